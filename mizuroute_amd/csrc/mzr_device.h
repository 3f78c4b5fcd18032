// Device-side view of one routing domain (gfx950).  Everything a kernel needs is in this POD,
// passed by value as kernel argument.  All per-reach arrays are in the library's INTERNAL reach
// order: reaches sorted by stage (= longest-path position counted from the outlet), and inside a
// stage in breadth-first order from the outlets, so that
//   * every stage is one contiguous index range  -> one coalesced launch per stage,
//   * the immediate upstreams of reach r are the contiguous range [upStart[r], upStart[r]+nUp[r])
//     in UREACHI order                             -> no index list, neighbouring lanes read
//                                                     neighbouring upstream rows.
// Ragged per-reach state of the one-lane-per-reach solvers (IRF convolution windows, sub-reach
// molecules) is stored "row-major by slot": element k of reach r lives at [k*N + r], so lanes of a
// wavefront that walk their rows in step touch consecutive addresses.  KWT particle rows are
// worked on by a group of lanes per reach and are contiguous per reach instead (MZR_KWI / MZR_OBI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MZR_MAXQPAR_DEV 20 // MAXQPAR, public_var.f90:36
#define MZR_KW_CAP   20   // at-rest particles per reach (MAXQPAR, public_var.f90:36)
#define MZR_OB_CAP   21   // outbox entries per reach: KWAVE(0:NR+1) + first non-routed
// The outbox of a reach is a ring over the time steps (slot = step mod MZR_OB_RING): step t of a reach overwrites what its
// downstream reach read in step t - MZR_OB_RING.  With two slots (rounds 1-3) a heavy reach r and its heavy downstream reach
// d were chained both ways -- d(t) needs r(t), r(t+2) needs d(t) -- so that two steps of r cost a pass of r AND a pass of d
// however fast r's own steps follow each other; with four (round 4), r runs up to four steps ahead and the loop no longer binds.
// (Round 5's flavour that visited a reach for four steps at a time needed a ring of eight; it was measured slower and is kept as a patch,
// tools/patches/r06_removed_flavours.patch.)
#ifndef MZR_OB_RING
#define MZR_OB_RING 4
#endif
static_assert((MZR_OB_RING & (MZR_OB_RING - 1)) == 0 && 21 + 5 * MZR_OB_RING <= 64, "outbox ring: a power of two whose particle counts fit the progress word");
// rows start on 64-byte sectors (24 doubles = 192 bytes per reach): a row of 20 / 21 doubles packed back to back straddles
// sector boundaries and every partial sector is fetched / written whole (profiles/r03a_summary.md)
#ifndef MZR_KW_STRIDE
#define MZR_KW_STRIDE 24
#endif
#ifndef MZR_OB_STRIDE
#define MZR_OB_STRIDE 24
#endif
#define MZR_KWI(k, r) ((size_t)(r) * MZR_KW_STRIDE + (k))   // particle k of reach r in kwTR, and PAIR index in kwQT
#define MZR_OBI(k, r) ((size_t)(r) * MZR_OB_STRIDE + (k))   // PAIR index of entry k of reach r in one parity of obQT
// Flow and time of a particle sit side by side (kwQT / obQT are rows of {Q, T} pairs, 16 bytes each): a lane moves a particle
// with ONE 16-byte access, and a list of n particles covers n * 16 contiguous bytes instead of two runs of n * 8 in two
// arrays -- fewer partly used 64-byte sectors, half the memory instructions.  Doubles inside the pair rows:
#define MZR_PQ(pairIndex) (2 * (size_t)(pairIndex))          // the flow
#define MZR_PT(pairIndex) (2 * (size_t)(pairIndex) + 1)      // the time (entry time TI in kwQT, exit time in obQT)
#define MZR_MAXUP    8    // immediate upstreams handled by the KWT merge
#define MZR_NMOL_KW  20   // init_model_data.f90:386-394
#define MZR_NMOL_MC  2
#define MZR_NMOL_DW  20
#define MZR_NLAKEPAR_DEV 56

// error record written by the first failing lane (atomicCAS on code).  The fields behind `where` are filled in by a
// persistent sweep that gives up waiting (code 93): which wait it was and what it saw, so that the host can tell a
// result that never became visible from one that was never produced (mzr_host.hip, stallReport).
#define MZR_BEAT 32   // ints per wavefront in swBeat
struct MzrErr {
  int code; int reach; int step; int where;
  int s;            // launch of the skewed schedule the waiting item belongs to
  int depReach;     // reach whose progress word the lane polled (internal index)
  int seen, need;   // the word it last saw, the step count it needs
  int queue, xcc;   // ticket queue the wavefront was serving / XCC it runs on
  int lane, nBad;   // first unsatisfied lane of the wavefront, number of unsatisfied lanes
  long long waited; // ticks of the 100 MHz clock since the polled words last changed
  int heads[8];     // ticket heads at the time
  int winSeq;                         // window (counted from the last synchronisation of the handle) whose kernel raised the error
  int nSlow; int raisedAt;            // (MZR_SWEEP_DEBUG) passes that took longer than 10 ms from the end of their wait to their publish; low word of the clock at the raise
  int slowT[32][24];                  // ... and (builds with -DMZR_SWEEP_TRACE) the clock at every section boundary of that pass, relative to the end of its wait
  int slow[32][8];                    // wavefront, launch, item, ticks, HW_ID, XCC, low word of the clock at the publish, -
};

// Static description of a reach that routes KWT particles, packed by the host into one 64-byte line
// (kwt_route.f90 reads the same values from NETOPO / RPARAM): everything the kernel would otherwise
// fetch through dependent loads.
struct MzrKwtRec {
  int r, sigma, u0;           // reach (internal index), its stage, first immediate upstream
  uint8_t nup;                // size(UREACHI)
  uint8_t flags;              // bits 0-3 count(goodBas), bit 6 an upstream reach is a lake, bit 7 outlet (DREACHK <= 0)
  uint8_t upGood;             // bit i: upstream i has upstream reaches of its own (publishes an outbox)
  uint8_t goodMask;           // bit i: goodBas(i+1)
  double width, CW, length;   // R_WIDTH, ALFA*K**(1/ALFA) with K = sqrt(R_SLOPE)/R_MAN_N (K itself: kwK[r], shock merges only), RLENGTH
  double scA, scB;            // R_WIDTH of the first / second non-headwater upstream over R_WIDTH (:929)
  int down, pad;              // downstream reach (internal index, -1 = outlet): whose progress the persistent sweep polls
};

#define MZR_KWT_HOLE (1 << 29)     // sigma of a record that stands for nobody: s - sigma < 0 in every launch, the lane group has no step (kwt_regroup)
static_assert(sizeof(MzrKwtRec) == 64, "MzrKwtRec is one 64-byte line");

// kwt traffic counters (particles), accumulated with wave-level reductions
struct MzrKwtStat { unsigned long long w_in, w_up, w_out, n_head, n_route, n_edges; };

struct MzrDev {
  int N, H;
  int W;                  // steps in the current window
  int stepBlock;          // steps a reach takes per launch of the Eulerian stage kernels (blocked time skew: launch s works on the
                          // steps [KB (s - stage), KB (s - stage + 1)) of every reach); 1 = one step per launch
  int nStages;            // longest path (reaches) in the domain
  // ---- topology (internal order)
  const int      *sigma;      // [N] stage of each reach (0 = farthest from the outlet)
  const int      *upStart;    // [N]
  const uint8_t  *nUp;        // [N] size(UREACHI)
  const uint8_t  *nGood;      // [N] count(goodBas)
  const uint32_t *goodMask;   // [N] bit i = goodBas(i+1)
  const uint8_t  *isOutlet;   // [N] DREACHK <= 0
  const int      *hruOff;     // [N+1]
  const int      *hruIdx;     // [nHru] 0-based index into a runoff row (caller HRU order)
  const double   *hruW;       // [nHru]
  // ---- parameters (RPARAM, dataTypes.f90:183-195)
  const double *slope, *mann, *width, *depth, *length, *storage, *side, *fldp, *basarea, *minflow;
  const double *chanTab;      // [10][N] what the Eulerian solvers derive from a reach's channel parameters alone (kernels_route.hip d_chan; null: computed per reach-step)
  const double *kwK, *kwCW;   // KWT: K = sqrt(slope)/n and ALFA*K**(1/ALFA), precomputed on the host
  // ---- configuration
  double dt, min_length_route, runoffMin, negRunoffTol, time_conv, length_conv, t_start;
  double mcTailTol;   // Muskingum-Cunge: closed-form tail of the sub-step sum once the outflow changes by less than this fraction of itself per sub-step (0 = iterate every sub-step)
  double T1_single;       // end of step for single-step windows (mzr_step passes TSEC(2) explicitly)
  int hw_drain_point, doesBasinRoute, is_flux_wm;
  const double *wm;           // [W][N] REACH_WM_FLUX of the window (null unless is_flux_wm)
  double *wmact;              // [N] REACH_WM_FLUX_actual of the method being launched
  // ---- hillslope
  int ntdhBas;
  const double *fracFuture;   // [ntdhBas]
  const double *fracPad;      // [32 + ntdhBas + 32] the same, zero-padded by the hillslope tile size on both sides
  const double *runoff;       // [W][H]
  double *qi;                 // [W][N] BASIN_QI per step
  double *qlat;               // [W+1][N] BASIN_QR(1); row 0 = value before the window
  const double *basS0;        // [ntdhBas][N] hillslope QFUTURE before the window
  double *basS1;              // [ntdhBas][N] ... after the window
  // ---- per-method flux rows
  double *Q;                  // [W][N] REACH_Q of the method being launched
  double *vol, *vol0, *inflow, *ele, *floodvol, *wb;   // [N] latest
  double *qsum;               // [N] running sum of REACH_Q (history mean)
  double *hInflow, *hEle, *hFlood;   // [N] running sums of REACH_INFLOW, REACH_ELE, FLOOD_VOL(1) (histVars_data.f90:229-246); null = not wanted
  // ---- IRF
  int maxtdh;
  const uint16_t *ntdh;       // [N]
  const double *uh;           // [maxtdh][N]
  double *irfQ;               // [maxtdh][N]
  // ---- KW / MC / DW molecules
  double *mol;                // [nMol][N]
  // One lane per reach, and a wavefront lasts as long as its slowest lane: Muskingum-Cunge reaches take 2 to 200 Courant sub-steps
  // (the same reaches step after step), the IRF convolution runs over 1 to maxtdh taps.  The reaches of every aligned block of
  // 256 are therefore dealt to the block's four wavefronts by that count (lanePerm: position -> reach, a permutation inside
  // each block, so every reach is still served exactly once and the block still touches the same 256-reach span of every array)
  unsigned short *mcSub;      // [N] Muskingum-Cunge sub-steps the reach executed in its last step (written by the kernel, read by the host now and then)
  const int *lanePerm;        // [ceil(N / 256) * 256 + nHeavyPos] or null: reach served by a lane position (-1 = none)
  // ... and the few reaches that take many times the usual trip count (Muskingum-Cunge: 4 to 200 sub-steps where the others take 2)
  // are taken out of their blocks altogether: they fill lane positions of their own behind the others (permN ..), heaviest first, and
  // the blocks over those positions are the FIRST of every launch -- the longest wavefronts of a launch start first and hold no
  // short ones hostage; their own positions hold -1
  int permN, nHeavyPos;       // first heavy lane position (= ceil(N / 256) * 256), number of heavy positions (a multiple of 256; 0 = none)
  // ---- KWT
  int    *kwN;                // [N] at-rest particle count (0 = not yet initialised)
  double *kwQT, *kwTR;        // [N][MZR_KW_STRIDE][2] {Q, TI} pairs; [N][MZR_KW_STRIDE] expected exit times (state only: written at the last step of a window)
  int    *obN;                // [MZR_OB_RING][N] routed-flag count of the outbox (NR+2)
  double *obQT;               // [MZR_OB_RING][N][MZR_OB_STRIDE][2] {Q, exit time} pairs
  const MzrKwtRec *kwtRouted;    // reaches that route particles (at most two upstream reaches), stage-major, class A: 16 lanes each
  const MzrKwtRec *kwtRoutedB;   // ... class B: reaches that lately needed at most 20 work-array entries, 8 lanes each (host regroups)
  const MzrKwtRec *kwtRoutedC;   // ... class C: at most 9 entries, 4 lanes each
  const MzrKwtRec *kwtGeneric;   // ... with more than two upstream reaches
  const int *kwtLight;        // headwater, lake and halo reaches, stage-major (one lane each)
  // ---- KWT persistent sweep (k_sweep_kwt): wavefronts draw items (blocks of reaches) of the skewed schedule in
  // launch order from per-XCD ticket counters and hand results on through kwDone
  unsigned long long *kwDone; // [N] steps of the current window a reach has completed (headwaters: W from the start), particle counts above them (kernels_kwt.hip, MZR_KWD_*)
  const int *down;            // [N] downstream reach (internal index), -1 = outlet
  const int *swItem;          // [nItems] stage-ordered; item = class << 28 | block index in the class list (0 A, 1 B, 2 generic, 3 lake / halo)
  const int *swLo, *swHi;     // [nItems] smallest / largest stage among the item's reaches
  const int *swRA;            // [nLaunch] first item that can be active in launch s (the last one follows from the tickets)
  const int *swP;             // [nLaunch+1][8] tickets of queue q before launch s (queue q = items i with i % 8 == q)
  int *swHead;                // [8][16] next ticket of each queue (one cache line each)
  const int *kwtHead;         // headwater reaches (bulk kernel before the sweep)
  const uint8_t *kwHeadFlag; double *kwHeadQ;      // round 6: k_hillslope_out writes a headwater reach's discharge rows of the KWT method itself (REACH_Q =
                                            // BASIN_QR(1), kwt_route.f90:181-205) where it has the value in a register; null: k_kwt_window_init copies the rows
  int nHead, nDepLight;       // entries of kwtHead / of kwtLight in persistent mode (lake and halo reaches only)
  int nA, nB, nC, nG;         // routed records per class
  // ---- lakes (null / 0 without lakes)
  const int *lakeSlot;        // [N] lake index of a lake reach, -1 otherwise
  const int *lakeModel;       // [nLake]
  const double *lakePar;      // [MZR_NLAKEPAR_DEV][nLake]
  double *lakeMut;            // [25][nLake] of the method being launched: I_months, D_months, E_rel_ini
  double *lakeRing;           // [nLake][12][lakeL] Hanasaki inflow memory of the method being launched
  int *lakeHead;              // [nLake][13] ring heads per month, [12] = initialised flag
  double *lakeRingD;          // [nLake][12][lakeLD] Hanasaki demand memory (lake_route.f90:288-331), same layout
  int *lakeHeadD;             // [nLake][13]
  const int *lakeTarg;        // [nLake] the lake follows a target volume (NETOPO%LakeTargVol, lake_route.f90:197-205); null = none does
  const double *lakeWmVol;    // [W][nLake] REACH_WM_VOL of the window
  int volJumpstart, lakeLD;   // is_vol_wm_jumpstart (lake_route.f90:140-142); ring length of the demand memory
  const double *lakeEvap, *lakePrecip;   // [W][nLake] m3/s
  const int *calMonth, *calDay, *calDoy; // [W]
  int nLake, LakeInputOption, calendarId, lakeL;
  long long iTime0;           // iTime of window step 0 is iTime0 + 1
  // ---- partition boundary (null / 0 in an unpartitioned domain)
  const int *haloSlot;        // [N] slot of a halo reach, -1 otherwise
  const int *exportSlot;      // [N] slot of an export reach, -1 otherwise
  int nHalo, nExp, Wmax;
  const double *imQ;          // [Wmax][nHalo] REACH_Q of the halo reaches (method being launched)
  const int    *imN;          // [Wmax][nHalo]
  const double *imOQ, *imOT;  // [Wmax][MZR_OB_CAP][nHalo]
  int    *exN;                // [Wmax][nExp]
  double *exOQ, *exOT;        // [Wmax][MZR_OB_CAP][nExp]
  // ---- constituent routing (tracer = T; main_route.f90:161-172,204-236,392-401, basinUH.f90:130-137, tracer.f90:43-207); null = off
  const double *solSrc;       // [W][H] basin constituent mass flux of the window
  double *solInst;            // [W][N] BASIN_solute_inst (hillslope routing on)
  double *basSol;             // [W+1][N] BASIN_solute, row t+1 = step t (as qlat)
  const double *solS0; double *solS1;   // [ntdhBas][N] solute_future before / after the window
  double *solFlux;            // [W][N] reach_solute_flux of the method being launched
  double *solMass;            // [N] reach_solute_mass(1)
  double *trVol0;             // [W][N] REACH_VOL(0) of every step (the solvers write it while routing); null = tracer off
  double time_conv_solute, mass_conv_solute;
  // ---- direct insertion of gauge observations (qmodOption = 1; main_route.f90:125-148, data_assimilation.f90:28-97); qmod = 0: off
  int qmod, qBlendPeriod, QerrTrend, nGauge;
  const int *gaugeFirst;      // [N] first gauge of the reach or -1
  const int *gaugeNext;       // [nGauge] next gauge of the same reach or -1 (a later gauge overrides an earlier one)
  const int *obsHave;         // [W] there is an observation time at this step
  const double *obsVal;       // [W][nGauge]
  double *qobs, *qerr;        // [N] RCHFLX%Qobs, ROUTE%Qerror (of the method being launched; Qobs / Qelapsed evolve alike in every method)
  int *qelapsed;              // [N] RCHFLX%Qelapsed
  // ---- persistent sweep of the Eulerian methods (k_sweep_route): items = up to 64 reaches of one stage, drawn in launch
  // order from per-XCD ticket counters like the KWT sweep's; rtDone[r] = steps of the window reach r has completed
  int *rtDone;                // [N] (of the method being launched)
  const int *rtItemR;         // [nItems][64] reach of every lane, -1 = none
  const int *rtItemInfo;      // [nItems] stage | 1 << 30 when the item holds lakes (their plain state needs fences)
  const int *rtRA, *rtP;      // per launch: first active item, ticket prefix per queue (as swRA / swP)
  int *rtHead;                // [8][16] ticket counters
  unsigned long long *swClock;   // [2] of this launch of the KWT sweep: the 100 MHz clock when its first wavefront arrived / when its last one left (mzr_get_sweep_clock)
  int *swBeat;                // [wavefronts][8] what every wavefront of a persistent sweep is doing (launch, item, queue, phase, items done): only with MZR_SWEEP_DEBUG=1
  int winSeq;                 // this window's number since the handle was last synchronised (goes into the error record)
  int sweepPrio;              // 1: wavefronts of this handle's persistent sweeps keep the highest wave priority (mzr_config.sweepPriority)
  int sweepAlways;            // wavefronts of a sweep launch that join however late they start (64; the whole grid for the small partner of two sweeps on one device)
  long long stallTicks;       // a polling wavefront gives up (code 93) when nothing it polls has changed for this many ticks of the 100 MHz clock
  MzrKwtStat *kwtStat;
  unsigned long long *dbgCycles;   // [32] per-section wave cycles (only with -DMZR_KWT_TIMING)
  MzrErr *err;
};

__device__ __forceinline__ void mzr_raise(const MzrDev &d, int code, int reach, int step, int where) {
  if (atomicCAS(&d.err->code, 0, code) == 0) {
    d.err->reach = reach; d.err->step = step; d.err->where = where; d.err->winSeq = d.winSeq;
  }
}
// a wavefront of a persistent sweep gives up waiting: the record says what it waited for (one lane calls this)
__device__ __noinline__ void mzr_raise_stall(const MzrDev &d, int where, int reach, int s, int depReach, int seen, int need, int queue,
                                             int lane, int nBad, long long waited, const int *heads) {
  if (atomicCAS(&d.err->code, 0, 93) == 0) {
    MzrErr *e = d.err;
    e->reach = reach; e->step = -1; e->where = where; e->s = s; e->depReach = depReach; e->seen = seen; e->need = need;
    e->queue = queue; e->xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7; e->lane = lane; e->nBad = nBad; e->waited = waited;
    e->raisedAt = (int)wall_clock64(); e->winSeq = d.winSeq;
    for (int q = 0; q < 8; ++q) e->heads[q] = heads ? __hip_atomic_load(heads + q * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
  }
}

// Census mode of a persistent sweep kernel: every wavefront counts itself in (cnt[0]), notes the highest count seen
// (cnt[1]), stays for 300 us and counts itself out; workgroups the device cannot hold start after others have left.
__device__ __forceinline__ void mzr_census(int *cnt) {
  if ((threadIdx.x & 63) == 0) {
    const int n = atomicAdd(cnt, 1) + 1;
    atomicMax(cnt + 1, n);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 30000) __builtin_amdgcn_s_sleep(32);
    atomicSub(cnt, 1);
  }
}

// Start of a wavefront of a persistent sweep (whole wavefront calls; head = the sweep's ticket heads, whose words 8*16+2 ..
// hold the launch's arrival count, the number of wavefronts that joined, the time of the first arrival (two words) and,
// from 8*16+16 on, a histogram of the start delays).  Returns the wavefront's number among those that joined, or -1 when it started late
// -- more than MZR_SWEEP_LATE_TICKS (100 MHz) after the first wavefront of the launch, because its workgroup had to wait
// for a slot (the grid did not fit the device, or another kernel held the slot) -- and must not join: the tickets do not
// need it, and a persistent kernel with workgroups launched behind time was measured (profiles/r03_soak.md) to freeze, now
// and then, the memory instructions of exactly those wavefronts for as long as the others keep running.
#ifndef MZR_SWEEP_LATE_TICKS
#define MZR_SWEEP_LATE_TICKS 2000
#endif
__device__ __forceinline__ int mzr_sweep_join(int *head, unsigned long long *clk = nullptr, int always = 64) {
  int j = 0;
  if ((threadIdx.x & 63) == 0) {
    const long long now = wall_clock64();
    const unsigned long long t0 = atomicCAS((unsigned long long *)(head + 8 * 16 + 4), 0ull, (unsigned long long)now);
    if (clk && t0 == 0ull) clk[0] = (unsigned long long)now;      // the launch's first wavefront
    const int arr = atomicAdd(head + 8 * 16 + 2, 1);
    const long long dt = t0 ? now - (long long)t0 : 0;
    atomicAdd(head + 8 * 16 + 16 + (dt <= 0 ? 0 : min(31, 64 - __clzll(dt))), 1);      // delays below 2^k ticks
    // (the first 64 to arrive always join -- a launch on a GPU that has just woken up can be slow as a whole -- so that
    // every queue has servers whatever happens: joiners 0..63 take queue j % 8)
    if (dt > MZR_SWEEP_LATE_TICKS && arr >= (always > 64 ? always : 64)) j = -1;
    else j = atomicAdd(head + 8 * 16 + 3, 1);
  }
  return __builtin_amdgcn_readfirstlane(j);
}
// before every launch of a sweep: arrival counters and start time back to zero (threads 8..11 of the heads kernel)
__device__ __forceinline__ void mzr_sweep_join_reset(int *head) {
  if (threadIdx.x >= 8 && threadIdx.x < 12) head[8 * 16 + 2 + (threadIdx.x - 8)] = 0;
}

// Accesses to data that another wavefront of the SAME launch produces or consumes (persistent sweep):
// relaxed agent-scope atomics = global_load / global_store ... sc1, which bypass the CU's L1 and are
// coherent across the per-XCD L2s (MI355X_MICROARCH.md, inter-workgroup visibility).  P = false: plain.
// The hand-off is: sc1 payload stores -> asm volatile "s_waitcnt vmcnt(0)" (with a memory clobber: also a compiler barrier)
// -> sc1 store of the progress word; and on the other side sc1 poll of the word -> compiler barrier -> sc1 loads of the
// payload.  That the drained stores are visible before the word relies on gfx9's single vmcnt for loads and stores and on
// in-order return of loads; it is written for gfx950 and nothing else:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mizuroute_amd device code is written for gfx950 (CDNA4) only"
#endif
template <bool P> __device__ __forceinline__ double ldx(const double *p) {
  if (P) return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  return *p;
}
template <bool P> __device__ __forceinline__ int ldx(const int *p) {
  if (P) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool P> __device__ __forceinline__ unsigned long long ldx(const unsigned long long *p) {
  if (P) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool P> __device__ __forceinline__ void stx(unsigned long long *p, unsigned long long v) {
  if (P) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool P> __device__ __forceinline__ void stx(double *p, double v) {
  if (P) __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// 16-byte accesses to the pair rows.  P: other wavefronts of the launch read / wrote them: buffer_load / buffer_store_dwordx4 sc1
// through a raw buffer descriptor over the whole array (offsets are bytes, 32 bits: arrays of up to 4 GB, i.e. 11 M reaches)
typedef double mzr_d2 __attribute__((ext_vector_type(2)));
typedef int mzr_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mzr_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
template <bool P> __device__ __forceinline__ mzr_d2 ldq(__amdgpu_buffer_rsrc_t rs, const double *base, size_t pairIndex) {
  if (P) return __builtin_bit_cast(mzr_d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(unsigned)(pairIndex * 16), 0, 16));
  return *(const mzr_d2 *)(base + 2 * pairIndex);
}
template <bool P> __device__ __forceinline__ void stq(__amdgpu_buffer_rsrc_t rs, double *base, size_t pairIndex, double a, double b) {
  mzr_d2 v; v.x = a; v.y = b;
  if (P) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mzr_i4, v), rs, (int)(unsigned)(pairIndex * 16), 0, 16);
  else *(mzr_d2 *)(base + 2 * pairIndex) = v;
}
template <bool P> __device__ __forceinline__ void stx(int *p, int v) {
  if (P) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
