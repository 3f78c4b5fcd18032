// Lagrangian kinematic-wave tracking (KWT) stage kernel for gfx950: a GROUP of G lanes per routed
// reach, one launch per stage of the time-skewed level sweep (see kernels_route.hip for the
// schedule).
//
// Replaces kwt_rch and its helpers, route/build/src/kwt_route.f90:
//   kwt_rch 36-346, getusq_rch 461-613, qexmul_rch 619-993, remove_rch 999-1123,
//   kinwav_rch 1130-1439 (+ rUpdate 1409-1437), interp_rch 1444-1622.
//
// Data-flow redesign (results are unchanged):
//  * The reference lets the DOWNSTREAM reach strip the routed particles out of its upstream
//    reach's list (kwt_route.f90:822-848).  The stripped list is a pure function of the upstream
//    reach's own result -- KWAVE(NR+1:NQ2+1), the same slice an outlet keeps for itself
//    (:325-344) -- so here every reach stores that at-rest slice itself (kwN/kwQT/kwTR,
//    <= 20 particles) and publishes what its downstream reach needs, KWAVE(0:NR+1) plus the first
//    non-routed particle, flow and exit time only, in a per-reach OUTBOX (obN/obQT).  No lane
//    ever writes another reach's state, and the outbox is double-buffered on the parity of the
//    time step so that reach u may already work on step t+1 while its downstream reach consumes
//    step t in the same launch.
//  * Expected exit times of a reach's own waiting particles are recomputed by kinwav every step,
//    so only TR of element 0 is read back; the others are written for restart files only.
//
// Execution model.  The reference algorithm is a chain of short sequential passes over a list of
// 10..60 particles per reach.  One lane per reach leaves a wavefront waiting on its slowest lane
// (the reach that thins 40 particles down to 20 while its neighbours hold 5), so a reach is worked
// on by G = 8 adjacent lanes instead and every pass is rewritten as a data-parallel step over the
// particles, with DPP reductions inside the group where the reference takes a minimum:
//    merge     each upstream particle finds its rank in the other tributary's list and
//              interpolates that tributary at its own time (qexmul_rch's k-way cursor walk);
//    remove    interpolation errors for all particles at once, group arg-min per removal, only the
//              two neighbours are re-evaluated (on two lanes);
//    kinwav    celerities (the x**0.4 of every particle) in parallel, shock search = group arg-min
//              of the pairwise crossing points, the particle list after routing is rebuilt with a
//              bit-mask prefix count instead of the running ICOUNT;
//    interp    time-step average: short serial sum (same order as the reference).
// Arithmetic per particle is the reference's statement by statement; where the order of a sequence
// of operations is observable (ties of MINLOC, summation order) it is kept.  Inputs for which the
// parallel form is not equivalent (duplicate times across tributaries, unordered times, more
// than two upstream reaches, more than 64 particles before thinning) take a serial path on lane
// 0 of the group that follows the reference loop literally.
// Headwater, lake and halo reaches are O(1) and take one lane each in the trailing blocks of the
// same launch (the host lists routed and light reaches separately, stage-major).
//
// Layout: particle rows are contiguous per reach, {flow, time} pairs of 16 bytes (kwQT[r][24][2], obQT[parity][r][24][2],
// rows on 64-byte sectors), so that the G lanes of a group move a list with one 16-byte access per lane; work arrays live in LDS, carved
// among the groups of a wavefront by need.
// Bound by HBM traffic of the particle rows: see DESIGN.md for the bytes-per-reach-step model.
#include <float.h>
#include <algorithm>
#include <mutex>
#include <hip/hip_ext.h>
#include "mzr_device.h"
#include "lake_device.h"
#include "mzr_math.h"

namespace {

__device__ __forceinline__ double interp3(double T0, double Q1, double Q2, double T1, double T2) {
  return Q1 + ((Q2 - Q1) / (T2 - T1)) * (T0 - T1);   // kwt_route.f90:1115-1121
}

// interp_rch with two output times (one averaging interval), kwt_route.f90:1444-1622.
// TOLD/QOLD hold NOLD points, addressed 1-based through T()/Q().
__device__ int d_interp_rch(const double *TOLD, const double *QOLD, int NOLD, double T0, double T1, double *QNEW) {
#define T(i) TOLD[(i) - 1]
#define Q(i) QOLD[(i) - 1]
  if (T(1) > T0 || T(NOLD) < T1) return 1;
  int IBEG = 1, IEND = 1;
  for (int i = 2; i <= NOLD; ++i) if (T0 <= T(i)) { IBEG = i; break; }
  for (int i = 1; i <= NOLD; ++i) if (T1 <= T(i)) { IEND = i; break; }
  double AREAB = 0.0, AREAE = 0.0, AREAM = 0.0;
  if (T1 < T(IBEG)) {
    const double SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    const double QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    const double QEST1 = SLOPE * (T1 - T(IBEG - 1)) + Q(IBEG - 1);
    *QNEW = 0.5 * (QEST0 + QEST1);
    return 0;
  }
  if (T0 < T(IBEG)) {
    const double SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    const double QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    AREAB = (T(IBEG) - T0) * 0.5 * (QEST0 + Q(IBEG));
  }
  if (T1 < T(IEND)) {
    const double SLOPE = (Q(IEND) - Q(IEND - 1)) / (T(IEND) - T(IEND - 1));
    const double QEST1 = SLOPE * (T1 - T(IEND - 1)) + Q(IEND - 1);
    AREAE = (T1 - T(IEND - 1)) * 0.5 * (Q(IEND - 1) + QEST1);
  }
  if (IBEG < IEND) {
    for (int IMID = IBEG + 1; IMID <= IEND; ++IMID) {
      if (IMID < IEND || (IMID == IEND && T1 == T(IEND) && T0 < T(IEND - 1)))
        AREAM = AREAM + (T(IMID) - T(IMID - 1)) * 0.5 * (Q(IMID - 1) + Q(IMID));
    }
  }
#undef T
#undef Q
  *QNEW = (AREAB + AREAE + AREAM) / (T1 - T0);
  return 0;
}


__device__ __forceinline__ int mzr_lane();
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Persistent sweep: wait until every lane's dependency counter (steps completed by an upstream or the
// downstream reach in this window) has reached the value the lane needs.  Whole wavefront; true = give
// up (another wavefront raised an error, or nothing moved for seconds: code 93 instead of a hung GPU).
// Progress word of a reach (kwDone): steps of the window it has completed in the low 16 bits; above them what its
// consumers would otherwise have to load before they know how much of its rows is alive -- the particle counts of its
// two outbox parities (5 bits each) and of its at-rest list (5 bits).
// debugging aid (MZR_SWEEP_DEBUG=1): slot `k` of this wavefront's record of what it is doing
__device__ __forceinline__ void kwt_beat(const MzrDev &d, int k, int v) {
#ifdef MZR_SWEEP_TRACE
  if (d.swBeat && (threadIdx.x & 63) == 0) __hip_atomic_store(d.swBeat + (size_t)blockIdx.x * MZR_BEAT + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
#ifdef MZR_SWEEP_TRACE
#define MZR_BEAT_ON(d) ((d).swBeat != nullptr)
#else
#define MZR_BEAT_ON(d) false
#endif
// (64-bit words: steps in bits 0-15, the at-rest count in 16-20, the counts of the MZR_OB_RING outbox slots from bit 21 on, five bits each)
#define MZR_KWD_STEPS(w) ((int)((w) & 0xffffull))
#define MZR_KWD_OWN(w) ((int)(((w) >> 16) & 31ull))
#define MZR_KWD_OUT(w, slot) ((int)(((w) >> (21 + 5 * (slot))) & 31ull))
#define MZR_KWD_OUTMASK(slot) (31ull << (21 + 5 * (slot)))
#define MZR_KWD_ALLOUT (((MZR_OB_RING * 5 + 21) >= 64 ? ~0ull : ((1ull << (MZR_OB_RING * 5 + 21)) - 1ull)) & ~((1ull << 21) - 1ull))      // the counts of every slot
typedef unsigned long long mzr_word;
__device__ __forceinline__ bool kwt_wait_deps(const MzrDev &d, const mzr_word *wp, int wneed, mzr_word *word = nullptr, int s = -1, int reach = -1) {
  long long t0 = 0;
  int spins = 0, vlast = 0;
  for (;;) {
    mzr_word w = (mzr_word)(unsigned)wneed;
    if (wp) w = ldx<true>(wp);
    if (word) *word = w;
    const int v = MZR_KWD_STEPS(w);
    if (__ballot(v < wneed) == 0ull) {
#ifdef MZR_KWT_TIMING
      if ((blockIdx.x & 15) == 0 && (threadIdx.x & 63) == 0) { atomicAdd(&d.dbgCycles[(((blockIdx.x >> 4) & 1023) << 5) + 23], spins ? 1ull : 0ull); atomicAdd(&d.dbgCycles[(((blockIdx.x >> 4) & 1023) << 5) + 24], (unsigned long long)spins); }
#endif
      asm volatile("" ::: "memory");      // what the progress words guard is read after them (compiler order; the loads are sc1)
      kwt_beat(d, 3, 3);
      if (MZR_BEAT_ON(d)) kwt_beat(d, 7, (int)wall_clock64());
      return false;
    }
#ifndef MZR_KWT_SLEEP
#define MZR_KWT_SLEEP 4
#endif
    // How soon to look again depends on how far the slowest dependency is behind: one step short, it can be there any
    // moment (the hand-off is on the window's critical path: poll at once); d >= 2 steps short, the reach still has d - 1
    // whole passes of >= 8 us each in front of it, and the wavefront -- one that drew a ticket launches ahead of the
    // frontier, as most of the waiting ones have -- sleeps through part of that.  Polling them all at the fast rate
    // (4 000 wavefronts x 64 lanes every microsecond) is what starved the passes at the frontier of their own memory
    // accesses for seconds at a time (profiles/r03_soak.md).
    if (__ballot(v + 1 < wneed) == 0ull) __builtin_amdgcn_s_sleep(MZR_KWT_SLEEP);
    else {
      const int n = __ballot(v + 8 < wneed) != 0ull ? 8 : __ballot(v + 4 < wneed) != 0ull ? 4 : __ballot(v + 2 < wneed) != 0ull ? 2 : 1;
      for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(127);      // 127 x 64 clocks = 3.4 us each
    }
    if ((++spins & 31) == 0) {
      if (ldx<true>(&d.err->code) != 0) return true;
      if (spins == 32) kwt_beat(d, 3, 2);
      const long long now = wall_clock64();     // 100 MHz
      if (!t0 || __ballot(v != vlast) != 0ull) t0 = now;      // something this wavefront polls has moved: not stuck
      else if (now - t0 > d.stallTicks) {
        const unsigned long long bad = __ballot(v < wneed);
        const int first = __ffsll((long long)bad) - 1;
        if (mzr_lane() == first)
          mzr_raise_stall(d, 20, reach, s, wp ? (int)(wp - d.kwDone) : -1, (int)(w & 0xffffffffull), wneed, -1, first, __popcll(bad), now - t0, d.swHead);
        return true;
      }
      vlast = v;
    }
  }
}

}  // namespace


// ---- group primitives: G adjacent lanes (G = 4, 8 or 16, aligned) cooperate on one reach ----------
namespace {

// Lane index through an opaque instruction pair: inside the persistent item loop everything derived
// from threadIdx would otherwise count as loop-invariant, be computed once up front and spilled.
__device__ __forceinline__ int mzr_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int CTRL> __device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Butterfly over the group: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror and
// row_mirror (once both quads / both halves agree, the mirrored lane holds the other side's value).
#define MZR_DPP_XOR1 0xB1
#define MZR_DPP_XOR2 0x4E
#define MZR_DPP_HALF_MIRROR 0x141
#define MZR_DPP_MIRROR 0x140

// G = 64 (one reach per wavefront): rows of 16 lanes are reduced with DPP, the four row results are
// combined through v_readlane, and everything "uniform" lives in scalar registers.
template <int G> __device__ __forceinline__ int uni(int v) { return G == 64 ? __builtin_amdgcn_readfirstlane(v) : v; }
template <int G> __device__ __forceinline__ double uni(double v) {
  if (G != 64) return v;
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// lanes i <-> i^16 inside each half of the wavefront (ds_swizzle bit mode: and 0x1f, xor 0x10)
__device__ __forceinline__ int swz16_i(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x401F); }
__device__ __forceinline__ double swz16_d(double v) { return __hiloint2double(swz16_i(__double2hiint(v)), swz16_i(__double2loint(v))); }
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int G, class F> __device__ __forceinline__ double grp_reduce_d(double v, F op) {
  v = op(v, dpp_d<MZR_DPP_XOR1>(v));
  v = op(v, dpp_d<MZR_DPP_XOR2>(v));
  if (G >= 8) v = op(v, dpp_d<MZR_DPP_HALF_MIRROR>(v));
  if (G >= 16) v = op(v, dpp_d<MZR_DPP_MIRROR>(v));
  if (G == 32) v = op(v, swz16_d(v));
  if (G == 64) v = op(op(readlane_d(v, 0), readlane_d(v, 16)), op(readlane_d(v, 32), readlane_d(v, 48)));
  return v;
}
template <int G> __device__ __forceinline__ double grp_min(double v) { return grp_reduce_d<G>(v, [](double a, double b) { return b < a ? b : a; }); }
// number of lanes of the group whose flag is set
template <int G> __device__ __forceinline__ int grp_count(bool p) {
  const unsigned long long b = __ballot(p);
  if (G == 64) return __popcll(b);
  const int gbase = mzr_lane() & ~(G - 1);
  return __popcll((b >> gbase) & ((1ull << (G & 63)) - 1ull));
}
template <int G> __device__ __forceinline__ bool grp_any(bool p) {
  const unsigned long long b = __ballot(p);
  if (G == 64) return b != 0ull;
  const int gbase = mzr_lane() & ~(G - 1);
  return ((b >> gbase) & ((1ull << (G & 63)) - 1ull)) != 0ull;
}
template <int G, bool MAXI> __device__ __forceinline__ int grp_minmax_i(int v) {
  auto op = [](int a, int b) { return MAXI ? (b > a ? b : a) : (b < a ? b : a); };
  v = op(v, dpp_i<MZR_DPP_XOR1>(v));
  v = op(v, dpp_i<MZR_DPP_XOR2>(v));
  if (G >= 8) v = op(v, dpp_i<MZR_DPP_HALF_MIRROR>(v));
  if (G >= 16) v = op(v, dpp_i<MZR_DPP_MIRROR>(v));
  if (G == 32) v = op(v, swz16_i(v));
  if (G == 64) v = op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
  return v;
}
// arg-min over the group of (v, i): the minimum value first, then among the lanes that hold it the
// smallest index (LAST = false, MINLOC) or the largest (LAST = true).  i >= 0.
template <int G, bool LAST> __device__ __forceinline__ void grp_argmin(double &v, int &i) {
  const double m = grp_min<G>(v);
  const int c = (v == m) ? i : (LAST ? -1 : 0x7fffffff);
  i = grp_minmax_i<G, LAST>(c);
  v = m;
}
// DPP move whose "old" operand is undefined (every lane has a source for the controls used here): the compiler folds
// it into the consuming VALU instruction (v_min_u32_dpp) instead of copying the register first
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int G> __device__ __forceinline__ unsigned grp_min_u(unsigned v) {
  static_assert(G <= 16, "one DPP row");
  v = min(v, dpp_u<MZR_DPP_XOR1>(v));
  v = min(v, dpp_u<MZR_DPP_XOR2>(v));
  if (G >= 8) v = min(v, dpp_u<MZR_DPP_HALF_MIRROR>(v));
  if (G >= 16) v = min(v, dpp_u<MZR_DPP_MIRROR>(v));
  return v;
}
// MINLOC over the group of (v, i) for v >= 0 and never NaN, i >= 1 (0 = "none"): non-negative doubles order like their
// bit patterns, so the minimum is found on the high words, then the low words of the lanes that tie, then the
// smallest index of the lanes that still tie -- twelve one-instruction DPP steps instead of 64-bit compares and selects.
template <int G> __device__ __forceinline__ int grp_argmin_pos(double v, int i) {
  const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
  const unsigned mh = grp_min_u<G>(hi);
  const unsigned ml = grp_min_u<G>(hi == mh ? lo : 0xffffffffu);
  return (int)grp_min_u<G>((hi == mh && lo == ml) ? (unsigned)i : 0x7fffffffu);
}
// flags of the group's lanes as a bit mask (bit 0 = first lane of the group)
template <int G> __device__ __forceinline__ unsigned long long grp_bits(bool p) {
  const unsigned long long b = __ballot(p);
  if (G == 64) return b;
  const int gbase = mzr_lane() & ~(G - 1);
  return (b >> gbase) & ((1ull << (G & 63)) - 1ull);
}
// value held by the first lane of the group
template <int G> __device__ __forceinline__ int grp_first(int v) {
  if (G == 64) return __builtin_amdgcn_readfirstlane(v);
  return __shfl(v, mzr_lane() & ~(G - 1), 64);
}

// Values read from LDS for ALL slots of a lane before any of them is used.  Written as `if (in range && arr[i] < x)` per slot the
// compiler puts every slot's read inside that slot's own divergent region, one LDS round trip (~130 cycles under the sweep's load) after
// the other; read with a clamped index and held here, the slots' reads are in flight together and the tests follow.
__device__ __forceinline__ void lds_held(double &v) { asm volatile("" : "+v"(v)); }
template <int N> __device__ __forceinline__ void lds_held(double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) asm volatile("" : "+v"(v[k]));
}

// LDS traffic of one group is ordered by the hardware (one wavefront, in-order LDS queue); this
// only keeps the compiler from moving accesses across a phase boundary.
__device__ __forceinline__ void grp_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// qexmul_rch (:619-993) for the binary confluence, literal cursor walk over the STAGED upstream
// series (LDS copies of the outbox rows, index 0 = last particle routed before this step, 1..nr-2
// = particles routed in this step, nr-1 = end-of-step particle at T1).  Serial path of the group
// kernel: taken when the parallel rank/interpolate form is not equivalent (duplicate or unordered
// times) and to produce the reference's error codes.
// Every upstream also contributes its hillslope series {BASIN_QR(0)@T0, BASIN_QR(1)@T1}: a
// straight line whose only own particle sits at T1 and is the LAST thing merged (MINLOC ties go to
// the lowest series index, and the basin series come first).
struct KwtBasin { double b0q0, b0q1, b0sl, b1q0, b1sl, bsc; };

__device__ __forceinline__ int kwt_merge_binary_serial(int nup, int ns, int nrA, int nrB, const double *SAq, const double *SAt,
                                                    const double *SBq, const double *SBt, double scA, double scB,
                                                    const KwtBasin &bs, double T0, double T1, double *QD, double *TD) {
  int kA = 1, kB = 1, IPRT = 0;
  double TIME_LAST = -DBL_MAX;
  for (;;) {
    const double cA = (ns > 0 && kA <= nrA - 2) ? SAt[kA] : DBL_MAX;
    const double cB = (ns > 1 && kB <= nrB - 2) ? SBt[kB] : DBL_MAX;
    if (cA == DBL_MAX && cB == DBL_MAX) break;
    const bool pickA = cA <= cB;               // MINLOC: ties -> lower series index
    const double CT = pickA ? cA : cB;
    if (!(CT < T1)) return -40;                // a routed particle leaves before the end of the step
    if (CT < TIME_LAST) return -30;
    if (CT != TIME_LAST) {
      double Q_AGG = 0.0;
      Q_AGG = Q_AGG + (bs.b0q0 + bs.b0sl * (CT - T0)) * bs.bsc;
      if (nup > 1) Q_AGG = Q_AGG + (bs.b1q0 + bs.b1sl * (CT - T0)) * bs.bsc;
      {
        double SFLOW;
        if (pickA) SFLOW = SAq[kA] * scA;
        else {
          const double tb = SAt[kA - 1], te = SAt[kA], qb = SAq[kA - 1], qe = SAq[kA];
          if (te < CT || tb > CT) return -40;
          const double SLOPE = (qe - qb) / (te - tb);
          SFLOW = (qb + SLOPE * (CT - tb)) * scA;
        }
        Q_AGG = Q_AGG + SFLOW;
      }
      if (ns > 1) {
        double SFLOW;
        if (!pickA) SFLOW = SBq[kB] * scB;
        else {
          const double tb = SBt[kB - 1], te = SBt[kB], qb = SBq[kB - 1], qe = SBq[kB];
          if (te < CT || tb > CT) return -40;
          const double SLOPE = (qe - qb) / (te - tb);
          SFLOW = (qb + SLOPE * (CT - tb)) * scB;
        }
        Q_AGG = Q_AGG + SFLOW;
      }
      QD[IPRT] = Q_AGG; TD[IPRT] = CT; TIME_LAST = CT; ++IPRT;
    }
    if (pickA) ++kA; else ++kB;
  }
  {   // the particle at T1, led by the first basin series
    const double CT = T1;
    double Q_AGG = 0.0;
    Q_AGG = Q_AGG + bs.b0q1 * bs.bsc;
    if (nup > 1) Q_AGG = Q_AGG + (bs.b1q0 + bs.b1sl * (CT - T0)) * bs.bsc;
    if (ns > 0) {
      const double tb = SAt[kA - 1], te = SAt[kA], qb = SAq[kA - 1], qe = SAq[kA];
      if (te < CT || tb > CT) return -40;
      const double SLOPE = (qe - qb) / (te - tb);
      Q_AGG = Q_AGG + (qb + SLOPE * (CT - tb)) * scA;
    }
    if (ns > 1) {
      const double tb = SBt[kB - 1], te = SBt[kB], qb = SBq[kB - 1], qe = SBq[kB];
      if (te < CT || tb > CT) return -40;
      const double SLOPE = (qe - qb) / (te - tb);
      Q_AGG = Q_AGG + (qb + SLOPE * (CT - tb)) * scB;
    }
    QD[IPRT] = Q_AGG; TD[IPRT] = CT; ++IPRT;
  }
  return IPRT;
}

// Confluences of more than two reaches are rare: the reference's k-way merge runs on lane 0 of the
// group, with the series cursors in the two work arrays the merge does not need (scrD: 32 doubles,
// scrI: 64 ints of LDS) and every particle fetched from the outbox on demand.
template <bool PERS>
__device__ __forceinline__ int kwt_merge_generic(int nup, int u0, int NUPS, double RW, double T0, double T1,
                                                 const uint8_t *nGood, const double *width, const double *qlat_prev,
                                                 const double *qlat_cur, const int *obN, const double *obQT,
                                                 int N, double *QD, double *TD, int IMAX,
                                                 double *scrD, int *scrI) {
  int *su = scrI, *slen = scrI + 2 * MZR_MAXUP, *snr = scrI + 4 * MZR_MAXUP, *ITIM = scrI + 6 * MZR_MAXUP;
  double *sc = scrD, *CTIME = scrD + 2 * MZR_MAXUP;
  int IUPR = 0;
#pragma unroll 1
  for (int i = 0; i < nup; ++i) { su[i] = u0 + i; slen[i] = 2; snr[i] = 2; sc[i] = 1.0 / RW; ITIM[i] = 1; CTIME[i] = T1; }
#pragma unroll 1
  for (int i = 0; i < nup; ++i) {
    const int u = u0 + i;
    if (nGood[u] > 0) {
      const int si = nup + IUPR; ++IUPR;
      const int nr = ldx<PERS>(obN + u);
      su[si] = u; snr[si] = nr; slen[si] = nr + 1; sc[si] = width[u] / RW; ITIM[si] = 1; CTIME[si] = ldx<PERS>(obQT + MZR_PT(MZR_OBI(1, u)));
    }
  }
  auto sQ = [&](int i, int k) -> double { return i < nup ? (k == 0 ? qlat_prev[su[i]] : qlat_cur[su[i]]) : ldx<PERS>(obQT + MZR_PQ(MZR_OBI(k, su[i]))); };
  auto sT = [&](int i, int k) -> double { return i < nup ? (k == 0 ? T0 : T1) : ldx<PERS>(obQT + MZR_PT(MZR_OBI(k, su[i]))); };
  unsigned done = 0;
  const unsigned all = (1u << NUPS) - 1u;
  int IPRT = 0, JUPS_OLD = 0x7fffffff, ITIM_OLD = 0x7fffffff;
  double TIME_LAST = -DBL_MAX;
#pragma unroll 1
  for (;;) {
    int JUPS = 0;
#pragma unroll 1
    for (int i = 1; i < NUPS; ++i) if (CTIME[i] < CTIME[JUPS]) JUPS = i;
    if (JUPS == JUPS_OLD && ITIM[JUPS] == ITIM_OLD) return -20;
    JUPS_OLD = JUPS; ITIM_OLD = ITIM[JUPS];
    if (!((done >> JUPS) & 1u)) {
      const int kj = ITIM[JUPS];
      if (kj >= snr[JUPS]) { done |= 1u << JUPS; CTIME[JUPS] = DBL_MAX; }
      else {
        const double CT = CTIME[JUPS];
        const double TIME_OLD = IPRT >= 1 ? TIME_LAST : -DBL_MAX;
        if (CT < TIME_OLD) return -30;
        if (CT != TIME_OLD) {
          double Q_AGG = 0.0;
#pragma unroll 1
          for (int i = 0; i < NUPS; ++i) {
            const int IWAV = ITIM[i];
            double SFLOW;
            if (i == JUPS) SFLOW = sQ(i, IWAV) * sc[i];
            else {
              int IBEG = IWAV;
              if (sT(i, IBEG) >= CT) IBEG = IWAV - 1;
              const int IEND = IBEG + 1;
              if (IEND >= slen[i] || IBEG < 0) return -40;
              const double tb = sT(i, IBEG), te = sT(i, IEND);
              if (te < CT || tb > CT) return -40;
              const double qb = sQ(i, IBEG), qe = sQ(i, IEND);
              const double SLOPE = (qe - qb) / (te - tb);
              SFLOW = (qb + SLOPE * (CT - tb)) * sc[i];
            }
            Q_AGG = Q_AGG + SFLOW;
          }
          if (IPRT >= IMAX) return -60;
          QD[IPRT] = Q_AGG; TD[IPRT] = CT; TIME_LAST = CT; ++IPRT;
        }
        if (kj == slen[JUPS] - 1) { done |= 1u << JUPS; CTIME[JUPS] = DBL_MAX; }
        else { ITIM[JUPS] = kj + 1; CTIME[JUPS] = sT(JUPS, kj + 1); }
      }
    }
    if (done == all) break;
  }
  return IPRT;
}

// interp_rch (:1444-1622) by a group of lanes, one averaging interval.  The two index searches are
// ballots, every trapezoid is evaluated by the lane of its right end point, and only the running
// sum (same order as the reference) is sequential.  TERM is scratch for NOLD values.
template <int G, int KS>
__device__ __forceinline__ int grp_interp_rch(const double *TOLD, const double *QOLD, double *TERM, int NOLD, double T0, double T1,
                                              int gl, double *QNEW) {
#define T(i) TOLD[(i) - 1]
#define Q(i) QOLD[(i) - 1]
  if (T(1) > T0 || T(NOLD) < T1) return 1;
  int IBEG = 1, IEND = 1;
  bool fb = false, fe = false;
#pragma unroll
  for (int sl = 0; sl < KS; ++sl) {
    const int i = gl + sl * G + 1;
    const double ti = i <= NOLD ? T(i) : 0.0;
    const unsigned long long mb = grp_bits<G>(i >= 2 && i <= NOLD && T0 <= ti), me = grp_bits<G>(i <= NOLD && T1 <= ti);
    if (!fb && mb) { IBEG = sl * G + __ffsll((long long)mb); fb = true; }
    if (!fe && me) { IEND = sl * G + __ffsll((long long)me); fe = true; }
  }
  if (T1 < T(IBEG)) {
    const double SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    const double QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    const double QEST1 = SLOPE * (T1 - T(IBEG - 1)) + Q(IBEG - 1);
    *QNEW = 0.5 * (QEST0 + QEST1);
    return 0;
  }
#pragma unroll
  for (int sl = 0; sl < KS; ++sl) {
    const int i = gl + sl * G + 1;
    if (i > IBEG && i <= IEND) TERM[i - 1] = (T(i) - T(i - 1)) * 0.5 * (Q(i - 1) + Q(i));
  }
  double AREAB = 0.0, AREAE = 0.0, AREAM = 0.0;
  if (T0 < T(IBEG)) {
    const double SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    const double QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    AREAB = (T(IBEG) - T0) * 0.5 * (QEST0 + Q(IBEG));
  }
  if (T1 < T(IEND)) {
    const double SLOPE = (Q(IEND) - Q(IEND - 1)) / (T(IEND) - T(IEND - 1));
    const double QEST1 = SLOPE * (T1 - T(IEND - 1)) + Q(IEND - 1);
    AREAE = (T1 - T(IEND - 1)) * 0.5 * (Q(IEND - 1) + QEST1);
  }
  grp_sync();
  if (IBEG < IEND) {
    for (int IMID = IBEG + 1; IMID < IEND; ++IMID) AREAM = AREAM + TERM[IMID - 1];
    if (T1 == T(IEND) && T0 < T(IEND - 1)) AREAM = AREAM + TERM[IEND - 1];
  }
#undef T
#undef Q
  *QNEW = (AREAB + AREAE + AREAM) / (T1 - T0);
  return 0;
}

// interp_rch as kwt_rch calls it for the time-step average (:257-311): the exit times X(0:NR+1) of the particles routed in this
// step with the last one routed before (X(0) <= T0 < X(1): the +1 s fix of kinwav) and the first one still waiting
// (X(NR) < T1 <= X(NR+1): that is what NR counts), strictly increasing.  Then the two index searches of interp_rch are known
// beforehand -- IBEG = 2, IEND = NOLD -- and are only VERIFIED here (four broadcast reads, one vote of the wavefront; anything
// else goes through the general routine above).  The leading and the trailing partial trapezoid are one instruction sequence
// on even / odd lanes, the interior trapezoids one per lane, and the running sum -- the reference's order -- takes its terms
// from LDS with constant offsets, six at a time.  Same operations on the same operands as interp_rch, term for term.
template <int G, int KS>
__device__ __forceinline__ int grp_interp_step(const double *X, const double *Q, double *TERM, int NOLD, double T0, double T1, int gl, double *QNEW) {
  const double x0 = X[0], x1 = X[1], xl = X[NOLD - 1], xp = X[NOLD - 2];
  const bool ok = NOLD >= 2 && x0 <= T0 && T0 <= x1 && T1 <= xl && (NOLD == 2 || xp < T1);
  if (__ballot(!ok) != 0ull) return grp_interp_rch<G, KS>(X, Q, TERM, NOLD, T0, T1, gl, QNEW);
  // ---- partial trapezoids: even lanes the one that starts at T0 (between points 1 and 2), odd lanes the one that ends at T1
  // (between points NOLD-1 and NOLD); with NOLD = 2 they are the same segment and QNEW is the mean of the two estimates
  const bool side = gl & 1;
  const int ia = side ? NOLD - 2 : 0, ib = ia + 1;              // 0-based
  const double tA = X[ia], tB = X[ib], qA = Q[ia], qB = Q[ib];
  const double Tq = side ? T1 : T0;
  const double SLOPE = (qB - qA) / (tB - tA);
  const double QEST = SLOPE * (Tq - tA) + qA;
  const double part = side ? (T1 - tA) * 0.5 * (qA + QEST) : (tB - T0) * 0.5 * (QEST + qB);
  const double estO = dpp_d<0xF5>(QEST), estE = dpp_d<0xA0>(QEST);        // quad_perm [1,1,3,3] / [0,0,2,2]: the odd / even lane of the pair
  if (NOLD == 2 || T1 < x1) {      // T1 < T(IBEG): both ends of the interval on one segment (:1545-1552)
    *QNEW = 0.5 * (estE + estO);
    // (uniform per group, but not per wavefront: the other groups go on)
  }
  double AREAB = dpp_d<0xA0>(part), AREAE = dpp_d<0xF5>(part);
  if (!(T0 < x1)) AREAB = 0.0;
  if (!(T1 < xl)) AREAE = 0.0;
  // ---- interior trapezoids between points i-1 and i (1-based i = 3 .. NOLD-1, and NOLD when T1 sits on the last point)
  {
    double xh[KS], xl_[KS], qh[KS], ql[KS];
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      const int i = gl + sl * G + 1, ii = (i >= 3 && i <= NOLD) ? i : 2;
      xh[sl] = X[ii - 1]; xl_[sl] = X[ii - 2]; qh[sl] = Q[ii - 1]; ql[sl] = Q[ii - 2];
    }
    lds_held(xh); lds_held(xl_); lds_held(qh); lds_held(ql);
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      const int i = gl + sl * G + 1;
      if (i >= 3 && i <= NOLD) TERM[i - 1] = (xh[sl] - xl_[sl]) * 0.5 * (ql[sl] + qh[sl]);
    }
  }
  grp_sync();
  double AREAM = 0.0;
  const int cnt = NOLD - 3 + ((T1 == xl && T0 < xp) ? 1 : 0);      // terms 3 .. NOLD-1 (+ NOLD)
  const double *tp = TERM + 2;
#pragma unroll 1
  for (int k0 = 0; __ballot(k0 < cnt) != 0ull; k0 += 6) {
    double v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = tp[k0 + k];
    lds_held(v);      // (six terms in flight together; what lies behind the last one is inside the wavefront's LDS and not added)
#pragma unroll
    for (int k = 0; k < 6; ++k) if (k0 + k < cnt) AREAM = AREAM + v[k];
  }
  if (!(NOLD == 2 || T1 < x1)) *QNEW = (AREAB + AREAE + AREAM) / (T1 - T0);
  return 0;
}

#ifdef MZR_KWT_TIMING
#define KCOUNT(i, v) do { if (gl == 0) atomicAdd(&d.dbgCycles[(((blockIdx.x >> 4) & 1023) << 5) + (i)], (unsigned long long)(v)); } while (0)
#define TSTAMP(i) do { const long long _n = clock64(); _sec[i] += (unsigned)(_n - _tprev); if ((blockIdx.x & 15) == 0 && (threadIdx.x & 63) == __ffsll(__ballot(1)) - 1) atomicAdd(&d.dbgCycles[(((blockIdx.x >> 4) & 1023) << 5) + (i)], (unsigned long long)(_n - _tprev)); _tprev = _n; } while (0)
// one record per sampled pass behind the counters: lane group size, largest list and most removals among the pass's reaches, cycles per section
#define MZR_REC_N 65536
#define TRECORD(G_, size_, nrem_) do { int _sz = (size_), _nr = (nrem_); for (int _o = 32; _o > 0; _o >>= 1) { _sz = max(_sz, __shfl_xor(_sz, _o, 64)); _nr = max(_nr, __shfl_xor(_nr, _o, 64)); } \
  if ((blockIdx.x & 15) == 0 && (threadIdx.x & 63) == 0) { unsigned *_rb = (unsigned *)(d.dbgCycles + 32 * 1024); const unsigned _k = atomicAdd(_rb, 1u) % MZR_REC_N; unsigned *_r = _rb + 16 + (size_t)_k * 16; \
    _r[0] = (G_); _r[1] = _sz; _r[2] = _nr; _r[3] = _sec[21] + _sec[25]; _r[4] = _sec[0]; _r[5] = _sec[1] + _sec[10] + _sec[11] + _sec[12]; _r[6] = _sec[2] + _sec[3]; _r[7] = _sec[4] + _sec[5]; _r[8] = _sec[6]; _r[9] = _sec[16] + _sec[17]; _r[10] = _sec[18] + _sec[19] + _sec[7]; _r[11] = _sec[22]; _r[12] = _sec[10]; _r[13] = _sec[11]; _r[14] = _sec[12]; } } while (0)
#define TSTAMP_WAVE(i) do { if ((blockIdx.x & 15) == 0 && (threadIdx.x & 63) == __ffsll(__ballot(1)) - 1) atomicAdd(&d.dbgCycles[(((blockIdx.x >> 4) & 1023) << 5) + (i)], 1ull); } while (0)
#elif defined(MZR_SWEEP_TRACE)
// debugging build (make EXTRA=-DMZR_SWEEP_TRACE, run with MZR_SWEEP_DEBUG=1): every wavefront keeps a record of what it is
// doing (kwt_beat), the schedule tables and records are checked against memory, and every section boundary leaves the
// clock in the record (slots 8 + i), so that a pass that took seconds can say where
#define KCOUNT(i, v) do { } while (0)
#define TSTAMP(i) do { if (PERS && d.swBeat && (i) != 11 && (i) != 12) kwt_beat(d, 8 + (i), (int)wall_clock64()); } while (0)
#define TSTAMP_WAVE(i) do { } while (0)
#define TRECORD(G_, size_, nrem_) do { } while (0)
#else
#define KCOUNT(i, v) do { } while (0)
#define TSTAMP(i) do { } while (0)
#define TSTAMP_WAVE(i) do { } while (0)
#define TRECORD(G_, size_, nrem_) do { } while (0)
#endif

namespace {

struct KwtStep {   // what a lane needs to know about its reach and window step
  int t, par;
  double T0, T1;
  double *Qrow;
  const double *qlat_prev, *qlat_cur;
};
__device__ __forceinline__ KwtStep kwt_step(const MzrDev &d, int t) {
  KwtStep k;
  const int tt = t < 0 ? 0 : t;
  k.t = t; k.par = tt & (MZR_OB_RING - 1);      // outbox slot of the step
  k.T0 = d.t_start + (double)tt * d.dt;
  k.T1 = (d.W == 1) ? d.T1_single : k.T0 + d.dt;   // mzr_step passes TSEC(2) explicitly
  k.Qrow = d.Q + (size_t)tt * d.N;
  k.qlat_prev = d.qlat + (size_t)tt * d.N;          // BASIN_QR(0)
  k.qlat_cur = d.qlat + (size_t)(tt + 1) * d.N;     // BASIN_QR(1)
  return k;
}

// Reaches that do not route particles: headwaters (kwt_route.f90:181-205), lake reaches
// (lake_route replaces kwt_rch) and halo reaches of a partition (replay of the imported record).
template <bool FULL, bool PERS>
__device__ __forceinline__ bool kwt_light(const MzrDev &d, int s, int item, int ltEnd) {
  const int N = d.N;
  unsigned long long st_head = 0;
  int r = -1, t = -1;
  bool act = false;
  if (item < ltEnd) {
    r = d.kwtLight[item];
    t = s - d.sigma[r];
    act = t >= 0 && t < d.W;
  }
  if (PERS) {   // a lake needs the discharge of its upstream reaches, a halo reach overwrites the outbox its downstream reach read two steps ago
    const bool halo = act && FULL && d.haloSlot && d.haloSlot[r] >= 0;
    const int nu = (act && !halo) ? (int)d.nUp[r] : 0, u0 = act ? d.upStart[r] : 0;
    const int dn = (halo && t >= MZR_OB_RING) ? d.down[r] : -1;
    for (int i = 0; __ballot(i < nu) != 0ull; ++i) if (kwt_wait_deps(d, i < nu ? d.kwDone + u0 + i : nullptr, t + 1, nullptr, s, r)) return true;
    if (kwt_wait_deps(d, dn >= 0 ? d.kwDone + dn : nullptr, t - (MZR_OB_RING - 1), nullptr, s, r)) return true;
    if (kwt_wait_deps(d, (act && t >= 1) ? d.kwDone + r : nullptr, t, nullptr, s, r)) return true;      // its own previous step
    // a lake's own state (volume, Hanasaki memory) was written by whichever wavefront took its last step
    if (__ballot(act && !halo) != 0ull) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  if (act) {
    {
      const KwtStep k = kwt_step(d, t);
      bool done = false;
      if (FULL && d.haloSlot) {   // tributary outlet computed in another partition
        const int hs = d.haloSlot[r];
        if (hs >= 0) {
          done = true;
          const size_t nH = d.nHalo;
          stx<PERS>(k.Qrow + r, d.imQ[(size_t)t * nH + hs]);
          const int n = d.imN[(size_t)t * nH + hs];
          stx<PERS>(d.obN + (size_t)k.par * N + r, n);
          double *ob = d.obQT + 2 * (size_t)k.par * MZR_OB_STRIDE * N;
          for (int j = 0; j <= n && n > 0; ++j) {
            stx<PERS>(ob + MZR_PQ(MZR_OBI(j, r)), d.imOQ[((size_t)t * MZR_OB_CAP + j) * nH + hs]);
            stx<PERS>(ob + MZR_PT(MZR_OBI(j, r)), d.imOT[((size_t)t * MZR_OB_CAP + j) * nH + hs]);
          }
        }
      }
      if (FULL && !done && d.lakeSlot) {
        const int ls = d.lakeSlot[r];
        if (ls >= 0) {   // a lake keeps one sentinel particle (init_model_data.f90:431-439)
          done = true;
          double vol = d.vol[r], vol0 = vol, ele = d.ele[r], wb = 0.0, wmAct = 0.0;
          const double Q = mzr_lake::lake_route(d, r, t, ls, k.Qrow, k.qlat_cur[r], vol, vol0, ele, wb, wmAct, PERS);
          stx<PERS>(k.Qrow + r, Q); d.vol[r] = vol; d.vol0[r] = vol0; d.ele[r] = ele; d.wb[r] = wb;
          if (d.trVol0) d.trVol0[(size_t)t * N + r] = vol0;      // REACH_VOL(0) of the step, for the constituent pass (zero for river reaches under KWT)
          if (d.kwN[r] != 1) { d.kwN[r] = 1; d.kwQT[MZR_PQ(MZR_KWI(0, r))] = -9999.0; d.kwQT[MZR_PT(MZR_KWI(0, r))] = -9999.0; d.kwTR[MZR_KWI(0, r)] = -9999.0; }
        }
      }
      if (!done) {   // headwater
        const double qlat_r = k.qlat_cur[r];
        stx<PERS>(k.Qrow + r, qlat_r);
        d.inflow[r] = 0.0;
        if (d.kwN[r] != 1) { d.kwN[r] = 1; d.kwQT[MZR_PQ(MZR_KWI(0, r))] = -9999.0; d.kwQT[MZR_PT(MZR_KWI(0, r))] = -9999.0; d.kwTR[MZR_KWI(0, r)] = -9999.0; }
        if (FULL && d.exportSlot && d.exportSlot[r] >= 0) d.exN[(size_t)t * d.nExp + d.exportSlot[r]] = 0;
        st_head = 1;
      }
    }
  }
  if (PERS) {
    if (__ballot(act && FULL && d.lakeSlot && d.lakeSlot[r] >= 0) != 0ull) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // plain lake state stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (act) stx<true>(d.kwDone + r, (mzr_word)(unsigned)(t + 1));
  }
  if (d.kwtStat) {
    const unsigned long long e = wave_sum(st_head);
    if ((threadIdx.x & 63) == 0 && e) atomicAdd(&d.kwtStat->n_head, e);
  }
  return false;
}

}  // namespace

// RPW = 64/G routed reaches per wavefront, each with a fixed slice of the LDS work arrays (own
// particles + everything its upstreams routed).
// FULL = false compiles out lakes, water management and partition boundaries (the common case).
// Values named "uniform" below are computed redundantly by all lanes of a group.
#ifndef MZR_KWT_KB
#define MZR_KWT_KB 3   // particle slots per lane of the 8-lane class
#endif
#ifndef MZR_KWT_KC
#define MZR_KWT_KC 3   // particle slots per lane of the 4-lane class
#endif
// The sweep exists a second time with MZR_KWT_KC_WIDE slots per lane of the 4-lane class (15 entries: all of the group's slice of the
// pool), for domains whose sweep is bound by instructions alone: reaches of up to 13 entries then share a pass sixteen at a time instead
// of eight.  The fourth slot costs every 4-lane pass, so it is a flavour, not the rule (kwt_regroup; profiles/r05_experiments.md 9), and
// it is this file compiled once more as a translation unit of its own (kernels_kwt_wide.hip: MZR_KWT_TU_WIDE) -- as a second
// instantiation next to the first it changed the register allocation of the first (100 k reaches: 442 -> 447 ms per window).
#ifndef MZR_KWT_KC_WIDE
#define MZR_KWT_KC_WIDE 4
#endif
#ifdef MZR_KWT_TU_WIDE
#undef MZR_KWT_KC
#define MZR_KWT_KC MZR_KWT_KC_WIDE
#endif
#ifndef MZR_KWT_OCC
#define MZR_KWT_OCC 4      // wavefronts per SIMD the kernels are compiled for: 16 one-wavefront workgroups per CU is what the device holds (round 5: 5 -> 4, c3 shard 346.2 -> 342.8 ms)
#endif
#ifndef MZR_KWT_POOL
#define MZR_KWT_POOL 240   // entries of each of the four LDS work arrays of a wavefront: 60 per 16-lane reach, 30 per 8-lane reach
#endif
#ifndef MZR_KWT_KTB
#define MZR_KWT_KTB 4      // entries per lane an 8-lane group can thin (capacity 8 * KTB - 1, and at most its slice of the pool)
#endif
// One reach by a group of G adjacent lanes with KS (OS) particle slots per lane for the own row
// (an outbox row).  `off` = the group's slice of the LDS work arrays, `cap` = how many entries the
// reach may need: a reach that needs more is left untouched and reported back (true), so that the
// caller can give it a wider group.  CAN_THIN = false leaves remove_rch out (cap <= MAXQPAR).  ctx = 8 doubles of LDS for values needed again late.
// PERS: the persistent sweep -- wait for the reaches this step depends on, exchange outbox rows and
// discharge with other wavefronts through sc1 accesses, publish the step in kwDone.  Returns bit 0: the
// reach needs more than `cap` entries, bit 1: the sweep is abandoned (error raised somewhere).
// One step of the reach per call: t = s - stage.
template <bool FULL, bool GEN, int G, int KS, int OS, bool CAN_THIN, bool PERS>
__device__ __forceinline__ int kwt_reach(const MzrDev &d, int s, const MzrKwtRec *recs, int item, bool have, int lastItem,
                                         int off, int cap, double *sA, double *sB, double *sC, double *sD, double *ctx) {
  bool ovf = false;
  // The sweep is as fast as its slowest chain of passes, and those are the wide ones (long particle lists, thinning):
  // they go first whenever the SIMD has a choice.
  const bool boost = PERS && d.sweepPrio;      // the whole sweep of this handle runs at priority 3 (set once in k_sweep_kwt)
#ifndef MZR_NO_PRIO
  if (G >= 16 && !PERS) __builtin_amdgcn_s_setprio(2);      // (the persistent sweep raises it after its wait: a wavefront that polls has no business in front of one that computes)
#endif
  // ---- round trip 1: the static record of the reach (host-packed, one 64-byte line), fetched by eight lanes of
  // the group into LDS (ctx[4..11]) and read from there when a field is needed, not held in registers
  {
    const int gl0 = mzr_lane() & (G - 1);
    double *rc0 = ctx + 4;
    for (int k = gl0; k < 8; k += G) rc0[k] = ((const double *)(recs + (have ? item : lastItem)))[k];
    grp_sync();
    if (PERS && MZR_BEAT_ON(d) && have && gl0 < 4) {      // debugging aid: the record as the caches hold it against what memory holds
      const int *rci0 = (const int *)rc0;
      const int fresh = ldx<true>((const int *)(recs + item) + gl0);
      if (fresh != rci0[gl0]) mzr_raise_stall(d, 40 + gl0, rci0[0], s, item, rci0[gl0], fresh, -1, mzr_lane(), 0, 0, d.swHead);
    }
  }
#ifdef MZR_KWT_TIMING
  long long _tprev = clock64();
  unsigned _sec[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int _recSize = 0, _recRem = 0;
#endif
  mzr_word wword = 0;      // the progress word this lane polled
  mzr_word wSelf = 0;      // the reach's own progress word as last seen / published (its outbox counts of the other slots ride on)
  int n_own = 0;           // at-rest particles of the reach
  const int lane = mzr_lane(), gl = lane & (G - 1);
  const int N = d.N;
  double *rc = ctx + 4;
  const int *rci = (const int *)rc;      // r, sigma | u0, nup flags upGood goodMask | width | CW | length | scA | scB | down, -
  const int r = uni<G>(rci[0]);
  const int tb = have ? s - rci[1] : -1;                                  // step of the reach in this launch of the schedule
  const unsigned rcb = (unsigned)rci[3];
  const __amdgpu_buffer_rsrc_t kwRs = mzr_rsrc(d.kwQT);
  const int nup = (int)(rcb & 0xff), ng = (int)((rcb >> 8) & 15), u0 = rci[2];
  const unsigned upGood = (rcb >> 16) & 0xff, goodMask = rcb >> 24;
  const bool isOut = (rcb & 0x8000u) != 0;
  const bool upLake = FULL && (rcb & 0x4000u) != 0;   // an upstream reach is a lake
  const double RW = rc[2];
  // reach series A / B: first and second non-headwater upstream in UREACHI order
  const int ns = __popc(upGood);
  const int uA = u0 + (upGood ? __ffs(upGood) - 1 : 0);
  const int uB = u0 + ((upGood & (upGood - 1u)) ? __ffs(upGood & (upGood - 1u)) - 1 : 0);
  const int t = uni<G>((have && tb >= 0) ? tb : -1);
  const bool live = t >= 0 && t < d.W;
  const KwtStep ks = kwt_step(d, t);
  const double T0 = ks.T0, T1 = ks.T1;
  const double T_START = T0, T_END = T1;                    // RSTEP = 0
  double *Qrow = ks.Qrow;
  const double *qlat_prev = ks.qlat_prev, *qlat_cur = ks.qlat_cur;
  const int par = ks.par;
  const int *obN = d.obN + (size_t)par * N;
  const double *obQT = d.obQT + 2 * (size_t)par * MZR_OB_STRIDE * N;      // this step's parity of the {Q, exit time} rows
  const __amdgpu_buffer_rsrc_t obRs = mzr_rsrc(obQT);
  if (PERS) {
    // step t of this reach needs step t of every upstream reach (their outbox rows and discharge), its own
    // step t - 1, and overwrites the outbox slot its downstream reach read in step t - MZR_OB_RING
    const int dn = rci[14];
    const mzr_word *wp = nullptr;
    int wneed = 0;
    if (live) {
      if (gl < nup) { wp = d.kwDone + u0 + gl; wneed = t + 1; }
      else if (gl == nup && dn >= 0 && t >= MZR_OB_RING) { wp = d.kwDone + dn; wneed = t - (MZR_OB_RING - 1); }
      else if (gl == nup + 1 && t >= 1) { wp = d.kwDone + r; wneed = t; }     // its own previous step (another wavefront's work)
    }
    if (kwt_wait_deps(d, wp, wneed, &wword, s, r)) return 2;
#ifndef MZR_NO_PRIO
    if (G >= 16 && !boost) __builtin_amdgcn_s_setprio(2);
#endif
    TSTAMP(21);
  }

  // ---- round trip 2: everything that depends on the step, issued together -- particle counts,
  // hillslope inflow of the upstream basins, upstream discharge, own particle row (getusq_rch
  // :598-608; element 0 = last routed particle) and, for the binary confluence, the outbox rows of
  // the upstream reaches.  Rows are fixed-size, so they are read whole before their counts are known.
  int need = 0, NUPS = 0, IMAX = 0, nrA = 0, nrB = 0;
  double q_up = 0.0;
  int st_up = 0;
  KwtBasin bs;
  bs.bsc = 1.0 / RW; bs.b0q0 = bs.b0q1 = bs.b0sl = bs.b1q0 = bs.b1sl = 0.0;   // UWIDTH(basin) = 1
  double q[KS], ti[KS], aq[OS], at[OS], bq[OS], bt[OS];
#pragma unroll
  for (int j = 0; j < KS; ++j) q[j] = ti[j] = 0.0;
#pragma unroll
  for (int j = 0; j < OS; ++j) aq[j] = at[j] = bq[j] = bt[j] = 0.0;
  // The counts that say how much of the fixed-size rows is alive came with the progress words (from the second step of
  // a window on, binary confluence below ordinary reaches): the rows are then read up to their counts only -- about
  // half of them -- and the count loads go away.  Otherwise counts and whole rows are fetched together.
  const bool exact = PERS && !GEN && !upLake && !(FULL && (rcb & 0x2000u)) && t >= 1;
  const bool exactNext = PERS && !GEN && !upLake && !(FULL && (rcb & 0x2000u));      // the next step of the window takes the count from the progress word
  if (live) {
    int n_own_v = 0, nrA_v = 0, nrB_v = 0;
    const int gbase = lane & ~(G - 1);
    if (PERS && t >= 1) wSelf = (mzr_word)__shfl((long long)wword, gbase + nup + 1, 64);      // (also carries the counts of the other outbox slots on)
    if (exact) {
      n_own_v = MZR_KWD_OWN(wSelf);
      if (ns > 0) nrA_v = MZR_KWD_OUT((mzr_word)__shfl((long long)wword, gbase + (uA - u0), 64), par);
      if (ns > 1) nrB_v = MZR_KWD_OUT((mzr_word)__shfl((long long)wword, gbase + (uB - u0), 64), par);
    } else {
      n_own_v = ldx<PERS>(d.kwN + r);
      if (!GEN && !upLake) { if (ns > 0) nrA_v = ldx<PERS>(obN + uA); if (ns > 1) nrB_v = ldx<PERS>(obN + uB); }
    }
    // exit time of the reach's last routed particle = the end of its previous step (the first at-rest element's TR, :1304): inside a
    // window of the sweep that is T0 + dt of the step before, the same expression that produced it -- one sector read (and, below,
    // written) per reach-step less; the first step of a window takes it from the state
    const double X0 = (PERS && t >= 1 && d.W > 1) ? kwt_step(d, t - 1).T1 : ldx<PERS>(d.kwTR + MZR_KWI(0, r));
    const double hin = d.hInflow ? ldx<PERS>(d.hInflow + r) : 0.0;      // history sum of REACH_INFLOW, when asked for
    const double qlat_r = qlat_cur[r];
    double b1q1 = 0.0, up0 = 0.0, up1 = 0.0;
    bs.b0q0 = qlat_prev[u0]; bs.b0q1 = qlat_cur[u0];
    if (nup > 1) { bs.b1q0 = qlat_prev[u0 + 1]; b1q1 = qlat_cur[u0 + 1]; }
    if (!GEN) { up0 = ldx<PERS>(Qrow + u0); if (nup > 1) up1 = ldx<PERS>(Qrow + u0 + 1); }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = gl + j * G, kk = k < MZR_KW_CAP ? k : 0;
      if (!exact || k < n_own_v) { const mzr_d2 v = ldq<PERS>(kwRs, d.kwQT, MZR_KWI(kk, r)); q[j] = v.x; ti[j] = v.y; }
    }
    if (!GEN && !upLake) {
#pragma unroll
      for (int j = 0; j < OS; ++j) {
        const int k = gl + j * G, kk = k < MZR_OB_CAP ? k : 0;
        if (ns > 0 && (!exact || k < nrA_v)) { const mzr_d2 v = ldq<PERS>(obRs, obQT, MZR_OBI(kk, uA)); aq[j] = v.x; at[j] = v.y; }
        if (ns > 1 && (!exact || k < nrB_v)) { const mzr_d2 v = ldq<PERS>(obRs, obQT, MZR_OBI(kk, uB)); bq[j] = v.x; bt[j] = v.y; }
      }
    }
    // ---- uniform: the work-array need
    n_own = uni<G>(n_own_v); nrA = uni<G>(nrA_v); nrB = uni<G>(nrB_v);
    IMAX = nup;
    int NUPR = 0;
    bool empty = false;
    if (upLake && nup > 1) mzr_raise(d, 10, r, t, 18);   // lake outlet reach should have one upstream lake, :551-553
    if (!upLake) {
      if (!GEN) {
        if (ns > 0) { if (nrA < 2) empty = true; ++NUPR; IMAX += nrA - 1; st_up += nrA + 1; }   // nr < 2: upstream published nothing
        if (ns > 1) { if (nrB < 2) empty = true; ++NUPR; IMAX += nrB - 1; st_up += nrB + 1; }
      } else {
        for (int i = 0; i < nup; ++i) {
          if ((upGood >> i) & 1u) {
            const int nr = uni<G>(ldx<PERS>(obN + u0 + i));
            if (nr < 2) empty = true;
            ++NUPR; IMAX += nr - 1; st_up += nr + 1;
          }
        }
      }
    }
    NUPS = nup + NUPR;
    const int NJ0 = n_own == 0 ? 0 : n_own - 1;
    need = NJ0 + 1 + ((NUPS == 1 || upLake) ? 1 : IMAX);
    if (upLake && nup > 1) need = 0;
    if (empty) { mzr_raise(d, 40, r, t, 11); need = 0; }
    if (need > cap) { ovf = true; need = 0; }
    const double dT10 = T1 - T0;
    bs.b0sl = (bs.b0q1 - bs.b0q0) / dT10;
    if (nup > 1) bs.b1sl = (b1q1 - bs.b1q0) / dT10;
    // upstream discharge, kwt_rch :163-174
    if (!GEN) {
      if (ng > 0 && (goodMask & 1u)) q_up = q_up + up0;
      if (ng > 1 && (goodMask & 2u)) q_up = q_up + up1;
    } else {
      for (int i = 0; i < ng; ++i) { if (!((goodMask >> i) & 1u)) continue; q_up = q_up + ldx<PERS>(Qrow + u0 + i); }
    }
    if (gl == 0) {
      double *c = ctx;
      c[2] = q_up;                     // REACH_INFLOW, stored with the other results at the end
      c[0] = n_own == 0 ? T0 : X0; c[3] = hin;     // getusq_rch :587-596: a reach without particles starts at T0
      c[1] = qlat_r;
      if (d.kwtStat && !ovf) {
        // particle-traffic counters (mzr_set_profiling 2).  The persistent sweep keeps them per reach slot in LDS (ctx[12..15]) and
        // adds them to the device counters once, when the wavefront leaves: five atomics per routed reach-step on six addresses
        // made a counted window last 5.9 s instead of 0.45 s (profiles/r05_soak.md)
        if (PERS) {
          unsigned long long *cs = (unsigned long long *)(c + 12);
          cs[0] += (unsigned long long)n_own;
          cs[1] += (unsigned long long)st_up; cs[3] += 1ull | ((unsigned long long)nup << 32);
        } else {
          atomicAdd(&d.kwtStat->w_in, (unsigned long long)n_own);
          atomicAdd(&d.kwtStat->w_up, (unsigned long long)st_up);
          atomicAdd(&d.kwtStat->n_route, 1ull); atomicAdd(&d.kwtStat->n_edges, (unsigned long long)nup);
        }
      }
    }
  }
  TSTAMP(0);

  // work arrays: a fixed slice of the wavefront's LDS pool per group (GP entries: a binary
  // confluence needs at most 20 + 1 + 2 + 2*19 of them)
  if (need > 0) {
    {
      double *const Qw = sA + off, *const Tw = sB + off, *const Xw = sC + off, *const Yw = sD + off;
      do {
        bool cold = (n_own == 0);
        int NJ = cold ? 0 : n_own - 1;
        const bool binary = !GEN && !upLake && NUPS != 1;   // GEN: launch over the confluences of more than two reaches
#pragma unroll
        for (int j = 0; j < KS; ++j) { const int k = gl + j * G; if (k < n_own) { Qw[k] = q[j]; Tw[k] = ti[j]; } }
        // minval(Q) < 0 (kwt_rch :163-174) is taken from the registers the list is made of -- the reach's own particles here, every merged
        // flow where it is computed -- instead of a pass over the list in LDS (one dependent round trip per G entries, in every pass)
        bool neg = false, negLds = GEN;
#pragma unroll
        for (int j = 0; j < KS; ++j) { const int k = gl + j * G; neg = neg || (k < n_own && q[j] < 0.0); }
        const bool negOwn = neg;
        if (binary) {
#pragma unroll
          for (int j = 0; j < OS; ++j) {
            const int k = gl + j * G;
            if (ns > 0 && k < nrA) { Xw[k] = aq[j]; Yw[k] = at[j]; }
            if (ns > 1 && k < nrB) { Xw[nrA + k] = bq[j]; Yw[nrA + k] = bt[j]; }
          }
          if (ns == 1 && gl == 0) Yw[nrA] = DBL_MAX;      // (the second series does not exist: a time nothing is later than, where the rank search looks)
        }
        grp_sync();
        TSTAMP(10);

        // ---- qexmul_rch
        int ND;
        double *QD = Qw + NJ + 1, *TD = Tw + NJ + 1;
        if (upLake) {      // lake outflow enters the river as one particle, getusq_rch :554-559
          const double ql = ldx<PERS>(Qrow + u0) / RW;
          if (gl == 0) { QD[0] = ql; TD[0] = T1; }
          neg = neg || ql < 0.0;
          ND = 1;
        } else if (NUPS == 1) {   // one upstream basin that is a headwater, :743-759
          const double qh = bs.b0q1 / RW;
          if (gl == 0) { QD[0] = qh; TD[0] = T1; }
          neg = neg || qh < 0.0;
          ND = 1;
        } else if (binary) {
          // Output = 2-way merge of the routed particles in time order, each with the other series
          // interpolated at its time, followed by one particle at T1 (:929-957).  Rank of a
          // particle = its index in its own series + the number of particles of the other series
          // that the cursor walk consumes before it (ties: series A first).
          const double *SAq = Xw, *SAt = Yw, *SBq = Xw + nrA, *SBt = Yw + nrA;
          const double scA = rc[5], scB = rc[6];
          const int nA = ns > 0 ? nrA - 2 : 0, nB = ns > 1 ? nrB - 2 : 0;
          ND = nA + nB + 1;
          bool slow = false;
          {   // the particle at T1, led by the first basin series: once per reach (every lane of the group the same; it used to sit
              // in the loop below, where one lane of one group makes the whole wavefront issue it in every turn)
            const double CT = T1;
            double Q_AGG = 0.0;
            Q_AGG = Q_AGG + bs.b0q1 * bs.bsc;
            if (nup > 1) Q_AGG = Q_AGG + (bs.b1q0 + bs.b1sl * (CT - T0)) * bs.bsc;
            if (ns > 0) {
              const double tb = SAt[nA], te = SAt[nA + 1], qb = SAq[nA], qe = SAq[nA + 1];
              if (te < CT || tb > CT) slow = true;
              const double SLOPE = (qe - qb) / (te - tb);
              Q_AGG = Q_AGG + (qb + SLOPE * (CT - tb)) * scA;
            }
            if (ns > 1) {
              const double tb = SBt[nB], te = SBt[nB + 1], qb = SBq[nB], qe = SBq[nB + 1];
              if (te < CT || tb > CT) slow = true;
              const double SLOPE = (qe - qb) / (te - tb);
              Q_AGG = Q_AGG + (qb + SLOPE * (CT - tb)) * scB;
            }
            if (gl == 0) { QD[nA + nB] = Q_AGG; TD[nA + nB] = CT; }
            neg = neg || Q_AGG < 0.0;
          }
          for (int m = gl; m < nA + nB; m += G) {
            double CT, Q_AGG = 0.0;
            int pos = m;
            {
              const bool isA = m < nA;
              const int i = isA ? m + 1 : m - nA + 1;                 // index in the own series
              const double *St = isA ? SAt : SBt, *Sq = isA ? SAq : SBq;
              const double *Ot = isA ? SBt : SAt, *Oq = isA ? SBq : SAq;
              const int nO = isA ? nB : nA;
              const bool other = isA ? ns > 1 : true;
              // (everything of the loop body that does not depend on the particle's own time is read together with it: the particle before
              // it, its flow, and the first four probes of the rank search)
              const int lim = other ? nO + 1 : 0;
              const int stride = (nA > nB ? nA : nB) <= 4 ? 1 : 4;
              double Sp = St[i - 1], SqI = Sq[i], p1[4];
              CT = St[i];
#pragma unroll
              for (int u = 0; u < 4; ++u) p1[u] = Ot[min(stride * (u + 1), lim)];
              lds_held(CT); lds_held(Sp); lds_held(SqI); lds_held(p1);
              if (!(CT < T1)) slow = true;                            // error 40 in the cursor walk
              if (i > 1 && !(Sp < CT)) slow = true;                   // duplicate or unordered
              // rank among the other tributary's particles = how many of them are earlier (an equal time anywhere sends the
              // group to the literal cursor walk, so "earlier" and "earlier or equal" need not be told apart).  Short lists:
              // one LDS round trip for four of them; long lists: bisection (each series is checked for order by its own lanes)
              // (no range tests on the probes: the entry behind the other series' routed particles is its end-of-step particle at T1 > CT
              // -- a probe clamped to it counts nothing -- and a series that does not exist is one sentinel, staged above)
              int cnt = 0;
              if (stride == 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  cnt += p1[u] < CT ? 1 : 0;
                  slow = slow || p1[u] == CT;
                }
              } else {
                // cnt = how many of Ot[1..nO] (ascending, nO <= 19) are earlier than CT, in two LDS round trips: the pivots 4, 8, 12, 16
                // together (k of them earlier: the answer lies in 4 k .. 4 k + 3), then the three entries behind pivot k together.  (Five
                // dependent probes of a bisection and a sixth for the equal time: 0.8 us of a pass's 15-25.)  An equal time is the first
                // entry that is not earlier, 4 k + 1 .. 4 k + 4: one of the seven values read.
                double p2[3];
                int k4 = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) { k4 += p1[u] < CT ? 4 : 0; slow = slow || p1[u] == CT; }
#pragma unroll
                for (int u = 0; u < 3; ++u) p2[u] = Ot[min(k4 + 1 + u, lim)];
                lds_held(p2);
                cnt = k4;
#pragma unroll
                for (int u = 0; u < 3; ++u) { cnt += p2[u] < CT ? 1 : 0; slow = slow || p2[u] == CT; }
              }
              pos = (i - 1) + cnt;
              TSTAMP(11);
              Q_AGG = Q_AGG + (bs.b0q0 + bs.b0sl * (CT - T0)) * bs.bsc;
              if (nup > 1) Q_AGG = Q_AGG + (bs.b1q0 + bs.b1sl * (CT - T0)) * bs.bsc;
              double SOWN = SqI * (isA ? scA : scB), SOTH = 0.0;
              if (other) {
                const double tb = Ot[cnt], te = Ot[cnt + 1], qb = Oq[cnt], qe = Oq[cnt + 1];
                if (te < CT || tb > CT) slow = true;
                const double SLOPE = (qe - qb) / (te - tb);
                SOTH = (qb + SLOPE * (CT - tb)) * (isA ? scB : scA);
              }
              // summation order of the series: A, then B (:929-957)
              if (isA) { Q_AGG = Q_AGG + SOWN; if (other) Q_AGG = Q_AGG + SOTH; }
              else { Q_AGG = Q_AGG + SOTH; Q_AGG = Q_AGG + SOWN; }
            }
            QD[pos] = Q_AGG; TD[pos] = CT;
            neg = neg || Q_AGG < 0.0;
            TSTAMP(12);
          }
          if (grp_any<G>(slow)) {
            KCOUNT(8, 1);
            grp_sync();
            int nd = 0;
            if (gl == 0) nd = kwt_merge_binary_serial(nup, ns, nrA, nrB, SAq, SAt, SBq, SBt, scA, scB, bs, T0, T1, QD, TD);
            ND = grp_first<G>(nd);
            negLds = true; neg = negOwn;      // (the list was made by one lane: looked at in LDS; what the lanes had worked out does not count)
          }
        } else {
          int nd = -60;
          if (GEN && gl == 0) nd = kwt_merge_generic<PERS>(nup, u0, NUPS, RW, T0, T1, d.nGood, d.width, qlat_prev, qlat_cur, obN, obQT, N, QD, TD, IMAX, Xw, (int *)Yw);
          ND = grp_first<G>(nd);
        }
        TSTAMP(1);
        if (ND < 0) { mzr_raise(d, -ND, r, t, 11); break; }
        grp_sync();
        if (cold) {   // getusq_rch :587-596
          const double DT = T1 - T0;
          if (gl == 0) { Qw[0] = Qw[1]; Tw[0] = T0 - DT - DT * 0; }
          grp_sync();
        }
        int size = NJ + 1 + ND;
#ifdef MZR_KWT_TIMING
        _recSize = size;
#endif

        {   // kwt_rch :163-174 (minval(Q) < 0: one vote of the group)
          if (negLds) { for (int k = gl + NJ + 1; k < size; k += G) neg = neg || Qw[k] < 0.0; }
          if (grp_any<G>(neg)) { mzr_raise(d, 20, r, t, 12); break; }
        }
        TSTAMP(2);

        // ---- remove_rch :999-1123: drop the particle with the least interpolation error until < MAXQPAR
        if (CAN_THIN && size > MZR_MAXQPAR_DEV) {
          KCOUNT(13, 1); KCOUNT(14, size - MZR_MAXQPAR_DEV);
#ifdef MZR_KWT_TIMING
          _recRem = size - MZR_MAXQPAR_DEV;
#endif
          // a pass that has to thin is the slowest kind, and the sweep is as fast as its slowest chain of passes:
          // it goes first whenever the SIMD has a choice
#ifndef MZR_NO_PRIO
          if (!boost) __builtin_amdgcn_s_setprio(3);
#endif
          const int NPRT = size - 1;
          const bool big = GEN && NPRT > 63;      // beyond the alive bit-mask: neighbours found by walking the error array
          unsigned long long mask = NPRT >= 63 ? ~0ull : ((2ull << NPRT) - 1ull);   // bits 0..NPRT
          int MPRT = NPRT;
          if (!big) {
            // Errors and the alive list live in LDS: Xw[i] = interpolation error of particle i (removed: +Inf), Yw[i] = its WORD,
            //   i << 24 | aa << 18 | a << 12 | b << 6 | bb     a / b = the alive particles before / behind i, aa / bb = the ones before a / behind b
            // (six bits each: a list that thins here holds at most 64 particles).  The word is also the PAYLOAD of the arg-min -- its
            // leading index makes the smallest word among equal errors MINLOC's first minimum -- so the winner's word names every
            // particle the removal touches, and what a turn reads behind the reduction (the six values of the two re-evaluations, the
            // four words it patches) is one LDS round trip.  History: round 3 kept errors and a 64-bit alive mask in registers (three
            // compares and six selects per slot to patch them), round 4 moved them to LDS with prev / next links (a removal: read the
            // winner's links, then the neighbour's, then the values: four dependent round trips), round 6 put the links into the
            // payload and then the neighbours' neighbours too.  Same errors, same MINLOC order, same survivors throughout.
            constexpr int KT = G >= 16 ? 64 / G : MZR_KWT_KTB;      // entries before thinning: at most 60 (16 lanes), 8 * KTB - 1 (8 lanes)
            const bool side = gl & 1;
            double *E = Xw;
            int *LK = (int *)Yw;      // LK[2 i]: the word of particle i (8-byte slots: E[i] and LK[2 i] are one ds_read2_b64)
            // (most lists that thin hold fewer than 2 G particles: the slots beyond the second are only looked at when some
            // group of the wavefront needs them)
            constexpr int KN = KT < 2 ? KT : 2;
            const bool wide = KT > KN && __ballot(NPRT >= KN * G) != 0ull;
            // (the six values of a slot's first error for two slots at a time, read before either is worked on)
            auto err0 = [&](int j0) {
              double v[2][6];
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int i = gl + (j0 + jj) * G, ic = (i >= 1 && i < NPRT) ? i : 1;
                v[jj][0] = Tw[ic]; v[jj][1] = Qw[ic - 1]; v[jj][2] = Qw[ic + 1]; v[jj][3] = Tw[ic - 1]; v[jj][4] = Tw[ic + 1]; v[jj][5] = Qw[ic];
              }
              lds_held(v[0]); lds_held(v[1]);
#pragma unroll
              for (int jj = 0; jj < 2; ++jj) {
                const int i = gl + (j0 + jj) * G;
                if (i <= NPRT) {
                  double ei = DBL_MAX;
                  if (i >= 1 && i < NPRT) ei = fabs(interp3(v[jj][0], v[jj][1], v[jj][2], v[jj][3], v[jj][4]) - v[jj][5]);
                  E[i] = ei; LK[2 * i] = (i << 24) | (max(i - 2, 0) << 18) | (max(i - 1, 0) << 12) | (min(i + 1, NPRT) << 6) | min(i + 2, NPRT);
                }
              }
            };
            static_assert(KN == 2 && (KT == 2 || KT == 4), "slots of the thinning in pairs");
            err0(0);
            if (wide) err0(2);
            grp_sync();
            // A removal is ~190 instructions of one wavefront, one after the other (the SIMD has other wavefronts to issue from, this
            // chain does not): on the reaches that thin 25 particles a step -- the window's longest chain at 100 k reaches -- every
            // instruction of the turn is ~0.2 % of the window.  So: ONE divergent region per turn for the writes (lanes 0-1 the two
            // neighbours, lane 2 the removed particle: a neighbour that is an end point gets the DBL_MAX it holds anyway instead of a
            // test, the removed particle its own word back), no test of the neighbour's range before the values are read (an end
            // point's far link points one past the list, inside the wavefront's LDS; what is computed from it is not used), and the
            // turn without a candidate leaves through the loop's own condition (MPRT < 0) instead of a break of its own.
            const int sideEnd = side ? NPRT : 0;      // the end point on this lane's side: never re-evaluated
            // where a lane finds its indices in the winner's word  i << 24 | aa << 18 | a << 12 | b << 6 | bb  (six bits each: a list that thins
            // here holds at most 64 particles), by its role gl & 3
            const int role = gl & 3;
            const int shx = role == 0 ? 12 : role == 1 ? 6 : role == 2 ? 18 : 0;       // the particle whose word the lane patches: a, b, aa, bb
            const int she = role == 0 ? 12 : role == 1 ? 6 : 24;                        // ... whose error it writes: a, b, ISEL
            const int slo = side ? 12 : 18, shi = side ? 0 : 6;                         // the neighbours of the re-evaluation: (aa, b) | (a, bb)
            const int shs = role == 2 ? 12 : role == 3 ? 0 : 6;                         // (pay << 6) >> shs puts the fields the lane moves in place
            const int wmask = role == 0 ? 0x00000fff : role == 1 ? 0x00fff000 : role == 2 ? 0x0000003f : 0x00fc0000;
            do {
              double emin = DBL_MAX;
              int pay = 0;
              // (error and word of a slot are read together and unconditionally -- one ds_read2_b64 -- and chosen by selects: a
              // word fetched only where its error wins is a second, dependent LDS round trip per slot)
              double evs[KT];
              long long pkw[KT];
              auto fetch = [&](int j) { const int i = gl + j * G, ii = i <= NPRT ? i : 0; evs[j] = E[ii]; pkw[j] = ((const long long *)Yw)[ii]; };      // (the word's 8-byte slot: fuses with E[ii])
              // (a slot beyond the list reads entry 0, whose error is DBL_MAX for good -- the first particle is never removed or
              // re-evaluated -- and is never below the minimum so far: no range test on the value.  A wide turn has its four reads in
              // flight together: fetched behind the first two slots' compares they were a second LDS round trip per removal)
              auto take = [&](int j) {
                const double ev = evs[j];
                const bool lt = ev < emin;
                emin = lt ? ev : emin; pay = lt ? (int)pkw[j] : pay;
              };
              // (`held`: the words are in registers before the first compare -- left to itself the compiler loads a slot's word only
              // where its error wins, the dependent round trip again)
              auto held = [&](int j) { asm volatile("" : "+v"(pkw[j])); };
              if (wide) {
#pragma unroll
                for (int j = 0; j < KT; ++j) fetch(j);
#pragma unroll
                for (int j = 0; j < KT; ++j) held(j);
#pragma unroll
                for (int j = 0; j < KT; ++j) take(j);
              } else {
#pragma unroll
                for (int j = 0; j < KN; ++j) fetch(j);
#pragma unroll
                for (int j = 0; j < KN; ++j) held(j);
#pragma unroll
                for (int j = 0; j < KN; ++j) take(j);
              }
              pay = grp_argmin_pos<G>(emin, pay);          // first minimum of ABSERR (removed entries hold +Inf); the index leads the word
              const bool none = (unsigned)(pay - 1) >= 0x7ffffffeu;      // 0 (or 0x7fffffff): no finite interpolation error left (NaN/Inf input)
              // The winner's word names its two neighbours a < ISEL < b AND theirs (aa = the one before a, bb = the one behind b): every index
              // the turn needs, so what it reads -- the six values of the two re-evaluations, the four words it patches -- is ONE LDS round
              // trip behind the reduction (reading the neighbour's word first to learn its far link was a second, dependent one).
              //   lane 0: a between aa and b, word of a: (next, next-but-one) <- (b, bb)       lane 2: word of aa: next-but-one <- b, E[ISEL] <- Inf
              //   lane 1: b between a and bb, word of b: (previous, the one before) <- (a, aa)   lane 3: word of bb: the one before <- a
              // The fields a lane moves sit at the same bits in the winner's word (lanes 0-1) or six bits off (lanes 2-3): one shift pair
              // and one v_bfi per lane.  A neighbour that is an end point gets its DBL_MAX back; end points' words are never a winner's.
              const unsigned up = (unsigned)pay;
              const int xw = (int)((up >> shx) & 63u), xe = (int)((up >> she) & 63u);
              const int a = (int)((up >> slo) & 63u), b = (int)((up >> shi) & 63u);
              const int c = xw;                                  // (lanes 0-1: the neighbour this lane re-evaluates)
              const int wold = LK[2 * xw];
              const double en = fabs(interp3(Tw[c], Qw[a], Qw[b], Tw[a], Tw[b]) - Qw[c]);
              const int wx = (wold & ~wmask) | ((int)(((unsigned)pay << 6) >> shs) & wmask);
              const double vx = gl >= 2 ? (double)INFINITY : c != sideEnd ? en : DBL_MAX;      // removed: never the minimum again
              grp_sync();
              if (gl < 4 && !none) { LK[2 * xw] = wx; E[xe] = vx; }
              grp_sync();
              MPRT = none ? -1 : MPRT - 1;
            } while (MPRT >= MZR_MAXQPAR_DEV);
            {   // who is left
              mask = 0ull;
              double ev[KT];
#pragma unroll
              for (int j = 0; j < KT; ++j) { const int i = gl + j * G; ev[j] = (j < KN || wide) ? E[i <= NPRT ? i : 0] : 0.0; }
              lds_held(ev);
#pragma unroll
              for (int j = 0; j < KN; ++j) { const int i = gl + j * G; mask |= grp_bits<G>(i <= NPRT && ev[j] != INFINITY) << (j * G); }
              if (wide) {
#pragma unroll
                for (int j = KN; j < KT; ++j) { const int i = gl + j * G; mask |= grp_bits<G>(i <= NPRT && ev[j] != INFINITY) << (j * G); }
              }
              grp_sync();
            }
          } else {
          for (int i = gl; i <= NPRT; i += G) {
            double e = DBL_MAX;
            if (i >= 1 && i < NPRT) e = fabs(interp3(Tw[i], Qw[i - 1], Qw[i + 1], Tw[i - 1], Tw[i + 1]) - Qw[i]);
            Xw[i] = e;
          }
          grp_sync();
          while (MPRT >= MZR_MAXQPAR_DEV) {
            double emin = DBL_MAX; int ISEL = 0;
            for (int i = gl; i <= NPRT; i += G) { const double e = Xw[i]; if (e < emin) { emin = e; ISEL = i; } }
            grp_argmin<G, false>(emin, ISEL);           // first minimum of ABSERR (removed entries hold +Inf)
            ISEL = uni<G>(ISEL);
            if (ISEL == 0) break;                         // no finite interpolation error left (NaN/Inf input)
            auto prevA = [&](int c) -> int { int i = c - 1; while (i > 0 && (i == ISEL || Xw[i] == INFINITY)) --i; return i; };
            auto nextA = [&](int c) -> int { int i = c + 1; while (i < NPRT && (i == ISEL || Xw[i] == INFINITY)) ++i; return i; };
            const int pm = prevA(ISEL), pn = nextA(ISEL);   // INDEX1(ISEL-1), INDEX1(ISEL+1)
            // the two neighbours are re-evaluated against their new neighbours, one on even and one on odd lanes
            const bool side = gl & 1;
            const int c = side ? pn : pm;
            const bool valid = side ? pn < NPRT : pm > 0;
            double e = 0.0;
            if (valid) {   // pm: between INDEX1(pm-1) and pn; pn: between pm and INDEX1(pn+1)
              const int a = side ? pm : prevA(pm), b = side ? nextA(pn) : pn;
              e = fabs(interp3(Tw[c], Qw[a], Qw[b], Tw[a], Tw[b]) - Qw[c]);
            }
            grp_sync();
            if (valid && gl < 2) Xw[c] = e;
            if (gl == 0) Xw[ISEL] = INFINITY;              // removed: never the minimum again
            --MPRT;
            grp_sync();
          }
          }
          if (MPRT >= MZR_MAXQPAR_DEV || MPRT < 0) { mzr_raise(d, 62, r, t, 16); break; }
          if (!big) {
            // compact in place: every lane takes its survivors into registers, then writes them to their new places (k <= i).  The
            // arrays keep their roles, so their addresses stay base + constant for the rest of the pass (the round-2 form compacted
            // into the two free arrays and swapped the pointers: every LDS access behind it paid an address addition)
            constexpr int KC_ = G >= 16 ? 64 / G : MZR_KWT_KTB;
            double cq[KC_], ct[KC_];
#pragma unroll
            for (int j = 0; j < KC_; ++j) { const int i = gl + j * G, ic = i <= NPRT ? i : 0; cq[j] = Qw[ic]; ct[j] = Tw[ic]; }      // (all slots' reads together; a slot that does not survive is not written back)
            lds_held(cq); lds_held(ct);
            grp_sync();
#pragma unroll
            for (int j = 0; j < KC_; ++j) {
              const int i = gl + j * G;
              if (i <= NPRT && ((mask >> i) & 1ull)) { const int k = __popcll(mask & ((1ull << i) - 1ull)); Qw[k] = cq[j]; Tw[k] = ct[j]; }
            }
          } else if (gl == 0) {
            int k = 0;
            for (int i = 0; i <= NPRT; ++i) if (Xw[i] != INFINITY) { Qw[k] = Qw[i]; Tw[k] = Tw[i]; ++k; }
          }
          size = MPRT + 1;
          grp_sync();
        }
        // ---- extract_from_rch :351-455 (water abstraction / injection on the particles).  Its
        // recomputed exit times are overwritten by kinwav below, so only the flows change.
        if (FULL && d.is_flux_wm && d.wm) {
          const double Qtake = d.wm[(size_t)t * N + r];
          if (Qtake != -9999.0) {
            double Qavg;
            if (d_interp_rch(Tw, Qw, size, T_START, T_END, &Qavg)) { mzr_raise(d, 1, r, t, 17); break; }
            grp_sync();
            const double totQ = Qavg * rc[2];
            if (Qtake > 0.0) {
              const double Qfrac = Qtake / totQ;
              for (int i = 1 + gl; i < size; i += G) Qw[i] = Qw[i] * (1.0 + Qfrac);
            } else if (Qtake < 0.0 && fabs(Qtake) < totQ) {
              const double Qfrac = fabs(Qtake) / totQ;
              for (int i = 1 + gl; i < size; i += G) Qw[i] = Qw[i] * (1.0 - Qfrac);
            } else {
              const double mf = d.minflow[r];
              for (int i = gl; i < size; i += G) Qw[i] = mf;
            }
            grp_sync();
          }
        }
        TSTAMP(3);

        // ---- kinwav_rch :1130-1439 on particles 1..NI (NI <= 19 after thinning):
        //   Xw[i]   wave celerity of the group whose first particle is i   (WC)
        //   Yw[i]   1/WC during the shock search, then the exit time of the group headed by i
        //   alive   bit i set while particle i still heads a group (uniform)
        //   Xw[i+1] entry time of a merged group (T1 after :1335); flows of a merged group are the
        //           min / max over its members (:1329-1330), recomputed when needed
        const int NI = size - 1;
        int NQ2 = 0;
        {
          // cw = ALFA*K**(1/ALFA) with K = sqrt(R_SLOPE)/R_MAN_N and ALFA = 5/3, XMX = RLENGTH: from the record (host, once)
          const double cw = rc[3], XMX = rc[4];
          double wcs[KS], rws[KS];     // celerity of the lane's own particles and its reciprocal
          bool shock = false;
          double tes[KS];
          bool zero = false;
          double qv[KS], tv_[KS];      // flow and entry time of the lane's own particles (read together, before the first power)
#pragma unroll
          for (int j = 0; j < KS; ++j) { const int i = gl + j * G, ic = (i >= 1 && i <= NI) ? i : 1; qv[j] = Qw[ic]; tv_[j] = Tw[ic]; }
          lds_held(qv); lds_held(tv_);
#pragma unroll
          for (int j = 0; j < KS; ++j) {
            const int i = gl + j * G;
            wcs[j] = 0.0; rws[j] = 0.0;
            if (i >= 1 && i <= NI) { const double wc = cw * pow_0p4(qv[j]); wcs[j] = wc; rws[j] = 1.0 / wc; Xw[i] = wc; Yw[i] = rws[j]; }
          }
          grp_sync();
          TSTAMP(4);
          // Does any wave break at all?  The first pass of the shock search (:1301-1320) with every particle still its
          // own group -- neighbours are i and i+1, entry times are the particles' own.  Shocks are rare (a handful per
          // 10^5 reach-steps on the benchmark forcing): without one the routed list is the list itself, every particle
          // with the exit time RLENGTH/celerity + entry time (:1363-1372, rUpdate :1409-1437), and the group machinery
          // below is skipped.
          // (The search keeps the smallest crossing point in [0, XMX] and gives up when that is XMX itself, :1301-1322: a wave
          // breaks iff SOME pair crosses inside [0, XMX) -- one vote of the group instead of an arg-min.)
          shock = false;
          if (NI > 1) {
            bool cross = false;
            double wn[KS], rn[KS], tn[KS];
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) { const int jw = gl + sl * G, jc = (jw >= 1 && jw < NI) ? jw + 1 : 1; wn[sl] = Xw[jc]; rn[sl] = Yw[jc]; tn[sl] = Tw[jc]; }
            lds_held(wn); lds_held(rn); lds_held(tn);
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) {
              const int jw = gl + sl * G;
              if (jw >= 1 && jw < NI) {
                const double wcj = wcs[sl], wci = wn[sl], tj = tv_[sl], ti = tn[sl];
                if (!(wci == 0.0 || wcj == 0.0) && !(wcj > wci && ti > tj)) {
                  const double WDIFF = rws[sl] - rn[sl];
                  // (a later, faster wave always catches up somewhere; nearly always far beyond the end of the reach.  The quotient
                  // is only formed where the product cannot rule that out: (ti - tj) > XMX * WDIFF * (1 + 1e-9) with both
                  // positive means (ti - tj) / WDIFF > XMX whatever the two roundings do)
                  const double dtt = ti - tj;
                  if (!(WDIFF == 0.0) && !(wci == wcj) && !(WDIFF > 0.0 && dtt > (XMX * WDIFF) * (1.0 + 1.e-9))) {
                    const double XXB = dtt / WDIFF;
                    if (!(XXB < 0.0 || XXB > XMX) && XXB < XMX) cross = true;
                  }
                }
              }
            }
            shock = grp_any<G>(cross);
          }
          zero = false;
          if (!shock) {
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) {
              const int h = gl + sl * G;
              tes[sl] = 0.0;
              if (h >= 1 && h <= NI) {
                if (wcs[sl] < DBL_MIN) zero = true;                                // zero flow :1365
                else {
                  double te = fmin(XMX / wcs[sl] + tv_[sl], DBL_MAX);
                  if (h == 1 && te <= T_START) te = T_START + 1.0;
                  tes[sl] = te;
                }
              }
            }
          }
          if (!shock) {
            if (grp_any<G>(zero)) { mzr_raise(d, 20, r, t, 13); break; }
            grp_sync();          // the neighbours have read the celerities this overwrites
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) { const int h = gl + sl * G; if (h >= 1 && h <= NI) Xw[h] = tes[sl]; }
            NQ2 = NI;
          } else {
          unsigned alive = (2u << NI) - 2u;   // bits 1..NI
          auto nextHead = [&](int h) -> int {           // next group head after h, or NI+1
            const unsigned m = alive & ~((2u << h) - 1u);
            return m ? __ffs(m) - 1 : NI + 1;
          };
          auto groupT = [&](int h, int hn) -> double { return hn - h > 1 ? Xw[h + 1] : Tw[h]; };
          if (NI > 1) {
            double X = 0.0;
            for (;;) {
              // crossing point of every pair of neighbouring groups; the reference keeps the smallest
              // one in [X, XMX], the later pair on ties (:1301-1320)
              double XB = DBL_MAX; int JXB = 0;
#pragma unroll
              for (int sl = 0; sl < KS; ++sl) {
                const int jw = gl + sl * G;
                if (jw >= 1 && jw <= NI && ((alive >> jw) & 1u)) {
                  const int iw = nextHead(jw);
                  if (iw <= NI) {
                    const int inx = nextHead(iw);
                    const double wcj = Xw[jw], wci = Xw[iw], tj = groupT(jw, iw), ti = groupT(iw, inx);
                    // earlier wave faster and later entry: XXB < 0 <= X (or WDIFF == 0) -> no break, no division
                    if (!(wci == 0.0 || wcj == 0.0) && !(wcj > wci && ti > tj)) {
                      const double WDIFF = Yw[jw] - Yw[iw];
                      if (!(WDIFF == 0.0) && !(wci == wcj)) {
                        const double XXB = (ti - tj) / WDIFF;
                        if (!(XXB < X || XXB > XMX) && XXB <= XB) { XB = XXB; JXB = jw; }
                      }
                    }
                  }
                }
              }
              grp_argmin<G, true>(XB, JXB);
              JXB = uni<G>(JXB); XB = uni<G>(XB);
              if (JXB == 0 || XB == XMX) break;
              // merge group IXB into group JXB (:1325-1346)
              KCOUNT(15, 1);
              const int IXB = nextHead(JXB);
              const int endI = nextHead(IXB);
              double q2 = Qw[JXB], q1 = Qw[JXB];
              for (int j = JXB + 1; j < endI; ++j) { const double q = Qw[j]; q2 = fmax(q2, q); q1 = fmin(q1, q); }
              const double K = d.kwK[r];                            // shock merges are rare: fetched on demand
              const double a = pow_0p6(((gl & 1) ? q1 : q2) / K);   // A2 on even, A1 on odd lanes
              const double b = dpp_d<MZR_DPP_XOR1>(a);
              const double A2 = (gl & 1) ? b : a, A1 = (gl & 1) ? a : b;
              const double CM = (q2 - q1) / (A2 - A1);
              const double tJ = groupT(JXB, IXB);
              const double wcJ = Xw[JXB];
              alive &= ~(1u << IXB);
              grp_sync();
              if (gl == 0) { Xw[JXB + 1] = tJ + XB / wcJ - XB / CM; Xw[JXB] = CM; Yw[JXB] = 1.0 / CM; }
              X = XB;
              grp_sync();
            }
          }
          TSTAMP(5);
          // exit time of every group (:1363-1372)
          bool zero = false;
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) {
            const int h = gl + sl * G;
            if (h >= 1 && h <= NI && ((alive >> h) & 1u)) {
              const double wc = Xw[h];
              if (wc < DBL_MIN) zero = true;                                // zero flow :1365
              else Yw[h] = fmin(XMX / wc + groupT(h, nextHead(h)), DBL_MAX);
            }
          }
          if (grp_any<G>(zero)) { mzr_raise(d, 20, r, t, 13); break; }
          grp_sync();
          // The routed list (rUpdate :1409-1437): a group of one particle stays one particle; a
          // merged group becomes two particles when it leaves within the step (:1381-1386), all
          // its members with a common exit time when it does not (:1388-1392), one particle when
          // its flows are all equal.  ICOUNT of a group = heads before it + what merged groups
          // before it add.
          const unsigned mergedHeads = alive & ~(alive >> 1) & ((1u << NI) - 1u);
          int hS[KS], adj[KS], extra = 0;
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) {
            const int J = gl + sl * G;
            hS[sl] = (J >= 1 && J <= NI) ? 31 - __clz((int)(alive & ((2u << J) - 1u))) : 0;
            adj[sl] = 0;
          }
          for (unsigned m = mergedHeads; m; m &= m - 1u) {
            const int g = __ffs(m) - 1, hn = nextHead(g);
            double q1 = Qw[g], q2 = q1;
            for (int j = g + 1; j < hn; ++j) { const double q = Qw[j]; q2 = fmax(q2, q); q1 = fmin(q1, q); }
            const int c = uni<G>((q1 != q2) ? (Yw[g] < T_END ? 2 : hn - g) : 1);
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) if (hS[sl] > g) adj[sl] += c - 1;
            extra += c - 1;
          }
          NQ2 = __popc(alive) + extra;
          int oI[KS]; double oQ[KS], oT[KS], oX[KS];
          bool bad30 = false;
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) {
            const int J = gl + sl * G;
            oI[sl] = 0; oQ[sl] = 0.0; oT[sl] = 0.0; oX[sl] = 0.0;
            if (J >= 1 && J <= NI) {
              const int h = hS[sl], hn = nextHead(h);
              const int base = __popc(alive & ((1u << h) - 1u)) + 1 + adj[sl];
              if (hn - h == 1) { oI[sl] = base; oQ[sl] = Qw[J]; oT[sl] = Tw[J]; oX[sl] = Yw[J]; }
              else {
                double q1 = Qw[h], q2 = q1;
                for (int j = h + 1; j < hn; ++j) { const double q = Qw[j]; q2 = fmax(q2, q); q1 = fmin(q1, q); }
                const double TEXIT = Yw[h], tg = Xw[h + 1];
                if (q1 != q2) {
                  if (TEXIT < T_END) {
                    const double TNEXT = hn <= NI ? Yw[hn] : DBL_MAX;           // = TEXIT of the next group (:1372)
                    const double TEXIT2 = fmin(TEXIT + 1.0, TEXIT + 0.5 * (fmin(TNEXT, T_END) - TEXIT));
                    if (TEXIT2 == TEXIT) bad30 = true;
                    if (J == h) { oI[sl] = base; oQ[sl] = q1; oT[sl] = tg; oX[sl] = TEXIT; }
                    else if (J == h + 1) { oI[sl] = base + 1; oQ[sl] = q2; oT[sl] = tg; oX[sl] = TEXIT2; }
                  } else { oI[sl] = base + (J - h); oQ[sl] = Qw[J]; oT[sl] = Tw[J]; oX[sl] = TEXIT; }
                } else if (J == h) { oI[sl] = base; oQ[sl] = q1; oT[sl] = tg; oX[sl] = TEXIT; }
              }
            }
          }
          if (grp_any<G>(bad30)) { mzr_raise(d, 30, r, t, 13); break; }
          grp_sync();
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) {
            if (oI[sl] > 0) {
              double te = oX[sl];
              if (oI[sl] == 1 && te <= T_START) te = T_START + 1.0;
              Qw[oI[sl]] = oQ[sl]; Tw[oI[sl]] = oT[sl]; Xw[oI[sl]] = te;
            }
          }
          }
          if (gl == 0) Xw[0] = ctx[0];
          grp_sync();
          // exit times must increase: te <= previous -> previous + 1 s (:1423-1426); sequential only when it happens
          bool viol = false;
          {
            double xa[KS], xb[KS];
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) { const int k2 = gl + sl * G, kk = (k2 >= 2 && k2 <= NQ2) ? k2 : 1; xa[sl] = Xw[kk]; xb[sl] = Xw[kk - 1]; }
            lds_held(xa); lds_held(xb);
#pragma unroll
            for (int sl = 0; sl < KS; ++sl) { const int k2 = gl + sl * G; if (k2 >= 2 && k2 <= NQ2 && xa[sl] <= xb[sl]) viol = true; }
          }
          if (grp_any<G>(viol)) {
            KCOUNT(9, 1);
            if (gl == 0) for (int k2 = 2; k2 <= NQ2; ++k2) { const double xp = Xw[k2 - 1]; if (Xw[k2] <= xp) Xw[k2] = xp + 1.0; }
            grp_sync();
          }
        }
        TSTAMP(6);

        // ---- time-step average and housekeeping, kwt_rch :257-311
        int NR = 0;
        {
          double xc[KS];
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) { const int i = gl + sl * G; xc[sl] = Xw[(i >= 1 && i <= NQ2) ? i : 0]; }
          lds_held(xc);
#pragma unroll
          for (int sl = 0; sl < KS; ++sl) { const int i = gl + sl * G; NR += grp_count<G>(i >= 1 && i <= NQ2 && xc[sl] < T_END); }   // count(FROUTE)-1
        }
        if (NR + 1 > NQ2) { mzr_raise(d, 61, r, t, 14); break; }      // no waiting particle left
        TSTAMP(16);
        const double qN = Qw[NR], qN1 = Qw[NR + 1], xN = Xw[NR], xN1 = Xw[NR + 1], tN = Tw[NR], tN1 = Tw[NR + 1];
        const double dTx = xN1 - xN;
        const double Q_END = qN + ((qN1 - qN) / dTx) * (T_END - xN);
        const double TIMEI = tN + ((tN1 - tN) / dTx) * (T_END - xN);
        const int NN2 = NQ2 - NR;
        // addresses of the result rows are rebuilt from the step index here rather than kept in
        // registers since the loads at the top
        int tq = t;
        if (G < 64) asm volatile("" : "+v"(tq));
        double QNEW;
        const int _ibad = grp_interp_step<G, OS>(Xw, Qw, Yw, NR + 2, T_START, T_END, gl, &QNEW);
        if (_ibad) { mzr_raise(d, 1, r, t, 15); break; }
        TSTAMP(17);
        const double Qout = QNEW * rc[2] + ctx[1];
        // (the history sum of REACH_Q is taken from the Q rows once per window, k_accum_qsum)
        // The particle count and REACH_INFLOW of a step are single words in sectors of their own: 64 bytes written for 4 / 8.  In
        // the sweep the count travels in the progress word (the reach reads it back from there, `exact` above), so both are
        // written where somebody reads them: at the last step of the window (state getters, regrouping, the next window's
        // first step), and the count every step for the reaches that do not take it from the progress word.
        const bool lastStep = tq == d.W - 1;
        if (gl == 0) {
          stx<PERS>(d.Q + (size_t)tq * N + r, Qout);
          if (!exactNext || lastStep) stx<PERS>(d.kwN + r, NN2 + 1);
          if (!PERS || lastStep) d.inflow[r] = ctx[2];
          if (d.hInflow) stx<PERS>(d.hInflow + r, ctx[3] + ctx[2]);
        }
        TSTAMP(18);
        // record for the downstream reach: KWAVE(0:NR+1) + first waiting particle (flow, exit time)
        const int es = (FULL && d.exportSlot) ? d.exportSlot[r] : -1;
        const bool outbox = !isOut;
        if (outbox || es >= 0) {
          const int pq = tq & (MZR_OB_RING - 1);
          int *obNw = d.obN + (size_t)pq * N;
          double *obW = d.obQT + 2 * (size_t)pq * MZR_OB_STRIDE * N;
          const __amdgpu_buffer_rsrc_t obWs = mzr_rsrc(obW);
          if (gl == 0 && outbox) stx<PERS>(obNw + r, NR + 2);
          if (gl == 0 && es >= 0) d.exN[(size_t)tq * d.nExp + es] = NR + 2;
          double oq[OS], ox[OS];      // (the routed particles of all slots read before the first store)
#pragma unroll
          for (int j = 0; j < OS; ++j) { const int k2 = gl + j * G, kc = k2 <= NR ? k2 : 0; oq[j] = Qw[kc]; ox[j] = Xw[kc]; }
          lds_held(oq); lds_held(ox);
#pragma unroll
          for (int j = 0; j < OS; ++j) {
            const int k2 = gl + j * G;
            if (k2 <= NR + 2) {
              const double q = k2 <= NR ? oq[j] : k2 == NR + 1 ? Q_END : qN1;
              const double x = k2 <= NR ? ox[j] : k2 == NR + 1 ? T_END : xN1;
              if (outbox) stq<PERS>(obWs, obW, MZR_OBI(k2, r), q, x);
              if (es >= 0) {   // tributary outlet of a partition: the same record goes to the time-indexed export buffer
                const size_t nE = d.nExp;
                d.exOQ[((size_t)tq * MZR_OB_CAP + k2) * nE + es] = q; d.exOT[((size_t)tq * MZR_OB_CAP + k2) * nE + es] = x;
              }
            }
          }
        }
        TSTAMP(19);
        // at-rest state: KWAVE(NR+1:NQ2+1)
        double rq[KS], rt[KS];
#pragma unroll
        for (int j = 0; j < KS; ++j) { const int k2 = gl + j * G, kc = NR + (k2 <= NN2 ? k2 : 0); rq[j] = Qw[kc]; rt[j] = Tw[kc]; }
        lds_held(rq); lds_held(rt);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          const int k2 = gl + j * G;
          if (k2 <= NN2) {
            const bool first = k2 == 0;
            stq<PERS>(kwRs, d.kwQT, MZR_KWI(k2, r), first ? Q_END : rq[j], first ? TIMEI : rt[j]);
            // expected exit times are recomputed every step: only element 0 is read back, the others are kept for restart files (last step of a window)
            if ((first && !(PERS && d.W > 1)) || tq == d.W - 1) stx<PERS>(d.kwTR + MZR_KWI(k2, r), first ? T_END : Xw[NR + k2]);
          }
        }
        if (d.kwtStat && gl == 0) {
          if (PERS) ((unsigned long long *)(ctx + 12))[2] += (unsigned long long)(NQ2 + 2);
          else atomicAdd(&d.kwtStat->w_out, (unsigned long long)(NQ2 + 2));
        }
        TSTAMP(7); TSTAMP_WAVE(20);
        if (PERS) {   // results written through (sc1) and drained, then the step is published
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (gl == 0) {
            const int pq = tq & (MZR_OB_RING - 1);
            const mzr_word keep = wSelf & MZR_KWD_ALLOUT & ~MZR_KWD_OUTMASK(pq);      // the other slots' counts stay (0 in the first step)
            const int nOut = isOut ? 0 : NR + 2;
            const mzr_word wNew = (mzr_word)(unsigned)((tq + 1) | ((NN2 + 1) << 16)) | ((mzr_word)(unsigned)nOut << (21 + 5 * pq)) | keep;
            stx<true>(d.kwDone + r, wNew);
          }
          TSTAMP(22);
          kwt_beat(d, 3, 4);
          if (MZR_BEAT_ON(d) && lane == 0) {      // debugging aid: a pass that took unreasonably long between the end of its wait and its publish
            const int *bt = d.swBeat + (size_t)blockIdx.x * MZR_BEAT;
            const int now = (int)wall_clock64(), dt = now - ldx<true>(bt + 7);
            if (dt > 1000000) {
              const int k = atomicAdd(&d.err->nSlow, 1);
              if (k < 32) {
                for (int j = 0; j < 24; ++j) d.err->slowT[k][j] = ldx<true>(bt + 8 + j) - ldx<true>(bt + 7);
                int *o = d.err->slow[k]; o[0] = blockIdx.x; o[1] = s; o[2] = ldx<true>(bt + 1); o[3] = dt; o[4] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4); o[5] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7; o[6] = now; o[7] = G;
              }
            }
          }
        }
      } while (0);
    }
  }
#ifdef MZR_KWT_TIMING
  if (PERS) { TRECORD(G, _recSize, _recRem); }
#endif
  if ((CAN_THIN || G >= 16) && !boost) __builtin_amdgcn_s_setprio(0);
  return ovf ? 1 : 0;
}

// Lane classes.  A routed reach is worked on by a group of adjacent lanes; how many is the host's choice
// (kwt_regroup, by the work-array entries the reach needed in the last step):
//   class C   4 lanes, 16 reaches per wavefront, at most 11 entries, no thinning code
//   class B   8 lanes,  8 reaches per wavefront, at most 30 entries
//   class A  16 lanes,  4 reaches per wavefront, at most 60 entries (a full binary confluence)
// A reach that has outgrown its class is left untouched by its group and routed right away by 16-lane groups
// of the same wavefront, four at a time, so the classification only has to be usually right.
#define MZR_CTX 16     // doubles of LDS per reach slot: X0, BASIN_QR(1), inflow, history sum, the 64-byte record, then (persistent sweep) four particle-traffic counters
struct KwtCls {
  static constexpr int GA = 16, RA = 4, KA = (MZR_KW_CAP + GA - 1) / GA, OA = (MZR_OB_CAP + GA - 1) / GA;
  static constexpr int GB = 8, RB = 8, KB = MZR_KWT_KB;
  static constexpr int GC = 4, RC = 16, KC = MZR_KWT_KC;
};
// class-B / class-C reaches of one item that overflowed (bit g = group g of the narrow pass): the k-th of them for wide group g16
__device__ __forceinline__ int kwt_pick(unsigned &ovfMask, int g16) {
  unsigned m = ovfMask;
  int sel = -1;
  for (int k = 0; k <= g16 && m; ++k) { sel = (k == g16) ? __ffs(m) - 1 : -1; m &= m - 1u; }
  for (int k = 0; k < KwtCls::RA && ovfMask; ++k) ovfMask &= ovfMask - 1u;
  return sel;
}

// One launch = every routed, headwater, lake and halo reach of the stages that are active in this
// launch: blocks of class-A, class-B and class-C reaches, then one lane per light reach.  GEN: confluences of
// more than two reaches.  Used for short windows (mzr_step: one step per call); long windows take k_sweep_kwt.
template <bool FULL, bool GEN, int POOL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GEN ? 1 : MZR_KWT_OCC, GEN ? 2 : MZR_KWT_OCC)))
k_stage_kwt(MzrDev d, int s, int haBegin, int haEnd, int nABlocks, int hbBegin, int hbEnd, int nBBlocks, int hcBegin, int hcEnd, int nCBlocks,
            int ltBegin, int ltEnd) {
  constexpr int GA = KwtCls::GA, RA = KwtCls::RA, KA = KwtCls::KA, OA = KwtCls::OA;
  constexpr int GB = KwtCls::GB, RB = KwtCls::RB, KB = KwtCls::KB, GC = KwtCls::GC, RC = KwtCls::RC, KC = KwtCls::KC;
  constexpr int GPA = POOL / RA, GPB = POOL / RB, GPC = POOL / RC;
  // entries 0..G*K-1 (the outbox write reaches index NR+2 <= size) within the group's slice of the pool
  constexpr int CAPB = GB * MZR_KWT_KTB - 1 < GPB ? GB * MZR_KWT_KTB - 1 : GPB, CAPC = GC * KC - 1 < GPC ? GC * KC - 1 : GPC;
  // the four work arrays are one block: an entry's address in one of them is its address in the first plus a constant, which the
  // LDS instructions carry as their immediate offset (one address register per entry instead of four)
  __shared__ double sW[4 * POOL];
  double *sA = sW, *sB = sW + POOL, *sC = sW + 2 * POOL, *sD = sW + 3 * POOL;
  __shared__ double sCtx[RC][MZR_CTX];
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  if (!GEN && b >= nABlocks + nBBlocks + nCBlocks) {
    kwt_light<FULL, false>(d, s, ltBegin + (b - nABlocks - nBBlocks - nCBlocks) * 64 + lane, ltEnd);
    return;
  }
  const int cls = (GEN || b < nABlocks) ? 0 : b < nABlocks + nBBlocks ? 1 : 2;
  unsigned ovfMask = 0;      // groups of the narrow pass whose reach needs the wide path (wave-uniform)
  int base = 0;
  if (cls == 1) {
    const int g8 = lane / GB;
    base = hbBegin + (b - nABlocks) * RB;
    const bool ovf = kwt_reach<FULL, false, GB, KB, KB, true, false>(d, s, d.kwtRoutedB, base + g8, base + g8 < hbEnd, hbEnd - 1, g8 * GPB, CAPB, sA, sB, sC, sD, sCtx[g8]) & 1;
    const unsigned long long bal = __ballot(ovf);
#pragma unroll
    for (int g = 0; g < RB; ++g) ovfMask |= (unsigned)((bal >> (g * GB)) & 1ull) << g;
    if (!ovfMask) return;
  } else if (cls == 2) {
    const int g4 = lane / GC;
    base = hcBegin + (b - nABlocks - nBBlocks) * RC;
    const bool ovf = kwt_reach<FULL, false, GC, KC, KC, false, false>(d, s, d.kwtRoutedC, base + g4, base + g4 < hcEnd, hcEnd - 1, g4 * GPC, CAPC, sA, sB, sC, sD, sCtx[g4]) & 1;
    const unsigned long long bal = __ballot(ovf);
#pragma unroll
    for (int g = 0; g < RC; ++g) ovfMask |= (unsigned)((bal >> (g * GC)) & 1ull) << g;
    if (!ovfMask) return;
  }
  // class A, or the reaches of this block that have outgrown their narrow group, four at a time; a reach that needs more than
  // a quarter of the pool (GPA entries; a full binary confluence can ask for 60) is routed once more ALONE with the whole
  // pool (`solo`: bit g = 16-lane group g of the pass before; the first group takes it)
  const int g16 = lane / GA;
  unsigned solo = 0;
  int itemKeep = 0;
#pragma unroll 1
  for (;;) {
    const MzrKwtRec *recs = cls == 0 ? d.kwtRouted : cls == 1 ? d.kwtRoutedB : d.kwtRoutedC;
    const int last = (cls == 0 ? haEnd : cls == 1 ? hbEnd : hcEnd) - 1;
    const bool isSolo = solo != 0;
    int item, offA = g16 * GPA, capA = GPA;
    bool have;
    if (isSolo) {
      const int g = __ffs(solo) - 1; solo &= solo - 1u;
      item = __shfl(itemKeep, g * GA, 64); have = g16 == 0; offA = 0; capA = POOL;
    } else {
      item = haBegin + b * RA + g16;
      have = item <= last;
      if (cls != 0) { const int sel = kwt_pick(ovfMask, g16); have = sel >= 0; item = base + (have ? sel : 0); }
      itemKeep = item;
    }
    const bool ovf = kwt_reach<FULL, GEN, GA, KA, OA, true, false>(d, s, recs, item, have, last, offA, capA, sA, sB, sC, sD, sCtx[g16]) & 1;
    if (isSolo || GEN) { if (ovf) mzr_raise(d, 60, recs[have ? item : last].r, s, 10); }      // work array bounds exceeded
    else { const unsigned long long bal = __ballot(ovf); solo = (unsigned)((bal & 1ull) | (((bal >> 16) & 1ull) << 1) | (((bal >> 32) & 1ull) << 2) | (((bal >> 48) & 1ull) << 3)); }
    if (!solo && !ovfMask) break;
  }
}

// The rare item kinds of the persistent sweep are real calls, so that their registers are not part of the loop body's.
template <bool FULL, int POOL>
__device__ __noinline__ int kwt_item_generic(const MzrDev &d, int s, int bi, double *sA, double *sB, double *sC, double *sD, double *ctx) {
  constexpr int GA = KwtCls::GA, KA = KwtCls::KA, OA = KwtCls::OA;
  const int g16 = mzr_lane() / GA;
  return kwt_reach<FULL, true, GA, KA, OA, true, true>(d, s, d.kwtGeneric, bi, g16 == 0, d.nG - 1, 0, POOL, sA, sB, sC, sD, ctx + MZR_CTX * g16);
}
template <bool FULL>
__device__ __noinline__ bool kwt_item_light(const MzrDev &d, int s, int bi) {
  return kwt_light<FULL, true>(d, s, bi * 64 + mzr_lane(), d.nDepLight);
}

// ------------------------------------------------------------------------------------------------
// Persistent sweep.  ONE launch advances the skewed schedule through launches s = sBegin .. sEnd-1
// of k_stage_kwt.  The items of a launch (blocks of 4 class-A reaches, 8 class-B reaches, one
// confluence of more than two reaches, or 64 lake / halo reaches; stage-ordered list made by the
// host) are numbered in launch order and drawn by the wavefronts from ticket counters -- eight of
// them, one per XCD (queue q = items i with i % 8 == q), because one address takes < 100 atomics
// per microsecond; a wavefront serves the queue of the XCD it runs on and, once that is empty,
// what is left in the others.  What a kernel boundary used to guarantee is now per reach: step t
// of reach r starts when kwDone of its upstream reaches has reached t+1, its own t, and that of
// its downstream reach t-1 (kwt_reach).  Every ticket depends only on tickets of the launch before,
// each queue is served in order and nothing is owned by a particular wavefront, so the sweep moves
// whichever wavefronts of the grid happen to be resident -- no co-residency assumption; wavefronts
// that wait sleep, and give up when an error was raised or nothing has moved for seconds.
// Results cross CUs, so state, outbox rows and discharge go through sc1 accesses (ldx / stx).
// Headwater reaches need nothing from anybody and are filled in by k_kwt_window_init.
// Wavefronts per workgroup of the sweep.  The wavefronts of the sweep are independent of each other (no barrier, an LDS slice each);
// workgroups of one wavefront stop at 16 per CU -- four per SIMD -- whatever registers and LDS would allow (the census of
// mzr_sweep_kwt_capacity: 4 008 of them on 256 CUs with 8 KB as with 9 KB of LDS each), so more than four wavefronts per SIMD
// take workgroups of two.
#ifndef MZR_KWT_WG
#define MZR_KWT_WG 1
#endif
#ifdef MZR_KWT_TU_WIDE
namespace mzr_kwt_wide {      // (the kernel's name must differ from the first translation unit's: template instantiations are merged by the linker)
#endif
template <bool FULL, int POOL>
__global__ void __launch_bounds__(64 * MZR_KWT_WG) __attribute__((amdgpu_waves_per_eu(MZR_KWT_OCC, MZR_KWT_OCC)))
k_sweep_kwt(MzrDev dArg, int sBegin, int sEnd) {
  // The domain description has ~100 fields; kept live around the item loop they spill.  They are read
  // through the kernel-argument segment instead (scalar loads, constant address space) and the
  // pointer is made opaque once per item, so that nothing is hoisted out of the loop.
  typedef const MzrDev __attribute__((address_space(4))) *MzrDevK;
  typedef const int __attribute__((address_space(4))) *IntK;
  MzrDevK dk0 = (MzrDevK)__builtin_amdgcn_kernarg_segment_ptr();
  const MzrDev &d0 = *(const MzrDev *)dk0;
  constexpr int GA = KwtCls::GA, RA = KwtCls::RA, KA = KwtCls::KA, OA = KwtCls::OA;
  constexpr int GB = KwtCls::GB, RB = KwtCls::RB, KB = KwtCls::KB, GC = KwtCls::GC, RC = KwtCls::RC, KC = KwtCls::KC;
  constexpr int GPA = POOL / RA, GPB = POOL / RB, GPC = POOL / RC;
  constexpr int CAPB = GB * MZR_KWT_KTB - 1 < GPB ? GB * MZR_KWT_KTB - 1 : GPB, CAPC = GC * KC - 1 < GPC ? GC * KC - 1 : GPC;
  __shared__ double sW_[MZR_KWT_WG][4 * POOL];      // the four work arrays, one block (constant distances: immediate offsets of the LDS instructions)
  __shared__ double sCtx_[MZR_KWT_WG][RC][MZR_CTX];
  const int wv = MZR_KWT_WG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // this wavefront's slice of the workgroup's LDS
  double *sA = sW_[wv], *sB = sA + POOL, *sC = sA + 2 * POOL, *sD = sA + 3 * POOL;
  double (*sCtx)[MZR_CTX] = sCtx_[wv];
  if (sEnd < 0) { mzr_census(d0.swHead + 8 * 16); return; }      // host: mzr_sweep_kwt_capacity
  if (ldx<true>(&d0.err->code) != 0) return;      // a window that failed stays as it is (and is not built upon)
  if (d0.kwtStat && mzr_lane() < RC) { unsigned long long *cs = (unsigned long long *)(sCtx[mzr_lane()] + 12); cs[0] = cs[1] = cs[2] = cs[3] = 0ull; }
  const int arr = mzr_sweep_join(d0.swHead, d0.swClock, d0.sweepAlways);      // (a wavefront that starts behind time does not join)
  if (arr < 0) return;
  if (d0.sweepPrio) __builtin_amdgcn_s_setprio(3);      // mzr_config.sweepPriority: a small, deep domain sweeping beside a large one
  const int Wm1 = d0.W - 1;
  const int q0 = arr < 64 ? (arr & 7) : (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7);     // HW_REG_XCC_ID: a speed hint only
  kwt_beat(d0, 5, q0); kwt_beat(d0, 3, 1); kwt_beat(d0, 4, 0);
  int nDone = 0;
  IntK P = (IntK)d0.swP, RAs = (IntK)d0.swRA;
  // (the item tables too: uniform indices into memory nobody writes during the launch -- scalar loads, not a vector-memory round trip per ticket)
  IntK ILo = (IntK)d0.swLo, IHi = (IntK)d0.swHi, IIt = (IntK)d0.swItem;
#pragma unroll 1
  for (int dq = 0; dq < 8; ++dq) {
    const int q = (q0 + dq) & 7;
    const int pEnd = P[sEnd * 8 + q];
    int sCur = sBegin, pLo = P[sBegin * 8 + q], pHi = P[(sBegin + 1) * 8 + q];   // tickets [pLo, pHi) of queue q belong to launch sCur
#pragma unroll 1
    for (;;) {
      MzrDevK dk = dk0;
      asm volatile("" : "+s"(dk));
      const MzrDev &d = *(const MzrDev *)dk;
      int k = 0;
      if (mzr_lane() == 0) k = atomicAdd(d.swHead + q * 16, 1);
      k = __builtin_amdgcn_readfirstlane(k);
      if (k >= pEnd) break;
      if (k >= pHi) {   // the next launch, or (after a pause, or in a queue taken over from another XCD) a later one
        ++sCur; pLo = pHi; pHi = P[(sCur + 1) * 8 + q];
        if (k >= pHi) {
          int lo = sCur + 1, hi = sEnd - 1;          // largest s with P[s] <= k
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P[mid * 8 + q] <= k) lo = mid; else hi = mid - 1; }
          sCur = lo; pLo = P[sCur * 8 + q]; pHi = P[(sCur + 1) * 8 + q];
        }
      }
      const int s = sCur;
      const int a = RAs[s];
      const int i = a + ((q - a) & 7) + 8 * (k - pLo);
      if (s < ILo[i] || s > IHi[i] + Wm1) continue;        // none of the item's reaches has a step in this launch
      if (MZR_BEAT_ON(d)) {
        kwt_beat(d, 0, s); kwt_beat(d, 1, i); kwt_beat(d, 2, q); kwt_beat(d, 6, k); kwt_beat(d, 3, 1); kwt_beat(d, 4, ++nDone);
        // debugging aid: the schedule tables as the scalar cache holds them against what memory holds (sc1 vector loads)
        const int c0 = pLo, c1 = pHi, c2 = a, c3 = d.swLo[i], c4 = d.swHi[i], c5 = d.swItem[i];
        const int f0 = ldx<true>(d.swP + sCur * 8 + q), f1 = ldx<true>(d.swP + (sCur + 1) * 8 + q), f2 = ldx<true>(d.swRA + s);
        const int f3 = ldx<true>(d.swLo + i), f4 = ldx<true>(d.swHi + i), f5 = ldx<true>(d.swItem + i);
        const int bad = c0 != f0 ? 0 : c1 != f1 ? 1 : c2 != f2 ? 2 : c3 != f3 ? 3 : c4 != f4 ? 4 : c5 != f5 ? 5 : -1;
        if (bad >= 0 && mzr_lane() == 0) {
          const int cv[6] = {c0, c1, c2, c3, c4, c5}, fv[6] = {f0, f1, f2, f3, f4, f5};
          mzr_raise_stall(d, 30 + bad, -1, s, i, cv[bad], fv[bad], q, k, 0, 0, d.swHead);
        }
      }
      const int lane = mzr_lane(), g16 = lane / GA;
      const int it = IIt[i];
      const int cls = it >> 28, bi = it & 0x0fffffff;      // 0 A, 1 B, 2 generic, 3 lake / halo, 4 C
      if (cls == 3) {
        if (kwt_item_light<FULL>(d, s, bi)) return;
        continue;
      }
      if (cls == 2) {   // one confluence of more than two reaches, first lane group, the whole pool
        const int st = kwt_item_generic<FULL, POOL>(d, s, bi, sA, sB, sC, sD, &sCtx[0][0]);
        if (__ballot(st & 2) != 0ull) return;
        if (st & 1) mzr_raise(d, 60, d.kwtGeneric[bi].r, s, 10);
        continue;
      }
      unsigned ovfMask = 0;
      int stNarrow = 0;      // per lane: what the narrow pass said about its group's reach (bit 0: outgrown)
      if (cls == 1) {
        const int g8 = lane / GB, item = bi * RB + g8;
        stNarrow = kwt_reach<FULL, false, GB, KB, KB, true, true>(d, s, d.kwtRoutedB, item, item < d.nB, d.nB - 1, g8 * GPB, CAPB, sA, sB, sC, sD, sCtx[g8]);
        if (__ballot(stNarrow & 2) != 0ull) return;
        const unsigned long long bal = __ballot(stNarrow & 1);
        if (bal == 0ull) continue;      // (nearly always)
#pragma unroll
        for (int g = 0; g < RB; ++g) ovfMask |= (unsigned)((bal >> (g * GB)) & 1ull) << g;
      } else if (cls == 4) {
        const int g4 = lane / GC, item = bi * RC + g4;
        stNarrow = kwt_reach<FULL, false, GC, KC, KC, false, true>(d, s, d.kwtRoutedC, item, item < d.nC, d.nC - 1, g4 * GPC, CAPC, sA, sB, sC, sD, sCtx[g4]);
        if (__ballot(stNarrow & 2) != 0ull) return;
        const unsigned long long bal = __ballot(stNarrow & 1);
        if (bal == 0ull) continue;
#pragma unroll
        for (int g = 0; g < RC; ++g) ovfMask |= (unsigned)((bal >> (g * GC)) & 1ull) << g;
      }
      // class A, or the reaches of this item that have outgrown their narrow group, four at a time; a reach that needs more than a
      // quarter of the pool (GPA entries; a full binary confluence can ask for 60) is taken up once more ALONE with the whole pool
      // (`solo`: bit g = 16-lane group g of the pass before)
      unsigned solo = 0;
      int itemKeep = 0;
#pragma unroll 1
      for (;;) {
        const MzrKwtRec *recs = cls == 0 ? d.kwtRouted : cls == 1 ? d.kwtRoutedB : d.kwtRoutedC;
        const int last = (cls == 0 ? d.nA : cls == 1 ? d.nB : d.nC) - 1;
        const bool isSolo = solo != 0;
        int item, offA = g16 * GPA, capA = GPA;
        bool have;
        if (isSolo) {
          const int g = __ffs(solo) - 1; solo &= solo - 1u;
          item = __shfl(itemKeep, g * GA, 64); have = g16 == 0; offA = 0; capA = POOL;
        } else {
          item = bi * RA + g16;
          have = item <= last;
          if (cls != 0) {
            const int sel = kwt_pick(ovfMask, g16);
            have = sel >= 0; item = bi * (cls == 1 ? RB : RC) + (have ? sel : 0);
          }
          itemKeep = item;
        }
        const int st = kwt_reach<FULL, false, GA, KA, OA, true, true>(d, s, recs, item, have, last, offA, capA, sA, sB, sC, sD, sCtx[g16]);
        if (__ballot(st & 2) != 0ull) return;
        if (isSolo) { if (st & 1) mzr_raise(d, 60, recs[have ? item : last].r, s, 10); }
        else {
          const unsigned long long bal = __ballot(st & 1);
          solo = (unsigned)((bal & 1ull) | (((bal >> 16) & 1ull) << 1) | (((bal >> 32) & 1ull) << 2) | (((bal >> 48) & 1ull) << 3));
        }
        if (!solo && !ovfMask) break;
      }
    }
  }
  kwt_beat(d0, 3, 9);
  if (d0.kwtStat) {      // the wavefront's particle-traffic counters (one set per reach slot) go to the device counters
    grp_sync();
    const int l = mzr_lane();
    const unsigned long long *cs = (const unsigned long long *)(sCtx[l < RC ? l : 0] + 12);
    const unsigned long long a = wave_sum(l < RC ? cs[0] : 0ull), b = wave_sum(l < RC ? cs[1] : 0ull), c = wave_sum(l < RC ? cs[2] : 0ull);
    const unsigned long long nr = wave_sum(l < RC ? (cs[3] & 0xffffffffull) : 0ull), ne = wave_sum(l < RC ? (cs[3] >> 32) : 0ull);
    if (l == 0) { atomicAdd(&d0.kwtStat->w_in, a); atomicAdd(&d0.kwtStat->w_up, b); atomicAdd(&d0.kwtStat->w_out, c); atomicAdd(&d0.kwtStat->n_route, nr); atomicAdd(&d0.kwtStat->n_edges, ne); }
  }
  // the launch's duration on the device's own clock (mzr_get_sweep_clock): the last wavefront to leave leaves the latest time
  if (d0.swClock && mzr_lane() == 0) atomicMax(d0.swClock + 1, (unsigned long long)wall_clock64());
}
#ifdef MZR_KWT_TU_WIDE
}  // namespace mzr_kwt_wide
using mzr_kwt_wide::k_sweep_kwt;
#endif

#ifndef MZR_KWT_TU_WIDE

// Start of a KWT window in persistent mode: progress counters back to zero, and the headwater reaches
// (kwt_route.f90:181-205: REACH_Q = BASIN_QR(1), one sentinel particle) for every step of the window.
template <bool FULL>
__global__ void __launch_bounds__(256) k_kwt_window_init(MzrDev d, int tBegin, int tEnd, int first) {
  if (d.err->code != 0) return;      // a window that failed stays as it is
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (first && blockIdx.y == 0) {   // first slab of the first chunk of steps: also the per-reach bookkeeping
    for (int r = i; r < d.N; r += gridDim.x * blockDim.x) d.kwDone[r] = 0;
  }
  if (i >= d.nHead) return;
  const int r = d.kwtHead[i];
  const int N = d.N;
  const int per = (tEnd - tBegin + gridDim.y - 1) / gridDim.y;
  const int tB = tBegin + blockIdx.y * per, tE = min(tEnd, tB + per);
  const int es = (FULL && d.exportSlot) ? d.exportSlot[r] : -1;
  int t = tB;
  if (!d.kwHeadQ) {      // (otherwise k_hillslope_out has written the rows: mzr_device.h kwHeadQ)
    for (; t + 8 <= tE; t += 8) {      // eight rows in flight per lane
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(d.qlat + (size_t)(t + j + 1) * N + r);
#pragma unroll
      for (int j = 0; j < 8; ++j) d.Q[(size_t)(t + j) * N + r] = v[j];
    }
    for (; t < tE; ++t) d.Q[(size_t)t * N + r] = d.qlat[(size_t)(t + 1) * N + r];
  }
  if (es >= 0) for (int t2 = tB; t2 < tE; ++t2) d.exN[(size_t)t2 * d.nExp + es] = 0;
  if (first && blockIdx.y == 0) {
    d.inflow[r] = 0.0;
    if (d.kwN[r] != 1) { d.kwN[r] = 1; d.kwQT[MZR_PQ(MZR_KWI(0, r))] = -9999.0; d.kwQT[MZR_PT(MZR_KWI(0, r))] = -9999.0; d.kwTR[MZR_KWI(0, r)] = -9999.0; }
  }
}
// second, tiny kernel (after the zeroing above has finished): headwaters are complete for the whole window
__global__ void __launch_bounds__(256) k_kwt_head_done(MzrDev d) {
  if (d.err->code != 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nHead) d.kwDone[d.kwtHead[i]] = (unsigned long long)(unsigned)d.W;
}

// History sum of REACH_Q (histVars_data.f90:229-231 accumulates step by step): the window's rows added in step order.
__global__ void __launch_bounds__(256) k_accum_qsum(const double *Q, double *qsum, int N, int W, const MzrErr *err) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  if (err && err->code != 0) return;      // the rows of a window that failed (or was never routed) are not a result
  double a = qsum[r];
  int t = 0;
  for (; t + 16 <= W; t += 16) {      // sixteen rows in flight per lane (100 k lanes are 1.5 wavefronts per SIMD: latency, not bandwidth); added in step order
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(Q + (size_t)(t + j) * N + r);
#pragma unroll
    for (int j = 0; j < 16; ++j) a = a + v[j];
  }
  for (; t < W; ++t) a = a + Q[(size_t)t * N + r];
  qsum[r] = a;
}

void mzr_launch_accum_qsum(const double *Q, double *qsum, int N, int W, hipStream_t stream, const MzrErr *err) {
  hipLaunchKernelGGL(k_accum_qsum, dim3((N + 255) / 256), dim3(256), 0, stream, Q, qsum, N, W, err);
}

void mzr_launch_stage_kwt(const MzrDev &d, int s, int haBegin, int haEnd, int hbBegin, int hbEnd, int hcBegin, int hcEnd, int gnBegin, int gnEnd,
                          int ltBegin, int ltEnd, hipStream_t stream) {
  constexpr int POOL = MZR_KWT_POOL, POOLG = 1024;   // binary confluence: 20 + 2 + 2 * 19 = 60 entries per reach at most, 4 reaches
  const int nA = haEnd - haBegin, nB = hbEnd - hbBegin, nC = hcEnd - hcBegin, nLt = ltEnd - ltBegin, nGn = gnEnd - gnBegin;
  const bool full = d.lakeSlot || d.haloSlot || d.exportSlot || (d.is_flux_wm && d.wm);
  dim3 block(64);
  if (nA > 0 || nB > 0 || nC > 0 || nLt > 0) {
    const int nABlocks = (nA + 3) / 4, nBBlocks = (nB + 7) / 8, nCBlocks = (nC + 15) / 16;
    dim3 grid(nABlocks + nBBlocks + nCBlocks + (nLt + 63) / 64);
    if (full) hipLaunchKernelGGL((k_stage_kwt<true, false, POOL>), grid, block, 0, stream, d, s, haBegin, haEnd, nABlocks, hbBegin, hbEnd, nBBlocks, hcBegin, hcEnd, nCBlocks, ltBegin, ltEnd);
    else hipLaunchKernelGGL((k_stage_kwt<false, false, POOL>), grid, block, 0, stream, d, s, haBegin, haEnd, nABlocks, hbBegin, hbEnd, nBBlocks, hcBegin, hcEnd, nCBlocks, ltBegin, ltEnd);
  }
  if (nGn > 0) {   // confluences of more than two reaches: the reference's k-way merge on one lane of the group
    MzrDev dg = d; dg.kwtRouted = d.kwtGeneric;
    const int nBlocks = (nGn + 3) / 4;
    dim3 grid(nBlocks);
    if (full) hipLaunchKernelGGL((k_stage_kwt<true, true, POOLG>), grid, block, 0, stream, dg, s, gnBegin, gnEnd, nBlocks, 0, 0, 0, 0, 0, 0, 0, 0);
    else hipLaunchKernelGGL((k_stage_kwt<false, true, POOLG>), grid, block, 0, stream, dg, s, gnBegin, gnEnd, nBlocks, 0, 0, 0, 0, 0, 0, 0, 0);
  }
}

#endif      // MZR_KWT_TU_WIDE
// ---- persistent sweep, host side
static bool kwt_full(const MzrDev &d) { return d.lakeSlot || d.haloSlot || d.exportSlot || (d.is_flux_wm && d.wm); }

#ifndef MZR_KWT_TU_WIDE
// Wavefronts of k_sweep_kwt the device really holds at once.  The grid of a sweep must not exceed this: a persistent
// kernel with workgroups still waiting for a slot was measured (profiles/r03_soak.md) to freeze, now and then and for
// as long as the others keep running, the vector-memory instructions of some of its last-launched resident wavefronts
// -- every other wavefront then waits on them and the sweep's watchdog fires (ierr 93).  The occupancy query is one
// workgroup per CU high for this kernel (17 against 16), so the number is measured: the kernel itself is launched in
// census mode (sEnd < 0: every wavefront counts itself in, stays 300 us, counts itself out; the peak is the answer).
// cnt: two ints of device memory (swHead + 128).  Measured once per process, device and kernel flavour.
template <bool FULL>
static int kwt_sweep_census(const MzrDev &d, hipStream_t stream, int cus) {
  int perCu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_sweep_kwt<FULL, MZR_KWT_POOL>, 64 * MZR_KWT_WG, 0) != hipSuccess) return 0;
  const int api = cus * perCu * MZR_KWT_WG;      // wavefronts
  int peak[2] = {0, 0};
  int *cnt = d.swHead + 8 * 16;
  if (hipMemsetAsync(cnt, 0, 2 * sizeof(int), stream) != hipSuccess) return 0;
  const int grid = api + api / 4;
  hipLaunchKernelGGL((k_sweep_kwt<FULL, MZR_KWT_POOL>), dim3((grid + MZR_KWT_WG - 1) / MZR_KWT_WG), dim3(64 * MZR_KWT_WG), 0, stream, d, 0, -1);
  if (hipStreamSynchronize(stream) != hipSuccess) return 0;
  if (hipMemcpy(peak, cnt, sizeof peak, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  const int cap = peak[1] > 0 ? std::min(api, peak[1]) : 0;
  return 2 * cap >= api ? cap : -cap;      // (negative: a census far below the occupancy query ran beside other work -- not to be remembered)
}
int mzr_sweep_kwt_capacity(bool full, const MzrDev &d, hipStream_t stream) {
  static int cached[16][2];
  static std::mutex mu;      // handles of several host threads share the cache; the census itself must not run twice at once either
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev >= 0 && dev < 16 && cached[dev][full]) return cached[dev][full];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const int c = full ? kwt_sweep_census<true>(d, stream, cus) : kwt_sweep_census<false>(d, stream, cus);
  if (dev >= 0 && dev < 16 && c > 0) cached[dev][full] = c;
  return c > 0 ? c : -c;
}

// headwater reaches for steps [tBegin, tEnd) of the window; tBegin == 0 also resets the progress counters
void mzr_launch_kwt_window_init(const MzrDev &d, int tBegin, int tEnd, hipStream_t stream) {
  const int first = tBegin == 0;
  const int n = first ? (d.nHead > d.N / 8 ? d.nHead : d.N / 8) : d.nHead;
  if (n < 1 && !first) return;
  const int slabs = tEnd - tBegin >= 64 ? 8 : 1;
  dim3 grid((n + 255) / 256 > 0 ? (n + 255) / 256 : 1, slabs), block(256);
  if (kwt_full(d)) hipLaunchKernelGGL(k_kwt_window_init<true>, grid, block, 0, stream, d, tBegin, tEnd, first);
  else hipLaunchKernelGGL(k_kwt_window_init<false>, grid, block, 0, stream, d, tBegin, tEnd, first);
  if (first && d.nHead > 0) hipLaunchKernelGGL(k_kwt_head_done, dim3((d.nHead + 255) / 256), block, 0, stream, d);
}

__global__ void k_sweep_heads(MzrDev d, int sBegin) {
  if (d.err->code != 0) return;
  if (threadIdx.x < 8) d.swHead[threadIdx.x * 16] = d.swP[sBegin * 8 + threadIdx.x];
  if (d.swClock && threadIdx.x >= 12 && threadIdx.x < 14) d.swClock[threadIdx.x - 12] = 0ull;
  mzr_sweep_join_reset(d.swHead);
}

#endif      // MZR_KWT_TU_WIDE
// evStart / evStop (profiling): attached to the sweep's own dispatch (hipExtLaunchKernelGGL), not recorded as markers around it --
// marker packets in front of and behind a persistent launch were measured to slow some windows by a quarter (446 -> 560 ms, a
// pattern with a period of eight windows; without events, and with events attached to the dispatch, every window takes 447 ms:
// profiles/r04_experiments.md)
template <bool FULL>
static void kwt_sweep_launch(const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop) {
  const dim3 grid((nWaves + MZR_KWT_WG - 1) / MZR_KWT_WG), block(64 * MZR_KWT_WG);
  if (evStart && evStop) hipExtLaunchKernelGGL((k_sweep_kwt<FULL, MZR_KWT_POOL>), grid, block, 0, stream, evStart, evStop, 0, d, sBegin, sEnd);
  else hipLaunchKernelGGL((k_sweep_kwt<FULL, MZR_KWT_POOL>), grid, block, 0, stream, d, sBegin, sEnd);
}
#ifdef MZR_KWT_TU_WIDE
void mzr_launch_sweep_kwt_wide(const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop) {
  if (kwt_full(d)) kwt_sweep_launch<true>(d, nWaves, sBegin, sEnd, stream, evStart, evStop); else kwt_sweep_launch<false>(d, nWaves, sBegin, sEnd, stream, evStart, evStop);
}
#else
void mzr_launch_sweep_kwt_wide(const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop);      // kernels_kwt_wide.hip
// kcWide: the flavour with MZR_KWT_KC_WIDE particle slots per lane of the 4-lane class (the class lists must have been cut for it:
// mzr_kwt_class_caps)
void mzr_launch_sweep_kwt(const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop, int kcWide) {
  if (nWaves < 1 || sEnd <= sBegin) return;
  hipLaunchKernelGGL(k_sweep_heads, dim3(1), dim3(64), 0, stream, d, sBegin);
  const bool full = kwt_full(d);
  if (kcWide) mzr_launch_sweep_kwt_wide(d, nWaves, sBegin, sEnd, stream, evStart, evStop);
  else if (full) kwt_sweep_launch<true>(d, nWaves, sBegin, sEnd, stream, evStart, evStop);
  else kwt_sweep_launch<false>(d, nWaves, sBegin, sEnd, stream, evStart, evStop);
}
// entries a class-B / class-C group holds (the host's regrouping stays below them); kcWide: in the sweep flavour with MZR_KWT_KC_WIDE slots per lane
int mzr_kwt_class_caps(int *capB, int *capC, int kcWide) {
  constexpr int GPB = MZR_KWT_POOL / KwtCls::RB, GPC = MZR_KWT_POOL / KwtCls::RC;
  const int kc = kcWide ? MZR_KWT_KC_WIDE : KwtCls::KC;
  *capB = KwtCls::GB * MZR_KWT_KTB - 1 < GPB ? KwtCls::GB * MZR_KWT_KTB - 1 : GPB;
  *capC = KwtCls::GC * kc - 1 < GPC ? KwtCls::GC * kc - 1 : GPC;
  return MZR_KWT_POOL / KwtCls::RA;
}
#endif      // MZR_KWT_TU_WIDE
