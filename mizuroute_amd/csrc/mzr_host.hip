// Host side of libmzr_hip.so: the C-ABI of include/mzr.h.
//
// Replaces, for one routing domain on one MI355X:
//   main_route / route_network          route/build/src/main_route.f90:29-409
//   put_data_struct (AoS -> device SoA) route/build/src/process_ntopo.f90:354-504
//   init_state_data (cold start)        route/build/src/init_model_data.f90:399-505
//   omp_domain_decomposition            route/build/src/domain_decomposition.f90:168-445
//       -> replaced by a stage schedule: stage(r) = Dmax - hops(r -> outlet).  Every edge of the
//          river tree then spans exactly one stage, which is what lets the sweep be skewed in
//          time: launch s advances reach r through step s - stage(r) of the window.
// There is no CPU fallback: every compute entry point launches HIP kernels.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <queue>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <rccl/rccl.h>     // types only: the library is loaded at run time (mzr_comm_*)

#include "../../include/mzr.h"
#include "mzr_device.h"

void mzr_launch_basin(const MzrDev &d, hipStream_t stream);
void mzr_launch_basin_chunk(const MzrDev &d, int tBegin, int tEnd, hipStream_t stream);
void mzr_launch_basin_state(const MzrDev &d, hipStream_t stream);
void mzr_launch_basin_solute(const MzrDev &d, hipStream_t stream);
void mzr_launch_tracer_stage(int method, const MzrDev &d, int s, int rBegin, int rEnd, hipStream_t stream);
int mzr_sweep_route_capacity(int method, const MzrDev &d, hipStream_t stream);
void mzr_launch_sweep_route(int method, const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream);
void mzr_launch_lake_forcing(const MzrDev &d, const int *lakeReachInt, const double *evap, const double *precip,
                             double *lakeEvap, double *lakePrecip, int nSteps, hipStream_t stream);
void mzr_launch_stage(int method, const MzrDev &d, int s, int rBegin, int rEnd, hipStream_t stream);
void mzr_launch_stage_pair(int method, const MzrDev &a, int sA, int rBeginA, int rEndA, const MzrDev &b, int sB, int rBeginB, int rEndB, hipStream_t stream);
int mzr_remap_ld(int nSteps);
void mzr_launch_remap(int H, int nSteps, int nSrc, const int *rowStart, const int *rowCnt, const int *srcIdx,
                      const double *weight, const double *src, double *srcT, double *dst, hipStream_t stream);
void mzr_launch_sort_flux(int H, int nSteps, int nSrc, const int *srcOf, int removeNegatives, const double *src, double *dst,
                          hipStream_t stream);
void mzr_launch_stage_kwt(const MzrDev &d, int s, int haBegin, int haEnd, int hbBegin, int hbEnd, int hcBegin, int hcEnd, int gnBegin, int gnEnd,
                          int ltBegin, int ltEnd, hipStream_t stream);

void mzr_launch_accum_qsum(const double *Q, double *qsum, int N, int W, hipStream_t stream, const MzrErr *err = nullptr);
int mzr_sweep_kwt_capacity(bool full, const MzrDev &d, hipStream_t stream);
int mzr_chan_table_doubles();
void mzr_launch_chan_table(const MzrDev &d, double *tab, hipStream_t stream);
int mzr_kwt_class_caps(int *capB, int *capC, int kcWide = 0);
void mzr_launch_kwt_window_init(const MzrDev &d, int tBegin, int tEnd, hipStream_t stream);
void mzr_launch_sweep_kwt(const MzrDev &d, int nWaves, int sBegin, int sEnd, hipStream_t stream, hipEvent_t evStart, hipEvent_t evStop, int kc);

namespace {

template <typename T> struct DBuf {
  T *p = nullptr; size_t n = 0;
  void alloc(size_t cnt) {
    free();
    n = cnt;
    if (cnt) { if (hipMalloc((void **)&p, cnt * sizeof(T)) != hipSuccess) { p = nullptr; n = 0; throw std::string("hipMalloc failed"); } }
  }
  // device memory that is not kept in the caches on its way in (experiment, MZR_H2D_UNCACHED=1: the staging buffers of host forcing)
  void allocStaging(size_t cnt) {
    static const bool unc = getenv("MZR_H2D_UNCACHED") && atoi(getenv("MZR_H2D_UNCACHED")) != 0;
    if (!unc) { alloc(cnt); return; }
    free();
    n = cnt;
    if (cnt) { if (hipExtMallocWithFlags((void **)&p, cnt * sizeof(T), hipDeviceMallocUncached) != hipSuccess) { p = nullptr; n = 0; throw std::string("hipExtMallocWithFlags failed"); } }
  }
  // zero(): set-up calls, no stream -- the fill runs on the null stream and the call returns when it is done.  (Round 6: it used to
  // return at once.  A handle's own stream does not wait for the null stream where it was made non-blocking -- the high-priority
  // stream of a mainstem domain, mzr_set_boundary -- so the first kernels of such a domain could overtake the fill of a buffer made
  // by a later set-up call (mzr_set_da, mzr_set_tracer), and the fill then wiped what they had written: a partitioned run with gauge
  // observations differed from the whole network once in a few runs on a fresh box.)  zero(stream): ordered in that stream.
  void zero() { if (p) { (void)hipMemsetAsync(p, 0, n * sizeof(T), 0); (void)hipStreamSynchronize(0); } }
  void zero(hipStream_t s) { if (p) (void)hipMemsetAsync(p, 0, n * sizeof(T), s); }
  void upload(const std::vector<T> &v) { alloc(v.size()); if (!v.empty()) (void)hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); }
  void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void swap(DBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
  ~DBuf() { free(); }
};

struct RouteBufs {
  int method = -1;
  DBuf<double> Q;                                   // [maxWindow][N]
  DBuf<double> Qalt;                                // the rows of the window before, while its last launches are kept back (overlapping windows)
  DBuf<double> vol, vol0, inflow, ele, floodvol, wb, qsum, wmact;   // [N]
  DBuf<double> hInflow, hEle, hFlood;               // [N] history sums beyond discharge (mzr_set_history)
  DBuf<double> mol;                                 // [nMol][N]
  DBuf<unsigned short> mcSub; long long mcWindows = 0;   // Muskingum-Cunge: sub-steps per reach as the kernel leaves them
  int permN = 0, nHeavyPos = 0;      // heavy lane positions behind the block-wise ones (mzr_device.h)
  DBuf<int> lanePerm; bool havePerm = false;           // reaches of every aligned block of 256 dealt to its wavefronts by loop trip count (mzr_device.h)
  DBuf<double> imQ;                                 // [maxWindow][nHalo] imported REACH_Q of halo reaches
  DBuf<double> imQAlt;                              // ... of the NEXT window while the last launches of the window before are kept back (overlapping windows of a mainstem domain)
  DBuf<double> lakeMut, lakeRing, lakeRingD; DBuf<int> lakeHead, lakeHeadD;   // per-method mutable Hanasaki parameters / inflow and demand memory
  DBuf<int> rtDone, rtHead;                         // persistent sweep of an Eulerian method: progress per reach, ticket counters
  DBuf<double> solFlux, solMass, trVol0;            // constituent routing (mzr_set_tracer): [maxWindow][N], [N], [maxWindow][N]
  DBuf<double> qobs, qerr; DBuf<int> qelapsed;      // [N] direct insertion (mzr_set_da): RCHFLX%Qobs, ROUTE%Qerror, RCHFLX%Qelapsed
  int rtCap = 0;                                    // wavefronts the device holds of this method's sweep kernel
  bool rtCapTried = false;
  long long nLaunches = 0, reachSteps = 0, meanSteps = 0; double kernel_ms = 0.0, kernel_ms_min = 0.0, kernel_ms_max = 0.0;   // meanSteps: steps summed into qsum since its last reset
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events; size_t evUsed = 0;
};

__global__ void k_gather_rows(const double *src, double *dst, const int *ext2int, int N, int rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (e < N && t < rows) dst[(size_t)t * N + e] = src[(size_t)t * N + ext2int[e]];
}

__global__ void k_scatter_rows(const double *src, double *dst, const int *ext2int, int N, int rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (e < N && t < rows) dst[(size_t)t * N + ext2int[e]] = src[(size_t)t * N + e];
}

// (overlapping windows: the window before keeps its rows, the new one has its own)
// the state a queue of windows can be taken back to (retry of a window whose sweep gave up): copied at the start of every window
// unless an earlier window of the queue has failed -- so what is kept is the state at the start of the FIRST failed window
__global__ void __launch_bounds__(256) k_snapshot(const MzrErr *err, int *sN, const int *kwN, size_t nN, double *sQ, const double *kwQ, size_t nQ,
                                                   double *sTR, const double *kwTR, size_t nTR, double *sS, const double *qsum, double *sH, const double *hIn, size_t nR) {
  if (err->code != 0) return;
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < nQ; i += stride) sQ[i] = kwQ[i];
  for (size_t i = i0; i < nTR; i += stride) sTR[i] = kwTR[i];
  for (size_t i = i0; i < nN; i += stride) sN[i] = kwN[i];
  for (size_t i = i0; i < nR; i += stride) { sS[i] = qsum[i]; if (sH) sH[i] = hIn[i]; }
}
__global__ void k_carry_qlat2(double *dst, const double *src, int lastW, int N, const int *haloSlot, const MzrErr *err) {
  if (err->code != 0) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  if (haloSlot && haloSlot[r] >= 0) return;
  dst[r] = src[(size_t)lastW * N + r];
}
__global__ void k_carry_qlat(double *qlat, int lastW, int N, const int *haloSlot, const MzrErr *err) {
  if (err->code != 0) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  if (haloSlot && haloSlot[r] >= 0) return;
  qlat[r] = qlat[(size_t)lastW * N + r];
}

// boundary record (include/mzr.h), round 6: what the importer READS and nothing else (mpi_process.f90:1245-1329 ships the outlet
// fluxes only) --
//   header[4] | Q[R][W][nB] | KWT among the methods: qlat[W+1][nB] | obN[W][nB] | obQ[W][21][nB] | obT[W][21][nB] | constituent on: solute flux[R][W][nB]
// (rounds 3-5 shipped the particle rows, their counts and the hillslope series of every domain: 3.2-3.8 GB per 2 048-step window
// into rank 0 of the Eulerian configurations where 0.07-0.15 GB are read)
// header = {magic = layout version, nRoutes, steps, reaches + 2^30 while the constituent is on + 2^31 with the KWT part}: what the
// sender packed, checked by the receiver (a sender and a receiver that disagree on any of them -- e.g. mzr_set_tracer on one side
// only -- would otherwise read each other's records at the wrong offsets without a word)
#define MZR_REC_HDR 4
#define MZR_REC_MAGIC 20260930.0
__host__ __device__ inline double recTag(int nB, int hasKwt, int tracer) { return (double)nB + (tracer ? 1073741824.0 : 0.0) + (hasKwt ? 2147483648.0 : 0.0); }
__host__ __device__ inline long long recKwtPart(long long W, long long nB, int hasKwt) { return hasKwt ? (W + 1) * nB + W * nB + 2 * W * MZR_OB_CAP * nB : 0; }
struct RecView { double *Q, *ql, *n, *oq, *ot; };
__host__ __device__ inline RecView recView(double *rec, int R, int W, int nB) {      // (ql .. ot exist with the KWT part only)
  RecView v;
  v.Q = rec + MZR_REC_HDR; v.ql = v.Q + (size_t)R * W * nB; v.n = v.ql + (size_t)(W + 1) * nB;
  v.oq = v.n + (size_t)W * nB; v.ot = v.oq + (size_t)W * MZR_OB_CAP * nB;
  return v;
}
struct QPtrs { const double *p[6]; };
struct QPtrsW { double *p[6]; };

// grid: x over export slots, y over steps 0..W (row W only carries the last BASIN_QR row)
__global__ void k_pack_boundary(double *rec, int R, int W, int nB, int N, const int *expInt, QPtrs Q, const double *qlat,
                                const int *exN, const double *exOQ, const double *exOT, int hasKwt, int tracer) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= nB) return;
  if (b == 0 && t == 0) { rec[0] = MZR_REC_MAGIC; rec[1] = (double)R; rec[2] = (double)W; rec[3] = recTag(nB, hasKwt, tracer); }
  const RecView v = recView(rec, R, W, nB);
  const int r = expInt[b];
  if (hasKwt) v.ql[(size_t)t * nB + b] = qlat[(size_t)t * N + r];      // BASIN_QR(0:1) of the outlet's basin: read by the KWT merge downstream only
  if (t >= W) return;
  for (int m = 0; m < R; ++m) v.Q[((size_t)m * W + t) * nB + b] = Q.p[m][(size_t)t * N + r];
  if (!hasKwt) return;
  const int n = exN[(size_t)t * nB + b];
  v.n[(size_t)t * nB + b] = (double)n;
  for (int k = 0; k < MZR_OB_CAP; ++k) {
    const size_t o = ((size_t)t * MZR_OB_CAP + k) * nB + b;
    const bool have = n > 0 && k <= n;
    v.oq[o] = have ? exOQ[o] : 0.0; v.ot[o] = have ? exOT[o] : 0.0;
  }
}

__global__ void k_unpack_boundary(const double *rec, int R, int W, int nB, int N, int nHalo, int haloBase, const int *haloInt,
                                  QPtrsW imQ, double *qlat, int *imN, double *imOQ, double *imOT, int hasKwt, int tracer, MzrErr *err) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= nB) return;
  // the record must be what this domain expects (ierr 20 at the next synchronisation; nothing of it is used)
  if (!(rec[0] == MZR_REC_MAGIC && rec[1] == (double)R && rec[2] == (double)W && rec[3] == recTag(nB, hasKwt, tracer))) {
    if (b == 0 && t == 0 && atomicCAS(&err->code, 0, 20) == 0) { err->reach = -1; err->step = (int)rec[2]; err->where = 30; }
    return;
  }
  const RecView v = recView(const_cast<double *>(rec), R, W, nB);
  const int hs = haloBase + b;
  const int r = haloInt[hs];
  if (hasKwt) qlat[(size_t)t * N + r] = v.ql[(size_t)t * nB + b];
  if (t >= W) return;
  for (int m = 0; m < R; ++m) imQ.p[m][(size_t)t * nHalo + hs] = v.Q[((size_t)m * W + t) * nB + b];
  if (hasKwt) {
    imN[(size_t)t * nHalo + hs] = (int)v.n[(size_t)t * nB + b];
    for (int k = 0; k < MZR_OB_CAP; ++k) {
      const size_t o = ((size_t)t * MZR_OB_CAP + k) * nB + b, oh = ((size_t)t * MZR_OB_CAP + k) * nHalo + hs;
      imOQ[oh] = v.oq[o]; imOT[oh] = v.ot[o];
    }
  }
}

// constituent routing in partitioned domains: reach_solute_flux of the export reaches, [R][W][nB] behind the record proper
__global__ void k_pack_solute(double *sf, int R, int W, int nB, int N, const int *expInt, QPtrs F) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= nB || t >= W) return;
  const int r = expInt[b];
  for (int m = 0; m < R; ++m) sf[((size_t)m * W + t) * nB + b] = F.p[m] ? F.p[m][(size_t)t * N + r] : 0.0;
}
__global__ void k_unpack_solute(const double *rec, const double *sf, int R, int W, int nB, int N, int haloBase, const int *haloInt, QPtrsW F, int hasKwt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (b >= nB || t >= W) return;
  // (a record this domain does not expect: k_unpack_boundary has raised ierr 20; nothing of it is used here either)
  if (!(rec[0] == MZR_REC_MAGIC && rec[1] == (double)R && rec[2] == (double)W && rec[3] == recTag(nB, hasKwt, 1))) return;
  const int r = haloInt[haloBase + b];
  for (int m = 0; m < R; ++m) if (F.p[m]) F.p[m][(size_t)t * N + r] = sf[((size_t)m * W + t) * nB + b];
}

}  // namespace

struct mzr_domain {
  mzr_config cfg;
  hipStream_t stream = nullptr;
  bool highPriority = false;
  hipStream_t basinStream = nullptr;          // hillslope pre-pass of the later parts of a window, behind the sweep
  hipStream_t routeStream[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // methods 2.. of a multi-method run
  hipEvent_t routeEvent[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // [0] start of the sweeps, [ix] end of method ix
  std::vector<hipEvent_t> basinEvents;        // [0] window start, [c] chunk c ready, [last] state ready
  std::string msg;
  int N = 0, H = 0, nStages = 0, maxStageWidth = 0, wk = 64;
  bool haveNet = false, haveState = false;
  std::vector<int> int2ext, ext2int, stageStart, reachId;
  std::vector<uint8_t> h_nGood, h_nUp;
  // device topology / params
  DBuf<int> sigma, upStart, hruOff, hruIdx, d_ext2int;
  DBuf<uint8_t> nUp, nGood, isOutlet;
  DBuf<uint32_t> goodMask;
  DBuf<double> hruW;
  DBuf<double> par[11];
  std::vector<double> h_slope, h_mann;   // host copies (internal order) for derived KWT constants
  DBuf<double> kwK, kwCW;
  static const char *parName(int i) {
    static const char *n[11] = {"R_SLOPE", "R_MAN_N", "R_WIDTH", "R_DEPTH", "RLENGTH", "R_STORAGE", "SIDE_SLOPE",
                                "FLDP_SLOPE", "BASAREA", "TOTAREA", "MINFLOW"};
    return n[i];
  }
  // unit hydrographs
  int ntdhBas = 0, maxtdh = 0;
  DBuf<double> fracFuture, fracPad, uh, irfQ;
  DBuf<uint16_t> ntdh;
  std::vector<int> uhOff;
  // window buffers
  DBuf<double> runoffW, runoffW2, qi, qlat, qr0Last, basS[2], scratchOut, wm;
  DBuf<float> runoffF[2];                        // single-precision forcing windows of mzr_run_async_f32 as they arrive, widened into runoffW / runoffW2
  hipStream_t copyStream = nullptr;             // host -> device forcing windows of mzr_run_async, behind the sweep of the window before
  hipEvent_t rwCopied[2] = {nullptr, nullptr}, rwRead[2] = {nullptr, nullptr};
  hipEvent_t exportDone = nullptr;              // recorded behind the last mzr_export_boundary_dev (main stream): what mzr_comm_send waits for
  hipEvent_t exportPrevDone = nullptr;          // ... behind the last mzr_export_boundary_prev_dev (expStream): an event of its own, so that an export on the main
                                                // stream in between cannot take the wait for the pack on expStream away
  // Export of the window BEFORE the last one (mzr_export_boundary_prev_dev): while the last launches of window k are kept back for window
  // k + 1 (overlapping windows), the rows of window k - 1 sit complete in the second set of rows from launch nS - 1 of window k on
  hipEvent_t sweepGo = nullptr;                 // recorded on the KWT stream right in front of the last sweep launch: that launch is eligible (mzr_run_async*: the next window's copy starts behind it)
  hipStream_t expStream = nullptr;              // the pack kernel of such an export runs here, behind prevRowsEv only -- not behind the rest of window k
  hipEvent_t prevRowsEv[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // per method: the rows of the window before are complete (recorded inside run_window)
  int prevW = 0;                                // steps of the window whose rows are the second set
  bool prevInAlt = false;                       // the second set of rows holds the window before the last one, untouched
  bool exportOnAux = false;                     // an export on expStream is in flight: the next window (which writes those rows) waits for exportDone
  bool rwUsed[2] = {false, false};
  int rwCur = 0;
  hipEvent_t rwOther = nullptr; bool rwOtherSet = false;   // a window queued by another entry point (mzr_run_src_dev) still reads runoffW

  int wmSteps = 0;
  int basCur = 0;
  int lastW = 0;
  bool havePrevQlat = false;
  // Overlapping windows of the Eulerian methods (kernels_route.hip, k_stage_pair): the launches in which a window drains
  // (s >= W) are kept back and issued together with the first launches of the next window, or on their own as soon as
  // anybody asks for a result (flushTail: every entry point but the run calls).  The window kept back owns its rows
  // (qlat / qi / Q / lake forcing: the *Alt buffers, swapped in and out); tail.d[ix] are the device views it was launched with.
  struct { bool pending = false; int W = 0, next = 0; MzrDev d[6]; } tail;      // (W: launches of the window = its steps in blocks of d.stepBlock; next: the first of its nStages - 1 kept-back launches not yet issued)
  DBuf<double> qiAlt, qlatAlt, lakeEvapAlt, lakePrecipAlt; DBuf<int> calMonthAlt, calDayAlt, calDoyAlt;
  bool lakeNextInAlt = false;                    // mzr_set_lake_forcing wrote the NEXT window's lake forcing into the *Alt buffers
  bool imNextInAlt = false;                      // mzr_import_boundary_dev wrote the NEXT window's halo discharge into imQAlt (run_window swaps it in)
  hipEvent_t importDone = nullptr;               // recorded behind the last mzr_import_boundary_dev: what mzr_wait_import waits for
  long long pairLaunches = 0, tailFlushes = 0;
  // A KWT window that ended with ierr 93 (the persistent sweep gave up waiting: DESIGN.md 2.4) is routed again through one launch
  // per stage -- when the state it started from is still known: the at-rest particles are copied aside before every sweep
  // (580 bytes per reach), and the failed window must be the only one queued since the last synchronisation (later windows
  // see the error, skip their sweeps, and their hillslope pre-pass has moved on: nothing to go back to).
  // Windows queued since the handle was last synchronised, each with what it takes to route it again and the host-side bookkeeping
  // as it was before the window was queued (retry of a KWT sweep that gave up, mzr_sync)
  struct Book { int basCur = 0, lastW = 0; long long stepsDone = 0, totalSteps = 0, histSteps = 0, kwtWindows = 0, kwtStepsSince = 0, kwtHeadSteps = 0; long long reachSteps[6] = {0}, meanSteps[6] = {0}, nLaunches[6] = {0}; };
  struct QWin { int W = 0; double t_start = 0.0, T1_single = 0.0; const double *runoff = nullptr; bool replayable = false, snap = false; Book before; };
  struct { std::vector<QWin> q; bool armed = false; int failAt = -1; long long seen = 0; int depth = 0; } retry;
  bool nextReplayable = false;      // the window being queued has its forcing in the caller's device memory (mzr_run_dev)
  DBuf<int> snapN; DBuf<double> snapQ, snapTR, snapQsum, snapHIn;
  long long sweepRetries = 0;
  DBuf<unsigned long long> swClock; long long swClockN = 0;      // {first wavefront in, last wavefront out} of the last MZR_CLOCK_LOG sweep launches (device clock)
  // kwt
  DBuf<int> kwN, obN, kwtLight;
  DBuf<uint8_t> kwHeadFlag;      // [N] 1 = headwater reach of the KWT sweep (k_hillslope_out writes its discharge rows)
  DBuf<MzrKwtRec> kwtRouted, kwtRoutedB, kwtRoutedC, kwtGeneric;
  DBuf<MzrKwtRec> kwtRoutedAll, kwtRoutedBAll, kwtRoutedCAll;   // classes A / B / C over all stages, heaviest first: used by launches in which every stage is active
  bool kwtAllValid = false;
  int swKcWide = 0;               // the sweep runs the flavour with MZR_KWT_KC_WIDE particle slots per lane of the 4-lane class (class C was cut for it: kwt_regroup)
  bool swHeavyFirst = false;      // the sweep's items in order of weight regardless of stage (class lists = the *All arrays): MZR_KWT_HEAVY_FIRST
  // persistent sweep (k_sweep_kwt): items dealt to wavefronts, progress counters
  DBuf<unsigned long long> kwDone; DBuf<int> down, swItem, swLo, swHi, swRA, swP, swHead, kwtHead, kwtDepLight;
  DBuf<int> swBeat;                                // [swCap][8] per-wavefront record of the sweep (MZR_SWEEP_DEBUG=1)
  DBuf<int> rtItemR, rtItemInfo, rtRA, rtP;        // items of the Eulerian sweeps (k_sweep_route) and their per-launch tables
  int tracer = 0, solSteps = 0, solCur = 0; double time_conv_solute = 1.0, mass_conv_solute = 1.0;      // constituent routing (mzr_set_tracer / mzr_set_solute)
  DBuf<double> solSrc, solInst, basSol, solS[2]; hipEvent_t trEvent = nullptr;
  int qmod = 0, qBlendPeriod = 10, QerrTrend = 1, nGauge = 0, obsSteps = 0;      // direct insertion of gauge observations (mzr_set_da / mzr_set_obs)
  DBuf<int> gaugeFirst, gaugeNext, obsHave; DBuf<double> obsVal;
  std::vector<int> h_rtStage; int rtItems = 0, rtTablesW = -1, rtMaxAct = 0;
  std::vector<int> h_down, h_kwtHead, h_kwtDepLight, h_swLo, h_swHiMax;
  std::vector<int> h_sigma, h_swCode, h_swP, h_swRA;   // host copies for the stall report of a sweep that gave up (code 93)
  std::vector<MzrKwtRec> h_kwtGeneric, h_swA, h_swB, h_swC;   // class lists of the sweep, host copies (stage order)
  int swWaves = 0, swCap = 0, swItems = 0, swTablesW = -1;
  long long kwtHeadSteps = 0;                   // headwater reach-steps filled in by the bulk kernel while the traffic counters were on
  std::vector<MzrKwtRec> h_kwtRouted;           // host copy of the routed list, stage-major (regrouped into classes A / B by load now and then)
  std::vector<int> kwtStageOff, kwtBOff, kwtCOff;   // [nStages+1] stage offsets in h_kwtRouted / in the class-B and class-C lists (kwtRoutedOff: class A)
  long long kwtWindows = 0, kwtStepsSince = 0; // KWT windows run since mzr_init_state, steps since the last regrouping
  std::vector<int> kwtRoutedOff, kwtGenericOff, kwtLightOff;   // [nStages+1] offsets of each stage in the two lists
  DBuf<double> kwQ, kwTR, obQ;      // kwQ: [N][stride] {Q, TI} pairs; obQ: [2][N][stride] {Q, exit time} pairs
  DBuf<MzrKwtStat> kwtStat;
  DBuf<unsigned long long> dbgCycles;
  // lakes
  int nLake = 0, LakeInputOption = 0, calendarId = 0, lakeL = 0, lakeLD = 0, lakeSteps = 0, volJumpstart = 0, wmVolSteps = 0;
  bool anyLakeTarget = false;
  DBuf<int> lakeTarg; DBuf<double> lakeWmVol;
  std::vector<double> h_lakePar; std::vector<int> h_lakeModel, h_lakeSlot;
  DBuf<int> lakeSlot, lakeModel, lakeReachInt, calMonth, calDay, calDoy;
  DBuf<double> lakePar, lakeEvap, lakePrecip, lakeFE, lakeFP;
  // partition boundary
  int nExp = 0, nHalo = 0;
  std::vector<int> h_expInt, h_haloInt, h_haloGood, h_haloSlot;
  DBuf<int> haloSlot, exportSlot, expInt, haloInt, imN, exN;
  DBuf<double> imOQ, imOT, exOQ, exOT;
  // forcing remap (0 = none, 1 = polygon vector, 2 = grid, 3 = sort_flux)
  int remapKind = 0, remapSrc = 0, remapRemoveNeg = 1;
  DBuf<int> rmRowStart, rmRowCnt, rmSrcIdx;
  DBuf<double> rmWeight, rmScratch;   // rmScratch: the window transposed to [cell][step]
  size_t rmScratchLen = 0;
  DBuf<MzrErr> err;
  RouteBufs route[6];
  bool profiling = false;       // HIP events around every stage launch
  hipStream_t timerStream = nullptr; hipEvent_t timerGate[2] = {nullptr, nullptr};      // the KWT sweep's timing events live here (run_window)
  bool countTraffic = false;    // KWT particle-traffic counters (atomics: not for timed runs)
  long long stepsDone = 0, totalSteps = 0;
  // steps handed over one at a time (mzr_step with stepBatch > 1): rows wait in page-locked host memory (two buffers, filled
  // alternately) until the batch is full or somebody asks for a result, and are then routed as one window
  double *stepHost[2] = {nullptr, nullptr}; hipEvent_t stepCopied[2] = {nullptr, nullptr}; bool stepInFlight[2] = {false, false};
  int stepCur = 0, stepN = 0, stepCap = 0; double stepT0 = 0.0, stepT1 = 0.0;
  // ... and so do the per-step rows of the per-window inputs handed over beside them (one-step calls of mzr_set_lake_forcing,
  // mzr_set_wm_flux, mzr_set_wm_vol, mzr_set_solute, mzr_set_obs): the row for the COMING step waits in `next`, mzr_step
  // moves it behind the rows of the pending steps, flushSteps hands them to the window setters in one piece.  A row whose
  // step never comes as mzr_step is handed to its setter as it is the next time anything is asked of the handle.
  struct StepRows { bool staged = false; std::vector<double> next[2], rows[2]; std::vector<int> nextI[3], rowsI[3]; };
  enum { SR_LAKE = 0, SR_WMFLUX, SR_WMVOL, SR_SOLUTE, SR_OBS, SR_KINDS };
  StepRows sr[SR_KINDS];
  unsigned srMask = 0;            // kinds whose rows the pending steps carry
  bool srAny = false;             // a row is staged for the coming step
  bool srApplying = false;        // flushSteps is handing rows to the setters (they neither stage nor flush then)
  int histFlags = 0;                            // MZR_H_*: which history sums beyond discharge are kept
  long long histSteps = 0;                      // steps in the runoff sums since the last reset
  DBuf<double> chanTab; bool chanDirty = true;      // channel table of the Eulerian solvers (kernels_route.hip d_chan), made again when a parameter changes
  DBuf<double> hInst, hDlay, hBas;              // [N], [N], [H] sums of BASIN_QI, BASIN_QR(1), basin runoff
};

namespace {

int fail(mzr_handle h, int code, const std::string &m) { h->msg = m; return code; }
// Window-sized staging buffers that only some entry points need are made on first use (a 625 k-reach domain with windows of
// 3072 steps would otherwise carry 30 GB it never touches): the library's own forcing window (mzr_run, mzr_run_async,
// mzr_run_src_dev, mzr_step) and the row-reordering scratch of the host getters / setters.  false = out of memory.
bool ensureRunoffW(mzr_handle h) {
  if (h->runoffW.p) return true;
  try { h->runoffW.allocStaging((size_t)h->cfg.maxWindow * h->H); } catch (const std::string &) { (void)hipGetLastError(); return false; }
  return true;
}
bool ensureScratch(mzr_handle h) {
  if (h->scratchOut.p) return true;
  try { h->scratchOut.alloc((size_t)h->cfg.maxWindow * h->N); } catch (const std::string &) { (void)hipGetLastError(); return false; }
  return true;
}
// checked copies of the state getters / setters: a failed copy is an error, not silently wrong state
#define MZR_COPY(dst, src, bytes, kind, who) do { if (hipMemcpy((dst), (src), (bytes), (kind)) != hipSuccess) return fail(h, 92, std::string(who) + "/hipMemcpy failed"); } while (0)

int idxOf(mzr_handle h, int method) {
  for (int i = 0; i < h->cfg.nRoutes; ++i) if (h->cfg.routeMethods[i] == method) return i;
  return -1;
}

void fillDev(mzr_handle h, MzrDev &d) {
  memset(&d, 0, sizeof d);
  d.stepBlock = 1;
  d.N = h->N; d.H = h->H; d.nStages = h->nStages;
  d.sigma = h->sigma.p; d.upStart = h->upStart.p; d.nUp = h->nUp.p; d.nGood = h->nGood.p;
  d.goodMask = h->goodMask.p; d.isOutlet = h->isOutlet.p;
  d.hruOff = h->hruOff.p; d.hruIdx = h->hruIdx.p; d.hruW = h->hruW.p;
  d.slope = h->par[0].p; d.mann = h->par[1].p; d.width = h->par[2].p; d.depth = h->par[3].p;
  d.length = h->par[4].p; d.storage = h->par[5].p; d.side = h->par[6].p; d.fldp = h->par[7].p;
  d.basarea = h->par[8].p; d.minflow = h->par[10].p;
  d.chanTab = h->chanDirty ? nullptr : h->chanTab.p;
  d.kwK = h->kwK.p; d.kwCW = h->kwCW.p;
  d.dt = h->cfg.dt; d.min_length_route = h->cfg.min_length_route; d.runoffMin = h->cfg.runoffMin;
  d.mcTailTol = h->cfg.mcTailTol >= 0.0 ? h->cfg.mcTailTol : 0.0;
  d.sweepPrio = h->cfg.sweepPriority != 0;
  // Round 6: the SMALL partner of two sweeps on one device (rank 0's mainstem beside its tributary sweep: sweepShare < 0.5) takes every
  // wavefront of its launch however late it starts.  Its few hundred workgroups are dispatched beside the other domain's kernels, and one
  // launch in ten or so most of them started 20+ us behind the first: only the 64 that always join took part and the window took 1.2-1.4 s
  // instead of 0.3 (bench.py --loopback --config c3: joined 64 of 480).  Together with the order of the two launches (PartitionedRouter:
  // the mainstem's window first, so that its sweep is resident when the large one's thousands of workgroups arrive): 0 slow windows in 32,
  // against 1 in 8 with either measure alone (profiles/r06_experiments.md 7).  Both grids together fit the device (the shares are of the
  // measured capacity): these wavefronts are late, not waiting for a slot; the watchdog and the retry stay behind the rule.
  d.sweepAlways = (h->cfg.sweepShare > 0.0 && h->cfg.sweepShare < 0.5) ? (1 << 30) : 64;
  d.stallTicks = (long long)((h->cfg.sweepTimeout > 0.0 ? h->cfg.sweepTimeout : 8.0) * 1.e8);      // wall_clock64: 100 MHz
  d.negRunoffTol = h->cfg.negRunoffTol; d.time_conv = h->cfg.time_conv; d.length_conv = h->cfg.length_conv;
  d.hw_drain_point = h->cfg.hw_drain_point; d.doesBasinRoute = h->cfg.doesBasinRoute;
  d.is_flux_wm = h->cfg.is_flux_wm; d.wm = (h->cfg.is_flux_wm && h->wmSteps > 0) ? h->wm.p : nullptr;
  d.ntdhBas = h->ntdhBas; d.fracFuture = h->fracFuture.p; d.fracPad = h->fracPad.p;
  d.qi = h->qi.p; d.qlat = h->qlat.p;
  d.basS0 = h->basS[h->basCur].p; d.basS1 = h->basS[h->basCur ^ 1].p;
  d.maxtdh = h->maxtdh; d.ntdh = h->ntdh.p; d.uh = h->uh.p; d.irfQ = h->irfQ.p;
  d.kwN = h->kwN.p; d.kwQT = h->kwQ.p; d.kwTR = h->kwTR.p;
  d.obN = h->obN.p; d.obQT = h->obQ.p;
  d.kwtRouted = h->kwtRouted.p; d.kwtRoutedB = h->kwtRoutedB.p; d.kwtRoutedC = h->kwtRoutedC.p; d.kwtGeneric = h->kwtGeneric.p; d.kwtLight = h->kwtLight.p;
  d.kwDone = h->kwDone.p; d.down = h->down.p; d.swItem = h->swItem.p; d.swLo = h->swLo.p; d.swHi = h->swHi.p;
  d.swRA = h->swRA.p; d.swP = h->swP.p; d.swHead = h->swHead.p; d.swBeat = h->swBeat.p;
  d.kwtHead = h->kwtHead.p; d.nHead = (int)h->h_kwtHead.size(); d.nDepLight = (int)h->h_kwtDepLight.size();
  d.nA = (int)h->h_swA.size(); d.nB = (int)h->h_swB.size(); d.nC = (int)h->h_swC.size(); d.nG = (int)h->h_kwtGeneric.size();
  d.kwtStat = h->countTraffic ? h->kwtStat.p : nullptr; d.err = h->err.p; d.dbgCycles = h->dbgCycles.p;
  d.lakeSlot = h->nLake ? h->lakeSlot.p : nullptr; d.lakeModel = h->lakeModel.p; d.lakePar = h->lakePar.p;
  d.lakeEvap = h->lakeEvap.p; d.lakePrecip = h->lakePrecip.p; d.calMonth = h->calMonth.p; d.calDay = h->calDay.p; d.calDoy = h->calDoy.p;
  d.nLake = h->nLake; d.LakeInputOption = h->LakeInputOption; d.calendarId = h->calendarId; d.lakeL = h->lakeL; d.lakeLD = h->lakeLD;
  d.lakeTarg = h->anyLakeTarget ? h->lakeTarg.p : nullptr; d.lakeWmVol = h->lakeWmVol.p; d.volJumpstart = h->volJumpstart;
  d.iTime0 = h->totalSteps;
  d.haloSlot = h->nHalo ? h->haloSlot.p : nullptr; d.exportSlot = h->nExp ? h->exportSlot.p : nullptr;
  d.nHalo = h->nHalo; d.nExp = h->nExp; d.Wmax = h->cfg.maxWindow;
  d.imN = h->imN.p; d.imOQ = h->imOQ.p; d.imOT = h->imOT.p;
  d.solSrc = h->solSrc.p; d.solInst = h->solInst.p; d.basSol = h->basSol.p; d.solS0 = h->solS[h->solCur].p; d.solS1 = h->solS[h->solCur ^ 1].p;
  d.solFlux = nullptr; d.solMass = nullptr; d.trVol0 = nullptr; d.time_conv_solute = h->time_conv_solute; d.mass_conv_solute = h->mass_conv_solute;
  d.qmod = h->qmod; d.qBlendPeriod = h->qBlendPeriod; d.QerrTrend = h->QerrTrend; d.nGauge = h->nGauge;
  d.gaugeFirst = h->gaugeFirst.p; d.gaugeNext = h->gaugeNext.p; d.obsHave = h->obsHave.p; d.obsVal = h->obsVal.p;
  d.qobs = nullptr; d.qerr = nullptr; d.qelapsed = nullptr;
  d.rtItemR = h->rtItemR.p; d.rtItemInfo = nullptr; d.rtRA = h->rtRA.p; d.rtP = h->rtP.p; d.rtDone = nullptr; d.rtHead = nullptr;
  d.exN = h->exN.p; d.exOQ = h->exOQ.p; d.exOT = h->exOT.p;
}

void setRoute(mzr_handle h, MzrDev &d, int ix) {
  RouteBufs &rb = h->route[ix];
  d.Q = rb.Q.p; d.vol = rb.vol.p; d.vol0 = rb.vol0.p; d.inflow = rb.inflow.p; d.ele = rb.ele.p;
  d.hInflow = rb.hInflow.p; d.hEle = rb.hEle.p; d.hFlood = rb.hFlood.p;
  d.floodvol = rb.floodvol.p; d.wb = rb.wb.p; d.qsum = rb.qsum.p; d.mol = rb.mol.p; d.imQ = rb.imQ.p; d.wmact = rb.wmact.p;
  d.lakeMut = rb.lakeMut.p; d.lakeRing = rb.lakeRing.p; d.lakeHead = rb.lakeHead.p; d.lakeRingD = rb.lakeRingD.p; d.lakeHeadD = rb.lakeHeadD.p;
  d.rtDone = rb.rtDone.p; d.rtHead = rb.rtHead.p;
  d.mcSub = rb.mcSub.p; d.lanePerm = rb.havePerm ? rb.lanePerm.p : nullptr; d.permN = rb.permN; d.nHeavyPos = rb.havePerm ? rb.nHeavyPos : 0;
  d.qobs = rb.qobs.p; d.qerr = rb.qerr.p; d.qelapsed = rb.qelapsed.p;
  d.solFlux = rb.solFlux.p; d.solMass = rb.solMass.p; d.trVol0 = h->tracer ? rb.trVol0.p : nullptr;
}


// A wavefront of the persistent KWT sweep gave up waiting (code 93).  What it recorded, set against the progress words
// and ticket heads as they are in memory now and against the host's copy of the schedule, tells the cases apart:
// the step it waited for IS in memory (a result that did not become visible to the poller), the ticket that would
// produce it was drawn and never finished, or it was never drawn (a queue nobody serves).
std::string stallReportKwt(mzr_handle h, const MzrErr &e) {
  char b[512];
  std::string out;
  const int N = h->N;
  std::vector<unsigned long long> done(N, 0); std::vector<int> heads(8 * 16 + 16, 0);
  if (h->kwDone.p) (void)hipMemcpy(done.data(), h->kwDone.p, (size_t)N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (h->swHead.p) (void)hipMemcpy(heads.data(), h->swHead.p, heads.size() * sizeof(int), hipMemcpyDeviceToHost);
  const int W = h->swTablesW;
  auto steps = [&](int r) { return (r >= 0 && r < N) ? (int)(done[r] & 0xffffull) : -1; };
  auto sig = [&](int r) { return (r >= 0 && r < N && !h->h_sigma.empty()) ? h->h_sigma[r] : -1; };
  // reach -> item of the sweep
  std::vector<int> itemOf(N, -1);
  {
    const std::vector<MzrKwtRec> *lists[5] = {&h->h_swA, &h->h_swB, &h->h_kwtGeneric, nullptr, &h->h_swC};
    const int per[5] = {4, 8, 1, 64, 16};
    for (size_t i = 0; i < h->h_swCode.size(); ++i) {
      const int cls = h->h_swCode[i] >> 28, bi = h->h_swCode[i] & 0x0fffffff;
      if (cls == 3) { for (size_t k = (size_t)bi * 64; k < std::min(h->h_kwtDepLight.size(), (size_t)(bi + 1) * 64); ++k) itemOf[h->h_kwtDepLight[k]] = (int)i; continue; }
      if (cls < 0 || cls > 4 || !lists[cls]) continue;
      const std::vector<MzrKwtRec> &v = *lists[cls];
      for (size_t k = (size_t)bi * per[cls]; k < std::min(v.size(), (size_t)(bi + 1) * per[cls]); ++k) if (v[k].r >= 0 && v[k].r < N && v[k].sigma < MZR_KWT_HOLE) itemOf[v[k].r] = (int)i;
    }
  }
  const int nL = (int)h->h_swRA.size();
  auto ticketOf = [&](int item, int s, int &q) {      // ticket number of (item, launch s) in its queue, -1 if outside the tables
    q = item & 7;
    if (item < 0 || s < 0 || s >= nL) return -1;
    const int a = h->h_swRA[s];
    const int first = a + (((q - a) % 8 + 8) % 8);
    if (item < first) return -1;
    return h->h_swP[(size_t)s * 8 + q] + (item - first) / 8;
  };
  auto launchOfHead = [&](int q) {      // launch the next ticket of queue q belongs to
    const int k = heads[q * 16];
    int lo = 0, hi = nL;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (h->h_swP[(size_t)mid * 8 + q] <= k) lo = mid; else hi = mid - 1; }
    return lo;
  };
  snprintf(b, sizeof b, "MZR STALL kwt sweep: window of %d steps, %d stages, %d items, %d wavefronts; waiting reach %d (stage %d) in launch %d = its step %d, "
           "lane %d of %d unsatisfied lanes, XCC %d, %.2f s without progress; ", W, h->nStages, h->swItems, h->swWaves, e.reach, sig(e.reach), e.s,
           e.s - sig(e.reach), e.lane, e.nBad, e.xcc, (double)e.waited * 1.e-8);
  out += b;
  const int u = e.depReach;
  const char *rel = u == e.reach ? "itself" : (e.reach >= 0 && e.reach < N && h->h_down[e.reach] == u) ? "its downstream reach" : "an upstream reach";
  snprintf(b, sizeof b, "polled %s %d (stage %d): saw word 0x%08x = %d steps, needs %d; in memory now 0x%08x = %d steps -> %s; ", rel, u, sig(u), (unsigned)e.seen,
           e.seen & 0xffff, e.need, (u >= 0 && u < N) ? (unsigned)done[u] : 0u, steps(u),
           steps(u) >= e.need ? "PRODUCED BUT NOT SEEN by the poller" : "not produced");
  out += b;
  if (u >= 0 && u < N && steps(u) < e.need) {
    const int su = steps(u) + sig(u);      // launch of the step the reach has not finished
    int q = 0;
    const int it = itemOf[u], tk = it >= 0 ? ticketOf(it, su, q) : -1;
    snprintf(b, sizeof b, "its next step %d is launch %d, item %d (class %d), queue %d ticket %d, head of that queue now %d -> %s; ", steps(u), su, it,
             it >= 0 ? h->h_swCode[it] >> 28 : -1, q, tk, heads[q * 16], tk < 0 ? "?" : heads[q * 16] > tk ? "ticket DRAWN, step not published" : "ticket NOT DRAWN");
    out += b;
    // what that step itself waits for
    if (u < (int)h->h_down.size()) {
      std::vector<int> up(1, 0);
      int u0 = 0, nu = 0;
      (void)hipMemcpy(&u0, h->upStart.p + u, sizeof(int), hipMemcpyDeviceToHost);
      nu = h->h_nUp[u];
      std::string deps;
      for (int k = 0; k < nu; ++k) { snprintf(b, sizeof b, " up %d:%d", u0 + k, steps(u0 + k)); deps += b; }
      snprintf(b, sizeof b, " down %d:%d", h->h_down[u], steps(h->h_down[u])); deps += b;
      out += "its own dependencies (reach:steps)" + deps + "; ";
    }
  }
  // frontier: the lowest launch that still has an unfinished routed reach
  long long lowest = 1LL << 60; int nLow = 0, rLow = -1;
  for (int r = 0; r < N; ++r) {
    if (itemOf[r] < 0) continue;
    const int st = steps(r);
    if (st >= W) continue;
    const long long l = (long long)st + sig(r);
    if (l < lowest) { lowest = l; nLow = 1; rLow = r; } else if (l == lowest) ++nLow;
  }
  snprintf(b, sizeof b, "frontier: launch %lld has %d unfinished reaches (e.g. %d); heads (queue:ticket@launch) now", lowest, nLow, rLow);
  out += b;
  for (int q = 0; q < 8; ++q) { snprintf(b, sizeof b, " %d:%d@%d", q, heads[q * 16], nL > 0 ? launchOfHead(q) : -1); out += b; }
  snprintf(b, sizeof b, "; wavefronts of the launch that arrived %d, joined %d", heads[8 * 16 + 2], heads[8 * 16 + 3]);
  out += b;
  out += "; heads when the wavefront gave up";
  for (int q = 0; q < 8; ++q) { snprintf(b, sizeof b, " %d", e.heads[q]); out += b; }
  if (rLow >= 0) {
    int q = 0;
    const int it = itemOf[rLow], tk = ticketOf(it, (int)lowest, q);
    snprintf(b, sizeof b, "; frontier reach %d: item %d queue %d ticket %d (%s)", rLow, it, q, tk, tk < 0 ? "?" : heads[q * 16] > tk ? "drawn" : "not drawn");
    out += b;
  }
  if (h->swBeat.p) {      // what every wavefront was doing: launch, item, queue, phase (1 drew a ticket, 2 waiting, 3 computing, 4 published, 9 left), items done, XCC, ticket
    const int nw = std::max(h->swWaves, 0);
    std::vector<int> bt8((size_t)nw * MZR_BEAT, 0), bt((size_t)nw * 8, 0);
    if (nw) (void)hipMemcpy(bt8.data(), h->swBeat.p, bt8.size() * sizeof(int), hipMemcpyDeviceToHost);
    for (int w = 0; w < nw; ++w) for (int k = 0; k < 8; ++k) bt[(size_t)w * 8 + k] = bt8[(size_t)w * MZR_BEAT + k];
    int phase[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, perX[8] = {0, 0, 0, 0, 0, 0, 0, 0}, servQ[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long minS[8], maxS[8];
    for (int q = 0; q < 8; ++q) { minS[q] = 1LL << 60; maxS[q] = -1; }
    for (int w = 0; w < nw; ++w) {
      const int *x = &bt[(size_t)w * 8];
      ++phase[std::min(std::max(x[3], 0), 9)];
      if (x[3] == 0) continue;
      ++perX[x[5] & 7];
      if (x[3] != 9 && x[4] > 0) { ++servQ[x[2] & 7]; minS[x[2] & 7] = std::min<long long>(minS[x[2] & 7], x[0]); maxS[x[2] & 7] = std::max<long long>(maxS[x[2] & 7], x[0]); }
    }
    snprintf(b, sizeof b, "; wavefronts by phase: never started %d, drew %d, waiting %d, computing %d, published %d, left %d; per XCC", phase[0], phase[1], phase[2], phase[3], phase[4], phase[9]);
    out += b;
    for (int q = 0; q < 8; ++q) { snprintf(b, sizeof b, " %d", perX[q]); out += b; }
    out += "; serving queue (count, launches held)";
    for (int q = 0; q < 8; ++q) { snprintf(b, sizeof b, " %d:%d[%lld..%lld]", q, servQ[q], servQ[q] ? minS[q] : -1, maxS[q]); out += b; }
    // who holds the frontier's unfinished items
    int shown = 0;
    for (int r = 0; r < N && shown < 6; ++r) {
      if (itemOf[r] < 0 || steps(r) >= W || (long long)steps(r) + sig(r) != lowest) continue;
      int holder = -1;
      for (int w = 0; w < nw; ++w) if (bt[(size_t)w * 8 + 3] != 0 && bt[(size_t)w * 8] == (int)lowest && bt[(size_t)w * 8 + 1] == itemOf[r]) { holder = w; break; }
      if (holder >= 0) snprintf(b, sizeof b, "; frontier reach %d item %d held by wavefront %d (XCC %d, phase %d, ticket %d, items done %d)", r, itemOf[r], holder, bt[(size_t)holder * 8 + 5],
                                bt[(size_t)holder * 8 + 3], bt[(size_t)holder * 8 + 6], bt[(size_t)holder * 8 + 4]);
      else snprintf(b, sizeof b, "; frontier reach %d item %d held by NO wavefront", r, itemOf[r]);
      out += b; ++shown;
    }
  }
  return out;
}

int checkDeviceError(mzr_handle h) {
  MzrErr e;
  if (hipMemcpy(&e, h->err.p, sizeof e, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, 99, "mzr/hipMemcpy(err) failed");
  if (e.code == 0) return 0;
  (void)hipMemset(h->err.p, 0, sizeof(MzrErr));
  const int ext = (e.reach >= 0 && e.reach < h->N) ? h->int2ext[e.reach] : -1;
  const int id = (ext >= 0 && !h->reachId.empty()) ? h->reachId[ext] : ext + 1;
  const char *what = "routing error";
  switch (e.where) {
    case 1: what = "basin2reach/exceeded negative runoff tolerance"; break;
    case 10: what = "kwt_rch/qexmul_rch/work array bounds exceeded"; break;
    case 11: what = e.code == 20 ? "kwt_rch/getusq_rch/qexmul_rch/stuck in the continuous do-loop"
                  : e.code == 30 ? "kwt_rch/getusq_rch/qexmul_rch/expect process in order of time"
                  : e.code == 40 ? "kwt_rch/getusq_rch/qexmul_rch/the times are not ordered as we assume"
                                 : "kwt_rch/getusq_rch/qexmul_rch/QD_TEMP bounds exceeded"; break;
    case 12: what = "kwt_rch/negative flow extracted from upstream reach"; break;
    case 13: what = e.code == 20 ? "kwt_rch/kinwav_rch/zero flow" : e.code == 30 ? "kwt_rch/kinwav_rch/TEXIT equals TEXIT2 in kinwav"
                                                                                  : "kwt_rch/kinwav_rch/RUPDATE/array bounds exceeded"; break;
    case 14: what = "kwt_rch/no waiting particle left in reach"; break;
    case 15: what = "kwt_rch/interp_rch/bad bounds"; break;
    case 17: what = "kwt_rch/extract_from_rch/interp_rch/bad bounds"; break;
    case 18: what = "kwt_rch/getusq_rch/lake outlet reach should have one upstream lake"; break;
    case 20: what = "persistent KWT sweep gave up waiting for a reach it depends on"; break;
    case 21: what = "persistent sweep of an Eulerian method gave up waiting for a reach it depends on"; break;
    case 30: what = "mzr_import_boundary_dev/the record is not what this domain expects (routing methods, window length, reach count, or the constituent on one side only; 'window step' = the steps the record holds)"; break;
  }
  char buf[768];
  snprintf(buf, sizeof buf, "main_routing/route_network/%s [where %d, reach index %d id %d, window step %d]", what, e.where, ext + 1, id, e.step);
  std::string msg = buf;
  if (e.code == 93) {
    snprintf(buf, sizeof buf, " [KWT windows run %lld, steps since the last regrouping %lld]", h->kwtWindows, h->kwtStepsSince);
    msg += buf;
    if (e.where >= 30 && e.where < 50) {
      static const char *tab[] = {"swP[s][q]", "swP[s+1][q]", "swRA[s]", "swLo[item]", "swHi[item]", "swItem[item]", "", "", "", "",
                                  "record.r", "record.sigma", "record.u0", "record.flags"};
      snprintf(buf, sizeof buf, " MZR STALE %s: launch %d, index %d, cached value %d, memory holds %d; queue %d, ticket/lane %d, XCC %d", tab[e.where - 30], e.s, e.depReach,
               e.seen, e.need, e.queue, e.lane, e.xcc);
      msg += buf;
    } else
    if (e.where == 20) {
      msg += " " + stallReportKwt(h, e);
      snprintf(buf, sizeof buf, "; slow passes (wait end -> publish > 10 ms): %d", e.nSlow);
      msg += buf;
      for (int k = 0; k < std::min(e.nSlow, 32); ++k) {
        const int *o = e.slow[k];
        snprintf(buf, sizeof buf, " [wavefront %d launch %d item %d G %d: %.3f s, published %.3f s %s the raise, HW_ID 0x%04x (wave %d simd %d cu %d sh %d se %d) XCC %d]", o[0], o[1], o[2], o[7], o[3] * 1.e-8,
                 std::abs(o[6] - e.raisedAt) * 1.e-8, (o[6] - e.raisedAt) >= 0 ? "after" : "before", o[4] & 0xffff, o[4] & 15, (o[4] >> 4) & 3, (o[4] >> 8) & 15, (o[4] >> 12) & 1, (o[4] >> 13) & 7, o[5]);
        msg += buf;
        msg += " sections(ms since wait end)";
        for (int j = 0; j < 24; ++j) if (e.slowT[k][j] != 0) { snprintf(buf, sizeof buf, " %d:%.3f", j, e.slowT[k][j] * 1.e-5); msg += buf; }
      }
    }
    else {
      snprintf(buf, sizeof buf, " MZR STALL route sweep: waiting reach %d in launch %d, lane %d of %d unsatisfied, queue %d, XCC %d, %.2f s without progress; polled reach %d: saw %d, needs %d; heads",
               e.reach, e.s, e.lane, e.nBad, e.queue, e.xcc, (double)e.waited * 1.e-8, e.depReach, e.seen, e.need);
      msg += buf;
      for (int q = 0; q < 8; ++q) { snprintf(buf, sizeof buf, " %d", e.heads[q]); msg += buf; }
    }
    fprintf(stderr, "%s\n", msg.c_str());
  }
  return fail(h, e.code, msg);
}

// ---- persistent KWT sweep: host side ------------------------------------------------------------
// Items of the sweep = consecutive blocks of the stage-ordered class lists (4 class-A reaches, 8 class-B reaches,
// single confluences of more than two reaches, 64 lake / halo reaches), merged into one list ordered by stage.
// The items of launch s are a contiguous range of that list, and the tickets of the kernel number them launch
// after launch in eight queues (item i belongs to queue i % 8): kwt_sweep_tables makes the per-launch ranges
// and ticket prefix sums for a window length.
// Grid of a persistent sweep from the wavefronts the device holds of its kernel (measured, mzr_sweep_*_capacity): a
// margin below it -- other kernels of the window (hillslope chunks, history sums) come and go beside the sweep, and a
// sweep with workgroups left waiting for a slot can stall -- times the handle's share of the device.
#define MZR_CLOCK_LOG 1024      // sweep launches whose device-clock pair is kept (a ring)
int sweepGrid(mzr_handle h, int held) {
  if (held < 1) return 0;      // the capacity could not be measured: the caller decides (the Eulerian methods then keep one launch per stage)
  const double share = (h->cfg.sweepShare > 0.0 && h->cfg.sweepShare <= 1.0) ? h->cfg.sweepShare : 1.0;
  const int g = (int)((held - std::max(64, held / 50)) * share);
  return std::max(8, g & ~7);
}

// Wavefronts the sweep is launched with at most: what the device holds of the kernel (measured when first used,
// mzr_sweep_kwt_capacity), less a margin, times the handle's share
void kwt_measure_cap(mzr_handle h) {
  if (h->swCap < 1) {
    const bool full = h->nLake || h->nHalo || h->nExp || h->cfg.is_flux_wm;
    if (!h->swHead.p) { h->swHead.alloc(8 * 16 + 16 + 32); h->swHead.zero(); }
    int cap = 0;
    { MzrDev dc; memset(&dc, 0, sizeof dc); dc.swHead = h->swHead.p; dc.err = h->err.p; cap = sweepGrid(h, mzr_sweep_kwt_capacity(full, dc, h->stream)); }
    if (cap < 1) {      // census failed: a conservative grid (four wavefronts per CU always fit) and a word about it
      int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device);
      cap = std::max(8, cus * 4);
      fprintf(stderr, "mzr: the wavefront capacity of the KWT sweep could not be measured on device %d; sweeping with %d wavefronts\n", h->cfg.device, cap);
    }
    if (const char *e = getenv("MZR_KWT_SWEEP_WAVES")) { const int v = atoi(e); if (v > 0) cap = v; }      // experiments only: any grid, also one the device does not hold
    h->swCap = cap;
  }
}

void kwt_build_sweep(mzr_handle h) {
  if (!h->swClock.p) { try { h->swClock.alloc(2 * MZR_CLOCK_LOG); h->swClock.zero(); h->swClockN = 0; } catch (const std::string &) { (void)hipGetLastError(); } }
  if (!h->swHead.p) { h->swHead.alloc(8 * 16 + 16 + 32); h->swHead.zero(); }      // eight ticket heads (one cache line each), census and arrival counters, histogram of the start delays
  kwt_measure_cap(h);
  struct It { int code, lo, hi; };
  std::vector<It> items;
  auto addRouted = [&](const std::vector<MzrKwtRec> &v, int cls, size_t per) {
    for (size_t b = 0; b * per < v.size(); ++b) {
      int lo = 1 << 30, hi = -1;
      for (size_t k = b * per; k < std::min(v.size(), (b + 1) * per); ++k) { if (v[k].sigma >= MZR_KWT_HOLE) continue; lo = std::min(lo, v[k].sigma); hi = std::max(hi, v[k].sigma); }
      items.push_back(It{(cls << 28) | (int)b, lo, hi});
    }
  };
  addRouted(h->h_swA, 0, 4);
  addRouted(h->h_swB, 1, 8);
  addRouted(h->h_swC, 4, 16);
  addRouted(h->h_kwtGeneric, 2, 1);
  if (!h->h_kwtDepLight.empty()) {
    std::vector<int> sg(h->N);
    (void)hipMemcpy(sg.data(), h->sigma.p, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost);
    for (size_t b = 0; b * 64 < h->h_kwtDepLight.size(); ++b)
      items.push_back(It{(3 << 28) | (int)b, sg[h->h_kwtDepLight[b * 64]], sg[h->h_kwtDepLight[std::min(h->h_kwtDepLight.size(), (b + 1) * 64) - 1]]});
  }
  if (!h->swHeavyFirst) std::stable_sort(items.begin(), items.end(), [](const It &a, const It &b) { return a.lo < b.lo; });
  h->h_swLo.clear(); h->h_swHiMax.clear();
  std::vector<int> code, hi;
  int run = -1;
  for (const It &it : items) { code.push_back(it.code); h->h_swLo.push_back(it.lo); hi.push_back(it.hi); run = std::max(run, it.hi); h->h_swHiMax.push_back(run); }
  const std::vector<int> lo = h->h_swLo;
  if (code.empty()) { code.push_back(0); hi.push_back(-1); }
  (void)hipStreamSynchronize(h->stream);
  h->swItem.upload(code); h->swLo.upload(lo.empty() ? std::vector<int>(1, 1 << 30) : lo); h->swHi.upload(hi);
  if (!h->swBeat.p && getenv("MZR_SWEEP_DEBUG") && atoi(getenv("MZR_SWEEP_DEBUG")) != 0) { h->swBeat.alloc((size_t)std::max(h->swCap, 8192) * MZR_BEAT); h->swBeat.zero(); }
  h->swItems = (int)items.size();
  h->h_swCode = code;
  h->swTablesW = -1;          // ticket tables have to be made again
}

// Items of the Eulerian sweeps (k_sweep_route): the reaches of every stage in blocks of 64, lakes in blocks of their own
// (their plain state is handed on with fences); ordered by stage, so the items of a launch are a contiguous range.
void rt_build_items(mzr_handle h) {
  std::vector<int> rec;       // [nItems][64][4]: reach (-1 = none), first upstream reach, number of upstream reaches | lake << 8, stage
  std::vector<int> upStart(h->N);
  (void)hipMemcpy(upStart.data(), h->upStart.p, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost);
  h->h_rtStage.clear();
  for (int sg = 0; sg < h->nStages; ++sg) {
    std::vector<int> plain, lakes;
    for (int i = h->stageStart[sg]; i < h->stageStart[sg + 1]; ++i)
      ((!h->h_lakeSlot.empty() && h->h_lakeSlot[i] >= 0) ? lakes : plain).push_back(i);
    for (int kind = 0; kind < 2; ++kind) {
      const std::vector<int> &v = kind ? lakes : plain;
      for (size_t b = 0; b * 64 < v.size(); ++b) {
        for (size_t l = 0; l < 64; ++l) {
          const int r = b * 64 + l < v.size() ? v[b * 64 + l] : -1;
          rec.push_back(r); rec.push_back(r >= 0 ? upStart[r] : 0); rec.push_back(r >= 0 ? ((int)h->h_nUp[r] | (kind << 8)) : 0); rec.push_back(sg);
        }
        h->h_rtStage.push_back(sg);
      }
    }
  }
  h->rtItems = (int)h->h_rtStage.size();
  if (rec.empty()) rec.assign(256, -1);
  (void)hipStreamSynchronize(h->stream);
  h->rtItemR.upload(rec);
  h->rtTablesW = -1;
}

// per launch of a W-step window: first active item and ticket prefix sums of the eight queues (as kwt_sweep_tables)
void rt_sweep_tables(mzr_handle h, int W) {
  if (h->rtTablesW == W) return;
  const int nS = h->nStages, nL = nS + W - 1;
  std::vector<int> ra(nL, 0), P((size_t)(nL + 1) * 8, 0);
  int maxAct = 0;
  for (int s = 0; s < nL; ++s) {
    const int b = (int)(std::upper_bound(h->h_rtStage.begin(), h->h_rtStage.end(), s) - h->h_rtStage.begin());          // items with stage <= s
    const int a = std::min(b, (int)(std::lower_bound(h->h_rtStage.begin(), h->h_rtStage.end(), s - W + 1) - h->h_rtStage.begin()));
    ra[s] = a;
    maxAct = std::max(maxAct, b - a);
    for (int q = 0; q < 8; ++q) {
      const int first = a + (((q - a) % 8 + 8) % 8);
      P[(size_t)(s + 1) * 8 + q] = P[(size_t)s * 8 + q] + (first < b ? (b - first + 7) / 8 : 0);
    }
  }
  (void)hipStreamSynchronize(h->stream);
  for (int ix = 0; ix < h->cfg.nRoutes; ++ix) if (h->routeStream[ix]) (void)hipStreamSynchronize(h->routeStream[ix]);
  h->rtRA.upload(ra); h->rtP.upload(P);
  h->rtMaxAct = maxAct;
  h->rtTablesW = W;
}

// launch ranges and ticket prefix sums of a window of W steps: launch s takes the reaches of stage j through step s - j
void kwt_sweep_tables(mzr_handle h, int W) {
  if (h->swTablesW == W) return;
  (void)hipStreamSynchronize(h->stream);     // (a sweep still in flight reads the old tables; the census of the kernel's first use runs on this stream)
  kwt_measure_cap(h);
  const int nS = h->nStages, nL = nS + W - 1, nI = h->swItems;
  std::vector<int> ra(nL, 0), P((size_t)(nL + 1) * 8, 0);
  int maxAct = 0;
  for (int s = 0; s < nL; ++s) {
    int b = (int)(std::upper_bound(h->h_swLo.begin(), h->h_swLo.end(), s) - h->h_swLo.begin());                 // items with lo <= s
    int a = std::min(b, (int)(std::lower_bound(h->h_swHiMax.begin(), h->h_swHiMax.end(), s - W + 1) - h->h_swHiMax.begin()));   // first with hi >= s-W+1
    if (h->swHeavyFirst) { a = 0; b = nI; }      // items in order of weight: every launch draws them all, an item without a step in the launch is dropped at once
    ra[s] = a;
    maxAct = std::max(maxAct, b - a);
    for (int q = 0; q < 8; ++q) {
      const int first = a + (((q - a) % 8 + 8) % 8);
      P[(size_t)(s + 1) * 8 + q] = P[(size_t)s * 8 + q] + (first < b ? (b - first + 7) / 8 : 0);
    }
  }
  (void)hipStreamSynchronize(h->stream);     // a sweep still in flight reads the old tables
  h->swRA.upload(ra); h->swP.upload(P);
  h->h_swRA = ra; h->h_swP = P;
  h->swWaves = h->swItems > 0 ? std::max(8, std::min(h->swCap, maxAct)) : 0;
  h->swTablesW = W;
}

}  // namespace

static int flushSteps(mzr_handle h, bool keepStaged = false);
static void flushTail(mzr_handle h);
static void issueTail(mzr_handle h, int jBegin, int jEnd);
// the row of one kind for the coming step (a later call for the same step replaces it, as a second call of the setter would)
static int stageRow(mzr_handle h, int kind, const double *a, size_t na, const double *b, size_t nb, const int *i0, const int *i1, const int *i2) {
  mzr_domain::StepRows &r = h->sr[kind];
  r.next[0].assign(a ? a : nullptr, a ? a + na : nullptr);
  r.next[1].assign(b ? b : nullptr, b ? b + nb : nullptr);
  const int *ip[3] = {i0, i1, i2};
  for (int k = 0; k < 3; ++k) { r.nextI[k].clear(); if (ip[k]) r.nextI[k].push_back(*ip[k]); }
  r.staged = true; h->srAny = true;
  return 0;
}
static void build_lane_perm(mzr_handle h, int ix, const std::vector<int> &key, int heavyMin = 0);
// steps handed over with mzr_step that have not been routed yet (mzr_config.stepBatch > 1) go first ...
#ifndef MZR_STEP_BLOCK_DEFAULT
#define MZR_STEP_BLOCK_DEFAULT 8
#endif
#define MZR_FLUSH_STEPS(h) do { if ((h) && ((h)->stepN > 0 || (h)->srAny) && !(h)->srApplying) { const int _rc = flushSteps(h); if (_rc) return _rc; } } while (0)
// a one-step call of a per-window setter on a handle that batches its steps: the row is put aside for the coming mzr_step
#define MZR_STAGES(h, nSteps) ((h) && (h)->cfg.stepBatch > 1 && (nSteps) == 1 && !(h)->srApplying && (h)->haveState)
// ... and so do the launches of the last window that were kept back for the next one (overlapping windows): every entry point
// that reads or changes anything a window touches takes this one; the run calls themselves take MZR_FLUSH_STEPS
#define MZR_FLUSH(h) do { MZR_FLUSH_STEPS(h); if ((h) && (h)->tail.pending) flushTail(h); } while (0)

extern "C" {

void mzr_default_config(mzr_config *c) {
  memset(c, 0, sizeof *c);
  c->dt = 3600.0; c->nRoutes = 1; c->routeMethods[0] = MZR_KWT;
  c->doesBasinRoute = 1; c->hw_drain_point = 2; c->min_length_route = 0.0; c->runoffMin = 0.0;
  c->negRunoffTol = -1.e-3; c->time_conv = 1.0; c->length_conv = 1.0; c->maxWindow = 64; c->device = 0;
  // the two knobs below can be preset from the environment (tests, tools); a host sets the fields
  c->mcTailTol = 1.e-7;
  if (const char *e = getenv("MZR_MC_TAIL_TOL")) c->mcTailTol = atof(e);
  c->sweepShare = 1.0;
  c->stepBatch = 1;
  c->sweepTimeout = 0.0;      // automatic (include/mzr.h)
  if (const char *e = getenv("MZR_SWEEP_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) c->sweepTimeout = v; }
}

int mzr_create(const mzr_config *cfg, mzr_handle *out) {
  if (!cfg || !out) return 1;
  mzr_domain *h = new mzr_domain();
  h->cfg = *cfg;
  *out = h;
  if (cfg->nRoutes < 1 || cfg->nRoutes > 6) return fail(h, 81, "mzr_create/nRoutes must be 1..6");
  for (int i = 0; i < cfg->nRoutes; ++i) {
    const int m = cfg->routeMethods[i];
    if (m < 0 || m > 5) return fail(h, 81, "route_network/routing method id expect digits 0-5. Check <route_opt> in control file");
    h->route[i].method = m;
  }
  if (cfg->maxWindow < 1) return fail(h, 1, "mzr_create/maxWindow must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(h, 90, "mzr_create/no HIP device available: this library has no CPU path");
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(h, 90, "mzr_create/hipSetDevice failed");
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(h, 90, "mzr_create/hipStreamCreate failed");
  return 0;
}

int mzr_destroy(mzr_handle h) {
  if (!h) return 0;
  (void)hipSetDevice(h->cfg.device);
  // Steps put aside by mzr_step (stepBatch > 1) that nobody asked the result of are NOT routed here -- there is nobody left
  // to read them -- but it is said; launches kept back for a next window are dropped likewise.  Everything queued has to
  // finish before the buffers it reads (page-locked step rows, forcing windows) are freed.
  if (h->stepN > 0) fprintf(stderr, "mzr_destroy: %d step(s) handed over with mzr_step (stepBatch %d) were never routed: no result was asked for after them\n", h->stepN, h->cfg.stepBatch);
  h->tail.pending = false;
  if (h->copyStream) (void)hipStreamSynchronize(h->copyStream);
  if (h->timerStream) (void)hipStreamSynchronize(h->timerStream);
  for (int ix = 0; ix < 6; ++ix) if (h->routeStream[ix]) (void)hipStreamSynchronize(h->routeStream[ix]);
  if (h->basinStream) (void)hipStreamSynchronize(h->basinStream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto &rb : h->route) for (auto &e : rb.events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->basinStream) (void)hipStreamDestroy(h->basinStream);
  if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
  if (h->exportDone) (void)hipEventDestroy(h->exportDone);
  if (h->exportPrevDone) (void)hipEventDestroy(h->exportPrevDone);
  if (h->importDone) (void)hipEventDestroy(h->importDone);
  if (h->expStream) (void)hipStreamDestroy(h->expStream);
  if (h->sweepGo) (void)hipEventDestroy(h->sweepGo);
  for (auto &e : h->prevRowsEv) if (e) (void)hipEventDestroy(e);
  if (h->rwOther) (void)hipEventDestroy(h->rwOther);
  for (int i = 0; i < 2; ++i) { if (h->stepHost[i]) (void)hipHostFree(h->stepHost[i]); if (h->stepCopied[i]) (void)hipEventDestroy(h->stepCopied[i]); }
  for (int i = 0; i < 2; ++i) { if (h->rwCopied[i]) (void)hipEventDestroy(h->rwCopied[i]); if (h->rwRead[i]) (void)hipEventDestroy(h->rwRead[i]); }
  for (int ix = 0; ix < 6; ++ix) { if (h->routeStream[ix]) (void)hipStreamDestroy(h->routeStream[ix]); if (h->routeEvent[ix]) (void)hipEventDestroy(h->routeEvent[ix]); }
  for (auto &e : h->basinEvents) (void)hipEventDestroy(e);
  delete h;
  return 0;
}

int mzr_last_error(mzr_handle h, char *buf, int len) {
  if (!h || !buf || len <= 0) return 1;
  snprintf(buf, len, "%s", h->msg.c_str());
  return 0;
}

int mzr_set_network(mzr_handle h, int N, int H, const int *downIndex, const int *upOffset, const int *upIndex,
                    const int *upGood, const int *hruOffset, const int *hruIndex, const double *hruWeight,
                    const int *reachId) {
  MZR_FLUSH(h);
  if (!h) return 1;
  (void)hipSetDevice(h->cfg.device);
  try {
    h->N = N; h->H = H;
    h->chanTab.free(); h->chanDirty = true;
    // ---- breadth-first levels from the outlets; upstreams appended in UREACHI order
    std::vector<int> level; level.reserve(N);
    std::vector<int> levelStart{0};
    for (int i = 0; i < N; ++i) if (downIndex[i] <= 0) level.push_back(i);
    size_t lo = 0;
    while (lo < level.size()) {
      const size_t hi = level.size();
      levelStart.push_back((int)hi);
      for (size_t k = lo; k < hi; ++k) {
        const int r = level[k];
        for (int e = upOffset[r]; e < upOffset[r + 1]; ++e) {
          const int u = upIndex[e] - 1;
          if (u < 0 || u >= N || downIndex[u] - 1 != r) return fail(h, 20, "mzr_set_network/upstream list inconsistent with downIndex");
          level.push_back(u);
        }
      }
      lo = hi;
    }
    if ((int)level.size() != N) return fail(h, 20, "mzr_set_network/network has a cycle or a reach not connected to an outlet");
    const int nLev = (int)levelStart.size() - 1;
    h->nStages = nLev;
    // internal order: stage 0 = deepest level
    h->int2ext.assign(N, 0); h->ext2int.assign(N, 0); h->stageStart.assign(nLev + 1, 0);
    std::vector<int> sigma(N);
    int pos = 0; h->maxStageWidth = 0;
    for (int s = 0; s < nLev; ++s) {
      const int lev = nLev - 1 - s;
      h->stageStart[s] = pos;
      const int w = levelStart[lev + 1] - levelStart[lev];
      h->maxStageWidth = std::max(h->maxStageWidth, w);
      for (int k = levelStart[lev]; k < levelStart[lev + 1]; ++k) { h->int2ext[pos] = level[k]; h->ext2int[level[k]] = pos; sigma[pos] = s; ++pos; }
    }
    h->stageStart[nLev] = N;
    std::vector<int> upStart(N, 0);
    std::vector<uint8_t> nUp(N), nGood(N), isOut(N);
    std::vector<uint32_t> gmask(N, 0);
    const bool kwt = idxOf(h, MZR_KWT) >= 0;
    int wkNeed = 21;
    for (int i = 0; i < N; ++i) {
      const int e = h->int2ext[i];
      const int nu = upOffset[e + 1] - upOffset[e];
      if (nu > 32) return fail(h, 20, "mzr_set_network/more than 32 immediate upstream reaches are not supported");
      nUp[i] = (uint8_t)nu;
      isOut[i] = downIndex[e] <= 0;
      int ng = 0;
      for (int k = 0; k < nu; ++k) {
        const int good = upGood ? (upGood[upOffset[e] + k] != 0) : 1;
        if (good) { gmask[i] |= 1u << k; ++ng; }
        const int ui = h->ext2int[upIndex[upOffset[e] + k] - 1];
        if (k == 0) upStart[i] = ui;
        else if (ui != upStart[i] + k) return fail(h, 20, "mzr_set_network/internal ordering error");
      }
      nGood[i] = (uint8_t)ng;
    }
    if (kwt) {
      for (int i = 0; i < N; ++i) {
        if (nGood[i] == 0) continue;
        if (nUp[i] > MZR_MAX_UPSTREAM) return fail(h, 20, "mzr_set_network/KWT supports at most 8 immediate upstream reaches per reach");
        int nr = 0;
        for (int k = 0; k < nUp[i]; ++k) nr += nGood[upStart[i] + k] > 0;
        wkNeed = std::max(wkNeed, 20 + nUp[i] + 19 * nr);
      }
    }
    h->wk = wkNeed <= 64 ? 64 : wkNeed <= 128 ? 128 : 192;
    h->h_nGood = nGood; h->h_nUp = nUp;
    h->h_down.assign(N, -1);
    for (int i = 0; i < N; ++i) { const int dn = downIndex[h->int2ext[i]]; if (dn >= 1 && dn <= N) h->h_down[i] = h->ext2int[dn - 1]; }
    h->down.upload(h->h_down);
    std::vector<int> hOff(N + 1, 0), hIdx; std::vector<double> hW;
    for (int i = 0; i < N; ++i) {
      const int e = h->int2ext[i];
      for (int k = hruOffset[e]; k < hruOffset[e + 1]; ++k) {
        if (hruIndex[k] < 1 || hruIndex[k] > H) return fail(h, 20, "mzr_set_network/HRU index out of range");
        hIdx.push_back(hruIndex[k] - 1); hW.push_back(hruWeight[k]);
      }
      hOff[i + 1] = (int)hIdx.size();
    }
    h->reachId.assign(N, 0);
    for (int i = 0; i < N; ++i) h->reachId[i] = reachId ? reachId[i] : i + 1;
    h->h_sigma = sigma;
    h->sigma.upload(sigma); h->upStart.upload(upStart); h->nUp.upload(nUp); h->nGood.upload(nGood);
    h->goodMask.upload(gmask); h->isOutlet.upload(isOut);
    h->hruOff.upload(hOff); h->hruIdx.upload(hIdx); h->hruW.upload(hW);
    h->d_ext2int.upload(h->ext2int);
    for (int p = 0; p < 11; ++p) { h->par[p].alloc(N); h->par[p].zero(); }
    h->haveNet = true; h->haveState = false;
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_network/" + e); }
  return 0;
}

int mzr_set_param(mzr_handle h, const char *name, const double *values) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_param/network not set") : 1;
  (void)hipSetDevice(h->cfg.device);
  for (int p = 0; p < 11; ++p) {
    if (strcmp(name, mzr_domain::parName(p)) == 0) {
      std::vector<double> v(h->N);
      for (int i = 0; i < h->N; ++i) v[i] = values[h->int2ext[i]];
      (void)hipMemcpy(h->par[p].p, v.data(), h->N * sizeof(double), hipMemcpyHostToDevice);
      if (p == 0) h->h_slope = v;
      if (p == 1) h->h_mann = v;
      h->chanDirty = true;      // the channel table of the Eulerian solvers is derived from slope, n, width, depth and the two side slopes
      // the KWT records hold derived copies (width ratios, K, celerity factor, length): a change after
      // mzr_init_state must not go unnoticed -- the state has to be initialised (or restored) again
      if (h->haveState && h->kwN.p) h->haveState = false;
      return 0;
    }
  }
  return fail(h, 20, std::string("mzr_set_param/unknown parameter ") + name);
}

int mzr_set_uh(mzr_handle h, const int *uhOffset, const double *uh) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_uh/network not set") : 1;
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N;
  h->uhOff.assign(uhOffset, uhOffset + N + 1);
  int mx = 1;
  for (int e = 0; e < N; ++e) mx = std::max(mx, uhOffset[e + 1] - uhOffset[e]);
  h->maxtdh = mx;
  std::vector<double> pad((size_t)mx * N, 0.0);
  std::vector<uint16_t> nt(N);
  for (int i = 0; i < N; ++i) {
    const int e = h->int2ext[i];
    const int n = uhOffset[e + 1] - uhOffset[e];
    if (n < 1) return fail(h, 20, "mzr_set_uh/empty unit hydrograph");
    nt[i] = (uint16_t)n;
    for (int j = 0; j < n; ++j) pad[(size_t)j * N + i] = uh[uhOffset[e] + j];
  }
  try { h->uh.upload(pad); h->ntdh.upload(nt); } catch (const std::string &e) { return fail(h, 91, "mzr_set_uh/" + e); }
  return 0;
}

int mzr_set_frac_future(mzr_handle h, int n, const double *frac) {
  MZR_FLUSH(h);
  if (!h || n < 1) return 1;
  (void)hipSetDevice(h->cfg.device);
  h->ntdhBas = n;
  try {
    h->fracFuture.upload(std::vector<double>(frac, frac + n));
    std::vector<double> pad((size_t)n + 64, 0.0);
    for (int k = 0; k < n; ++k) pad[32 + k] = frac[k];
    h->fracPad.upload(pad);
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_frac_future/" + e); }
  return 0;
}

int mzr_set_lakes(mzr_handle h, int LakeInputOption, int calendarId, int nLake, const int *lakeReach, const int *modelType, const double *par) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_lakes/network not set") : 1;
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N;
  std::vector<int> slot(N, -1), lri(nLake, 0);
  h->h_lakeModel.assign(modelType, modelType + nLake);
  h->h_lakePar.assign(par, par + (size_t)MZR_NLAKEPAR * nLake);
  int L = 0, LD = 0;
  for (int l = 0; l < nLake; ++l) {
    const int e = lakeReach[l] - 1;
    if (e < 0 || e >= N) return fail(h, 20, "mzr_set_lakes/lake reach index out of range");
    if (modelType[l] < 0 || modelType[l] > 3) return fail(h, 20, "lake_route/unable to identify the parametric lake model type");
    slot[h->ext2int[e]] = l; lri[l] = h->ext2int[e];
    if (modelType[l] == 2 && h->cfg.nRoutes > 1 && !h->cfg.lakeMemoryPerMethod && (par[(size_t)52 * nLake + l] != 0.0 || par[(size_t)53 * nLake + l] != 0.0))
      return fail(h, 20, "mzr_set_lakes/Hanasaki reservoirs with inflow or demand memory and more than one routing method: the reference's methods share "
                         "one set of mutable parameters per lake (lake_route.f90:258-276,360), this library keeps one per method; set "
                         "mzr_config.lakeMemoryPerMethod = 1 to accept that, or route one method per handle");
    if (modelType[l] == 2 && par[(size_t)52 * nLake + l] != 0.0)
      L = std::max(L, (int)std::floor(par[(size_t)54 * nLake + l] * 31 * 86400.0 / h->cfg.dt));
    if (modelType[l] == 2 && par[(size_t)53 * nLake + l] != 0.0)
      LD = std::max(LD, (int)std::floor(par[(size_t)55 * nLake + l] * 31 * 86400.0 / h->cfg.dt));
  }
  try {
    h->nLake = nLake; h->LakeInputOption = LakeInputOption; h->calendarId = calendarId; h->lakeL = L; h->lakeLD = LD;
    h->anyLakeTarget = false; h->volJumpstart = 0; h->lakeTarg.upload(std::vector<int>(std::max(nLake, 1), 0));
    h->h_lakeSlot = slot;
    h->lakeSlot.upload(slot); h->lakeModel.upload(h->h_lakeModel); h->lakePar.upload(h->h_lakePar); h->lakeReachInt.upload(lri);
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_lakes/" + e); }
  h->haveState = false;
  return 0;
}

// evaporation / precipitation of the window on the river-network HRUs (host or device memory) -> per-lake fluxes.
// With LakeInputOption = 1 (runoff only, lake_route.f90:146-160) the two fluxes are not used and not moved.
static int set_lake_forcing(mzr_handle h, int nSteps, const double *evap, const double *precip, bool onDevice,
                            const int *month, const int *day, const int *dayofyear) {
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_lake_forcing/state not initialised") : 1;
  if (!h->nLake) return fail(h, 20, "mzr_set_lake_forcing/no lakes in this domain");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_set_lake_forcing/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  hipStream_t st = h->stream;
  const bool fluxes = h->LakeInputOption != 1;
  if (fluxes && (!evap || !precip)) return fail(h, 20, "mzr_set_lake_forcing/evaporation and precipitation are needed unless LakeInputOption = 1");
  const double *fe = evap, *fp = precip;
  if (fluxes && !onDevice) {
    const size_t need = (size_t)h->cfg.maxWindow * h->H;
    if (h->lakeFE.n < need) {
      try { h->lakeFE.alloc(need); h->lakeFP.alloc(need); } catch (const std::string &e) { return fail(h, 91, "mzr_set_lake_forcing/" + e); }
    }
    (void)hipMemcpyAsync(h->lakeFE.p, evap, (size_t)nSteps * h->H * sizeof(double), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(h->lakeFP.p, precip, (size_t)nSteps * h->H * sizeof(double), hipMemcpyHostToDevice, st);
    fe = h->lakeFE.p; fp = h->lakeFP.p;
  }
  // the last window's final launches may still be kept back (overlapping windows) and read ITS forcing: the next window's
  // goes into the second set of buffers, which run_window swaps in
  const bool beside = h->tail.pending;
  if (beside && !h->calMonthAlt.p) {
    try {
      h->lakeEvapAlt.alloc(h->lakeEvap.n); h->lakePrecipAlt.alloc(h->lakePrecip.n); h->lakeEvapAlt.zero(st); h->lakePrecipAlt.zero(st);
      h->calMonthAlt.alloc(h->calMonth.n); h->calDayAlt.alloc(h->calDay.n); h->calDoyAlt.alloc(h->calDoy.n);
    } catch (const std::string &e) { return fail(h, 91, "mzr_set_lake_forcing/" + e); }
  }
  (void)hipMemcpyAsync(beside ? h->calMonthAlt.p : h->calMonth.p, month, nSteps * sizeof(int), hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(beside ? h->calDayAlt.p : h->calDay.p, day, nSteps * sizeof(int), hipMemcpyHostToDevice, st);
  (void)hipMemcpyAsync(beside ? h->calDoyAlt.p : h->calDoy.p, dayofyear, nSteps * sizeof(int), hipMemcpyHostToDevice, st);
  if (fluxes) {
    MzrDev d; fillDev(h, d);
    mzr_launch_lake_forcing(d, h->lakeReachInt.p, fe, fp, beside ? h->lakeEvapAlt.p : h->lakeEvap.p, beside ? h->lakePrecipAlt.p : h->lakePrecip.p, nSteps, st);
  }
  h->lakeNextInAlt = beside;
  if (hipStreamSynchronize(st) != hipSuccess) return fail(h, 92, "mzr_set_lake_forcing/device error");
  h->lakeSteps = nSteps;
  return checkDeviceError(h);
}
int mzr_set_lake_forcing(mzr_handle h, int nSteps, const double *evap, const double *precip, const int *month, const int *day, const int *dayofyear) {
  if (MZR_STAGES(h, nSteps) && h->nLake && month && day && dayofyear && (h->LakeInputOption == 1 || (evap && precip))) {
    const bool fluxes = h->LakeInputOption != 1;
    return stageRow(h, mzr_domain::SR_LAKE, fluxes ? evap : nullptr, h->H, fluxes ? precip : nullptr, h->H, month, day, dayofyear);
  }
  MZR_FLUSH_STEPS(h);      // (a window kept back keeps its own lake forcing: set_lake_forcing writes beside it)
  return set_lake_forcing(h, nSteps, evap, precip, false, month, day, dayofyear);
}
int mzr_set_lake_forcing_dev(mzr_handle h, int nSteps, const double *evap_dev, const double *precip_dev, const int *month, const int *day, const int *dayofyear) {
  MZR_FLUSH_STEPS(h);      // (a window kept back keeps its own lake forcing: set_lake_forcing writes beside it)
  return set_lake_forcing(h, nSteps, evap_dev, precip_dev, true, month, day, dayofyear);
}

// Lakes that follow a target volume (is_vol_wm: NETOPO%LakeTargVol, lake_route.f90:197-205; jump start :140-142)
int mzr_set_lake_target(mzr_handle h, const int *targVol, int jumpstart) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_lake_target/network not set") : 1;
  if (!h->nLake) return fail(h, 20, "mzr_set_lake_target/no lakes in this domain (mzr_set_lakes first)");
  (void)hipSetDevice(h->cfg.device);
  std::vector<int> f(targVol, targVol + h->nLake);
  h->anyLakeTarget = std::any_of(f.begin(), f.end(), [](int x) { return x != 0; });
  h->volJumpstart = jumpstart != 0;
  try { h->lakeTarg.upload(f); } catch (const std::string &e) { return fail(h, 91, "mzr_set_lake_target/" + e); }
  return 0;
}

__global__ void k_gather_lake_rows(const double *src, double *dst, const int *lakeReachInt, const int *ext2int_unused, int N, int nLake, int rows) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (l < nLake && t < rows) dst[(size_t)t * nLake + l] = src[(size_t)t * N + lakeReachInt[l]];
}

// REACH_WM_VOL of the next window: vol[nSteps][nRch] in the caller's reach order (main_route.f90:115-122); only lake reaches are read
int mzr_set_wm_vol(mzr_handle h, int nSteps, const double *vol) {
  if (MZR_STAGES(h, nSteps) && h->nLake && vol) return stageRow(h, mzr_domain::SR_WMVOL, vol, h->N, nullptr, 0, nullptr, nullptr, nullptr);
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_wm_vol/state not initialised") : 1;
  if (!h->nLake) return fail(h, 20, "mzr_set_wm_vol/no lakes in this domain");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_set_wm_vol/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N;
  if (!ensureScratch(h)) return fail(h, 91, "mzr/out of device memory (row scratch)");
  (void)hipMemcpyAsync(h->scratchOut.p, vol, (size_t)nSteps * N * sizeof(double), hipMemcpyHostToDevice, h->stream);
  // scratchOut is in the caller's order: row t, reach e -> the lake's external index
  std::vector<int> ext(h->nLake);
  std::vector<int> lri(h->nLake);
  (void)hipMemcpy(lri.data(), h->lakeReachInt.p, h->nLake * sizeof(int), hipMemcpyDeviceToHost);
  for (int l = 0; l < h->nLake; ++l) ext[l] = h->int2ext[lri[l]];
  DBuf<int> dext; try { dext.upload(ext); } catch (const std::string &e) { return fail(h, 91, "mzr_set_wm_vol/" + e); }
  dim3 block(64), grid((h->nLake + 63) / 64, nSteps);
  hipLaunchKernelGGL(k_gather_lake_rows, grid, block, 0, h->stream, h->scratchOut.p, h->lakeWmVol.p, dext.p, (const int *)nullptr, N, h->nLake, nSteps);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, 92, "mzr_set_wm_vol/device error");
  h->wmVolSteps = nSteps;
  return 0;
}

static int pullRow(mzr_handle h, const double *src, double *out);
// Constituent routing (public_var tracer = T): a conservative constituent enters with the runoff (mass flux per HRU and
// step), takes the hillslope delay and is routed reach by reach with the water of every active method except the runoff
// accumulation (main_route.f90:161-172,204-236,392-401, basinUH.f90:130-137, tracer.f90:43-207).  Call after
// mzr_init_state; on = 0 switches it off.  Costs three more window buffers per method and a second pass over the window.
int mzr_set_tracer(mzr_handle h, int on, double time_conv_solute, double mass_conv_solute) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_tracer/state not initialised (call mzr_init_state)") : 1;
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  h->tracer = on ? 1 : 0; h->time_conv_solute = time_conv_solute; h->mass_conv_solute = mass_conv_solute; h->solSteps = 0; h->solCur = 0;
  if (!h->tracer) return 0;
  const size_t N = h->N, W = h->cfg.maxWindow;
  try {
    h->solSrc.alloc(W * h->H); h->basSol.alloc((W + 1) * N); h->basSol.zero();
    if (h->cfg.doesBasinRoute == 1) {
      h->solInst.alloc(W * N);
      h->solS[0].alloc((size_t)h->ntdhBas * N); h->solS[1].alloc((size_t)h->ntdhBas * N); h->solS[0].zero(); h->solS[1].zero();
    }
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      if (rb.method == MZR_SUM) continue;
      rb.solFlux.alloc(W * N); rb.solFlux.zero(); rb.solMass.alloc(N); rb.solMass.zero(); rb.trVol0.alloc(W * N); rb.trVol0.zero();
    }
    if (!h->trEvent) (void)hipEventCreateWithFlags(&h->trEvent, hipEventDisableTiming);
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_tracer/" + e); }
  return 0;
}

// basin constituent mass flux of the next window, solute[nSteps][nHru] in the order of the runoff
int mzr_set_solute(mzr_handle h, int nSteps, const double *solute) {
  if (MZR_STAGES(h, nSteps) && h->tracer && solute) return stageRow(h, mzr_domain::SR_SOLUTE, solute, h->H, nullptr, 0, nullptr, nullptr, nullptr);
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_solute/state not initialised") : 1;
  if (!h->tracer) return fail(h, 20, "mzr_set_solute/constituent routing is off (mzr_set_tracer)");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_set_solute/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  MZR_COPY(h->solSrc.p, solute, (size_t)nSteps * h->H * sizeof(double), hipMemcpyHostToDevice, "mzr_set_solute");
  h->solSteps = nSteps;
  return 0;
}

// which = 0: reach_solute_flux of the last routed step, 1: reach_solute_mass(1) (caller's reach order)
int mzr_get_solute(mzr_handle h, int method, int which, double *out) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_solute/state not initialised") : 1;
  if (!h->tracer) return fail(h, 20, "mzr_get_solute/constituent routing is off");
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0 || method == MZR_SUM) return fail(h, 81, "mzr_get_solute/method not active (or the runoff accumulation)");
  if (h->lastW < 1) return fail(h, 20, "mzr_get_solute/no step has been routed");
  RouteBufs &rb = h->route[ix];
  return pullRow(h, which == 0 ? rb.solFlux.p + (size_t)(h->lastW - 1) * h->N : rb.solMass.p, out);
}

// reach_solute_flux of every step of the last window, out[nSteps][nRch]
int mzr_get_window_solute(mzr_handle h, int method, double *out) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_window_solute/state not initialised") : 1;
  if (!h->tracer) return fail(h, 20, "mzr_get_window_solute/constituent routing is off");
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = method < 0 ? -1 : idxOf(h, method);
  if (method >= 0 && (ix < 0 || method == MZR_SUM)) return fail(h, 81, "mzr_get_window_solute/method not active (or the runoff accumulation)");
  const int N = h->N, W = h->lastW;
  if (W < 1) return fail(h, 20, "mzr_get_window_solute/no window has been run");
  dim3 block(256), grid((N + 255) / 256, W);
  const double *src = method < 0 ? h->basSol.p + N : h->route[ix].solFlux.p;      // method < 0: BASIN_solute (rows 1..W)
  if (!ensureScratch(h)) return fail(h, 91, "mzr/out of device memory (row scratch)");
  hipLaunchKernelGGL(k_gather_rows, grid, block, 0, h->stream, src, h->scratchOut.p, h->d_ext2int.p, N, W);
  if (hipMemcpyAsync(out, h->scratchOut.p, (size_t)W * N * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess)
    return fail(h, 92, "mzr_get_window_solute/hipMemcpy failed");
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, 92, "mzr_get_window_solute/sync failed");
  return 0;
}

// Direct insertion of gauge observations (public_var qmodOption = 1, qBlendPeriod, QerrTrend; main_route.f90:125-148,
// data_assimilation.f90:28-97) for IRF, KW, MC and DW (the reference's KWT and lake solvers do not call it).
// gaugeReach: 1-based reach (caller's order) of every gauge, < 1 = the gauge is not in this network.  Resets Qobs,
// Qelapsed and Qerror.  nGauge = 0 switches it off.
int mzr_set_da(mzr_handle h, int qBlendPeriod, int QerrTrend, int nGauge, const int *gaugeReach) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_da/state not initialised (call mzr_init_state)") : 1;
  if (nGauge < 0 || (nGauge > 0 && !gaugeReach)) return fail(h, 20, "mzr_set_da/bad gauge list");
  if (nGauge > 0 && (QerrTrend < 1 || QerrTrend > 4)) return fail(h, 81, "direct_insertion/discharge error trend model must be 1(const),2(liear), or 3(logistic)");
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  h->qmod = nGauge > 0 ? 1 : 0; h->qBlendPeriod = qBlendPeriod; h->QerrTrend = QerrTrend; h->nGauge = nGauge; h->obsSteps = 0;
  if (!h->qmod) return 0;
  const int N = h->N;
  std::vector<int> first(N, -1), next(nGauge, -1), last(N, -1);
  for (int g = 0; g < nGauge; ++g) {
    const int e = gaugeReach[g] - 1;
    if (e < 0 || e >= N) continue;
    const int i = h->ext2int[e];
    if (first[i] < 0) first[i] = g; else next[last[i]] = g;
    last[i] = g;
  }
  try {
    h->gaugeFirst.upload(first); h->gaugeNext.upload(next);
    h->obsHave.alloc(h->cfg.maxWindow); h->obsVal.alloc((size_t)h->cfg.maxWindow * nGauge);
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      rb.qobs.alloc(N); rb.qobs.zero(); rb.qerr.alloc(N); rb.qerr.zero(); rb.qelapsed.alloc(N); rb.qelapsed.zero();
    }
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_da/" + e); }
  return 0;
}

// gauge observations of the next window: have[nSteps] (1 = there is an observation time at this step), obs[nSteps][nGauge]
// (NaN or negative = no value at this gauge)
int mzr_set_obs(mzr_handle h, int nSteps, const int *have, const double *obs) {
  if (MZR_STAGES(h, nSteps) && h->qmod && have && obs) return stageRow(h, mzr_domain::SR_OBS, obs, h->nGauge, nullptr, 0, have, nullptr, nullptr);
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_obs/state not initialised") : 1;
  if (!h->qmod) return fail(h, 20, "mzr_set_obs/direct insertion is off (mzr_set_da)");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_set_obs/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);       // the window before may still read the buffers
  for (int ix = 0; ix < h->cfg.nRoutes; ++ix) if (h->routeStream[ix]) (void)hipStreamSynchronize(h->routeStream[ix]);
  MZR_COPY(h->obsHave.p, have, (size_t)nSteps * sizeof(int), hipMemcpyHostToDevice, "mzr_set_obs");
  MZR_COPY(h->obsVal.p, obs, (size_t)nSteps * h->nGauge * sizeof(double), hipMemcpyHostToDevice, "mzr_set_obs");
  h->obsSteps = nSteps;
  return 0;
}

int mzr_set_boundary(mzr_handle h, int nExport, const int *exportReach, int nHalo, const int *haloReach, const int *haloGood) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_boundary/network not set") : 1;
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N;
  std::vector<int> hs(N, -1), es(N, -1);
  h->h_expInt.assign(nExport, 0); h->h_haloInt.assign(nHalo, 0); h->h_haloGood.assign(nHalo, 0);
  std::vector<uint8_t> ng = h->h_nGood;
  for (int b = 0; b < nExport; ++b) {
    const int e = exportReach[b] - 1;
    if (e < 0 || e >= N) return fail(h, 20, "mzr_set_boundary/export reach index out of range");
    const int i = h->ext2int[e];
    es[i] = b; h->h_expInt[b] = i;
  }
  for (int b = 0; b < nHalo; ++b) {
    const int e = haloReach[b] - 1;
    if (e < 0 || e >= N) return fail(h, 20, "mzr_set_boundary/halo reach index out of range");
    const int i = h->ext2int[e];
    if (h->h_nUp[i] != 0) return fail(h, 20, "mzr_set_boundary/a halo reach must not have upstream reaches in this domain");
    hs[i] = b; h->h_haloInt[b] = i; h->h_haloGood[b] = (haloGood[b] & 1) | (haloGood[b] & 2);      // bit 0: good; bit 1: a lake where it is routed
    ng[i] = (haloGood[b] & 1) ? 1 : 0;      // what its downstream reach sees: count(goodBas) of the full network
  }
  if (nHalo > 0 && !h->highPriority) {
    // a domain that consumes halo records is the mainstem of a partitioned network: few reaches, many
    // stages, on the critical path of the next exchange, and it shares the GPU with a tributary domain
    // whose launches fill every slot -- its small launches go to a high-priority stream
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
      hipStream_t st2 = nullptr;
      if (hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, hi) == hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
        h->stream = st2; h->highPriority = true;
      }
    }
  }
  try {
    h->nExp = nExport; h->nHalo = nHalo;
    h->h_haloSlot = hs;
    h->haloSlot.upload(hs); h->exportSlot.upload(es);
    h->expInt.upload(h->h_expInt); h->haloInt.upload(h->h_haloInt);
    h->h_nGood = ng; h->nGood.upload(ng);
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_boundary/" + e); }
  h->haveState = false;
  return 0;
}

long long mzr_boundary_size(mzr_handle h, int nSteps, int nReach) {
  if (!h) return -1;
  const long long R = h->cfg.nRoutes, W = nSteps, B = nReach;
  const int hasKwt = idxOf(h, MZR_KWT) >= 0;
  return MZR_REC_HDR + R * W * B + recKwtPart(W, B, hasKwt) + (h->tracer ? R * W * B : 0);      // (header; discharge; the KWT part; reach_solute_flux while the tracer is on)
}

/* The record of the window BEFORE the last one, while the last launches of the last one are still kept back (overlapping windows of
   the Eulerian methods): its rows are the second set, complete from launch nS - 1 of the last window on; packed on a stream of its own
   behind exactly that, so the record is there long before the last window is.  (mpi_process.f90:1281-1312 ships every step's outlet
   fluxes before the mainstem's step; here a tributary domain's windows overlap, and the record travels one window later.) */
int mzr_export_boundary_prev_dev(mzr_handle h, double *rec_dev) {
  MZR_FLUSH_STEPS(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_export_boundary_prev/state not initialised") : 1;
  if (h->nExp == 0) return 0;
  if (!h->tail.pending || !h->prevInAlt || h->prevW < 1) return fail(h, 20, "mzr_export_boundary_prev/the rows of the window before the last one are not kept (mzr_get_export_lag)");
  (void)hipSetDevice(h->cfg.device);
  if (!h->expStream && hipStreamCreateWithFlags(&h->expStream, hipStreamNonBlocking) != hipSuccess) return fail(h, 90, "mzr_export_boundary_prev/hipStreamCreate failed");
  for (int ix = 0; ix < h->cfg.nRoutes && ix < 6; ++ix) if (h->prevRowsEv[ix]) (void)hipStreamWaitEvent(h->expStream, h->prevRowsEv[ix], 0);
  QPtrs q; for (int m = 0; m < 6; ++m) q.p[m] = m < h->cfg.nRoutes ? h->route[m].Qalt.p : nullptr;
  dim3 block(64), grid((h->nExp + 63) / 64, h->prevW);      // (overlapping windows: no KWT among the methods, no row W)
  hipLaunchKernelGGL(k_pack_boundary, grid, block, 0, h->expStream, rec_dev, h->cfg.nRoutes, h->prevW, h->nExp, h->N,
                     h->expInt.p, q, h->qlatAlt.p, h->exN.p, h->exOQ.p, h->exOT.p, 0, 0);
  if (!h->exportPrevDone) (void)hipEventCreateWithFlags(&h->exportPrevDone, hipEventDisableTiming);
  (void)hipEventRecord(h->exportPrevDone, h->expStream);
  h->exportOnAux = true;
  h->prevInAlt = false;      // (exported once)
  return hipGetLastError() == hipSuccess ? 0 : fail(h, 92, "mzr_export_boundary_prev/launch failed");
}
/* 1: the last window's final launches are kept back for the next window -- its record is exported by mzr_export_boundary_prev_dev
   after the next mzr_run* of the same length (or by mzr_export_boundary_dev, which issues the kept-back launches first) */
int mzr_get_export_lag(mzr_handle h) { return (h && h->tail.pending && h->nExp > 0) ? 1 : 0; }
/* the host waits for the handle's last export (either kind) and for nothing else the handle has queued */
int mzr_wait_export(mzr_handle h) {
  if (!h) return 1;
  if (!h->exportDone && !h->exportPrevDone) return 0;
  (void)hipSetDevice(h->cfg.device);
  if (h->exportPrevDone && hipEventSynchronize(h->exportPrevDone) != hipSuccess) return fail(h, 92, "mzr_wait_export/device error");
  if (h->exportDone && hipEventSynchronize(h->exportDone) != hipSuccess) return fail(h, 92, "mzr_wait_export/device error");
  return 0;
}

int mzr_export_boundary_dev(mzr_handle h, double *rec_dev) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_export_boundary/state not initialised") : 1;
  if (h->nExp == 0) return 0;
  if (h->lastW < 1) return fail(h, 20, "mzr_export_boundary/no window has been run");
  (void)hipSetDevice(h->cfg.device);
  QPtrs q; for (int m = 0; m < 6; ++m) q.p[m] = m < h->cfg.nRoutes ? h->route[m].Q.p : nullptr;
  const int hasKwt = h->kwN.p ? 1 : 0;
  dim3 block(64), grid((h->nExp + 63) / 64, h->lastW + hasKwt);      // (row W carries the last BASIN_QR row: KWT only)
  hipLaunchKernelGGL(k_pack_boundary, grid, block, 0, h->stream, rec_dev, h->cfg.nRoutes, h->lastW, h->nExp, h->N,
                     h->expInt.p, q, h->qlat.p, h->exN.p, h->exOQ.p, h->exOT.p, hasKwt, h->tracer ? 1 : 0);
  if (h->tracer) {
    QPtrs f; for (int m = 0; m < 6; ++m) f.p[m] = m < h->cfg.nRoutes ? h->route[m].solFlux.p : nullptr;
    const long long base = MZR_REC_HDR + (long long)h->cfg.nRoutes * h->lastW * h->nExp + recKwtPart(h->lastW, h->nExp, hasKwt);
    hipLaunchKernelGGL(k_pack_solute, dim3((h->nExp + 63) / 64, h->lastW), block, 0, h->stream, rec_dev + base, h->cfg.nRoutes, h->lastW, h->nExp, h->N, h->expInt.p, f);
  }
  if (!h->exportDone) (void)hipEventCreateWithFlags(&h->exportDone, hipEventDisableTiming);
  (void)hipEventRecord(h->exportDone, h->stream);
  return hipGetLastError() == hipSuccess ? 0 : fail(h, 92, "mzr_export_boundary/launch failed");
}

int mzr_import_boundary_dev(mzr_handle h, int nSteps, const double *rec_dev, int nSrc, int haloBase) {
  MZR_FLUSH_STEPS(h);
  // Round 6: a mainstem domain of the Eulerian methods keeps its windows overlapping too (mpi_process.f90:1281-1312 is one serial
  // sweep of the mainstem per step on rank 0; here its nStages + W - 1 dependent launches per window were what rank 0 waited for).
  // The launches kept back of the window before still read THAT window's halo discharge, so the next window's goes into a second
  // buffer, which run_window swaps in -- the lake forcing's scheme (mzr_set_lake_forcing).
  if (h && h->tail.pending && (h->kwN.p || h->tracer)) flushTail(h);      // (cannot happen: such domains never keep launches back)
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_import_boundary/state not initialised") : 1;
  if (nSrc == 0) return 0;
  if (haloBase < 0 || haloBase + nSrc > h->nHalo) return fail(h, 20, "mzr_import_boundary/halo slot range out of bounds");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_import_boundary/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  const bool beside = h->tail.pending || h->imNextInAlt;      // (every import of one window goes to the same buffer)
  if (beside) {
    try {
      for (int m = 0; m < h->cfg.nRoutes; ++m) if (!h->route[m].imQAlt.p) { h->route[m].imQAlt.alloc(h->route[m].imQ.n); h->route[m].imQAlt.zero(h->stream); }
    } catch (const std::string &e) { return fail(h, 91, "mzr_import_boundary/" + e); }
    h->imNextInAlt = true;
  }
  QPtrsW q; for (int m = 0; m < 6; ++m) q.p[m] = m < h->cfg.nRoutes ? (beside ? h->route[m].imQAlt.p : h->route[m].imQ.p) : nullptr;
  const int hasKwt = h->kwN.p ? 1 : 0;
  dim3 block(64), grid((nSrc + 63) / 64, nSteps + hasKwt);
  hipLaunchKernelGGL(k_unpack_boundary, grid, block, 0, h->stream, rec_dev, h->cfg.nRoutes, nSteps, nSrc, h->N, h->nHalo,
                     haloBase, h->haloInt.p, q, h->qlat.p, h->imN.p, h->imOQ.p, h->imOT.p, hasKwt, h->tracer ? 1 : 0, h->err.p);
  if (h->tracer) {      // the halo reaches' reach_solute_flux goes straight into the window's rows: the constituent pass skips halo reaches
    QPtrsW f; for (int m = 0; m < 6; ++m) f.p[m] = m < h->cfg.nRoutes ? h->route[m].solFlux.p : nullptr;
    const long long base = MZR_REC_HDR + (long long)h->cfg.nRoutes * nSteps * nSrc + recKwtPart(nSteps, nSrc, hasKwt);
    hipLaunchKernelGGL(k_unpack_solute, dim3((nSrc + 63) / 64, nSteps), block, 0, h->stream, rec_dev, rec_dev + base, h->cfg.nRoutes, nSteps, nSrc, h->N, haloBase, h->haloInt.p, f, hasKwt);
  }
  if (!h->importDone) (void)hipEventCreateWithFlags(&h->importDone, hipEventDisableTiming);
  (void)hipEventRecord(h->importDone, h->stream);
  return hipGetLastError() == hipSuccess ? 0 : fail(h, 92, "mzr_import_boundary/launch failed");
}
/* the host waits until the handle's last mzr_import_boundary_dev has read its record (the buffer may go) -- and for nothing the
   handle keeps back: unlike mzr_sync it leaves overlapping windows overlapping */
int mzr_wait_import(mzr_handle h) {
  if (!h) return 1;
  if (!h->importDone) return 0;
  (void)hipSetDevice(h->cfg.device);
  return hipEventSynchronize(h->importDone) == hipSuccess ? 0 : fail(h, 92, "mzr_wait_import/device error");
}

int mzr_init_state(mzr_handle h) {
  MZR_FLUSH(h);
  if (h) h->rtTablesW = -1;
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_init_state/network not set") : 1;
  (void)hipSetDevice(h->cfg.device);
  const size_t N = h->N, W = h->cfg.maxWindow;
  if (h->cfg.doesBasinRoute == 1 && h->ntdhBas < 1) return fail(h, 20, "mzr_init_state/FRAC_FUTURE not set");
  try {
    h->runoffW.free();      // (made on first use: ensureRunoffW)
    // what earlier windows of this handle left behind in buffers sized for the old network / maxWindow: the second set of rows of
    // the overlapping windows, the retry snapshot, a window kept back, lake forcing written beside a window kept back
    h->qlatAlt.free(); h->qiAlt.free();
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) h->route[ix].Qalt.free();
    h->lakeEvapAlt.free(); h->lakePrecipAlt.free(); h->calMonthAlt.free(); h->calDayAlt.free(); h->calDoyAlt.free();
    h->lakeNextInAlt = false; h->imNextInAlt = false;
    h->snapN.free(); h->snapQ.free(); h->snapTR.free(); h->snapQsum.free(); h->snapHIn.free();
    h->retry.q.clear(); h->retry.seen = 0;
    h->tail.pending = false;
    if (h->cfg.doesBasinRoute == 1) {
      h->qi.alloc(W * N); h->qi.zero();      // halo reaches have no HRUs of their own: their rows stay zero (basin state getters)
      h->basS[0].alloc((size_t)h->ntdhBas * N); h->basS[1].alloc((size_t)h->ntdhBas * N);
      h->basS[0].zero(); h->basS[1].zero();
    }
    h->basCur = 0;
    h->qlat.alloc((W + 1) * N); h->qlat.zero();
    h->qr0Last.alloc(N); h->qr0Last.zero();
    h->scratchOut.free();   // (made on first use: ensureScratch)
    if (h->cfg.is_flux_wm) { h->wm.alloc(W * N); h->wm.zero(); h->wmSteps = 0; }
    h->err.alloc(1); h->err.zero();
    if (h->nLake) {
      h->lakeEvap.alloc(W * h->nLake); h->lakePrecip.alloc(W * h->nLake); h->lakeEvap.zero(); h->lakePrecip.zero();
      h->lakeFE.free(); h->lakeFP.free();      // staging of host-side evaporation / precipitation: on first use
      h->calMonth.alloc(W); h->calDay.alloc(W); h->calDoy.alloc(W);
      h->lakeWmVol.alloc(W * h->nLake); h->lakeWmVol.zero(); h->wmVolSteps = 0;
    }
    const bool kwtAmong = idxOf(h, MZR_KWT) >= 0;      // (particle rows of the boundary reaches: only where KWT routes)
    h->imN.free(); h->imOQ.free(); h->imOT.free(); h->exN.free(); h->exOQ.free(); h->exOT.free();
    if (h->nHalo && kwtAmong) {
      h->imN.alloc(W * h->nHalo); h->imN.zero();
      h->imOQ.alloc(W * MZR_OB_CAP * h->nHalo); h->imOT.alloc(W * MZR_OB_CAP * h->nHalo); h->imOQ.zero(); h->imOT.zero();
    }
    if (h->nExp && kwtAmong) {
      h->exN.alloc(W * h->nExp); h->exN.zero();
      h->exOQ.alloc(W * MZR_OB_CAP * h->nExp); h->exOT.alloc(W * MZR_OB_CAP * h->nExp); h->exOQ.zero(); h->exOT.zero();
    }
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      const int m = rb.method;
      rb.Q.alloc(W * N); rb.Q.zero();
      if (h->nHalo) { rb.imQ.alloc(W * h->nHalo); rb.imQ.zero(); }
      rb.imQAlt.free();
      if (h->nLake) {   // mutable Hanasaki parameters start from the static ones: I_months, D_months, E_rel_ini
        std::vector<double> mut((size_t)25 * h->nLake);
        for (int l = 0; l < h->nLake; ++l) {
          for (int m2 = 0; m2 < 24; ++m2) mut[(size_t)m2 * h->nLake + l] = h->h_lakePar[(size_t)(27 + m2) * h->nLake + l];
          mut[(size_t)24 * h->nLake + l] = h->h_lakePar[(size_t)26 * h->nLake + l];
        }
        rb.lakeMut.upload(mut);
        if (h->lakeL > 0) { rb.lakeRing.alloc((size_t)h->nLake * 12 * h->lakeL); rb.lakeRing.zero(); rb.lakeHead.alloc((size_t)h->nLake * 13); rb.lakeHead.zero(); }
        if (h->lakeLD > 0) { rb.lakeRingD.alloc((size_t)h->nLake * 12 * h->lakeLD); rb.lakeRingD.zero(); rb.lakeHeadD.alloc((size_t)h->nLake * 13); rb.lakeHeadD.zero(); }
      }
      for (DBuf<double> *b : {&rb.vol, &rb.vol0, &rb.inflow, &rb.ele, &rb.floodvol, &rb.wb, &rb.qsum, &rb.wmact}) { b->alloc(N); b->zero(); }
      if (m != MZR_KWT) {      // persistent sweep of an Eulerian method
        rb.rtDone.alloc(N); rb.rtDone.zero(); rb.rtHead.alloc(8 * 16 + 16 + 32); rb.rtHead.zero();
        rb.rtCap = 0; rb.rtCapTried = false;      // (measured by the first window short enough to be swept: run_window)
        if (h->rtTablesW != -2) { rt_build_items(h); h->rtTablesW = -2; }      // once per mzr_init_state (-2: built, tables not yet)
      }
      if (m == MZR_KW || m == MZR_DW) { rb.mol.alloc((size_t)MZR_NMOL_KW * N); rb.mol.zero(); }
      if (m == MZR_MC) { rb.mol.alloc((size_t)MZR_NMOL_MC * N); rb.mol.zero(); rb.mcSub.alloc(N); rb.mcSub.zero(); rb.havePerm = false; }
      if (m == MZR_IRF) {
        if (h->maxtdh < 1) return fail(h, 20, "mzr_init_state/reach unit hydrographs not set (IRF)");
        h->irfQ.alloc((size_t)h->maxtdh * N); h->irfQ.zero();
        // (dealing the lanes to wavefronts by tap count -- build_lane_perm with ntdh as the key -- was measured slower for IRF once its
        // loads were batched: 5.3 against 6.0 x 10^9 reach-steps/s on the 625 k shard; the kernel is bound by what it fetches, and
        // a wavefront of scattered reaches fetches four times the sectors it uses.  MZR_IRF_PERM=1 switches it on.)
        if (const char *e = getenv("MZR_IRF_PERM")) if (atoi(e) != 0) {
          std::vector<uint16_t> nt(N);
          (void)hipMemcpy(nt.data(), h->ntdh.p, N * sizeof(uint16_t), hipMemcpyDeviceToHost);
          build_lane_perm(h, ix, std::vector<int>(nt.begin(), nt.end()));
        }
      }
      if (m == MZR_KWT) {
        if ((size_t)h->h_slope.size() != N || (size_t)h->h_mann.size() != N) return fail(h, 20, "mzr_init_state/R_SLOPE and R_MAN_N must be set before KWT state is initialised");
        // the particle rows are addressed with 32-bit byte offsets (16-byte buffer accesses): 384 bytes per reach
        if ((unsigned long long)N * 2ull * MZR_KW_STRIDE * sizeof(double) >= (1ull << 32)) return fail(h, 20, "mzr_init_state/KWT: more than 11 million reaches in one domain (partition the network)");
        std::vector<double> K(N), CW(N);
        {   // kinwav_rch constants, kwt_route.f90:1273-1274,1290: evaluated once, with the host libm
          const double ALFA = 5.0 / 3.0;
          for (size_t i = 0; i < N; ++i) { K[i] = std::sqrt(h->h_slope[i]) / h->h_mann[i]; CW[i] = ALFA * std::pow(K[i], 1.0 / ALFA); }
          h->kwK.upload(K); h->kwCW.upload(CW);
        }
        {   // reaches that route particles get a group of lanes each, the O(1) ones a lane each
          std::vector<double> width(N), length(N);
          (void)hipMemcpy(width.data(), h->par[2].p, N * sizeof(double), hipMemcpyDeviceToHost);
          (void)hipMemcpy(length.data(), h->par[4].p, N * sizeof(double), hipMemcpyDeviceToHost);
          std::vector<int> upStart(N); std::vector<uint32_t> gmask(N); std::vector<uint8_t> isOut(N);
          (void)hipMemcpy(upStart.data(), h->upStart.p, N * sizeof(int), hipMemcpyDeviceToHost);
          (void)hipMemcpy(gmask.data(), h->goodMask.p, N * sizeof(uint32_t), hipMemcpyDeviceToHost);
          (void)hipMemcpy(isOut.data(), h->isOutlet.p, N * sizeof(uint8_t), hipMemcpyDeviceToHost);
          std::vector<MzrKwtRec> routed, generic; std::vector<int> light, head, depLight;
          h->kwtRoutedOff.assign(h->nStages + 1, 0); h->kwtGenericOff.assign(h->nStages + 1, 0); h->kwtLightOff.assign(h->nStages + 1, 0);
          for (int sg = 0; sg < h->nStages; ++sg) {
            h->kwtRoutedOff[sg] = (int)routed.size(); h->kwtGenericOff[sg] = (int)generic.size(); h->kwtLightOff[sg] = (int)light.size();
            for (int i = h->stageStart[sg]; i < h->stageStart[sg + 1]; ++i) {
              const bool halo = h->nHalo && !h->h_haloSlot.empty() && h->h_haloSlot[i] >= 0;
              const bool lake = !h->h_lakeSlot.empty() && h->h_lakeSlot[i] >= 0;
              if (halo || lake || h->h_nGood[i] == 0) { light.push_back(i); (halo || lake ? depLight : head).push_back(i); continue; }
              MzrKwtRec rc; memset(&rc, 0, sizeof rc);
              rc.r = i; rc.sigma = sg; rc.u0 = upStart[i]; rc.nup = h->h_nUp[i]; rc.down = h->h_down[i];
              rc.flags = (uint8_t)(h->h_nGood[i] & 15) | (isOut[i] ? 0x80 : 0);
              rc.goodMask = (uint8_t)(gmask[i] & 0xff);
              int nsr = 0;
              for (int k = 0; k < rc.nup; ++k) {
                const int u = rc.u0 + k;
                if (!h->h_lakeSlot.empty() && h->h_lakeSlot[u] >= 0) rc.flags |= 0x40;
                if (h->nHalo && !h->h_haloSlot.empty() && h->h_haloSlot[u] >= 0) {
                  rc.flags |= 0x20;   // its progress word carries no particle counts
                  // a tributary outlet that is a LAKE where it is routed: its outflow enters this reach as one particle, and it
                  // has to be this reach's only upstream reach (getusq_rch, kwt_route.f90:540-559), exactly as if the lake were here
                  if (h->h_haloGood[h->h_haloSlot[u]] & 2) rc.flags |= 0x40;
                }
                if (h->h_nGood[u] > 0) {
                  rc.upGood |= (uint8_t)(1u << k);
                  if (nsr == 0) rc.scA = width[u] / width[i]; else if (nsr == 1) rc.scB = width[u] / width[i];
                  ++nsr;
                }
              }
              rc.width = width[i]; rc.CW = CW[i]; rc.length = length[i];
              if (rc.nup > 2) generic.push_back(rc); else routed.push_back(rc);
            }
          }
          h->kwtRoutedOff[h->nStages] = (int)routed.size(); h->kwtGenericOff[h->nStages] = (int)generic.size(); h->kwtLightOff[h->nStages] = (int)light.size();
          MzrKwtRec none; memset(&none, 0, sizeof none);
          if (routed.empty()) routed.push_back(none);
          if (generic.empty()) generic.push_back(none);
          if (light.empty()) light.push_back(0);
          h->kwtRouted.upload(routed); h->kwtRoutedB.upload(routed); h->kwtRoutedC.upload(routed); h->kwtRoutedAll.upload(routed); h->kwtRoutedBAll.upload(routed); h->kwtRoutedCAll.upload(routed); h->kwtAllValid = false; h->kwtGeneric.upload(generic); h->kwtLight.upload(light);
          h->h_kwtRouted = routed; h->kwtWindows = 0; h->kwtStepsSince = 0;
          h->kwtStageOff = h->kwtRoutedOff;                       // every routed reach starts in class A
          h->kwtBOff.assign(h->nStages + 1, 0); h->kwtCOff.assign(h->nStages + 1, 0);
          // persistent sweep: every routed reach starts in class A, in stage order
          h->h_swA = h->kwtRoutedOff[h->nStages] > 0 ? routed : std::vector<MzrKwtRec>();
          h->h_swB.clear(); h->h_swC.clear(); h->swKcWide = 0;
          h->h_kwtGeneric = h->kwtGenericOff[h->nStages] > 0 ? generic : std::vector<MzrKwtRec>();
          h->h_kwtHead = head; h->h_kwtDepLight = depLight;
          if (head.empty()) head.push_back(0);
          if (depLight.empty()) depLight.push_back(0);
          h->kwtHead.upload(head); h->kwtDepLight.upload(depLight);
          { std::vector<uint8_t> hf(N, 0); for (int r : h->h_kwtHead) hf[r] = 1; h->kwHeadFlag.upload(hf); }
          h->kwDone.alloc(N); h->kwDone.zero();
          h->swCap = 0;
          h->kwtHeadSteps = 0;
          kwt_build_sweep(h);
        }
        h->kwN.alloc(N); h->kwN.zero();
        h->kwQ.alloc((size_t)2 * MZR_KW_STRIDE * N); h->kwTR.alloc((size_t)MZR_KW_STRIDE * N);      // kwQ: {Q, TI} pairs
        h->kwQ.zero(); h->kwTR.zero();
        h->obN.alloc((size_t)MZR_OB_RING * N); h->obN.zero();
        h->obQ.alloc((size_t)MZR_OB_RING * 2 * MZR_OB_STRIDE * N);      // a ring of MZR_OB_RING steps of {Q, exit time} pairs
        h->obQ.zero();
        h->kwtStat.alloc(1); h->kwtStat.zero();
        h->dbgCycles.alloc(32 * 1024 + 8 + 65536 * 8); h->dbgCycles.zero();   // counters, then (timing builds) one record per sampled pass
      }
      rb.nLaunches = 0; rb.kernel_ms = 0; rb.reachSteps = 0; rb.meanSteps = 0;
      rb.hInflow.free(); rb.hEle.free(); rb.hFlood.free();
      if (h->histFlags & MZR_H_INFLOW) { rb.hInflow.alloc(N); rb.hInflow.zero(); }
      if (h->histFlags & MZR_H_HEIGHT) { rb.hEle.alloc(N); rb.hEle.zero(); rb.hFlood.alloc(N); rb.hFlood.zero(); }
    }
    h->hInst.free(); h->hDlay.free(); h->hBas.free(); h->histSteps = 0;
    if (h->histFlags & MZR_H_RUNOFF) {
      h->hInst.alloc(N); h->hInst.zero(); h->hDlay.alloc(N); h->hDlay.zero(); h->hBas.alloc(h->H); h->hBas.zero();
    }
  } catch (const std::string &e) { return fail(h, 91, "mzr_init_state/" + e); }
  if (hipDeviceSynchronize() != hipSuccess) return fail(h, 92, "mzr_init_state/device error");
  h->haveState = true; h->stepsDone = 0; h->lastW = 0; h->totalSteps = 0;
  return 0;
}

// The order of the reaches inside a stage is free.  A wavefront works on 64/G reaches in lockstep and
// pays for its busiest one (thinning iterations, second particle slot), and how many particles a
// reach holds changes slowly, so every now and then the routed list of each stage is regrouped by
// the particle counts of the last step: saturated reaches share wavefronts, light ones do too, and
// reaches that hold only a few particles go to the class that gives them 8 lanes instead of 16.
// One lane per reach in the Eulerian stage kernels: a wavefront lasts as long as its slowest lane.  Muskingum-Cunge reaches take
// 2 to 200 Courant sub-steps (mc_route.f90:283-330: ntSub = ceil(dt / L * celerity) -- short reaches with a fast wave, the same
// ones step after step), the IRF convolution 1 to maxtdh taps (irf_route.f90:210-264).  The reaches of every aligned block of 256
// lane positions are dealt to the block's four wavefronts by that count, largest first (lanePerm): slow lanes sit with slow
// lanes.  A permutation inside each block: every reach is served exactly once, by whichever lane -- results cannot change.
// heavyMin > 0: the reaches with a count of heavyMin and more (at most an eighth of all) are taken out of their blocks and fill lane
// positions of their own behind the others, heaviest first (mzr_device.h nHeavyPos; Muskingum-Cunge, MZR_MC_HEAVY_MIN)
static void build_lane_perm(mzr_handle h, int ix, const std::vector<int> &key, int heavyMin) {
  RouteBufs &rb = h->route[ix];
  const int N = h->N, Np = (N + 255) & ~255, heavyCap = (((N / 8) + 255) & ~255);
  std::vector<int> perm(Np + heavyCap, -1);
  std::vector<std::pair<int, int>> blk;
  bool any = false;
  std::vector<char> isHeavy(N, 0);
  int nHeavy = 0;
  try { if (!rb.lanePerm.p) rb.lanePerm.alloc(Np + heavyCap); } catch (const std::string &) { rb.havePerm = false; rb.nHeavyPos = 0; (void)hipGetLastError(); return; }      // allocated once: a window kept back may still point at it
  if (heavyMin > 0) {
    std::vector<std::pair<int, int>> hv;
    for (int r = 0; r < N; ++r) if (key[r] >= heavyMin) hv.emplace_back(-key[r], r);
    std::stable_sort(hv.begin(), hv.end());
    if ((int)hv.size() > heavyCap) hv.resize(heavyCap);
    for (size_t k = 0; k < hv.size(); ++k) { perm[Np + k] = hv[k].second; isHeavy[hv[k].second] = 1; }
    nHeavy = (int)hv.size();
    if (nHeavy > 0) any = true;
  }
  for (int b0 = 0; b0 < N; b0 += 256) {
    blk.clear();
    for (int r = b0; r < std::min(N, b0 + 256); ++r) if (!isHeavy[r]) blk.emplace_back(-key[r], r);
    if (blk.empty()) continue;
    std::stable_sort(blk.begin(), blk.end());
    if (blk.front().first != blk.back().first) any = true;
    for (size_t k = 0; k < blk.size(); ++k) perm[b0 + k] = blk[k].second;
  }
  if (const char *e = getenv("MZR_LANE_PERM")) if (atoi(e) == 0) any = false;
  try {
    if (any) (void)hipMemcpy(rb.lanePerm.p, perm.data(), (size_t)(Np + heavyCap) * sizeof(int), hipMemcpyHostToDevice);
    rb.havePerm = any; rb.permN = Np; rb.nHeavyPos = any ? ((nHeavy + 255) & ~255) : 0;
  } catch (const std::string &) { rb.havePerm = false; rb.nHeavyPos = 0; }
  // a window whose last launches are kept back (overlapping windows) goes out with the new positions
  if (h->tail.pending) { h->tail.d[ix].lanePerm = rb.havePerm ? rb.lanePerm.p : nullptr; h->tail.d[ix].permN = rb.permN; h->tail.d[ix].nHeavyPos = rb.havePerm ? rb.nHeavyPos : 0; }
}

static void mc_regroup(mzr_handle h, int ix) {
  RouteBufs &rb = h->route[ix];
  if (!rb.mcSub.p) return;
  const int N = h->N;
  (void)hipStreamSynchronize(h->stream);
  if (h->routeStream[ix]) (void)hipStreamSynchronize(h->routeStream[ix]);
  std::vector<unsigned short> sub(N);
  if (hipMemcpy(sub.data(), rb.mcSub.p, (size_t)N * sizeof(unsigned short), hipMemcpyDeviceToHost) != hipSuccess) return;
  std::vector<int> key(sub.begin(), sub.end());
  // (c4 shard, 625 k reaches IRF + MC, reach-steps/s by threshold: none 7.37, 4: 7.25, 6: 7.55, 10: 7.78-7.85, 16: 7.56, 24: 7.36 x 10^9;
  // profiles/r05_experiments.md)
  int heavyMin = 10;
  if (const char *e = getenv("MZR_MC_HEAVY_MIN")) heavyMin = atoi(e);
  if (getenv("MZR_MC_LOG")) {      // debugging aid: the sub-steps the reaches executed in their last step (the longest one is the launch)
    std::vector<int> srt(key); std::sort(srt.begin(), srt.end(), std::greater<int>());
    int c[5] = {0, 0, 0, 0, 0};
    for (int v : srt) { c[0] += v >= 10; c[1] += v >= 20; c[2] += v >= 40; c[3] += v >= 80; c[4] += v >= 160; }
    fprintf(stderr, "[mzr] MC sub-steps executed: top %d %d %d %d %d %d; reaches with >= 10 / 20 / 40 / 80 / 160: %d %d %d %d %d of %d\n",
            srt[0], srt[std::min(1, N - 1)], srt[std::min(2, N - 1)], srt[std::min(7, N - 1)], srt[std::min(63, N - 1)], srt[std::min(255, N - 1)], c[0], c[1], c[2], c[3], c[4], N);
  }
  build_lane_perm(h, ix, key, heavyMin);
}

static void kwt_regroup(mzr_handle h) {
  if (h->h_kwtRouted.size() < 2 || !h->kwN.p) return;
  (void)hipStreamSynchronize(h->stream);
  const int N = h->N;
  std::vector<int> n(N), ob((size_t)MZR_OB_RING * N);
  (void)hipMemcpy(n.data(), h->kwN.p, N * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipMemcpy(ob.data(), h->obN.p, (size_t)MZR_OB_RING * N * sizeof(int), hipMemcpyDeviceToHost);
  // work-array entries the reach needed in the last step (the kernel's `need`), from the larger outbox parity
  auto need = [&](const MzrKwtRec &rc) {
    int l = std::max(n[rc.r], 1) + rc.nup;
    for (int k = 0; k < rc.nup; ++k) if ((rc.upGood >> k) & 1) {
      int m = 0;
      for (int sl = 0; sl < MZR_OB_RING; ++sl) m = std::max(m, ob[(size_t)sl * N + rc.u0 + k]);
      l += std::max(m - 1, 0);
    }
    return l;
  };
  // class C (4 lanes, capacity 11 entries): reaches that needed at most 9 entries in the last step; class B (8 lanes,
  // capacity 30): at most 20 (wider lists thin, and a pass is as slow as its busiest reach: measured best); class A
  // (16 lanes): the others.  MZR_KWT_CLASSB_MAX / MZR_KWT_CLASSC_MAX override the
  // thresholds (tests: 0 = nobody, 64 = everybody, through the fall-back).
  int classBMax = 20, classCMax = 9;
  // Round 4: the cut between B and A depends on what bounds the sweep.  With many more items per launch than wavefronts
  // (the 375 k-reach shards: 25 k items for 4 000 wavefronts) the sweep is bound by VALU issue, and 8 reaches per pass
  // cost fewer instructions per reach than 4: class B up to 28 entries (its capacity is 30) measured +13 % there.  With
  // about as many items as wavefronts (100 k reaches: 6.7 k) the window is bound by its longest chain of passes, which
  // runs through the reaches that thin every step, and those are faster in 16-lane groups: 28 there measured -8 %.
  // (round 4, with the outbox ring of four steps and the split 16-lane pass the chain no longer punishes the narrower groups:
  // 24 there measured +4.6 %, 28 +-0)
  const bool byInstructions = h->swCap > 0 && (double)h->h_kwtRouted.size() / 7.0 > 3.0 * h->swCap;
  classBMax = byInstructions ? 28 : 24;
  // Round 5: the cut between C and B the same way.  The 4-lane groups' slice of the pool holds 15 entries, three slots per lane 11; the
  // sweep flavour with four slots per lane takes reaches of up to 13 entries sixteen to a pass instead of eight, and the fourth slot
  // costs every 4-lane pass: the 375 k shard (bound by instructions) 674.1 -> 655.4 ms per window of 8 192, 100 k reaches 442 -> 447.5 ms
  // (profiles/r05_experiments.md 9).  MZR_KWT_KC_WIDE_RUN forces it (tests).  The stage launches keep three slots (a reach beyond them
  // takes their wide fall-back).
  // (round 6: also where the window is bound by its chain -- 100 k reaches 441 -> 435 ms per window, twice, turn about, once thinning and LDS
  // addressing had been made cheaper: the fourth slot no longer costs what the reaches moved out of the 8-lane class save)
  int kcWide = 1;
  if (const char *e = getenv("MZR_KWT_KC_WIDE_RUN")) kcWide = atoi(e) != 0;
  h->swKcWide = kcWide;
  if (kcWide) classCMax = 13;
  { int capB = 0, capC = 0; (void)mzr_kwt_class_caps(&capB, &capC, kcWide); classBMax = std::min(classBMax, capB - 2); classCMax = std::min(classCMax, capC - 2); }      // (room to grow by two before the fall-back)
  if (const char *e = getenv("MZR_KWT_CLASSB_MAX")) classBMax = atoi(e);
  if (const char *e = getenv("MZR_KWT_CLASSC_MAX")) classCMax = atoi(e);
  const std::vector<MzrKwtRec> &v = h->h_kwtRouted;
  std::vector<MzrKwtRec> L[3];    // A, B, C
  for (auto &l : L) l.reserve(v.size());
  std::vector<std::pair<int, int>> key;
  // Round 6, the latency path: a reach's step t + 1 waits for its step t, so the reaches with the longest lists -- 25 removals a step, each
  // a chain of ~150 dependent instructions -- are the window's longest chain, and a pass is as long as its longest lane group: the 16-lane
  // reaches that needed 45 entries or more (40 of 100 000 on the benchmark network) have their pass to themselves.  100 k reaches: 399.4 ->
  // 387.0 ms per window of 16 384 (4 such reaches: the same; 294: the same; 581: slower than without -- the passes are paid for in wavefront
  // slots).  Where instructions bound the sweep and not its chain (the rule of the B / A cut above) the empty lane groups only cost: 375 k
  // shard 637.3 ms without, 644.2 with.  At most swCap / 32 of them (the threshold moves up).  MZR_KWT_SOLO_MIN=n (0 = off) /
  // MZR_KWT_SOLO_PER=k (k reaches in such a pass, 1) override.
  int soloMin = byInstructions ? 0 : 45, soloPer = 1, soloIn = 0;
  bool soloOpen = false;
  if (const char *e = getenv("MZR_KWT_SOLO_MIN")) soloMin = atoi(e);
  if (const char *e = getenv("MZR_KWT_SOLO_PER")) soloPer = std::max(1, std::min(3, atoi(e)));
  if (soloMin > 0 && !getenv("MZR_KWT_SOLO_MIN")) {
    int hist[66] = {0};
    for (const auto &rc : v) { const int nd = need(rc); if (nd > classBMax) hist[std::min(65, std::max(0, nd))]++; }
    const int most = std::max(64, h->swCap / 32);
    int cnt = 0, T = 66;
    while (T > soloMin && cnt + hist[T - 1] <= most) { --T; cnt += hist[T]; }
    soloMin = T > 65 ? 0 : T;
  }
  std::vector<int> offA(h->nStages + 1), offB(h->nStages + 1), offC(h->nStages + 1);      // (committed once the lists are on the device)
  for (int sg = 0; sg < h->nStages; ++sg) {
    offA[sg] = (int)L[0].size(); offB[sg] = (int)L[1].size(); offC[sg] = (int)L[2].size();
    key.clear();
    for (int i = h->kwtStageOff[sg]; i < h->kwtStageOff[sg + 1]; ++i) key.emplace_back(-need(v[i]), i);
    std::sort(key.begin(), key.end());
    for (const auto &k : key) {
      const int cl = -k.first <= classCMax ? 2 : -k.first <= classBMax ? 1 : 0;
      if (cl == 0 && soloMin > 0) {
        // latency path: the heaviest reaches share their pass with fewer others (soloPer per item of four lane groups; the other groups get
        // HOLES -- records whose stage is out of reach of any launch, so the group never has a step).  A pass costs what the entries in it
        // cost, and a reach's step t + 1 waits for its step t: the reaches with the longest lists set the window's longest chain.
        const bool solo = -k.first >= soloMin;
        const int inItem = (int)(L[0].size() & 3);
        if (inItem != 0 && (solo ? (!soloOpen || soloIn >= soloPer) : soloOpen)) {      // close the item that is open
          MzrKwtRec hole = v[k.second]; hole.sigma = MZR_KWT_HOLE;
          while (L[0].size() & 3) L[0].push_back(hole);
        }
        if ((L[0].size() & 3) == 0) { soloOpen = solo; soloIn = 0; }
        if (solo) ++soloIn;
      }
      L[cl].push_back(v[k.second]);
    }
    if (soloMin > 0 && soloOpen && (L[0].size() & 3)) {      // (the next stage starts an item of its own)
      MzrKwtRec hole = L[0].back(); hole.sigma = MZR_KWT_HOLE;
      while (L[0].size() & 3) L[0].push_back(hole);
      soloOpen = false;
    }
  }
  offA[h->nStages] = (int)L[0].size(); offB[h->nStages] = (int)L[1].size(); offC[h->nStages] = (int)L[2].size();
  DBuf<MzrKwtRec> *devStage[3] = {&h->kwtRouted, &h->kwtRoutedB, &h->kwtRoutedC}, *devAll[3] = {&h->kwtRoutedAll, &h->kwtRoutedBAll, &h->kwtRoutedCAll};
  // (holes: the 16-lane list may hold more records than there are routed reaches.  Nothing is in flight here; without room the reaches go on
  // in the classes they are in)
  {
    DBuf<MzrKwtRec> a[3], b[3];
    bool grow[3];
    for (int c = 0; c < 3; ++c) {
      grow[c] = L[c].size() > devStage[c]->n || L[c].size() > devAll[c]->n;
      if (!grow[c]) continue;
      try { a[c].alloc(L[c].size() + L[c].size() / 4); b[c].alloc(L[c].size() + L[c].size() / 4); } catch (const std::string &) { (void)hipGetLastError(); return; }
    }
    for (int c = 0; c < 3; ++c) if (grow[c]) { devStage[c]->swap(a[c]); devAll[c]->swap(b[c]); }
  }
  for (int c = 0; c < 3; ++c) {
    if (L[c].empty()) continue;
    (void)hipMemcpy(devStage[c]->p, L[c].data(), L[c].size() * sizeof(MzrKwtRec), hipMemcpyHostToDevice);
    // the same reaches, heaviest first regardless of stage
    std::vector<std::pair<int, int>> k2; k2.reserve(L[c].size());
    for (size_t i = 0; i < L[c].size(); ++i) k2.emplace_back(-need(L[c][i]), (int)i);
    std::sort(k2.begin(), k2.end());
    std::vector<MzrKwtRec> S; S.reserve(L[c].size());
    for (const auto &k : k2) S.push_back(L[c][k.second]);
    (void)hipMemcpy(devAll[c]->p, S.data(), S.size() * sizeof(MzrKwtRec), hipMemcpyHostToDevice);
  }
  h->kwtAllValid = true;
  h->kwtRoutedOff = offA; h->kwtBOff = offB; h->kwtCOff = offC;
  if (getenv("MZR_KWT_CLASS_LOG")) {      // debugging aid: how the routed reaches spread over the work-array need and the lane classes
    int hist[64] = {0};
    for (const auto &rc : v) hist[std::min(63, std::max(0, need(rc)))]++;
    size_t holes = 0, alone = 0;
    for (size_t i = 0; i < L[0].size(); ++i) { holes += L[0][i].sigma >= MZR_KWT_HOLE; if (soloMin > 0 && L[0][i].sigma < MZR_KWT_HOLE && need(L[0][i]) >= soloMin) ++alone; }
    fprintf(stderr, "[mzr] kwt classes: A %zu (+ %zu holes, %zu reaches of >= %d entries in passes of %d) B %zu C %zu reaches (cuts %d / %d); need histogram:", L[0].size() - holes, holes, alone, soloMin, soloPer, L[1].size(), L[2].size(), classBMax, classCMax);
    for (int i = 0; i < 64; ++i) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]);
    fprintf(stderr, "\n");
  }
  h->h_swA = L[0]; h->h_swB = L[1]; h->h_swC = L[2];
  // MZR_KWT_HEAVY_FIRST=1: the sweep draws its items heaviest first regardless of stage (the class lists in order of weight: the
  // *All arrays) -- the longest passes of a launch of the schedule start first
  h->swHeavyFirst = getenv("MZR_KWT_HEAVY_FIRST") && atoi(getenv("MZR_KWT_HEAVY_FIRST")) != 0;
  if (h->swHeavyFirst) for (int c = 0; c < 3; ++c) {
    std::vector<std::pair<int, int>> k2; k2.reserve(L[c].size());
    for (size_t i = 0; i < L[c].size(); ++i) k2.emplace_back(-need(L[c][i]), (int)i);
    std::sort(k2.begin(), k2.end());
    std::vector<MzrKwtRec> S; S.reserve(L[c].size());
    for (const auto &k : k2) S.push_back(L[c][k.second]);
    (c == 0 ? h->h_swA : c == 1 ? h->h_swB : h->h_swC) = S;
  }
  kwt_build_sweep(h);
}

// Steps a reach takes per launch of the Eulerian stage kernels.  More than one keeps a reach's state in the caches from one step to
// the next and divides the launches of a long window; it also makes the schedule's fill and drain (S - 1 launches each, part of
// the stages idle) KB times longer, which overlapping windows hide and windows that cannot overlap (partitioned domains) pay.
// MZR_STEP_BLOCK forces it (tests run several); one step per launch wherever KWT's launches share the loop, and in short windows.
// Measured (profiles/r04_experiments.md, reach-steps/s with 1 / 4 steps per launch, windows overlapping): 100 k reaches, windows of
// 4096: IRF 6.2 / 7.35, KW 4.6 / 5.65, DW 4.5 / 5.5 x 10^9; 625 k reaches, windows of 2048: IRF 6.5 / 6.7, DW 5.1 / 5.6; Muskingum-
// Cunge loses (2.15 / 2.08, 4.0 / 2.8 x 10^9: a launch waits for its slowest reach, now through four steps of sub-steps).  So:
// Windows of 16 384 at 100 k reaches, 1 / 4 / 8 / 16 steps: IRF 6.5 / 8.0 / 8.3 / 8.5, KW 4.9 / 6.1 / 6.45 / 6.4, DW 4.7 / 5.9 / 6.2 / 6.2;
// at 625 k DW is best with 2-4 (5.6; 4.7 with 8).  So: where the windows can overlap with it (at least as many blocks as stages)
// and no method is Muskingum-Cunge, the largest of 8 (4 from 300 k reaches on), 4, 2 that leaves as many blocks as stages; else one.
static int stepBlockFor(mzr_handle h, int W, bool canOverlap) {
  int kb = 1;
  // (the blocked kernels are instantiations of their own and need more registers -- the loop costs them even when it runs once: DW 208 ->
  // 262 VGPRs, with the lake / water-management / observation branches 217 -> 358: those configurations keep one step per launch)
  const bool full = h->nLake || h->cfg.is_flux_wm || h->qmod || h->tracer;
  if (canOverlap && !full && idxOf(h, MZR_MC) < 0)
    for (int k = (h->N >= 300000 ? std::min(4, MZR_STEP_BLOCK_DEFAULT) : MZR_STEP_BLOCK_DEFAULT); k >= 2; k /= 2)
      if (W / k >= h->nStages) { kb = k; break; }
  if (const char *e = getenv("MZR_STEP_BLOCK")) kb = atoi(e);
  if (idxOf(h, MZR_KWT) >= 0 || W <= 8) kb = 1;
  return std::max(1, std::min(kb, W));
}

static mzr_domain::Book saveBook(mzr_handle h) {
  mzr_domain::Book b;
  b.basCur = h->basCur; b.lastW = h->lastW; b.stepsDone = h->stepsDone; b.totalSteps = h->totalSteps; b.histSteps = h->histSteps;
  b.kwtWindows = h->kwtWindows; b.kwtStepsSince = h->kwtStepsSince; b.kwtHeadSteps = h->kwtHeadSteps;
  for (int ix = 0; ix < h->cfg.nRoutes && ix < 6; ++ix) { b.reachSteps[ix] = h->route[ix].reachSteps; b.meanSteps[ix] = h->route[ix].meanSteps; b.nLaunches[ix] = h->route[ix].nLaunches; }
  return b;
}
static void restoreBook(mzr_handle h, const mzr_domain::Book &b) {
  h->basCur = b.basCur; h->lastW = b.lastW; h->stepsDone = b.stepsDone; h->totalSteps = b.totalSteps; h->histSteps = b.histSteps;
  h->kwtWindows = b.kwtWindows; h->kwtStepsSince = b.kwtStepsSince; h->kwtHeadSteps = b.kwtHeadSteps;
  for (int ix = 0; ix < h->cfg.nRoutes && ix < 6; ++ix) { h->route[ix].reachSteps = b.reachSteps[ix]; h->route[ix].meanSteps = b.meanSteps[ix]; h->route[ix].nLaunches = b.nLaunches[ix]; }
}

static int run_window(mzr_handle h, int W, double t_start, double T1_single, const double *runoff_dev) {
  if (!h->haveState) return fail(h, 20, "mzr_run/state not initialised (call mzr_init_state)");
  if (W < 1 || W > h->cfg.maxWindow) return fail(h, 20, "mzr_run/nSteps exceeds maxWindow");
  if (h->cfg.is_flux_wm && h->wmSteps < W) return fail(h, 20, "mzr_run/is_flux_wm is on: call mzr_set_wm_flux for this window first");
  if (h->qmod && h->obsSteps < W) return fail(h, 20, "mzr_run/direct insertion is on: call mzr_set_obs for this window first");
  if (h->tracer && h->solSteps < W) return fail(h, 20, "mzr_run/constituent routing is on: call mzr_set_solute for this window first");
  if (h->nLake && h->lakeSteps < W) return fail(h, 20, "mzr_run/lakes are on: call mzr_set_lake_forcing for this window first");
  if (h->anyLakeTarget && h->wmVolSteps < W) return fail(h, 20, "mzr_run/target-volume lakes are on: call mzr_set_wm_vol for this window first");
  (void)hipSetDevice(h->cfg.device);
  const mzr_domain::Book bookBefore = saveBook(h);
  if (h->kwN.p) {   // regroup after the first two windows (not before the first: no particles yet), then every 8 windows / 512 steps
    const bool early = W > 1 && (h->kwtWindows == 1 || h->kwtWindows == 2);
    if (h->kwtWindows > 0 && (early || h->kwtStepsSince >= std::max(512LL, 8LL * W))) { kwt_regroup(h); h->kwtStepsSince = 0; }
    ++h->kwtWindows; h->kwtStepsSince += W;
  }
  for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {      // Muskingum-Cunge: the slow reaches are listed after the first two windows, then every 8 windows
    RouteBufs &rb = h->route[ix];
    if (rb.method != MZR_MC || W < 2) continue;
    const long long k = rb.mcWindows++;
    if (k == 1 || k == 2 || (k > 2 && k % 8 == 0)) mc_regroup(h, ix);
  }
  const int N = h->N;
  hipStream_t st = h->stream;
  // The channel table (kernels_route.hip d_chan): built and bit-identical, but OFF unless MZR_CHAN_TABLE=1 -- measured on the c4 shard
  // (625 k reaches, IRF + Muskingum-Cunge): k_stage<4> 92.8 us per launch with the table against 86.8 us without.  The launch lasts as
  // long as its reach with the most sub-steps (a chain of dependent FP64 operations), the other wavefronts wait for memory, and ten
  // more loads per lane cost them more than the ~300 instructions they save (profiles/r06_experiments.md).
  const bool chanTableOn = getenv("MZR_CHAN_TABLE") && atoi(getenv("MZR_CHAN_TABLE")) != 0;
  if (chanTableOn && h->chanDirty && (idxOf(h, MZR_MC) >= 0 || idxOf(h, MZR_KW) >= 0 || idxOf(h, MZR_DW) >= 0)) {
    // once per parameter set, on the handle's stream in front of everything the window queues
    if (h->tail.pending) flushTail(h);
    try {
      if (!h->chanTab.p) h->chanTab.alloc((size_t)mzr_chan_table_doubles() * N);
      MzrDev dc; fillDev(h, dc);
      mzr_launch_chan_table(dc, h->chanTab.p, st);
      (void)hipStreamSynchronize(st);      // (the methods' own streams read it)
      h->chanDirty = false;
    } catch (const std::string &) { (void)hipGetLastError(); }      // no room: the solvers compute the values per reach-step as before
  }
  if (h->imNextInAlt) {        // this window's halo discharge was imported beside that of a window kept back
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) h->route[ix].imQ.swap(h->route[ix].imQAlt);
    h->imNextInAlt = false;
  }
  if (h->lakeNextInAlt) {      // this window's lake forcing was written beside the forcing of a window kept back
    h->lakeEvap.swap(h->lakeEvapAlt); h->lakePrecip.swap(h->lakePrecipAlt);
    h->calMonth.swap(h->calMonthAlt); h->calDay.swap(h->calDayAlt); h->calDoy.swap(h->calDoyAlt);
    h->lakeNextInAlt = false;
  }
  // Overlapping windows (kernels_route.hip, k_stage_pair): the launches s >= W of this window are kept back and go out with the
  // first launches of the next one.  Only where a window is nothing but stage launches of the Eulerian methods over rows
  // of its own: no KWT, constituent, gauge observations, water-management fluxes or imported halo rows, and at least as many
  // steps as the network has stages (so that never more than two windows are in flight).
  const int nSt = h->nStages;
  // Steps per launch of the Eulerian stage kernels (kernels_route.hip, stage_reach_block): see stepBlockFor
  // (round 6: halo rows no longer rule it out -- the imported discharge exists twice, mzr_import_boundary_dev)
  bool pipe = nSt >= 2 && idxOf(h, MZR_KWT) < 0 && !h->tracer && !h->qmod && !h->cfg.is_flux_wm &&
              !h->anyLakeTarget && !(W <= 8 && h->rtItems > 0);
  if (h->nHalo && getenv("MZR_OVERLAP_HALO") && atoi(getenv("MZR_OVERLAP_HALO")) == 0) pipe = false;
  if (const char *e = getenv("MZR_OVERLAP_WINDOWS")) pipe = pipe && atoi(e) != 0;
  if (const char *e = getenv("MZR_ROUTE_SWEEP")) pipe = pipe && atoi(e) == 0;      // (a forced persistent sweep routes whole windows)
  const int KB = stepBlockFor(h, W, pipe && h->nExp == 0);      // (an export asks for the window's last launches at once: no overlap)
  const int WB = (W + KB - 1) / KB;      // the window in blocks = launches in which a stage is active
  // Round 6: a window SHORTER than the network is deep overlaps too.  Its nStages - 1 kept-back launches no longer fit the next
  // window's WB launches one for one: the first (nStages - 1 - WB) of them go out on their own in front of the next window, the last WB
  // ride with its launches -- kept-back launch j + o with launch j, o = nStages - 1 - WB: the stages above j + o and the stages up to j,
  // disjoint reaches; step 0 of a reach of stage s in the new window (launch s) needs its last step of the old one, kept-back launch
  // s - 1, which went out before (on its own, or with launch s - 1 - o).  A window then costs nStages - 1 launches instead of
  // nStages + WB - 1 (a mainstem domain of 3 756 stages in windows of 2 048: 3 755 instead of 5 803), and still no more than two
  // windows are in flight.  MZR_OVERLAP_SHORT=0: only windows of at least nStages launches overlap (rounds 4-5).
  if (WB < nSt && getenv("MZR_OVERLAP_SHORT") && atoi(getenv("MZR_OVERLAP_SHORT")) == 0) pipe = false;
  if (h->tail.pending && !pipe) flushTail(h);
  const double *prevQlat = h->qlat.p;      // rows of the window before (row lastW = its last BASIN_QR(1))
  if (pipe) {
    try {
      if (!h->qlatAlt.p) { h->qlatAlt.alloc(h->qlat.n); h->qlatAlt.zero(st); }
      if (h->qi.p && !h->qiAlt.p) { h->qiAlt.alloc(h->qi.n); h->qiAlt.zero(st); }
      for (int ix = 0; ix < h->cfg.nRoutes; ++ix) if (!h->route[ix].Qalt.p) { h->route[ix].Qalt.alloc(h->route[ix].Q.n); h->route[ix].Qalt.zero(st); }
    } catch (const std::string &e) {      // no room for the second set of rows: windows one after the other, as before
      pipe = false; (void)hipGetLastError();
      h->qlatAlt.free(); h->qiAlt.free(); for (int ix = 0; ix < h->cfg.nRoutes; ++ix) h->route[ix].Qalt.free();
      if (h->tail.pending) flushTail(h);
    }
  }
  if (h->exportOnAux) {      // a record of the window before the last one is being packed from the rows this window is about to write
    if (h->exportPrevDone) (void)hipStreamWaitEvent(st, h->exportPrevDone, 0);
    h->exportOnAux = false;
  }
  h->prevInAlt = false;
  if (pipe) {      // this window's rows: the second set (the set of the window before stays as it is until its last launches are out)
    h->qlat.swap(h->qlatAlt); if (h->qi.p) h->qi.swap(h->qiAlt);
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) h->route[ix].Q.swap(h->route[ix].Qalt);
    h->prevInAlt = h->lastW > 0 && h->nExp > 0; h->prevW = h->lastW;
  }
  MzrDev d; fillDev(h, d);
  d.W = W; d.t_start = t_start; d.T1_single = T1_single; d.runoff = runoff_dev; d.stepBlock = KB;
  // carry BASIN_QR(1) of the last step of the previous window into row 0 (halo columns already
  // hold the imported row 0 of this window)
  if (h->lastW > 0) {
    if (prevQlat != h->qlat.p) hipLaunchKernelGGL(k_carry_qlat2, dim3((N + 255) / 256), dim3(256), 0, st, h->qlat.p, prevQlat, h->lastW, N, d.haloSlot, h->err.p);
    else hipLaunchKernelGGL(k_carry_qlat, dim3((N + 255) / 256), dim3(256), 0, st, h->qlat.p, h->lastW, N, d.haloSlot, h->err.p);
  }
  // Which methods go through a persistent sweep (decided here because it decides their stream).  KWT: always (one launch
  // per chunk of the skewed schedule, progress words instead of kernel boundaries; single steps too: 842 dependent stages
  // cost 15 ms per step as hand-offs, 18 ms as launches); MZR_KWT_SWEEP=0 keeps one launch per stage (k_stage_kwt), the
  // form the sweep is tested against.  The Eulerian methods (k_sweep_route): where launches are the cost -- single steps
  // and short windows (mzr_step: 42 ms instead of 49 ms per step of the 625 k-reach IRF + Muskingum-Cunge shard) -- and not
  // in long windows, where these kernels stream their per-reach state at 1.4-1.7 TB/s either way and plain cached accesses
  // from full-width launches move more bytes than sc1 accesses from resident wavefronts (6.1 against 4.9 x 10^9
  // reach-steps/s on the same shard).  Default: windows of up to 8 steps; MZR_ROUTE_SWEEP=1 / 0 forces it (tests run both).
  bool sweep = true;
  if (const char *e = getenv("MZR_KWT_SWEEP")) sweep = atoi(e) != 0;
  const int kwtIx = idxOf(h, MZR_KWT);
  if (kwtIx < 0 || h->swItems < 1 || W > 65535) sweep = false;      // (progress words hold the step count in 16 bits)
  bool rtSweep = W <= 8;
  if (const char *e = getenv("MZR_ROUTE_SWEEP")) rtSweep = atoi(e) != 0;
  if (h->rtItems < 1) rtSweep = false;
  if (rtSweep)
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      if (rb.method == MZR_KWT || rb.rtCapTried) continue;
      rb.rtCapTried = true;
      MzrDev dc; memset(&dc, 0, sizeof dc); dc.rtHead = rb.rtHead.p; dc.err = h->err.p;
      rb.rtCap = sweepGrid(h, mzr_sweep_route_capacity(rb.method, dc, h->stream));
      if (rb.rtCap < 1) fprintf(stderr, "mzr: the wavefront capacity of the sweep of method %d could not be measured on device %d; one launch per stage instead\n", rb.method, h->cfg.device);
    }
  bool anyPersistent = sweep;
  for (int ix = 0; ix < h->cfg.nRoutes; ++ix) if (h->route[ix].method != MZR_KWT && rtSweep && h->route[ix].rtCap >= 1) anyPersistent = true;
  // hillslope pre-pass.  The fold is causal and launch s of the sweep only touches steps <= s, so only the
  // first chunk has to be ready before the sweep starts; the others are produced on a second stream
  // while the sweep runs and the sweep waits for chunk c right before launch s = c * chunk.
  const int CH = 1024;
  const int nChunks = (W + CH - 1) / CH;
  // several routing methods are independent of each other once the hillslope series exist: each gets its own
  // stream (their stage launches are small and latency-bound, so they fill each other's gaps)
  const bool multi = h->cfg.nRoutes > 1;
  // ... unless the window goes through a persistent sweep: that is ONE launch for the whole window, made when nothing else
  // is being dispatched -- a sweep some of whose workgroups start behind time (because another kernel holds their slots
  // for a while) was measured to freeze the memory instructions of exactly those wavefronts now and then, for as long as
  // the others keep running (ierr 93; profiles/r03_soak.md).  The whole hillslope pre-pass then runs first.
  bool chunked = nChunks > 2 && !multi && !anyPersistent;
  if (const char *e = getenv("MZR_BASIN_CHUNKED")) chunked = chunked && atoi(e) != 0;   // debugging aid
  // (persistent KWT sweep with the hillslope delay on: k_hillslope_out also writes the headwater reaches' discharge rows of the KWT
  // method -- the value it has just made -- and k_kwt_window_init no longer reads them back and copies them)
  if (!chunked && sweep && h->cfg.doesBasinRoute == 1 && h->kwHeadFlag.p && !h->h_kwtHead.empty() && !getenv("MZR_NO_HEAD_FUSE")) {
    d.kwHeadFlag = h->kwHeadFlag.p; d.kwHeadQ = h->route[kwtIx].Q.p;
  }
  if (!chunked) mzr_launch_basin(d, st);
  else {
    if (!h->basinStream) (void)hipStreamCreateWithFlags(&h->basinStream, hipStreamNonBlocking);
    while ((int)h->basinEvents.size() < nChunks + 2) { hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); h->basinEvents.push_back(e); }
    mzr_launch_basin_chunk(d, 0, CH, st);
    (void)hipEventRecord(h->basinEvents[0], st);                       // everything before this window is done
    (void)hipStreamWaitEvent(h->basinStream, h->basinEvents[0], 0);
    for (int c = 1; c < nChunks; ++c) {
      mzr_launch_basin_chunk(d, c * CH, std::min(W, (c + 1) * CH), h->basinStream);
      (void)hipEventRecord(h->basinEvents[c], h->basinStream);
    }
    mzr_launch_basin_state(d, h->basinStream);
    (void)hipEventRecord(h->basinEvents[nChunks], h->basinStream);
  }
  if (h->cfg.doesBasinRoute == 1) h->basCur ^= 1;
  const int nS = h->nStages;
  const int nR = h->cfg.nRoutes;
  hipStream_t rst[6];
  MzrDev dr[6];
  for (int ix = 0; ix < nR; ++ix) {
    rst[ix] = st;
    // persistent sweeps of several methods run one after the other on the handle's stream: each is sized to fill the
    // device, and a sweep whose grid does not fit beside another one can stall
    const bool persistent = h->route[ix].method == MZR_KWT ? sweep : (rtSweep && h->route[ix].rtCap >= 1);
    if (multi && ix > 0 && !persistent) {
      if (!h->routeStream[ix]) {      // (the other methods of a mainstem domain go to high-priority streams like its first: mzr_set_boundary)
        int lo = 0, hi = 0;
        // (experiment, MZR_MC_STREAM_PRIO=1: Muskingum-Cunge -- a launch as long as its longest sub-step chain -- on a high-priority stream
        // beside the bandwidth-bound method of the first stream)
        const bool mcPrio = h->route[ix].method == MZR_MC && getenv("MZR_MC_STREAM_PRIO") && atoi(getenv("MZR_MC_STREAM_PRIO")) != 0;
        if (!((h->highPriority || mcPrio) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo &&
              hipStreamCreateWithPriority(&h->routeStream[ix], hipStreamNonBlocking, hi) == hipSuccess))
          (void)hipStreamCreateWithFlags(&h->routeStream[ix], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&h->routeEvent[ix], hipEventDisableTiming);
      }
      rst[ix] = h->routeStream[ix];
    }
    dr[ix] = d;
    setRoute(h, dr[ix], ix);
    // (event pairs of windows queued without a synchronisation in between pile up; mzr_sync reads and releases them)
  }
  if (multi) {
    if (!h->routeEvent[0]) (void)hipEventCreateWithFlags(&h->routeEvent[0], hipEventDisableTiming);
    (void)hipEventRecord(h->routeEvent[0], st);                    // hillslope series of the window (and everything before) done
    for (int ix = 1; ix < nR; ++ix) if (rst[ix] != st) (void)hipStreamWaitEvent(rst[ix], h->routeEvent[0], 0);
  }
  const bool prof = h->profiling;
  const int kc = h->swKcWide ? 1 : 0;      // the sweep flavour with four particle slots per lane of the 4-lane class (kwt_regroup)
  if (sweep) {
    kwt_sweep_tables(h, W);
    RouteBufs &rb = h->route[kwtIx];
    hipStream_t sx = rst[kwtIx];
    MzrDev dk = dr[kwtIx];
    dk.swRA = h->swRA.p; dk.swP = h->swP.p;
    dk.kwtLight = h->kwtDepLight.p;
    if (h->swHeavyFirst && h->kwtAllValid) { dk.kwtRouted = h->kwtRoutedAll.p; dk.kwtRoutedB = h->kwtRoutedBAll.p; dk.kwtRoutedC = h->kwtRoutedCAll.p; }
    const int nLaunch = nS + W - 1;
    {      // the state the queue can be taken back to, and what it takes to route this window again (mzr_sync, retryKwtQueue)
      // (not for a domain that exports a boundary record -- the record of a stalled window may have been packed and sent before
      // mzr_sync gets to route the window again -- and not when another method went through a persistent sweep in this window:
      // k_sweep_route returns at once after an error, so that method would be left behind with nobody to say so)
      bool otherSweep = false;
      for (int ix = 0; ix < h->cfg.nRoutes; ++ix) if (h->route[ix].method != MZR_KWT && rtSweep && h->route[ix].rtCap >= 1) otherSweep = true;
      const bool can = !h->nLake && !h->tracer && !h->cfg.is_flux_wm && h->nExp == 0 && !otherSweep;
      mzr_domain::QWin qw;
      qw.W = W; qw.t_start = t_start; qw.T1_single = T1_single; qw.runoff = runoff_dev; qw.replayable = h->nextReplayable; qw.before = bookBefore;
      if (can) {
        try {
          if (!h->snapN.p) { h->snapN.alloc(h->kwN.n); h->snapQ.alloc(h->kwQ.n); h->snapTR.alloc(h->kwTR.n); h->snapQsum.alloc(N); }
          if (rb.hInflow.p && !h->snapHIn.p) h->snapHIn.alloc(N);      // (mzr_set_history may have switched the sum on since the first snapshot)
          hipLaunchKernelGGL(k_snapshot, dim3(1024), dim3(256), 0, sx, h->err.p, h->snapN.p, h->kwN.p, h->kwN.n, h->snapQ.p, h->kwQ.p, h->kwQ.n,
                             h->snapTR.p, h->kwTR.p, h->kwTR.n, h->snapQsum.p, rb.qsum.p, rb.hInflow.p ? h->snapHIn.p : (double *)nullptr, rb.hInflow.p, (size_t)N);
          qw.snap = true;
        } catch (const std::string &) { (void)hipGetLastError(); }
      }
      h->retry.q.push_back(qw);
      // debugging aid (tests): MZR_SWEEP_FAIL_AT=n gives the n-th KWT window of the handle (0-based) a watchdog of one clock tick
      // The watchdog of the KWT sweep measures time WITHOUT PROGRESS on the words a wavefront polls; how long that may be before the
      // window is given up follows from what the window is expected to take (round 6: it was 8 s whatever the window -- a stall cost
      // 8 s plus the window once more): four times the window's expected duration -- W steps x 15 us x the passes a wavefront takes
      // per launch of the schedule, the figure both operating points show (100 k reaches: 6 022 items for 4 008 wavefronts, 27 us per
      // step; 375 k: 22 416 items, 80 us) -- but at least 1 s and at most 8 s.  mzr_config.sweepTimeout > 0 fixes it.
      if (!(h->cfg.sweepTimeout > 0.0)) {
        const double perStep = 15.e-6 * std::max(1.8, (double)h->swItems / (double)std::max(1, h->swWaves));
        dk.stallTicks = (long long)(std::min(8.0, std::max(1.0, 4.0 * perStep * (double)W)) * 1.e8);
      }
      if (h->retry.failAt == -1) { const char *e = getenv("MZR_SWEEP_FAIL_AT"); h->retry.failAt = e ? atoi(e) : -2; }
      if (h->retry.failAt >= 0 && h->retry.seen == h->retry.failAt && h->retry.depth == 0) dk.stallTicks = 1;
      if (h->retry.depth == 0) ++h->retry.seen;
      dk.winSeq = (int)h->retry.q.size() - 1;
    }
    if (h->swClock.p) { dk.swClock = h->swClock.p + 2 * (size_t)(h->swClockN % MZR_CLOCK_LOG); ++h->swClockN; }
    mzr_launch_kwt_window_init(dk, 0, W, sx);
    // (mzr_run_async*: the copy of the next window starts behind this point plus a pause, not beside the launch itself; only where
    // that entry point is in use -- the event is one more packet in front of k_sweep_heads, two launches in front of the sweep)
    if (h->copyStream && getenv("MZR_H2D_HEAD_START") && atoi(getenv("MZR_H2D_HEAD_START")) != 0) {
      if (!h->sweepGo) (void)hipEventCreateWithFlags(&h->sweepGo, hipEventDisableTiming);
      (void)hipEventRecord(h->sweepGo, sx);
    }
    if (prof) {      // the event pair rides on the sweep's own dispatch (see mzr_launch_sweep_kwt)
      if (rb.evUsed == rb.events.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b); }
      // Timing-enabled events on the sweep's own stream -- as markers around the launch or attached to its dispatch -- slow some
      // windows by up to a quarter (profiles/r04_experiments.md).  So the timing events go to a stream of their own, chained to the
      // sweep's stream by events without timing: start = everything in front of the sweep has finished, stop = the sweep has.
      static const int how = getenv("MZR_EVENT_MARKERS") ? atoi(getenv("MZR_EVENT_MARKERS")) : 0;      // debugging aid: 1 markers, 2 attached
      if (how == 1) {
        (void)hipEventRecord(rb.events[rb.evUsed].first, sx);
        mzr_launch_sweep_kwt(dk, h->swWaves, 0, nLaunch, sx, nullptr, nullptr, kc);
        (void)hipEventRecord(rb.events[rb.evUsed].second, sx);
      } else if (how == 2) mzr_launch_sweep_kwt(dk, h->swWaves, 0, nLaunch, sx, rb.events[rb.evUsed].first, rb.events[rb.evUsed].second, kc);
      else {
        if (!h->timerStream) {
          (void)hipStreamCreateWithFlags(&h->timerStream, hipStreamNonBlocking);
          (void)hipEventCreateWithFlags(&h->timerGate[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&h->timerGate[1], hipEventDisableTiming);
        }
        (void)hipEventRecord(h->timerGate[0], sx); (void)hipStreamWaitEvent(h->timerStream, h->timerGate[0], 0);
        (void)hipEventRecord(rb.events[rb.evUsed].first, h->timerStream);
        mzr_launch_sweep_kwt(dk, h->swWaves, 0, nLaunch, sx, nullptr, nullptr, kc);
        (void)hipEventRecord(h->timerGate[1], sx); (void)hipStreamWaitEvent(h->timerStream, h->timerGate[1], 0);
        (void)hipEventRecord(rb.events[rb.evUsed].second, h->timerStream);
      }
      ++rb.evUsed;
    } else mzr_launch_sweep_kwt(dk, h->swWaves, 0, nLaunch, sx, nullptr, nullptr, kc);
    ++rb.nLaunches;
    if (h->countTraffic) h->kwtHeadSteps += (long long)h->h_kwtHead.size() * W;
  }
  bool anyStage = false;
  for (int ix = 0; ix < nR; ++ix) {
    RouteBufs &rb = h->route[ix];
    if (rb.method == MZR_KWT) { if (!sweep) anyStage = true; continue; }
    if (!rtSweep || rb.rtCap < 1) { anyStage = true; continue; }
    rt_sweep_tables(h, W);
    hipStream_t sx = rst[ix];
    MzrDev dx = dr[ix];
    dx.rtRA = h->rtRA.p; dx.rtP = h->rtP.p;
    const int nLaunch = nS + W - 1;
    const int waves = std::max(8, std::min(rb.rtCap, h->rtMaxAct));
    (void)hipMemsetAsync(rb.rtDone.p, 0, (size_t)N * sizeof(int), sx);
    if (prof) {
      if (rb.evUsed == rb.events.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b); }
      (void)hipEventRecord(rb.events[rb.evUsed].first, sx);
    }
    mzr_launch_sweep_route(rb.method, dx, waves, 0, nLaunch, sx);
    if (prof) { (void)hipEventRecord(rb.events[rb.evUsed].second, sx); ++rb.evUsed; }
    ++rb.nLaunches;
  }
  const bool withTail = pipe && h->tail.pending;      // the window before drains in this window's first launches
  int nPair = 0, tailOff = 0;      // kept-back launches tailOff .. tailOff + nPair - 1 of the window before ride with launches 0 .. nPair - 1 of this one
  if (withTail) {
    const int left = (nS - 1) - h->tail.next;
    nPair = std::min(left, WB);
    issueTail(h, h->tail.next, h->tail.next + (left - nPair));      // (a window shorter than the network is deep: the others first, on their own)
    tailOff = h->tail.next + (left - nPair);
    h->tail.next = nS - 1;
  }
  int nextChunk = 1;
  for (int s = 0; s < (pipe ? WB : nS + WB - 1); ++s) {
    if (!anyStage) break;
    // (launch s reads the hillslope series up to step KB (s + 1) - 1)
    while (chunked && nextChunk < nChunks && (long long)nextChunk * CH <= (long long)KB * (s + 1) - 1) { (void)hipStreamWaitEvent(st, h->basinEvents[nextChunk], 0); ++nextChunk; }
    const int sLo = std::max(0, s - (WB - 1)), sHi = std::min(s, nS - 1);
    const int rB = h->stageStart[sLo], rE = h->stageStart[sHi + 1];
    if (rE <= rB) continue;
    for (int ix = 0; ix < nR; ++ix) {
      RouteBufs &rb = h->route[ix];
      hipStream_t sx = rst[ix];
      if (sweep && ix == kwtIx) continue;
      if (rb.method != MZR_KWT && rtSweep && rb.rtCap >= 1) continue;
      if (withTail && s < nPair) {      // launch W' + tailOff + s of the window before (the stages above tailOff + s) and launch s of this one (stages 0 .. s) as one
        if (prof) {
          if (rb.evUsed == rb.events.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b); }
          (void)hipEventRecord(rb.events[rb.evUsed].first, sx);
        }
        mzr_launch_stage_pair(rb.method, h->tail.d[ix], h->tail.W + tailOff + s, h->stageStart[tailOff + s + 1], N, dr[ix], s, rB, rE, sx);
        if (prof) { (void)hipEventRecord(rb.events[rb.evUsed].second, sx); ++rb.evUsed; }
        ++rb.nLaunches; ++h->pairLaunches;
        continue;
      }
      if (prof) {
        if (rb.evUsed == rb.events.size()) {
          hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b);
        }
        (void)hipEventRecord(rb.events[rb.evUsed].first, sx);
      }
      if (rb.method == MZR_KWT) {
        MzrDev dk = dr[ix];
        // every stage active: heaviest reaches first regardless of stage (shorter tail of the launch).  This gives up
        // the stage-major locality of the rows, so only while the rows of all routed reaches (about 1 KB each) sit in
        // the 256 MB Infinity Cache anyway; measured +12 % at 100 k reaches, -8 % at 400 k.
        if (h->kwtAllValid && sLo == 0 && sHi == nS - 1 && h->h_kwtRouted.size() <= 150000) { dk.kwtRouted = h->kwtRoutedAll.p; dk.kwtRoutedB = h->kwtRoutedBAll.p; dk.kwtRoutedC = h->kwtRoutedCAll.p; }
        mzr_launch_stage_kwt(dk, s, h->kwtRoutedOff[sLo], h->kwtRoutedOff[sHi + 1], h->kwtBOff[sLo], h->kwtBOff[sHi + 1], h->kwtCOff[sLo], h->kwtCOff[sHi + 1],
                             h->kwtGenericOff[sLo], h->kwtGenericOff[sHi + 1], h->kwtLightOff[sLo], h->kwtLightOff[sHi + 1], sx);
      }
      else mzr_launch_stage(rb.method, dr[ix], s, rB, rE, sx);
      if (prof) { (void)hipEventRecord(rb.events[rb.evUsed].second, sx); ++rb.evUsed; }
      ++rb.nLaunches;
    }
    // the rows of the window before are complete from here on (its kept-back launches went out with launches 0 .. nS - 2 of this
    // one): what mzr_export_boundary_prev_dev waits for
    if (h->prevInAlt && s == (withTail ? std::max(0, nPair - 1) : 0))
      for (int ix = 0; ix < nR && ix < 6; ++ix) {
        if (!h->prevRowsEv[ix]) (void)hipEventCreateWithFlags(&h->prevRowsEv[ix], hipEventDisableTiming);
        (void)hipEventRecord(h->prevRowsEv[ix], rst[ix]);
      }
  }
  if (pipe && anyStage) { h->tail.pending = true; h->tail.W = WB; h->tail.next = 0; for (int ix = 0; ix < nR; ++ix) h->tail.d[ix] = dr[ix]; }
  else h->tail.pending = false;
  if (h->tracer) {
    // constituent: lateral mass flux and its hillslope delay for the whole window (after the water's), then, behind every
    // method's routing, the constituent pass over the same skewed schedule
    if (chunked) (void)hipStreamWaitEvent(st, h->basinEvents[nChunks], 0);
    if (h->lastW > 0 && h->basSol.p)
      hipLaunchKernelGGL(k_carry_qlat, dim3((N + 255) / 256), dim3(256), 0, st, h->basSol.p, h->lastW, N, (const int *)nullptr, h->err.p);
    MzrDev dt2 = d;
    dt2.solS0 = h->solS[h->solCur].p; dt2.solS1 = h->solS[h->solCur ^ 1].p;
    mzr_launch_basin_solute(dt2, st);
    if (h->cfg.doesBasinRoute == 1) h->solCur ^= 1;
    (void)hipEventRecord(h->trEvent, st);
    for (int ix = 0; ix < nR; ++ix) {
      RouteBufs &rb = h->route[ix];
      if (rb.method == MZR_SUM) continue;
      hipStream_t sx = rst[ix];
      if (sx != st) (void)hipStreamWaitEvent(sx, h->trEvent, 0);
      for (int s = 0; s < nS + W - 1; ++s) {
        const int sLo = std::max(0, s - (W - 1)), sHi = std::min(s, nS - 1);
        mzr_launch_tracer_stage(rb.method, dr[ix], s, h->stageStart[sLo], h->stageStart[sHi + 1], sx);
      }
    }
  }
  for (int ix = 0; ix < nR; ++ix) {
    if (h->route[ix].method == MZR_KWT) mzr_launch_accum_qsum(h->route[ix].Q.p, h->route[ix].qsum.p, N, W, rst[ix], h->err.p);
    h->route[ix].reachSteps += (long long)N * W;
    h->route[ix].meanSteps += W;
    if (multi && ix > 0 && rst[ix] != st) { (void)hipEventRecord(h->routeEvent[ix], rst[ix]); (void)hipStreamWaitEvent(st, h->routeEvent[ix], 0); }
  }
  if (chunked) (void)hipStreamWaitEvent(st, h->basinEvents[nChunks], 0);   // QFUTURE of the window is part of its result
  if (h->histFlags & MZR_H_RUNOFF) {      // histVars_data.f90:196-211: basin runoff, instantaneous and delayed runoff into the reaches, step by step
    const double *inst = (h->cfg.doesBasinRoute == 1 && h->qi.p) ? h->qi.p : h->qlat.p + N;
    mzr_launch_accum_qsum(inst, h->hInst.p, N, W, st, h->err.p);
    mzr_launch_accum_qsum(h->qlat.p + N, h->hDlay.p, N, W, st, h->err.p);
    mzr_launch_accum_qsum(runoff_dev, h->hBas.p, h->H, W, st, h->err.p);
    h->histSteps += W;
  }
  h->lastW = W; h->stepsDone += W; h->obsSteps = 0; h->solSteps = 0; h->wmSteps = 0; h->totalSteps += W; h->lakeSteps = 0; h->wmVolSteps = 0;
  if (hipGetLastError() != hipSuccess) return fail(h, 92, "mzr_run/kernel launch failed");
  return 0;
}

int mzr_set_wm_flux(mzr_handle h, int nSteps, const double *flux) {
  if (MZR_STAGES(h, nSteps) && h->cfg.is_flux_wm && flux) return stageRow(h, mzr_domain::SR_WMFLUX, flux, h->N, nullptr, 0, nullptr, nullptr, nullptr);
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_wm_flux/state not initialised") : 1;
  if (!h->cfg.is_flux_wm) return fail(h, 20, "mzr_set_wm_flux/is_flux_wm is off in the configuration");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_set_wm_flux/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  const int N = h->N;
  if (!ensureScratch(h)) return fail(h, 91, "mzr/out of device memory (row scratch)");
  (void)hipMemcpyAsync(h->scratchOut.p, flux, (size_t)nSteps * N * sizeof(double), hipMemcpyHostToDevice, h->stream);
  dim3 block(256), grid((N + 255) / 256, nSteps);
  hipLaunchKernelGGL(k_scatter_rows, grid, block, 0, h->stream, h->scratchOut.p, h->wm.p, h->d_ext2int.p, N, nSteps);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, 92, "mzr_set_wm_flux/device error");
  h->wmSteps = nSteps;
  return 0;
}

// A window of the queue ended with ierr 93 (its persistent KWT sweep gave up waiting) and the state it started from was kept
// (k_snapshot: the later windows of the queue left everything as it was -- every kernel of theirs returns at once after an error):
// back to that state, the window's KWT routing once more through one launch per stage (k_stage_kwt: no progress words, no waiting --
// the form the sweep is tested against, bit for bit; the hillslope series of the window (qlat) are untouched by the sweep and still
// there), and the windows queued behind it once more as they were queued.
static int retryKwtQueue(mzr_handle h, int k) {
  const int kwtIx = idxOf(h, MZR_KWT);
  const std::vector<mzr_domain::QWin> q = h->retry.q;      // (run_window below appends to the handle's own list)
  if (kwtIx < 0 || k < 0 || k >= (int)q.size() || !q[k].snap || !h->snapN.p) return 1;
  for (size_t j = k + 1; j < q.size(); ++j) if (!q[j].replayable) return 1;      // forcing that is no longer where it was
  if (q.size() > (size_t)k + 1 && (h->cfg.nRoutes != 1 || h->nHalo || (h->histFlags & MZR_H_RUNOFF))) return 1;      // only the plain KWT domain is taken back across windows
  // runoff history sums: the three accumulations of the failed window ran behind its sweep and returned at once (error word set);
  // they are made again below -- the basin-runoff sum from the window's forcing, which must still be where it was
  const bool histRunoff = (h->histFlags & MZR_H_RUNOFF) != 0;
  if (histRunoff && !(q[k].replayable && q[k].runoff)) return 1;
  RouteBufs &rb = h->route[kwtIx];
  const int N = h->N, W = q[k].W, nS = h->nStages;
  hipStream_t st = h->stream;
  (void)hipMemset(h->err.p, 0, sizeof(MzrErr));
  (void)hipMemcpyAsync(h->kwN.p, h->snapN.p, h->kwN.n * sizeof(int), hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(h->kwQ.p, h->snapQ.p, h->kwQ.n * sizeof(double), hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(h->kwTR.p, h->snapTR.p, h->kwTR.n * sizeof(double), hipMemcpyDeviceToDevice, st);
  (void)hipMemcpyAsync(rb.qsum.p, h->snapQsum.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st);
  if (rb.hInflow.p && h->snapHIn.p) (void)hipMemcpyAsync(rb.hInflow.p, h->snapHIn.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st);
  MzrDev d; fillDev(h, d);
  d.W = W; d.t_start = q[k].t_start; d.T1_single = q[k].T1_single; d.runoff = nullptr;
  setRoute(h, d, kwtIx);
  for (int s = 0; s < nS + W - 1; ++s) {
    const int sLo = std::max(0, s - (W - 1)), sHi = std::min(s, nS - 1);
    mzr_launch_stage_kwt(d, s, h->kwtRoutedOff[sLo], h->kwtRoutedOff[sHi + 1], h->kwtBOff[sLo], h->kwtBOff[sHi + 1], h->kwtCOff[sLo], h->kwtCOff[sHi + 1],
                         h->kwtGenericOff[sLo], h->kwtGenericOff[sHi + 1], h->kwtLightOff[sLo], h->kwtLightOff[sHi + 1], st);
  }
  mzr_launch_accum_qsum(rb.Q.p, rb.qsum.p, N, W, st, h->err.p);
  if (histRunoff) {      // as at the end of run_window (histSteps has counted the window already)
    const double *inst = (h->cfg.doesBasinRoute == 1 && h->qi.p) ? h->qi.p : h->qlat.p + N;
    mzr_launch_accum_qsum(inst, h->hInst.p, N, W, st, h->err.p);
    mzr_launch_accum_qsum(h->qlat.p + N, h->hDlay.p, N, W, st, h->err.p);
    mzr_launch_accum_qsum(q[k].runoff, h->hBas.p, h->H, W, st, h->err.p);
  }
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  ++h->sweepRetries;
  // the windows behind it: the bookkeeping as it was before the first of them was queued, then the same calls again
  h->retry.q.clear();
  if ((size_t)k + 1 < q.size()) {
    restoreBook(h, q[k + 1].before);
    ++h->retry.depth;
    for (size_t j = k + 1; j < q.size(); ++j) {
      h->nextReplayable = true;
      const int rc = run_window(h, q[j].W, q[j].t_start, q[j].T1_single, q[j].runoff);
      h->nextReplayable = false;
      if (rc) { --h->retry.depth; return 1; }
    }
    --h->retry.depth;
  }
  return 0;
}

int mzr_sync(mzr_handle h) {
  MZR_FLUSH(h);
  if (!h) return 1;
  (void)hipSetDevice(h->cfg.device);
  for (int round = 0; ; ++round) {
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, 92, std::string("mzr_sync/") + hipGetErrorString(e));
    if (h->retry.q.empty() || round >= 4) break;
    int code = 0;
    MzrErr e93;
    // (where 20 = a wait of the KWT sweep; a stall raised anywhere else -- the Eulerian sweeps raise 21 -- is not this queue's to repair)
    if (!(hipMemcpy(&code, h->err.p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && code == 93 &&
          hipMemcpy(&e93, h->err.p, sizeof e93, hipMemcpyDeviceToHost) == hipSuccess && e93.where == 20)) break;
    const int k = e93.winSeq, nq = (int)h->retry.q.size();
    fprintf(stderr, "mzr: the persistent KWT sweep of window %d of %d queued (%d steps) gave up waiting (ierr 93, reach index %d, schedule step %d); the window is routed "
                    "again with one launch per stage%s\n", k + 1, nq, (k >= 0 && k < nq) ? h->retry.q[k].W : 0, e93.reach, e93.s, k + 1 < nq ? ", the windows behind it as they were queued" : "");
    if (retryKwtQueue(h, k) != 0) return fail(h, 93, "mzr_sync/the persistent KWT sweep gave up waiting and the queued windows could not be routed again");
  }
  h->retry.q.clear();
  if (h->profiling) {
    if (h->timerStream) (void)hipStreamSynchronize(h->timerStream);
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      const bool logEach = getenv("MZR_LAUNCH_LOG") != nullptr;      // debugging aid: every event-timed launch on stderr
      for (size_t k = 0; k < rb.evUsed; ++k) {
        float ms = 0; (void)hipEventElapsedTime(&ms, rb.events[k].first, rb.events[k].second); rb.kernel_ms += ms;
        if (rb.kernel_ms_min == 0.0 || ms < rb.kernel_ms_min) rb.kernel_ms_min = ms;
        if (ms > rb.kernel_ms_max) rb.kernel_ms_max = ms;
        if (logEach) fprintf(stderr, "mzr launch method %d #%zu %.3f ms\n", rb.method, k, ms);
      }
      rb.evUsed = 0;
    }
  }
  return checkDeviceError(h);
}

int mzr_run_dev(mzr_handle h, int nSteps, double t_start, const double *runoff_dev) {
  MZR_FLUSH_STEPS(h);
  if (!h) return 1;
  h->nextReplayable = true;      // the forcing stays in the caller's device memory, unchanged until the next synchronisation (include/mzr.h)
  const int rc = run_window(h, nSteps, t_start, t_start + h->cfg.dt, runoff_dev);
  h->nextReplayable = false;
  return rc;
}

int mzr_run(mzr_handle h, int nSteps, double t_start, const double *runoff) {
  MZR_FLUSH(h);
  if (!h) return 1;
  if (!h->haveState) return fail(h, 20, "mzr_run/state not initialised (call mzr_init_state)");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_run/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  if (!ensureRunoffW(h)) return fail(h, 91, "mzr/out of device memory (forcing window)");
  (void)hipMemcpyAsync(h->runoffW.p, runoff, (size_t)nSteps * h->H * sizeof(double), hipMemcpyHostToDevice, h->stream);
  const int rc = run_window(h, nSteps, t_start, t_start + h->cfg.dt, h->runoffW.p);
  if (rc) return rc;
  return mzr_sync(h);
}

// The stand-alone driver's loop (standalone/route_runoff.f90:80-108) reads forcing and routes, step after step;
// here a whole window of forcing is handed over in host memory and the call returns at once: the copy runs on its
// own stream into one of two device buffers while the window before is still being routed.
// one wavefront that does nothing for `ticks` of the 100 MHz clock (mzr_run_async*: the head start of a sweep launch over the next window's copy)
__global__ void __launch_bounds__(64) k_pause(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void __launch_bounds__(256) k_widen_f32(const float *src, double *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

static int run_async_impl(mzr_handle h, int nSteps, double t_start, double T1_single, const double *runoff, hipEvent_t copied, const float *runoff32 = nullptr) {
  if (!h->haveState) return fail(h, 20, "mzr_run/state not initialised (call mzr_init_state)");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_run/nSteps exceeds maxWindow");
  (void)hipSetDevice(h->cfg.device);
  try {
    if (!h->runoffW2.p) h->runoffW2.allocStaging((size_t)h->cfg.maxWindow * h->H);
  } catch (const std::string &e) { return fail(h, 91, "mzr_run_async/" + e); }
  if (!h->copyStream) {
    if (hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking) != hipSuccess) return fail(h, 90, "mzr_run_async/hipStreamCreate failed");
    for (int i = 0; i < 2; ++i) { (void)hipEventCreateWithFlags(&h->rwCopied[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&h->rwRead[i], hipEventDisableTiming); }
  }
  const int k = h->rwCur;
  if (!ensureRunoffW(h)) return fail(h, 91, "mzr/out of device memory (forcing window)");
  double *buf = k == 0 ? h->runoffW.p : h->runoffW2.p;
  if (h->rwUsed[k]) (void)hipStreamWaitEvent(h->copyStream, h->rwRead[k], 0);     // the window that last read this buffer
  if (k == 0 && h->rwOtherSet) { (void)hipStreamWaitEvent(h->copyStream, h->rwOther, 0); h->rwOtherSet = false; }   // ... also one queued by mzr_run_src_dev
  // Round 5, measured and left OFF (MZR_H2D_HEAD_START=1 turns it on): the copy of window k + 1 starts when window k - 1 has finished, so it
  // is in full swing when the persistent sweep of window k is launched ~50 ms later, and a sweep launched beside a copy now and then gets
  // ~12 % of its wavefronts 20-80 us late (histogram of start delays, tools/r05_h2d2.py; without a copy every wavefront starts within
  // 0.7 us); those do not join (DESIGN.md 2.4 iii) and the window takes 480-520 ms instead of 442.  Holding the copy back until that
  // launch has been eligible for 300 us takes the late starts away (every sweep 444-446 ms) -- and the bench's host-forcing leg got
  // SLOWER with it, twice on one box (value_with_h2d / value 0.907, 0.886 against 0.920, 0.952: tools/r05_h2d3.sh).  Not understood.
  static const bool headStart = getenv("MZR_H2D_HEAD_START") && atoi(getenv("MZR_H2D_HEAD_START")) != 0;
  if (headStart && h->sweepGo) {
    (void)hipStreamWaitEvent(h->copyStream, h->sweepGo, 0);
    hipLaunchKernelGGL(k_pause, dim3(1), dim3(64), 0, h->copyStream, 30000LL);
  }
  if (runoff32) {      // single-precision forcing: half the bytes across PCIe, widened behind the copy on the copy's stream
    const size_t n = (size_t)nSteps * h->H;
    try { if (h->runoffF[k].n < (size_t)h->cfg.maxWindow * h->H) h->runoffF[k].allocStaging((size_t)h->cfg.maxWindow * h->H); } catch (const std::string &e) { return fail(h, 91, "mzr_run_async_f32/" + e); }
    if (hipMemcpyAsync(h->runoffF[k].p, runoff32, n * sizeof(float), hipMemcpyHostToDevice, h->copyStream) != hipSuccess)
      return fail(h, 92, "mzr_run_async_f32/hipMemcpyAsync failed");
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, h->copyStream, h->runoffF[k].p, buf, n);
  } else
  if (hipMemcpyAsync(buf, runoff, (size_t)nSteps * h->H * sizeof(double), hipMemcpyHostToDevice, h->copyStream) != hipSuccess)
    return fail(h, 92, "mzr_run_async/hipMemcpyAsync failed");
  (void)hipEventRecord(h->rwCopied[k], h->copyStream);
  if (copied) (void)hipEventRecord(copied, h->copyStream);      // (the caller's host buffer is free again)
  (void)hipStreamWaitEvent(h->stream, h->rwCopied[k], 0);
  const int rc = run_window(h, nSteps, t_start, T1_single, buf);
  if (rc) return rc;
  (void)hipEventRecord(h->rwRead[k], h->stream);
  h->rwUsed[k] = true; h->rwCur = k ^ 1;
  return 0;
}

int mzr_run_async(mzr_handle h, int nSteps, double t_start, const double *runoff) {
  MZR_FLUSH_STEPS(h);
  if (!h) return 1;
  return run_async_impl(h, nSteps, t_start, t_start + h->cfg.dt, runoff, nullptr);
}

int mzr_run_async_f32(mzr_handle h, int nSteps, double t_start, const float *runoff) {
  MZR_FLUSH_STEPS(h);
  if (!h) return 1;
  if (!runoff) return fail(h, 20, "mzr_run_async_f32/runoff is null");
  return run_async_impl(h, nSteps, t_start, t_start + h->cfg.dt, nullptr, nullptr, runoff);
}

int mzr_set_remap(mzr_handle h, int kind, int nMap, const int *hru_ix, const int *num_qhru, int nOverlap, const int *qhru_ix,
                  const int *i_index, const int *j_index, const double *weight, int n1, int n2,
                  const long long *qhru_id, const long long *src_id) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_remap/network not set") : 1;
  if (kind != 1 && kind != 2) return fail(h, 20, "mzr_set_remap/kind must be 1 (polygon vector) or 2 (grid)");
  if ((kind == 1 && !qhru_ix) || (kind == 2 && (!i_index || !j_index || n2 < 1)) || n1 < 1) return fail(h, 20, "mzr_set_remap/missing index arrays");
  (void)hipSetDevice(h->cfg.device);
  const int H = h->H, IMISS = -9999;   // integerMissing, public_var.f90:44
  std::vector<int> rowStart(H, 0), rowCnt(H, -1), srcIdx((size_t)std::max(nOverlap, 1), -1);
  long long cur = 0;   // the reference's running ixOverlap (0-based)
  for (int i = 0; i < nMap; ++i) {
    const int j = hru_ix[i];
    if (j == IMISS) { if (num_qhru[i] != IMISS) cur += num_qhru[i]; continue; }   // process_remap.f90:189-194
    if (j < 1 || j > H) return fail(h, 20, "mzr_set_remap/hru_ix out of range");
    const int n = num_qhru[i] > 0 ? num_qhru[i] : 0;
    if (cur + n > nOverlap) return fail(h, 20, "mzr_set_remap/num_qhru runs past the overlap arrays");
    rowStart[j - 1] = (int)cur; rowCnt[j - 1] = n;      // a later row that names the same HRU overwrites it
    for (int k = 0; k < n; ++k) {
      const size_t e = (size_t)cur + k;
      if (kind == 1) {
        const int q = qhru_ix[e];
        if (q == IMISS) { srcIdx[e] = -1; continue; }
        if (q < 1 || q > n1) return fail(h, 20, "mzr_set_remap/qhru_ix out of range");
        if (qhru_id && src_id && qhru_id[e] != src_id[q - 1])
          return fail(h, 20, "remap_runoff/remap_1D_runoff/mismatch in HRU ids for polygons in the runoff layer");
        srcIdx[e] = q - 1;
      } else {
        const int ii = i_index[e], jj = j_index[e];
        srcIdx[e] = (ii < 1 || ii > n1 || jj < 1 || jj > n2) ? -1 : (jj - 1) * n1 + (ii - 1);
      }
    }
    cur += n;
  }
  try {
    h->rmRowStart.upload(rowStart); h->rmRowCnt.upload(rowCnt); h->rmSrcIdx.upload(srcIdx);
    h->rmWeight.upload(std::vector<double>(weight, weight + std::max(nOverlap, 1)));
  } catch (const std::string &e) { return fail(h, 91, "mzr_set_remap/" + e); }
  h->remapKind = kind; h->remapSrc = kind == 1 ? n1 : n1 * n2;
  return 0;
}

int mzr_set_sort_map(mzr_handle h, int nSrc, const int *ix_in, int remove_negatives) {
  MZR_FLUSH(h);
  if (!h || !h->haveNet) return h ? fail(h, 20, "mzr_set_sort_map/network not set") : 1;
  if (nSrc < 1 || !ix_in) return fail(h, 20, "mzr_set_sort_map/empty map");
  (void)hipSetDevice(h->cfg.device);
  std::vector<int> srcOf(h->H, -1);
  for (int i = 0; i < nSrc; ++i) {
    const int j = ix_in[i];
    if (j == -9999) continue;                        // process_remap.f90:304-307
    if (j < 1 || j > h->H) return fail(h, 20, "mzr_set_sort_map/index out of range");
    srcOf[j - 1] = i;
  }
  try { h->rmRowStart.upload(srcOf); } catch (const std::string &e) { return fail(h, 91, "mzr_set_sort_map/" + e); }
  h->remapKind = 3; h->remapSrc = nSrc; h->remapRemoveNeg = remove_negatives != 0;
  return 0;
}

int mzr_remap_runoff_dev(mzr_handle h, int nSteps, const double *src_dev, double *dst_dev) {
  MZR_FLUSH(h);
  if (!h || !h->remapKind) return h ? fail(h, 20, "mzr_remap_runoff/no mapping set (mzr_set_remap / mzr_set_sort_map)") : 1;
  if (nSteps < 1) return fail(h, 20, "mzr_remap_runoff/nSteps must be positive");
  (void)hipSetDevice(h->cfg.device);
  if (h->remapKind == 3) mzr_launch_sort_flux(h->H, nSteps, h->remapSrc, h->rmRowStart.p, h->remapRemoveNeg, src_dev, dst_dev, h->stream);
  else {
    const size_t need = (size_t)h->remapSrc * mzr_remap_ld(nSteps);
    if (need > h->rmScratchLen) {
      (void)hipStreamSynchronize(h->stream);
      try { h->rmScratch.alloc(need); } catch (const std::string &e) { return fail(h, 91, "mzr_remap_runoff/" + e); }
      h->rmScratchLen = need;
    }
    mzr_launch_remap(h->H, nSteps, h->remapSrc, h->rmRowStart.p, h->rmRowCnt.p, h->rmSrcIdx.p, h->rmWeight.p, src_dev, h->rmScratch.p, dst_dev, h->stream);
  }
  if (hipGetLastError() != hipSuccess) return fail(h, 92, "mzr_remap_runoff/kernel launch failed");
  return 0;
}

int mzr_run_src_dev(mzr_handle h, int nSteps, double t_start, const double *src_dev) {
  MZR_FLUSH(h);
  if (!h) return 1;
  if (!h->haveState) return fail(h, 20, "mzr_run/state not initialised (call mzr_init_state)");
  if (nSteps < 1 || nSteps > h->cfg.maxWindow) return fail(h, 20, "mzr_run/nSteps exceeds maxWindow");
  if (!ensureRunoffW(h)) return fail(h, 91, "mzr/out of device memory (forcing window)");
  const int rc = mzr_remap_runoff_dev(h, nSteps, src_dev, h->runoffW.p);
  if (rc) return rc;
  const int rc2 = run_window(h, nSteps, t_start, t_start + h->cfg.dt, h->runoffW.p);
  if (rc2) return rc2;
  // the window is queued, not finished: a later mzr_run_async copies into runoffW on its own stream and has to wait for it
  if (!h->rwOther) (void)hipEventCreateWithFlags(&h->rwOther, hipEventDisableTiming);
  (void)hipEventRecord(h->rwOther, h->stream);
  h->rwOtherSet = true;
  return 0;
}

// One time step == one main_route call.  stepBatch = 1 (default): routed and synchronised at once, errors come back with
// the call, as the reference's driver expects (standalone/route_runoff.f90:80-108 calls handle_err after every step).
// stepBatch > 1: the row is put aside and the call returns; the steps are routed as ONE window (time-skewed over the
// stages) when stepBatch of them have come together, or as soon as anything is asked of the handle (a getter,
// mzr_sync, a setter) -- a host that keeps its time loop and fetches results at output frequency then runs at the speed
// of the windows.  Results are bit-identical either way.  A step whose (T0, T1) does not continue the pending ones by
// exactly dt starts a window of its own; the per-step inputs beside the runoff (lake forcing, abstraction / injection,
// target volumes, observations, constituent) handed over by one-step setter calls before the step travel with it
// (mzr_domain::StepRows) -- pending steps that carry other kinds of rows than this one are routed first.
int mzr_step(mzr_handle h, double T0, double T1, const double *runoff) {
  if (!h) return 1;
  if (!h->haveState) return fail(h, 20, "mzr_step/state not initialised (call mzr_init_state)");
  (void)hipSetDevice(h->cfg.device);
  const int cap = std::min(h->cfg.stepBatch, h->cfg.maxWindow);
  if (cap <= 1) {
    MZR_FLUSH(h);
    if (!ensureRunoffW(h)) return fail(h, 91, "mzr/out of device memory (forcing window)");
    (void)hipMemcpyAsync(h->runoffW.p, runoff, (size_t)h->H * sizeof(double), hipMemcpyHostToDevice, h->stream);
    const int rc = run_window(h, 1, T0, T1, h->runoffW.p);
    if (rc) return rc;
    return mzr_sync(h);
  }
  const double dt = h->cfg.dt;
  // not the continuation of what is pending, or a step of another length (a window of its own: the kernels take TSEC(2) of a
  // single step as given and T0 + dt otherwise)
  // the rows handed over for this step (lake forcing, abstraction / injection, ...): the pending steps must carry the same kinds
  unsigned mask = 0;
  for (int k = 0; k < mzr_domain::SR_KINDS; ++k) if (h->sr[k].staged) mask |= 1u << k;
  if (h->stepN > 0 && (!(T0 == h->stepT0 + (double)h->stepN * dt) || !(T1 == T0 + dt) || mask != h->srMask)) { const int rc = flushSteps(h, true); if (rc) return rc; }
  if (!h->stepHost[0] || h->stepCap != cap) {
    { const int rc = flushSteps(h, true); if (rc) return rc; }
    if (h->tail.pending) flushTail(h);
    for (int i = 0; i < 2; ++i) {
      if (h->stepInFlight[i]) { (void)hipEventSynchronize(h->stepCopied[i]); h->stepInFlight[i] = false; }
      if (h->stepHost[i]) { (void)hipHostFree(h->stepHost[i]); h->stepHost[i] = nullptr; }
      if (hipHostMalloc((void **)&h->stepHost[i], (size_t)cap * h->H * sizeof(double), hipHostMallocDefault) != hipSuccess) { h->stepHost[i] = nullptr; return fail(h, 91, "mzr_step/hipHostMalloc failed"); }
      if (!h->stepCopied[i]) (void)hipEventCreateWithFlags(&h->stepCopied[i], hipEventDisableTiming);
    }
    h->stepCap = cap;
  }
  if (h->stepN == 0) {
    h->stepT0 = T0;
    if (h->stepInFlight[h->stepCur]) { (void)hipEventSynchronize(h->stepCopied[h->stepCur]); h->stepInFlight[h->stepCur] = false; }      // its last copy has left
  }
  memcpy(h->stepHost[h->stepCur] + (size_t)h->stepN * h->H, runoff, (size_t)h->H * sizeof(double));
  if (h->stepN == 0) {
    h->srMask = mask;
    for (int k = 0; k < mzr_domain::SR_KINDS; ++k) { for (auto &v : h->sr[k].rows) v.clear(); for (auto &v : h->sr[k].rowsI) v.clear(); }
  }
  for (int k = 0; k < mzr_domain::SR_KINDS; ++k) {
    mzr_domain::StepRows &r = h->sr[k];
    if (!r.staged) continue;
    for (int j = 0; j < 2; ++j) r.rows[j].insert(r.rows[j].end(), r.next[j].begin(), r.next[j].end());
    for (int j = 0; j < 3; ++j) r.rowsI[j].insert(r.rowsI[j].end(), r.nextI[j].begin(), r.nextI[j].end());
    r.staged = false;
  }
  h->srAny = false;
  h->stepT1 = T1;
  ++h->stepN;
  if (h->stepN == cap || !(T1 == T0 + dt)) return flushSteps(h);
  return 0;
}

}  // extern "C"

// The launches s = W .. W + nS - 2 of the last window, kept back for the next one (overlapping windows), on their own: somebody
// wants a result, or the next window cannot take them along.  Same streams as the window's other launches; the handle's
// stream then waits for the others, as it does at the end of every window.
static void issueTail(mzr_handle h, int jBegin, int jEnd);
static void flushTail(mzr_handle h) {
  if (!h->tail.pending) return;
  h->tail.pending = false;
  ++h->tailFlushes;
  (void)hipSetDevice(h->cfg.device);
  issueTail(h, h->tail.next, h->nStages - 1);
  h->tail.next = h->nStages - 1;
  for (int ix = 1; ix < h->cfg.nRoutes; ++ix)
    if (h->routeStream[ix]) { (void)hipEventRecord(h->routeEvent[ix], h->routeStream[ix]); (void)hipStreamWaitEvent(h->stream, h->routeEvent[ix], 0); }
}
// the kept-back launches [jBegin, jEnd) of the last window (launch W + j: the stages above j), each on its method's stream
static void issueTail(mzr_handle h, int jBegin, int jEnd) {
  const int N = h->N, nR = h->cfg.nRoutes, W = h->tail.W;
  const bool prof = h->profiling;
  for (int j = jBegin; j < jEnd; ++j) {
    const int rB = h->stageStart[j + 1];
    if (rB >= N) break;
    for (int ix = 0; ix < nR; ++ix) {
      RouteBufs &rb = h->route[ix];
      hipStream_t sx = (nR > 1 && ix > 0 && h->routeStream[ix]) ? h->routeStream[ix] : h->stream;
      if (prof) {
        if (rb.evUsed == rb.events.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b); }
        (void)hipEventRecord(rb.events[rb.evUsed].first, sx);
      }
      mzr_launch_stage(rb.method, h->tail.d[ix], W + j, rB, N, sx);
      if (prof) { (void)hipEventRecord(rb.events[rb.evUsed].second, sx); ++rb.evUsed; }
      ++rb.nLaunches;
    }
  }
}

// n rows of one kind to its window setter (srApplying: the setter neither stages them again nor flushes)
static int applyRows(mzr_handle h, int kind, int n, std::vector<double> *a, std::vector<int> *ai) {
  h->srApplying = true;
  int rc = 0;
  switch (kind) {
    case mzr_domain::SR_LAKE:   rc = mzr_set_lake_forcing(h, n, a[0].empty() ? nullptr : a[0].data(), a[1].empty() ? nullptr : a[1].data(), ai[0].data(), ai[1].data(), ai[2].data()); break;
    case mzr_domain::SR_WMFLUX: rc = mzr_set_wm_flux(h, n, a[0].data()); break;
    case mzr_domain::SR_WMVOL:  rc = mzr_set_wm_vol(h, n, a[0].data()); break;
    case mzr_domain::SR_SOLUTE: rc = mzr_set_solute(h, n, a[0].data()); break;
    case mzr_domain::SR_OBS:    rc = mzr_set_obs(h, n, ai[0].data(), a[0].data()); break;
    default: break;
  }
  h->srApplying = false;
  return rc;
}

static int flushSteps(mzr_handle h, bool keepStaged) {
  const int n = h->stepN;
  if (n >= 1) {
    h->stepN = 0;
    const int k = h->stepCur;
    h->stepCur ^= 1;
    h->stepInFlight[k] = true;
    for (int kind = 0; kind < mzr_domain::SR_KINDS; ++kind)
      if (h->srMask >> kind & 1) { const int rc = applyRows(h, kind, n, h->sr[kind].rows, h->sr[kind].rowsI); if (rc) return rc; }
    h->srMask = 0;
    const double T1_single = n == 1 ? h->stepT1 : h->stepT0 + h->cfg.dt;
    const int rc = run_async_impl(h, n, h->stepT0, T1_single, h->stepHost[k], h->stepCopied[k]);
    if (rc) return rc;
  }
  if (!keepStaged && h->srAny) {      // rows whose step has not come as mzr_step: to their setters as they are
    h->srAny = false;
    for (int kind = 0; kind < mzr_domain::SR_KINDS; ++kind) {
      mzr_domain::StepRows &r = h->sr[kind];
      if (!r.staged) continue;
      r.staged = false;
      const int rc = applyRows(h, kind, 1, r.next, r.nextI);
      if (rc) return rc;
    }
  }
  return 0;
}

extern "C" {

// ---- getters: device (internal order) -> host (caller order)
static int pullRow(mzr_handle h, const double *src, double *out) {
  std::vector<double> tmp(h->N);
  if (hipMemcpy(tmp.data(), src, h->N * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(h, 92, "mzr_get/hipMemcpy failed");
  for (int e = 0; e < h->N; ++e) out[e] = tmp[h->ext2int[e]];
  return 0;
}

int mzr_get_flux(mzr_handle h, int method, int which, double *out) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_flux/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const size_t N = h->N;
  if (which >= MZR_F_BASIN_QR1) {
    if (which == MZR_F_BASIN_QR1) return pullRow(h, h->qlat.p + (size_t)h->lastW * N, out);
    if (which == MZR_F_BASIN_QR0) return pullRow(h, h->qlat.p + (size_t)std::max(0, h->lastW - 1) * N, out);
    if (which == MZR_F_BASIN_QI && h->cfg.doesBasinRoute == 1 && h->lastW > 0) return pullRow(h, h->qi.p + (size_t)(h->lastW - 1) * N, out);
    return fail(h, 20, "mzr_get_flux/field not available");
  }
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_flux/method not active");
  RouteBufs &rb = h->route[ix];
  switch (which) {
    case MZR_F_Q: return pullRow(h, rb.Q.p + (size_t)std::max(0, h->lastW - 1) * N, out);
    case MZR_F_VOL0: return pullRow(h, rb.vol0.p, out);
    case MZR_F_VOL1: return pullRow(h, rb.vol.p, out);
    case MZR_F_INFLOW: return pullRow(h, rb.inflow.p, out);
    case MZR_F_ELE: return pullRow(h, rb.ele.p, out);
    case MZR_F_FLOODVOL: return pullRow(h, rb.floodvol.p, out);
    case MZR_F_WB: return pullRow(h, rb.wb.p, out);
  }
  return fail(h, 20, "mzr_get_flux/unknown field");
}

// comp_global_wb (water_balance.f90:191-323) for the last routed step: the seven sums over the whole domain and the
// error term 8 = 1 - (2+3+4+5+6).  Reaches are summed in the library's internal order (the reference sums mainstem,
// then tributaries, then across MPI ranks -- another order of the same additions).
int mzr_get_global_wb(mzr_handle h, int method, double *out8) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_global_wb/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_global_wb/method not active");
  if (h->lastW < 1) return fail(h, 20, "mzr_get_global_wb/no step has been routed");
  const size_t N = h->N;
  const int tl = h->lastW - 1;
  RouteBufs &rb = h->route[ix];
  std::vector<double> vol(N), vol0(N), qr(N), q(N), act(N, 0.0), dem(N, 0.0);
  auto pull = [&](std::vector<double> &v, const double *src, size_t n) { return hipMemcpy(v.data(), src, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess; };
  bool ok = pull(vol, rb.vol.p, N) && pull(vol0, rb.vol0.p, N) && pull(qr, h->qlat.p + (size_t)h->lastW * N, N) && pull(q, rb.Q.p + (size_t)tl * N, N);
  const bool wm = h->cfg.is_flux_wm && h->wm.p;          // (the window's series stay in place until the next upload)
  if (wm) ok = ok && pull(act, rb.wmact.p, N) && pull(dem, h->wm.p + (size_t)tl * N, N);
  std::vector<double> ev(h->nLake, 0.0), pr(h->nLake, 0.0);
  if (h->nLake && h->lakeEvap.p) ok = ok && pull(ev, h->lakeEvap.p + (size_t)tl * h->nLake, h->nLake) && pull(pr, h->lakePrecip.p + (size_t)tl * h->nLake, h->nLake);
  if (!ok) return fail(h, 92, "mzr_get_global_wb/hipMemcpy failed");
  const double dt = h->cfg.dt;
  double b[7] = {0, 0, 0, 0, 0, 0, 0};
  for (size_t r = 0; r < N; ++r) {
    b[0] += vol[r] - vol0[r];
    b[1] += qr[r] * dt;
    const int ls = h->nLake ? h->h_lakeSlot[r] : -1;
    if (ls >= 0) { b[2] += pr[ls] * dt; b[4] -= ev[ls] * dt; }
    if (wm) b[3] -= act[r] * dt;                            // (a reach the data set does not name carries realMissing here, as in the reference)
    if (h->h_down[r] < 0) b[5] -= q[r] * dt;
    if (wm) b[6] -= dem[r] * dt;
  }
  for (int i = 0; i < 7; ++i) out8[i] = b[i];
  out8[7] = b[0] - (b[1] + b[2] + b[3] + b[4] + b[5]);
  return 0;
}

int mzr_get_window_q(mzr_handle h, int method, double *out) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_window_q/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_window_q/method not active");
  const int N = h->N, W = h->lastW;
  if (W < 1) return fail(h, 20, "mzr_get_window_q/no window has been run");
  dim3 block(256), grid((N + 255) / 256, W);
  if (!ensureScratch(h)) return fail(h, 91, "mzr/out of device memory (row scratch)");
  hipLaunchKernelGGL(k_gather_rows, grid, block, 0, h->stream, h->route[ix].Q.p, h->scratchOut.p, h->d_ext2int.p, N, W);
  if (hipMemcpyAsync(out, h->scratchOut.p, (size_t)W * N * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess)
    return fail(h, 92, "mzr_get_window_q/hipMemcpy failed");
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, 92, "mzr_get_window_q/sync failed");
  return 0;
}

int mzr_get_mean_q(mzr_handle h, int method, double *out, int reset) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_mean_q/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_mean_q/method not active");
  rc = pullRow(h, h->route[ix].qsum.p, out); if (rc) return rc;
  const double n = (double)std::max<long long>(1, h->route[ix].meanSteps);      // steps accumulated in THIS method's sum
  for (int e = 0; e < h->N; ++e) out[e] /= n;
  if (reset) { h->route[ix].qsum.zero(h->stream); h->route[ix].meanSteps = 0; (void)hipStreamSynchronize(h->stream); }
  return 0;
}

// History accumulation beyond discharge (histVars_data.f90:154-305): which sums are kept.  Takes effect at the next mzr_init_state.
int mzr_set_history(mzr_handle h, int flags) {
  MZR_FLUSH(h);
  if (!h) return 1;
  h->histFlags = flags & (MZR_H_INFLOW | MZR_H_HEIGHT | MZR_H_RUNOFF);
  h->haveState = false;
  return 0;
}

int mzr_get_mean(mzr_handle h, int method, int which, double *out) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_mean/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  if (which >= MZR_M_INST_RUNOFF) {
    if (!(h->histFlags & MZR_H_RUNOFF)) return fail(h, 20, "mzr_get_mean/runoff sums are off (mzr_set_history)");
    const double n = (double)std::max<long long>(1, h->histSteps);
    if (which == MZR_M_BAS_RUNOFF) {
      MZR_COPY(out, h->hBas.p, (size_t)h->H * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_mean");
      for (int i = 0; i < h->H; ++i) out[i] /= n;
      return 0;
    }
    rc = pullRow(h, which == MZR_M_INST_RUNOFF ? h->hInst.p : h->hDlay.p, out); if (rc) return rc;
    for (int e = 0; e < h->N; ++e) out[e] /= n;
    return 0;
  }
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_mean/method not active");
  RouteBufs &rb = h->route[ix];
  const double *src = which == MZR_M_Q ? rb.qsum.p : which == MZR_M_INFLOW ? rb.hInflow.p : which == MZR_M_HEIGHT ? rb.hEle.p : which == MZR_M_FLOODVOL ? rb.hFlood.p : nullptr;
  if (!src) return fail(h, 20, "mzr_get_mean/this sum is off (mzr_set_history) or unknown");
  rc = pullRow(h, src, out); if (rc) return rc;
  const double n = (double)std::max<long long>(1, rb.meanSteps);
  for (int e = 0; e < h->N; ++e) out[e] /= n;
  return 0;
}

// histVars%refresh: every sum back to zero
int mzr_reset_means(mzr_handle h) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_reset_means/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
    RouteBufs &rb = h->route[ix];
    rb.qsum.zero(h->stream); rb.hInflow.zero(h->stream); rb.hEle.zero(h->stream); rb.hFlood.zero(h->stream); rb.meanSteps = 0;
  }
  h->hInst.zero(h->stream); h->hDlay.zero(h->stream); h->hBas.zero(h->stream); h->histSteps = 0;
  (void)hipStreamSynchronize(h->stream);
  return 0;
}

int mzr_get_kwt_state(mzr_handle h, int *numWaves, double *qwave, double *tentry, double *texit, int *routed) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->kwN.p) return h ? fail(h, 20, "mzr_get_kwt_state/KWT not active") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N;
  std::vector<int> n(N);
  std::vector<double> q((size_t)2 * MZR_KW_STRIDE * N), tr((size_t)MZR_KW_STRIDE * N);      // q: {Q, TI} pairs
  MZR_COPY(n.data(), h->kwN.p, N * sizeof(int), hipMemcpyDeviceToHost, "mzr_get_kwt_state");
  MZR_COPY(q.data(), h->kwQ.p, q.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_kwt_state");
  MZR_COPY(tr.data(), h->kwTR.p, tr.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_kwt_state");
  for (int e = 0; e < N; ++e) {
    const int i = h->ext2int[e];
    numWaves[e] = n[i];
    for (int k = 0; k < MZR_WCAP; ++k) {
      const size_t o = (size_t)e * MZR_WCAP + k;
      if (k < n[i]) {
        qwave[o] = q[MZR_PQ(MZR_KWI(k, i))]; tentry[o] = q[MZR_PT(MZR_KWI(k, i))]; texit[o] = tr[MZR_KWI(k, i)];
        const bool lake = !h->h_lakeSlot.empty() && h->h_lakeSlot[i] >= 0;   // a lake keeps one sentinel particle
        routed[o] = (k == 0 && h->h_nGood[i] > 0 && !lake) ? 1 : 0;          // element 0 = last routed particle
      } else { qwave[o] = tentry[o] = texit[o] = -9999.0; routed[o] = 0; }
    }
  }
  return 0;
}

int mzr_set_kwt_state(mzr_handle h, const int *numWaves, const double *qwave, const double *tentry, const double *texit, const int *routed) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->kwN.p) return h ? fail(h, 20, "mzr_set_kwt_state/KWT not active") : 1;
  (void)routed;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N;
  std::vector<int> n(N);
  std::vector<double> q((size_t)2 * MZR_KW_STRIDE * N, 0.0), tr((size_t)MZR_KW_STRIDE * N, 0.0);      // q: {Q, TI} pairs
  for (int e = 0; e < N; ++e) {
    const int i = h->ext2int[e];
    if (numWaves[e] > MZR_KW_CAP) return fail(h, 20, "mzr_set_kwt_state/more than MAXQPAR waves in a reach");
    if (numWaves[e] < 0) return fail(h, 20, "mzr_set_kwt_state/negative number of waves in a reach");
    n[i] = numWaves[e];
    for (int k = 0; k < numWaves[e]; ++k) {
      const size_t o = (size_t)e * MZR_WCAP + k;
      q[MZR_PQ(MZR_KWI(k, i))] = qwave[o]; q[MZR_PT(MZR_KWI(k, i))] = tentry[o]; tr[MZR_KWI(k, i)] = texit[o];
    }
  }
  MZR_COPY(h->kwN.p, n.data(), N * sizeof(int), hipMemcpyHostToDevice, "mzr_set_kwt_state");
  MZR_COPY(h->kwQ.p, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_kwt_state");
  MZR_COPY(h->kwTR.p, tr.data(), tr.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_kwt_state");
  return 0;
}

int mzr_get_irf_state(mzr_handle h, double *qfuture) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->irfQ.p) return h ? fail(h, 20, "mzr_get_irf_state/IRF not active") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N;
  std::vector<double> v((size_t)h->maxtdh * N);
  MZR_COPY(v.data(), h->irfQ.p, v.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_irf_state");
  for (int e = 0; e < N; ++e) {
    const int i = h->ext2int[e];
    for (int j = 0; j < h->uhOff[e + 1] - h->uhOff[e]; ++j) qfuture[h->uhOff[e] + j] = v[(size_t)j * N + i];
  }
  return 0;
}

int mzr_get_mol_state(mzr_handle h, int method, double *qout) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_get_mol_state/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0 || !h->route[ix].mol.p) return fail(h, 81, "mzr_get_mol_state/method not active");
  const int N = h->N, nm = method == MZR_MC ? MZR_NMOL_MC : MZR_NMOL_KW;
  std::vector<double> v((size_t)nm * N);
  MZR_COPY(v.data(), h->route[ix].mol.p, v.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_mol_state");
  for (int e = 0; e < N; ++e) for (int j = 0; j < nm; ++j) qout[(size_t)e * nm + j] = v[(size_t)j * N + h->ext2int[e]];
  return 0;
}

int mzr_get_basin_state(mzr_handle h, double *qfuture) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || h->cfg.doesBasinRoute != 1) return h ? fail(h, 20, "mzr_get_basin_state/hillslope routing not active") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N, n = h->ntdhBas;
  std::vector<double> v((size_t)n * N);
  MZR_COPY(v.data(), h->basS[h->basCur].p, v.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_basin_state");
  for (int e = 0; e < N; ++e) for (int j = 0; j < n; ++j) qfuture[(size_t)e * n + j] = v[(size_t)j * N + h->ext2int[e]];
  return 0;
}

// ---- state setters (restart, read_restart.f90:152-742): caller order -> device (internal order)
int mzr_set_irf_state(mzr_handle h, const double *qfuture) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->irfQ.p) return h ? fail(h, 20, "mzr_set_irf_state/IRF not active") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N;
  std::vector<double> v((size_t)h->maxtdh * N, 0.0);
  for (int e = 0; e < N; ++e) {
    const int i = h->ext2int[e];
    for (int j = 0; j < h->uhOff[e + 1] - h->uhOff[e]; ++j) v[(size_t)j * N + i] = qfuture[h->uhOff[e] + j];
  }
  MZR_COPY(h->irfQ.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_irf_state");
  return 0;
}

int mzr_set_mol_state(mzr_handle h, int method, const double *q) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_mol_state/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0 || !h->route[ix].mol.p) return fail(h, 81, "mzr_set_mol_state/method not active");
  const int N = h->N, nm = method == MZR_MC ? MZR_NMOL_MC : MZR_NMOL_KW;
  std::vector<double> v((size_t)nm * N);
  for (int e = 0; e < N; ++e) for (int j = 0; j < nm; ++j) v[(size_t)j * N + h->ext2int[e]] = q[(size_t)e * nm + j];
  MZR_COPY(h->route[ix].mol.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_mol_state");
  return 0;
}

// qfuture[nRch][n] = hillslope QFUTURE, basin_q[nRch] = BASIN_QR(1) (read_restart.f90:190,246)
int mzr_set_basin_state(mzr_handle h, const double *qfuture, const double *basin_q) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_basin_state/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N, n = h->ntdhBas;
  if (qfuture) {
    if (h->cfg.doesBasinRoute != 1) return fail(h, 20, "mzr_set_basin_state/hillslope routing not active");
    std::vector<double> v((size_t)n * N);
    for (int e = 0; e < N; ++e) for (int j = 0; j < n; ++j) v[(size_t)j * N + h->ext2int[e]] = qfuture[(size_t)e * n + j];
    MZR_COPY(h->basS[h->basCur].p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_basin_state");
  }
  if (basin_q) {   // row lastW of qlat is BASIN_QR(1) of the last step; the next window starts from it
    std::vector<double> v(N);
    for (int e = 0; e < N; ++e) v[h->ext2int[e]] = basin_q[e];
    MZR_COPY(h->qlat.p + (size_t)h->lastW * N, v.data(), N * sizeof(double), hipMemcpyHostToDevice, "mzr_set_basin_state");
  }
  return 0;
}

// Constituent state (restart variables tfuture(seg, tdh) and solute_mass(seg), write_restart_pio.f90:941-971,1292-):
// tfuture [nRch][ntdhBas] (hillslope routing on) and / or the mass in the reaches of one method; null = leave out.
// The lateral flux of the last step (BASIN_solute) is not part of the reference's restart and is not needed: every step
// derives it anew.
int mzr_get_tracer_state(mzr_handle h, int method, double *tfuture, double *mass) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->tracer) return h ? fail(h, 20, "mzr_get_tracer_state/constituent routing is off") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N, n = h->ntdhBas;
  if (tfuture) {
    if (h->cfg.doesBasinRoute != 1) return fail(h, 20, "mzr_get_tracer_state/hillslope routing not active");
    std::vector<double> v((size_t)n * N);
    MZR_COPY(v.data(), h->solS[h->solCur].p, v.size() * sizeof(double), hipMemcpyDeviceToHost, "mzr_get_tracer_state");
    for (int e = 0; e < N; ++e) for (int j = 0; j < n; ++j) tfuture[(size_t)e * n + j] = v[(size_t)j * N + h->ext2int[e]];
  }
  if (mass) {
    const int ix = idxOf(h, method);
    if (ix < 0 || method == MZR_SUM) return fail(h, 81, "mzr_get_tracer_state/method not active (or the runoff accumulation)");
    return pullRow(h, h->route[ix].solMass.p, mass);
  }
  return 0;
}
int mzr_set_tracer_state(mzr_handle h, int method, const double *tfuture, const double *mass) {
  MZR_FLUSH(h);
  if (!h || !h->haveState || !h->tracer) return h ? fail(h, 20, "mzr_set_tracer_state/constituent routing is off") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int N = h->N, n = h->ntdhBas;
  if (tfuture) {
    if (h->cfg.doesBasinRoute != 1) return fail(h, 20, "mzr_set_tracer_state/hillslope routing not active");
    std::vector<double> v((size_t)n * N);
    for (int e = 0; e < N; ++e) for (int j = 0; j < n; ++j) v[(size_t)j * N + h->ext2int[e]] = tfuture[(size_t)e * n + j];
    MZR_COPY(h->solS[h->solCur].p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_tracer_state");
  }
  if (mass) {
    const int ix = idxOf(h, method);
    if (ix < 0 || method == MZR_SUM) return fail(h, 81, "mzr_set_tracer_state/method not active (or the runoff accumulation)");
    std::vector<double> v(N);
    for (int e = 0; e < N; ++e) v[h->ext2int[e]] = mass[e];
    MZR_COPY(h->route[ix].solMass.p, v.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, "mzr_set_tracer_state");
  }
  return 0;
}

// REACH_VOL(1) of a method (volume_<method> of the restart file)
int mzr_set_volume(mzr_handle h, int method, const double *vol) {
  MZR_FLUSH(h);
  if (!h || !h->haveState) return h ? fail(h, 20, "mzr_set_volume/state not initialised") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_set_volume/method not active");
  std::vector<double> v(h->N);
  for (int e = 0; e < h->N; ++e) v[h->ext2int[e]] = vol[e];
  MZR_COPY(h->route[ix].vol.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, "mzr_set_volume");
  return 0;
}

// debug: per-section wave cycles of the KWT kernel (library built with -DMZR_KWT_TIMING)
int mzr_debug_cycles(mzr_handle h, unsigned long long *out32, int reset) {
  if (!h || !h->dbgCycles.p) return 1;
  (void)hipStreamSynchronize(h->stream);
  std::vector<unsigned long long> all((size_t)32 * 1024);   // 1024 slots of 32 counters (uncontended atomics), summed here
  (void)hipMemcpy(all.data(), h->dbgCycles.p, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i) { out32[i] = 0; for (int k = 0; k < 1024; ++k) out32[i] += all[(size_t)k * 32 + i]; }
  if (reset) (void)hipMemset(h->dbgCycles.p, 0, all.size() * sizeof(unsigned long long));
  return 0;
}

// debug: per-pass records of a timing build (16 unsigned each), newest `n` at most
int mzr_debug_records(mzr_handle h, unsigned *out, int n) {
  if (!h || !h->dbgCycles.p) return -1;
  (void)hipStreamSynchronize(h->stream);
  std::vector<unsigned> all((size_t)16 + (size_t)65536 * 16);
  (void)hipMemcpy(all.data(), h->dbgCycles.p + 32 * 1024, all.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  const int have = (int)std::min<unsigned>(all[0], 65536u), m = std::min(n, have);
  memcpy(out, all.data() + 16, (size_t)m * 16 * sizeof(unsigned));
  return m;
}

// debug: raw 64-bit words of the debug buffer
int mzr_debug_raw(mzr_handle h, long long first, long long n, unsigned long long *out) {
  if (!h || !h->dbgCycles.p || first < 0 || n < 0 || (size_t)(first + n) > h->dbgCycles.n) return 1;
  (void)hipStreamSynchronize(h->stream);
  return hipMemcpy(out, h->dbgCycles.p + first, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}

int mzr_get_schedule(mzr_handle h, int *nStages, int *maxStageWidth) {
  if (!h || !h->haveNet) return 1;
  *nStages = h->nStages; *maxStageWidth = h->maxStageWidth;
  return 0;
}

int mzr_get_sweep_info(mzr_handle h, int *nWaves, int *capacity, int *nItems) {
  if (!h || !h->haveState) return 1;
  *nWaves = h->swWaves; *capacity = h->swCap; *nItems = h->swItems;
  return 0;
}

int mzr_get_sweep_retries(mzr_handle h, long long *nRetries) {
  if (!h || !nRetries) return 1;
  *nRetries = h->sweepRetries;
  return 0;
}

int mzr_get_sweep_clock(mzr_handle h, int maxN, double *ms, int *n, int reset) {
  MZR_FLUSH(h);
  if (!h || !n) return 1;
  *n = 0;
  if (!h->swClock.p) return 0;
  (void)hipSetDevice(h->cfg.device);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, 92, "mzr_get_sweep_clock/device error");
  const long long have = std::min<long long>(h->swClockN, MZR_CLOCK_LOG);
  std::vector<unsigned long long> v(2 * MZR_CLOCK_LOG);
  if (hipMemcpy(v.data(), h->swClock.p, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return fail(h, 92, "mzr_get_sweep_clock/hipMemcpy failed");
  const long long take = std::min<long long>(have, maxN > 0 ? maxN : 0);
  for (long long k = 0; k < take; ++k) {      // the latest `take` launches, oldest first
    const long long seq = h->swClockN - take + k;
    const unsigned long long a = v[2 * (seq % MZR_CLOCK_LOG)], b = v[2 * (seq % MZR_CLOCK_LOG) + 1];
    if (ms) ms[k] = (a && b > a) ? (double)(b - a) * 1e-5 : 0.0;      // 100 MHz ticks -> ms (0: the launch did not run to its end)
  }
  *n = (int)take;
  if (reset) h->swClockN = 0;
  return 0;
}

int mzr_get_sweep_arrivals(mzr_handle h, int *arrivedLast, int *joinedLast, long long *hist32) {
  MZR_FLUSH(h);
  if (!h || !h->swHead.p) return 1;
  int v[16 + 32];
  if (hipMemcpy(v, h->swHead.p + 8 * 16, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, 92, "mzr_get_sweep_arrivals/hipMemcpy failed");
  if (arrivedLast) *arrivedLast = v[2];
  if (joinedLast) *joinedLast = v[3];
  if (hist32) for (int k = 0; k < 32; ++k) hist32[k] = v[16 + k];
  return 0;
}

int mzr_set_profiling(mzr_handle h, int mode) {
  if (!h) return 1;
  const bool on = (mode & 1) != 0;
  if (on != h->profiling) for (auto &rb : h->route) rb.evUsed = 0;      // (event pairs recorded but never read belong to nobody)
  h->profiling = on; h->countTraffic = (mode & 2) != 0;
  if (on)      // event pairs made ahead: none is created inside a window (the KWT sweep takes one pair per window)
    for (int ix = 0; ix < h->cfg.nRoutes; ++ix) {
      RouteBufs &rb = h->route[ix];
      if (rb.method != MZR_KWT) continue;
      (void)hipSetDevice(h->cfg.device);
      while (rb.events.size() < 64) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); rb.events.emplace_back(a, b); }
    }
  return 0;
}

int mzr_get_timing(mzr_handle h, int method, long long *nLaunches, double *kernel_ms, long long *reachSteps, int reset) {
  MZR_FLUSH(h);
  if (!h) return 1;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_timing/method not active");
  RouteBufs &rb = h->route[ix];
  *nLaunches = rb.nLaunches; *kernel_ms = rb.kernel_ms; *reachSteps = rb.reachSteps;
  if (reset) { rb.nLaunches = 0; rb.kernel_ms = 0; rb.reachSteps = 0; }
  return 0;
}

// shortest and longest event-timed launch since the last reset (mode 1); 0 = none
int mzr_get_timing_range(mzr_handle h, int method, double *min_ms, double *max_ms, int reset) {
  MZR_FLUSH(h);
  if (!h) return 1;
  const int ix = idxOf(h, method);
  if (ix < 0) return fail(h, 81, "mzr_get_timing_range/method not active");
  RouteBufs &rb = h->route[ix];
  if (min_ms) *min_ms = rb.kernel_ms_min;
  if (max_ms) *max_ms = rb.kernel_ms_max;
  if (reset) { rb.kernel_ms_min = 0.0; rb.kernel_ms_max = 0.0; }
  return 0;
}

int mzr_get_kwt_traffic(mzr_handle h, long long *w_in, long long *w_up, long long *w_out, long long *n_head,
                        long long *n_route, long long *n_edges, int reset) {
  MZR_FLUSH(h);
  if (!h || !h->kwtStat.p) return h ? fail(h, 20, "mzr_get_kwt_traffic/KWT not active") : 1;
  int rc = mzr_sync(h); if (rc) return rc;
  MzrKwtStat s;
  (void)hipMemcpy(&s, h->kwtStat.p, sizeof s, hipMemcpyDeviceToHost);
  *w_in = (long long)s.w_in; *w_up = (long long)s.w_up; *w_out = (long long)s.w_out;
  *n_head = (long long)s.n_head + h->kwtHeadSteps; *n_route = (long long)s.n_route; *n_edges = (long long)s.n_edges;
  if (reset) { (void)hipMemset(h->kwtStat.p, 0, sizeof s); h->kwtHeadSteps = 0; }
  return 0;
}

// ---- boundary-record transport: RCCL point-to-point, loaded at run time -------------------------------------
namespace {
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_commMsg;
int commFail(int code, const std::string &m) { g_commMsg = m; return code; }
int rcclLoad() {
  if (g_rccl.lib) return 0;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
  if (!g_rccl.lib) return commFail(94, std::string("mzr_comm/cannot load librccl.so: ") + dlerror());
#define MZR_SYM(field, sym) do { *(void **)(&g_rccl.field) = dlsym(g_rccl.lib, sym); if (!g_rccl.field) { g_rccl.lib = nullptr; return commFail(94, "mzr_comm/librccl.so lacks " sym); } } while (0)
  MZR_SYM(GetUniqueId, "ncclGetUniqueId"); MZR_SYM(CommInitRank, "ncclCommInitRank"); MZR_SYM(CommDestroy, "ncclCommDestroy");
  MZR_SYM(Send, "ncclSend"); MZR_SYM(Recv, "ncclRecv"); MZR_SYM(GroupStart, "ncclGroupStart"); MZR_SYM(GroupEnd, "ncclGroupEnd");
  MZR_SYM(GetErrorString, "ncclGetErrorString");
#undef MZR_SYM
  return 0;
}
int rcclCheck(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return 0;
  return commFail(95, std::string("mzr_comm/") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
}  // namespace

struct mzr_comm_s { ncclComm_t comm = nullptr; int rank = 0, nRanks = 1, device = 0; hipStream_t stream = nullptr; hipEvent_t ev = nullptr; };

int mzr_comm_last_error(char *buf, int len) {
  if (!buf || len <= 0) return 1;
  snprintf(buf, len, "%s", g_commMsg.c_str());
  return 0;
}

int mzr_comm_unique_id(char id[128]) {
  if (!id) return 1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (int rc = rcclLoad()) return rc;
  ncclUniqueId u;
  if (int rc = rcclCheck(g_rccl.GetUniqueId(&u), "ncclGetUniqueId")) return rc;
  memcpy(id, &u, 128);
  return 0;
}

int mzr_comm_init(int rank, int nRanks, const char id[128], int device, mzr_comm *out) {
  if (!out || !id || nRanks < 1 || rank < 0 || rank >= nRanks) return commFail(1, "mzr_comm_init/bad arguments");
  if (int rc = rcclLoad()) return rc;
  if (hipSetDevice(device) != hipSuccess) return commFail(90, "mzr_comm_init/hipSetDevice failed");
  mzr_comm_s *c = new mzr_comm_s();
  c->rank = rank; c->nRanks = nRanks; c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, 128);
  if (int rc = rcclCheck(g_rccl.CommInitRank(&c->comm, nRanks, u, rank), "ncclCommInitRank")) { delete c; return rc; }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) {
    delete c; return commFail(90, "mzr_comm_init/hipStreamCreate failed");
  }
  *out = c;
  return 0;
}

int mzr_comm_destroy(mzr_comm c) {
  if (!c) return 0;
  int rc = 0;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) rc = rcclCheck(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->ev) (void)hipEventDestroy(c->ev);
  delete c;
  return rc;
}

int mzr_comm_send(mzr_comm c, mzr_handle h, const double *dev, long long n, int peer) {
  MZR_FLUSH_STEPS(h);      // (not the launches kept back for the next window: the record was packed by an export, which issued them if it needed them)
  if (!c || !h || !dev || n < 0 || peer < 0 || peer >= c->nRanks || peer == c->rank) return commFail(1, "mzr_comm_send/bad arguments");
  (void)hipSetDevice(c->device);
  // the record was packed by the handle's last mzr_export_boundary_dev; whatever the handle has queued since (the next
  // window) must not hold the transfer up, so it runs on the communicator's own stream behind that export only
  if (h->exportDone) (void)hipStreamWaitEvent(c->stream, h->exportDone, 0);
  if (h->exportPrevDone) (void)hipStreamWaitEvent(c->stream, h->exportPrevDone, 0);
  return rcclCheck(g_rccl.Send(dev, (size_t)n, ncclDouble, peer, c->comm, c->stream), "ncclSend");
}

int mzr_comm_sync(mzr_comm c) {
  if (!c) return 1;
  (void)hipSetDevice(c->device);
  return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : commFail(92, "mzr_comm_sync/device error");
}

int mzr_comm_recv(mzr_comm c, mzr_handle h, double *dev, long long n, int peer) {
  MZR_FLUSH(h);
  if (!c || !h || !dev || n < 0 || peer < 0 || peer >= c->nRanks || peer == c->rank) return commFail(1, "mzr_comm_recv/bad arguments");
  (void)hipSetDevice(c->device);
  if (int rc = rcclCheck(g_rccl.Recv(dev, (size_t)n, ncclDouble, peer, c->comm, c->stream), "ncclRecv")) return rc;
  (void)hipEventRecord(c->ev, c->stream);
  (void)hipStreamWaitEvent(h->stream, c->ev, 0);      // what the handle queues from now on (the import) sees the record
  return 0;
}

int mzr_comm_recv_many(mzr_comm c, mzr_handle h, int nPeers, double *const *dev, const long long *n, const int *peers) {
  MZR_FLUSH(h);
  if (!c || !h || nPeers < 0 || (nPeers > 0 && (!dev || !n || !peers))) return commFail(1, "mzr_comm_recv_many/bad arguments");
  (void)hipSetDevice(c->device);
  if (int rc = rcclCheck(g_rccl.GroupStart(), "ncclGroupStart")) return rc;
  int rc = 0;
  for (int i = 0; i < nPeers && !rc; ++i) {
    if (peers[i] < 0 || peers[i] >= c->nRanks || peers[i] == c->rank) rc = commFail(1, "mzr_comm_recv_many/bad peer");
    else rc = rcclCheck(g_rccl.Recv(dev[i], (size_t)n[i], ncclDouble, peers[i], c->comm, c->stream), "ncclRecv");
  }
  const int rc2 = rcclCheck(g_rccl.GroupEnd(), "ncclGroupEnd");
  (void)hipEventRecord(c->ev, c->stream);
  (void)hipStreamWaitEvent(h->stream, c->ev, 0);
  return rc ? rc : rc2;
}

}  // extern "C"
