// Lagrangian kinematic-wave tracking (KWT) stage kernel for gfx950: one lane per reach, one launch
// per stage of the time-skewed level sweep (see kernels_route.hip for the schedule).
//
// Replaces kwt_rch and its helpers, route/build/src/kwt_route.f90:
//   kwt_rch 36-346, getusq_rch 461-613, qexmul_rch 619-993, remove_rch 999-1123,
//   kinwav_rch 1130-1439 (+ rUpdate 1409-1437), interp_rch 1444-1622.
//
// Data-flow redesign (results are unchanged):
//  * The reference lets the DOWNSTREAM reach strip the routed particles out of its upstream
//    reach's list (kwt_route.f90:822-848).  The stripped list is a pure function of the upstream
//    reach's own result -- KWAVE(NR+1:NQ2+1), the same slice an outlet keeps for itself
//    (:325-344) -- so here every reach stores that at-rest slice itself (kwN/kwQ/kwTI/kwTR,
//    <= 20 particles) and publishes what its downstream reach needs, KWAVE(0:NR+1) plus the first
//    non-routed particle, flow and exit time only, in a per-reach OUTBOX (obN/obQ/obT).  No lane
//    ever writes another reach's state, and the outbox is double-buffered on the parity of the
//    time step so that reach u may already work on step t+1 while its downstream reach consumes
//    step t in the same launch.
//  * Expected exit times of a reach's own waiting particles are recomputed by kinwav every step,
//    so only TR of element 0 is read back; the others are written for restart files only.
//  * Work arrays (own particles + merged upstream particles) live in LDS: each wavefront carves a
//    1024-particle pool among its 64 reaches by need (prefix sum), no private-memory arrays.
//
// Bound by HBM traffic of the particle rows: see DESIGN.md for the bytes-per-reach-step model.
#include <float.h>
#include "mzr_device.h"
#include "lake_device.h"

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


// ---- x**0.4 and x**0.6 for x >= 0 -------------------------------------------------------------
// The celerity law wc = (5/3) K**0.6 q**0.4 (kwt_route.f90:1290) and the stage inversion
// A = (q/K)**0.6 of the shock speed (:1331-1332) are the only transcendental work of this kernel,
// and a generic FP64 pow() costs more instructions than everything else a reach does.
//   x**(2/5) = z*v * 2**(2k),  v = (z**3)**(-1/5)      x = z * 2**(5k), z in [1,32)
//   x**(3/5) = z*w * 2**(3k),  w = (z**2)**(-1/5)
// Inverse fifth root: single-precision hardware seed (v_log_f32 / v_exp_f32), two division-free
// Newton steps v <- v*(1.2 - 0.2*a*v**5), the last with fused residual.  The reference raises to
// the DOUBLES nearest 0.4 / 0.6 (exponents (ALFA-1)/ALFA and 1/ALFA evaluated in FP64), which
// differ from 2/5 and 3/5 by -+2.22e-17; the factor (1 + delta*ln x) restores that.  Measured
// against the correctly rounded power: max 2.3 ulp, mean 0.44 ulp (same class as libm's pow).
__device__ __forceinline__ double pow_fifths(double x, bool three) {
  if (!(x > 0.0)) return x == 0.0 ? 0.0 : NAN;
  if (isinf(x)) return x;
  int e;
  const double m = frexp(x, &e);            // x = m * 2**e, m in [0.5,1)
  const int e1 = e - 1;                     // x = (2m) * 2**e1 ; e1 = 5k + j, j in 0..4
  const int k = (e1 >= 0) ? e1 / 5 : -((4 - e1) / 5);
  const int j = e1 - 5 * k;
  const double z = ldexp(m, j + 1);
  const float lz = __builtin_amdgcn_logf((float)z);                        // log2(z)
  const double a = three ? z * z : z * z * z;
  double v = (double)__builtin_amdgcn_exp2f((three ? -0.4f : -0.6f) * lz);
  {
    const double v2 = v * v, v4 = v2 * v2;
    v = v * (1.2 - 0.2 * (a * (v4 * v)));
  }
  {
    const double v2 = v * v, v4 = v2 * v2, v5 = v4 * v;
    v = fma(v, 0.2 * fma(-a, v5, 1.0), v);
  }
  const double lnx = ((double)lz + 5.0 * (double)k) * 0.6931471805599453;
  const double dl = three ? -2.2204460492503132e-17 : 2.2204460492503132e-17;
  double y = z * v;
  y = fma(y, dl * lnx, y);
  return ldexp(y, three ? 3 * k : 2 * k);
}
__device__ __forceinline__ double pow_0p4(double x) { return pow_fifths(x, false); }   // x**((ALFA-1)/ALFA)
__device__ __forceinline__ double pow_0p6(double x) { return pow_fifths(x, true); }    // x**(1/ALFA)

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// qexmul_rch (:619-993) for the binary confluence (at most two upstream reaches), the shape of
// almost every reach of a river network.  The reference's generic k-way merge then reduces to:
//   * every upstream contributes its hillslope series {BASIN_QR(0)@T0, BASIN_QR(1)@T1}: a straight
//     line, whose only own particle sits at T1 and is the LAST thing merged (MINLOC ties go to
//     the lowest series index, and the basin series come first);
//   * each non-headwater upstream contributes its routed particles (exit time < T1) and the
//     interpolated end-of-step particle at exactly T1;
//   so the output is the 2-way merge of the routed particles in time order (duplicated times
//   emitted once), each with the other series interpolated at that time, followed by one particle
//   at T1.  Arithmetic (scaling by width ratio, slope/prediction form, summation order over the
//   series) is the reference's, statement by statement (:929-957).
// The two particles bracketing each cursor live in registers and the next one is prefetched, so
// the global-memory latency of the outbox overlaps the arithmetic of the current particle.
__device__ __forceinline__ int kwt_merge_binary(int nup, int u0, double RW, double T0, double T1,
                                                const uint8_t *nGood, const double *width,
                                                const double *qlat_prev, const double *qlat_cur, const int *obN,
                                                const double *obQ, const double *obT, int N, double *QD, double *TD) {
  const double bsc = 1.0 / RW;               // UWIDTH(basin) = 1
  const double dT10 = T1 - T0;
  double b0q0, b0q1, b0sl, b1q0 = 0.0, b1sl = 0.0;
  {
    b0q0 = qlat_prev[u0]; b0q1 = qlat_cur[u0]; b0sl = (b0q1 - b0q0) / dT10;
    if (nup > 1) { b1q0 = qlat_prev[u0 + 1]; const double q1 = qlat_cur[u0 + 1]; b1sl = (q1 - b1q0) / dT10; }
  }
  // reach series A (first non-headwater upstream in UREACHI order) and B (second)
  int ns = 0, uA = 0, uB = 0;
  if (nGood[u0] > 0) { uA = u0; ns = 1; }
  if (nup > 1 && nGood[u0 + 1] > 0) { if (ns == 0) uA = u0 + 1; else uB = u0 + 1; ++ns; }
  int nrA = 0, nrB = 0, kA = 1, kB = 1;
  double scA = 0.0, qbA = 0.0, tbA = 0.0, qeA = 0.0, teA = DBL_MAX, qnA = 0.0, tnA = 0.0;
  double scB = 0.0, qbB = 0.0, tbB = 0.0, qeB = 0.0, teB = DBL_MAX, qnB = 0.0, tnB = 0.0;
  if (ns > 0) {
    nrA = obN[uA]; scA = width[uA] / RW;
    qbA = obQ[uA]; tbA = obT[uA]; qeA = obQ[(size_t)N + uA]; teA = obT[(size_t)N + uA];
    if (nrA > 2) { qnA = obQ[(size_t)2 * N + uA]; tnA = obT[(size_t)2 * N + uA]; }
  }
  if (ns > 1) {
    nrB = obN[uB]; scB = width[uB] / RW;
    qbB = obQ[uB]; tbB = obT[uB]; qeB = obQ[(size_t)N + uB]; teB = obT[(size_t)N + uB];
    if (nrB > 2) { qnB = obQ[(size_t)2 * N + uB]; tnB = obT[(size_t)2 * N + uB]; }
  }
  if ((ns > 0 && nrA < 2) || (ns > 1 && nrB < 2)) return -40;   // upstream published nothing
  int IPRT = 0;
  double TIME_LAST = -DBL_MAX;
  for (;;) {
    // next routed particle of each series: indices 1 .. nr-2 (index nr-1 is the end-of-step particle)
    const double cA = (ns > 0 && kA <= nrA - 2) ? teA : DBL_MAX;
    const double cB = (ns > 1 && kB <= nrB - 2) ? teB : DBL_MAX;
    if (cA == DBL_MAX && cB == DBL_MAX) break;
    const bool pickA = cA <= cB;               // MINLOC: ties -> lower series index
    const double CT = pickA ? cA : cB;
    if (!(CT < T1)) return -40;                // a routed particle leaves before the end of the step
    if (CT < TIME_LAST) return -30;
    if (CT != TIME_LAST) {
      double Q_AGG = 0.0;
      Q_AGG = Q_AGG + (b0q0 + b0sl * (CT - T0)) * bsc;
      if (nup > 1) Q_AGG = Q_AGG + (b1q0 + b1sl * (CT - T0)) * bsc;
      {
        double SFLOW;
        if (pickA) SFLOW = qeA * scA;
        else {
          if (teA < CT || tbA > CT) return -40;
          const double SLOPE = (qeA - qbA) / (teA - tbA);
          SFLOW = (qbA + SLOPE * (CT - tbA)) * scA;
        }
        Q_AGG = Q_AGG + SFLOW;
      }
      if (ns > 1) {
        double SFLOW;
        if (!pickA) SFLOW = qeB * scB;
        else {
          if (teB < CT || tbB > CT) return -40;
          const double SLOPE = (qeB - qbB) / (teB - tbB);
          SFLOW = (qbB + SLOPE * (CT - tbB)) * scB;
        }
        Q_AGG = Q_AGG + SFLOW;
      }
      QD[IPRT] = Q_AGG; TD[IPRT] = CT; TIME_LAST = CT; ++IPRT;
    }
    if (pickA) {
      qbA = qeA; tbA = teA; qeA = qnA; teA = tnA; ++kA;
      if (kA + 1 <= nrA - 1) { qnA = obQ[(size_t)(kA + 1) * N + uA]; tnA = obT[(size_t)(kA + 1) * N + uA]; }
    } else {
      qbB = qeB; tbB = teB; qeB = qnB; teB = tnB; ++kB;
      if (kB + 1 <= nrB - 1) { qnB = obQ[(size_t)(kB + 1) * N + uB]; tnB = obT[(size_t)(kB + 1) * N + uB]; }
    }
  }
  {   // the particle at T1, led by the first basin series
    const double CT = T1;
    double Q_AGG = 0.0;
    Q_AGG = Q_AGG + b0q1 * bsc;
    if (nup > 1) Q_AGG = Q_AGG + (b1q0 + b1sl * (CT - T0)) * bsc;
    if (ns > 0) {
      if (teA < CT || tbA > CT) return -40;
      const double SLOPE = (qeA - qbA) / (teA - tbA);
      Q_AGG = Q_AGG + (qbA + SLOPE * (CT - tbA)) * scA;
    }
    if (ns > 1) {
      if (teB < CT || tbB > CT) return -40;
      const double SLOPE = (qeB - qbB) / (teB - tbB);
      Q_AGG = Q_AGG + (qbB + SLOPE * (CT - tbB)) * scB;
    }
    QD[IPRT] = Q_AGG; TD[IPRT] = CT; ++IPRT;
  }
  return IPRT;
}

// Confluences of more than two reaches are rare: their merge stays out of line with the series
// cursors in private memory and every particle fetched from the outbox on demand, so that the
// common (binary) path keeps a small register footprint.
__device__ __noinline__ int kwt_merge_generic(int nup, int u0, int NUPS, double RW, double T0, double T1,
                                              const uint8_t *nGood, const double *width, const double *qlat_prev,
                                              const double *qlat_cur, const int *obN, const double *obQ,
                                              const double *obT, int N, double *QD, double *TD, int IMAX) {
  int su[2 * MZR_MAXUP], slen[2 * MZR_MAXUP], snr[2 * MZR_MAXUP], ITIM[2 * MZR_MAXUP];
  double sc[2 * MZR_MAXUP], CTIME[2 * MZR_MAXUP];
  int IUPR = 0;
#pragma unroll 1
  for (int i = 0; i < nup; ++i) { su[i] = u0 + i; slen[i] = 2; snr[i] = 2; sc[i] = 1.0 / RW; ITIM[i] = 1; CTIME[i] = T1; }
#pragma unroll 1
  for (int i = 0; i < nup; ++i) {
    const int u = u0 + i;
    if (nGood[u] > 0) {
      const int si = nup + IUPR; ++IUPR;
      const int nr = obN[u];
      su[si] = u; snr[si] = nr; slen[si] = nr + 1; sc[si] = width[u] / RW; ITIM[si] = 1; CTIME[si] = obT[(size_t)N + u];
    }
  }
  auto sQ = [&](int i, int k) -> double { return i < nup ? (k == 0 ? qlat_prev[su[i]] : qlat_cur[su[i]]) : obQ[(size_t)k * N + su[i]]; };
  auto sT = [&](int i, int k) -> double { return i < nup ? (k == 0 ? T0 : T1) : obT[(size_t)k * N + su[i]]; };
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

#ifdef MZR_KWT_TIMING
#define TSTAMP(i) do { const long long _n = clock64(); if ((threadIdx.x & 63) == __ffsll(__ballot(1)) - 1) atomicAdd(&d.dbgCycles[i], (unsigned long long)(_n - _tprev)); _tprev = _n; } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
// particles of LDS work space per wavefront (3 x 8 B + 2 B each): 1024 (26 KB, 6 wavefronts per CU)
// when a launch has fewer wavefronts than the chip can hold, 768 (20 KB, 8 per CU) for larger domains

// One wavefront per block.  Each lane first works out how many work-array entries its reach needs
// (own particles + everything its upstreams routed), the wave carves the LDS pool with a prefix
// sum, and lanes that do not fit wait for the next round of the same wave.
// FULL = false compiles out lakes, water management and partition boundaries (the common case).
template <bool FULL, int KWT_POOL>
__global__ void __launch_bounds__(64) k_stage_kwt(MzrDev d, int s, int rBegin, int rEnd) {
  __shared__ double sQ[KWT_POOL], sT[KWT_POOL], sX[KWT_POOL];
  __shared__ unsigned short sL[KWT_POOL];
  const int r = rBegin + blockIdx.x * 64 + threadIdx.x;
  const int N = d.N;
  unsigned long long st_in = 0, st_up = 0, st_out = 0, st_head = 0, st_route = 0, st_edges = 0;
  int t = -1;
  const bool live = (r < rEnd) && ((t = s - d.sigma[r]) >= 0) && (t < d.W);
  const double T0 = d.t_start + (double)(t < 0 ? 0 : t) * d.dt;
  const double T1 = (d.W == 1) ? d.T1_single : T0 + d.dt;   // mzr_step passes TSEC(2) explicitly
  const double T_START = T0, T_END = T1;                    // RSTEP = 0
  double *Qrow = d.Q + (size_t)(t < 0 ? 0 : t) * N;
  const double *qlat_prev = d.qlat + (size_t)(t < 0 ? 0 : t) * N;       // BASIN_QR(0)
  const double *qlat_cur = d.qlat + (size_t)((t < 0 ? 0 : t) + 1) * N;  // BASIN_QR(1)
  const int par = t & 1;
  const int *obN = d.obN + (size_t)par * N;
  const double *obQ = d.obQ + (size_t)par * MZR_OB_CAP * N;
  const double *obT = d.obT + (size_t)par * MZR_OB_CAP * N;

#ifdef MZR_KWT_TIMING
  long long _tprev = clock64();
#endif
  int need = 0, nup = 0, u0 = 0, ng = 0, n_own = 0, NUPS = 0, IMAX = 0;
  double qlat_r = 0.0;
  bool halo = false;
  if (FULL && live && d.haloSlot) {   // tributary outlet computed in another partition: replay its imported record
    const int hs = d.haloSlot[r];
    if (hs >= 0) {
      halo = true;
      const size_t nH = d.nHalo;
      Qrow[r] = d.imQ[(size_t)t * nH + hs];
      const int n = d.imN[(size_t)t * nH + hs];
      d.obN[(size_t)par * N + r] = n;
      double *oq = d.obQ + (size_t)par * MZR_OB_CAP * N, *ot = d.obT + (size_t)par * MZR_OB_CAP * N;
      for (int k = 0; k <= n && n > 0; ++k) {
        oq[(size_t)k * N + r] = d.imOQ[((size_t)t * MZR_OB_CAP + k) * nH + hs];
        ot[(size_t)k * N + r] = d.imOT[((size_t)t * MZR_OB_CAP + k) * nH + hs];
      }
    }
  }
  bool lake = false, upLake = false;
  if (FULL && live && !halo && d.lakeSlot) {
    const int ls = d.lakeSlot[r];
    if (ls >= 0) {   // lake reach: lake_route replaces kwt_rch; it keeps one sentinel particle (init_model_data.f90:431-439)
      lake = true;
      double vol = d.vol[r], vol0 = vol, ele = d.ele[r], wb = 0.0, wmAct = 0.0;
      const double Q = mzr_lake::lake_route(d, r, t, ls, Qrow, qlat_cur[r], vol, vol0, ele, wb, wmAct);
      Qrow[r] = Q; d.vol[r] = vol; d.vol0[r] = vol0; d.ele[r] = ele; d.wb[r] = wb; d.qsum[r] += Q;
      if (d.kwN[r] != 1) { d.kwN[r] = 1; d.kwQ[r] = -9999.0; d.kwTI[r] = -9999.0; d.kwTR[r] = -9999.0; }
    }
  }
  if (live && !halo && !lake) {
    qlat_r = qlat_cur[r];
    ng = d.nGood[r];
    if (ng == 0) {   // headwater: kwt_route.f90:181-205
      Qrow[r] = qlat_r;
      d.qsum[r] += qlat_r;
      d.inflow[r] = 0.0;
      if (d.kwN[r] != 1) { d.kwN[r] = 1; d.kwQ[r] = -9999.0; d.kwTI[r] = -9999.0; d.kwTR[r] = -9999.0; }
      if (FULL && d.exportSlot && d.exportSlot[r] >= 0) d.exN[(size_t)t * d.nExp + d.exportSlot[r]] = 0;
      st_head = 1;
    } else {
      st_route = 1;
      nup = d.nUp[r]; u0 = d.upStart[r];
      st_edges = nup;
      n_own = d.kwN[r];
      int NUPR = 0;
      IMAX = nup;
      if (FULL && d.lakeSlot) for (int i = 0; i < nup; ++i) if (d.lakeSlot[u0 + i] >= 0) upLake = true;
      if (upLake && nup > 1) { mzr_raise(d, 10, r, t, 18); }   // lake outlet reach should have one upstream lake, :551-553
      for (int i = 0; i < nup && !upLake; ++i) {
        if (d.nGood[u0 + i] > 0) { ++NUPR; const int nr = obN[u0 + i]; IMAX += nr - 1; st_up += nr + 1; }
      }
      NUPS = nup + NUPR;
      const int NJ0 = n_own == 0 ? 0 : n_own - 1;
      need = NJ0 + 1 + ((NUPS == 1 || upLake) ? 1 : IMAX);
      if (upLake && nup > 1) need = 0;
      if (need > KWT_POOL) { mzr_raise(d, 60, r, t, 10); need = 0; }
    }
  }

  TSTAMP(0);
  bool pending = need > 0;
  while (__any(pending)) {
    // wave-wide inclusive prefix sum of the pending lanes' needs
    int incl = pending ? need : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += v; }
    const bool go = pending && incl <= KWT_POOL;
    if (go) {
      pending = false;
      const int off = incl - need;
      double *Qw = sQ + off, *Tw = sT + off, *Xw = sX + off;
      unsigned short *Lw = sL + off;
      do {
        const double RW = d.width[r];
        // ---- own particles (getusq_rch :598-608); element 0 = last routed particle
        const bool cold = (n_own == 0);
        const int NJ = cold ? 0 : n_own - 1;
        double X0 = cold ? 0.0 : d.kwTR[r];
        for (int k = 0; k < n_own; k += 4) {   // four particles per trip: eight loads in flight before the LDS writes
          double q[4], ti[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int kk = k + j < n_own ? k + j : n_own - 1;
            q[j] = d.kwQ[(size_t)kk * N + r]; ti[j] = d.kwTI[(size_t)kk * N + r];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < n_own) { Qw[k + j] = q[j]; Tw[k + j] = ti[j]; }
        }
        st_in = n_own;

        // ---- qexmul_rch
        int ND;
        if (upLake) {      // lake outflow enters the river as one particle, getusq_rch :554-559
          Qw[NJ + 1] = Qrow[u0] / RW; Tw[NJ + 1] = T1; ND = 1;
        } else if (NUPS == 1) {   // one upstream basin that is a headwater, :743-759
          Qw[NJ + 1] = qlat_cur[u0] / RW; Tw[NJ + 1] = T1; ND = 1;
        } else if (nup <= 2) {   // binary confluence
          ND = kwt_merge_binary(nup, u0, RW, T0, T1, d.nGood, d.width, qlat_prev, qlat_cur, obN, obQ, obT, N, Qw + NJ + 1, Tw + NJ + 1);
        } else {
          ND = kwt_merge_generic(nup, u0, NUPS, RW, T0, T1, d.nGood, d.width, qlat_prev, qlat_cur, obN, obQ, obT, N, Qw + NJ + 1, Tw + NJ + 1, IMAX);
        }
        TSTAMP(1);
        if (ND < 0) { mzr_raise(d, -ND, r, t, 11); break; }
        if (cold) {   // getusq_rch :587-596
          const double DT = T1 - T0;
          Qw[0] = Qw[1]; Tw[0] = T0 - DT - DT * 0; X0 = T0 - DT * 0;
        }
        int size = NJ + 1 + ND;

        {   // kwt_rch :163-174
          double mn = Qw[0];
          for (int k = 1; k < size; ++k) { const double q = Qw[k]; mn = q < mn ? q : mn; }
          if (mn < 0.0) { mzr_raise(d, 20, r, t, 12); break; }
          double q_up = 0.0;
          const uint32_t gm = d.goodMask[r];
          for (int i = 0; i < ng; ++i) { if (!((gm >> i) & 1u)) continue; q_up = q_up + Qrow[u0 + i]; }
          d.inflow[r] = q_up;
        }

        TSTAMP(2);
#ifdef MZR_KWT_TIMING
        {
          const unsigned long long bm = __ballot(size > MZR_MAXQPAR_DEV);
          if (size > MZR_MAXQPAR_DEV) { atomicAdd(&d.dbgCycles[8], 1ull); atomicAdd(&d.dbgCycles[9], (unsigned long long)(size - MZR_MAXQPAR_DEV)); atomicMax(&d.dbgCycles[10], (unsigned long long)size); atomicAdd(&d.dbgCycles[12], (unsigned long long)size); }
          if (bm && (int)(threadIdx.x & 63) == __ffsll(bm) - 1) { atomicAdd(&d.dbgCycles[11], 1ull); atomicAdd(&d.dbgCycles[13], (unsigned long long)__popcll(bm)); }
          atomicAdd(&d.dbgCycles[14], (unsigned long long)size); atomicAdd(&d.dbgCycles[15], 1ull);
        }
#endif
        // ---- remove_rch :999-1123: drop the particle with the least interpolation error until < MAXQPAR
        if (size > MZR_MAXQPAR_DEV) {
          const int NPRT = size - 1;
          for (int i = 0; i <= NPRT; ++i) Lw[i] = (unsigned short)(((i - 1) & 0xff) | ((i + 1) << 8));
          Xw[NPRT] = DBL_MAX; Xw[0] = DBL_MAX;
          {
            double qa = Qw[0], ta = Tw[0], qb2 = Qw[1], tb2 = Tw[1];
            for (int i = 1; i <= NPRT - 1; ++i) {
              const double qc = Qw[i + 1], tc = Tw[i + 1];
              Xw[i] = fabs(interp3(tb2, qa, qc, ta, tc) - qb2);
              qa = qb2; ta = tb2; qb2 = qc; tb2 = tc;
            }
          }
          int MPRT = NPRT;
          while (MPRT >= MZR_MAXQPAR_DEV) {
            int ISEL = 0; double emin = DBL_MAX;
            {   // first minimum of ABSERR over 1..NPRT (removed entries hold +Inf); four LDS reads in flight
              int i = 1;
              for (; i + 3 <= NPRT; i += 4) {
                const double e0 = Xw[i], e1 = Xw[i + 1], e2 = Xw[i + 2], e3 = Xw[i + 3];
                if (e0 < emin) { emin = e0; ISEL = i; }
                if (e1 < emin) { emin = e1; ISEL = i + 1; }
                if (e2 < emin) { emin = e2; ISEL = i + 2; }
                if (e3 < emin) { emin = e3; ISEL = i + 3; }
              }
              for (; i <= NPRT; ++i) { const double e = Xw[i]; if (e < emin) { emin = e; ISEL = i; } }
            }
            if (ISEL == 0) break;                         // no finite interpolation error left (NaN/Inf input)
            const unsigned short ls = Lw[ISEL];
            const int pm = ls & 0xff, pn = ls >> 8;     // INDEX1(ISEL-1), INDEX1(ISEL+1)
            const double qm = Qw[pm], tm = Tw[pm], qn = Qw[pn], tn = Tw[pn];
            if (pm > 0) {
              const int INEG = Lw[pm] & 0xff;
              Xw[pm] = fabs(interp3(tm, Qw[INEG], qn, Tw[INEG], tn) - qm);
            }
            if (pn < NPRT) {
              const int IPOS = Lw[pn] >> 8;
              Xw[pn] = fabs(interp3(tn, qm, Qw[IPOS], tm, Tw[IPOS]) - qn);
            }
            Xw[ISEL] = INFINITY;                        // removed: never the minimum again
            Lw[pm] = (unsigned short)((Lw[pm] & 0xff) | (pn << 8));
            Lw[pn] = (unsigned short)((Lw[pn] & 0xff00) | pm);
            --MPRT;
          }
          if (MPRT >= MZR_MAXQPAR_DEV) { mzr_raise(d, 62, r, t, 16); break; }
          int k = 0;
          for (int i = 0; i <= NPRT; i = Lw[i] >> 8) { Qw[k] = Qw[i]; Tw[k] = Tw[i]; ++k; }
          size = MPRT + 1;
        }
        // ---- extract_from_rch :351-455 (water abstraction / injection on the particles).  Its
        // recomputed exit times are overwritten by kinwav below, so only the flows change.
        if (FULL && d.is_flux_wm && d.wm) {
          const double Qtake = d.wm[(size_t)t * N + r];
          if (Qtake != -9999.0) {
            double Qavg;
            if (d_interp_rch(Tw, Qw, size, T_START, T_END, &Qavg)) { mzr_raise(d, 1, r, t, 17); break; }
            const double totQ = Qavg * RW;
            if (Qtake > 0.0) {
              const double Qfrac = Qtake / totQ;
              for (int i = 1; i < size; ++i) Qw[i] = Qw[i] * (1.0 + Qfrac);
            } else if (Qtake < 0.0 && fabs(Qtake) < totQ) {
              const double Qfrac = fabs(Qtake) / totQ;
              for (int i = 1; i < size; ++i) Qw[i] = Qw[i] * (1.0 - Qfrac);
            } else {
              const double mf = d.minflow[r];
              for (int i = 0; i < size; ++i) Qw[i] = mf;
            }
          }
        }
        const int NQ1 = size - 1;
        TSTAMP(3);

        // ---- kinwav_rch :1130-1439 on particles 1..NQ1, in place:
        //   Xw[i]   wave celerity of the group whose first particle is i   (WC)
        //   alive   bit i set while particle i still heads a group
        //   Xw[i+1] entry time of a merged group (T1 after :1335); flows of a merged group are the
        //           min / max over its members (:1329-1330), recomputed when needed
        int NQ2 = 0;
        {
          const double K = d.kwK[r];        // sqrt(R_SLOPE)/R_MAN_N           (host, once)
          const double cw = d.kwCW[r];      // ALFA*K**(1/ALFA), ALFA = 5/3     (host, once)
          const double XMX = d.length[r];
          const int NI = NQ1;
          for (int i = 1; i <= NI; ++i) Xw[i] = cw * pow_0p4(Qw[i]);
          TSTAMP(4);
          unsigned alive = NI >= 31 ? 0xfffffffeu : ((1u << (NI + 1)) - 2u);   // bits 1..NI
          auto nextHead = [&](int h) -> int {           // next group head after h, or NI+1
            const unsigned m = alive & ~((2u << h) - 1u);
            return m ? __ffs(m) - 1 : NI + 1;
          };
          auto groupT = [&](int h, int hn) -> double { return hn - h > 1 ? Xw[h + 1] : Tw[h]; };
          if (NI > 1) {
            double X = 0.0;
            for (;;) {
              double XB = XMX; int IXB = 0, JXBsel = 0;
              int jw = 1, iw = nextHead(1);
              double wcj = Xw[jw], tj = groupT(jw, iw);
              while (iw <= NI) {
                const int inx = nextHead(iw);
                const double wci = Xw[iw], ti = groupT(iw, inx);
                // earlier wave faster and later entry: XXB < 0 <= X (or WDIFF == 0) -> no break, no division
                if (!(wci == 0.0 || wcj == 0.0) && !(wcj > wci && ti > tj)) {
                  const double WDIFF = 1.0 / wcj - 1.0 / wci;
                  if (!(WDIFF == 0.0) && !(wci == wcj)) {
                    const double XXB = (ti - tj) / WDIFF;
                    if (!(XXB < X || XXB > XB)) { XB = XXB; IXB = iw; JXBsel = jw; }
                  }
                }
                jw = iw; wcj = wci; tj = ti; iw = inx;
              }
              if (XB == XMX) break;
              // merge group IXB into group JXB (:1325-1346)
              const int JXB = JXBsel;
              const int endI = nextHead(IXB);
              double q2 = Qw[JXB], q1 = Qw[JXB];
              for (int j = JXB + 1; j < endI; ++j) { const double q = Qw[j]; q2 = fmax(q2, q); q1 = fmin(q1, q); }
              const double A2 = pow_0p6(q2 / K);
              const double A1 = pow_0p6(q1 / K);
              const double CM = (q2 - q1) / (A2 - A1);
              const double tJ = groupT(JXB, IXB);
              const double wcJ = Xw[JXB];
              alive &= ~(1u << IXB);
              Xw[JXB + 1] = tJ + XB / wcJ - XB / CM;
              Xw[JXB] = CM;
              X = XB;
            }
          }
          TSTAMP(5);
          int ICOUNT = 0, bad = 0;
          double xprev = 0.0;
          auto rUpdate = [&](double QNEW, double TOLD, double TNEW) {   // :1409-1437
            ++ICOUNT;
            if (ICOUNT > NI) { bad = 60; return; }
            double te = TNEW;
            if (ICOUNT > 1) { if (te <= xprev) te = xprev + 1.0; }
            if (ICOUNT == 1 && te <= T_START) te = T_START + 1.0;
            Qw[ICOUNT] = QNEW; Tw[ICOUNT] = TOLD; Xw[ICOUNT] = te; xprev = te;
          };
          int h = 1;
          int hn = NI >= 1 ? nextHead(1) : NI + 1;
          double wc = NI >= 1 ? Xw[1] : 0.0, tg = NI >= 1 ? groupT(1, hn) : 0.0;
          double TEXIT = (NI >= 1 && !(wc < DBL_MIN)) ? fmin(XMX / wc + tg, DBL_MAX) : 0.0;
          while (h <= NI && !bad) {
            // look ahead to the next group before this group's slots are overwritten
            const int hnn = hn <= NI ? nextHead(hn) : NI + 1;
            const double wcn = hn <= NI ? Xw[hn] : 0.0;
            const double tgn = hn <= NI ? groupT(hn, hnn) : 0.0;
            if (wc < DBL_MIN) { bad = 20; break; }                       // zero flow :1365
            double TNEXT = DBL_MAX;                                      // = TEXIT of the next group (:1372)
            if (hn <= NI) TNEXT = fmin(XMX / wcn + tgn, DBL_MAX);
            double q1 = Qw[h], q2 = q1;
            for (int j = h + 1; j < hn; ++j) { const double q = Qw[j]; q2 = fmax(q2, q); q1 = fmin(q1, q); }
            if (q1 != q2) {
              if (TEXIT < T_END) {
                const double TEXIT2 = fmin(TEXIT + 1.0, TEXIT + 0.5 * (fmin(TNEXT, T_END) - TEXIT));
                if (TEXIT2 == TEXIT) { bad = 30; break; }
                rUpdate(q1, tg, TEXIT);
                if (bad) break;
                rUpdate(q2, tg, TEXIT2);
              } else {
                for (int J = h; J < hn && !bad; ++J) rUpdate(Qw[J], Tw[J], TEXIT);
              }
            } else {
              rUpdate(q1, tg, TEXIT);
            }
            h = hn; hn = hnn; wc = wcn; tg = tgn; TEXIT = TNEXT;
          }
          if (bad) { mzr_raise(d, bad, r, t, 13); break; }
          NQ2 = ICOUNT;
        }

        TSTAMP(6);
        // ---- time-step average and housekeeping, kwt_rch :257-311
        Xw[0] = X0;
        int NR = 0;
        for (int i = 1; i <= NQ2; ++i) NR += Xw[i] < T_END ? 1 : 0;   // count(FROUTE)-1
        if (NR + 1 > NQ2) { mzr_raise(d, 61, r, t, 14); break; }      // no waiting particle left
        double QNEW;
        if (d_interp_rch(Xw, Qw, NR + 2, T_START, T_END, &QNEW)) { mzr_raise(d, 1, r, t, 15); break; }
        const double Qout = QNEW * RW + qlat_r;
        Qrow[r] = Qout;
        d.qsum[r] += Qout;
        const double qN = Qw[NR], qN1 = Qw[NR + 1], xN = Xw[NR], xN1 = Xw[NR + 1], tN = Tw[NR], tN1 = Tw[NR + 1];
        const double dTx = xN1 - xN;
        const double Q_END = qN + ((qN1 - qN) / dTx) * (T_END - xN);
        const double TIMEI = tN + ((tN1 - tN) / dTx) * (T_END - xN);
        const int NN2 = NQ2 - NR;
        // tributary outlet of a partition: the same record goes to the time-indexed export buffer
        const int es = (FULL && d.exportSlot) ? d.exportSlot[r] : -1;
        if (es >= 0) {
          const size_t nE = d.nExp;
          d.exN[(size_t)t * nE + es] = NR + 2;
          double *eq = d.exOQ + (size_t)t * MZR_OB_CAP * nE, *et = d.exOT + (size_t)t * MZR_OB_CAP * nE;
          for (int k = 0; k <= NR; ++k) { eq[(size_t)k * nE + es] = Qw[k]; et[(size_t)k * nE + es] = Xw[k]; }
          eq[(size_t)(NR + 1) * nE + es] = Q_END; et[(size_t)(NR + 1) * nE + es] = T_END;
          eq[(size_t)(NR + 2) * nE + es] = qN1;   et[(size_t)(NR + 2) * nE + es] = xN1;
        }
        // outbox for the downstream reach: KWAVE(0:NR+1) + first waiting particle (flow, exit time)
        if (!d.isOutlet[r]) {
          int *obNw = d.obN + (size_t)par * N;
          double *obQw = d.obQ + (size_t)par * MZR_OB_CAP * N;
          double *obTw = d.obT + (size_t)par * MZR_OB_CAP * N;
          obNw[r] = NR + 2;
          for (int k = 0; k <= NR; ++k) { obQw[(size_t)k * N + r] = Qw[k]; obTw[(size_t)k * N + r] = Xw[k]; }
          obQw[(size_t)(NR + 1) * N + r] = Q_END; obTw[(size_t)(NR + 1) * N + r] = T_END;
          obQw[(size_t)(NR + 2) * N + r] = qN1;   obTw[(size_t)(NR + 2) * N + r] = xN1;
        }
        // at-rest state: KWAVE(NR+1:NQ2+1)
        d.kwN[r] = NN2 + 1;
        d.kwQ[r] = Q_END; d.kwTI[r] = TIMEI; d.kwTR[r] = T_END;
        for (int j = 1; j <= NN2; ++j) {
          d.kwQ[(size_t)j * N + r] = Qw[NR + j]; d.kwTI[(size_t)j * N + r] = Tw[NR + j]; d.kwTR[(size_t)j * N + r] = Xw[NR + j];
        }
        st_out = NQ2 + 2;
        TSTAMP(7);
      } while (0);
    }
  }
  if (d.kwtStat) {
    const unsigned long long a = wave_sum(st_in), b = wave_sum(st_up), c = wave_sum(st_out);
    const unsigned long long e = wave_sum(st_head), f = wave_sum(st_route), g = wave_sum(st_edges);
    if ((threadIdx.x & 63) == 0 && (e | f)) {
      atomicAdd(&d.kwtStat->w_in, a); atomicAdd(&d.kwtStat->w_up, b); atomicAdd(&d.kwtStat->w_out, c);
      atomicAdd(&d.kwtStat->n_head, e); atomicAdd(&d.kwtStat->n_route, f); atomicAdd(&d.kwtStat->n_edges, g);
    }
  }
}

void mzr_launch_stage_kwt(const MzrDev &d, int wk, int s, int rBegin, int rEnd, hipStream_t stream) {
  (void)wk;
  const int n = rEnd - rBegin;
  if (n <= 0) return;
  dim3 block(64), grid((n + 63) / 64);
  const bool full = d.lakeSlot || d.haloSlot || d.exportSlot || (d.is_flux_wm && d.wm);
  const bool big = d.N > 250000;      // more wavefronts per launch than 6 per CU can hold at once
  if (full) hipLaunchKernelGGL((k_stage_kwt<true, 1024>), grid, block, 0, stream, d, s, rBegin, rEnd);
  else if (big) hipLaunchKernelGGL((k_stage_kwt<false, 768>), grid, block, 0, stream, d, s, rBegin, rEnd);
  else hipLaunchKernelGGL((k_stage_kwt<false, 1024>), grid, block, 0, stream, d, s, rBegin, rEnd);
}
