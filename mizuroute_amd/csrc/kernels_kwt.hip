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
//  * Work arrays (own particles + merged upstream particles, up to WK) live in private memory.
//
// Bound by HBM traffic of the particle rows: see DESIGN.md for the bytes-per-reach-step model.
#include <float.h>
#include "mzr_device.h"

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

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace

template <int WK>
__global__ void __launch_bounds__(256) k_stage_kwt(MzrDev d, int s, int rBegin, int rEnd) {
  const int r = rBegin + blockIdx.x * blockDim.x + threadIdx.x;
  const int N = d.N;
  unsigned long long st_in = 0, st_up = 0, st_out = 0, st_head = 0, st_route = 0, st_edges = 0;
  int t = -1;
  const bool live = (r < rEnd) && ((t = s - d.sigma[r]) >= 0) && (t < d.W);
  if (live) {
    const double T0 = d.t_start + (double)t * d.dt;
    const double T1 = (d.W == 1) ? d.T1_single : T0 + d.dt;   // mzr_step passes TSEC(2) explicitly
    const double T_START = T0, T_END = T1;                 // RSTEP = 0
    double *Qrow = d.Q + (size_t)t * N;
    const double *qlat_prev = d.qlat + (size_t)t * N;      // BASIN_QR(0)
    const double *qlat_cur = d.qlat + (size_t)(t + 1) * N; // BASIN_QR(1)
    const double qlat_r = qlat_cur[r];
    const int ng = d.nGood[r];
    do {
      if (ng == 0) {   // headwater: kwt_route.f90:181-205
        Qrow[r] = qlat_r;
        d.qsum[r] += qlat_r;
        d.inflow[r] = 0.0;
        if (d.kwN[r] != 1) {   // single sentinel particle; static afterwards
          d.kwN[r] = 1; d.kwQ[r] = -9999.0; d.kwTI[r] = -9999.0; d.kwTR[r] = -9999.0;
        }
        st_head = 1;
        break;
      }
      st_route = 1;
      const int nup = d.nUp[r];
      const int u0 = d.upStart[r];
      const double RW = d.width[r];
      const int par = t & 1;
      const int *obN = d.obN + (size_t)par * N;
      const double *obQ = d.obQ + (size_t)par * MZR_OB_CAP * N;
      const double *obT = d.obT + (size_t)par * MZR_OB_CAP * N;
      st_edges = nup;

      double Qw[WK], Tw[WK], Xw[WK];   // Q_JRCH, TENTRY, T_EXIT (Xw doubles as ABSERR in remove)

      // ---- own particles (getusq_rch :598-608); element 0 = last routed particle
      const int n_own = d.kwN[r];
      const bool cold = (n_own == 0);
      const int NJ = cold ? 0 : n_own - 1;
      for (int k = 0; k < n_own; ++k) { Qw[k] = d.kwQ[(size_t)k * N + r]; Tw[k] = d.kwTI[(size_t)k * N + r]; }
      if (!cold) Xw[0] = d.kwTR[r];
      st_in = n_own;

      // ---- qexmul_rch: merge upstream series into (QD,TD) = Qw/Tw[NJ+1 ...]
      int ND = 0;
      {
        int NUPR = 0;
        for (int i = 0; i < nup; ++i) NUPR += d.nGood[u0 + i] > 0 ? 1 : 0;
        const int NUPS = nup + NUPR;
        if (NUPS == 1) {   // one upstream basin that is a headwater, :743-759
          Qw[NJ + 1] = qlat_cur[u0] / RW;
          Tw[NJ + 1] = T1;
          ND = 1;
        } else {
          int su[2 * MZR_MAXUP], slen[2 * MZR_MAXUP], snr[2 * MZR_MAXUP], ITIM[2 * MZR_MAXUP];
          double sc[2 * MZR_MAXUP], CTIME[2 * MZR_MAXUP];
          int IMAX = nup, IUPR = 0;
          for (int i = 0; i < nup; ++i) {            // basins :771-787
            su[i] = u0 + i; slen[i] = 2; snr[i] = 2; sc[i] = 1.0 / RW; ITIM[i] = 1; CTIME[i] = T1;
          }
          for (int i = 0; i < nup; ++i) {            // reaches :792-858
            const int u = u0 + i;
            if (d.nGood[u] > 0) {
              const int si = nup + IUPR; ++IUPR;
              const int nr = obN[u];                 // count(RF) = NR_u + 2
              su[si] = u; snr[si] = nr; slen[si] = nr + 1;   // NQ = min(NR+1, NS) = NR+1 (one waiting particle always exists)
              sc[si] = d.width[u] / RW; ITIM[si] = 1; CTIME[si] = obT[(size_t)N + u];
              IMAX += nr - 1;
              st_up += nr + 1;
            }
          }
          if (NJ + 1 + IMAX > WK) { mzr_raise(d, 60, r, t, 10); break; }
          // element k of series i: flow / exit time
          auto sQ = [&](int i, int k) -> double {
            return i < nup ? (k == 0 ? qlat_prev[su[i]] : qlat_cur[su[i]]) : obQ[(size_t)k * N + su[i]];
          };
          auto sT = [&](int i, int k) -> double {
            return i < nup ? (k == 0 ? T0 : T1) : obT[(size_t)k * N + su[i]];
          };
          unsigned done = 0;                         // MFLG bits
          const unsigned all = (1u << NUPS) - 1u;
          int IPRT = 0, JUPS_OLD = 0x7fffffff, ITIM_OLD = 0x7fffffff, bad = 0;
          double TIME_LAST = -DBL_MAX;
          for (;;) {
            int JUPS = 0;                            // MINLOC(CTIME): first minimum
            for (int i = 1; i < NUPS; ++i) if (CTIME[i] < CTIME[JUPS]) JUPS = i;
            if (JUPS == JUPS_OLD && ITIM[JUPS] == ITIM_OLD) { bad = 20; break; }   // :901-903
            JUPS_OLD = JUPS; ITIM_OLD = ITIM[JUPS];
            if (!((done >> JUPS) & 1u)) {
              const int kj = ITIM[JUPS];
              if (kj >= snr[JUPS]) {                 // particle not routed: series finished :910-912
                done |= 1u << JUPS; CTIME[JUPS] = DBL_MAX;
              } else {
                const double CT = CTIME[JUPS];
                const double TIME_OLD = IPRT >= 1 ? TIME_LAST : -DBL_MAX;
                if (CT < TIME_OLD) { bad = 30; break; }
                if (CT != TIME_OLD) {
                  double Q_AGG = 0.0;
                  for (int i = 0; i < NUPS; ++i) {
                    const int IWAV = ITIM[i];
                    double SFLOW;
                    if (i == JUPS) {
                      SFLOW = sQ(i, IWAV) * sc[i];
                    } else {
                      int IBEG = IWAV;
                      if (sT(i, IBEG) >= CT) IBEG = IWAV - 1;
                      const int IEND = IBEG + 1;
                      const double tb = sT(i, IBEG), te = sT(i, IEND);
                      if (IEND >= slen[i] || IBEG < 0 || te < CT || tb > CT) { bad = 40; break; }
                      const double qb = sQ(i, IBEG), qe = sQ(i, IEND);
                      const double SLOPE = (qe - qb) / (te - tb);
                      const double PREDV = qb + SLOPE * (CT - tb);
                      SFLOW = PREDV * sc[i];
                    }
                    Q_AGG = Q_AGG + SFLOW;
                  }
                  if (bad) break;
                  if (IPRT >= IMAX) { bad = 60; break; }
                  Qw[NJ + 1 + IPRT] = Q_AGG; Tw[NJ + 1 + IPRT] = CT; TIME_LAST = CT; ++IPRT;
                }
                if (kj == slen[JUPS] - 1) { done |= 1u << JUPS; CTIME[JUPS] = DBL_MAX; }
                else { ITIM[JUPS] = kj + 1; CTIME[JUPS] = sT(JUPS, kj + 1); }
              }
            }
            if (done == all) break;
          }
          if (bad) { mzr_raise(d, bad, r, t, 11); break; }
          ND = IPRT;
        }
      }
      if (cold) {   // getusq_rch :587-596
        const double DT = T1 - T0;
        Qw[0] = Qw[1]; Tw[0] = T0 - DT - DT * 0; Xw[0] = T0 - DT * 0;
      }
      int size = NJ + 1 + ND;

      {   // kwt_rch :163-174
        double mn = Qw[0];
        for (int k = 1; k < size; ++k) mn = Qw[k] < mn ? Qw[k] : mn;
        if (mn < 0.0) { mzr_raise(d, 20, r, t, 12); break; }
        double q_up = 0.0;
        const uint32_t gm = d.goodMask[r];
        for (int i = 0; i < ng; ++i) { if (!((gm >> i) & 1u)) continue; q_up = q_up + Qrow[u0 + i]; }
        d.inflow[r] = q_up;
      }

      // ---- remove_rch :999-1123: drop the particle with the least interpolation error until < MAXQPAR
      if (size > MZR_MAXQPAR_DEV) {
        const int NPRT = size - 1;
        uint8_t prv[WK], nxt[WK];
        for (int i = 0; i <= NPRT; ++i) { prv[i] = (uint8_t)(i - 1); nxt[i] = (uint8_t)(i + 1); }
        Xw[NPRT] = DBL_MAX;
        const double X0keep = Xw[0];
        Xw[0] = DBL_MAX;
        for (int i = 1; i <= NPRT - 1; ++i)
          Xw[i] = fabs(interp3(Tw[i], Qw[i - 1], Qw[i + 1], Tw[i - 1], Tw[i + 1]) - Qw[i]);
        int MPRT = NPRT;
        while (MPRT >= MZR_MAXQPAR_DEV) {
          int ISEL = 0; double emin = Xw[0];
          for (int i = 1; i <= NPRT; ++i) { const double e = Xw[i]; if (e < emin) { emin = e; ISEL = i; } }
          const int pm = prv[ISEL], pn = nxt[ISEL];       // INDEX1(ISEL-1), INDEX1(ISEL+1)
          if (pm > 0) {
            const int INEG = prv[pm];
            Xw[pm] = fabs(interp3(Tw[pm], Qw[INEG], Qw[pn], Tw[INEG], Tw[pn]) - Qw[pm]);
          }
          if (pn < NPRT) {
            const int IPOS = nxt[pn];
            Xw[pn] = fabs(interp3(Tw[pn], Qw[pm], Qw[IPOS], Tw[pm], Tw[IPOS]) - Qw[pn]);
          }
          Xw[ISEL] = INFINITY;                           // removed: never the minimum again
          nxt[pm] = (uint8_t)pn; prv[pn] = (uint8_t)pm;
          --MPRT;
        }
        int k = 0;
        for (int i = 0; i <= NPRT; i = nxt[i]) { Qw[k] = Qw[i]; Tw[k] = Tw[i]; ++k; }
        size = MPRT + 1;
        Xw[0] = X0keep;
      }
      const int NQ1 = size - 1;

      // ---- kinwav_rch :1130-1439 on particles 1..NQ1 (Q0/T0 = Qw/Tw in place)
      int NQ2 = 0;
      {
        const double ALFA = 5.0 / 3.0;
        const double K = sqrt(d.slope[r]) / d.mann[r];
        const double XMX = d.length[r];
        constexpr int KC = MZR_MAXQPAR_DEV + 1;
        double Q1[KC], Q2[KC], TT[KC], WC[KC];
        uint8_t IX[KC], MF[KC];
        int NN = NQ1;
        const int NI = NQ1;
        const double e1 = 1.0 / ALFA, e2 = (ALFA - 1.0) / ALFA;
        const double cw = ALFA * pow(K, e1);
        for (int i = 1; i <= NI; ++i) {
          MF[i] = (uint8_t)i; IX[i] = (uint8_t)i;
          Q1[i] = Q2[i] = Qw[i]; TT[i] = Tw[i];
          WC[i] = cw * pow(Qw[i], e2);
        }
        if (NN > 1) {
          double X = 0.0;
          for (;;) {
            double XB = XMX; int IXB = 0;
            for (int IW = 2; IW <= NN; ++IW) {
              const int JW = IW - 1;
              if (WC[IW] == 0.0 || WC[JW] == 0.0) continue;
              const double WDIFF = 1.0 / WC[JW] - 1.0 / WC[IW];
              if (WDIFF == 0.0) continue;
              if (WC[IW] == WC[JW]) continue;
              const double XXB = (TT[IW] - TT[JW]) / WDIFF;
              if (XXB < X || XXB > XB) continue;
              XB = XXB; IXB = IW;
            }
            if (XB == XMX) break;
            NN = NN - 1;
            const int JXB = IXB - 1;
            Q2[JXB] = fmax(Q2[JXB], Q2[IXB]);
            Q1[JXB] = fmin(Q1[JXB], Q1[IXB]);
            const double A2 = pow(Q2[JXB] / K, 1.0 / ALFA);
            const double A1 = pow(Q1[JXB] / K, 1.0 / ALFA);
            const double CM = (Q2[JXB] - Q1[JXB]) / (A2 - A1);
            TT[JXB] = TT[JXB] + XB / WC[JXB] - XB / CM;
            WC[JXB] = CM;
            for (int i = IX[IXB]; i <= NI; ++i) MF[i] = (uint8_t)(MF[i] - 1);
            for (int i = IXB; i <= NN; ++i) { IX[i] = IX[i + 1]; TT[i] = TT[i + 1]; WC[i] = WC[i + 1]; Q1[i] = Q1[i + 1]; Q2[i] = Q2[i + 1]; }
            X = XB;
          }
        }
        int ICOUNT = 0, bad = 0;
        auto rUpdate = [&](double QNEW, double TOLD, double TNEW) {   // :1409-1437
          ++ICOUNT;
          if (ICOUNT > NI) { bad = 60; return; }
          Qw[ICOUNT] = QNEW; Tw[ICOUNT] = TOLD;
          double te = TNEW;
          if (ICOUNT > 1) { if (te <= Xw[ICOUNT - 1]) te = Xw[ICOUNT - 1] + 1.0; }
          if (ICOUNT == 1 && te <= T_START) te = T_START + 1.0;
          Xw[ICOUNT] = te;
        };
        for (int IROUTE = 1; IROUTE <= NN && !bad; ++IROUTE) {
          if (WC[IROUTE] < DBL_MIN) { bad = 20; break; }                       // zero flow :1365
          const double TEXIT = fmin(XMX / WC[IROUTE] + TT[IROUTE], DBL_MAX);
          double TNEXT = DBL_MAX;
          if (IROUTE < NN) TNEXT = fmin(XMX / WC[IROUTE + 1] + TT[IROUTE + 1], DBL_MAX);
          if (Q1[IROUTE] != Q2[IROUTE]) {
            if (TEXIT < T_END) {
              const double TEXIT2 = fmin(TEXIT + 1.0, TEXIT + 0.5 * (fmin(TNEXT, T_END) - TEXIT));
              if (TEXIT2 == TEXIT) { bad = 30; break; }
              rUpdate(Q1[IROUTE], TT[IROUTE], TEXIT);
              if (bad) break;
              rUpdate(Q2[IROUTE], TT[IROUTE], TEXIT2);
            } else {
              for (int J = 1; J <= NI && !bad; ++J) if (MF[J] == IROUTE) rUpdate(Qw[J], Tw[J], TEXIT);
            }
          } else {
            rUpdate(Q1[IROUTE], TT[IROUTE], TEXIT);
          }
        }
        if (bad) { mzr_raise(d, bad, r, t, 13); break; }
        NQ2 = ICOUNT;
      }

      // ---- time-step average and housekeeping, kwt_rch :257-311
      int NR = 0;
      for (int i = 1; i <= NQ2; ++i) NR += Xw[i] < T_END ? 1 : 0;   // count(FROUTE)-1
      if (NR + 1 > NQ2) { mzr_raise(d, 61, r, t, 14); break; }      // no waiting particle left
      double QNEW;
      if (d_interp_rch(Xw, Qw, NR + 2, T_START, T_END, &QNEW)) { mzr_raise(d, 1, r, t, 15); break; }
      const double Qout = QNEW * RW + qlat_r;
      Qrow[r] = Qout;
      d.qsum[r] += Qout;
      const double dTx = Xw[NR + 1] - Xw[NR];
      const double Q_END = Qw[NR] + ((Qw[NR + 1] - Qw[NR]) / dTx) * (T_END - Xw[NR]);
      const double TIMEI = Tw[NR] + ((Tw[NR + 1] - Tw[NR]) / dTx) * (T_END - Xw[NR]);
      const int NN2 = NQ2 - NR;
      // outbox for the downstream reach: KWAVE(0:NR+1) + first waiting particle (flow, exit time)
      if (!d.isOutlet[r]) {
        int *obNw = d.obN + (size_t)par * N;
        double *obQw = d.obQ + (size_t)par * MZR_OB_CAP * N;
        double *obTw = d.obT + (size_t)par * MZR_OB_CAP * N;
        obNw[r] = NR + 2;
        for (int k = 0; k <= NR; ++k) { obQw[(size_t)k * N + r] = Qw[k]; obTw[(size_t)k * N + r] = Xw[k]; }
        obQw[(size_t)(NR + 1) * N + r] = Q_END;      obTw[(size_t)(NR + 1) * N + r] = T_END;
        obQw[(size_t)(NR + 2) * N + r] = Qw[NR + 1]; obTw[(size_t)(NR + 2) * N + r] = Xw[NR + 1];
      }
      // at-rest state: KWAVE(NR+1:NQ2+1)
      d.kwN[r] = NN2 + 1;
      d.kwQ[r] = Q_END; d.kwTI[r] = TIMEI; d.kwTR[r] = T_END;
      for (int j = 1; j <= NN2; ++j) {
        d.kwQ[(size_t)j * N + r] = Qw[NR + j]; d.kwTI[(size_t)j * N + r] = Tw[NR + j]; d.kwTR[(size_t)j * N + r] = Xw[NR + j];
      }
      st_out = NQ2 + 2;
    } while (0);
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
  const int n = rEnd - rBegin;
  if (n <= 0) return;
  dim3 block(256), grid((n + 255) / 256);
  if (wk <= 64) hipLaunchKernelGGL(k_stage_kwt<64>, grid, block, 0, stream, d, s, rBegin, rEnd);
  else if (wk <= 128) hipLaunchKernelGGL(k_stage_kwt<128>, grid, block, 0, stream, d, s, rBegin, rEnd);
  else hipLaunchKernelGGL(k_stage_kwt<192>, grid, block, 0, stream, d, s, rBegin, rEnd);
}
