/* CPU oracle: Lagrangian kinematic-wave tracking, kwt_route.f90 (test infrastructure; see
   mzr_oracle.h).  Follows the Fortran routine by routine, including the in-place truncation of the
   upstream reach's particle list (kwt_route.f90:822-848). */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

#define KW(o, r) ((o)->kw + (size_t)(r) * ORC_KWSTORE)

static int fail(orc_t *o, int code, const char *msg) {
  snprintf(o->msg, sizeof o->msg, "%s", msg);
  return code;
}

/* kwt_route.f90:1444-1622 with NNEW = 2 (one averaging interval) */
static int interp_rch(const double *TOLD, const double *QOLD, int NOLD, double T0, double T1, double *QNEW) {
  /* arrays are 1-based in the Fortran; here TOLD[0..NOLD-1] */
  if (TOLD[0] > T0 || TOLD[NOLD - 1] < T1) return 1;   /* bad bounds */
  int IBEG = 1, IEND = 1;
  for (int i = 2; i <= NOLD; i++) { if (T0 <= TOLD[i - 1]) { IBEG = i; break; } }
  for (int i = 1; i <= NOLD; i++) { if (T1 <= TOLD[i - 1]) { IEND = i; break; } }
  double AREAB = 0.0, AREAE = 0.0, AREAM = 0.0, SLOPE, QEST0, QEST1;
#define T(i) TOLD[(i) - 1]
#define Q(i) QOLD[(i) - 1]
  if (T1 < T(IBEG)) {
    SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    QEST1 = SLOPE * (T1 - T(IBEG - 1)) + Q(IBEG - 1);
    *QNEW = 0.5 * (QEST0 + QEST1);
    return 0;
  }
  if (T0 < T(IBEG)) {
    SLOPE = (Q(IBEG) - Q(IBEG - 1)) / (T(IBEG) - T(IBEG - 1));
    QEST0 = SLOPE * (T0 - T(IBEG - 1)) + Q(IBEG - 1);
    AREAB = (T(IBEG) - T0) * 0.5 * (QEST0 + Q(IBEG));
  }
  if (T1 < T(IEND)) {
    SLOPE = (Q(IEND) - Q(IEND - 1)) / (T(IEND) - T(IEND - 1));
    QEST1 = SLOPE * (T1 - T(IEND - 1)) + Q(IEND - 1);
    AREAE = (T1 - T(IEND - 1)) * 0.5 * (Q(IEND - 1) + QEST1);
  }
  if (IBEG < IEND) {
    for (int IMID = IBEG + 1; IMID <= IEND; IMID++) {
      if (IMID < IEND || (IMID == IEND && T1 == T(IEND) && T0 < T(IEND - 1)))
        AREAM = AREAM + (T(IMID) - T(IMID - 1)) * 0.5 * (Q(IMID - 1) + Q(IMID));
    }
  }
#undef T
#undef Q
  *QNEW = (AREAB + AREAE + AREAM) / (T1 - T0);
  return 0;
}

/* kwt_route.f90:619-993 */
static int qexmul_rch(orc_t *o, int JRCH, double T0, double T1, int *ND_out, double **QD_out, double **TD_out) {
  const double DT = T1 - T0;
  (void)DT;
  int NUPB = o->upOff[JRCH + 1] - o->upOff[JRCH];
  const int *ups = o->upIdx + o->upOff[JRCH];
  int NUPR = 0;
  for (int i = 0; i < NUPB; i++) if (o->nGood[ups[i]] > 0) NUPR++;
  int NUPS = NUPB + NUPR;
  double RW = o->par[ORC_P_WIDTH][JRCH];
  o->n_edges += NUPB;
  if (NUPB > 2 && NUPS > 1) o->paths[7]++;

  if (NUPS == 1) {   /* one upstream basin that is a headwater, :743-759 */
    double *QD = (double *)malloc(sizeof(double)), *TD = (double *)malloc(sizeof(double));
    int IR = ups[0];
    QD[0] = o->BASIN_QR1[IR] / RW;
    TD[0] = T1;
    *ND_out = 1; *QD_out = QD; *TD_out = TD;
    return 0;
  }

  orc_fpoint **US = (orc_fpoint **)calloc(NUPS, sizeof(orc_fpoint *));
  int *USN = (int *)calloc(NUPS, sizeof(int));
  double *UWIDTH = (double *)calloc(NUPS, sizeof(double));
  double *CTIME = (double *)calloc(NUPS, sizeof(double));
  int IMAX = NUPB;
  /* (2) basins, :771-787 */
  for (int i = 0; i < NUPB; i++) {
    int IR = ups[i];
    US[i] = (orc_fpoint *)calloc(2, sizeof(orc_fpoint)); USN[i] = 2;
    US[i][0].QF = o->BASIN_QR0[IR]; US[i][1].QF = o->BASIN_QR1[IR];
    US[i][0].TI = T0; US[i][1].TI = T1;
    US[i][0].TR = T0; US[i][1].TR = T1;
    US[i][0].RF = 1;  US[i][1].RF = 1;
    UWIDTH[i] = 1.0;
    CTIME[i] = US[i][1].TR;
  }
  /* (3) reaches, :792-858 */
  int IUPR = 0, ierr = 0;
  for (int i = 0; i < NUPB; i++) {
    int IR = ups[i];
    if (o->nGood[IR] > 0) {
      IUPR++;
      int s = NUPB + IUPR - 1;
      if (o->nkw[IR] < 0) { ierr = fail(o, 20, "qexmul_rch/RCHSTA_out%LKW_ROUTE%KWAVE is not associated"); goto done; }
      int NS = o->nkw[IR];
      orc_fpoint *K = KW(o, IR);
      int NR = 0;
      for (int k = 0; k < NS; k++) NR += K[k].RF ? 1 : 0;
      int NQ = NR + 1 < NS ? NR + 1 : NS;
      US[s] = (orc_fpoint *)calloc(NQ, sizeof(orc_fpoint)); USN[s] = NQ;
      memcpy(US[s], K, NQ * sizeof(orc_fpoint));
      o->w_up += NQ;
      /* remove the routed particles from the upstream reach: KWAVE(0:NS-NR) = OLD(NR-1:NS-1) */
      memmove(K, K + (NR - 1), (size_t)(NS - NR + 1) * sizeof(orc_fpoint));
      o->nkw[IR] = NS - NR + 1;
      UWIDTH[s] = o->par[ORC_P_WIDTH][IR];
      CTIME[s] = US[s][1].TR;
      IMAX = IMAX + (NR - 1);
    }
  }
  /* (4) merge, :880-976 */
  {
    double *QD = (double *)malloc((IMAX > 0 ? IMAX : 1) * sizeof(double));
    double *TD = (double *)malloc((IMAX > 0 ? IMAX : 1) * sizeof(double));
    int *MFLG = (int *)calloc(NUPS, sizeof(int));
    int *ITIM = (int *)calloc(NUPS, sizeof(int));
    for (int i = 0; i < NUPS; i++) ITIM[i] = 1;
    int IPRT = 0, JUPS_OLD = 2147483647, ITIM_OLD = 2147483647;
    for (;;) {
      int JUPS = 0;   /* MINLOC: first index of the minimum */
      for (int i = 1; i < NUPS; i++) if (CTIME[i] < CTIME[JUPS]) JUPS = i;
      if (JUPS == JUPS_OLD && ITIM[JUPS] == ITIM_OLD) { ierr = fail(o, 20, "qexmul_rch/stuck in the continuous do-loop"); }
      if (ierr) { free(MFLG); free(ITIM); free(QD); free(TD); goto done; }
      JUPS_OLD = JUPS; ITIM_OLD = ITIM[JUPS];
      if (!MFLG[JUPS]) {
        if (!US[JUPS][ITIM[JUPS]].RF) {
          MFLG[JUPS] = 1;
          CTIME[JUPS] = DBL_MAX;
        } else {
          double TIME_OLD = IPRT >= 1 ? TD[IPRT - 1] : -DBL_MAX;
          if (CTIME[JUPS] < TIME_OLD) { ierr = fail(o, 30, "qexmul_rch/expect process in order of time"); }
          if (!ierr && CTIME[JUPS] == TIME_OLD && CTIME[JUPS] < T1) o->paths[4]++;   /* not the common end-of-step time */
          if (!ierr && CTIME[JUPS] != TIME_OLD) {
            double Q_AGG = 0.0;
            for (int i = 0; i < NUPS; i++) {
              int IWAV = ITIM[i];
              double SCFAC = UWIDTH[i] / RW;
              double SFLOW;
              if (i == JUPS) {
                SFLOW = US[i][IWAV].QF * SCFAC;
              } else {
                int IBEG = IWAV;
                if (US[i][IBEG].TR >= CTIME[JUPS]) IBEG = IWAV - 1;
                int IEND = IBEG + 1;
                if (IEND >= USN[i] || IBEG < 0 ||
                    US[i][IEND].TR < CTIME[JUPS] || US[i][IBEG].TR > CTIME[JUPS]) {
                  ierr = fail(o, 40, "qexmul_rch/the times are not ordered as we assume");
                  break;
                }
                double SLOPE = (US[i][IEND].QF - US[i][IBEG].QF) / (US[i][IEND].TR - US[i][IBEG].TR);
                double PREDV = US[i][IBEG].QF + SLOPE * (CTIME[JUPS] - US[i][IBEG].TR);
                SFLOW = PREDV * SCFAC;
              }
              Q_AGG = Q_AGG + SFLOW;
            }
            if (!ierr) {
              if (IPRT >= IMAX) { ierr = fail(o, 60, "qexmul_rch/QD_TEMP bounds exceeded"); }
              else { QD[IPRT] = Q_AGG; TD[IPRT] = CTIME[JUPS]; IPRT++; }
            }
          }
          if (ierr) { free(MFLG); free(ITIM); free(QD); free(TD); goto done; }
          if (ITIM[JUPS] == USN[JUPS] - 1) {
            MFLG[JUPS] = 1;
            CTIME[JUPS] = DBL_MAX;
          } else {
            ITIM[JUPS] = ITIM[JUPS] + 1;
            CTIME[JUPS] = US[JUPS][ITIM[JUPS]].TR;
          }
        }
      }
      int cnt = 0;
      for (int i = 0; i < NUPS; i++) cnt += MFLG[i];
      if (cnt == NUPS) break;
    }
    free(MFLG); free(ITIM);
    *ND_out = IPRT; *QD_out = QD; *TD_out = TD;
  }
done:
  for (int i = 0; i < NUPS; i++) free(US[i]);
  free(US); free(USN); free(UWIDTH); free(CTIME);
  return ierr;
}

/* kwt_route.f90:461-613 (lakes disabled) */
static int getusq_rch(orc_t *o, int JRCH, double T0, double T1, int *NK1_out, double **Q_JRCH, double **TENTRY, double **T_EXIT) {
  double DT = T1 - T0;
  int ND = 0; double *QD = NULL, *TD = NULL;
  int ierr = 0, isUpLake = 0, iUp = -1;
  if (o->is_lake_sim) {   /* lake outflow enters the river as a single particle, :540-559 */
    int nUps = o->upOff[JRCH + 1] - o->upOff[JRCH];
    for (int e = o->upOff[JRCH]; e < o->upOff[JRCH + 1]; e++) if (o->lakeSlot[o->upIdx[e]] >= 0) { isUpLake = 1; iUp = o->upIdx[e]; }
    if (isUpLake && nUps > 1) return fail(o, 10, "getusq_rch/lake outlet reach should have one upstream lake");
  }
  if (isUpLake) {
    ND = 1; QD = (double *)malloc(sizeof(double)); TD = (double *)malloc(sizeof(double));
    QD[0] = HYD(o, ORC_KWT, iUp).REACH_Q / o->par[ORC_P_WIDTH][JRCH];
    TD[0] = T1;
  } else {
    ierr = qexmul_rch(o, JRCH, T0, T1, &ND, &QD, &TD);
  }
  if (ierr) return ierr;
  orc_fpoint *K = KW(o, JRCH);
  if (o->nkw[JRCH] < 0) {   /* cold start, :587-596 */
    o->nkw[JRCH] = 1;
    K[0].QF = QD[0];
    K[0].TI = T0 - DT - DT * 0;
    K[0].TR = T0 - DT * 0;
    K[0].RF = 1;
  }
  int NJ = o->nkw[JRCH] - 1;
  int NK = NJ + ND;
  o->w_in += NJ + 1;
  double *Q = (double *)malloc((NK + 1) * sizeof(double));
  double *TE = (double *)malloc((NK + 1) * sizeof(double));
  double *TX = (double *)malloc((NK + 1) * sizeof(double));
  for (int k = 0; k <= NJ; k++) { Q[k] = K[k].QF; TE[k] = K[k].TI; TX[k] = K[k].TR; }
  for (int k = 0; k < ND; k++) { Q[NJ + 1 + k] = QD[k]; TE[NJ + 1 + k] = TD[k]; TX[NJ + 1 + k] = -9999.0; }
  free(QD); free(TD);
  *NK1_out = NK + 1; *Q_JRCH = Q; *TENTRY = TE; *T_EXIT = TX;
  return 0;
}

static inline double INTERP(double T0, double Q1, double Q2, double T1, double T2) {
  return Q1 + ((Q2 - Q1) / (T2 - T1)) * (T0 - T1);   /* kwt_route.f90:1115-1121 */
}

/* kwt_route.f90:999-1123; arrays are reallocated to the reduced size */
static int remove_rch(int MAXQPAR, int *size_io, double **Q_JRCH, double **TENTRY, double **T_EXIT) {
  int NPRT = *size_io - 1;
  double *Q = *Q_JRCH, *T = *TENTRY, *Z = *T_EXIT;
  int *PARFLG = (int *)malloc((NPRT + 1) * sizeof(int));
  double *ABSERR = (double *)malloc((NPRT + 1) * sizeof(double));
  int *INDEX1 = (int *)malloc((NPRT + 1) * sizeof(int));
  for (int i = 0; i <= NPRT; i++) { PARFLG[i] = 1; ABSERR[i] = DBL_MAX; }
  for (int i = 1; i <= NPRT - 1; i++) {
    double Q_INTP = INTERP(T[i], Q[i - 1], Q[i + 1], T[i - 1], T[i + 1]);
    ABSERR[i] = fabs(Q_INTP - Q[i]);
  }
  int MPRT;
  for (;;) {
    MPRT = -1;
    for (int i = 0; i <= NPRT; i++) if (PARFLG[i]) INDEX1[++MPRT] = i;   /* pack(INDEX0, PARFLG) */
    if (MPRT < MAXQPAR) break;
    int ISEL = 0;   /* minloc over E_TEMP = pack(ABSERR, PARFLG): first minimum */
    for (int k = 1; k <= MPRT; k++) if (ABSERR[INDEX1[k]] < ABSERR[INDEX1[ISEL]]) ISEL = k;
    if (INDEX1[ISEL - 1] > 0) {
      int INEG = INDEX1[ISEL - 2], IMID = INDEX1[ISEL - 1], IPOS = INDEX1[ISEL + 1];
      ABSERR[IMID] = fabs(INTERP(T[IMID], Q[INEG], Q[IPOS], T[INEG], T[IPOS]) - Q[IMID]);
    }
    if (INDEX1[ISEL + 1] < NPRT) {
      int INEG = INDEX1[ISEL - 1], IMID = INDEX1[ISEL + 1], IPOS = INDEX1[ISEL + 2];
      ABSERR[IMID] = fabs(INTERP(T[IMID], Q[INEG], Q[IPOS], T[INEG], T[IPOS]) - Q[IMID]);
    }
    PARFLG[INDEX1[ISEL]] = 0;
  }
  double *Qn = (double *)malloc((MPRT + 1) * sizeof(double));
  double *Tn = (double *)malloc((MPRT + 1) * sizeof(double));
  double *Zn = (double *)malloc((MPRT + 1) * sizeof(double));
  for (int k = 0; k <= MPRT; k++) { Qn[k] = Q[INDEX1[k]]; Tn[k] = T[INDEX1[k]]; Zn[k] = Z[INDEX1[k]]; }
  free(Q); free(T); free(Z); free(PARFLG); free(ABSERR); free(INDEX1);
  *Q_JRCH = Qn; *TENTRY = Tn; *T_EXIT = Zn; *size_io = MPRT + 1;
  return 0;
}

/* kwt_route.f90:1130-1439.  Arrays here are the (1:NQ1) sections, addressed 1-based via macros. */
static int kinwav_rch(orc_t *o, int JRCH, double T_START, double T_END, double *Q_JRCH, double *TENTRY,
                      double *T_EXIT, int *FROUTE, int NN_in, int *NQ2_out) {
  const double ALFA = 5.0 / 3.0;
  const double K = sqrt(o->par[ORC_P_SLOPE][JRCH]) / o->par[ORC_P_MAN_N][JRCH];
  const double XMX = o->par[ORC_P_LENGTH][JRCH];
  int NN = NN_in, NI = NN_in, NM;
  *NQ2_out = 0;
  if (NN == 0) return 0;
  int *IX = (int *)malloc((NI + 2) * sizeof(int)), *MF = (int *)malloc((NI + 2) * sizeof(int));
  double *T0 = (double *)malloc((NI + 2) * sizeof(double)), *T1 = (double *)malloc((NI + 2) * sizeof(double));
  double *Q0 = (double *)malloc((NI + 2) * sizeof(double)), *Q1 = (double *)malloc((NI + 2) * sizeof(double));
  double *Q2 = (double *)malloc((NI + 2) * sizeof(double)), *WC = (double *)malloc((NI + 2) * sizeof(double));
  int ierr = 0;
  for (int i = 1; i <= NI; i++) {
    MF[i] = i; IX[i] = i;
    Q0[i] = Q1[i] = Q2[i] = Q_JRCH[i - 1];
    T0[i] = T1[i] = TENTRY[i - 1];
  }
  /* WC = ALFA*K**(1./ALFA)*Q1**((ALFA-1.)/ALFA), :1290 */
  {
    double e1 = 1.0 / ALFA, e2 = (ALFA - 1.0) / ALFA;
    double c = ALFA * pow(K, e1);
    for (int i = 1; i <= NN; i++) WC[i] = c * pow(Q1[i], e2);
  }
  if (NN > 1) {
    double X = 0.0;
    for (;;) {
      double XB = XMX;
      int IXB = 0;
      for (int IW = 2; IW <= NN; IW++) {
        int JW = IW - 1;
        if (WC[IW] == 0.0 || WC[JW] == 0.0) continue;
        double WDIFF = 1.0 / WC[JW] - 1.0 / WC[IW];
        if (WDIFF == 0.0) continue;
        if (WC[IW] == WC[JW]) continue;
        double XXB = (T1[IW] - T1[JW]) / WDIFF;
        if (XXB < X || XXB > XB) continue;
        XB = XXB;
        IXB = IW;
      }
      if (XB == XMX) break;
      o->paths[0]++;
      NN = NN - 1;
      int JXB = IXB - 1;
      NM = NI - NN; (void)NM;
      Q2[JXB] = fmax(Q2[JXB], Q2[IXB]);
      Q1[JXB] = fmin(Q1[JXB], Q1[IXB]);
      double A2 = pow(Q2[JXB] / K, 1.0 / ALFA);
      double A1 = pow(Q1[JXB] / K, 1.0 / ALFA);
      double CM = (Q2[JXB] - Q1[JXB]) / (A2 - A1);
      T1[JXB] = T1[JXB] + XB / WC[JXB] - XB / CM;
      WC[JXB] = CM;
      for (int i = IX[IXB]; i <= NI; i++) MF[i] = MF[i] - 1;
      for (int i = IXB; i <= NN; i++) {
        IX[i] = IX[i + 1]; T1[i] = T1[i + 1]; WC[i] = WC[i + 1]; Q1[i] = Q1[i + 1]; Q2[i] = Q2[i + 1];
      }
      X = XB;
    }
  }
  int ICOUNT = 0;
#define RUPDATE(QNEW, TOLD, TNEW)                                                              \
  do {                                                                                        \
    ICOUNT++;                                                                                 \
    if (ICOUNT > NI) { ierr = fail(o, 60, "kinwav_rch/RUPDATE/array bounds exceeded"); break; } \
    Q_JRCH[ICOUNT - 1] = (QNEW); TENTRY[ICOUNT - 1] = (TOLD); T_EXIT[ICOUNT - 1] = (TNEW);    \
    if (ICOUNT > 1) { if (T_EXIT[ICOUNT - 1] <= T_EXIT[ICOUNT - 2]) { T_EXIT[ICOUNT - 1] = T_EXIT[ICOUNT - 2] + 1.0; o->paths[3]++; } } \
    if (ICOUNT == 1 && T_EXIT[ICOUNT - 1] <= T_START) { T_EXIT[ICOUNT - 1] = T_START + 1.0; o->paths[8]++; } \
    if (T_EXIT[ICOUNT - 1] < T_END) FROUTE[ICOUNT - 1] = 1;                                   \
  } while (0)
  for (int IROUTE = 1; IROUTE <= NN && !ierr; IROUTE++) {
    if (WC[IROUTE] < DBL_MIN) {   /* verySmall = tiny(1.0_dp) */
      snprintf(o->msg, sizeof o->msg, "kinwav_rch/zero flow for reach index %d", JRCH + 1);
      ierr = 20; break;
    }
    double TEXIT = fmin(XMX / WC[IROUTE] + T1[IROUTE], DBL_MAX);
    double TNEXT = DBL_MAX;
    if (IROUTE < NN) TNEXT = fmin(XMX / WC[IROUTE + 1] + T1[IROUTE + 1], DBL_MAX);
    if (Q1[IROUTE] != Q2[IROUTE]) {
      if (TEXIT < T_END) {
        double TEXIT2 = fmin(TEXIT + 1.0, TEXIT + 0.5 * (fmin(TNEXT, T_END) - TEXIT));
        if (TEXIT2 == TEXIT) { ierr = fail(o, 30, "kinwav_rch/TEXIT equals TEXIT2 in kinwav"); break; }
        o->paths[1]++;
        RUPDATE(Q1[IROUTE], T1[IROUTE], TEXIT);
        if (ierr) break;
        RUPDATE(Q2[IROUTE], T1[IROUTE], TEXIT2);
      } else {
        o->paths[2]++;
        for (int JROUTE = 1; JROUTE <= NI && !ierr; JROUTE++) {
          if (MF[JROUTE] == IROUTE) RUPDATE(Q0[JROUTE], T0[JROUTE], TEXIT);
        }
      }
    } else {
      RUPDATE(Q1[IROUTE], T1[IROUTE], TEXIT);
    }
  }
#undef RUPDATE
  *NQ2_out = ICOUNT;
  free(IX); free(MF); free(T0); free(T1); free(Q0); free(Q1); free(Q2); free(WC);
  return ierr;
}

/* kwt_route.f90:351-455 water abstraction/injection on particles (only if is_flux_wm) */
static int extract_from_rch(orc_t *o, int JRCH, double T_START, double T_END, double Qtake,
                            double *Q_JRCH, double *T_EXIT, double *TENTRY, int NR) {
  const double alfa = 5.0 / 3.0;
  const double K = sqrt(o->par[ORC_P_SLOPE][JRCH]) / o->par[ORC_P_MAN_N][JRCH];
  double Qavg;
  if (interp_rch(TENTRY, Q_JRCH, NR, T_START, T_END, &Qavg)) return fail(o, 1, "extract_from_rch/interp_rch/bad bounds");
  double totQ = Qavg * o->par[ORC_P_WIDTH][JRCH];
  double *qm = (double *)malloc(NR * sizeof(double));
  qm[0] = Q_JRCH[0];
  if (Qtake > 0.0) {
    double Qfrac = Qtake / totQ;
    for (int i = 1; i < NR; i++) qm[i] = Q_JRCH[i] * (1.0 + Qfrac);
  } else if (Qtake < 0.0 && fabs(Qtake) < totQ) {
    double Qfrac = fabs(Qtake) / totQ;
    for (int i = 1; i < NR; i++) qm[i] = Q_JRCH[i] * (1.0 - Qfrac);
  } else {
    for (int i = 0; i < NR; i++) qm[i] = o->par[ORC_P_MINFLOW][JRCH];
  }
  double c = alfa * pow(K, 1.0 / alfa);
  for (int i = 1; i < NR; i++) {
    /* Reference quirk (kwt_route.f90:434): `wc = ...Q_jrch_mod**...` assigns an expression of
       extent NR (Q_jrch_mod is 0:NR-1) to allocatable wc(1:NR-1); F2003 reallocation gives
       wc(k) = f(Q_jrch_mod(k-1)), i.e. the celerity of particle i uses the flow of particle i-1. */
    double wc = c * pow(qm[i - 1], (alfa - 1.0) / alfa);
    T_EXIT[i] = fmin(o->par[ORC_P_LENGTH][JRCH] / wc + TENTRY[i], DBL_MAX);
  }
  for (int i = 0; i < NR; i++) Q_JRCH[i] = qm[i];
  free(qm);
  return 0;
}

/* kwt_route.f90:36-346 */
int orc_kwt_rch(orc_t *o, int r, double T0, double T1) {
  orc_hyd *h = &HYD(o, ORC_KWT, r);
  orc_fpoint *K = KW(o, r);
  int NUPS = o->nGood[r];
  double *Q_JRCH = NULL, *TENTRY = NULL, *T_EXIT = NULL;
  int size = 0, ierr;
  if (NUPS > 0) {
    o->n_route++;
    ierr = getusq_rch(o, r, T0, T1, &size, &Q_JRCH, &TENTRY, &T_EXIT);
    if (ierr) return ierr;
    double mn = Q_JRCH[0];
    for (int k = 1; k < size; k++) if (Q_JRCH[k] < mn) mn = Q_JRCH[k];
    if (mn < 0.0) { free(Q_JRCH); free(TENTRY); free(T_EXIT); return fail(o, 20, "kwt_rch/negative flow extracted from upstream reach"); }
    double q_upstream = 0.0;
    for (int i = 0; i < NUPS; i++) {
      int e = o->upOff[r] + i;
      if (!o->upGood[e]) continue;
      q_upstream = q_upstream + HYD(o, ORC_KWT, o->upIdx[e]).REACH_Q;
    }
    h->REACH_INFLOW = q_upstream;
  } else {
    o->n_head++;
    h->REACH_INFLOW = 0.0;
    h->REACH_Q = o->BASIN_QR1[r];
    o->nkw[r] = 1;
    K[0].QF = -9999; K[0].TI = -9999; K[0].TR = -9999; K[0].RF = 0;
    return 0;
  }
  if (size > ORC_MAXQPAR) {
    o->paths[5]++; if (size > 64) o->paths[6]++;
    ierr = remove_rch(ORC_MAXQPAR, &size, &Q_JRCH, &TENTRY, &T_EXIT);
    if (ierr) return ierr;
  }
  int NQ1 = size - 1;
  double T_START = T0, T_END = T1;   /* RSTEP = 0 */
  if (o->is_flux_wm && o->REACH_WM_FLUX[r] != ORC_REALMISSING) {
    ierr = extract_from_rch(o, r, T_START, T_END, o->REACH_WM_FLUX[r], Q_JRCH, T_EXIT, TENTRY, size);
    if (ierr) { free(Q_JRCH); free(TENTRY); free(T_EXIT); return ierr; }
  }
  int *FROUTE = (int *)calloc(NQ1 + 2, sizeof(int));
  FROUTE[0] = 1;
  int NQ2 = 0;
  ierr = kinwav_rch(o, r, T_START, T_END, Q_JRCH + 1, TENTRY + 1, T_EXIT + 1, FROUTE + 1, NQ1, &NQ2);
  if (ierr) { free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE); return ierr; }
  int NR = -1;
  for (int k = 0; k <= NQ1; k++) NR += FROUTE[k];
  int NN = NQ2 - NR;
  if (NR + 1 > NQ1) { free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE); return fail(o, 61, "kwt_rch/no non-routed particle left (reference would read out of bounds)"); }
  double QNEW;
  if (interp_rch(T_EXIT, Q_JRCH, NR + 2, T_START, T_END, &QNEW)) {
    free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE);
    return fail(o, 1, "kwt_rch/interp_rch/bad bounds");
  }
  h->REACH_Q = QNEW * o->par[ORC_P_WIDTH][r] + o->BASIN_QR1[r];
  double Q_END = Q_JRCH[NR] + ((Q_JRCH[NR + 1] - Q_JRCH[NR]) / (T_EXIT[NR + 1] - T_EXIT[NR])) * (T_END - T_EXIT[NR]);
  double TIMEI = TENTRY[NR] + ((TENTRY[NR + 1] - TENTRY[NR]) / (T_EXIT[NR + 1] - T_EXIT[NR])) * (T_END - T_EXIT[NR]);
  if (o->nkw[r] < 0) { free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE); return fail(o, 20, "kwt_rch/RCHSTA_out is not associated"); }
  /* KWAVE(0:NQ2+1) */
  if (NQ2 + 2 > ORC_KWSTORE) { free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE); return fail(o, 62, "kwt_rch/oracle KWAVE storage exceeded"); }
  o->nkw[r] = NQ2 + 2;
  K[NR + 1].QF = Q_END; K[NR + 1].TI = TIMEI; K[NR + 1].TR = T_END; K[NR + 1].RF = 1;
  for (int k = 0; k <= NR; k++) { K[k].QF = Q_JRCH[k]; K[k].TI = TENTRY[k]; K[k].TR = T_EXIT[k]; K[k].RF = FROUTE[k]; }
  for (int k = NR + 1; k <= NQ2; k++) { K[k + 1].QF = Q_JRCH[k]; K[k + 1].TI = TENTRY[k]; K[k + 1].TR = T_EXIT[k]; K[k + 1].RF = FROUTE[k]; }
  o->w_out += NQ2 + 2;
  free(Q_JRCH); free(TENTRY); free(T_EXIT); free(FROUTE);
  /* outlet: strip routed particles itself, :325-344 */
  if (o->down[r] < 0 || (o->is_lake_sim && o->lakeInlet[r])) {
    memmove(K, K + (NR + 1), (size_t)(NN + 1) * sizeof(orc_fpoint));
    o->nkw[r] = NN + 1;
  }
  return 0;
}
