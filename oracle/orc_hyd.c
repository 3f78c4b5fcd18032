/* CPU oracle: hydraulic.f90, advection_diffusion.f90, mc_route.f90, dfw_route.f90, kwe_route.f90
   (test infrastructure; see mzr_oracle.h). */
#include <math.h>
#include <stdlib.h>
#include "orc_internal.h"

static const double const13 = 1.0 / 3.0, const23 = 2.0 / 3.0, const53 = 5.0 / 3.0, const103 = 10.0 / 3.0;
static const double err_thresh = 0.005;   /* hydraulic.f90:37 */
static const double Qmin = 1.e-50;

/* Integer powers x**n: AMD flang 22 (the compiler of the oracle/_ref pin) expands them into the
   left-to-right product ((x*x)*x)*...; verified bitwise against the reference build. */
static inline double pw2(double x) { return x * x; }
static inline double pw3(double x) { return x * x * x; }
static inline double pw4(double x) { return x * x * x * x; }
static inline double pw5(double x) { return x * x * x * x * x; }

/* hydraulic.f90:46-78 */
double orc_Btop(double yin, double b, double zc, double zf, double bankDepth) {
  if (yin <= bankDepth) return b + 2 * yin * zc;
  double Bt = b + 2 * bankDepth * zc;
  return Bt + zf * (yin - bankDepth) * 2;
}
/* hydraulic.f90:83-115 */
double orc_Pwet(double yin, double b, double zc, double zf, double bankDepth) {
  if (yin <= bankDepth) return b + 2 * yin * sqrt(1 + zc * zc);
  double P = b + 2 * bankDepth * sqrt(1 + zc * zc);
  return P + 2 * (yin - bankDepth) * sqrt(1 + zf * zf);
}
/* hydraulic.f90:120-156 */
double orc_flow_area(double yin, double b, double zc, double zf, double bankDepth) {
  if (yin <= bankDepth) return yin * (b + zc * yin);
  double A = bankDepth * (b + zc * bankDepth);
  double Bt = orc_Btop(yin, b, zc, zf, bankDepth);
  double Bt_bank = orc_Btop(bankDepth, b, zc, zf, bankDepth);
  return A + (yin - bankDepth) * (Bt + Bt_bank) / 2.0;
}
/* hydraulic.f90:161-202 */
double orc_water_height(double flowArea, double b, double zc, double zf, double bankDepth) {
  double A_bank = orc_flow_area(bankDepth, b, zc, zf, bankDepth);
  if (flowArea > A_bank) {
    double Bt_bank = orc_Btop(bankDepth, b, zc, zf, bankDepth);
    double disc = Bt_bank * Bt_bank - 4.0 * zf * (A_bank - flowArea);
    return bankDepth + (-Bt_bank + sqrt(disc)) / (2.0 * zf);
  }
  if (zc == 0) return flowArea / b;
  return (-b + sqrt(b * b + 4.0 * flowArea * zc)) / (2.0 * zc);
}
/* hydraulic.f90:306-433; every hot-path caller passes zf and bankDepth, so floodplain = .true. */
double orc_flow_depth(double Qin, double b, double zc, double S, double n, double zf, double bankDepth) {
  double err = 100.0, fd = 0.0;
  if (!(Qin > Qmin)) return 0.0;
  double Abf = orc_flow_area(bankDepth, b, zc, zf, bankDepth);
  double Pbf = orc_Pwet(bankDepth, b, zc, zf, bankDepth);
  double Bbf = orc_Btop(bankDepth, b, zc, zf, bankDepth);
  double Qbf = Abf * pow(Abf / Pbf, const23) * sqrt(S) / n;
  if (Qin < Qbf) {
    double Coef1 = pw3(sqrt(S) / n / Qin);
    double Coef2 = 2 * sqrt(zc * zc + 1.0);
    double y0 = pow(1.0 / Coef1 / pw3(b), 1.0 / 5.0);
    while (err > err_thresh) {
      double A = orc_flow_area(y0, b, zc, zf, bankDepth);
      double Bt = orc_Btop(y0, b, zc, zf, bankDepth);
      double P = orc_Pwet(y0, b, zc, zf, bankDepth);
      double h = Coef1 * pw5(A) / pw2(P) - 1.0;
      double dhdy = Coef1 * (5 * pw4(A) * Bt * P - 2 * Coef2 * pw5(A)) / pw3(P);
      fd = y0 - h / dhdy;
      err = fabs((fd - y0) / fd);
      y0 = fd;
    }
  } else {
    double y0 = bankDepth + 2.0;
    double Coef1 = sqrt(S) / n / pow(Pbf, const23);
    double Coef2 = 2 * pow(zf / 2, const53) * sqrt(S) / n / pow(zf * zf + 1.0, const13);
    while (err > err_thresh) {
      double ye = y0 - bankDepth;
      double h = Coef1 * pow(Abf + Bbf * ye, const53) + Coef2 * pow(ye, const103) / pow(ye, const23) - Qin;
      double dhdy = Coef1 * const53 * Bbf * pow(Abf + Bbf * ye, const23) + Coef2 * (const103 - const23) * pow(ye, const53);
      fd = y0 - h / dhdy;
      err = fabs((fd - y0) / fd);
      y0 = fd;
    }
  }
  return fd;
}
/* hydraulic.f90:438-484 */
double orc_celerity(double Qin, double y, double b, double zc, double S, double n, double zf, double bankDepth) {
  (void)S;
  if (!(y > 0.0)) return 0.0;
  double Bt = orc_Btop(y, b, zc, zf, bankDepth);
  double A = orc_flow_area(y, b, zc, zf, bankDepth);
  double P = orc_Pwet(y, b, zc, zf, bankDepth);
  double Sf = pw2(Qin * n / A / pow(A / P, const23));
  return const53 * pow(Sf, 0.3) * pow(Qin, 0.4) / pow(Bt, 0.4) / pow(n, 0.6);
}
/* hydraulic.f90:489-535 */
double orc_diffusivity(double Qin, double y, double b, double zc, double S, double n, double zf, double bankDepth) {
  (void)S;
  if (!(y > 0.0)) return 0.0;
  double Bt = orc_Btop(y, b, zc, zf, bankDepth);
  double A = orc_flow_area(y, b, zc, zf, bankDepth);
  double P = orc_Pwet(y, b, zc, zf, bankDepth);
  double Sf = pw2(Qin * n / A / pow(A / P, const23));
  return fabs(Qin) / Sf / Bt / 2.0;
}

/* advection_diffusion.f90:19-258 with advec_scheme = central (2), downstreamBC = Neumann (2),
   wck = wdk = 1 -- the only configuration the hot path uses (dfw_route.f90:36-37,301-312;
   kwe_route.f90 relies on the same defaults). */
void orc_solve_ade(double L, int nMol, double dt_local, double FluxUpstream, double ck, double dk,
                   const double *FluxPrev, double *FluxSolved) {
  double up[32], mid[32], low[32], b[32], D[32], b1[32];   /* diagonal(:,1..3), rhs */
  const double wck = 1.0, wdk = 1.0;
  int Nx = nMol - 1;
  double dx = L / (Nx - 1);
  double Cd = dk * dt_local / (dx * dx);
  double Ca = ck * dt_local / dx;
  /* 1-based indexing below mirrors the Fortran */
  mid[1] = 1.0;
  for (int i = 2; i <= nMol - 1; i++) mid[i] = 2.0 + 4 * wdk * Cd;
  mid[nMol] = 1.0;
  for (int i = 1; i <= nMol; i++) up[i] = 0.0;
  for (int i = 3; i <= nMol; i++) up[i] = wck * Ca - 2.0 * wdk * Cd;
  for (int i = 1; i <= nMol; i++) low[i] = 0.0;
  for (int i = 1; i <= nMol - 2; i++) low[i] = -wck * Ca - 2.0 * wdk * Cd;
  low[nMol - 1] = -1.0;
  b[1] = FluxUpstream;
  b[nMol] = FluxPrev[nMol - 1] - FluxPrev[nMol - 2];   /* Sbc */
  for (int i = 2; i <= nMol - 1; i++) {
    b[i] = ((1.0 - wck) * Ca + 2.0 * (1.0 - wdk) * Cd) * FluxPrev[i - 2]
         + (2.0 - 4.0 * (1.0 - wdk) * Cd) * FluxPrev[i - 1]
         - ((1.0 - wck) * Ca - 2.0 * (1.0 - wdk) * Cd) * FluxPrev[i];
  }
  /* TDMA, advection_diffusion.f90:211-258 */
  for (int i = 1; i <= nMol; i++) { D[i] = mid[i]; b1[i] = b[i]; }
  for (int i = 2; i <= nMol; i++) {
    double coef = low[i - 1] / D[i - 1];
    D[i] = D[i] - coef * up[i];
    b1[i] = b1[i] - coef * b1[i - 1];
  }
  FluxSolved[nMol - 1] = b1[nMol] / D[nMol];
  for (int i = nMol - 1; i >= 1; i--) FluxSolved[i - 1] = (b1[i] - up[i + 1] * FluxSolved[i]) / D[i];
}

/* mc_route.f90:46-416 */
int orc_mc_rch(orc_t *o, int r, double T0, double T1) {
  (void)T0; (void)T1;
  orc_hyd *h = &HYD(o, ORC_MC, r);
  double q_upstream, q_upstream_mod, Qlat; int isHW;
  orc_preamble(o, r, ORC_MC, &q_upstream, &q_upstream_mod, &Qlat, &isHW);
  double *mol = o->molMC + (size_t)r * ORC_NMOL_MC;
  const double Y = 0.5, QminMC = 1.e-50;
  double dt = o->dt;
  double r_slope = o->par[ORC_P_SLOPE][r], r_man_n = o->par[ORC_P_MAN_N][r], r_width = o->par[ORC_P_WIDTH][r];
  double r_depth = o->par[ORC_P_DEPTH][r], side_slope = o->par[ORC_P_SIDE_SLOPE][r];
  double fldp_slope = o->par[ORC_P_FLDP_SLOPE][r], r_storage = o->par[ORC_P_STORAGE][r];
  double rlength = o->par[ORC_P_LENGTH][r];
  double Q00 = mol[0], Q01 = mol[1], Q10, Q11;
  if (!isHW || o->hw_drain_point == 1) {
    if (rlength > o->min_length_route) {
      double theta = dt / rlength;
      Q10 = q_upstream_mod;
      double Qbar = (Q00 + Q10 + Q01) / 3.0;
      if (Qbar > QminMC) {
        double depth = orc_flow_depth(fabs(Qbar), r_width, side_slope, r_slope, r_man_n, fldp_slope, r_depth);
        double ck = orc_celerity(fabs(Qbar), depth, r_width, side_slope, r_slope, r_man_n, fldp_slope, r_depth);
        double Cn = ck * theta;
        int ntSub = 1;
        double dTsub = dt;
        if (Cn > 1.0) {
          ntSub = (int)ceil(dt / rlength * ck);
          dTsub = dt / ntSub;
        }
        double *QoutLocal = (double *)malloc((ntSub + 1) * sizeof(double));
        double *QinLocal = (double *)malloc((ntSub + 1) * sizeof(double));
        QoutLocal[0] = Q01; QinLocal[0] = Q00;
        for (int ix = 1; ix <= ntSub; ix++) QinLocal[ix] = Q10;
        for (int ix = 1; ix <= ntSub; ix++) {
          Qbar = (QinLocal[ix] + QinLocal[ix - 1] + QoutLocal[ix - 1]) / 3.0;
          if (Qbar > QminMC) {
            depth = orc_flow_depth(fabs(Qbar), r_width, side_slope, r_slope, r_man_n, fldp_slope, r_depth);
            double topWidth = orc_Btop(depth, r_width, side_slope, fldp_slope, r_depth);
            ck = orc_celerity(fabs(Qbar), depth, r_width, side_slope, r_slope, r_man_n, fldp_slope, r_depth);
            double X = 0.5 * (1.0 - Qbar / (topWidth * r_slope * ck * rlength));
            Cn = ck * dTsub / rlength;
            double C0 = (-X + Cn * (1 - Y)) / (1 - X + Cn * (1 - Y));
            double C1 = (X + Cn * Y) / (1 - X + Cn * (1 - Y));
            double C2 = (1 - X - Cn * Y) / (1 - X + Cn * (1 - Y));
            QoutLocal[ix] = C0 * QinLocal[ix] + C1 * QinLocal[ix - 1] + C2 * QoutLocal[ix - 1];
            QoutLocal[ix] = fmax(0.0, QoutLocal[ix]);
          } else {
            QoutLocal[ix] = 0.0;
          }
        }
        double s = 0.0;
        for (int ix = 1; ix <= ntSub; ix++) s = s + QoutLocal[ix];   /* sum(QoutLocal(1:nTsub)) */
        Q11 = s / (double)ntSub;
        free(QoutLocal); free(QinLocal);
        if (fabs(Q11) > 0.0) {
          /* `*0.999` default-real literal, mc_route.f90:352 */
          double pcntReduc = fmin((h->REACH_VOL[1] / dt + Q10) * (double)0.999f / Q11, 1.0);
          Q11 = Q11 * pcntReduc;
        }
        h->REACH_VOL[1] = h->REACH_VOL[1] + (Q10 - Q11) * dt;
        h->FLOOD_VOL[1] = h->REACH_VOL[1] > r_storage ? h->REACH_VOL[1] - r_storage : 0.0;
        h->REACH_ELE = orc_water_height(h->REACH_VOL[1] / rlength, r_width, side_slope, fldp_slope, r_depth);
        h->REACH_Q = Q11 + Qlat;
      } else {
        Q11 = 0.0;
        h->REACH_Q = Q11 + Qlat;
        h->REACH_VOL[1] = h->REACH_VOL[1] + (Q10 - Q11) * dt;
        h->FLOOD_VOL[1] = h->REACH_VOL[1] > r_storage ? h->REACH_VOL[1] - r_storage : 0.0;
        h->REACH_ELE = orc_water_height(h->REACH_VOL[1] / rlength, r_width, side_slope, fldp_slope, r_depth);
      }
    } else {
      Q10 = q_upstream_mod; Q11 = q_upstream_mod;
      h->REACH_Q = q_upstream_mod + Qlat;
      h->REACH_VOL[0] = 0.0; h->REACH_VOL[1] = 0.0; h->FLOOD_VOL[1] = 0.0; h->REACH_ELE = 0.0;
    }
  } else {
    Q10 = 0.0; Q11 = 0.0;
    h->REACH_Q = Qlat;
    h->REACH_VOL[0] = 0.0; h->REACH_VOL[1] = 0.0; h->FLOOD_VOL[1] = 0.0; h->REACH_ELE = 0.0;
  }
  mol[0] = Q10; mol[1] = Q11;
  return orc_finish_rch(o, r, ORC_MC, q_upstream, Qlat);
}

/* dfw_route.f90:49-370 (method == ORC_DW) and kwe_route.f90:46-363 (method == ORC_KW, dk = 0) */
int orc_dw_rch(orc_t *o, int r, int method) {
  orc_hyd *h = &HYD(o, method, r);
  double Qupstream, Qupstream_mod, Qlat; int isHW;
  orc_preamble(o, r, method, &Qupstream, &Qupstream_mod, &Qlat, &isHW);
  int nMol = method == ORC_DW ? ORC_NMOL_DW : ORC_NMOL_KW;
  double *mol = (method == ORC_DW ? o->molDW : o->molKW) + (size_t)r * nMol;
  double dt = o->dt;
  double S = o->par[ORC_P_SLOPE][r], n = o->par[ORC_P_MAN_N][r], bt = o->par[ORC_P_WIDTH][r];
  double bankDepth = o->par[ORC_P_DEPTH][r], zc = o->par[ORC_P_SIDE_SLOPE][r];
  double zf = o->par[ORC_P_FLDP_SLOPE][r], bankVol = o->par[ORC_P_STORAGE][r], L = o->par[ORC_P_LENGTH][r];
  double Qu = Qupstream_mod;
  if (!isHW || o->hw_drain_point == 1) {
    if (L > o->min_length_route) {
      double Qprev[32], Qlocal[32];
      for (int i = 0; i < nMol; i++) Qprev[i] = mol[i];
      double dTsub = dt / 1;
      double Qbar = (Qu + Qprev[0] + Qprev[nMol - 2]) / 3.0;
      double depth = orc_flow_depth(fabs(Qbar), bt, zc, S, n, zf, bankDepth);
      double ck = orc_celerity(fabs(Qbar), depth, bt, zc, S, n, zf, bankDepth);
      double dk = method == ORC_DW ? orc_diffusivity(fabs(Qbar), depth, bt, zc, S, n, zf, bankDepth) : 0.0;
      orc_solve_ade(L, nMol, dTsub, Qu, ck, dk, Qprev, Qlocal);
      if (fabs(Qlocal[nMol - 2]) > 0.0) {
        double volTmp = fmax(0.0, h->REACH_VOL[1]);
        double qoutTmp = Qlocal[nMol - 2] * dt;
        double pcntReduc = fmin((volTmp + dt * Qu) * 0.999 / qoutTmp, 1.0);   /* 0.999_dp */
        for (int i = 1; i < nMol; i++) Qlocal[i] = Qlocal[i] * pcntReduc;
      }
      h->REACH_VOL[1] = h->REACH_VOL[1] + (Qu - Qlocal[nMol - 2]) * dt;
      h->FLOOD_VOL[1] = h->REACH_VOL[1] > bankVol ? h->REACH_VOL[1] - bankVol : 0.0;
      h->REACH_ELE = orc_water_height(h->REACH_VOL[1] / L, bt, zc, zf, bankDepth);
      h->REACH_Q = Qlocal[nMol - 2] + Qlat;
      for (int i = 0; i < nMol; i++) mol[i] = Qlocal[i];
    } else {
      h->REACH_Q = Qu + Qlat;
      for (int i = 0; i < nMol; i++) mol[i] = 0.0;
      mol[nMol - 1] = h->REACH_Q;
      h->REACH_VOL[0] = 0.0; h->REACH_VOL[1] = 0.0; h->FLOOD_VOL[1] = 0.0; h->REACH_ELE = 0.0;
    }
  } else {
    h->REACH_Q = Qlat;
    h->REACH_VOL[0] = 0.0; h->REACH_VOL[1] = 0.0; h->FLOOD_VOL[1] = 0.0; h->REACH_ELE = 0.0;
    for (int i = 0; i < nMol; i++) mol[i] = 0.0;
    mol[nMol - 1] = h->REACH_Q;
  }
  return orc_finish_rch(o, r, method, Qupstream, Qlat);
}
