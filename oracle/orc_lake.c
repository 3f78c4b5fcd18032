/* CPU oracle: lakes and reservoirs, lake_route.f90:28-472 (test infrastructure; see mzr_oracle.h). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

enum { P_D03_MaxStorage = 0, P_D03_Coefficient, P_D03_Power, P_D03_S0,
       P_HYP_E_emr, P_HYP_E_lim, P_HYP_E_min, P_HYP_E_zero, P_HYP_Qrate_emr, P_HYP_Erate_emr, P_HYP_Qrate_prim,
       P_HYP_Qrate_amp, P_HYP_Qrate_phs, P_HYP_prim_F, P_HYP_A_avg, P_HYP_Qsim_mode,
       P_H06_Smax, P_H06_alpha, P_H06_envfact, P_H06_S_ini, P_H06_c1, P_H06_c2, P_H06_exponent, P_H06_denominator,
       P_H06_c_compare, P_H06_frac_Sdead, P_H06_E_rel_ini,
       P_H06_I_Jan, P_H06_D_Jan = P_H06_I_Jan + 12,
       P_H06_purpose = P_H06_D_Jan + 12, P_H06_I_mem_F, P_H06_D_mem_F, P_H06_I_mem_L, P_H06_D_mem_L };

static const double secprday = 86400.0, pi = 3.14159265359;   /* public_var.f90:16 (truncated on purpose) */
static const int days_per_yr = 365, months_per_yr = 12;

int orc_set_lakes(orc_t *o, int LakeInputOption, int calendarId, int nLake, const int *lakeReach,
                  const int *modelType, const double *par) {
  int N = o->N;
  o->is_lake_sim = 1; o->LakeInputOption = LakeInputOption; o->calendarId = calendarId; o->nLake = nLake;
  o->lakeSlot = (int *)malloc(N * sizeof(int)); o->lakeModel = (int *)calloc(nLake ? nLake : 1, sizeof(int));
  o->lakeInlet = (int *)calloc(N, sizeof(int));
  for (int r = 0; r < N; r++) o->lakeSlot[r] = -1;
  o->lakePar = (double *)calloc((size_t)(nLake ? nLake : 1) * ORC_NLAKEPAR, sizeof(double));
  for (int l = 0; l < nLake; l++) {
    o->lakeSlot[lakeReach[l] - 1] = l; o->lakeModel[l] = modelType[l];
    for (int p = 0; p < ORC_NLAKEPAR; p++) o->lakePar[(size_t)l * ORC_NLAKEPAR + p] = par[(size_t)p * nLake + l];
  }
  for (int r = 0; r < N; r++) if (o->down[r] >= 0 && o->lakeSlot[o->down[r]] >= 0) o->lakeInlet[r] = 1;
  o->basinEvapo = (double *)calloc(N, sizeof(double)); o->basinPrecip = (double *)calloc(N, sizeof(double));
  o->qpast = (double **)calloc(nLake ? nLake : 1, sizeof(double *)); o->dpast = (double **)calloc(nLake ? nLake : 1, sizeof(double *));
  o->qpastLen = (int *)calloc(nLake ? nLake : 1, sizeof(int)); o->dpastLen = (int *)calloc(nLake ? nLake : 1, sizeof(int));
  o->iTime = 0;
  /* KWT: a lake reach holds one sentinel particle from the start (init_model_data.f90:431-439) */
  if (o->kw) for (int r = 0; r < N; r++) if (o->lakeSlot[r] >= 0) {
    orc_fpoint *K = o->kw + (size_t)r * ORC_KWSTORE;
    o->nkw[r] = 1; K[0].QF = -9999; K[0].TI = -9999; K[0].TR = -9999; K[0].RF = 0;
  }
  return 0;
}

/* sum(x(1:n)) / n with the elements in storage order */
static double mean_row(const double *row, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s = s + row[i];
  return s / n;
}

/* Hanasaki memory update for one family (inflow or demand), lake_route.f90:227-276 / :284-336 */
static int h06_memory(orc_t *o, double **past, int *pastLen, double *P, int base, int mem_L, double newval) {
  const double dt = o->dt;
  int L31 = (int)floor(mem_L * 31 * secprday / dt);
  if (!*past) {
    *pastLen = L31;
    *past = (double *)malloc((size_t)12 * L31 * sizeof(double));
    for (int m = 0; m < 12; m++) for (int k = 0; k < L31; k++) (*past)[(size_t)m * L31 + k] = P[base + m];
  } else {
    int L = *pastLen, m = o->month - 1;
    double *row = *past + (size_t)m * L;
    memmove(row + 1, row, (size_t)(L - 1) * sizeof(double));
    row[0] = newval;
  }
  const int L = *pastLen;
  static const int m31[7] = {0, 2, 4, 6, 7, 9, 11};
  for (int k = 0; k < 7; k++) P[base + m31[k]] = mean_row(*past + (size_t)m31[k] * L, L31);
  int L30 = (int)floor(mem_L * 30 * secprday / dt);
  /* the reference updates Apr, Jun, Sep for the inflow family and Apr, Jun, Sep, Nov for the demand family */
  static const int m30[4] = {3, 5, 8, 10};
  const int n30 = (base == P_H06_I_Jan) ? 3 : 4;
  for (int k = 0; k < n30; k++) P[base + m30[k]] = mean_row(*past + (size_t)m30[k] * L, L30);
  int LF = (o->calendarId == 0) ? (int)floor(mem_L * 28 * secprday / dt) : (int)floor(mem_L * 28.25 * secprday / dt);
  P[base + 1] = mean_row(*past + (size_t)1 * L, LF);
  return 0;
}

int orc_lake_route(orc_t *o, int r, int method) {
  orc_hyd *h = &HYD(o, method, r);
  const int l = o->lakeSlot[r];
  double *P = o->lakePar + (size_t)l * ORC_NLAKEPAR;
  const int type = o->lakeModel[l];
  const double dt = o->dt;
  double q_upstream = 0.0;
  for (int e = o->upOff[r]; e < o->upOff[r + 1]; e++) q_upstream = q_upstream + HYD(o, method, o->upIdx[e]).REACH_Q;
  const int targ = o->lakeTarg && o->lakeTarg[l];
  const double wmvol = (targ && o->wmVol) ? o->wmVol[(size_t)(o->iTime - 1 - o->wmVolFirst) * o->N + r] : 0.0;   /* REACH_WM_VOL, main_route.f90:115-122 */
  if (o->iTime == 1 && o->volJumpstart && targ) {   /* lake_route.f90:139-142 */
    h->REACH_VOL[1] = wmvol;
  } else if (o->iTime == 1) {   /* cold start, lake_route.f90:143-158 */
    switch (type) {
      case 0: h->REACH_VOL[1] = P[P_D03_S0]; break;
      case 1: h->REACH_VOL[1] = P[P_D03_MaxStorage]; break;
      case 2: h->REACH_VOL[1] = P[P_H06_Smax]; break;
      case 3: h->REACH_VOL[1] = (P[P_HYP_E_emr] - P[P_HYP_E_zero]) * P[P_HYP_A_avg]; break;
      default: snprintf(o->msg, sizeof o->msg, "lake_route/unable to identify the parametric lake model type"); return 20;
    }
  }
  h->REACH_VOL[0] = h->REACH_VOL[1];
  h->REACH_VOL[1] = h->REACH_VOL[1] + q_upstream * dt;
  if (o->LakeInputOption == 1 || o->LakeInputOption == 2) h->REACH_VOL[1] = h->REACH_VOL[1] + o->BASIN_QR1[r] * dt;
  if (o->LakeInputOption == 0 || o->LakeInputOption == 2) {
    h->REACH_VOL[1] = h->REACH_VOL[1] + o->basinPrecip[r] * dt;
    if (h->REACH_VOL[1] > o->basinEvapo[r] * dt) {
      h->REACH_VOL[1] = h->REACH_VOL[1] - o->basinEvapo[r] * dt;
    } else {
      o->basinEvapo[r] = h->REACH_VOL[1] / dt;
      h->REACH_VOL[1] = 0.0;
    }
  }
  h->REACH_WM_FLUX_actual = o->REACH_WM_FLUX[r];
  if (o->REACH_WM_FLUX[r] != ORC_REALMISSING && o->is_flux_wm) {
    if (o->REACH_WM_FLUX[r] <= 0) {
      h->REACH_VOL[1] = h->REACH_VOL[1] - o->REACH_WM_FLUX[r] * dt;
      h->REACH_WM_FLUX_actual = o->REACH_WM_FLUX[r];
    } else if (o->REACH_WM_FLUX[r] * dt <= h->REACH_VOL[1]) {
      h->REACH_VOL[1] = h->REACH_VOL[1] - o->REACH_WM_FLUX[r] * dt;
      h->REACH_WM_FLUX_actual = o->REACH_WM_FLUX[r];
    } else {
      h->REACH_WM_FLUX_actual = h->REACH_VOL[1] / dt;
      h->REACH_VOL[1] = 0.0;
    }
  }
  if (targ) {   /* the lake follows the given target volume, lake_route.f90:197-205 */
    if (h->REACH_VOL[1] < wmvol) {
      h->REACH_Q = 0;
    } else {
      h->REACH_Q = (h->REACH_VOL[1] - wmvol) / dt;
      h->REACH_VOL[1] = wmvol;
    }
  } else switch (type) {
    case 0: h->REACH_Q = 0.0; break;
    case 1: {
      if ((h->REACH_VOL[1] - P[P_D03_S0]) > 0) {
        h->REACH_Q = P[P_D03_Coefficient] * (h->REACH_VOL[1] - P[P_D03_S0]) *
                     pow((h->REACH_VOL[1] - P[P_D03_S0]) / (P[P_D03_MaxStorage] - P[P_D03_S0]), P[P_D03_Power]);
      } else {
        h->REACH_Q = 0;
      }
      h->REACH_Q = h->REACH_Q / secprday;
      h->REACH_Q = fmin(h->REACH_Q, h->REACH_VOL[1] / dt);
      h->REACH_VOL[1] = h->REACH_VOL[1] - h->REACH_Q * dt;
      break;
    }
    case 2: {
      if (P[P_H06_I_mem_F] != 0.0) h06_memory(o, &o->qpast[l], &o->qpastLen[l], P, P_H06_I_Jan, (int)P[P_H06_I_mem_L], q_upstream);
      if (P[P_H06_D_mem_F] != 0.0 && o->REACH_WM_FLUX[r] != ORC_REALMISSING && o->is_flux_wm) {
        if (o->REACH_WM_FLUX[r] < 0) o->REACH_WM_FLUX[r] = 0.0;
        h06_memory(o, &o->dpast[l], &o->dpastLen[l], P, P_H06_D_Jan, (int)P[P_H06_D_mem_L], o->REACH_WM_FLUX[r]);
      }
      const double *I_months = P + P_H06_I_Jan, *D_months = P + P_H06_D_Jan;
      double sI = 0.0, sD = 0.0;
      for (int m = 0; m < 12; m++) { sI = sI + I_months[m]; sD = sD + D_months[m]; }
      const double I_yearly = sI / months_per_yr, D_yearly = sD / months_per_yr;
      const double c = P[P_H06_Smax] / (I_yearly * days_per_yr * secprday);
      int start_month = 0;   /* implicit SAVE in the reference but always reassigned by the wettest-month scan
                                unless no month reaches the mean (then the previous lake's value leaks; not restated) */
      for (int i = 1; i <= months_per_yr; i++) if (I_yearly <= I_months[i - 1]) start_month = i + 1;
      if (o->month == start_month && o->day == 1) P[P_H06_E_rel_ini] = h->REACH_VOL[1] / (P[P_H06_alpha] * P[P_H06_Smax]);
      double target_r;
      if ((int)P[P_H06_purpose] == 1) {
        if (P[P_H06_envfact] * I_yearly <= D_yearly)
          target_r = I_months[o->month - 1] * P[P_H06_c1] + I_yearly * P[P_H06_c2] * (D_months[o->month - 1] / D_yearly);
        else
          target_r = I_yearly + D_months[o->month - 1] - D_yearly;
      } else {
        target_r = I_yearly;
      }
      if (c >= P[P_H06_c_compare]) {
        h->REACH_Q = target_r * P[P_H06_E_rel_ini];
      } else if (0 <= c && c < P[P_H06_c_compare]) {
        const double f = pow(c / P[P_H06_denominator], P[P_H06_exponent]);
        h->REACH_Q = P[P_H06_E_rel_ini] * target_r * f + q_upstream * (1 - f);
      }
      if (h->REACH_VOL[1] < (P[P_H06_Smax] * P[P_H06_frac_Sdead])) {
        h->REACH_Q = h->REACH_Q - (P[P_H06_Smax] * P[P_H06_frac_Sdead] - h->REACH_VOL[1]) / dt;
        if (h->REACH_Q < 0) h->REACH_Q = 0;
      } else if (h->REACH_VOL[1] > P[P_H06_Smax]) {
        h->REACH_Q = h->REACH_Q + (h->REACH_VOL[1] - P[P_H06_Smax]) / dt;
      }
      h->REACH_VOL[1] = h->REACH_VOL[1] - h->REACH_Q * dt;
      break;
    }
    case 3: {
      h->REACH_ELE = h->REACH_VOL[1] / P[P_HYP_A_avg] + P[P_HYP_E_zero];
      const double Day_of_year = (double)o->dayofyear;
      const double F_sin = fmax(0.0, (1 + P[P_HYP_Qrate_amp] * sin(2 * pi * (Day_of_year + (int)P[P_HYP_Qrate_phs]) / 365)));
      const double F_lin = fmin(fmax((h->REACH_ELE - P[P_HYP_E_min]) / (P[P_HYP_E_lim] - P[P_HYP_E_min]), 0.0), 1.0);
      const int F_prim = P[P_HYP_prim_F] != 0.0 ? 1 : 0;
      const double Q_prim = F_sin * F_lin * F_prim * P[P_HYP_Qrate_prim];
      double Q_spill = 0.0;
      if (h->REACH_ELE > P[P_HYP_E_emr]) Q_spill = P[P_HYP_Qrate_emr] * pow(h->REACH_ELE - P[P_HYP_E_emr], P[P_HYP_Erate_emr]);
      const double Q_sim = P[P_HYP_Qsim_mode] != 0.0 ? Q_prim + Q_spill : fmax(Q_prim, Q_spill);
      h->REACH_Q = fmin(Q_sim, fmax(0.0, (h->REACH_ELE - P[P_HYP_E_min]) * P[P_HYP_A_avg]) / dt);
      h->REACH_VOL[1] = h->REACH_VOL[1] - h->REACH_Q * dt;
      break;
    }
    default: snprintf(o->msg, sizeof o->msg, "lake_route/unable to identify the parametric lake model type"); return 20;
  }
  /* comp_reach_wb with lakeFlag, water_balance.f90:52-85 */
  {
    const double dVol = h->REACH_VOL[1] - h->REACH_VOL[0];
    const double Qin = q_upstream * dt, Qlateral = o->BASIN_QR1[r] * dt, precip = o->basinPrecip[r] * dt;
    const double Qout = -1.0 * h->REACH_Q * dt, Qtake_actual = -1.0 * h->REACH_WM_FLUX_actual * dt;
    const double evapo = -1.0 * o->basinEvapo[r] * dt;
    h->WB = dVol - (Qin + Qlateral + precip + Qtake_actual + Qout + evapo);
  }
  return 0;
}

/* target-volume lakes: flags[nLake], is_vol_wm_jumpstart, and REACH_WM_VOL[nSteps][N] for steps firstStep+1 .. (kept by reference) */
int orc_set_lake_target(orc_t *o, const int *flags, int jumpstart, int firstStep, const double *wmvol) {
  if (!o->is_lake_sim) return 1;
  if (!o->lakeTarg) o->lakeTarg = (int *)calloc(o->nLake ? o->nLake : 1, sizeof(int));
  for (int l = 0; l < o->nLake; l++) o->lakeTarg[l] = flags[l];
  o->volJumpstart = jumpstart; o->wmVol = wmvol; o->wmVolFirst = firstStep;
  return 0;
}

