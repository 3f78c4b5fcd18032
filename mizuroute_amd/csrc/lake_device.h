// Lakes and reservoirs on the device: lake_route, route/build/src/lake_route.f90:28-472.
// A lake reach replaces the reach solver of whatever method is being swept (main_route.f90:375-381);
// lakes are rare (<< 1 % of reaches), so this is a branch of the stage kernels, not a kernel.
#pragma once
#include "mzr_device.h"

namespace mzr_lake {

enum { P_D03_MaxStorage = 0, P_D03_Coefficient, P_D03_Power, P_D03_S0,
       P_HYP_E_emr, P_HYP_E_lim, P_HYP_E_min, P_HYP_E_zero, P_HYP_Qrate_emr, P_HYP_Erate_emr, P_HYP_Qrate_prim,
       P_HYP_Qrate_amp, P_HYP_Qrate_phs, P_HYP_prim_F, P_HYP_A_avg, P_HYP_Qsim_mode,
       P_H06_Smax, P_H06_alpha, P_H06_envfact, P_H06_S_ini, P_H06_c1, P_H06_c2, P_H06_exponent, P_H06_denominator,
       P_H06_c_compare, P_H06_frac_Sdead, P_H06_E_rel_ini,
       P_H06_I_Jan, P_H06_D_Jan = P_H06_I_Jan + 12,
       P_H06_purpose = P_H06_D_Jan + 12, P_H06_I_mem_F, P_H06_D_mem_F, P_H06_I_mem_L, P_H06_D_mem_L };

// mean of the first n entries of month m's ring in storage order newest -> oldest (Fortran sum(x(1:n))/n)
__device__ inline double ring_mean(const double *ring, int L, int head, int n) {
  double s = 0.0;
  int k = head;
  for (int i = 0; i < n; ++i) { s = s + ring[k]; k = k + 1 == L ? 0 : k + 1; }
  return s / n;
}

// Hanasaki memory of one family (inflow: lake_route.f90:227-276, demand: :288-331): the new value enters the ring of
// the current month, and the monthly means that can have changed are recomputed.  M = the family's twelve mutable
// monthly parameters (stride nL).  The reference never updates the November inflow (:262-265) but does update the
// November demand (:322).
__device__ inline void h06_memory(double *ring, int *head, int L, double *M, size_t nL, int ls, bool demand, int memL, double newval,
                                  int month, double dt, int calendarId) {
  const double secprday = 86400.0;
  const int L31 = (int)floor(memL * 31 * secprday / dt), L30 = (int)floor(memL * 30 * secprday / dt);
  const int LF = calendarId == 0 ? (int)floor(memL * 28 * secprday / dt) : (int)floor(memL * 28.25 * secprday / dt);
  const bool first = head[12] == 0;
  if (first) {
    for (int m = 0; m < 12; ++m) { head[m] = 0; const double v = M[(size_t)m * nL + ls]; for (int k = 0; k < L31; ++k) ring[(size_t)m * L + k] = v; }
    head[12] = 1;
  } else {
    const int m = month - 1;
    int hd = head[m] - 1; if (hd < 0) hd = L31 - 1;
    ring[(size_t)m * L + hd] = newval; head[m] = hd;
  }
  // the reference recomputes every monthly mean each step; only the row just shifted can change (all rows on the first call)
  for (int m = 0; m < 12; ++m) {
    if (!first && m != month - 1) continue;
    if (m == 10 && !demand) continue;
    const int n = (m == 1) ? LF : (m == 3 || m == 5 || m == 8 || m == 10) ? L30 : L31;
    M[(size_t)m * nL + ls] = ring_mean(ring + (size_t)m * L, L31, head[m], n);
  }
}

// returns REACH_Q; updates vol (REACH_VOL(1)), vol0, ele, wb
__device__ inline double lake_route(const MzrDev &d, int r, int t, int ls, const double *Qrow, double qlat,
                                    double &vol, double &vol0, double &ele, double &wb, double &wmAct, bool coherent = false) {
  const int nL = d.nLake;
  auto P = [&](int p) -> double { return d.lakePar[(size_t)p * nL + ls]; };
  double *mut = d.lakeMut;                       // [25][nLake]
  const int type = d.lakeModel[ls];
  const double dt = d.dt, secprday = 86400.0;
  const int month = d.calMonth[t], day = d.calDay[t];
  double q_up = 0.0;
  {
    const int nu = d.nUp[r], u0 = d.upStart[r];
    // coherent: the upstream discharge was written by another wavefront of the same launch (persistent KWT sweep)
    for (int i = 0; i < nu; ++i)
      q_up = q_up + (coherent ? __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)(Qrow + u0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                              : Qrow[u0 + i]);
  }
  const bool targ = d.lakeTarg && d.lakeTarg[ls] != 0;
  const double wmvol = (targ && d.lakeWmVol) ? d.lakeWmVol[(size_t)t * nL + ls] : 0.0;      // REACH_WM_VOL
  if (d.iTime0 + t + 1 == 1) {   // first step: jump start to the target volume, or cold start (lake_route.f90:139-158)
    if (d.volJumpstart && targ) vol = wmvol;
    else vol = type == 0 ? P(P_D03_S0) : type == 1 ? P(P_D03_MaxStorage) : type == 2 ? P(P_H06_Smax)
                                                                                 : (P(P_HYP_E_emr) - P(P_HYP_E_zero)) * P(P_HYP_A_avg);
  }
  vol0 = vol;
  vol = vol + q_up * dt;
  if (d.LakeInputOption == 1 || d.LakeInputOption == 2) vol = vol + qlat * dt;
  double evapo = 0.0, precip = 0.0;
  if (d.LakeInputOption == 0 || d.LakeInputOption == 2) {
    precip = d.lakePrecip[(size_t)t * nL + ls];
    evapo = d.lakeEvap[(size_t)t * nL + ls];
    vol = vol + precip * dt;
    if (vol > evapo * dt) vol = vol - evapo * dt;
    else { evapo = vol / dt; vol = 0.0; }
  } else {
    precip = d.lakePrecip[(size_t)t * nL + ls];
    evapo = d.lakeEvap[(size_t)t * nL + ls];
  }
  const double wmflux = (d.is_flux_wm && d.wm) ? d.wm[(size_t)t * d.N + r] : 0.0;
  wmAct = wmflux;
  if (d.is_flux_wm && wmflux != -9999.0) {
    if (wmflux <= 0) { vol = vol - wmflux * dt; }
    else if (wmflux * dt <= vol) { vol = vol - wmflux * dt; }
    else { wmAct = vol / dt; vol = 0.0; }
  }
  double Q = 0.0;
  if (targ) {               // the lake follows the given target volume, :197-205
    if (vol < wmvol) Q = 0;
    else { Q = (vol - wmvol) / dt; vol = wmvol; }
  } else if (type == 0) {
    Q = 0.0;
  } else if (type == 1) {   // Doll 2003, :208-224
    const double S0 = P(P_D03_S0);
    if ((vol - S0) > 0) Q = P(P_D03_Coefficient) * (vol - S0) * pow((vol - S0) / (P(P_D03_MaxStorage) - S0), P(P_D03_Power));
    else Q = 0;
    Q = Q / secprday;
    Q = fmin(Q, vol / dt);
    vol = vol - Q * dt;
  } else if (type == 2) {   // Hanasaki 2006, :225-370
    double *Im = mut, *Dm = mut + (size_t)12 * nL;   // Im[m*nL+ls]
    if (P(P_H06_I_mem_F) != 0.0 && d.lakeRing)       // inflow memory, :227-276
      h06_memory(d.lakeRing + (size_t)ls * 12 * d.lakeL, d.lakeHead + (size_t)ls * 13, d.lakeL, Im, nL, ls, false, (int)P(P_H06_I_mem_L), q_up, month, dt, d.calendarId);
    if (P(P_H06_D_mem_F) != 0.0 && d.lakeRingD && d.is_flux_wm && wmflux != -9999.0)      // demand memory, :288-331 (a demand cannot be negative)
      h06_memory(d.lakeRingD + (size_t)ls * 12 * d.lakeLD, d.lakeHeadD + (size_t)ls * 13, d.lakeLD, Dm, nL, ls, true, (int)P(P_H06_D_mem_L),
                 wmflux < 0 ? 0.0 : wmflux, month, dt, d.calendarId);
    double sI = 0.0, sD = 0.0;
    for (int m = 0; m < 12; ++m) { sI = sI + Im[(size_t)m * nL + ls]; sD = sD + Dm[(size_t)m * nL + ls]; }
    const double I_yearly = sI / 12, D_yearly = sD / 12;
    const double c = P(P_H06_Smax) / (I_yearly * 365 * secprday);
    int start_month = 0;
    for (int i = 1; i <= 12; ++i) if (I_yearly <= Im[(size_t)(i - 1) * nL + ls]) start_month = i + 1;
    double &E_rel = mut[(size_t)24 * nL + ls];
    if (month == start_month && day == 1) E_rel = vol / (P(P_H06_alpha) * P(P_H06_Smax));
    double target_r;
    const double Imon = Im[(size_t)(month - 1) * nL + ls], Dmon = Dm[(size_t)(month - 1) * nL + ls];
    if ((int)P(P_H06_purpose) == 1) {
      if (P(P_H06_envfact) * I_yearly <= D_yearly) target_r = Imon * P(P_H06_c1) + I_yearly * P(P_H06_c2) * (Dmon / D_yearly);
      else target_r = I_yearly + Dmon - D_yearly;
    } else {
      target_r = I_yearly;
    }
    if (c >= P(P_H06_c_compare)) {
      Q = target_r * E_rel;
    } else if (0 <= c && c < P(P_H06_c_compare)) {
      const double f = pow(c / P(P_H06_denominator), P(P_H06_exponent));
      Q = E_rel * target_r * f + q_up * (1 - f);
    }
    const double Sdead = P(P_H06_Smax) * P(P_H06_frac_Sdead);
    if (vol < Sdead) { Q = Q - (Sdead - vol) / dt; if (Q < 0) Q = 0; }
    else if (vol > P(P_H06_Smax)) Q = Q + (vol - P(P_H06_Smax)) / dt;
    vol = vol - Q * dt;
  } else {                  // HYPE, :371-400
    const double pi = 3.14159265359;   // public_var.f90:16
    ele = vol / P(P_HYP_A_avg) + P(P_HYP_E_zero);
    const double doy = (double)d.calDoy[t];
    const double F_sin = fmax(0.0, (1 + P(P_HYP_Qrate_amp) * sin(2 * pi * (doy + (int)P(P_HYP_Qrate_phs)) / 365)));
    const double F_lin = fmin(fmax((ele - P(P_HYP_E_min)) / (P(P_HYP_E_lim) - P(P_HYP_E_min)), 0.0), 1.0);
    const int F_prim = P(P_HYP_prim_F) != 0.0 ? 1 : 0;
    const double Q_prim = F_sin * F_lin * F_prim * P(P_HYP_Qrate_prim);
    double Q_spill = 0.0;
    if (ele > P(P_HYP_E_emr)) Q_spill = P(P_HYP_Qrate_emr) * pow(ele - P(P_HYP_E_emr), P(P_HYP_Erate_emr));
    const double Q_sim = P(P_HYP_Qsim_mode) != 0.0 ? Q_prim + Q_spill : fmax(Q_prim, Q_spill);
    Q = fmin(Q_sim, fmax(0.0, (ele - P(P_HYP_E_min)) * P(P_HYP_A_avg)) / dt);
    vol = vol - Q * dt;
  }
  {   // comp_reach_wb with lakeFlag, water_balance.f90:52-85
    const double dVol = vol - vol0;
    const double Qin = q_up * dt, Qlateral = qlat * dt, pr = precip * dt;
    const double Qout = -1.0 * Q * dt, Qtake = -1.0 * wmAct * dt, ev = -1.0 * evapo * dt;
    wb = dVol - (Qin + Qlateral + pr + Qtake + Qout + ev);
  }
  return Q;
}

}  // namespace mzr_lake
