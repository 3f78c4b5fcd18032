/* CPU oracle: driver, basin2reach, hillslope UH, SUM, IRF (test infrastructure; see mzr_oracle.h). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) { fprintf(stderr, "orc: out of memory\n"); abort(); }
  return p;
}

orc_t *orc_create(int N, int H, const int *downIndex, const int *upOff, const int *upIdx,
                  const int *upGood, const int *hruOff, const int *hruIdx, const double *hruW,
                  const double *par) {
  orc_t *o = (orc_t *)xcalloc(1, sizeof(orc_t));
  o->N = N; o->H = H;
  o->down = (int *)xcalloc(N, sizeof(int));
  o->upOff = (int *)xcalloc(N + 1, sizeof(int));
  o->nGood = (int *)xcalloc(N, sizeof(int));
  o->hruOff = (int *)xcalloc(N + 1, sizeof(int));
  memcpy(o->upOff, upOff, (N + 1) * sizeof(int));
  memcpy(o->hruOff, hruOff, (N + 1) * sizeof(int));
  int nUp = upOff[N], nHru = hruOff[N];
  o->upIdx = (int *)xcalloc(nUp, sizeof(int));
  o->upGood = (int *)xcalloc(nUp, sizeof(int));
  o->hruIdx = (int *)xcalloc(nHru, sizeof(int));
  o->hruW = (double *)xcalloc(nHru, sizeof(double));
  for (int i = 0; i < N; i++) o->down[i] = downIndex[i] > 0 ? downIndex[i] - 1 : -1;
  for (int e = 0; e < nUp; e++) { o->upIdx[e] = upIdx[e] - 1; o->upGood[e] = upGood[e] != 0; }
  for (int e = 0; e < nHru; e++) { o->hruIdx[e] = hruIdx[e] - 1; o->hruW[e] = hruW[e]; }
  for (int i = 0; i < N; i++) {
    int c = 0;
    for (int e = upOff[i]; e < upOff[i + 1]; e++) c += o->upGood[e];
    o->nGood[i] = c;   /* count(NETOPO_in(i)%goodBas) */
  }
  for (int p = 0; p < ORC_NPAR; p++) {
    o->par[p] = (double *)xcalloc(N, sizeof(double));
    memcpy(o->par[p], par + (size_t)p * N, N * sizeof(double));
  }
  /* processing order: any upstream->downstream order gives identical results because a reach
     reads only its immediate upstreams' current-step outputs (main_route.f90:356-405). */
  o->order = (int *)xcalloc(N, sizeof(int));
  int *indeg = (int *)xcalloc(N, sizeof(int));
  for (int i = 0; i < N; i++) indeg[i] = upOff[i + 1] - upOff[i];
  int head = 0, tail = 0;
  for (int i = 0; i < N; i++) if (indeg[i] == 0) o->order[tail++] = i;
  while (head < tail) {
    int r = o->order[head++];
    int d = o->down[r];
    if (d >= 0 && --indeg[d] == 0) o->order[tail++] = d;
  }
  free(indeg);
  if (tail != N) { snprintf(o->msg, sizeof o->msg, "orc_create/network has a cycle or dangling upstream"); }
  o->BASIN_QI = (double *)xcalloc(N, sizeof(double));
  o->BASIN_QR0 = (double *)xcalloc(N, sizeof(double));
  o->BASIN_QR1 = (double *)xcalloc(N, sizeof(double));
  o->REACH_WM_FLUX = (double *)xcalloc(N, sizeof(double));
  for (int m = 0; m < 6; m++) o->idx[m] = -1;
  o->hw_drain_point = 2; o->doesBasinRoute = 1;
  return o;
}

void orc_destroy(orc_t *o) {
  if (o) { free(o->gaugeReach); free(o->Qobs); free(o->Qelapsed); free(o->Qerror); free(o->BASIN_solute); free(o->BASIN_solute_inst); free(o->solute_future); free(o->sol_mass); free(o->sol_flux); }
  if (!o) return;
  free(o->down); free(o->upOff); free(o->upIdx); free(o->upGood); free(o->nGood);
  free(o->hruOff); free(o->hruIdx); free(o->hruW); free(o->order);
  for (int p = 0; p < ORC_NPAR; p++) free(o->par[p]);
  free(o->fracFuture); free(o->uhOff); free(o->uh);
  free(o->BASIN_QI); free(o->BASIN_QR0); free(o->BASIN_QR1); free(o->REACH_WM_FLUX);
  free(o->QFUTURE); free(o->route); free(o->QFUTURE_IRF); free(o->kw); free(o->nkw);
  free(o->molKW); free(o->molMC); free(o->molDW);
  free(o);
}

const char *orc_last_error(const orc_t *o) { return o->msg; }

/* cold-start state, init_model_data.f90:399-505 */
int orc_config(orc_t *o, double dt, int nRoutes, const int *methods, int doesBasinRoute,
               int hw_drain_point, double min_length_route, double runoffMin, int is_flux_wm) {
  int N = o->N;
  o->dt = dt; o->nRoutes = nRoutes; o->doesBasinRoute = doesBasinRoute;
  o->hw_drain_point = hw_drain_point; o->min_length_route = min_length_route;
  o->runoffMin = runoffMin; o->is_flux_wm = is_flux_wm;
  for (int m = 0; m < 6; m++) o->idx[m] = -1;
  for (int i = 0; i < nRoutes; i++) {
    if (methods[i] < 0 || methods[i] > 5) { snprintf(o->msg, sizeof o->msg, "route_network/routing method id expect digits 0-5"); return 81; }
    o->methods[i] = methods[i]; o->idx[methods[i]] = i;
  }
  free(o->route); o->route = (orc_hyd *)xcalloc((size_t)nRoutes * N, sizeof(orc_hyd));
  free(o->kw); free(o->nkw); o->kw = NULL; o->nkw = NULL;
  free(o->molKW); free(o->molMC); free(o->molDW); o->molKW = o->molMC = o->molDW = NULL;
  if (o->idx[ORC_KWT] >= 0) {
    o->kw = (orc_fpoint *)xcalloc((size_t)N * ORC_KWSTORE, sizeof(orc_fpoint));
    o->nkw = (int *)xcalloc(N, sizeof(int));
    for (int i = 0; i < N; i++) o->nkw[i] = -1;   /* KWAVE unallocated */
  }
  if (o->idx[ORC_KW] >= 0) o->molKW = (double *)xcalloc((size_t)N * ORC_NMOL_KW, sizeof(double));
  if (o->idx[ORC_MC] >= 0) o->molMC = (double *)xcalloc((size_t)N * ORC_NMOL_MC, sizeof(double));
  if (o->idx[ORC_DW] >= 0) o->molDW = (double *)xcalloc((size_t)N * ORC_NMOL_DW, sizeof(double));
  return 0;
}

int orc_set_uh(orc_t *o, int ntdhBas, const double *fracFuture, const int *uhOff, const double *uh) {
  int N = o->N;
  o->ntdhBas = ntdhBas;
  free(o->fracFuture); o->fracFuture = (double *)xcalloc(ntdhBas, sizeof(double));
  memcpy(o->fracFuture, fracFuture, ntdhBas * sizeof(double));
  free(o->QFUTURE); o->QFUTURE = (double *)xcalloc((size_t)N * ntdhBas, sizeof(double));
  free(o->uhOff); free(o->uh); free(o->QFUTURE_IRF);
  o->uhOff = (int *)xcalloc(N + 1, sizeof(int));
  if (uhOff) {
    memcpy(o->uhOff, uhOff, (N + 1) * sizeof(int));
    o->uh = (double *)xcalloc(uhOff[N], sizeof(double));
    memcpy(o->uh, uh, uhOff[N] * sizeof(double));
    o->QFUTURE_IRF = (double *)xcalloc(uhOff[N], sizeof(double));
  } else {
    o->uh = NULL; o->QFUTURE_IRF = NULL;
  }
  return 0;
}

/* process_remap.f90:319-422 (time_conv = length_conv = 1 are applied by the caller's units) */
static int basin2reach(orc_t *o, const double *basinRunoff, double *reachRunoff) {
  const double negRunoffTol = -1.e-3;   /* public_var.f90:31 */
  for (int r = 0; r < o->N; r++) {
    int n = o->hruOff[r + 1] - o->hruOff[r];
    if (n > 0) {
      double acc = 0.0;
      for (int e = o->hruOff[r]; e < o->hruOff[r + 1]; e++) {
        double ro = basinRunoff[o->hruIdx[e]];
        if (ro < negRunoffTol) {
          snprintf(o->msg, sizeof o->msg, "basin2reach/exceeded negative runoff tolerance for HRU %d", o->hruIdx[e] + 1);
          return 20;
        }
        acc = acc + o->hruW[e] * ro * 1.0 * 1.0;   /* *time_conv*length_conv */
      }
      if (acc < o->runoffMin) acc = o->runoffMin;
      reachRunoff[r] = acc * o->par[ORC_P_BASAREA][r];
    } else {
      reachRunoff[r] = o->runoffMin;
    }
  }
  return 0;
}

/* basinUH.f90:70-178 (hru_irf + irf_conv), non-lake reaches */
static void hru_irf(orc_t *o, int r) {
  int n = o->ntdhBas;
  double *qf = o->QFUTURE + (size_t)r * n;
  o->BASIN_QR0[r] = o->BASIN_QR1[r];
  const int lake = o->is_lake_sim && o->lakeSlot[r] >= 0;   /* impulse for lakes, basinUH.f90:116-119 */
  for (int j = 0; j < n; j++) qf[j] = qf[j] + (lake ? (j == 0 ? 1.0 : 0.0) : o->fracFuture[j]) * o->BASIN_QI[r];
  o->BASIN_QR1[r] = qf[0];
  for (int j = 1; j < n; j++) qf[j - 1] = qf[j];
  qf[n - 1] = 0.0;
  if (o->tracer) {   /* the same fold for the constituent, basinUH.f90:130-137 */
    double *sf = o->solute_future + (size_t)r * n;
    for (int j = 0; j < n; j++) sf[j] = sf[j] + (lake ? (j == 0 ? 1.0 : 0.0) : o->fracFuture[j]) * o->BASIN_solute_inst[r];
    o->BASIN_solute[r] = sf[0];
    for (int j = 1; j < n; j++) sf[j - 1] = sf[j];
    sf[n - 1] = 0.0;
  }
}

/* process_remap.f90:425-500 basin2reach_mass */
static int basin2reach_mass(orc_t *o, const double *basinSolute, double *reachSolute) {
  for (int r = 0; r < o->N; r++) {
    if (o->hruOff[r + 1] - o->hruOff[r] > 0) {
      double acc = 0.0;
      for (int e = o->hruOff[r]; e < o->hruOff[r + 1]; e++) {
        const double s = basinSolute[o->hruIdx[e]];
        if (s < 0.0) { snprintf(o->msg, sizeof o->msg, "basin2reach_mass/Negative solute mass flux: HRU = %d", o->hruIdx[e] + 1); return 20; }
        acc = acc + o->hruW[e] * s * o->time_conv_solute * o->mass_conv_solute;
      }
      reachSolute[r] = acc * o->par[ORC_P_BASAREA][r];
    }
  }
  return 0;
}

/* tracer.f90:43-207: constituent_rch + comp_mass_flux (the mass-balance check only prints) */
static void constituent_rch(orc_t *o, int r, int method) {
  const size_t ix = (size_t)o->idx[method];
  const int N = o->N;
  const int nUps = o->nGood[r];
  int isHW = 1;
  double Cupstream = 0.0, Clat = 0.0;
  if (nUps > 0) {
    isHW = 0;
    for (int i = 0; i < nUps; i++) {
      const int e = o->upOff[r] + i;
      if (!o->upGood[e]) continue;
      Cupstream = Cupstream + o->sol_flux[ix * N + o->upIdx[e]];
    }
    Clat = o->BASIN_solute[r];
  } else {
    if (o->hw_drain_point == 1) { Cupstream = Cupstream + o->BASIN_solute[r]; Clat = 0.0; }
    else if (o->hw_drain_point == 2) Clat = o->BASIN_solute[r];
  }
  orc_hyd *h = &HYD(o, method, r);
  double *mass = o->sol_mass + (ix * N + r) * 2;
  const double dt = o->dt;
  mass[0] = mass[1];
  if (!isHW || o->hw_drain_point == 1) {
    const double reach_mass = Cupstream * dt + mass[0];
    const double reach_vol = h->REACH_INFLOW * dt + h->REACH_VOL[0];
    double solute_per_vol = 0.0;
    if (reach_vol > 0.0) solute_per_vol = reach_mass / reach_vol;
    double solute_out = (h->REACH_Q - o->BASIN_QR1[r]) * solute_per_vol;
    const double max_outMass = mass[1] / dt + Cupstream;
    if (solute_out > max_outMass) { solute_out = max_outMass; mass[1] = 0; }
    else mass[1] = mass[1] + (Cupstream - solute_out) * dt;
    o->sol_flux[ix * N + r] = solute_out + Clat;
  } else {
    o->sol_flux[ix * N + r] = Clat;
    mass[1] = 0.0;
  }
}

int orc_set_tracer(orc_t *o, double time_conv_solute, double mass_conv_solute, int firstStep, const double *solute) {
  o->tracer = 1; o->time_conv_solute = time_conv_solute; o->mass_conv_solute = mass_conv_solute;
  o->soluteFirst = firstStep; o->solute = solute;
  if (!o->BASIN_solute) {
    const size_t N = o->N, R = o->nRoutes > 0 ? o->nRoutes : 1;
    o->BASIN_solute = (double *)xcalloc(N, sizeof(double)); o->BASIN_solute_inst = (double *)xcalloc(N, sizeof(double));
    o->solute_future = (double *)xcalloc(N * (o->ntdhBas > 0 ? o->ntdhBas : 1), sizeof(double));
    o->sol_mass = (double *)xcalloc(R * N * 2, sizeof(double)); o->sol_flux = (double *)xcalloc(R * N, sizeof(double));
  }
  return 0;
}
int orc_get_solute(const orc_t *o, int route, double *flux, double *mass) {
  if (!o->tracer || route < 0 || route >= o->nRoutes) return 1;
  for (int r = 0; r < o->N; r++) { if (flux) flux[r] = o->sol_flux[(size_t)route * o->N + r]; if (mass) mass[r] = o->sol_mass[((size_t)route * o->N + r) * 2 + 1]; }
  return 0;
}
const double *orc_basin_solute(const orc_t *o) { return o->BASIN_solute; }

/* accum_runoff.f90:32-93 */
int orc_sum_rch(orc_t *o, int r) {
  orc_hyd *h = &HYD(o, ORC_SUM, r);
  h->REACH_Q = o->BASIN_QR1[r];
  double q_upstream = 0.0;
  if (o->upOff[r + 1] > o->upOff[r]) {
    for (int e = o->upOff[r]; e < o->upOff[r + 1]; e++) q_upstream = q_upstream + HYD(o, ORC_SUM, o->upIdx[e]).REACH_Q;
    h->REACH_Q = h->REACH_Q + q_upstream;
  }
  return 0;
}

/* Shared preamble of irf_rch / mc_rch / dfw_rch / kw_rch: irf_route.f90:81-142, mc_route.f90:83-146,
   dfw_route.f90:87-149, kwe_route.f90:83-145 */
void orc_preamble(orc_t *o, int r, int method, double *q_upstream_out, double *q_upstream_mod_out,
                  double *Qlat_out, int *isHW_out) {
  orc_hyd *h = &HYD(o, method, r);
  int nUps = o->nGood[r];
  int isHW = 1;
  double q_upstream = 0.0, q_upstream_mod = 0.0, Qlat = 0.0;
  double Qabs = o->REACH_WM_FLUX[r];
  h->REACH_WM_FLUX_actual = o->REACH_WM_FLUX[r];
  h->REACH_VOL[0] = h->REACH_VOL[1];
  if (nUps > 0) {
    isHW = 0;
    for (int i = 0; i < nUps; i++) {        /* do iUps = 1,nUps ; cycle if .not.goodBas(iUps) */
      int e = o->upOff[r] + i;
      if (!o->upGood[e]) continue;
      q_upstream = q_upstream + HYD(o, method, o->upIdx[e]).REACH_Q;
    }
    q_upstream_mod = q_upstream;
    Qlat = o->BASIN_QR1[r];
  } else {
    if (o->hw_drain_point == 1) {
      q_upstream = q_upstream + o->BASIN_QR1[r];
      q_upstream_mod = q_upstream;
      Qlat = 0.0;
    } else if (o->hw_drain_point == 2) {
      q_upstream_mod = q_upstream;
      Qlat = o->BASIN_QR1[r];
    }
  }
  h->REACH_INFLOW = q_upstream;
  if (o->REACH_WM_FLUX[r] != ORC_REALMISSING && o->is_flux_wm) {
    double dt = o->dt;
    if (Qabs > 0) {
      if (h->REACH_VOL[1] / dt > Qabs) {
        h->REACH_VOL[1] = h->REACH_VOL[1] - Qabs * dt;
      } else {
        Qabs = Qabs - h->REACH_VOL[1] / dt;
        h->REACH_VOL[1] = 0.0;
        if (q_upstream > Qabs) {
          q_upstream_mod = q_upstream - Qabs;
        } else {
          Qabs = Qabs - q_upstream;
          q_upstream_mod = 0.0;
          if (Qlat > Qabs) {
            Qlat = Qlat - Qabs;
          } else {
            Qabs = Qabs - Qlat;
            Qlat = 0.0;
            h->REACH_WM_FLUX_actual = o->REACH_WM_FLUX[r] - Qabs;
          }
        }
      }
    } else {
      Qlat = Qlat - Qabs;
    }
  }
  *q_upstream_out = q_upstream; *q_upstream_mod_out = q_upstream_mod; *Qlat_out = Qlat; *isHW_out = isHW;
}

/* water_balance.f90:22-112 (non-lake) */
void orc_comp_reach_wb(orc_t *o, int r, int method, double Qupstream, double Qlat) {
  orc_hyd *h = &HYD(o, method, r);
  double dt = o->dt;
  double dVol = h->REACH_VOL[1] - h->REACH_VOL[0];
  double Qin = Qupstream * dt;
  double Qlateral = Qlat * dt;
  double precip = 0.0;
  double Qout = -1.0 * h->REACH_Q * dt;
  double Qtake_actual = -1.0 * h->REACH_WM_FLUX_actual * dt;
  double evapo = 0.0;
  h->WB = dVol - (Qin + Qlateral + precip + Qtake_actual + Qout + evapo);
}

/* data_assimilation.f90:28-97 */
static int direct_insertion(orc_t *o, int r, int method) {
  orc_hyd *h = &HYD(o, method, r);
  double *Qerror = &o->Qerror[(size_t)o->idx[method] * o->N + r];
  const int Qelapsed = o->Qelapsed[r], qBlendPeriod = o->qBlendPeriod;
  double Qcorrect;
  if (o->Qobs[r] > 0.0) *Qerror = h->REACH_Q - o->Qobs[r];     /* there is observation */
  if (Qelapsed > qBlendPeriod) *Qerror = 0.0;
  if (Qelapsed <= qBlendPeriod) {
    switch (o->QerrTrend) {
      case 1: Qcorrect = *Qerror; break;
      case 2: Qcorrect = *Qerror * (1.0 - (double)Qelapsed / (double)qBlendPeriod); break;
      case 3: {
        const double x0 = 0.25, y0 = (double)0.90f;      /* default-real literals in the reference (:78) */
        const double k = log(1.0 / y0 - 1.0) / (qBlendPeriod / 2.0 - qBlendPeriod * x0);
        Qcorrect = *Qerror / (1.0 + exp(-k * (1.0 * Qelapsed - qBlendPeriod / 2.0)));
        break;
      }
      case 4:
        if (*Qerror != 0.0) {
          const double k = log(0.1 / fabs(*Qerror)) / (1.0 * qBlendPeriod);
          Qcorrect = *Qerror * exp(k * Qelapsed);
        } else Qcorrect = 0.0;
        break;
      default:
        snprintf(o->msg, sizeof o->msg, "direct_insertion/discharge error trend model must be 1(const),2(liear), or 3(logistic)");
        return 81;
    }
  } else Qcorrect = 0.0;
  h->REACH_Q = fmax(h->REACH_Q - Qcorrect, 0.0);
  return 0;
}

/* the tail the four Eulerian solvers share (irf_route.f90:188-202, kwe_route.f90:183-197, mc_route.f90:183-197,
   dfw_route.f90:187-201): direct insertion when qmodOption = 1, the reach water balance only when it is off */
int orc_finish_rch(orc_t *o, int r, int method, double Qupstream, double Qlat) {
  if (o->qmodOption == 1) { const int ierr = direct_insertion(o, r, method); if (ierr) return ierr; }
  if (o->qmodOption == 0) orc_comp_reach_wb(o, r, method, Qupstream, Qlat);
  return 0;
}

int orc_set_da(orc_t *o, int qBlendPeriod, int QerrTrend, int nGauge, const int *gaugeReach, int firstStep,
               const int *obsHave, const double *obsVal) {
  o->qmodOption = 1; o->qBlendPeriod = qBlendPeriod; o->QerrTrend = QerrTrend; o->nGauge = nGauge; o->obsFirst = firstStep;
  free(o->gaugeReach); o->gaugeReach = (int *)xcalloc(nGauge > 0 ? nGauge : 1, sizeof(int));
  for (int g = 0; g < nGauge; g++) o->gaugeReach[g] = (gaugeReach[g] >= 1 && gaugeReach[g] <= o->N) ? gaugeReach[g] - 1 : -1;
  o->obsHave = obsHave; o->obsVal = obsVal;
  if (!o->Qobs) {
    o->Qobs = (double *)xcalloc(o->N, sizeof(double)); o->Qelapsed = (int *)xcalloc(o->N, sizeof(int));
    o->Qerror = (double *)xcalloc((size_t)(o->nRoutes > 0 ? o->nRoutes : 1) * o->N, sizeof(double));
  }
  return 0;
}

/* irf_route.f90:40-264 */
int orc_irf_rch(orc_t *o, int r) {
  orc_hyd *h = &HYD(o, ORC_IRF, r);
  double q_upstream, q_upstream_mod, Qlat; int isHW;
  orc_preamble(o, r, ORC_IRF, &q_upstream, &q_upstream_mod, &Qlat, &isHW);
  /* conv_upsbas_qr, irf_route.f90:210-264; called with q_upstream_mod */
  double qu = q_upstream_mod;
  int nTDH = o->uhOff[r + 1] - o->uhOff[r];
  double *qf = o->QFUTURE_IRF + o->uhOff[r];
  const double *uh = o->uh + o->uhOff[r];
  if (o->par[ORC_P_LENGTH][r] > o->min_length_route) {
    for (int j = 0; j < nTDH; j++) qf[j] = qf[j] + uh[j] * qu;
    /* `*0.999` is a default-real literal in the reference (irf_route.f90:245) */
    double lim = (fmax(0.0, h->REACH_VOL[1]) / o->dt + qu) * (double)0.999f;
    qf[0] = fmin(lim, qf[0]);
    h->REACH_VOL[1] = h->REACH_VOL[1] - (qf[0] - qu) * o->dt;
    h->REACH_Q = qf[0] + Qlat;
    for (int j = 1; j < nTDH; j++) qf[j - 1] = qf[j];   /* eoshift(shift=1) */
    qf[nTDH - 1] = 0.0;
  } else {
    for (int j = 0; j < nTDH; j++) qf[j] = 0.0;
    qf[0] = qu;
    h->REACH_Q = qf[0] + Qlat;
    h->REACH_VOL[0] = 0.0;
    h->REACH_VOL[1] = 0.0;
  }
  return orc_finish_rch(o, r, ORC_IRF, q_upstream, Qlat);
}

/* main_route.f90:29-268 + route_network 273-409 */
int orc_step(orc_t *o, double T0, double T1, const double *runoff, const double *wmflux) {
  return orc_step_lake(o, T0, T1, runoff, wmflux, NULL, NULL, 1, 1, 1);
}

int orc_step_lake(orc_t *o, double T0, double T1, const double *runoff, const double *wmflux,
                  const double *evap, const double *precip, int month, int day, int dayofyear) {
  int N = o->N, ierr = 0;
  o->iTime += 1; o->month = month; o->day = day; o->dayofyear = dayofyear;
  o->msg[0] = 0;
  if (o->is_flux_wm && wmflux) { for (int r = 0; r < N; r++) o->REACH_WM_FLUX[r] = wmflux[r]; }
  else { for (int r = 0; r < N; r++) o->REACH_WM_FLUX[r] = 0.0; }
  if (o->qmodOption == 1) {   /* main_route.f90:125-148: gauge observations of this step, or one more step since the last ones */
    const long long row = o->iTime - 1 - o->obsFirst;
    if (o->obsHave[row]) {
      for (int g = 0; g < o->nGauge; g++) {
        const int r = o->gaugeReach[g];
        if (r < 0) continue;
        const double qobs = o->obsVal[(size_t)row * o->nGauge + g];
        if ((qobs != qobs) || (qobs < 0)) continue;
        o->Qobs[r] = qobs; o->Qelapsed[r] = 0;
      }
    } else {
      for (int r = 0; r < N; r++) o->Qelapsed[r] = o->Qelapsed[r] + 1;
    }
  }
  double *reachRunoff = (double *)xcalloc(N, sizeof(double));
  ierr = basin2reach(o, runoff, reachRunoff);
  if (ierr) { free(reachRunoff); return ierr; }
  if (o->is_lake_sim) {   /* main_route.f90:172-200: evaporation and precipitation through the same mapping */
    if (!evap || !precip) { free(reachRunoff); snprintf(o->msg, sizeof o->msg, "main_routing/lake simulation needs evaporation and precipitation"); return 20; }
    ierr = basin2reach(o, evap, o->basinEvapo);
    if (!ierr) ierr = basin2reach(o, precip, o->basinPrecip);
    if (ierr) { free(reachRunoff); return ierr; }
  }
  double *reachSolute = NULL;
  if (o->tracer) {   /* main_route.f90:161-172 */
    reachSolute = (double *)xcalloc(N, sizeof(double));
    ierr = basin2reach_mass(o, o->solute + (size_t)(o->iTime - 1 - o->soluteFirst) * o->H, reachSolute);
    if (ierr) { free(reachRunoff); free(reachSolute); return ierr; }
  }
  if (o->doesBasinRoute == 1) {
    for (int r = 0; r < N; r++) {
      o->BASIN_QI[r] = reachRunoff[r];
      if (o->tracer) o->BASIN_solute_inst[r] = o->BASIN_QI[r] > 0 ? reachSolute[r] : 0.0;      /* :207-213 */
    }
    for (int r = 0; r < N; r++) hru_irf(o, r);
  } else {
    for (int r = 0; r < N; r++) { o->BASIN_QR0[r] = o->BASIN_QR1[r]; o->BASIN_QR1[r] = reachRunoff[r]; }
    if (o->tracer) for (int r = 0; r < N; r++) o->BASIN_solute[r] = o->BASIN_QR1[r] > 0 ? reachSolute[r] : 0.0;   /* :228-236 */
  }
  free(reachSolute);
  free(reachRunoff);
  o->w_in = o->w_up = o->w_out = o->n_head = o->n_route = o->n_edges = 0;
  for (int ix = 0; ix < o->nRoutes; ix++) {
    int m = o->methods[ix];
    for (int k = 0; k < N; k++) {
      int r = o->order[k];
      if (o->is_lake_sim && o->lakeSlot[r] >= 0 && m != ORC_SUM) {   /* main_route.f90:375-381 */
        ierr = orc_lake_route(o, r, m);
        if (ierr) return ierr;
        if (o->tracer) constituent_rch(o, r, m);
        continue;
      }
      switch (m) {
        case ORC_SUM: ierr = orc_sum_rch(o, r); break;
        case ORC_IRF: ierr = orc_irf_rch(o, r); break;
        case ORC_KWT: ierr = orc_kwt_rch(o, r, T0, T1); break;
        case ORC_KW:  ierr = orc_dw_rch(o, r, ORC_KW); break;
        case ORC_MC:  ierr = orc_mc_rch(o, r, T0, T1); break;
        case ORC_DW:  ierr = orc_dw_rch(o, r, ORC_DW); break;
      }
      if (ierr) return ierr;
      if (o->tracer && m != ORC_SUM) constituent_rch(o, r, m);      /* main_route.f90:392-401 */
    }
  }
  orc_hist_aggregate(o, runoff);
  return 0;
}

/* ---- history accumulation, histVars_data.f90:154-305 (aggregate / finalize / refresh).  The reference's module cannot be
   compiled here (it USEs the ParallelIO wrappers), so this restatement is NOT pinned by execution; the arithmetic is
   sum-then-divide in step order. */
void orc_hist_aggregate(orc_t *o, const double *basRunoff) {
  const int N = o->N, H = o->H, R = o->nRoutes;
  if (!o->h_bas) {
    o->h_bas = (double *)xcalloc(H, sizeof(double)); o->h_inst = (double *)xcalloc(N, sizeof(double)); o->h_dlay = (double *)xcalloc(N, sizeof(double));
    o->h_q = (double *)xcalloc((size_t)N * R, sizeof(double)); o->h_vol = (double *)xcalloc((size_t)N * R, sizeof(double));
    o->h_ele = (double *)xcalloc((size_t)N * R, sizeof(double)); o->h_flood = (double *)xcalloc((size_t)N * R, sizeof(double));
    o->h_inflow = (double *)xcalloc((size_t)N * R, sizeof(double));
  }
  o->h_nt += 1;
  for (int i = 0; i < H; i++) o->h_bas[i] = o->h_bas[i] + basRunoff[i];                 /* :196-198 */
  for (int r = 0; r < N; r++) o->h_inst[r] = o->h_inst[r] + o->BASIN_QI[r];             /* :201-203 */
  for (int r = 0; r < N; r++) o->h_dlay[r] = o->h_dlay[r] + o->BASIN_QR1[r];            /* :206-208 */
  for (int ix = 0; ix < R; ix++) for (int r = 0; r < N; r++) {                          /* :211-246 */
    const orc_hyd *h = &o->route[(size_t)ix * N + r];
    const size_t k = (size_t)ix * N + r;
    o->h_q[k] = o->h_q[k] + h->REACH_Q;
    o->h_vol[k] = h->REACH_VOL[1];
    o->h_flood[k] = o->h_flood[k] + h->FLOOD_VOL[1];
    o->h_ele[k] = o->h_ele[k] + h->REACH_ELE;
    o->h_inflow[k] = o->h_inflow[k] + h->REACH_INFLOW;
  }
}

/* finalize (:251-297) of one variable: which 0 discharge, 1 inflow, 2 height, 3 floodVolume, 4 volume (last), 10 instRunoff,
   11 dlayRunoff, 12 basRunoff [H] */
int orc_hist_get(orc_t *o, int route, int which, double *out) {
  if (!o->h_bas || o->h_nt < 1) return 1;
  const int N = o->N;
  const double nt = (double)o->h_nt;
  if (which == 12) { for (int i = 0; i < o->H; i++) out[i] = o->h_bas[i] / nt; return 0; }
  if (which == 10 || which == 11) { const double *s = which == 10 ? o->h_inst : o->h_dlay; for (int r = 0; r < N; r++) out[r] = s[r] / nt; return 0; }
  const double *s = which == 0 ? o->h_q : which == 1 ? o->h_inflow : which == 2 ? o->h_ele : which == 3 ? o->h_flood : o->h_vol;
  for (int r = 0; r < N; r++) out[r] = which == 4 ? s[(size_t)route * N + r] : s[(size_t)route * N + r] / nt;
  return 0;
}

void orc_hist_refresh(orc_t *o) {   /* :300-305 */
  if (!o->h_bas) return;
  const size_t NR = (size_t)o->N * o->nRoutes;
  memset(o->h_bas, 0, o->H * sizeof(double)); memset(o->h_inst, 0, o->N * sizeof(double)); memset(o->h_dlay, 0, o->N * sizeof(double));
  memset(o->h_q, 0, NR * sizeof(double)); memset(o->h_vol, 0, NR * sizeof(double)); memset(o->h_ele, 0, NR * sizeof(double));
  memset(o->h_flood, 0, NR * sizeof(double)); memset(o->h_inflow, 0, NR * sizeof(double));
  o->h_nt = 0;
}

int orc_run_wm(orc_t *o, int nSteps, double t_start, const double *runoff, const double *wmflux, double *Qout, double *volOut);
int orc_run(orc_t *o, int nSteps, double t_start, const double *runoff, double *Qout, double *volOut) {
  return orc_run_wm(o, nSteps, t_start, runoff, NULL, Qout, volOut);
}

int orc_run_wm(orc_t *o, int nSteps, double t_start, const double *runoff, const double *wmflux, double *Qout, double *volOut) {
  int N = o->N;
  for (int it = 0; it < nSteps; it++) {
    double T0 = t_start + (double)it * o->dt;
    double T1 = T0 + o->dt;
    int ierr = orc_step(o, T0, T1, runoff + (size_t)it * o->H, wmflux ? wmflux + (size_t)it * N : NULL);
    if (ierr) return ierr;
    for (int ix = 0; ix < o->nRoutes; ix++) {
      size_t base = ((size_t)it * o->nRoutes + ix) * N;
      if (Qout)   for (int r = 0; r < N; r++) Qout[base + r] = o->route[(size_t)ix * N + r].REACH_Q;
      if (volOut) for (int r = 0; r < N; r++) volOut[base + r] = o->route[(size_t)ix * N + r].REACH_VOL[1];
    }
  }
  return 0;
}

int orc_run_lake(orc_t *o, int nSteps, double t_start, const double *runoff, const double *wmflux,
                 const double *evap, const double *precip, const int *ymd, double *Qout, double *volOut) {
  static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  int N = o->N;
  for (int it = 0; it < nSteps; it++) {
    double T0 = t_start + (double)it * o->dt, T1 = T0 + o->dt;
    const int y = ymd[3 * it], mo = ymd[3 * it + 1], d = ymd[3 * it + 2];
    int doy = d;
    const int leap = o->calendarId == 1 && ((y % 4 == 0 && y % 100 != 0) || y % 400 == 0);
    for (int k = 0; k < mo - 1; k++) doy += mdays[k] + (k == 1 && leap ? 1 : 0);
    int ierr = orc_step_lake(o, T0, T1, runoff + (size_t)it * o->H, wmflux ? wmflux + (size_t)it * N : NULL,
                             evap + (size_t)it * o->H, precip + (size_t)it * o->H, mo, d, doy);
    if (ierr) return ierr;
    for (int ix = 0; ix < o->nRoutes; ix++) {
      size_t base = ((size_t)it * o->nRoutes + ix) * N;
      if (Qout)   for (int r = 0; r < N; r++) Qout[base + r] = o->route[(size_t)ix * N + r].REACH_Q;
      if (volOut) for (int r = 0; r < N; r++) volOut[base + r] = o->route[(size_t)ix * N + r].REACH_VOL[1];
    }
  }
  return 0;
}

int orc_get_flux(const orc_t *o, int route, int which, double *out) {
  int N = o->N;
  if (which >= ORC_F_BASIN_QR1) {
    const double *src = which == ORC_F_BASIN_QR1 ? o->BASIN_QR1 : which == ORC_F_BASIN_QR0 ? o->BASIN_QR0 : o->BASIN_QI;
    memcpy(out, src, N * sizeof(double));
    return 0;
  }
  if (route < 0 || route >= o->nRoutes) return 1;
  const orc_hyd *h = o->route + (size_t)route * N;
  for (int r = 0; r < N; r++) {
    switch (which) {
      case ORC_F_Q: out[r] = h[r].REACH_Q; break;
      case ORC_F_VOL0: out[r] = h[r].REACH_VOL[0]; break;
      case ORC_F_VOL1: out[r] = h[r].REACH_VOL[1]; break;
      case ORC_F_INFLOW: out[r] = h[r].REACH_INFLOW; break;
      case ORC_F_ELE: out[r] = h[r].REACH_ELE; break;
      case ORC_F_FLOODVOL: out[r] = h[r].FLOOD_VOL[1]; break;
      case ORC_F_WB: out[r] = h[r].WB; break;
      default: return 1;
    }
  }
  return 0;
}

int orc_get_kwt_state(const orc_t *o, int *nw, double *qf, double *ti, double *tr, int *rf) {
  if (!o->kw) return 1;
  for (int r = 0; r < o->N; r++) {
    int n = o->nkw[r] < 0 ? 0 : o->nkw[r];
    nw[r] = n;
    for (int k = 0; k < ORC_WCAP; k++) {
      size_t d = (size_t)r * ORC_WCAP + k;
      if (k < n) {
        const orc_fpoint *p = &o->kw[(size_t)r * ORC_KWSTORE + k];
        qf[d] = p->QF; ti[d] = p->TI; tr[d] = p->TR; rf[d] = p->RF;
      } else { qf[d] = ti[d] = tr[d] = ORC_REALMISSING; rf[d] = 0; }
    }
  }
  return 0;
}

int orc_set_kwt_state(orc_t *o, const int *nw, const double *qf, const double *ti, const double *tr, const int *rf) {
  if (!o->kw) return 1;
  for (int r = 0; r < o->N; r++) {
    o->nkw[r] = nw[r] > 0 ? nw[r] : -1;
    for (int k = 0; k < nw[r]; k++) {
      size_t s = (size_t)r * ORC_WCAP + k;
      orc_fpoint *p = &o->kw[(size_t)r * ORC_KWSTORE + k];
      p->QF = qf[s]; p->TI = ti[s]; p->TR = tr[s]; p->RF = rf[s];
    }
  }
  return 0;
}

int orc_get_irf_state(const orc_t *o, double *qfuture) {
  if (!o->QFUTURE_IRF) return 1;
  memcpy(qfuture, o->QFUTURE_IRF, o->uhOff[o->N] * sizeof(double));
  return 0;
}

int orc_get_mol_state(const orc_t *o, int method, double *q) {
  const double *src = method == ORC_KW ? o->molKW : method == ORC_MC ? o->molMC : method == ORC_DW ? o->molDW : NULL;
  int n = method == ORC_MC ? ORC_NMOL_MC : ORC_NMOL_KW;
  if (!src) return 1;
  memcpy(q, src, (size_t)o->N * n * sizeof(double));
  return 0;
}

int orc_get_basin_state(const orc_t *o, double *qfuture) {
  if (!o->QFUTURE) return 1;
  memcpy(qfuture, o->QFUTURE, (size_t)o->N * o->ntdhBas * sizeof(double));
  return 0;
}

int orc_get_kwt_paths(const orc_t *o, long long *out) {
  for (int i = 0; i < ORC_NPATHS; i++) out[i] = o->paths[i];
  return 0;
}

int orc_get_kwt_traffic(const orc_t *o, long long *w_in, long long *w_up, long long *w_out,
                        long long *n_head, long long *n_route, long long *n_edges) {
  *w_in = o->w_in; *w_up = o->w_up; *w_out = o->w_out;
  *n_head = o->n_head; *n_route = o->n_route; *n_edges = o->n_edges;
  return 0;
}
