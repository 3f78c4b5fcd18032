/* Internal definitions of the CPU oracle (test infrastructure; see mzr_oracle.h). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "mzr_oracle.h"

#define ORC_REALMISSING (-9999.0)   /* public_var.f90:45 */
#define ORC_NMOL_KW 20              /* init_model_data.f90:386-394 */
#define ORC_NMOL_MC 2
#define ORC_NMOL_DW 20
#define ORC_KWSTORE 40
#define ORC_NPATHS 9
#define ORC_NLAKEPAR 56             /* mizuroute_amd/casefile.py LAKE_PAR order */              /* storage per reach for KWAVE (size <= MAXQPAR+1) */

/* dataTypes.f90:291-302 (QM is always -9999 on this path and is not stored) */
typedef struct { double QF, TI, TR; int RF; } orc_fpoint;

/* dataTypes.f90:346-358, one per (reach, active method) */
typedef struct {
  double REACH_ELE, REACH_INFLOW, FLOOD_VOL[2], REACH_Q, REACH_VOL[2], REACH_WM_FLUX_actual, WB;
} orc_hyd;

struct orc {
  int N, H;
  /* topology (0-based internally) */
  int *down;            /* [N] downstream index or -1 */
  int *upOff, *upIdx;   /* CSR of immediate upstreams (UREACHI order) */
  int *upGood;          /* per edge goodBas flag */
  int *nGood;           /* [N] count(goodBas) */
  int *hruOff, *hruIdx; double *hruW;
  int *order;           /* [N] processing order, upstream -> downstream */
  double *par[ORC_NPAR];
  /* config */
  double dt, min_length_route, runoffMin;
  int nRoutes, methods[6], idx[6];   /* idx[method] = slot or -1 */
  int doesBasinRoute, hw_drain_point, is_flux_wm;
  /* unit hydrographs */
  int ntdhBas; double *fracFuture; int *uhOff; double *uh;
  /* fluxes / state */
  double *BASIN_QI, *BASIN_QR0, *BASIN_QR1, *REACH_WM_FLUX;
  double *QFUTURE;      /* [N][ntdhBas] hillslope */
  orc_hyd *route;       /* [nRoutes][N] */
  double *QFUTURE_IRF;  /* concatenated per uhOff */
  orc_fpoint *kw; int *nkw;   /* [N][ORC_KWSTORE]; nkw = -1 unallocated */
  double *molKW, *molMC, *molDW;
  /* lakes (lake_route.f90); lakeSlot[r] = -1 for river reaches */
  int is_lake_sim, LakeInputOption, calendarId, nLake;
  int *lakeSlot, *lakeModel, *lakeInlet;
  double *lakePar;           /* [nLake][ORC_NLAKEPAR] (mutable: H06 monthly means, E_rel_ini) */
  double *basinEvapo, *basinPrecip;   /* [N] m3/s */
  double **qpast, **dpast; int *qpastLen, *dpastLen;   /* Hanasaki memory [12][L] per lake */
  int h_nt; double *h_bas, *h_inst, *h_dlay, *h_q, *h_vol, *h_ele, *h_flood, *h_inflow;   /* history sums, histVars_data.f90 */
  int *lakeTarg, volJumpstart, wmVolFirst; const double *wmVol;   /* target-volume lakes: NETOPO%LakeTargVol, is_vol_wm_jumpstart, REACH_WM_VOL[step][N] */
  long long iTime; int month, day, dayofyear;
  /* direct insertion of gauge observations (qmodOption = 1; main_route.f90:125-148, data_assimilation.f90:28-97) */
  int qmodOption, qBlendPeriod, QerrTrend, nGauge, obsFirst; int *gaugeReach;   /* [nGauge] 0-based reach, -1 = not in the network */
  const int *obsHave; const double *obsVal;   /* [step], [step][nGauge]: is there an observation time at this step, and the values */
  double *Qobs, *Qerror; int *Qelapsed;       /* [N], [nRoutes][N], [N] */
  /* constituent routing (tracer = T; main_route.f90:161-172,204-236, basinUH.f90:130-137, tracer.f90:43-207) */
  int tracer, soluteFirst; double time_conv_solute, mass_conv_solute;
  const double *solute;                        /* [step][H] basin constituent mass flux */
  double *BASIN_solute, *BASIN_solute_inst, *solute_future;   /* [N], [N], [N][ntdhBas] */
  double *sol_mass, *sol_flux;                 /* [nRoutes][N][2] reach_solute_mass(0:1), [nRoutes][N] reach_solute_flux */
  /* KWT traffic statistics of the last step */
  long long w_in, w_up, w_out, n_head, n_route, n_edges;
  /* how often the less common branches of kwt_rch ran since creation (test coverage evidence):
     0 shock merges, 1 merged group leaving within the step, 2 merged group staying, 3 exit-time +1 s fixes,
     4 duplicate times dropped in qexmul, 5 remove_rch calls, 6 ... with more than 64 particles,
     7 confluences of more than two reaches merged, 8 first-particle T_START+1 fixes */
  long long paths[ORC_NPATHS];
  char msg[512];
};

#define HYD(o, m, r) ((o)->route[(size_t)(o)->idx[m] * (o)->N + (r)])

/* hydraulic.f90 */
double orc_Btop(double yin, double b, double zc, double zf, double bankDepth);
double orc_Pwet(double yin, double b, double zc, double zf, double bankDepth);
double orc_flow_area(double yin, double b, double zc, double zf, double bankDepth);
double orc_water_height(double flowArea, double b, double zc, double zf, double bankDepth);
double orc_flow_depth(double Qin, double b, double zc, double S, double n, double zf, double bankDepth);
double orc_celerity(double Qin, double y, double b, double zc, double S, double n, double zf, double bankDepth);
double orc_diffusivity(double Qin, double y, double b, double zc, double S, double n, double zf, double bankDepth);
/* advection_diffusion.f90 */
void orc_solve_ade(double L, int nMol, double dt_local, double FluxUpstream, double ck, double dk,
                   const double *FluxPrev, double *FluxSolved);

int orc_sum_rch(orc_t *o, int r);
int orc_irf_rch(orc_t *o, int r);
int orc_mc_rch(orc_t *o, int r, double T0, double T1);
int orc_dw_rch(orc_t *o, int r, int method);   /* ORC_DW or ORC_KW */
int orc_kwt_rch(orc_t *o, int r, double T0, double T1);
int orc_lake_route(orc_t *o, int r, int method);
/* shared preamble of irf/mc/dw/kw (irf_route.f90:81-142) */
void orc_preamble(orc_t *o, int r, int method, double *q_upstream, double *q_upstream_mod,
                  double *Qlat, int *isHW);
void orc_comp_reach_wb(orc_t *o, int r, int method, double Qupstream, double Qlat);
int orc_finish_rch(orc_t *o, int r, int method, double Qupstream, double Qlat);   /* direct insertion or water balance, e.g. irf_route.f90:188-202 */
void orc_hist_aggregate(orc_t *o, const double *basRunoff);
#endif
