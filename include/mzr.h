/*
 * mzr.h -- C-ABI of the MI355X-native river-routing hot path (libmzr_hip.so).
 *
 * Drop-in boundary for mizuRoute's per-timestep sweep.  The reference's plugin point is the
 * per-reach `base_route_rch%route` (route/build/src/base_route.f90:29-61) called from
 * `route_network` (main_route.f90:273-409); that granularity is useless for a GPU, so the cut is
 * one level up: one call per (domain, time window), replacing `main_route`
 * (main_route.f90:29-268) as called from `mpi_route` (mpi_process.f90:1217,1294).
 *
 * Conventions shared with the reference:
 *   - every real is FP64 (nrtype.f90:8), every index int32, reach/HRU indices are 1-based,
 *     a downstream index <= 0 marks an outlet (NETOPO%DREACHI);
 *   - per-reach arrays are in the CALLER's reach order (the order of NETOPO_in / RCHFLX_out);
 *     the library keeps its own level-sorted device layout and permutes at the boundary;
 *   - every entry point returns the reference's integer `ierr` (0 = ok; 20/30/40/60/81 keep the
 *     meaning they have in the Fortran sources) and mzr_last_error() returns the message chain.
 *
 * All pointers are plain host pointers unless the name ends in `_dev`; nothing crosses the
 * boundary but pointers, sizes and scalars (bindable with ISO_C_BINDING, see
 * mizuroute_amd/fortran/mzr_c.f90 and INTEGRATION.md).  A handle is bound to one HIP device and is
 * not thread-safe (like the reference: one host thread per rank enters main_route).
 */
#ifndef MZR_H
#define MZR_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mzr_domain *mzr_handle;

/* routing method ids == digits of <route_opt>, public_var.f90:73-80 */
enum { MZR_SUM = 0, MZR_IRF = 1, MZR_KWT = 2, MZR_KW = 3, MZR_MC = 4, MZR_DW = 5 };

/* flux selectors (fields of dataTypes.f90:346-377 STRFLX / hydraulic) */
enum { MZR_F_Q = 0, MZR_F_VOL0 = 1, MZR_F_VOL1 = 2, MZR_F_INFLOW = 3, MZR_F_ELE = 4,
       MZR_F_FLOODVOL = 5, MZR_F_WB = 6, MZR_F_BASIN_QR1 = 7, MZR_F_BASIN_QR0 = 8, MZR_F_BASIN_QI = 9 };

#define MZR_MAXQPAR 20     /* public_var.f90:36 */
#define MZR_WCAP    32     /* padded wave capacity of mzr_get/set_kwt_state rows */
#define MZR_MAX_UPSTREAM 8 /* immediate upstream reaches per reach supported by the KWT merge */

/* Mirrors the public_var / globalData knobs the hot path reads (read_control.f90:118-600). */
typedef struct mzr_config {
  double dt;                  /* <dt_qsim> simulation time step [s]                        */
  int    nRoutes;             /* number of active methods                                  */
  int    routeMethods[6];     /* their ids, in <route_opt> order (flux slot = position)    */
  int    doesBasinRoute;      /* 1: hillslope UH delay (basinUH.f90), 0: runoff is routed  */
  int    hw_drain_point;      /* 1 top_reach, 2 bottom_reach (default)                     */
  double min_length_route;    /* pass-through below this reach length [m]                  */
  double runoffMin;           /* public_var.f90:145                                        */
  double negRunoffTol;        /* public_var.f90:31 (-1e-3)                                 */
  double time_conv, length_conv;  /* <units_qsim> conversion to m/s (read_control.f90:458-469) */
  int    maxWindow;           /* largest number of time steps per mzr_run call             */
  int    device;              /* HIP device ordinal                                        */
  int    is_flux_wm;          /* 1: water-management abstraction/injection fluxes are applied */
  int    lakeMemoryPerMethod; /* Hanasaki reservoirs with inflow / demand memory (lake_route.f90:227-331) keep mutable parameters
                                 (I_months, D_months, E_rel_ini) that the reference holds ONCE per lake in RPARAM: with several
                                 active routing methods every method feeds and reads the same memory, once per method and step
                                 (:258-276,360), so its methods interact.  Here the methods run side by side, each with a copy
                                 of its own.  0 (default): mzr_set_lakes refuses that combination (ierr 20) instead of silently
                                 deviating; 1: accept per-method copies knowingly.  One method, or no memory: no difference. */
  double mcTailTol;           /* Muskingum-Cunge: once the outflow of a sub-step (mc_route.f90:246-330) moves by less than
                                 this fraction of itself and the differences contract, the rest of the sub-step sum is
                                 added in closed form.  Default 1e-7 (deviation from iterating on: <= 1.4e-10 of the
                                 discharge, measured); 0 = iterate every sub-step, the reference's arithmetic.         */
  double sweepShare;          /* share of the device's wavefront slots the persistent sweeps of THIS handle may fill (default
                                 1).  Handles whose windows run side by side on one GPU (a tributary and the mainstem
                                 domain of rank 0) must share them out: the grids of concurrently running sweeps must
                                 fit the device together                                               */
  int    stepBatch;           /* mzr_step: 1 (default) = every call routes its step and returns its ierr; n > 1 = up to n
                                 steps (and at most maxWindow) are put aside and routed as one window when the batch is
                                 full or anything else is asked of the handle; errors then surface at that later call.
                                 One-step calls (nSteps = 1, host pointers) of mzr_set_lake_forcing, mzr_set_wm_flux,
                                 mzr_set_wm_vol, mzr_set_obs, mzr_set_solute made before a mzr_step travel with that step */
  int    sweepPriority;       /* 0 (default) / 1: the wavefronts of THIS handle's persistent sweeps run at the highest wave
                                 priority throughout.  For a small, deep domain that sweeps beside a large one on the same GPU
                                 (rank 0's mainstem beside its tributaries): its window is one long chain of dependent passes
                                 and a hundredth of the work, so its passes go first wherever they meet the other sweep's  */
  double sweepTimeout;        /* seconds without any progress on the reaches it waits for after which a wavefront of a
                                 persistent sweep gives up with ierr 93 instead of hanging the device.  0 (default): the KWT
                                 sweep takes four times the window's expected duration, between 1 s and 8 s; the sweeps of the
                                 Eulerian methods 8 s */
} mzr_config;

void mzr_default_config(mzr_config *cfg);

/* replaces init_route_method (init_model_data.f90:753-805) */
int mzr_create(const mzr_config *cfg, mzr_handle *out);
int mzr_destroy(mzr_handle h);
int mzr_last_error(mzr_handle h, char *buf, int len);

/* replaces put_data_struct's topology part (process_ntopo.f90:354-504): NETOPO%DREACHI, UREACHI,
   goodBas, HRUIX, HRUWGT, REACHID.  upGood may be NULL (all .true.). */
int mzr_set_network(mzr_handle h, int nRch, int nHru, const int *downIndex, const int *upOffset,
                    const int *upIndex, const int *upGood, const int *hruOffset, const int *hruIndex,
                    const double *hruWeight, const int *reachId);
/* RPARAM fields by name: R_SLOPE R_MAN_N R_WIDTH R_DEPTH RLENGTH R_STORAGE SIDE_SLOPE FLDP_SLOPE
   BASAREA TOTAREA MINFLOW (dataTypes.f90:183-195).  Set them before mzr_init_state: with KWT active a
   later change invalidates the state (derived per-reach constants are packed at initialisation) */
int mzr_set_param(mzr_handle h, const char *name, const double *values);
/* NETOPO%UH per reach (process_param.f90:99-262 make_uh), CSR by reach */
int mzr_set_uh(mzr_handle h, const int *uhOffset, const double *uh);
/* FRAC_FUTURE (process_param.f90:13-92 basinUH) */
int mzr_set_frac_future(mzr_handle h, int n, const double *frac);
/* Sub-basin partitioning (replaces the MPI domain decomposition, mpi_process.f90:473-606,1245-1329):
   exportReach[] = reaches of THIS domain (tributary outlets) whose per-step boundary records are
   shipped to the domain that owns their downstream reach; haloReach[] = reaches of this domain that
   stand for tributary outlets computed elsewhere (no upstreams, no HRUs here; haloGood bit 0 = their
   count(goodBas) > 0 in the full network, bit 1 (value 2) = the reach is a lake where it is routed: the reach below
   it then takes the lake's outflow as one particle and must have no other upstream reach, kwt_route.f90:540-559).  Indices are 1-based, caller's reach order.  Call
   after mzr_set_network, before mzr_init_state. */
int mzr_set_boundary(mzr_handle h, int nExport, const int *exportReach, int nHalo, const int *haloReach,
                     const int *haloGood);
/* number of doubles of a boundary record of nReach reaches over nSteps steps (the per-partition wire format; it carries what the
   importing domain reads and nothing else -- mpi_process.f90:1245-1329 ships the outlets' fluxes):
   header[4] | Q[nRoutes][nSteps][nReach]
   | only when KWT is among routeMethods: BASIN_QR[nSteps+1][nReach] | obN[nSteps][nReach] | obQ[nSteps][21][nReach] | obT[nSteps][21][nReach]
   | only while constituent routing is on (mzr_set_tracer: the same on EVERY domain, before the first mzr_boundary_size):
     reach_solute_flux[nRoutes][nSteps][nReach].
   header = {format tag (layout version), nRoutes, nSteps, nReach (+ 2^30 with the constituent, + 2^31 with the KWT part)}: written by
   mzr_export_boundary_dev, checked by mzr_import_boundary_dev -- a record that is not what the importing domain expects
   (other methods, window length, reach count, constituent on one side only) raises ierr 20 at the next synchronisation
   and nothing of it is used */
long long mzr_boundary_size(mzr_handle h, int nSteps, int nReach);
/* pack the export reaches' records of the last window into rec_dev (device memory) */
int mzr_export_boundary_dev(mzr_handle h, double *rec_dev);
/* Windows of the Eulerian methods overlap (the last nStages - 1 launches of a window go out with the first launches of the next one),
   and mzr_export_boundary_dev has to issue the launches kept back before it can pack the last window's record: correct, but the
   overlap is lost.  A tributary domain keeps it by exporting one window later:
     mzr_run_dev(window k); if (mzr_get_export_lag(h)) mzr_export_boundary_prev_dev(h, rec)  -> the record of window k - 1,
   packed on a stream of the library's own as soon as window k - 1 is complete (launch nStages - 1 of window k), mzr_wait_export(h)
   = the host waits for that record and for nothing else.  mzr_get_export_lag: 1 while the last window's final launches are kept
   back (same window length and options as the window before: the rule of DESIGN.md 2.6); the record of the LAST window of a run
   comes from mzr_export_boundary_dev as before.  mzr_export_boundary_prev_dev refuses (ierr 20) when the rows of the window
   before the last one are not kept (no overlap, or already exported).  (mpi_process.f90:1281-1312: the reference ships the outlet
   fluxes of every step before the mainstem's step.) */
int mzr_get_export_lag(mzr_handle h);
int mzr_export_boundary_prev_dev(mzr_handle h, double *rec_dev);
int mzr_wait_export(mzr_handle h);
/* unpack a record of nSrc reaches (one source partition) into halo slots [haloBase, haloBase+nSrc)
   for the next window of nSteps steps */
int mzr_import_boundary_dev(mzr_handle h, int nSteps, const double *rec_dev, int nSrc, int haloBase);
/* The host waits until the handle's last mzr_import_boundary_dev has read its record (rec_dev may then be reused or freed) and for
   nothing else.  Windows of a mainstem domain of the Eulerian methods overlap like any other domain's (round 6: the imported discharge
   is kept twice, the next window's beside that of the window whose last launches are kept back), and mzr_sync -- which issues the
   launches kept back -- between the import and the next mzr_run* would undo that: mzr_wait_import does not.  (The reference's rank 0
   sweeps the mainstem once per step after the gather, mpi_process.f90:1281-1312.) */
int mzr_wait_import(mzr_handle h);

/* Lakes and reservoirs (lake_route.f90:28-472; <is_lake_sim> = T).  lakeReach[nLake] 1-based;
   modelType 0 endorheic, 1 Doll03, 2 Hanasaki06, 3 HYPE (lake_route.f90:14-17);
   par[MZR_NLAKEPAR][nLake], rows in the order of RPARAM's lake fields (dataTypes.f90:196-254):
   D03_MaxStorage D03_Coefficient D03_Power D03_S0 | HYP_E_emr E_lim E_min E_zero Qrate_emr Erate_emr
   Qrate_prim Qrate_amp Qrate_phs prim_F A_avg Qsim_mode | H06_Smax alpha envfact S_ini c1 c2 exponent
   denominator c_compare frac_Sdead E_rel_ini | H06_I_Jan..Dec | H06_D_Jan..Dec | H06_purpose I_mem_F
   D_mem_F I_mem_L D_mem_L.   calendarId 0 noleap/365_day, 1 standard/gregorian.
   The Hanasaki inflow and demand memories (H06_I_mem_F, H06_D_mem_F; the demand is REACH_WM_FLUX, so it needs
   is_flux_wm) and target-volume lakes (mzr_set_lake_target) are included.
   With several routing methods each method keeps its own copy of the mutable Hanasaki
   parameters (the reference shares them through RPARAM, so its methods interact).  Call after mzr_set_network. */
#define MZR_NLAKEPAR 56
int mzr_set_lakes(mzr_handle h, int LakeInputOption, int calendarId, int nLake, const int *lakeReach,
                  const int *modelType, const double *par);
/* lake forcing of the NEXT window: evaporation and precipitation [nSteps][nHru] in the runoff units,
   and month / day / day-of-year of simDatetime(1) for every step */
int mzr_set_lake_forcing(mzr_handle h, int nSteps, const double *evap, const double *precip,
                         const int *month, const int *day, const int *dayofyear);
/* the same with evaporation and precipitation already in device memory (e.g. written by mzr_remap_runoff_dev, so that
   the two fluxes take the runoff's path from the file to the lakes without visiting the host); the calendar stays on
   the host.  With LakeInputOption = 1 neither flux is used: both entry points accept NULL for them. */
int mzr_set_lake_forcing_dev(mzr_handle h, int nSteps, const double *evap_dev, const double *precip_dev,
                             const int *month, const int *day, const int *dayofyear);
/* Lakes that follow a target volume instead of their parametric release (is_vol_wm; NETOPO%LakeTargVol,
   lake_route.f90:197-205): targVol[nLake] flags, jumpstart = is_vol_wm_jumpstart (the first step starts from the
   target, :140-142).  The targets of a window, REACH_WM_VOL [nSteps][nRch] in the caller's reach order
   (main_route.f90:115-122; only lake reaches are read), are set before every mzr_run / mzr_step while a flag is on. */
int mzr_set_lake_target(mzr_handle h, const int *targVol, int jumpstart);
int mzr_set_wm_vol(mzr_handle h, int nSteps, const double *vol);

/* Constituent routing (public_var tracer = T; main_route.f90:161-172,204-236,392-401, basinUH.f90:130-137,
   tracer.f90:43-207): a conservative constituent enters with the runoff as a mass flux per HRU and step
   (x time_conv_solute x mass_conv_solute x basin area), takes the hillslope delay, and is routed with the water of every
   active method except the runoff accumulation.  mzr_set_tracer after mzr_init_state (on = 0: off; in partitioned domains the
   flux of the tributary outlets travels with the boundary record); mzr_set_solute hands over solute[nSteps][nHru] (order of the runoff) before every window; after a window
   mzr_get_solute gives reach_solute_flux of the last step (which = 0) or reach_solute_mass(1) (which = 1) and
   mzr_get_window_solute the flux of every step, out[nSteps][nRch] (method < 0: BASIN_solute, the lateral mass flux into
   the reaches after the hillslope delay). */
int mzr_set_tracer(mzr_handle h, int on, double time_conv_solute, double mass_conv_solute);
int mzr_set_solute(mzr_handle h, int nSteps, const double *solute);
int mzr_get_solute(mzr_handle h, int method, int which, double *out);
int mzr_get_window_solute(mzr_handle h, int method, double *out);
/* constituent state for restarts (tfuture(seg, tdh), solute_mass(seg): write_restart_pio.f90:941-971,1292-): tfuture
   [nRch][ntdhBas] in the layout of mzr_get_basin_state, mass [nRch] of one method; a null pointer leaves that part out */
int mzr_get_tracer_state(mzr_handle h, int method, double *tfuture, double *mass);
int mzr_set_tracer_state(mzr_handle h, int method, const double *tfuture, const double *mass);

/* Direct insertion of gauge observations (data assimilation; public_var qmodOption = 1 with qBlendPeriod and QerrTrend,
   main_route.f90:125-148, data_assimilation.f90:28-97): after IRF, KW, MC or DW have routed a reach, the error against
   the last observed discharge -- constant (QerrTrend 1), linearly (2), logistically (3) or exponentially (4) decaying
   over qBlendPeriod steps -- is taken off REACH_Q; with it on, the reach water balance is not evaluated (as in the
   reference).  gaugeReach: 1-based reach of every gauge in the caller's order, < 1 = not in this network.  Call after
   mzr_init_state; nGauge = 0 switches it off.  mzr_set_obs hands over the observations of the next window (like
   mzr_set_wm_flux): have[nSteps] = there is an observation time at this step, obs[nSteps][nGauge] with NaN or a
   negative value for "none at this gauge". */
int mzr_set_da(mzr_handle h, int qBlendPeriod, int QerrTrend, int nGauge, const int *gaugeReach);
int mzr_set_obs(mzr_handle h, int nSteps, const int *have, const double *obs);

/* cold start (init_model_data.f90:399-505); must follow the setters above */
int mzr_init_state(mzr_handle h);

/* One time step == one main_route call (main_route.f90:29): T0,T1 = TSEC(1:2). runoff[nHru].  With cfg.stepBatch > 1 the
   call only puts the step aside (see mzr_config); results are identical, bit for bit. */
int mzr_step(mzr_handle h, double T0, double T1, const double *runoff);
/* nSteps <= maxWindow steps in one call, time-skewed over the level schedule.
   runoff[nSteps][nHru]; step k covers [t_start + k*dt, t_start + (k+1)*dt]. */
int mzr_run(mzr_handle h, int nSteps, double t_start, const double *runoff);
/* same, runoff already resident in device memory (zero copy); launches are asynchronous on the
   handle's stream, errors surface at the next mzr_sync / mzr_get_* call.  runoff_dev has to stay as it is until that
   call returns (the kernels read it when they run; a window whose KWT sweep gave up is routed again from it, mzr_get_sweep_retries) */
int mzr_run_dev(mzr_handle h, int nSteps, double t_start, const double *runoff_dev);
/* same, runoff in HOST memory (page-locked for the copy to overlap), returns at once: the window is copied on a
   stream of its own into one of two device buffers while the window before is still being routed -- the loop of
   standalone/route_runoff.f90:80-108 (read forcing, route) with the read hidden.  The host buffer must stay
   unchanged until the next-but-one mzr_run_async returns, or until mzr_sync. */
int mzr_run_async(mzr_handle h, int nSteps, double t_start, const double *runoff);
/* same, runoff as the forcing files store it: single precision.  The reference reads its forcing through get_nc into
   real(dp) (standalone/read_runoff.f90:264-306), i.e. it widens every float of the file; here the widening happens on
   the device behind the copy, so a window crosses PCIe at half the bytes and the values routed are the same, bit for
   bit, whenever the file variable is a float. */
int mzr_run_async_f32(mzr_handle h, int nSteps, double t_start, const float *runoff);
int mzr_sync(mzr_handle h);

/* Forcing remap in front of basin2reach (get_basin_runoff.f90:86-98 -> process_remap.f90:32-316),
   for when the hydrologic model's runoff is not on the river-network HRUs.  The mapping-file arrays
   are passed as the reference holds them (dataTypes.f90:132-143; 1-based indices, -9999 =
   integerMissing):  kind 1 = runoff on n1 polygons (remap_1D_runoff; qhru_ix, and optionally
   qhru_id[nOverlap] / src_id[n1] for the reference's id check, ierr 20);  kind 2 = runoff on an
   n1 x n2 grid stored like the reference's sim2d(n1,n2) (remap_2D_runoff; i_index, j_index).
   hru_ix[nMap] = position of each mapping row's HRU in the runoff rows given to mzr_run*. */
int mzr_set_remap(mzr_handle h, int kind, int nMap, const int *hru_ix, const int *num_qhru, int nOverlap,
                  const int *qhru_ix, const int *i_index, const int *j_index, const double *weight,
                  int n1, int n2, const long long *qhru_id, const long long *src_id);
/* runoff already on the river-network HRUs but in file order (sort_flux, process_remap.f90:268-316):
   ix_in[nSrc] = 1-based HRU position of every file entry (-9999: not in the network) */
int mzr_set_sort_map(mzr_handle h, int nSrc, const int *ix_in, int remove_negatives);
/* src_dev [nSteps][n1] or [nSteps][n2][n1] -> dst_dev [nSteps][nHru], both in device memory,
   asynchronous on the handle's stream */
int mzr_remap_runoff_dev(mzr_handle h, int nSteps, const double *src_dev, double *dst_dev);
/* remap + mzr_run_dev in one call: the window's forcing never visits the host */
int mzr_run_src_dev(mzr_handle h, int nSteps, double t_start, const double *src_dev);
/* water-management flux REACH_WM_FLUX for the NEXT window: flux[nSteps][nRch] in m3/s, caller's
   reach order; > 0 abstraction, < 0 injection for IRF/KW/MC/DW (irf_route.f90:118-142), Qtake for
   KWT (kwt_route.f90:351-455); -9999 = missing.  Only read when cfg.is_flux_wm = 1
   (main_route.f90:110-116). */
int mzr_set_wm_flux(mzr_handle h, int nSteps, const double *flux);

/* latest value of a flux field, caller's reach order, out[nRch] */
int mzr_get_flux(mzr_handle h, int method, int which, double *out);
/* discharge of every step of the last window: out[nSteps][nRch] */
int mzr_get_window_q(mzr_handle h, int method, double *out);
/* device-side history accumulation (histVars_data.f90:154-246): mean REACH_Q since the last reset */
int mzr_get_mean_q(mzr_handle h, int method, double *out, int reset);
/* comp_global_wb (water_balance.f90:191-323) of the last routed step, whole domain of this handle [m3]:
   out8 = { dVol, lateral flow, lake precipitation, -actual water take, -lake evaporation, -outflow at the outlets,
   -demanded water take, error = [0] - ([1]+...+[5]) }; the reference warns when |error| > 1 m3.  With several
   partitions the host adds the first seven numbers of all handles (shr_mpi_reduce in the reference). */
int mzr_get_global_wb(mzr_handle h, int method, double *out8);
/* The other history variables of the reference (histVars_data.f90:154-305; names popMetadat.f90:238-271):
   which sums are accumulated on the device -- call before mzr_init_state.
     MZR_H_INFLOW  <M>inflow: mean REACH_INFLOW per method (outputInflow)
     MZR_H_HEIGHT  <M>height and <M>floodVolume: mean REACH_ELE and FLOOD_VOL(1) per method (floodplain)
     MZR_H_RUNOFF  instRunoff, dlayRunoff (mean BASIN_QI, BASIN_QR(1), per reach) and basRunoff (mean runoff per HRU)
   mzr_get_mean returns sum / steps since the last reset (finalize, :251-297), `which` = MZR_M_*; method is ignored for the
   three runoff means; MZR_M_BAS_RUNOFF is [nHru] in the caller's HRU order, everything else [nRch].  The last volume
   (<M>volume) is mzr_get_flux(..., MZR_F_VOL1).  mzr_reset_means = histVars%refresh (every sum, discharge included). */
#define MZR_H_INFLOW 1
#define MZR_H_HEIGHT 2
#define MZR_H_RUNOFF 4
#define MZR_M_Q 0
#define MZR_M_INFLOW 1
#define MZR_M_HEIGHT 2
#define MZR_M_FLOODVOL 3
#define MZR_M_INST_RUNOFF 10
#define MZR_M_DLAY_RUNOFF 11
#define MZR_M_BAS_RUNOFF 12
int mzr_set_history(mzr_handle h, int flags);
int mzr_get_mean(mzr_handle h, int method, int which, double *out);
int mzr_reset_means(mzr_handle h);

/* state in the restart-file layout (write_restart_pio.f90:1039-1290) */
int mzr_get_kwt_state(mzr_handle h, int *numWaves, double *qwave, double *tentry, double *texit, int *routed);
int mzr_set_kwt_state(mzr_handle h, const int *numWaves, const double *qwave, const double *tentry,
                      const double *texit, const int *routed);
int mzr_get_irf_state(mzr_handle h, double *qfuture /* CSR by uhOffset */);
int mzr_get_mol_state(mzr_handle h, int method, double *q /* [nRch][nMolecule] */);
int mzr_get_basin_state(mzr_handle h, double *qfuture /* [nRch][n] */);
/* restart (read_restart.f90:152-742): the same layouts back in; basin_q = BASIN_QR(1) (either pointer may be NULL) */
int mzr_set_irf_state(mzr_handle h, const double *qfuture /* CSR by uhOffset */);
int mzr_set_mol_state(mzr_handle h, int method, const double *q /* [nRch][nMolecule] */);
int mzr_set_basin_state(mzr_handle h, const double *qfuture /* [nRch][n] */, const double *basin_q /* [nRch] */);
int mzr_set_volume(mzr_handle h, int method, const double *vol /* [nRch] REACH_VOL(1) */);

/* schedule / measurement introspection */
int mzr_get_schedule(mzr_handle h, int *nStages, int *maxStageWidth);
/* KWT persistent sweep: wavefronts the sweep kernel is launched with (0 = not in use), wavefronts the device
   holds at once, items (blocks of reaches) dealt to them */
int mzr_get_sweep_info(mzr_handle h, int *nWaves, int *capacity, int *nItems);
/* Windows whose persistent KWT sweep gave up waiting (ierr 93) and that were routed again through one launch per stage, since
   mzr_create.  That happens at mzr_sync / any getter: the state the first failed window of the queue started from is kept (every
   kernel of the windows queued behind it returns at once after an error), that window is routed again, and the windows behind it are
   queued once more as they were -- which takes their forcing where it was: windows queued with mzr_run_dev, whose forcing the caller
   leaves unchanged in device memory until the next synchronisation returns.  The caller sees no error, only a slow window and a line
   on stderr.  Not taken back: domains that export a boundary record, lakes, water management, the constituent, a window in which
   another method went through a persistent sweep; across several windows only the plain KWT domain.  The error is reported then. */
int mzr_get_sweep_retries(mzr_handle h, long long *nRetries);
/* KWT persistent sweep, start of its wavefronts: how many of the last launch arrived and how many of them joined (a
   wavefront that starts more than 20 us (MZR_SWEEP_LATE_TICKS = 2000 ticks of the 100 MHz clock) after the first one of its launch leaves at once), and since
   mzr_init_state the number of wavefronts by start delay: hist32[k] counts delays below 2^k ticks of 10 ns */
int mzr_get_sweep_arrivals(mzr_handle h, int *arrivedLast, int *joinedLast, long long *hist32);
/* KWT persistent sweep, duration of its launches on the DEVICE's clock: the first wavefront of a launch to arrive and the last one
   to leave write the 100 MHz counter (s_memrealtime) into a pair of words of the launch's own; no host events involved (events
   with timing on a stream next to a persistent launch were measured to slow it).  ms[0..*n) = the latest min(maxN,
   launches since the last reset, 1024) launches, oldest first, in milliseconds (0 = the launch did not run to its end);
   synchronises the handle's stream; reset != 0 starts counting anew. */
int mzr_get_sweep_clock(mzr_handle h, int maxN, double *ms, int *n, int reset);
/* measurement modes (bit mask, default 0):
   1  kernel-time accounting of the routing sweep: launches and summed device time [ms] per method
      (HIP events around every stage launch on the handle's stream), read with mzr_get_timing;
   2  KWT particle-traffic counters (device atomics -- perturbs timing; use on a separate window),
      read with mzr_get_kwt_traffic */
int mzr_set_profiling(mzr_handle h, int mode);
int mzr_get_timing(mzr_handle h, int method, long long *nLaunches, double *kernel_ms, long long *reachSteps, int reset);
/* shortest and longest event-timed launch [ms] since the last reset (mode 1; 0 = none).  Timing-enabled events on the stream of a
   persistent sweep slow some of its windows (profiles/r04_experiments.md): the shortest launch is the undisturbed kernel */
int mzr_get_timing_range(mzr_handle h, int method, double *min_ms, double *max_ms, int reset);
/* particle traffic counters of the KWT sweep since the last reset (for the roofline model);
   counted only while profiling mode 2 is on */
int mzr_get_kwt_traffic(mzr_handle h, long long *w_in, long long *w_up, long long *w_out,
                        long long *n_head, long long *n_route, long long *n_edges, int reset);

/* ---- boundary-record transport between the partitions of one node: RCCL point-to-point over xGMI, inside the library,
   so that a host without torch (the Fortran driver of mpi_process.f90:1217-1329) can exchange the records of
   mzr_export_boundary_dev / mzr_import_boundary_dev.  RCCL is loaded at run time (librccl.so); nothing else of the
   library depends on it.  One process per GPU: rank 0 calls mzr_comm_unique_id and hands the 128 bytes to the other
   ranks by whatever the host has (MPI_Bcast in the reference's driver), then every rank calls mzr_comm_init.
   Transfers run on the communicator's own stream: a send starts behind the handle's last mzr_export_boundary_dev (not
   behind the window the handle has queued since), and what the handle queues after a receive (mzr_import_boundary_dev)
   waits for it.  Buffers must stay valid until mzr_comm_sync (or the handle's mzr_sync on the receiving side). */
typedef struct mzr_comm_s *mzr_comm;
int mzr_comm_unique_id(char id[128]);
int mzr_comm_init(int rank, int nRanks, const char id[128], int device, mzr_comm *out);
int mzr_comm_send(mzr_comm c, mzr_handle h, const double *dev, long long n, int peer);
int mzr_comm_recv(mzr_comm c, mzr_handle h, double *dev, long long n, int peer);
/* several receives as one group: the transfers run side by side, one xGMI link per peer */
int mzr_comm_recv_many(mzr_comm c, mzr_handle h, int nPeers, double *const *dev, const long long *n, const int *peers);
int mzr_comm_sync(mzr_comm c);
int mzr_comm_destroy(mzr_comm c);
/* message of the last failed mzr_comm_* call of this thread */
int mzr_comm_last_error(char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif
