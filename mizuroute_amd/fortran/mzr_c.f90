! ISO_C_BINDING interface to libmzr_hip.so (C-ABI: include/mzr.h).
!
! This is the module a mizuRoute maintainer adds to route/build/src/ to call the MI355X hot path
! from the Fortran driver: `mpi_route` (mpi_process.f90:1217,1294) calls mzr_step (or mzr_run for a
! window of steps) instead of `main_route` (main_route.f90:29).  See INTEGRATION.md.
! All arrays are contiguous, caller-owned, in the caller's reach order; indices are 1-based.
MODULE mzr_c
  USE, INTRINSIC :: iso_c_binding
  implicit none
  private

  integer(c_int), parameter, public :: MZR_SUM=0, MZR_IRF=1, MZR_KWT=2, MZR_KW=3, MZR_MC=4, MZR_DW=5
  integer(c_int), parameter, public :: MZR_F_Q=0, MZR_F_VOL0=1, MZR_F_VOL1=2, MZR_F_INFLOW=3, MZR_F_ELE=4, &
                                       MZR_F_FLOODVOL=5, MZR_F_WB=6, MZR_F_BASIN_QR1=7, MZR_F_BASIN_QR0=8, MZR_F_BASIN_QI=9
  integer(c_int), parameter, public :: MZR_WCAP=32

  ! struct mzr_config (include/mzr.h)
  type, bind(C), public :: mzr_config
    real(c_double)  :: dt
    integer(c_int)  :: nRoutes
    integer(c_int)  :: routeMethods(6)
    integer(c_int)  :: doesBasinRoute
    integer(c_int)  :: hw_drain_point
    real(c_double)  :: min_length_route
    real(c_double)  :: runoffMin
    real(c_double)  :: negRunoffTol
    real(c_double)  :: time_conv, length_conv
    integer(c_int)  :: maxWindow
    integer(c_int)  :: device
    integer(c_int)  :: is_flux_wm
    integer(c_int)  :: lakeMemoryPerMethod   ! 1: several methods with Hanasaki memory lakes keep per-method copies (refused otherwise)
    real(c_double)  :: mcTailTol       ! Muskingum-Cunge closed-form tail of the sub-step sum (0 = iterate every sub-step)
    real(c_double)  :: sweepShare      ! share of the device's wavefront slots this handle's persistent sweeps fill (1 = all)
    integer(c_int)  :: stepBatch       ! mzr_step: steps put aside and routed as one window (1 = every call routes its step)
    integer(c_int)  :: sweepPriority   ! 1: this handle's persistent sweeps run at the highest wave priority (rank 0's mainstem beside its tributaries)
    real(c_double)  :: sweepTimeout    ! seconds without progress before a persistent sweep gives up (ierr 93)
  end type mzr_config

  public :: mzr_default_config, mzr_create, mzr_destroy, mzr_last_error, mzr_set_network, mzr_set_param, &
            mzr_set_uh, mzr_set_frac_future, mzr_init_state, mzr_step, mzr_run, mzr_sync, mzr_get_flux, &
            mzr_get_window_q, mzr_get_mean_q, mzr_get_kwt_state, mzr_set_kwt_state, mzr_get_irf_state, &
            mzr_get_mol_state, mzr_get_basin_state, mzr_get_schedule, mzr_set_boundary, mzr_boundary_size, &
            mzr_export_boundary_dev, mzr_export_boundary_prev_dev, mzr_get_export_lag, mzr_wait_export, mzr_import_boundary_dev, mzr_wait_import, mzr_run_dev, mzr_set_wm_flux, &
            mzr_set_remap, mzr_set_sort_map, mzr_remap_runoff_dev, mzr_run_src_dev, &
            mzr_set_irf_state, mzr_set_mol_state, mzr_set_basin_state, mzr_set_volume, &
            mzr_set_lakes, mzr_set_lake_forcing, mzr_get_sweep_info, mzr_run_async, mzr_run_async_f32, &
            mzr_set_lake_target, mzr_set_wm_vol, mzr_comm_unique_id, mzr_comm_init, mzr_comm_send, mzr_comm_recv, mzr_comm_recv_many, mzr_comm_destroy, mzr_comm_last_error, mzr_comm_sync, &
            mzr_get_global_wb, mzr_set_lake_forcing_dev, mzr_set_da, mzr_set_obs, &
            mzr_set_tracer, mzr_set_solute, mzr_get_solute, mzr_get_window_solute, mzr_get_tracer_state, mzr_set_tracer_state, &
            mzr_set_history, mzr_get_mean, mzr_reset_means, mzr_get_sweep_arrivals, mzr_get_sweep_retries, mzr_get_sweep_clock
  public :: mzr_message

  INTERFACE
    subroutine mzr_default_config(cfg) bind(C, name='mzr_default_config')
      import :: mzr_config
      type(mzr_config), intent(out) :: cfg
    end subroutine
    integer(c_int) function mzr_create(cfg, h) bind(C, name='mzr_create')
      import :: mzr_config, c_ptr, c_int
      type(mzr_config), intent(in) :: cfg
      type(c_ptr), intent(out) :: h
    end function
    integer(c_int) function mzr_destroy(h) bind(C, name='mzr_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_last_error(h, buf, len) bind(C, name='mzr_last_error')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(out) :: buf(*)
      integer(c_int), value :: len
    end function
    integer(c_int) function mzr_set_network(h, nRch, nHru, downIndex, upOffset, upIndex, upGood, hruOffset, &
                                            hruIndex, hruWeight, reachId) bind(C, name='mzr_set_network')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nRch, nHru
      integer(c_int), intent(in) :: downIndex(*), upOffset(*), upIndex(*), upGood(*), hruOffset(*), hruIndex(*), reachId(*)
      real(c_double), intent(in) :: hruWeight(*)
    end function
    integer(c_int) function mzr_set_param(h, name, values) bind(C, name='mzr_set_param')
      import :: c_ptr, c_int, c_double, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*)
      real(c_double), intent(in) :: values(*)
    end function
    integer(c_int) function mzr_set_uh(h, uhOffset, uh) bind(C, name='mzr_set_uh')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), intent(in) :: uhOffset(*)
      real(c_double), intent(in) :: uh(*)
    end function
    integer(c_int) function mzr_set_frac_future(h, n, frac) bind(C, name='mzr_set_frac_future')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: n
      real(c_double), intent(in) :: frac(*)
    end function
    integer(c_int) function mzr_init_state(h) bind(C, name='mzr_init_state')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_step(h, T0, T1, runoff) bind(C, name='mzr_step')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: T0, T1
      real(c_double), intent(in) :: runoff(*)
    end function
    integer(c_int) function mzr_run(h, nSteps, t_start, runoff) bind(C, name='mzr_run')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), value :: t_start
      real(c_double), intent(in) :: runoff(*)
    end function
    integer(c_int) function mzr_sync(h) bind(C, name='mzr_sync')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_get_flux(h, method, which, out) bind(C, name='mzr_get_flux')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method, which
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_get_window_q(h, method, out) bind(C, name='mzr_get_window_q')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_get_mean_q(h, method, out, reset) bind(C, name='mzr_get_mean_q')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method, reset
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_set_tracer(h, on, time_conv_solute, mass_conv_solute) bind(C, name='mzr_set_tracer')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: on
      real(c_double), value :: time_conv_solute, mass_conv_solute
    end function
    integer(c_int) function mzr_set_solute(h, nSteps, solute) bind(C, name='mzr_set_solute')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), intent(in) :: solute(*)
    end function
    integer(c_int) function mzr_get_solute(h, method, which, out) bind(C, name='mzr_get_solute')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method, which
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_get_window_solute(h, method, out) bind(C, name='mzr_get_window_solute')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_get_tracer_state(h, method, tfuture, mass) bind(C, name='mzr_get_tracer_state')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, tfuture, mass        ! c_loc of real(c_double) arrays, or c_null_ptr
      integer(c_int), value :: method
    end function
    integer(c_int) function mzr_set_tracer_state(h, method, tfuture, mass) bind(C, name='mzr_set_tracer_state')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, tfuture, mass
      integer(c_int), value :: method
    end function
    integer(c_int) function mzr_set_da(h, qBlendPeriod, QerrTrend, nGauge, gaugeReach) bind(C, name='mzr_set_da')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: qBlendPeriod, QerrTrend, nGauge
      integer(c_int), intent(in) :: gaugeReach(*)
    end function
    integer(c_int) function mzr_set_obs(h, nSteps, have, obs) bind(C, name='mzr_set_obs')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      integer(c_int), intent(in) :: have(*)
      real(c_double), intent(in) :: obs(*)
    end function
    integer(c_int) function mzr_get_global_wb(h, method, out8) bind(C, name='mzr_get_global_wb')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(out) :: out8(8)
    end function
    integer(c_int) function mzr_get_kwt_state(h, numWaves, qwave, tentry, texit, routed) bind(C, name='mzr_get_kwt_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), intent(out) :: numWaves(*), routed(*)
      real(c_double), intent(out) :: qwave(*), tentry(*), texit(*)
    end function
    integer(c_int) function mzr_set_kwt_state(h, numWaves, qwave, tentry, texit, routed) bind(C, name='mzr_set_kwt_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), intent(in) :: numWaves(*), routed(*)
      real(c_double), intent(in) :: qwave(*), tentry(*), texit(*)
    end function
    integer(c_int) function mzr_get_irf_state(h, qfuture) bind(C, name='mzr_get_irf_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: qfuture(*)
    end function
    integer(c_int) function mzr_get_mol_state(h, method, q) bind(C, name='mzr_get_mol_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(out) :: q(*)
    end function
    integer(c_int) function mzr_get_basin_state(h, qfuture) bind(C, name='mzr_get_basin_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: qfuture(*)
    end function
    integer(c_int) function mzr_get_schedule(h, nStages, maxStageWidth) bind(C, name='mzr_get_schedule')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), intent(out) :: nStages, maxStageWidth
    end function
    ! history sums beyond discharge (histVars_data.f90:154-305): flags MZR_H_INFLOW = 1, MZR_H_HEIGHT = 2, MZR_H_RUNOFF = 4, before
    ! mzr_init_state; which = MZR_M_Q 0, INFLOW 1, HEIGHT 2, FLOODVOL 3, INST_RUNOFF 10, DLAY_RUNOFF 11, BAS_RUNOFF 12 (per HRU)
    integer(c_int) function mzr_set_history(h, flags) bind(C, name='mzr_set_history')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: flags
    end function
    integer(c_int) function mzr_get_mean(h, method, which, out) bind(C, name='mzr_get_mean')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method, which
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function mzr_reset_means(h) bind(C, name='mzr_reset_means')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    ! windows whose persistent KWT sweep gave up waiting (ierr 93) and that were routed again through one launch per stage
    integer(c_int) function mzr_get_sweep_retries(h, nRetries) bind(C, name='mzr_get_sweep_retries')
      import :: c_ptr, c_int, c_long_long
      type(c_ptr), value :: h
      integer(c_long_long), intent(out) :: nRetries
    end function
    ! durations [ms] of the latest KWT sweep launches on the device's own clock
    integer(c_int) function mzr_get_sweep_clock(h, maxN, ms, n, reset) bind(C, name='mzr_get_sweep_clock')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: maxN, reset
      real(c_double), intent(out) :: ms(*)
      integer(c_int), intent(out) :: n
    end function
    integer(c_int) function mzr_get_sweep_arrivals(h, arrivedLast, joinedLast, hist32) bind(C, name='mzr_get_sweep_arrivals')
      import :: c_ptr, c_int, c_long_long
      type(c_ptr), value :: h
      integer(c_int), intent(out) :: arrivedLast, joinedLast
      integer(c_long_long), intent(out) :: hist32(32)
    end function
    integer(c_int) function mzr_get_sweep_info(h, nWaves, capacity, nItems) bind(C, name='mzr_get_sweep_info')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), intent(out) :: nWaves, capacity, nItems
    end function
    integer(c_int) function mzr_set_boundary(h, nExport, exportReach, nHalo, haloReach, haloGood) bind(C, name='mzr_set_boundary')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: nExport, nHalo
      integer(c_int), intent(in) :: exportReach(*), haloReach(*), haloGood(*)
    end function
    integer(c_long_long) function mzr_boundary_size(h, nSteps, nReach) bind(C, name='mzr_boundary_size')
      import :: c_ptr, c_int, c_long_long
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps, nReach
    end function
    integer(c_int) function mzr_export_boundary_dev(h, rec_dev) bind(C, name='mzr_export_boundary_dev')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, rec_dev
    end function
    ! overlapping windows of a tributary domain: the record of the window BEFORE the last one (include/mzr.h)
    integer(c_int) function mzr_export_boundary_prev_dev(h, rec_dev) bind(C, name='mzr_export_boundary_prev_dev')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, rec_dev
    end function
    integer(c_int) function mzr_get_export_lag(h) bind(C, name='mzr_get_export_lag')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_wait_export(h) bind(C, name='mzr_wait_export')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_import_boundary_dev(h, nSteps, rec_dev, nSrc, haloBase) bind(C, name='mzr_import_boundary_dev')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, rec_dev
      integer(c_int), value :: nSteps, nSrc, haloBase
    end function
    integer(c_int) function mzr_wait_import(h) bind(C, name='mzr_wait_import')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function
    integer(c_int) function mzr_set_wm_flux(h, nSteps, flux) bind(C, name='mzr_set_wm_flux')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), intent(in) :: flux(*)
    end function
    integer(c_int) function mzr_set_lakes(h, LakeInputOption, calendarId, nLake, lakeReach, modelType, par) bind(C, name='mzr_set_lakes')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: LakeInputOption, calendarId, nLake
      integer(c_int), intent(in) :: lakeReach(*), modelType(*)
      real(c_double), intent(in) :: par(*)
    end function
    integer(c_int) function mzr_set_lake_forcing(h, nSteps, evap, precip, month, day, dayofyear) bind(C, name='mzr_set_lake_forcing')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), intent(in) :: evap(*), precip(*)
      integer(c_int), intent(in) :: month(*), day(*), dayofyear(*)
    end function
    integer(c_int) function mzr_set_lake_forcing_dev(h, nSteps, evap_dev, precip_dev, month, day, dayofyear) bind(C, name='mzr_set_lake_forcing_dev')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, evap_dev, precip_dev      ! device pointers (c_null_ptr with LakeInputOption = 1)
      integer(c_int), value :: nSteps
      integer(c_int), intent(in) :: month(*), day(*), dayofyear(*)
    end function
    ! lakes that follow a target volume (is_vol_wm; lake_route.f90:139-142,197-205) and their targets REACH_WM_VOL per window
    integer(c_int) function mzr_set_lake_target(h, targVol, jumpstart) bind(C, name='mzr_set_lake_target')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), intent(in) :: targVol(*)
      integer(c_int), value :: jumpstart
    end function
    integer(c_int) function mzr_set_wm_vol(h, nSteps, vol) bind(C, name='mzr_set_wm_vol')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), intent(in) :: vol(*)
    end function
    integer(c_int) function mzr_run_dev(h, nSteps, t_start, runoff_dev) bind(C, name='mzr_run_dev')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h, runoff_dev
      integer(c_int), value :: nSteps
      real(c_double), value :: t_start
    end function
    integer(c_int) function mzr_run_async(h, nSteps, t_start, runoff) bind(C, name='mzr_run_async')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), value :: t_start
      real(c_double), intent(in) :: runoff(*)
    end function
    ! the same with the forcing as the files store it (real(sp), what get_nc widens into real(dp): read_runoff.f90:264-306)
    integer(c_int) function mzr_run_async_f32(h, nSteps, t_start, runoff) bind(C, name='mzr_run_async_f32')
      import :: c_ptr, c_int, c_double, c_float
      type(c_ptr), value :: h
      integer(c_int), value :: nSteps
      real(c_double), value :: t_start
      real(c_float), intent(in) :: runoff(*)
    end function
    ! boundary-record transport between partitions (RCCL point-to-point inside the library; replaces the gather / scatter
    ! of mpi_route, mpi_process.f90:1245-1329): id from rank 0 to every rank with the host's MPI_Bcast, then mzr_comm_init
    integer(c_int) function mzr_comm_unique_id(id) bind(C, name='mzr_comm_unique_id')
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: id(128)
    end function
    integer(c_int) function mzr_comm_init(rank, nRanks, id, device, comm) bind(C, name='mzr_comm_init')
      import :: c_int, c_char, c_ptr
      integer(c_int), value :: rank, nRanks, device
      character(kind=c_char), intent(in) :: id(128)
      type(c_ptr), intent(out) :: comm
    end function
    integer(c_int) function mzr_comm_send(comm, h, dev, n, peer) bind(C, name='mzr_comm_send')
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm, h, dev
      integer(c_long_long), value :: n
      integer(c_int), value :: peer
    end function
    integer(c_int) function mzr_comm_recv(comm, h, dev, n, peer) bind(C, name='mzr_comm_recv')
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm, h, dev
      integer(c_long_long), value :: n
      integer(c_int), value :: peer
    end function
    integer(c_int) function mzr_comm_recv_many(comm, h, nPeers, dev, n, peers) bind(C, name='mzr_comm_recv_many')
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm, h
      integer(c_int), value :: nPeers
      type(c_ptr), intent(in) :: dev(*)
      integer(c_long_long), intent(in) :: n(*)
      integer(c_int), intent(in) :: peers(*)
    end function
    integer(c_int) function mzr_comm_sync(comm) bind(C, name='mzr_comm_sync')
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
    end function
    integer(c_int) function mzr_comm_destroy(comm) bind(C, name='mzr_comm_destroy')
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
    end function
    integer(c_int) function mzr_comm_last_error(buf, len) bind(C, name='mzr_comm_last_error')
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: buf(*)
      integer(c_int), value :: len
    end function
    ! restart (read_restart.f90:152-742): state back in, the layouts of the getters
    integer(c_int) function mzr_set_irf_state(h, qfuture) bind(C, name='mzr_set_irf_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: qfuture(*)
    end function
    integer(c_int) function mzr_set_mol_state(h, method, q) bind(C, name='mzr_set_mol_state')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(in) :: q(*)
    end function
    integer(c_int) function mzr_set_basin_state(h, qfuture, basin_q) bind(C, name='mzr_set_basin_state')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, qfuture, basin_q
    end function
    integer(c_int) function mzr_set_volume(h, method, vol) bind(C, name='mzr_set_volume')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: method
      real(c_double), intent(in) :: vol(*)
    end function
    ! forcing remap (process_remap.f90:32-316): remap_data as the reference holds it; pass c_null_ptr for the
    ! index arrays of the other kind and for the optional id arrays
    integer(c_int) function mzr_set_remap(h, kind, nMap, hru_ix, num_qhru, nOverlap, qhru_ix, i_index, j_index, weight, &
                                          n1, n2, qhru_id, src_id) bind(C, name='mzr_set_remap')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h, qhru_ix, i_index, j_index, qhru_id, src_id
      integer(c_int), value :: kind, nMap, nOverlap, n1, n2
      integer(c_int), intent(in) :: hru_ix(*), num_qhru(*)
      real(c_double), intent(in) :: weight(*)
    end function
    integer(c_int) function mzr_set_sort_map(h, nSrc, ix_in, remove_negatives) bind(C, name='mzr_set_sort_map')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: nSrc, remove_negatives
      integer(c_int), intent(in) :: ix_in(*)
    end function
    integer(c_int) function mzr_remap_runoff_dev(h, nSteps, src_dev, dst_dev) bind(C, name='mzr_remap_runoff_dev')
      import :: c_ptr, c_int
      type(c_ptr), value :: h, src_dev, dst_dev
      integer(c_int), value :: nSteps
    end function
    integer(c_int) function mzr_run_src_dev(h, nSteps, t_start, src_dev) bind(C, name='mzr_run_src_dev')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h, src_dev
      integer(c_int), value :: nSteps
      real(c_double), value :: t_start
    end function
  END INTERFACE

CONTAINS

  ! the reference's (ierr, message) convention: message chain of the last failing call
  FUNCTION mzr_message(h) result(msg)
    type(c_ptr), intent(in) :: h
    character(len=512) :: msg
    character(kind=c_char) :: buf(512)
    integer :: i, rc
    msg = ''
    rc = mzr_last_error(h, buf, 512_c_int)
    do i = 1, 512
      if (buf(i) == c_null_char) exit
      msg(i:i) = buf(i)
    end do
  END FUNCTION mzr_message

END MODULE mzr_c
