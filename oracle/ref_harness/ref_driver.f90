! Harness driver for the reference routing hot path (test infrastructure; not product code).
!
! Reads a binary "case" file written by oracle/casefile.py, fills the reference's
! own derived types the way put_data_struct / init_state_data do
! (process_ntopo.f90:354-504, init_model_data.f90:399-505), calls the UNMODIFIED reference
! `main_route` (main_route.f90:29) once per time step and dumps per-step discharge/volume plus the
! final solver state.  No routing arithmetic lives in this file.
!
! usage: ref_route <case.bin> <out.bin> [nthreads]
PROGRAM ref_driver
  USE nrtype
  USE public_var
  USE dataTypes,  ONLY: RCHTOPO, RCHPRP, STRFLX, STRSTA, subbasin_omp, dlength
  USE datetime_data, ONLY: datetime
  USE globalData, ONLY: simDatetime
  USE globalData, ONLY: rch_routes, nRoutes, routeMethods, onRoute, &
                        idxSUM, idxIRF, idxKWT, idxKW, idxMC, idxDW, &
                        iTime, TSEC, nMolecule, isColdStart, FRAC_FUTURE, &
                        time_conv, length_conv, maxtdh, nThreads, time_conv_solute, mass_conv_solute
  USE obs_data,   ONLY: gageObs
  USE model_utils, ONLY: harness_last_err, harness_last_msg
  USE process_param, ONLY: basinUH, make_uh
  USE main_route_module,   ONLY: main_route
  USE accum_runoff_module, ONLY: accum_runoff_rch
  USE irf_route_module,    ONLY: irf_route_rch
  USE kwt_route_module,    ONLY: kwt_route_rch
  USE kw_route_module,     ONLY: kwe_route_rch
  USE mc_route_module,     ONLY: mc_route_rch
  USE dfw_route_module,    ONLY: dfw_route_rch
  USE omp_lib
  implicit none

  integer(i4b), parameter :: MAGIC_IN  = 1297765955   ! 'MZRC' bytes read as little-endian int32
  integer(i4b), parameter :: MAGIC_DA  = 1145133645   ! 'MZAD': gauge-observation section
  integer(i4b), parameter :: MAGIC_TR  = 1381259853   ! 'MZTR': constituent section
  integer(i4b) :: daOn, daMagic, ios, nGauge, daBlend, daTrend, trOn, utr
  real(dp), allocatable :: solute(:,:), trflux(:,:), trmass(:,:), trbas(:)
  integer(i4b), parameter :: MAGIC_OUT = 1297765967   ! 'MZRO'
  integer(i4b), parameter :: WCAP = 32                ! padded wave capacity in the state dump

  character(len=1024) :: fcase, fout, arg
  integer(i4b) :: uin, uout, magic, version
  integer(i4b) :: N, H, nSteps, methodsIn(6), nUpTot, nHruTot, nOrder, nBranch
  integer(i4b) :: uhSource, ntdhBasIn, nUhTotIn, dumpEvery, isFluxWm, isLakeSim, calendarId, nLake, il, isVolWm, volJump
  integer(i4b), allocatable :: lakeTarg(:)
  real(dp), allocatable :: wmvol(:,:)
  integer(i4b), allocatable :: ymd(:,:), lakeReach(:), lakeModel(:)
  real(dp), allocatable :: lakePar(:,:), evap(:,:), precip(:,:)
  real(dp)     :: fshape, tscale, velo, diff, t_start
  integer(i4b), allocatable :: downIndex(:), reachId(:), upOffset(:), upIndex(:), upGood(:)
  integer(i4b), allocatable :: hruOffset(:), hruIndex(:), orderOffset(:), branchOffset(:), seg(:)
  integer(i4b), allocatable :: uhOffsetIn(:)
  real(dp), allocatable :: hruWeight(:), par(:,:), fracIn(:), uhIn(:), runoff(:,:), wmflux(:,:)
  real(dp), allocatable :: lengths(:)
  type(dlength), allocatable :: seg_uh(:)

  type(RCHTOPO), allocatable :: NETOPO(:)
  type(RCHPRP),  allocatable :: RPARAM(:)
  type(STRFLX),  allocatable :: RCHFLX(:)
  type(STRSTA),  allocatable :: RCHSTA(:)
  type(subbasin_omp), allocatable :: river_basin(:)
  type(gageObs) :: gage_obs
  integer(i4b), allocatable :: ixRch(:)
  real(dp), allocatable :: basinRunoff(:), basinEvapo(:), basinPrecip(:), basinSolute(:), reachflux(:), reachvol(:)
  real(dp), allocatable :: qout(:,:), volout(:,:), qr1(:)

  integer(i4b) :: ierr, i, j, k, ix, it, nu, nh, io, ib, nb, nr, nw, ntdh, first_err, first_err_step
  integer(i8b) :: c0, c1, crate
  real(dp)     :: wall
  integer      :: nSkip, envstat
  character(len=32) :: envbuf
  character(len=strLen) :: message
  real(dp), allocatable :: wbuf(:,:)
  integer(i4b), allocatable :: ibuf(:)

  call get_command_argument(1, fcase)
  call get_command_argument(2, fout)
  nThreads = 1
  if (command_argument_count() >= 3) then
    call get_command_argument(3, arg); read(arg,*) nThreads
  end if
  call omp_set_num_threads(nThreads)

  open(newunit=uin, file=trim(fcase), access='stream', form='unformatted', status='old', action='read')
  read(uin) magic, version
  if (magic /= MAGIC_IN) then
    write(*,*) 'bad case-file magic', magic; stop 3
  end if
  read(uin) N, H, nSteps, nRoutes, methodsIn, doesBasinRoute, hw_drain_point, nUpTot, nHruTot, &
            nOrder, nBranch, uhSource, ntdhBasIn, nUhTotIn, dumpEvery, isFluxWm, isLakeSim
  read(uin) dt, min_length_route, runoffMin, fshape, tscale, velo, diff, t_start
  allocate(downIndex(N), reachId(N), upOffset(N+1), upIndex(nUpTot), upGood(nUpTot))
  allocate(hruOffset(N+1), hruIndex(nHruTot), hruWeight(nHruTot), par(N,11))
  allocate(orderOffset(nOrder+1), branchOffset(nBranch+1), seg(N))
  read(uin) downIndex, reachId, upOffset, upIndex, upGood, hruOffset, hruIndex, hruWeight
  read(uin) par            ! columns: slope, man_n, width, depth, length, storage, sideSlope, fldpSlope, basArea, totArea, minflow
  read(uin) orderOffset, branchOffset, seg
  if (uhSource == 1) then
    allocate(fracIn(ntdhBasIn), uhOffsetIn(N+1), uhIn(nUhTotIn))
    read(uin) fracIn, uhOffsetIn, uhIn
  end if
  allocate(runoff(H, nSteps))
  read(uin) runoff
  if (isFluxWm == 1) then
    allocate(wmflux(N, nSteps))
    read(uin) wmflux
  end if
  isVolWm = 0
  if (isLakeSim == 2) then      ! lakes + a target-volume section
    isLakeSim = 1; isVolWm = 1
  end if
  if (isLakeSim == 1) then
    read(uin) LakeInputOption, calendarId, nLake
    allocate(ymd(3, nSteps), lakeReach(nLake), lakeModel(nLake), lakePar(nLake, 56), evap(H, nSteps), precip(H, nSteps))
    read(uin) ymd, lakeReach, lakeModel, lakePar, evap, precip
    if (calendarId == 0) then
      calendar = 'noleap'
    else
      calendar = 'standard'
    end if
    if (isVolWm == 1) then      ! NETOPO%LakeTargVol flags, is_vol_wm_jumpstart, REACH_WM_VOL per step
      allocate(lakeTarg(nLake), wmvol(N, nSteps))
      read(uin) lakeTarg
      read(uin) volJump
      read(uin) wmvol
    end if
  end if
  ! optional trailing section: gauge observations for direct insertion (qmodOption = 1)
  daOn = 0; trOn = 0
  do      ! trailing sections, each behind its magic
    read(uin, iostat=ios) daMagic
    if (ios /= 0) exit
    if (daMagic == MAGIC_DA) then
      daOn = 1
      read(uin) nGauge, daBlend, daTrend
      allocate(gage_obs%link(nGauge), gage_obs%have(nSteps), gage_obs%val(nGauge, nSteps))
      read(uin) gage_obs%link
      read(uin) gage_obs%have
      read(uin) gage_obs%val
      where (gage_obs%link < 1 .or. gage_obs%link > N) gage_obs%link = integerMissing
    else if (daMagic == MAGIC_TR) then      ! constituent: lateral mass flux per HRU and step
      trOn = 1
      allocate(solute(H, nSteps))
      read(uin) solute
    else
      exit
    end if
  end do
  close(uin)

  ! ---- configuration the reference keeps in public_var / globalData (read_control.f90:580-600)
  is_lake_sim = (isLakeSim == 1); is_flux_wm = (isFluxWm == 1); is_vol_wm = (isVolWm == 1); tracer = (trOn == 1)
  time_conv_solute = 1._dp; mass_conv_solute = 1._dp
  is_vol_wm_jumpstart = .false.
  if (isVolWm == 1) is_vol_wm_jumpstart = (volJump /= 0)
  qmodOption = 0
  if (daOn == 1) then
    qmodOption = 1; qBlendPeriod = daBlend; QerrTrend = daTrend
  end if
  time_conv = 1._dp; length_conv = 1._dp
  allocate(routeMethods(nRoutes))
  routeMethods(1:nRoutes) = methodsIn(1:nRoutes)
  onRoute = .false.
  idxSUM=-1; idxIRF=-1; idxKWT=-1; idxKW=-1; idxMC=-1; idxDW=-1
  do ix = 1, nRoutes
    onRoute(routeMethods(ix)) = .true.
    select case (routeMethods(ix))
      case (accumRunoff);           idxSUM = ix
      case (impulseResponseFunc);   idxIRF = ix
      case (kinematicWaveTracking); idxKWT = ix
      case (kinematicWave);         idxKW  = ix
      case (muskingumCunge);        idxMC  = ix
      case (diffusiveWave);         idxDW  = ix
    end select
  end do
  nMolecule%KW_ROUTE = 20; nMolecule%MC_ROUTE = 2; nMolecule%DW_ROUTE = 20   ! init_model_data.f90:386-394

  ! ---- routing objects (init_model_data.f90:753-805)
  allocate(rch_routes(nRoutes))
  do ix = 1, nRoutes
    select case (routeMethods(ix))
      case (accumRunoff);           allocate(accum_runoff_rch :: rch_routes(ix)%rch_route)
      case (impulseResponseFunc);   allocate(irf_route_rch    :: rch_routes(ix)%rch_route)
      case (kinematicWaveTracking); allocate(kwt_route_rch    :: rch_routes(ix)%rch_route)
      case (kinematicWave);         allocate(kwe_route_rch    :: rch_routes(ix)%rch_route)
      case (muskingumCunge);        allocate(mc_route_rch     :: rch_routes(ix)%rch_route)
      case (diffusiveWave);         allocate(dfw_route_rch    :: rch_routes(ix)%rch_route)
    end select
  end do

  ! ---- unit hydrographs: reference routines (process_param.f90:13-92,99-262) or case-file arrays
  allocate(lengths(N)); lengths = par(:,5)
  if (uhSource == 1) then
    allocate(FRAC_FUTURE(ntdhBasIn)); FRAC_FUTURE = fracIn
  else
    call basinUH(dt, fshape, tscale, ierr, message)
    if (ierr/=0) then; write(*,*) trim(message); stop 4; end if
    call make_uh(lengths, dt, velo, diff, seg_uh, ierr, message)
    if (ierr/=0) then; write(*,*) trim(message); stop 4; end if
  end if

  ! ---- topology + parameters (process_ntopo.f90:354-504)
  allocate(NETOPO(N), RPARAM(N), RCHFLX(N), RCHSTA(N))
  do i = 1, N
    NETOPO(i)%REACHIX = i
    NETOPO(i)%REACHID = reachId(i)
    NETOPO(i)%DREACHI = downIndex(i)
    if (downIndex(i) > 0) then
      NETOPO(i)%DREACHK = reachId(downIndex(i))
    else
      NETOPO(i)%DREACHK = -1
    end if
    nu = upOffset(i+1) - upOffset(i)
    allocate(NETOPO(i)%UREACHI(nu), NETOPO(i)%UREACHK(nu), NETOPO(i)%goodBas(nu))
    do j = 1, nu
      NETOPO(i)%UREACHI(j) = upIndex(upOffset(i)+j)
      NETOPO(i)%UREACHK(j) = reachId(upIndex(upOffset(i)+j))
      NETOPO(i)%goodBas(j) = (upGood(upOffset(i)+j) /= 0)
    end do
    nh = hruOffset(i+1) - hruOffset(i)
    allocate(NETOPO(i)%HRUID(nh), NETOPO(i)%HRUIX(nh), NETOPO(i)%HRUWGT(nh))
    do j = 1, nh
      NETOPO(i)%HRUIX(j)  = hruIndex(hruOffset(i)+j)
      NETOPO(i)%HRUID(j)  = hruIndex(hruOffset(i)+j)
      NETOPO(i)%HRUWGT(j) = hruWeight(hruOffset(i)+j)
    end do
    NETOPO(i)%RHORDER = i
    NETOPO(i)%LAKINLT = .false.; NETOPO(i)%ISLAKE = .false.; NETOPO(i)%LAKETARGVOL = .false.
    NETOPO(i)%LAKEMODELTYPE = 0
    if (uhSource == 1) then
      ntdh = uhOffsetIn(i+1) - uhOffsetIn(i)
      allocate(NETOPO(i)%UH(ntdh)); NETOPO(i)%UH = uhIn(uhOffsetIn(i)+1:uhOffsetIn(i+1))
    else
      allocate(NETOPO(i)%UH(size(seg_uh(i)%dat))); NETOPO(i)%UH = seg_uh(i)%dat
    end if
    RPARAM(i)%R_SLOPE = par(i,1);  RPARAM(i)%R_MAN_N = par(i,2);   RPARAM(i)%R_WIDTH = par(i,3)
    RPARAM(i)%R_DEPTH = par(i,4);  RPARAM(i)%RLENGTH = par(i,5);   RPARAM(i)%R_STORAGE = par(i,6)
    RPARAM(i)%SIDE_SLOPE = par(i,7); RPARAM(i)%FLDP_SLOPE = par(i,8)
    RPARAM(i)%BASAREA = par(i,9);  RPARAM(i)%TOTAREA = par(i,10);  RPARAM(i)%MINFLOW = par(i,11)
    RPARAM(i)%UPSAREA = par(i,10) - par(i,9)
  end do

  ! ---- lakes: flags, parameters (process_ntopo.f90:479-494; lake inlet = immediate downstream is a lake,
  !      network_topo.f90:958-981)
  if (is_lake_sim) then
    do il = 1, nLake
      i = lakeReach(il)
      NETOPO(i)%ISLAKE = .true.
      NETOPO(i)%LAKEMODELTYPE = lakeModel(il)
      if (isVolWm == 1) NETOPO(i)%LAKETARGVOL = (lakeTarg(il) /= 0)
      if (onRoute(impulseResponseFunc)) then
        NETOPO(i)%UH = 0._dp; NETOPO(i)%UH(1) = 1._dp          ! process_ntopo.f90:501-505
      end if
      RPARAM(i)%D03_MaxStorage = lakePar(il,1);  RPARAM(i)%D03_Coefficient = lakePar(il,2)
      RPARAM(i)%D03_Power = lakePar(il,3);       RPARAM(i)%D03_S0 = lakePar(il,4)
      RPARAM(i)%HYP_E_emr = lakePar(il,5);       RPARAM(i)%HYP_E_lim = lakePar(il,6);   RPARAM(i)%HYP_E_min = lakePar(il,7)
      RPARAM(i)%HYP_E_zero = lakePar(il,8);      RPARAM(i)%HYP_Qrate_emr = lakePar(il,9); RPARAM(i)%HYP_Erate_emr = lakePar(il,10)
      RPARAM(i)%HYP_Qrate_prim = lakePar(il,11); RPARAM(i)%HYP_Qrate_amp = lakePar(il,12)
      RPARAM(i)%HYP_Qrate_phs = nint(lakePar(il,13)); RPARAM(i)%HYP_prim_F = (lakePar(il,14) /= 0._dp)
      RPARAM(i)%HYP_A_avg = lakePar(il,15);      RPARAM(i)%HYP_Qsim_mode = (lakePar(il,16) /= 0._dp)
      RPARAM(i)%H06_Smax = lakePar(il,17);       RPARAM(i)%H06_alpha = lakePar(il,18);  RPARAM(i)%H06_envfact = lakePar(il,19)
      RPARAM(i)%H06_S_ini = lakePar(il,20);      RPARAM(i)%H06_c1 = lakePar(il,21);     RPARAM(i)%H06_c2 = lakePar(il,22)
      RPARAM(i)%H06_exponent = lakePar(il,23);   RPARAM(i)%H06_denominator = lakePar(il,24); RPARAM(i)%H06_c_compare = lakePar(il,25)
      RPARAM(i)%H06_frac_Sdead = lakePar(il,26); RPARAM(i)%H06_E_rel_ini = lakePar(il,27)
      RPARAM(i)%H06_I_Jan = lakePar(il,28); RPARAM(i)%H06_I_Feb = lakePar(il,29); RPARAM(i)%H06_I_Mar = lakePar(il,30)
      RPARAM(i)%H06_I_Apr = lakePar(il,31); RPARAM(i)%H06_I_May = lakePar(il,32); RPARAM(i)%H06_I_Jun = lakePar(il,33)
      RPARAM(i)%H06_I_Jul = lakePar(il,34); RPARAM(i)%H06_I_Aug = lakePar(il,35); RPARAM(i)%H06_I_Sep = lakePar(il,36)
      RPARAM(i)%H06_I_Oct = lakePar(il,37); RPARAM(i)%H06_I_Nov = lakePar(il,38); RPARAM(i)%H06_I_Dec = lakePar(il,39)
      RPARAM(i)%H06_D_Jan = lakePar(il,40); RPARAM(i)%H06_D_Feb = lakePar(il,41); RPARAM(i)%H06_D_Mar = lakePar(il,42)
      RPARAM(i)%H06_D_Apr = lakePar(il,43); RPARAM(i)%H06_D_May = lakePar(il,44); RPARAM(i)%H06_D_Jun = lakePar(il,45)
      RPARAM(i)%H06_D_Jul = lakePar(il,46); RPARAM(i)%H06_D_Aug = lakePar(il,47); RPARAM(i)%H06_D_Sep = lakePar(il,48)
      RPARAM(i)%H06_D_Oct = lakePar(il,49); RPARAM(i)%H06_D_Nov = lakePar(il,50); RPARAM(i)%H06_D_Dec = lakePar(il,51)
      RPARAM(i)%H06_purpose = nint(lakePar(il,52)); RPARAM(i)%H06_I_mem_F = (lakePar(il,53) /= 0._dp)
      RPARAM(i)%H06_D_mem_F = (lakePar(il,54) /= 0._dp)
      RPARAM(i)%H06_I_mem_L = nint(lakePar(il,55)); RPARAM(i)%H06_D_mem_L = nint(lakePar(il,56))
    end do
    do i = 1, N
      if (downIndex(i) > 0) then
        if (NETOPO(downIndex(i))%ISLAKE) NETOPO(i)%LAKINLT = .true.
      end if
    end do
  end if

  ! ---- cold-start state (init_model_data.f90:399-505)
  isColdStart = .true.
  do i = 1, N
    RCHFLX(i)%BASIN_QI = 0._dp; RCHFLX(i)%BASIN_QR(0) = 0._dp; RCHFLX(i)%BASIN_QR(1) = 0._dp
    RCHFLX(i)%BASIN_solute = 0._dp; RCHFLX(i)%BASIN_solute_inst = 0._dp
    RCHFLX(i)%Qelapsed = 0; RCHFLX(i)%Qobs = 0._dp
    RCHFLX(i)%REACH_WM_FLUX = 0._dp; RCHFLX(i)%REACH_WM_VOL = 0._dp
    RCHFLX(i)%basinEvapo = 0._dp; RCHFLX(i)%basinPrecip = 0._dp
    allocate(RCHFLX(i)%ROUTE(nRoutes))
    do ix = 1, nRoutes
      RCHFLX(i)%ROUTE(ix)%REACH_VOL(0:1) = 0._dp; RCHFLX(i)%ROUTE(ix)%REACH_Q = 0._dp
      RCHFLX(i)%ROUTE(ix)%Qerror = 0._dp;         RCHFLX(i)%ROUTE(ix)%FLOOD_VOL(0:1) = 0._dp
      RCHFLX(i)%ROUTE(ix)%REACH_ELE = 0._dp;      RCHFLX(i)%ROUTE(ix)%REACH_INFLOW = 0._dp
      RCHFLX(i)%ROUTE(ix)%WB = 0._dp;             RCHFLX(i)%ROUTE(ix)%REACH_WM_FLUX_actual = 0._dp
      RCHFLX(i)%ROUTE(ix)%reach_solute_mass(0:1) = 0._dp; RCHFLX(i)%ROUTE(ix)%reach_solute_flux = 0._dp      ! init_model_data.f90:467-500
    end do
    if (onRoute(impulseResponseFunc)) then
      allocate(RCHFLX(i)%QFUTURE_IRF(size(NETOPO(i)%UH))); RCHFLX(i)%QFUTURE_IRF = 0._dp
    end if
    if (onRoute(kinematicWaveTracking) .and. is_lake_sim .and. NETOPO(i)%ISLAKE) then   ! init_model_data.f90:431-439
      allocate(RCHSTA(i)%LKW_ROUTE%KWAVE(0:0))
      RCHSTA(i)%LKW_ROUTE%KWAVE(0)%QF=-9999; RCHSTA(i)%LKW_ROUTE%KWAVE(0)%TI=-9999; RCHSTA(i)%LKW_ROUTE%KWAVE(0)%TR=-9999
      RCHSTA(i)%LKW_ROUTE%KWAVE(0)%RF=.False.; RCHSTA(i)%LKW_ROUTE%KWAVE(0)%QM=-9999
    end if
    if (onRoute(kinematicWave)) then
      allocate(RCHSTA(i)%KW_ROUTE%molecule%Q(nMolecule%KW_ROUTE)); RCHSTA(i)%KW_ROUTE%molecule%Q = 0._dp
    end if
    if (onRoute(muskingumCunge)) then
      allocate(RCHSTA(i)%MC_ROUTE%molecule%Q(nMolecule%MC_ROUTE)); RCHSTA(i)%MC_ROUTE%molecule%Q = 0._dp
    end if
    if (onRoute(diffusiveWave)) then
      allocate(RCHSTA(i)%DW_ROUTE%molecule%Q(nMolecule%DW_ROUTE)); RCHSTA(i)%DW_ROUTE%molecule%Q = 0._dp
    end if
  end do

  ! ---- processing schedule: orders x branches (dataTypes.f90:58-65), supplied by the case file
  allocate(river_basin(nOrder))
  do io = 1, nOrder
    nb = orderOffset(io+1) - orderOffset(io)
    allocate(river_basin(io)%branch(nb))
    do ib = 1, nb
      k  = orderOffset(io) + ib
      nr = branchOffset(k+1) - branchOffset(k)
      river_basin(io)%branch(ib)%nRch = nr
      allocate(river_basin(io)%branch(ib)%segIndex(nr))
      river_basin(io)%branch(ib)%segIndex = seg(branchOffset(k)+1:branchOffset(k+1))
    end do
  end do

  allocate(ixRch(N)); ixRch = [(i, i=1,N)]
  if (trOn == 1) then
    allocate(basinRunoff(H), basinSolute(H), trflux(N, nRoutes), trmass(N, nRoutes), trbas(N))
    open(newunit=utr, file=trim(fout)//'.tr', access='stream', form='unformatted', status='replace', action='write')
  else
    allocate(basinRunoff(H), basinSolute(0))
  end if
  if (isVolWm == 1) then
    allocate(reachvol(N))
  else
    allocate(reachvol(0))
  end if
  if (is_lake_sim) then
    allocate(basinEvapo(H), basinPrecip(H))
  else
    allocate(basinEvapo(0), basinPrecip(0))
  end if
  if (isFluxWm == 1) then
    allocate(reachflux(N))
  else
    allocate(reachflux(0))
  end if
  allocate(qout(N, nRoutes), volout(N, nRoutes), qr1(N))

  open(newunit=uout, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(uout) MAGIC_OUT, N, nSteps, nRoutes, size(FRAC_FUTURE), dumpEvery

  ! ---- time loop (standalone/route_runoff.f90:80-108 without I/O)
  first_err = 0; first_err_step = 0
  wall = 0._dp
  nSkip = 0     ! leading steps that run but are not timed (spin-up), from the environment
  call get_environment_variable('MZR_REF_SKIP', envbuf, status=envstat)
  if (envstat == 0) read(envbuf, *, iostat=envstat) nSkip
  if (envstat /= 0 .or. nSkip < 0 .or. nSkip >= nSteps) nSkip = 0
  do it = 1, nSteps
    iTime = it
    TSEC(1) = t_start + real(it-1, dp)*dt       ! init_model_data.f90:311-312,600
    TSEC(2) = TSEC(1) + dt
    basinRunoff = runoff(:, it)
    gage_obs%step = it
    if (trOn == 1) basinSolute = solute(:, it)
    if (isFluxWm == 1) reachflux = wmflux(:, it)
    if (isVolWm == 1) reachvol = wmvol(:, it)
    if (is_lake_sim) then
      basinEvapo = evap(:, it); basinPrecip = precip(:, it)
      simDatetime(1) = datetime(ymd(1,it), ymd(2,it), ymd(3,it), 0, 0, 0._dp, calendar=trim(calendar))
    end if
    call system_clock(c0, crate)
    call main_route(basinRunoff, basinEvapo, basinPrecip, basinSolute, reachflux, reachvol, ixRch, &
                    river_basin, NETOPO, RPARAM, RCHFLX, RCHSTA, gage_obs, ierr, message)
    call system_clock(c1)
    if (it > nSkip) wall = wall + real(c1-c0, dp)/real(crate, dp)
    if (ierr == 0 .and. harness_last_err /= 0) then
      ierr = harness_last_err; message = harness_last_msg
    end if
    if (ierr /= 0 .and. first_err == 0) then
      first_err = ierr; first_err_step = it
      write(*,'(a,i0,a,i0,2a)') 'ref_route: ierr=', ierr, ' at step ', it, ' : ', trim(message)
      exit
    end if
    if (dumpEvery > 0) then
      if (mod(it, dumpEvery) == 0 .or. it == nSteps) then
        do ix = 1, nRoutes
          do i = 1, N
            qout(i, ix)   = RCHFLX(i)%ROUTE(ix)%REACH_Q
            volout(i, ix) = RCHFLX(i)%ROUTE(ix)%REACH_VOL(1)
          end do
        end do
        do i = 1, N
          qr1(i) = RCHFLX(i)%BASIN_QR(1)
        end do
        write(uout) it, qout, volout, qr1
        if (trOn == 1) then      ! constituent fluxes of the step go to a file of their own
          do ix = 1, nRoutes
            do i = 1, N
              trflux(i, ix) = RCHFLX(i)%ROUTE(ix)%reach_solute_flux
              trmass(i, ix) = RCHFLX(i)%ROUTE(ix)%reach_solute_mass(1)
            end do
          end do
          do i = 1, N
            trbas(i) = RCHFLX(i)%BASIN_solute
          end do
          write(utr) it, trflux, trmass, trbas
        end if
      end if
    end if
  end do
  write(uout) -1, first_err, first_err_step, wall
  if (trOn == 1) close(utr)

  ! ---- setup products and final state
  write(uout) FRAC_FUTURE
  allocate(ibuf(N+1)); ibuf(1) = 0
  do i = 1, N
    ibuf(i+1) = ibuf(i) + size(NETOPO(i)%UH)
  end do
  write(uout) ibuf
  do i = 1, N
    write(uout) NETOPO(i)%UH
  end do
  ! hillslope QFUTURE [ntdh_bas, N]
  do i = 1, N
    if (allocated(RCHFLX(i)%QFUTURE)) then
      write(uout) RCHFLX(i)%QFUTURE
    else
      write(uout) (0._dp, k=1,size(FRAC_FUTURE))
    end if
  end do
  do ix = 1, nRoutes
    write(uout) routeMethods(ix)
    do i = 1, N
      write(uout) RCHFLX(i)%ROUTE(ix)%REACH_Q, RCHFLX(i)%ROUTE(ix)%REACH_VOL(0), RCHFLX(i)%ROUTE(ix)%REACH_VOL(1), &
                  RCHFLX(i)%ROUTE(ix)%REACH_INFLOW, RCHFLX(i)%ROUTE(ix)%REACH_ELE, RCHFLX(i)%ROUTE(ix)%FLOOD_VOL(1), &
                  RCHFLX(i)%ROUTE(ix)%WB
    end do
    select case (routeMethods(ix))
      case (impulseResponseFunc)
        do i = 1, N
          write(uout) RCHFLX(i)%QFUTURE_IRF
        end do
      case (kinematicWaveTracking)
        allocate(wbuf(WCAP, 4))
        do i = 1, N
          wbuf = -9999._dp
          nw = 0
          if (allocated(RCHSTA(i)%LKW_ROUTE%KWAVE)) then
            nw = size(RCHSTA(i)%LKW_ROUTE%KWAVE)
            do k = 1, min(nw, WCAP)
              wbuf(k,1) = RCHSTA(i)%LKW_ROUTE%KWAVE(k-1)%QF
              wbuf(k,2) = RCHSTA(i)%LKW_ROUTE%KWAVE(k-1)%TI
              wbuf(k,3) = RCHSTA(i)%LKW_ROUTE%KWAVE(k-1)%TR
              wbuf(k,4) = merge(1._dp, 0._dp, RCHSTA(i)%LKW_ROUTE%KWAVE(k-1)%RF)
            end do
          end if
          write(uout) nw, wbuf
        end do
        deallocate(wbuf)
      case (kinematicWave)
        do i = 1, N
          write(uout) RCHSTA(i)%KW_ROUTE%molecule%Q
        end do
      case (muskingumCunge)
        do i = 1, N
          write(uout) RCHSTA(i)%MC_ROUTE%molecule%Q
        end do
      case (diffusiveWave)
        do i = 1, N
          write(uout) RCHSTA(i)%DW_ROUTE%molecule%Q
        end do
    end select
  end do
  close(uout)
  write(*,'(a,i0,a,i0,a,i0,a,f10.4,a,es12.4)') 'ref_route: N=', N, ' steps=', nSteps, ' threads=', nThreads, &
        ' wall_s=', wall, ' reach_steps_per_s=', real(N,dp)*real(nSteps-nSkip,dp)*real(nRoutes,dp)/max(wall,1.e-9_dp)
END PROGRAM ref_driver
