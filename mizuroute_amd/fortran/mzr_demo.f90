! Minimal Fortran host for the C-ABI: routes a small network read from a case file step by step,
! the way mizuRoute's time loop (standalone/route_runoff.f90:80-108) calls mpi_route/main_route.
! usage: mzr_demo <case.bin> <out.bin>     (case format: the flat binary file the test harness writes, tests/ + INTEGRATION.md; uhSource=1)
! Writes REACH_Q of every step: int32 N, nSteps, nRoutes; float64 Q(N, nRoutes, nSteps).
PROGRAM mzr_demo
  USE, INTRINSIC :: iso_c_binding
  USE mzr_c
  implicit none
  integer, parameter :: dp = c_double
  character(len=1024) :: fcase, fout
  integer(c_int) :: magic, version, N, nHru, nSteps, nRoutes, methodsIn(6), doesBasinRoute, hw_drain_point
  integer(c_int) :: nUpTot, nHruTot, nOrder, nBranch, uhSource, ntdhBas, nUhTot, dumpEvery, isFluxWm, isLakeSim, ierr
  real(dp) :: dt, min_length_route, runoffMin, fshape, tscale, velo, diff, t_start
  integer(c_int), allocatable :: downIndex(:), reachId(:), upOffset(:), upIndex(:), upGood(:), hruOffset(:), hruIndex(:)
  integer(c_int), allocatable :: orderOffset(:), branchOffset(:), seg(:), uhOffset(:)
  real(dp), allocatable :: hruWeight(:), par(:,:), frac(:), uh(:), runoff(:,:), q(:)
  character(len=10), parameter :: pname(11) = [character(len=10) :: 'R_SLOPE','R_MAN_N','R_WIDTH','R_DEPTH','RLENGTH', &
      'R_STORAGE','SIDE_SLOPE','FLDP_SLOPE','BASAREA','TOTAREA','MINFLOW']
  type(mzr_config) :: cfg
  type(c_ptr) :: dom
  integer :: uin, uout, it, ix, p
  real(dp) :: T0, T1

  call get_command_argument(1, fcase); call get_command_argument(2, fout)
  open(newunit=uin, file=trim(fcase), access='stream', form='unformatted', status='old', action='read')
  read(uin) magic, version
  read(uin) N, nHru, nSteps, nRoutes, methodsIn, doesBasinRoute, hw_drain_point, nUpTot, nHruTot, nOrder, nBranch, &
            uhSource, ntdhBas, nUhTot, dumpEvery, isFluxWm, isLakeSim
  read(uin) dt, min_length_route, runoffMin, fshape, tscale, velo, diff, t_start
  allocate(downIndex(N), reachId(N), upOffset(N+1), upIndex(nUpTot), upGood(nUpTot), hruOffset(N+1), hruIndex(nHruTot))
  allocate(hruWeight(nHruTot), par(N,11), orderOffset(nOrder+1), branchOffset(nBranch+1), seg(N))
  read(uin) downIndex, reachId, upOffset, upIndex, upGood, hruOffset, hruIndex, hruWeight
  read(uin) par
  read(uin) orderOffset, branchOffset, seg
  if (uhSource /= 1) stop 'mzr_demo needs a case file that carries FRAC_FUTURE and UH'
  allocate(frac(ntdhBas), uhOffset(N+1), uh(nUhTot), runoff(nHru, nSteps), q(N))
  read(uin) frac, uhOffset, uh
  read(uin) runoff
  close(uin)

  call mzr_default_config(cfg)
  cfg%dt = dt; cfg%nRoutes = nRoutes; cfg%routeMethods = methodsIn
  cfg%doesBasinRoute = doesBasinRoute; cfg%hw_drain_point = hw_drain_point
  cfg%min_length_route = min_length_route; cfg%runoffMin = runoffMin; cfg%maxWindow = 1
  ierr = mzr_create(cfg, dom);                                             call check('mzr_create')
  ierr = mzr_set_network(dom, N, nHru, downIndex, upOffset, upIndex, upGood, hruOffset, hruIndex, hruWeight, reachId)
  call check('mzr_set_network')
  do p = 1, 11
    ierr = mzr_set_param(dom, trim(pname(p))//c_null_char, par(:,p));      call check('mzr_set_param')
  end do
  ierr = mzr_set_frac_future(dom, ntdhBas, frac);                          call check('mzr_set_frac_future')
  ierr = mzr_set_uh(dom, uhOffset, uh);                                    call check('mzr_set_uh')
  ierr = mzr_init_state(dom);                                              call check('mzr_init_state')

  open(newunit=uout, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(uout) N, nSteps, nRoutes
  do it = 1, nSteps                      ! the reference's time loop
    T0 = t_start + real(it-1, dp)*dt; T1 = T0 + dt
    ierr = mzr_step(dom, T0, T1, runoff(:, it));                           call check('mzr_step')
    do ix = 1, nRoutes
      ierr = mzr_get_flux(dom, methodsIn(ix), MZR_F_Q, q);                 call check('mzr_get_flux')
      write(uout) q
    end do
  end do
  close(uout)
  ierr = mzr_destroy(dom)
  write(*,'(a,i0,a,i0,a)') 'mzr_demo: routed ', N, ' reaches for ', nSteps, ' steps'
CONTAINS
  subroutine check(where)
    character(*), intent(in) :: where
    if (ierr /= 0) then
      write(*,'(a,a,a,i0,a,a)') 'mzr_demo: ', where, ' ierr=', ierr, ' ', trim(mzr_message(dom))
      stop 1
    end if
  end subroutine
END PROGRAM mzr_demo
