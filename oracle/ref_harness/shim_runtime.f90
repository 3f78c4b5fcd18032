! No-op shims for three reference modules whose real versions need libraries this image
! lacks (MPI, GPTL/PIO timers, netCDF).  None of them contains routing arithmetic; the hot
! path reaches them only for timers, fatal-error exit and the (disabled) gauge-obs reader.
!   perf_mod    : GPTL timers  (main_route.f90:288,353,407)        -> no-ops
!   mpi_utils   : MPI wrappers (water_balance.f90:200, comp_global_wb only) -> single-process identity
!   model_utils : handle_err   (model_utils.f90:45-57, calls MPI_ABORT)      -> print + stop
!   obs_data    : gauge observations via netCDF (main_route.f90:9,125-148; only if qmodOption=1)
MODULE perf_mod
  implicit none
CONTAINS
  SUBROUTINE t_startf(name); character(*), intent(in) :: name; END SUBROUTINE
  SUBROUTINE t_stopf(name);  character(*), intent(in) :: name; END SUBROUTINE
END MODULE perf_mod

MODULE mpi_utils
  USE nrtype
  implicit none
  private
  public :: shr_mpi_reduce, shr_mpi_abort
CONTAINS
  SUBROUTINE shr_mpi_reduce(localArray, method, reducedArray, ierr, message)
    real(dp),     intent(in)  :: localArray(:)
    character(*), intent(in)  :: method
    real(dp),     intent(out) :: reducedArray(:)
    integer(i4b), intent(out) :: ierr
    character(*), intent(out) :: message
    ierr=0; message='shr_mpi_reduce(single process)/'
    reducedArray = localArray
  END SUBROUTINE
  SUBROUTINE shr_mpi_abort(message, ierr, comm)
    character(*), intent(in)           :: message
    integer(i4b), intent(in)           :: ierr
    integer(i4b), intent(in), optional :: comm
    write(*,*) 'ABORT: ', trim(message), ierr
    stop 2
  END SUBROUTINE
END MODULE mpi_utils

MODULE model_utils
  USE nrtype
  implicit none
  integer(i4b), save :: harness_last_err = 0
  character(len=strLen), save :: harness_last_msg = ''
CONTAINS
  SUBROUTINE handle_err(err, message)
    ! The reference aborts the MPI job here.  The harness records the first error so the
    ! driver can report the reference's ierr for the step, and keeps going.
    integer(i4b), intent(in) :: err
    character(*), intent(in) :: message
    if (err/=0 .and. harness_last_err==0) then
      harness_last_err = err
      harness_last_msg = message
    end if
  END SUBROUTINE
END MODULE model_utils

MODULE obs_data
  ! stand-in for the reference's gauge-observation reader (obs_data.f90 needs ncio_utils / netCDF): the driver puts the
  ! observations of the case file into the object, main_route (unmodified) asks for them through the reference's interface
  USE nrtype
  USE public_var,    ONLY: integerMissing, realMissing
  USE datetime_data, ONLY: datetime
  implicit none
  type :: gageObs
    integer(i4b) :: step = 0                      ! simulation step main_route is about to route (set by the driver)
    integer(i4b) :: cur = 0                       ! record read_obs made current
    integer(i4b), allocatable :: have(:)          ! (nSteps) 1 = there is an observation time at this step
    integer(i4b), allocatable :: link(:)          ! (nGauge) reach index of every gauge, integerMissing = none
    real(dp),     allocatable :: val(:,:)         ! (nGauge, nSteps)
  CONTAINS
    procedure, pass :: time_ix
    procedure, pass :: read_obs
    procedure, pass :: link_ix
    procedure, pass :: get_obs
  end type gageObs
CONTAINS
  FUNCTION time_ix(this, dt) result(ix)
    class(gageObs), intent(in) :: this
    type(datetime), intent(in) :: dt
    integer(i4b) :: ix
    ix = integerMissing
    if (allocated(this%have)) then
      if (this%step >= 1 .and. this%step <= size(this%have)) then
        if (this%have(this%step) == 1) ix = this%step
      end if
    end if
  END FUNCTION
  SUBROUTINE read_obs(this, ierr, message, index_time)
    class(gageObs), intent(inout) :: this
    integer(i4b), intent(out) :: ierr
    character(*), intent(out) :: message
    integer(i4b), intent(in), optional :: index_time
    ierr=0; message=''
    if (present(index_time)) this%cur = index_time
  END SUBROUTINE
  FUNCTION link_ix(this) result(ix)
    class(gageObs), intent(in) :: this
    integer(i4b), allocatable :: ix(:)
    if (allocated(this%link)) then
      allocate(ix(size(this%link))); ix = this%link
    else
      allocate(ix(0))
    end if
  END FUNCTION
  FUNCTION get_obs(this, tix, six) result(q)
    class(gageObs), intent(in) :: this
    integer(i4b), intent(in), optional :: tix, six
    real(dp) :: q
    q = realMissing
    if (allocated(this%val) .and. present(six)) q = this%val(six, this%cur)
  END FUNCTION
END MODULE obs_data
