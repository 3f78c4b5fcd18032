! Data-only shim for the reference module `globalData`
! (/root/reference/route/build/src/globalData.f90).
!
! The reference's own globalData.f90 cannot be compiled in this image: it USEs the `pio`
! module of ParallelIO (an un-vendored submodule) for history-file descriptors that the
! routing hot path never touches.  This shim declares ONLY the module variables that the
! unmodified reference solver sources import (globalData.f90:88-279), with the same names,
! types and default values.  It contains no routing arithmetic.
MODULE globalData
  USE nrtype
  USE public_var
  USE datetime_data, ONLY: datetime
  USE dataTypes,     ONLY: RCHTOPO, STRFLX, cMolecule, subbasin_mpi, var_info
  USE var_lookup,    ONLY: nVarsSEG
  USE base_route,    ONLY: routeContainer
  implicit none
  save
  public
  integer(i4b)                      :: nRch_mainstem = 0
  integer(i4b)                      :: nRch_trib = 0
  type(routeContainer), allocatable :: rch_routes(:)
  integer(i4b)                      :: nRoutes
  integer(i4b), allocatable         :: routeMethods(:)
  logical(lgt)                      :: onRoute(0:nRouteMethods-1)
  integer(i4b)                      :: idxSUM, idxIRF, idxKWT, idxKW, idxMC, idxDW
  integer(i4b)                      :: iTime
  real(dp)                          :: TSEC(1:2)
  type(datetime)                    :: simDatetime(0:2)
  integer(i4b)                      :: maxtdh=0
  type(cMolecule)                   :: nMolecule
  logical(lgt)                      :: isColdStart=.true.
  integer(i4b)                      :: nThreads = 1
  logical(lgt)                      :: masterproc = .true.
  real(dp)                          :: time_conv
  real(dp)                          :: length_conv
  real(dp)                          :: time_conv_solute
  real(dp)                          :: mass_conv_solute
  real(dp)                          :: high_depth=100000._dp
  type(RCHTOPO), allocatable        :: NETOPO_trib(:)
  type(RCHTOPO), allocatable        :: NETOPO_main(:)
  real(dp),      allocatable        :: FRAC_FUTURE(:)
  type(STRFLX),  allocatable        :: RCHFLX_trib(:)
  type(subbasin_mpi), allocatable   :: domains_mpi(:)
  integer(i4b)                      :: nDomain_mpi
  integer(i4b), allocatable         :: nTribOutlet
  ! what augment_ntopo (process_ntopo.f90:61-67) imports: spatially constant routing parameters (globalData.f90:181-188,
  ! same defaults) and the metadata flags "read from the file or computed" of the reach properties (globalData.f90:204)
  real(dp)                          :: fshape
  real(dp)                          :: tscale
  real(dp)                          :: velo
  real(dp)                          :: diff
  real(dp)                          :: mann_n
  real(dp)                          :: wscale
  real(dp)                          :: dscale=0.000045
  real(dp)                          :: floodplainSlope=1000
  type(var_info)                    :: meta_SEG(nVarsSEG)
END MODULE globalData
