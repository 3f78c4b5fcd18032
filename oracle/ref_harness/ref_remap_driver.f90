! Harness around the UNMODIFIED reference remap routines (process_remap.f90: remap_runoff ->
! remap_1D_runoff / remap_2D_runoff, sort_flux).  TEST INFRASTRUCTURE ONLY (oracle/README.md).
! usage: ref_remap <case.bin> <out.bin>
! case (stream, native endian):
!   int32 magic(1380798800) kind(1 = 1-D, 2 = 2-D, 3 = sort_flux) nMap nOverlap n1 n2 H nSteps removeNeg
!   kind 1/2: int32 hru_ix(nMap) num_qhru(nMap); kind 1: int32 qhru_ix(nOverlap), int64 qhru_id(nOverlap), int64 src_id(n1)
!             kind 2: int32 i_index(nOverlap) j_index(nOverlap);  real64 weight(nOverlap)
!   kind 3:   int32 ix_in(n1)
!   real64 sim(n1[,n2], nSteps)
! out: int32 ierr, real64 basinRunoff(H, nSteps)
PROGRAM ref_remap_driver
  USE nrtype
  USE dataTypes, ONLY: remap, runoff
  USE public_var, ONLY: integerMissing
  USE process_remap_module, ONLY: remap_runoff, sort_flux
  implicit none
  character(len=1024) :: fcase, fout
  character(len=strLen) :: message
  integer(i4b) :: magic, mode, nMap, nOverlap, n1, n2, H, nSteps, removeNeg, ierr, uin, uout, it, first_err
  type(remap)  :: rmp
  type(runoff) :: ro
  integer(i4b), allocatable :: ix_in(:)
  real(dp), allocatable :: sim1(:,:), sim2(:,:,:), outp(:,:)

  call get_command_argument(1, fcase); call get_command_argument(2, fout)
  open(newunit=uin, file=trim(fcase), access='stream', form='unformatted', status='old', action='read')
  read(uin) magic, mode, nMap, nOverlap, n1, n2, H, nSteps, removeNeg
  if (magic /= 1380798800) stop 'ref_remap: bad magic'
  allocate(ro%basinRunoff(H), outp(H, nSteps))
  ro%basinRunoff = 0._dp
  if (mode == 1 .or. mode == 2) then
    allocate(rmp%hru_ix(nMap), rmp%num_qhru(nMap), rmp%weight(nOverlap), rmp%hru_id(nMap))
    read(uin) rmp%hru_ix, rmp%num_qhru
    rmp%hru_id = 0
    if (mode == 1) then
      allocate(rmp%qhru_ix(nOverlap), rmp%qhru_id(nOverlap), ro%hru_id(n1))
      read(uin) rmp%qhru_ix, rmp%qhru_id, ro%hru_id
    else
      allocate(rmp%i_index(nOverlap), rmp%j_index(nOverlap))
      read(uin) rmp%i_index, rmp%j_index
    end if
    read(uin) rmp%weight
  else
    allocate(ix_in(n1), ro%hru_id(n1))
    read(uin) ix_in
    ro%hru_id = 0
  end if
  if (mode == 2) then
    allocate(sim2(n1, n2, nSteps), ro%sim2d(n1, n2)); read(uin) sim2
    ro%nSpace = (/n1, n2/)
  else
    allocate(sim1(n1, nSteps), ro%sim(n1)); read(uin) sim1
    ro%nSpace(1) = n1; ro%nSpace(2) = integerMissing
  end if
  close(uin)

  first_err = 0
  do it = 1, nSteps
    if (mode == 2) then
      ro%sim2d = sim2(:, :, it)
    else
      ro%sim = sim1(:, it)
    end if
    if (mode == 3) then
      call sort_flux(ro%hru_id, ix_in, ro%sim, removeNeg /= 0, ro%basinRunoff, ierr, message)
    else
      call remap_runoff(ro, rmp, ro%basinRunoff, ierr, message)
    end if
    if (ierr /= 0 .and. first_err == 0) then
      first_err = ierr
      write(*,'(a,i0,2a)') 'ref_remap: ierr=', ierr, ' : ', trim(message)
    end if
    outp(:, it) = ro%basinRunoff
  end do
  open(newunit=uout, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(uout) first_err, outp
  close(uout)
END PROGRAM ref_remap_driver
