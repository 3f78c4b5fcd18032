! Harness around the reference's UNMODIFIED start-up routines for the river network:
!   augment_ntopo              (process_ntopo.f90:39-266) with the network_topo.f90 routines it calls
!                              (hru2segment, up2downSegment, reachOrder, streamOrdering, reach_list, reach_mask)
!   mpi_domain_decomposition   (domain_decomposition.f90:41-163 -> classify_river_basin :450-590, assign_node :724-819)
! It fills the reference's data structures from a plain-text case (what read_streamSeg would have read from the
! topology file: ids, lengths, slopes, HRU areas), calls the routines, and dumps what they made.  Test infrastructure:
! mizuroute_amd/standalone.py (augmentation) and mizuroute_amd/partition.py (decomposition) are compared with it.
!
!   ref_topo <case.txt> <out.txt>
program ref_topo
  USE nrtype
  USE public_var
  USE dataTypes,  ONLY: var_ilength, var_dlength
  USE var_lookup, ONLY: ixHRU, ixSEG, ixHRU2SEG, ixNTOPO, nVarsHRU, nVarsSEG, nVarsHRU2SEG, nVarsNTOPO
  USE globalData, ONLY: onRoute, fshape, tscale, velo, diff, mann_n, wscale, dscale, floodplainSlope, meta_SEG, &
                        domains_mpi, nDomain_mpi
  USE process_ntopo,        ONLY: augment_ntopo
  USE domain_decomposition, ONLY: mpi_domain_decomposition
  implicit none
  character(len=1024)            :: fin, fout
  character(len=strLen)          :: message
  integer(i4b)                   :: nSeg, nHRU, nNodes, ierr, i, k, iFlood, iIRF, nContribHRU
  integer(i4b)                   :: tot_hru, tot_upseg, tot_upstream, tot_uh
  integer(i4b), allocatable      :: segId(:), downSegId(:), hruId(:), hruSegId(:)
  real(dp),     allocatable      :: length(:), slope(:), area(:)
  type(var_dlength), allocatable :: structHRU(:), structSEG(:)
  type(var_ilength), allocatable :: structHRU2SEG(:), structNTOPO(:)
  integer(i4b), allocatable      :: ixHRU_desired(:), ixSeg_desired(:)

  call get_command_argument(1, fin); call get_command_argument(2, fout)
  open(11, file=trim(fin), status='old', action='read')
  read(11,*) nSeg, nHRU, nNodes
  read(11,*) dt, fshape, tscale, velo, diff, mann_n, wscale, dscale, iFlood, iIRF
  allocate(segId(nSeg), downSegId(nSeg), length(nSeg), slope(nSeg), hruId(nHRU), hruSegId(nHRU), area(nHRU))
  read(11,*) segId; read(11,*) downSegId; read(11,*) length; read(11,*) slope
  read(11,*) hruId; read(11,*) hruSegId;  read(11,*) area
  close(11)
  floodplain = (iFlood /= 0)
  onRoute = .false.; onRoute(impulseResponseFunc) = (iIRF /= 0)
  topoNetworkOption = compute; computeReachList = compute; idSegOut = -9999
  ! hydraulic geometry is not in the file: computed from wscale / dscale / mann_n (augment_ntopo :171-182)
  meta_SEG(:)%varFile = .true.
  meta_SEG(ixSEG%width)%varFile = .false.; meta_SEG(ixSEG%depth)%varFile = .false.; meta_SEG(ixSEG%man_n)%varFile = .false.
  meta_SEG(ixSEG%sideSlope)%varFile = .false.; meta_SEG(ixSEG%floodplainSlope)%varFile = .false.

  ! the data structures as the reader leaves them: every scalar variable one element, the ragged ones unallocated
  allocate(structHRU(nHRU), structHRU2SEG(nHRU), structSEG(nSeg), structNTOPO(nSeg))
  do i = 1, nHRU
    allocate(structHRU(i)%var(nVarsHRU), structHRU2SEG(i)%var(nVarsHRU2SEG))
    allocate(structHRU(i)%var(ixHRU%area)%dat(1)); structHRU(i)%var(ixHRU%area)%dat(1) = area(i)
    do k = 1, nVarsHRU2SEG
      allocate(structHRU2SEG(i)%var(k)%dat(1)); structHRU2SEG(i)%var(k)%dat(1) = integerMissing
    end do
    structHRU2SEG(i)%var(ixHRU2SEG%HRUid)%dat(1) = hruId(i)
    structHRU2SEG(i)%var(ixHRU2SEG%hruSegId)%dat(1) = hruSegId(i)
  end do
  do i = 1, nSeg
    allocate(structSEG(i)%var(nVarsSEG), structNTOPO(i)%var(nVarsNTOPO))
    do k = 1, nVarsSEG
      if (k==ixSEG%hruArea .or. k==ixSEG%weight .or. k==ixSEG%timeDelayHist) cycle
      allocate(structSEG(i)%var(k)%dat(1)); structSEG(i)%var(k)%dat(1) = realMissing
    end do
    structSEG(i)%var(ixSEG%length)%dat(1) = length(i); structSEG(i)%var(ixSEG%slope)%dat(1) = slope(i)
    do k = 1, nVarsNTOPO
      if (k==ixNTOPO%hruContribIx .or. k==ixNTOPO%hruContribId .or. k==ixNTOPO%upSegIds .or. k==ixNTOPO%upSegIndices .or. &
          k==ixNTOPO%allUpSegIndices .or. k==ixNTOPO%goodBasin) cycle
      allocate(structNTOPO(i)%var(k)%dat(1)); structNTOPO(i)%var(k)%dat(1) = integerMissing
    end do
    structNTOPO(i)%var(ixNTOPO%segId)%dat(1) = segId(i); structNTOPO(i)%var(ixNTOPO%downSegId)%dat(1) = downSegId(i)
    structNTOPO(i)%var(ixNTOPO%segIndex)%dat(1) = i
    structNTOPO(i)%var(ixNTOPO%islake)%dat(1) = 0; structNTOPO(i)%var(ixNTOPO%userTake)%dat(1) = 0
  end do

  call augment_ntopo(nHRU, nSeg, structHRU, structSEG, structHRU2SEG, structNTOPO, ierr, message, &
                     tot_hru=tot_hru, tot_upseg=tot_upseg, tot_upstream=tot_upstream, tot_uh=tot_uh, &
                     ixHRU_desired=ixHRU_desired, ixSeg_desired=ixSeg_desired)
  open(12, file=trim(fout), status='replace', action='write')
  write(12,'(A,I6,1x,A)') 'augment_ntopo ', ierr, trim(message)
  if (ierr /= 0) stop 1
  write(12,'(4(I12,1x))') tot_hru, tot_upseg, tot_upstream, tot_uh
  do i = 1, nSeg      ! scalars of every reach
    write(12,'(7(I10,1x),8(ES24.16E3,1x))') structNTOPO(i)%var(ixNTOPO%segIndex)%dat(1), structNTOPO(i)%var(ixNTOPO%downSegIndex)%dat(1), &
      structNTOPO(i)%var(ixNTOPO%nHRU)%dat(1), size(structNTOPO(i)%var(ixNTOPO%upSegIndices)%dat), &
      size(structNTOPO(i)%var(ixNTOPO%allUpSegIndices)%dat), structNTOPO(i)%var(ixNTOPO%rchOrder)%dat(1), &
      structNTOPO(i)%var(ixNTOPO%streamOrder)%dat(1), &
      structSEG(i)%var(ixSEG%basArea)%dat(1), structSEG(i)%var(ixSEG%upsArea)%dat(1), structSEG(i)%var(ixSEG%totalArea)%dat(1), &
      structSEG(i)%var(ixSEG%width)%dat(1), structSEG(i)%var(ixSEG%depth)%dat(1), structSEG(i)%var(ixSEG%storage)%dat(1), &
      structSEG(i)%var(ixSEG%man_n)%dat(1), structSEG(i)%var(ixSEG%floodplainSlope)%dat(1)
  end do
  do i = 1, nSeg      ! ragged lists: HRUs and their weights, immediate upstream reaches and their goodBasin flags, all upstream reaches
    write(12,'(*(I10,1x))') structNTOPO(i)%var(ixNTOPO%hruContribIx)%dat
    write(12,'(*(ES24.16E3,1x))') structSEG(i)%var(ixSEG%weight)%dat
    write(12,'(*(I10,1x))') structNTOPO(i)%var(ixNTOPO%upSegIndices)%dat
    write(12,'(*(I10,1x))') structNTOPO(i)%var(ixNTOPO%goodBasin)%dat
    write(12,'(*(I10,1x))') structNTOPO(i)%var(ixNTOPO%allUpSegIndices)%dat
    if (iIRF /= 0) then
      write(12,'(*(ES24.16E3,1x))') structSEG(i)%var(ixSEG%timeDelayHist)%dat
    else
      write(12,*)
    end if
  end do

  call mpi_domain_decomposition(nNodes, nSeg, structNTOPO, structHRU2SEG, nContribHRU, ierr, message)
  write(12,'(A,I6,1x,A)') 'mpi_domain_decomposition ', ierr, trim(message)
  if (ierr /= 0) stop 2
  write(12,'(2(I12,1x))') nDomain_mpi, nContribHRU
  do i = 1, nDomain_mpi
    if (allocated(domains_mpi(i)%segIndex)) then
      k = size(domains_mpi(i)%segIndex)
    else
      k = 0
    end if
    write(12,'(4(I10,1x))') domains_mpi(i)%basinType, domains_mpi(i)%idNode, k, size(domains_mpi(i)%hruIndex)
    if (k > 0) then
      write(12,'(*(I10,1x))') domains_mpi(i)%segIndex
    else
      write(12,*)
    end if
    write(12,'(*(I10,1x))') domains_mpi(i)%hruIndex
  end do
  close(12)
end program ref_topo
