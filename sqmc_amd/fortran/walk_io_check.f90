!> Reads the two text files a walk can start from with exactly the statements of the reference
!> (psit_connections: do_walk.f90:702-741 read(56,*) ...; deterministic matrix elements:
!> do_walk.f90:898-940 read(57,*) ...) and writes them back with the statements that produce them
!> (semistoch.f90:86-126; do_walk.f90:970-1010).  Pins sqmc_amd.host.write/read_psit_connections
!> and write/read_dtm_elems to what a Fortran list-directed read accepts and a formatted write emits.
!>   usage: walk_io_check psit <in> <out> nup ndn norb   |   walk_io_check dtm <in> <out> nup ndn
program walk_io_check
  implicit none
  integer, parameter :: rk = kind(1.d0), i8b = selected_int_kind(18)
  character(len=512) :: what, fin, fout, arg
  character(len=16) :: fmt
  integer :: nup, ndn, norb, ndet_psi_t, ndet_con, i, j, ind, n_imp
  integer(i8b) :: nnz, k
  integer, allocatable :: tmp_reader(:), orbs(:,:), counts(:), dets(:,:)
  integer(i8b), allocatable :: indices(:)
  real(rk), allocatable :: num(:), den(:), vals(:)
  real(rk) :: dtm_energy, e_t
  call get_command_argument(1, what); call get_command_argument(2, fin); call get_command_argument(3, fout)
  call get_command_argument(4, arg); read(arg, *) nup
  call get_command_argument(5, arg); read(arg, *) ndn
  allocate(tmp_reader(nup + ndn))
  if (trim(what) == 'psit') then
    call get_command_argument(6, arg); read(arg, *) norb
    open(56, file=trim(fin), status='old')
    read(56, *) ndet_psi_t, ndet_con
    read(56, *)
    allocate(orbs(nup + ndn, ndet_con), num(ndet_con), den(ndet_con))
    do i = 1, ndet_con
      read(56, *) tmp_reader, num(i), den(i)
      orbs(:, i) = tmp_reader
    enddo
    close(56)
    e_t = num(1) / den(1)
    write(6, '(a,2i10,3es25.16)') 'walk_io_check psit:', ndet_psi_t, ndet_con, sum(num), sum(den), real(sum(orbs), rk)
    open(8, file=trim(fout), status='replace')
    write(8,'(i8,i12,2i4,i6,i3,f15.8,'' ndet_psi_t, ndet_connections_nonzero, nup-n_core_orb, ndn-n_core_orb, norb, 0, E_T'')') &
   &  ndet_psi_t, ndet_con, nup, ndn, norb, 0, e_t
    write(8,'(''orb_up          orb_dn           e_loc_num            e_loc_den'')')
    write (fmt, '(i5)') nup+ndn-1
    do i = 1, ndet_con
      write(8,'(i3,' // trim(fmt) // 'i4,f22.15,f19.15)') orbs(1:nup, i), orbs(nup+1:nup+ndn, i), num(i), den(i)
    enddo
    close(8)
  else
    open(57, file=trim(fin), status='old')
    read(57, *) n_imp, nnz, dtm_energy
    allocate(counts(n_imp), dets(nup + ndn, n_imp), indices(nnz), vals(nnz))
    read(57, *) counts(1:n_imp)
    do i = 1, n_imp
      read(57, *) ind, tmp_reader
      dets(:, ind) = tmp_reader
    enddo
    do k = 1, nnz
      read(57, *) indices(k), vals(k)
    enddo
    close(57)
    write(6, '(a,i10,i14,4es25.16)') 'walk_io_check dtm:', n_imp, nnz, dtm_energy, sum(vals), real(sum(indices), rk), real(sum(counts), rk)
    open(8, file=trim(fout), status='replace')
    write(8,*) n_imp, nnz, dtm_energy, "number of deterministic dets, number of nonzero deterministic Hamiltonian elements, ground state energy within deterministic space"
    write (fmt, '(i15)') n_imp
    write(8, '(' // trim(fmt) // 'i8)') counts
    do i = 1, n_imp
      write(8,*) i, dets(1:nup, i), dets(nup+1:nup+ndn, i)
    enddo
    do k = 1, nnz
      write (8,*) indices(k), vals(k)
    enddo
    close(8)
  endif
end program
