!> Reads a variational-wavefunction file with exactly the statements of perform_hci
!> (hci.f90:203-214: open(form='unformatted'); read ndets; read dets_up; read dets_dn; read wts;
!> read energy, determinants as integer(16)) and writes it back with the statements of
!> hci.f90:606-612.  Pins sqmc_amd.host.read_wf_var / write_wf_var to the compiler's record layout.
!>   usage: wf_io_check <in> <out> <n_states>
program wf_io_check
  implicit none
  integer, parameter :: ik = selected_int_kind(38), rk = kind(1.d0)
  character(len=512) :: fin, fout, arg
  integer :: ndets, n_states
  integer(ik), allocatable :: up(:), dn(:)
  real(rk), allocatable :: wts(:,:), energy(:)
  call get_command_argument(1, fin); call get_command_argument(2, fout)
  call get_command_argument(3, arg); read(arg, *) n_states
  open(1, file=trim(fin), form='unformatted', status='old')
  read(1) ndets
  allocate(up(ndets), dn(ndets), wts(ndets, n_states), energy(n_states))
  read(1) up(1:ndets)
  read(1) dn(1:ndets)
  read(1) wts(1:ndets, 1:n_states)
  read(1) energy(1:n_states)
  close(1)
  write(6, '(a,i10,2i24,3es25.16)') 'wf_io_check:', ndets, int(sum(mod(up, 1000003_ik)), 8), int(sum(mod(dn, 1000003_ik)), 8), &
    sum(wts(:, 1)), sum(wts(:, n_states)**2), energy(n_states)
  open(2, file=trim(fout), form='unformatted', status='replace')
  write(2) ndets
  write(2) up(1:ndets)
  write(2) dn(1:ndets)
  write(2) wts(1:ndets, 1:n_states)
  write(2) energy(1:n_states)
  close(2)
end program
