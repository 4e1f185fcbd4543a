!> A Fortran host driving the semistochastic walk on the GPU through sqmc_gpu_mod, the way the
!> reference's walk loop would (INTEGRATION.md, patch sketch 1; do_walk.f90:1350-1366 for the
!> set-up calls, :2171-2934 for the loop that sqmc_gpu_run replaces).  The tables the reference
!> builds in system_setup_chem / read_integrals / generate_psi_t_connected_e_loc come from a deck
!> file written by sqmc_amd.host.dump_walk_deck.
!>   usage: example_walk <deck> <nsteps> [block]
program example_walk
  use iso_c_binding
  use sqmc_gpu_mod
  implicit none
  character(len=512) :: deck, arg
  integer :: u, nblocks, ib
  integer(c_int64_t) :: hdr(22), nsteps, block, nthis, done
  real(c_double) :: scal(4), totals(16), e_num, e_den, e_num_all, e_den_all
  integer(c_int32_t), allocatable, target :: prod(:), osym(:), c2(:)
  real(c_double), allocatable, target :: ints(:)
  integer(c_int64_t), allocatable :: cnt(:), idx(:), ct_up(:), ct_dn(:), up(:), dn(:)
  real(c_double), allocatable :: val(:), ct_num(:), ct_den(:), wt(:), me(:), en(:), ed(:)
  real(c_double), allocatable, target :: stats(:,:)
  integer(c_int8_t), allocatable :: impd(:), init(:), psgn(:)
  type(sqmc_chem_cfg) :: cfg
  type(sqmc_popctl) :: pc
  type(c_ptr) :: gpu
  integer(c_int64_t) :: nwalk

  if (command_argument_count() < 2) stop 'usage: example_walk <deck> <nsteps> [block]'
  call get_command_argument(1, deck)
  call get_command_argument(2, arg); read(arg, *) nsteps
  block = 20
  if (command_argument_count() >= 3) then
    call get_command_argument(3, arg); read(arg, *) block
  endif

  open(newunit=u, file=trim(deck), access='stream', form='unformatted', status='old')
  read(u) hdr
  if (hdr(1) /= int(z'73716D63', c_int64_t)) stop 'not a walk deck'
  read(u) scal
  allocate(prod(hdr(20)), osym(hdr(21)), c2(hdr(22)), ints(hdr(9) + 1))
  allocate(cnt(hdr(16)), idx(hdr(17)), val(hdr(17)), ct_up(hdr(18)), ct_dn(hdr(18)), ct_num(hdr(18)), ct_den(hdr(18)))
  allocate(up(hdr(19)), dn(hdr(19)), wt(hdr(19)), impd(hdr(19)), init(hdr(19)), psgn(hdr(19)), me(hdr(19)), en(hdr(19)), ed(hdr(19)))
  read(u) prod, osym, c2, ints, cnt, idx, val, ct_up, ct_dn, ct_num, ct_den, up, dn, wt, impd, init, psgn, me, en, ed
  close(u)

  cfg%norb = int(hdr(2), c_int32_t); cfg%nup = int(hdr(3), c_int32_t); cfg%ndn = int(hdr(4), c_int32_t)
  cfg%n_core_orb = int(hdr(5), c_int32_t); cfg%time_sym = int(hdr(6), c_int32_t); cfg%z = int(hdr(7), c_int32_t)
  cfg%n_group = int(hdr(8), c_int32_t)
  cfg%product_table = c_loc(prod); cfg%orbital_symmetries = c_loc(osym); cfg%combine_2 = c_loc(c2)
  cfg%n_integrals = hdr(9); cfg%integrals = c_loc(ints)
  cfg%rng_mode = int(hdr(10), c_int32_t); cfg%irand_seed = int(hdr(11:14), c_int32_t); cfg%mwalk = hdr(15)

  call sqmc_gpu_check(sqmc_gpu_set_device(0_c_int), 'set_device')
  call sqmc_gpu_check(sqmc_gpu_init_chem(cfg, gpu), 'init_chem')
  call sqmc_gpu_check(sqmc_gpu_set_projector(gpu, hdr(16), hdr(17), cnt, idx, val), 'set_projector')
  call sqmc_gpu_check(sqmc_gpu_set_ct_table(gpu, hdr(18), ct_up, ct_dn, ct_num, ct_den), 'set_ct_table')
  call sqmc_gpu_check(sqmc_gpu_upload_walkers(gpu, hdr(19), up, dn, wt, impd, init, psgn, me, en, ed), 'upload_walkers')

  ! population control as the reference starts it (do_walk.f90:1170-1260): ramp of tau and r_initiator
  ! until w_abs_gen first reaches the target, e_trial following e_est during equilibration
  pc%tau_sav = scal(1); pc%tau = scal(1); pc%tau_prev = scal(1)
  pc%e_trial = scal(2); pc%e_est = scal(2)
  pc%w_abs_gen_target = scal(3); pc%w_abs_gen = scal(4)
  pc%r_initiator_sav = 1._c_double; pc%r_initiator = 1._c_double; pc%initiator_rescale_power = 1._c_double
  pc%population_control_exponent = 10._c_double
  pc%reweight_factor_inv = 1._c_double; pc%reweight_factor_inv_max = 1._c_double + scal(1)
  pc%e_num_cum = 0; pc%e_den_cum = 0; pc%min_wt = 0.5_c_double; pc%always_spawn_cutoff_wt = 0.5_c_double
  pc%reached_w_abs_gen = 0; pc%initiator_power = 0; pc%initiator_min_distance = 0; pc%c_t_initiator = 0
  pc%semistochastic = 1; pc%reserved = 0; pc%istep = 0; pc%n_equil = 1000000000_c_int64_t

  allocate(stats(16, block))
  ! one sqmc_gpu_run per block: with chained runs the last step of a block enqueues the head of the next block's first step
  call sqmc_gpu_check(sqmc_gpu_set_chained_runs(gpu, 1_c_int32_t), 'set_chained_runs')
  nblocks = int((nsteps + block - 1) / block)
  done = 0; e_num_all = 0; e_den_all = 0
  write(6, '(a)') '   block     steps        w_abs_gen     nwalk          e_trial      e_block (num/den)'
  do ib = 1, nblocks
    nthis = min(block, nsteps - done)
    call sqmc_gpu_check(sqmc_gpu_run(gpu, pc, nthis, c_loc(stats), totals), 'run')
    done = done + nthis
    e_num = sum(stats(4, 1:nthis) * sign(1._c_double, stats(3, 1:nthis))); e_den = sum(abs(stats(3, 1:nthis)))
    e_num_all = e_num_all + e_num; e_den_all = e_den_all + e_den
    write(6, '(i8,i10,f17.4,i10,2f17.9)') ib, done, pc%w_abs_gen, nint(stats(6, nthis)), pc%e_trial, e_num / e_den
  enddo
  call sqmc_gpu_check(sqmc_gpu_num_walkers(gpu, nwalk), 'num_walkers')
  write(6, '(a,i8,i10,4es26.17)') 'fortran walk:', done, nwalk, pc%w_abs_gen, pc%e_trial, pc%e_num_cum, pc%e_den_cum
  call sqmc_gpu_check(sqmc_gpu_finalize(gpu), 'finalize')
end program
