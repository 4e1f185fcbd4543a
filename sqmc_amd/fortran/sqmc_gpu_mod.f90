!> sqmc_gpu_mod -- iso_c_binding view of include/sqmc_gpu.h for a Fortran host such as the
!> reference's do_walk.f90 / more_tools.f90 / hci.f90.  Every interface names the C entry point
!> it binds; INTEGRATION.md lists the reference call sites each one replaces.
!> Determinants cross the boundary as integer(c_int64_t) (the low word of the reference's
!> 128-bit ik, exactly as conv_128_to_64 does for MPI, mpi_routines.f90:671-680).
module sqmc_gpu_mod
  use iso_c_binding
  implicit none
  private
  public :: sqmc_chem_cfg, sqmc_heg_cfg, sqmc_gpu_init_heg, sqmc_hubbard_cfg, sqmc_gpu_init_hubbard, sqmc_step_params, sqmc_popctl, sqmc_gpu_run
  public :: sqmc_gpu_set_device, sqmc_gpu_init_chem, sqmc_gpu_finalize, sqmc_gpu_last_error, sqmc_gpu_set_hb_tables
  public :: sqmc_gpu_set_projector, sqmc_gpu_scale_projector, sqmc_gpu_set_ct_table, sqmc_gpu_upload_walkers
  public :: sqmc_gpu_num_walkers, sqmc_gpu_download_walkers, sqmc_gpu_step, sqmc_gpu_get_rng, sqmc_gpu_set_rng
  public :: sqmc_gpu_spmv_prepare, sqmc_gpu_spmv_apply, sqmc_gpu_spmv_free, sqmc_gpu_spmv_sym_upper
  public :: sqmc_gpu_hamiltonian_batch, sqmc_gpu_hamiltonian_chem_batch, sqmc_gpu_build_sparse_ham, sqmc_gpu_propose_batch
  public :: sqmc_gpu_hci_connections, sqmc_gpu_free, sqmc_gpu_set_timing, sqmc_gpu_get_timing
  public :: sqmc_gpu_det_owner, sqmc_gpu_shard_config, sqmc_gpu_shard_begin, sqmc_gpu_shard_pack, sqmc_gpu_shard_finish
  public :: sqmc_gpu_annihilate, sqmc_gpu_build_spmv_plan, sqmc_gpu_hci_connections_slice
  public :: sqmc_gpu_comm_unique_id, sqmc_gpu_comm_init, sqmc_gpu_comm_size, sqmc_gpu_set_owner_hash, sqmc_gpu_tail_stats, sqmc_gpu_slowest_steps, sqmc_gpu_set_chained_runs, sqmc_gpu_hci_pt2, sqmc_gpu_hci_set_active_space, sqmc_gpu_set_heatbath_tables, sqmc_gpu_propose_heatbath_batch, sqmc_heatbath_tables, sqmc_gpu_shard_step, sqmc_gpu_shard_run, sqmc_gpu_shard_time_split, sqmc_gpu_davidson
  public :: sqmc_gpu_set_hf_to_psit, sqmc_gpu_setup_efficient_heatbath, sqmc_gpu_get_heatbath_tables
  public :: sqmc_gpu_check

  integer(c_int), parameter, public :: SQMC_RNG_REPLAY = 0, SQMC_RNG_COUNTER = 1

  type, bind(C) :: sqmc_chem_cfg
    integer(c_int32_t) :: norb, nup, ndn, n_core_orb
    integer(c_int32_t) :: time_sym, z
    integer(c_int32_t) :: n_group
    type(c_ptr) :: product_table       ! int32 (0:8,0:8) stored [i*9+j]
    type(c_ptr) :: orbital_symmetries  ! int32 (0:norb)
    type(c_ptr) :: combine_2           ! int32 (0:norb+1,0:norb+1) stored [i*(norb+2)+j]
    integer(c_int64_t) :: n_integrals
    type(c_ptr) :: integrals           ! real(c_double) (0:n_integrals)
    integer(c_int32_t) :: rng_mode
    integer(c_int32_t) :: irand_seed(4)
    integer(c_int64_t) :: mwalk
  end type

  type, bind(C) :: sqmc_heg_cfg
    integer(c_int32_t) :: n_dim, norb, nup, ndn
    real(c_double) :: length_cell
    type(c_ptr) :: k_vectors           ! real(c_double) k_vectors(n_dim, norb), as in module heg
    integer(c_int32_t) :: rng_mode
    integer(c_int32_t) :: irand_seed(4)
    integer(c_int64_t) :: mwalk
  end type

  type, bind(C) :: sqmc_hubbard_cfg    ! hamiltonian_type 'hubbard2': the scalars of read_hubbard
    integer(c_int32_t) :: l_x, l_y, pbc, nup, ndn
    real(c_double) :: t, U
    integer(c_int32_t) :: rng_mode
    integer(c_int32_t) :: irand_seed(4)
    integer(c_int64_t) :: mwalk
  end type

  type, bind(C) :: sqmc_step_params
    real(c_double) :: tau, e_trial, reweight_factor_inv, r_initiator, min_wt, always_spawn_cutoff_wt
    integer(c_int32_t) :: initiator_power, initiator_min_distance, c_t_initiator, semistochastic, reached_w_abs_gen
    integer(c_int32_t) :: reserved
  end type

  ! the tables of setup_efficient_heatbath (chemistry.f90:1002-1225) as the reference holds them: c_loc of its module arrays
  type, bind(C) :: sqmc_heatbath_tables
    integer(c_int32_t) :: norb, reserved
    type(c_ptr) :: one_orbital_probabilities, two_orbital_probabilities
    type(c_ptr) :: three_orbital_probabilities_same_spin, three_orbital_probabilities_opposite_spin
    type(c_ptr) :: J_three_orbital_probabilities_same_spin, J_three_orbital_probabilities_opposite_spin
    type(c_ptr) :: q_three_orbital_probabilities_same_spin, q_three_orbital_probabilities_opposite_spin
    integer(c_int64_t) :: size_same, size_opposite
    type(c_ptr) :: four_orbital_probabilities_same_spin, four_orbital_probabilities_opposite_spin
    type(c_ptr) :: J_four_orbital_probabilities_same_spin, J_four_orbital_probabilities_opposite_spin
    type(c_ptr) :: q_four_orbital_probabilities_same_spin, q_four_orbital_probabilities_opposite_spin
    type(c_ptr) :: Htot_same, Htot_opposite
  end type

  type, bind(C) :: sqmc_popctl
    real(c_double) :: tau_sav, tau, tau_prev, e_trial, e_est, w_abs_gen_target, w_abs_gen
    real(c_double) :: r_initiator_sav, r_initiator, initiator_rescale_power, population_control_exponent
    real(c_double) :: reweight_factor_inv, reweight_factor_inv_max, e_num_cum, e_den_cum, min_wt, always_spawn_cutoff_wt
    integer(c_int32_t) :: reached_w_abs_gen, initiator_power, initiator_min_distance, c_t_initiator, semistochastic, reserved
    integer(c_int64_t) :: istep, n_equil
  end type

  interface
    integer(c_int) function sqmc_gpu_run(ctx, pc, nsteps, stats, totals) bind(C, name='sqmc_gpu_run')
      import; type(c_ptr), value :: ctx; type(sqmc_popctl), intent(inout) :: pc; integer(c_int64_t), value :: nsteps
      type(c_ptr), value :: stats; real(c_double), intent(out) :: totals(16)
    end function
    integer(c_int) function sqmc_gpu_build_spmv_plan(ctx, n, up, dn, plan, diag, out_nnz) bind(C, name='sqmc_gpu_build_spmv_plan')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: up(*), dn(*)
      type(c_ptr), intent(out) :: plan; real(c_double), intent(out) :: diag(*); integer(c_int64_t), intent(out) :: out_nnz
    end function
    integer(c_int) function sqmc_gpu_annihilate(ctx, p, n_spawn, up, dn, wt, imp_distance, initiator, out_stats) bind(C, name='sqmc_gpu_annihilate')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; integer(c_int64_t), value :: n_spawn
      integer(c_int64_t), intent(in) :: up(*), dn(*); real(c_double), intent(in) :: wt(*)
      integer(c_int8_t), intent(in) :: imp_distance(*), initiator(*); real(c_double), intent(out) :: out_stats(16)
    end function
    integer(c_int) function sqmc_gpu_comm_unique_id(id) bind(C, name='sqmc_gpu_comm_unique_id')
      import; integer(c_int8_t), intent(out) :: id(128)
    end function
    integer(c_int) function sqmc_gpu_comm_init(ctx, id) bind(C, name='sqmc_gpu_comm_init')
      import; type(c_ptr), value :: ctx; integer(c_int8_t), intent(in) :: id(128)
    end function
    integer(c_int) function sqmc_gpu_set_owner_hash(ctx, mode) bind(C, name='sqmc_gpu_set_owner_hash')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: mode
    end function
    integer(c_int) function sqmc_gpu_davidson(plan, diag, n_states, v0, tol, evals, evecs, n_matvec) bind(C, name='sqmc_gpu_davidson')
      import; type(c_ptr), value :: plan; real(c_double), intent(in) :: diag(*); integer(c_int32_t), value :: n_states
      type(c_ptr), value :: v0; real(c_double), value :: tol; real(c_double), intent(out) :: evals(*), evecs(*); integer(c_int32_t), intent(out) :: n_matvec
    end function
    integer(c_int) function sqmc_gpu_shard_time_split(ctx, us, steps, reset) bind(C, name='sqmc_gpu_shard_time_split')
      import; type(c_ptr), value :: ctx; real(c_double), intent(out) :: us(4); integer(c_int64_t), intent(out) :: steps; integer(c_int32_t), value :: reset
    end function
    integer(c_int) function sqmc_gpu_tail_stats(ctx, bucket_steps, bucket_retries) bind(C, name='sqmc_gpu_tail_stats')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), intent(out) :: bucket_steps, bucket_retries
    end function
    integer(c_int) function sqmc_gpu_set_chained_runs(ctx, on) bind(C, name='sqmc_gpu_set_chained_runs')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: on
    end function
    integer(c_int) function sqmc_gpu_slowest_steps(ctx, us, step) bind(C, name='sqmc_gpu_slowest_steps')
      import; type(c_ptr), value :: ctx; real(c_double), intent(out) :: us(4); integer(c_int64_t), intent(out) :: step(4)
    end function
    integer(c_int) function sqmc_gpu_hci_set_active_space(ctx, core_up, core_dn, virt_up, virt_dn, mode) bind(C, name='sqmc_gpu_hci_set_active_space')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: core_up, core_dn, virt_up, virt_dn; integer(c_int32_t), value :: mode
    end function
    integer(c_int) function sqmc_gpu_hci_pt2(ctx, n_var, var_up, var_dn, coeffs, e_var, eps_pt, n_slices, delta_e, n_connections) &
        bind(C, name='sqmc_gpu_hci_pt2')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_var; integer(c_int64_t), intent(in) :: var_up(*), var_dn(*)
      real(c_double), intent(in) :: coeffs(*); real(c_double), value :: e_var, eps_pt; integer(c_int32_t), value :: n_slices
      real(c_double), intent(out) :: delta_e; integer(c_int64_t), intent(out) :: n_connections
    end function
    integer(c_int) function sqmc_gpu_set_heatbath_tables(ctx, t) bind(C, name='sqmc_gpu_set_heatbath_tables')
      import; type(c_ptr), value :: ctx; type(sqmc_heatbath_tables), intent(in) :: t
    end function
    ! setup_efficient_heatbath (chemistry.f90:1002-1225) + check_heatbath_unbiased (9330-9375) done by the library; the tables it built
    integer(c_int) function sqmc_gpu_setup_efficient_heatbath(ctx, is_heatbath_unbiased) bind(C, name='sqmc_gpu_setup_efficient_heatbath')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), intent(out) :: is_heatbath_unbiased
    end function
    integer(c_int) function sqmc_gpu_get_heatbath_tables(ctx, t, n_orb_uniq_sym) bind(C, name='sqmc_gpu_get_heatbath_tables')
      import; type(c_ptr), value :: ctx; type(sqmc_heatbath_tables), intent(out) :: t; integer(c_int32_t), intent(out) :: n_orb_uniq_sym
    end function
    integer(c_int) function sqmc_gpu_propose_heatbath_batch(ctx, n, tau, up, dn, seeds, det_j_up, det_j_dn, weight_j, seeds_after) &
        bind(C, name='sqmc_gpu_propose_heatbath_batch')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; real(c_double), value :: tau
      integer(c_int64_t), intent(in) :: up(*), dn(*); integer(c_int32_t), intent(in) :: seeds(*)
      integer(c_int64_t), intent(out) :: det_j_up(*), det_j_dn(*); real(c_double), intent(out) :: weight_j(*); integer(c_int32_t), intent(out) :: seeds_after(*)
    end function
    integer(c_int) function sqmc_gpu_comm_size(ctx, nranks) bind(C, name='sqmc_gpu_comm_size')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), intent(out) :: nranks
    end function
    integer(c_int) function sqmc_gpu_shard_step(ctx, p, out_stats) bind(C, name='sqmc_gpu_shard_step')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; real(c_double), intent(out) :: out_stats(16)
    end function
    integer(c_int) function sqmc_gpu_shard_run(ctx, pc, nsteps, stats, totals) bind(C, name='sqmc_gpu_shard_run')
      import; type(c_ptr), value :: ctx; type(sqmc_popctl), intent(inout) :: pc; integer(c_int64_t), value :: nsteps
      type(c_ptr), value :: stats; real(c_double), intent(out) :: totals(16)
    end function
    integer(c_int) function sqmc_gpu_det_owner(ctx, n, up, dn, nranks, owner) bind(C, name='sqmc_gpu_det_owner')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: up(*), dn(*)
      integer(c_int32_t), value :: nranks; integer(c_int32_t), intent(out) :: owner(*)
    end function
    integer(c_int) function sqmc_gpu_shard_config(ctx, rank, nranks, n_imp_local, global_row) bind(C, name='sqmc_gpu_shard_config')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), value :: rank, nranks; integer(c_int64_t), value :: n_imp_local
      integer(c_int32_t), intent(in) :: global_row(*)
    end function
    integer(c_int) function sqmc_gpu_shard_begin(ctx, p, x_global_dev, n_children) bind(C, name='sqmc_gpu_shard_begin')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; type(c_ptr), value :: x_global_dev
      integer(c_int64_t), intent(out) :: n_children
    end function
    integer(c_int) function sqmc_gpu_shard_pack(ctx, p, x_global_dev, send_dev, cap_records, send_counts) bind(C, name='sqmc_gpu_shard_pack')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; type(c_ptr), value :: x_global_dev, send_dev
      integer(c_int64_t), value :: cap_records; integer(c_int64_t), intent(out) :: send_counts(*)
    end function
    integer(c_int) function sqmc_gpu_shard_finish(ctx, p, recv_dev, n_recv, out_stats) bind(C, name='sqmc_gpu_shard_finish')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; type(c_ptr), value :: recv_dev
      integer(c_int64_t), value :: n_recv; real(c_double), intent(out) :: out_stats(16)
    end function
    integer(c_int) function sqmc_gpu_set_device(device) bind(C, name='sqmc_gpu_set_device')
      import; integer(c_int), value :: device
    end function
    integer(c_int) function sqmc_gpu_init_chem(cfg, ctx) bind(C, name='sqmc_gpu_init_chem')
      import; type(sqmc_chem_cfg), intent(in) :: cfg; type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function sqmc_gpu_init_heg(cfg, ctx) bind(C, name='sqmc_gpu_init_heg')
      import; type(sqmc_heg_cfg), intent(in) :: cfg; type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function sqmc_gpu_init_hubbard(cfg, ctx) bind(C, name='sqmc_gpu_init_hubbard')
      import; type(sqmc_hubbard_cfg), intent(in) :: cfg; type(c_ptr), intent(out) :: ctx
    end function
    integer(c_int) function sqmc_gpu_finalize(ctx) bind(C, name='sqmc_gpu_finalize')
      import; type(c_ptr), value :: ctx
    end function
    type(c_ptr) function sqmc_gpu_last_error() bind(C, name='sqmc_gpu_last_error')
      import
    end function
    integer(c_int) function sqmc_gpu_set_hb_tables(ctx, n_hb, hb_r, hb_s, hb_absH, n_pq, pq_ind, pq_count, max_double) &
        bind(C, name='sqmc_gpu_set_hb_tables')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_hb; integer(c_int32_t), intent(in) :: hb_r(*), hb_s(*)
      real(c_double), intent(in) :: hb_absH(*); integer(c_int32_t), value :: n_pq; integer(c_int64_t), intent(in) :: pq_ind(*)
      integer(c_int32_t), intent(in) :: pq_count(*); real(c_double), value :: max_double
    end function
    integer(c_int) function sqmc_gpu_set_projector(ctx, n_imp, nnz, row_counts, indices, values) bind(C, name='sqmc_gpu_set_projector')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_imp, nnz
      integer(c_int64_t), intent(in) :: row_counts(*), indices(*); real(c_double), intent(in) :: values(*)
    end function
    integer(c_int) function sqmc_gpu_scale_projector(ctx, ratio) bind(C, name='sqmc_gpu_scale_projector')
      import; type(c_ptr), value :: ctx; real(c_double), value :: ratio
    end function
    integer(c_int) function sqmc_gpu_set_ct_table(ctx, n, up, dn, e_num, e_den) bind(C, name='sqmc_gpu_set_ct_table')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: up(*), dn(*)
      real(c_double), intent(in) :: e_num(*), e_den(*)
    end function
    ! hf_to_psit = .true. (do_walk.f90:378): psit_ct_index = my_locations_of_psit (1849-1886), cdet_psi_t in label order (1258), diag_elems (1091-1116)
    integer(c_int) function sqmc_gpu_set_hf_to_psit(ctx, n_psit, psit_ct_index, cdet_psi_t, diag_elems, sum_order) bind(C, name='sqmc_gpu_set_hf_to_psit')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_psit; integer(c_int64_t), intent(in) :: psit_ct_index(*)
      real(c_double), intent(in) :: cdet_psi_t(*), diag_elems(*); integer(c_int32_t), value :: sum_order
    end function
    integer(c_int) function sqmc_gpu_upload_walkers(ctx, n, up, dn, wt, imp_distance, initiator, perm_sign, matrix_elements, e_num, e_den) &
        bind(C, name='sqmc_gpu_upload_walkers')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: up(*), dn(*)
      real(c_double), intent(in) :: wt(*), matrix_elements(*), e_num(*), e_den(*)
      integer(c_int8_t), intent(in) :: imp_distance(*), initiator(*), perm_sign(*)
    end function
    integer(c_int) function sqmc_gpu_num_walkers(ctx, n) bind(C, name='sqmc_gpu_num_walkers')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), intent(out) :: n
    end function
    integer(c_int) function sqmc_gpu_download_walkers(ctx, cap, n, up, dn, wt, imp_distance, initiator, matrix_elements, e_num, e_den) &
        bind(C, name='sqmc_gpu_download_walkers')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: cap; integer(c_int64_t), intent(out) :: n
      integer(c_int64_t), intent(out) :: up(*), dn(*); real(c_double), intent(out) :: wt(*), matrix_elements(*), e_num(*), e_den(*)
      integer(c_int8_t), intent(out) :: imp_distance(*), initiator(*)
    end function
    integer(c_int) function sqmc_gpu_step(ctx, p, out_stats) bind(C, name='sqmc_gpu_step')
      import; type(c_ptr), value :: ctx; type(sqmc_step_params), intent(in) :: p; real(c_double), intent(out) :: out_stats(16)
    end function
    integer(c_int) function sqmc_gpu_get_rng(ctx, seed) bind(C, name='sqmc_gpu_get_rng')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), intent(out) :: seed(4)
    end function
    integer(c_int) function sqmc_gpu_set_rng(ctx, seed) bind(C, name='sqmc_gpu_set_rng')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), intent(in) :: seed(4)
    end function
    integer(c_int) function sqmc_gpu_spmv_prepare(n, row_counts, indices, values, plan) bind(C, name='sqmc_gpu_spmv_prepare')
      import; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: row_counts(*), indices(*)
      real(c_double), intent(in) :: values(*); type(c_ptr), intent(out) :: plan
    end function
    integer(c_int) function sqmc_gpu_spmv_apply(plan, x, y, on_device) bind(C, name='sqmc_gpu_spmv_apply')
      import; type(c_ptr), value :: plan; real(c_double), intent(in) :: x(*); real(c_double), intent(out) :: y(*); integer(c_int), value :: on_device
    end function
    integer(c_int) function sqmc_gpu_spmv_free(plan) bind(C, name='sqmc_gpu_spmv_free')
      import; type(c_ptr), value :: plan
    end function
    integer(c_int) function sqmc_gpu_spmv_sym_upper(n, row_counts, indices, values, x, y) bind(C, name='sqmc_gpu_spmv_sym_upper')
      import; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: row_counts(*), indices(*)
      real(c_double), intent(in) :: values(*), x(*); real(c_double), intent(out) :: y(*)
    end function
    integer(c_int) function sqmc_gpu_hamiltonian_batch(ctx, n, iu, id, ju, jd, h) bind(C, name='sqmc_gpu_hamiltonian_batch')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: iu(*), id(*), ju(*), jd(*)
      real(c_double), intent(out) :: h(*)
    end function
    integer(c_int) function sqmc_gpu_hamiltonian_chem_batch(ctx, n, iu, id, ju, jd, h) bind(C, name='sqmc_gpu_hamiltonian_chem_batch')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: iu(*), id(*), ju(*), jd(*)
      real(c_double), intent(out) :: h(*)
    end function
    integer(c_int) function sqmc_gpu_build_sparse_ham(ctx, n, up, dn, nnz, row_counts, indices, values) bind(C, name='sqmc_gpu_build_sparse_ham')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; integer(c_int64_t), intent(in) :: up(*), dn(*)
      integer(c_int64_t), intent(out) :: nnz; type(c_ptr), intent(out) :: row_counts, indices, values
    end function
    integer(c_int) function sqmc_gpu_propose_batch(ctx, n, tau, up, dn, seeds, ju, jd, weight_j, seeds_after) bind(C, name='sqmc_gpu_propose_batch')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n; real(c_double), value :: tau
      integer(c_int64_t), intent(in) :: up(*), dn(*); integer(c_int32_t), intent(in) :: seeds(*)
      integer(c_int64_t), intent(out) :: ju(*), jd(*); real(c_double), intent(out) :: weight_j(*); integer(c_int32_t), intent(out) :: seeds_after(*)
    end function
    integer(c_int) function sqmc_gpu_hci_connections(ctx, n_ref, ref_up, ref_dn, coeffs, eps, diag_mode, out_n, out_up, out_dn, out_num, out_den) &
        bind(C, name='sqmc_gpu_hci_connections')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_ref; integer(c_int64_t), intent(in) :: ref_up(*), ref_dn(*)
      real(c_double), intent(in) :: coeffs(*); real(c_double), value :: eps; integer(c_int), value :: diag_mode
      integer(c_int64_t), intent(out) :: out_n; type(c_ptr), intent(out) :: out_up, out_dn, out_num, out_den
    end function
    integer(c_int) function sqmc_gpu_hci_connections_slice(ctx, n_ref, ref_up, ref_dn, coeffs, eps, diag_mode, slice, n_slices, out_n, out_up, out_dn, &
        out_num, out_den) bind(C, name='sqmc_gpu_hci_connections_slice')
      import; type(c_ptr), value :: ctx; integer(c_int64_t), value :: n_ref; integer(c_int64_t), intent(in) :: ref_up(*), ref_dn(*)
      real(c_double), intent(in) :: coeffs(*); real(c_double), value :: eps; integer(c_int), value :: diag_mode
      integer(c_int32_t), value :: slice, n_slices
      integer(c_int64_t), intent(out) :: out_n; type(c_ptr), intent(out) :: out_up, out_dn, out_num, out_den
    end function
    subroutine sqmc_gpu_free(p) bind(C, name='sqmc_gpu_free')
      import; type(c_ptr), value :: p
    end subroutine
    integer(c_int) function sqmc_gpu_set_timing(ctx, on) bind(C, name='sqmc_gpu_set_timing')
      import; type(c_ptr), value :: ctx; integer(c_int), value :: on
    end function
    integer(c_int) function sqmc_gpu_get_timing(ctx, n, names, ms) bind(C, name='sqmc_gpu_get_timing')
      import; type(c_ptr), value :: ctx; integer(c_int32_t), intent(out) :: n; type(c_ptr), intent(out) :: names(*); real(c_float), intent(out) :: ms(*)
    end function
  end interface

contains

  !> Turns a non-zero status into the reference's way of failing: print the text and stop
  !> (the reference stops with these very texts: 'nwalk>MWALK' do_walk.f90:3690, 'my_nwalk=0' :2492, ...).
  subroutine sqmc_gpu_check(status, where)
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i
    if (status == 0) return
    p = sqmc_gpu_last_error()
    write(6, '(a,a,a,i4)', advance='no') 'sqmc_gpu: ', where, ' failed with status', status
    if (c_associated(p)) then
      call c_f_pointer(p, msg, [512])
      write(6, '(a)', advance='no') ': '
      do i = 1, 512
        if (msg(i) == c_null_char) exit
        write(6, '(a)', advance='no') msg(i)
      enddo
    endif
    write(6, *)
    stop 'sqmc_gpu call failed'
  end subroutine

end module sqmc_gpu_mod
