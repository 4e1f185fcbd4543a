﻿!mod$ v1 sum:9bce2a0b82d6970c
!need$ 34bfdfda242a4e50 n types
module common_selected_ci
use types,only:rk
use types,only:ik
use types,only:ik_vec
use types,only:i8b
use types,only:optional_rk
use types,only:optional_integer
integer(4)::det_sel_iters
integer(4),allocatable::norb_det_sel(:)
integer(4),allocatable::n_sym_uniq_det_det_sel(:)
integer(4)::ndet_det_sel
real(8),allocatable::cdet_det_sel(:)
integer(16),allocatable::dets_up_det_sel(:)
integer(16),allocatable::dets_dn_det_sel(:)
integer(4)::lanczos_iters
integer(4)::lanczos_initiators
integer(4)::lanczos_truncate
integer(8),parameter::max_nonzero_elements=5000000000_8
intrinsic::int
logical(4)::too_big_to_store
real(8)::log_num_nonzero_elements
integer(4)::cdets
integer(4)::tdets
real(8)::eps_var_sched(1_8:30_8)
real(8)::eps_var
real(8)::eps_pt
real(8)::eps_pt_big
real(8)::target_error
logical(4)::dump_wf_var
type::sparse_mat
integer(8)::ndet=0_8
integer(8),allocatable::indices(:)
integer(8),allocatable::nonzero_elements(:)
real(8),allocatable::values(:)
end type
type(sparse_mat),save::sparse_ham
integer(4)::n_states
real(8)::eps_pt_big_energy
real(8)::n_max_connections
integer(4)::n_mc
integer(4)::n_energy_batch
logical(4)::use_hash_generation
logical(4)::get_auto_hf
integer(4)::hf_symmetry
integer(4),allocatable::up(:)
integer(4),allocatable::dn(:)
integer(4)::lz
logical(4)::g
logical(4)::u
integer(4),allocatable::irreps(:)
integer(4),allocatable::irrep_occs_up(:)
integer(4),allocatable::irrep_occs_dn(:)
integer(4)::n_irrep
logical(4)::get_natorbs
logical(4)::use_pt
logical(4)::get_greens_function
integer(4)::n_w
real(8)::w_min
real(8)::w_max
integer(4)::n_var_e_up
integer(4)::n_var_e_dn
integer(4)::n_var_orbs
integer(4),allocatable::var_orbs(:)
namelist/selected_ci/eps_var_sched,eps_var,eps_pt,eps_pt_big,eps_pt_big_energy,target_error,dump_wf_var,n_max_connections,n_states,n_mc,n_energy_batch,use_hash_generation
namelist/hf_det/up,dn,hf_symmetry,lz,g,u,n_irrep,irreps,irrep_occs_up,irrep_occs_dn
namelist/natorb/get_natorbs,use_pt
namelist/greens_function/get_greens_function,n_w,w_min,w_max
namelist/active_space/n_var_e_up,n_var_e_dn,n_var_orbs,var_orbs
end
