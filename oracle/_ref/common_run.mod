﻿!mod$ v1 sum:2b0382f44fcb4917
!need$ 34bfdfda242a4e50 n types
!need$ 0bde2ac47243ead2 i iso_c_binding
module common_run
use types,only:rk
use types,only:ik
use types,only:ik_vec
use,intrinsic::iso_c_binding,only:c_ptr
integer(4)::nstep
integer(4)::nblk
integer(4)::nblk_eq
integer(4)::ipr
integer(4)::nwalk
integer(4)::importance_sampling
real(8)::tau
real(8)::tau_multiplier
real(8)::population_control_exponent
real(8)::reweight_factor_inv_max_multiplier
real(8)::reweight_factor_inv_max
real(8)::initiator_rescale_power
real(8)::partial_node_eps
character(16_8,1)::proposal_method
character(16_8,1)::run_type
integer(4)::max_connected_dets
integer(16),allocatable::connected_dets_up(:)
integer(16),allocatable::connected_dets_dn(:)
integer(4)::n_connected_dets_hf
real(8),allocatable::connected_matrix_elements(:)
real(8),allocatable::connected_matrix_elements_fn(:)
logical(4)::first_time
integer(4)::debug_counter
integer(4)::ndet_outside_ct
type(c_ptr)::h_psi_pointer
type(c_ptr)::psi_pointer
type(c_ptr)::qp_pointer
type(c_ptr)::ps_pointer
integer(4),allocatable::occ(:)
logical(4)::use_efficient_heatbath
type::diag_elem_info
real(8)::old_diag_elem
integer(4)::p
integer(4)::q
integer(4)::r
integer(4)::s
end type
type(diag_elem_info),allocatable::connected_diag_elems_info(:)
integer(4),parameter::input_copy_unit=111_4
end
