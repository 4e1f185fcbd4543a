﻿!mod$ v1 sum:4c0798394cd6ed20
!need$ 34bfdfda242a4e50 n types
module common_psi_t
use types,only:rk
use types,only:ik
use types,only:ik_vec
integer(4)::ndeg
integer(4)::ndet_psi_t
integer(4)::ndet_psi_t_in
logical(4)::use_psit_in
logical(4)::use_psit_out
logical(4)::print_psit_wo_sqmc
logical(4)::use_psit_con_in
logical(4)::use_psit_con_out
logical(4)::use_elems_in
logical(4)::use_elems_out
character(100_4,1)::psit_in_file
character(100_4,1)::psit_out_file
character(100_4,1)::psit_con_in_file
character(100_4,1)::psit_con_out_file
character(100_4,1)::dtm_elems_in_file
character(100_4,1)::dtm_elems_out_file
real(8)::dtm_energy
real(8)::e_trial
integer(16),allocatable::dets_up_psi_t(:)
integer(16),allocatable::dets_dn_psi_t(:)
integer(16),allocatable::dets_up_psi_t_in(:)
integer(16),allocatable::dets_dn_psi_t_in(:)
integer(16),allocatable::psi_t_connected_dets_up(:)
integer(16),allocatable::psi_t_connected_dets_dn(:)
integer(4),allocatable::iwdet_psi_t(:)
real(8),allocatable::cdet_psi_t(:)
real(8),allocatable::psi_g(:)
real(8),allocatable::real_wt(:)
real(8),allocatable::cdet_psi_t_in(:)
real(8)::psi_g_energy
real(8)::psi_g_epsilon
real(8)::psi_g_epsilon_inv
integer(4)::trial_wf_iters
integer(4),allocatable::norb_trial_wf(:)
integer(4),allocatable::n_initiators_trial_wf(:)
integer(4),allocatable::n_truncate_trial_wf(:)
integer(4)::ndet_psi_t_connected
real(8),allocatable::psi_t_connected_e_loc_num(:)
real(8),allocatable::psi_t_connected_e_loc_den(:)
logical(4)::hf_to_psit
end
