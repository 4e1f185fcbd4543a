﻿!mod$ v1 sum:1774047b7d99e66e
!need$ 2b0382f44fcb4917 n common_run
!need$ 34bfdfda242a4e50 n types
module tools
use types,only:ik
use types,only:ik_vec
use types,only:i8b
use types,only:i16b
use types,only:rk
use common_run,only:diag_elem_info
private::ik
private::ik_vec
private::i8b
private::i16b
private::rk
private::diag_elem_info
private::n_choose_k_int4
private::n_choose_k_int16
private::merge_real
private::merge_original_with_spawned3_nowts
private::merge_original_with_spawned3_nowts_replace
private::merge_original_with_spawned3_num_denom
private::merge_original_with_spawned3_num_denom_diag_elems_info
private::merge_original_with_spawned3_5items
private::merge_original_with_spawned3_num
private::merge_original_with_spawned3_num_denom_replace
private::merge_original_with_spawned3_num_denom_diag_elems_info_replace
private::merge_original_with_spawned3_5items_replace
private::merge_original_with_spawned3_add_wts
private::verbose_allocate_int
private::verbose_allocate_ik
private::verbose_allocate_rk
private::verbose_allocate_rk2
private::verbose_allocate_int_test
private::sort_and_merge_int
private::sort_and_merge_int_rk
private::sort_and_merge_det
private::alloc_i
private::alloc_ik
private::alloc_rk_1
private::alloc_rk_2
private::alloc_diag_elem_info
interface merge_original_with_spawned3
procedure::merge_original_with_spawned3_nowts
procedure::merge_original_with_spawned3_nowts_replace
procedure::merge_original_with_spawned3_num_denom
procedure::merge_original_with_spawned3_num_denom_diag_elems_info
procedure::merge_original_with_spawned3_num
procedure::merge_original_with_spawned3_5items
procedure::merge_original_with_spawned3_num_denom_replace
procedure::merge_original_with_spawned3_num_denom_diag_elems_info_replace
procedure::merge_original_with_spawned3_5items_replace
procedure::merge_original_with_spawned3_add_wts
end interface
interface sort_and_merge
procedure::sort_and_merge_int
procedure::sort_and_merge_det
procedure::sort_and_merge_int_rk
end interface
interface n_choose_k
procedure::n_choose_k_int4
procedure::n_choose_k_int16
end interface
interface verbose_allocate
procedure::verbose_allocate_int
procedure::verbose_allocate_ik
procedure::verbose_allocate_rk
procedure::verbose_allocate_rk2
procedure::verbose_allocate_int_test
end interface
interface alloc
procedure::alloc_i
procedure::alloc_ik
procedure::alloc_rk_1
procedure::alloc_rk_2
procedure::alloc_diag_elem_info
end interface
contains
function n_choose_k_int4(n,k)
integer(4),intent(in)::n
integer(4),intent(in)::k
integer(16)::n_choose_k_int4
end
function n_choose_k_int16(n,k)
integer(16),intent(in)::n
integer(16),intent(in)::k
integer(16)::n_choose_k_int16
end
function random_int(n)
integer(4),intent(in)::n
integer(4)::random_int
end
subroutine count_bits_naive(bit_pattern,n_bits,n_set)
integer(16),intent(in)::bit_pattern
integer(4),intent(in)::n_bits
integer(4),intent(out)::n_set
end
subroutine count_bits_imp1(bit_pattern,n_set)
integer(16),intent(inout)::bit_pattern
integer(4),intent(out)::n_set
end
recursive subroutine merge_sort2_up_dn(key_up,key_dn,iorder,nwalk,temp_i16_up,temp_i16_dn,temp_i_2)
integer(4),intent(in)::nwalk
integer(16),intent(inout)::key_up(1_8:int(nwalk,kind=8))
integer(16),intent(inout)::key_dn(1_8:int(nwalk,kind=8))
integer(4),intent(inout)::iorder(1_8:int(nwalk,kind=8))
integer(16),intent(out)::temp_i16_up(1_8:int((nwalk+1_4)/2_4,kind=8))
integer(16),intent(out)::temp_i16_dn(1_8:int((nwalk+1_4)/2_4,kind=8))
integer(4),intent(out)::temp_i_2(1_8:int((nwalk+1_4)/2_4,kind=8))
end
recursive subroutine merge_sort_real(key_real,iorder,nelts,temp_real,temp_i_2)
integer(4),intent(in)::nelts
real(8),intent(inout)::key_real(1_8:int(nelts,kind=8))
integer(4),intent(inout)::iorder(1_8:int(nelts,kind=8))
real(8),intent(out)::temp_real(1_8:int((nelts+1_4)/2_4,kind=8))
integer(4),intent(out)::temp_i_2(1_8:int((nelts+1_4)/2_4,kind=8))
end
subroutine merge2_up_dn(a_up,a_dn,a2,na,b_up,b_dn,b2,nb,c_up,c_dn,c2,nc)
integer(4),intent(in)::na
integer(16),intent(inout)::a_up(1_8:int(na,kind=8))
integer(16),intent(inout)::a_dn(1_8:int(na,kind=8))
integer(4),intent(inout)::a2(1_8:int(na,kind=8))
integer(4),intent(in)::nb
integer(16),intent(in)::b_up(1_8:int(nb,kind=8))
integer(16),intent(in)::b_dn(1_8:int(nb,kind=8))
integer(4),intent(in)::b2(1_8:int(nb,kind=8))
integer(4),intent(in)::nc
integer(16),intent(inout)::c_up(1_8:int(nc,kind=8))
integer(16),intent(inout)::c_dn(1_8:int(nc,kind=8))
integer(4),intent(inout)::c2(1_8:int(nc,kind=8))
end
subroutine merge_real(a,a2,na,b,b2,nb,c,c2,nc)
integer(4),intent(in)::na
real(8),intent(inout)::a(1_8:int(na,kind=8))
integer(4),intent(inout)::a2(1_8:int(na,kind=8))
integer(4),intent(in)::nb
real(8),intent(in)::b(1_8:int(nb,kind=8))
integer(4),intent(in)::b2(1_8:int(nb,kind=8))
integer(4),intent(in)::nc
real(8),intent(inout)::c(1_8:int(nc,kind=8))
integer(4),intent(inout)::c2(1_8:int(nc,kind=8))
end
subroutine merge_original_with_spawned3_nowts(old_up,old_dn,nwalk,new_up,new_dn)
integer(4),intent(inout)::nwalk
integer(16),intent(in)::old_up(1_8:int(nwalk,kind=8))
integer(16),intent(in)::old_dn(1_8:int(nwalk,kind=8))
integer(16),intent(out)::new_up(:)
integer(16),intent(out)::new_dn(:)
end
subroutine merge_original_with_spawned3_nowts_replace(nwalk,dets_up,dets_dn)
integer(4),intent(inout)::nwalk
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
end
subroutine merge_original_with_spawned3_num_denom(old_up,old_dn,nwalk,new_up,new_dn,old_e_mix_num,old_e_mix_den,e_mix_num,e_mix_den)
integer(4),intent(inout)::nwalk
integer(16),intent(in)::old_up(1_8:int(nwalk,kind=8))
integer(16),intent(in)::old_dn(1_8:int(nwalk,kind=8))
integer(16),allocatable,intent(out)::new_up(:)
integer(16),allocatable,intent(out)::new_dn(:)
real(8),intent(in)::old_e_mix_num(1_8:int(nwalk,kind=8))
real(8),intent(in)::old_e_mix_den(1_8:int(nwalk,kind=8))
real(8),allocatable,intent(out)::e_mix_num(:)
real(8),allocatable,intent(out)::e_mix_den(:)
end
subroutine merge_original_with_spawned3_num_denom_diag_elems_info(old_up,old_dn,nwalk,new_up,new_dn,old_e_mix_num,old_e_mix_den,old_diag_elems_info,e_mix_num,e_mix_den,diag_elems_info)
use common_run,only:diag_elem_info
integer(4),intent(inout)::nwalk
integer(16),intent(in)::old_up(1_8:int(nwalk,kind=8))
integer(16),intent(in)::old_dn(1_8:int(nwalk,kind=8))
integer(16),allocatable,intent(out)::new_up(:)
integer(16),allocatable,intent(out)::new_dn(:)
real(8),intent(in)::old_e_mix_num(1_8:int(nwalk,kind=8))
real(8),intent(in)::old_e_mix_den(1_8:int(nwalk,kind=8))
type(diag_elem_info),intent(in)::old_diag_elems_info(:)
real(8),allocatable,intent(out)::e_mix_num(:)
real(8),allocatable,intent(out)::e_mix_den(:)
type(diag_elem_info),allocatable,intent(out)::diag_elems_info(:)
end
subroutine merge_original_with_spawned3_5items(old_up,old_dn,nwalk,new_up,new_dn,old_e_mix_num,old_e_mix_den,old_diag_elems_info,old_term1_big,old_term2_big,e_mix_num,e_mix_den,diag_elems_info,term1_big,term2_big)
use common_run,only:diag_elem_info
integer(4),intent(inout)::nwalk
integer(16),intent(in)::old_up(1_8:int(nwalk,kind=8))
integer(16),intent(in)::old_dn(1_8:int(nwalk,kind=8))
integer(16),allocatable,intent(out)::new_up(:)
integer(16),allocatable,intent(out)::new_dn(:)
real(8),intent(in)::old_e_mix_num(1_8:int(nwalk,kind=8))
real(8),intent(in)::old_e_mix_den(1_8:int(nwalk,kind=8))
type(diag_elem_info),intent(in)::old_diag_elems_info(:)
real(8),intent(in)::old_term1_big(:)
real(8),intent(in)::old_term2_big(:)
real(8),allocatable,intent(out)::e_mix_num(:)
real(8),allocatable,intent(out)::e_mix_den(:)
type(diag_elem_info),allocatable,intent(out)::diag_elems_info(:)
real(8),allocatable,intent(out)::term1_big(:)
real(8),allocatable,intent(out)::term2_big(:)
end
subroutine merge_original_with_spawned3_num(old_up,old_dn,nwalk,new_up,new_dn,old_e_mix_num,e_mix_num)
integer(4),intent(inout)::nwalk
integer(16),intent(in)::old_up(1_8:int(nwalk,kind=8))
integer(16),intent(in)::old_dn(1_8:int(nwalk,kind=8))
integer(16),allocatable,intent(out)::new_up(:)
integer(16),allocatable,intent(out)::new_dn(:)
real(8),intent(in)::old_e_mix_num(1_8:int(nwalk,kind=8))
real(8),allocatable,intent(out)::e_mix_num(:)
end
subroutine merge_original_with_spawned3_num_denom_replace(nwalk,dets_up,dets_dn,e_mix_num,e_mix_den)
integer(4),intent(inout)::nwalk
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
real(8),intent(inout)::e_mix_num(:)
real(8),intent(inout),optional::e_mix_den(:)
end
subroutine merge_original_with_spawned3_num_denom_diag_elems_info_replace(nwalk,dets_up,dets_dn,e_mix_num,e_mix_den,diag_elems_info)
use common_run,only:diag_elem_info
integer(4),intent(inout)::nwalk
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
real(8),intent(inout)::e_mix_num(:)
real(8),intent(inout)::e_mix_den(:)
type(diag_elem_info),intent(inout)::diag_elems_info(:)
end
subroutine merge_original_with_spawned3_5items_replace(nwalk,dets_up,dets_dn,e_mix_num,e_mix_den,diag_elems_info,term1_big,term2_big)
use common_run,only:diag_elem_info
integer(4),intent(inout)::nwalk
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
real(8),intent(inout)::e_mix_num(:)
real(8),intent(inout)::e_mix_den(:)
type(diag_elem_info),intent(inout)::diag_elems_info(:)
real(8),intent(inout)::term1_big(:)
real(8),intent(inout)::term2_big(:)
end
subroutine merge_original_with_spawned3_add_wts(dets_up,dets_dn,walk_wt,nwalk)
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
real(8),intent(inout)::walk_wt(:)
integer(4),intent(inout)::nwalk
end
function get_free_memory()
real(4)::get_free_memory
end
subroutine verbose_allocate_int(variable,length)
integer(4),allocatable,intent(inout)::variable(:)
integer(4),intent(in)::length
end
subroutine verbose_allocate_ik(variable,length)
integer(16),allocatable,intent(inout)::variable(:)
integer(4),intent(in)::length
end
subroutine verbose_allocate_rk(variable,length)
real(8),allocatable,intent(inout)::variable(:)
integer(4),intent(in)::length
end
subroutine verbose_allocate_rk2(variable,length1,length2)
real(8),allocatable,intent(inout)::variable(:,:)
integer(4),intent(in)::length1
integer(4),intent(in)::length2
end
subroutine verbose_allocate_int_test(variable,length,ratio)
integer(4),allocatable,intent(inout)::variable(:)
integer(4),intent(in)::length
real(8),intent(out)::ratio
end
function permutation_factor(det1,det2)
integer(16),intent(in)::det1
integer(16),intent(in)::det2
integer(4)::permutation_factor
end
subroutine permutation_factor2(det_i,det_j,gamma,first_i_bit,second_i_bit,first_j_bit,second_j_bit,nosign)
integer(16),intent(in)::det_i
integer(16),intent(in)::det_j
integer(4),intent(out)::gamma
integer(4),intent(out)::first_i_bit
integer(4),intent(out)::second_i_bit
integer(4),intent(out)::first_j_bit
integer(4),intent(out)::second_j_bit
logical(4),intent(in),optional::nosign
end
function count_excitations(det1,det2)
integer(16),intent(in)::det1
integer(16),intent(in)::det2
integer(4)::count_excitations
end
subroutine print_excitation_levels_and_wts(n_det,dets_up,dets_dn,lowest_eigenvector,norb,orbital_symmetries,hf_up_remote,hf_dn_remote)
integer(4),intent(in)::n_det
integer(16),intent(in)::dets_up(:)
integer(16),intent(in)::dets_dn(:)
real(8),intent(in)::lowest_eigenvector(:)
integer(4),intent(in)::norb
integer(4),intent(in),optional::orbital_symmetries(:)
integer(16),intent(in),optional::hf_up_remote
integer(16),intent(in),optional::hf_dn_remote
end
subroutine matrix_inversion(n,a,x,b,b0,ipiv,work)
integer(4),intent(in)::n
real(8),intent(in)::a(:,:)
real(8),intent(out)::x(:)
real(8),intent(in)::b(:)
real(8),intent(out)::b0(:,:)
integer(4),intent(out)::ipiv(:)
real(8),intent(out)::work(:)
end
function do_once()
logical(4)::do_once
end
function do_n_times(n)
integer(4),intent(in)::n
logical(4)::do_n_times
end
subroutine choose_det_with_prob_prop_to_abs_wt(iwalk,tot_wt)
integer(4),intent(out)::iwalk
real(8),intent(out)::tot_wt
end
subroutine sort_and_merge_count_repeats(n,arr,counts)
integer(4),intent(inout)::n
integer(4),intent(inout)::arr(:)
integer(4),intent(out)::counts(:)
end
subroutine sort_and_merge_int(n,arr)
integer(4),intent(inout)::n
integer(4),intent(inout)::arr(:)
end
subroutine sort_and_merge_int_rk(n,arr,arr_rk)
integer(4),intent(inout)::n
integer(4),intent(inout)::arr(:)
real(8),intent(inout)::arr_rk(:)
end
subroutine sort_and_merge_det(n_dets,dets_up,dets_dn)
integer(4),intent(inout)::n_dets
integer(16),intent(inout)::dets_up(:)
integer(16),intent(inout)::dets_dn(:)
end
subroutine welford(n,x,m,s,var)
integer(4),intent(in)::n
real(8),intent(in)::x
real(8),intent(inout)::m
real(8),intent(inout)::s
real(8),intent(out)::var
end
function round_r(r,n)
real(8),intent(in)::r
integer(4),intent(in)::n
real(8)::round_r
end
function round_i(i,n)
integer(4),intent(in)::i
integer(4),intent(in)::n
integer(4)::round_i
end
subroutine alloc_i(array_name,array,dim1)
character(*,1),intent(in)::array_name
integer(4),allocatable,intent(inout)::array(:)
integer(8),intent(in)::dim1
end
subroutine alloc_ik(array_name,array,dim1)
character(*,1),intent(in)::array_name
integer(16),allocatable,intent(inout)::array(:)
integer(8),intent(in)::dim1
end
subroutine alloc_rk_1(array_name,array,dim1)
character(*,1),intent(in)::array_name
real(8),allocatable::array(:)
integer(8),intent(in)::dim1
end
subroutine alloc_rk_2(array_name,array,dim1,dim2)
character(*,1),intent(in)::array_name
real(8),allocatable::array(:,:)
integer(8),intent(in)::dim1
integer(8),intent(in)::dim2
end
subroutine alloc_diag_elem_info(array_name,array,dim1)
character(*,1),intent(in)::array_name
type(diag_elem_info),allocatable,intent(inout)::array(:)
integer(8),intent(in)::dim1
end
end
