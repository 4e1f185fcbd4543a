﻿!mod$ v1 sum:bc21d31206536f60
!need$ 1774047b7d99e66e n tools
!need$ 821e2c31b61140dc n generic_sort
!need$ 0bde2ac47243ead2 i iso_c_binding
!need$ 34bfdfda242a4e50 n types
module ref_shim
use types,only:ik
use tools,only:permutation_factor
use tools,only:permutation_factor2
use tools,only:count_excitations
use tools,only:random_int
use tools,only:merge_sort2_up_dn
use generic_sort,only:sort
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
use generic_sort,only:generic_sort$generic_sort$shell_sort_real_rank2=>shell_sort_real_rank2
private::generic_sort$generic_sort$shell_sort_real_rank2
contains
subroutine ref_setrn(seed) bind(c,name="ref_setrn")
integer(4),intent(in)::seed(1_8:4_8)
end
subroutine ref_savern(seed) bind(c,name="ref_savern")
integer(4),intent(out)::seed(1_8:4_8)
end
subroutine ref_rannyu_fill(n,out) bind(c,name="ref_rannyu_fill")
integer(4),value::n
real(8),intent(out)::out(1_8:int(n,kind=8))
end
subroutine ref_random_int_fill(nmax,n,out) bind(c,name="ref_random_int_fill")
integer(4),value::nmax
integer(4),value::n
integer(4),intent(out)::out(1_8:int(n,kind=8))
end
subroutine ref_permutation_factor(n,a,b,out) bind(c,name="ref_permutation_factor")
integer(4),value::n
integer(8),intent(in)::a(1_8:int(n,kind=8))
integer(8),intent(in)::b(1_8:int(n,kind=8))
integer(4),intent(out)::out(1_8:int(n,kind=8))
end
subroutine ref_permutation_factor2(n,a,b,out) bind(c,name="ref_permutation_factor2")
integer(4),value::n
integer(8),intent(in)::a(1_8:int(n,kind=8))
integer(8),intent(in)::b(1_8:int(n,kind=8))
integer(4),intent(out)::out(1_8:5_8,1_8:int(n,kind=8))
end
subroutine ref_merge_sort2_up_dn(n,up,dn,iorder) bind(c,name="ref_merge_sort2_up_dn")
integer(4),value::n
integer(8),intent(inout)::up(1_8:int(n,kind=8))
integer(8),intent(inout)::dn(1_8:int(n,kind=8))
integer(4),intent(out)::iorder(1_8:int(n,kind=8))
end
subroutine ref_sort_real_rank2(ndim,n,arr) bind(c,name="ref_sort_real_rank2")
integer(4),value::ndim
integer(4),value::n
real(8),intent(inout)::arr(1_8:int(ndim,kind=8),1_8:int(n,kind=8))
end
end
