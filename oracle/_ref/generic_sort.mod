﻿!mod$ v1 sum:821e2c31b61140dc
!need$ 34bfdfda242a4e50 n types
module generic_sort
use types,only:rk
use types,only:ik
use types,only:ik_vec
private::rk
private::ik
private::ik_vec
private::sort_by_first_argument_int_int
private::sort_by_first_argument_int_rk
private::sort_by_first_argument_rk_int
private::sort_by_first_argument_ik
private::shell_sort_simple_int
private::shell_sort_int
private::quick_sort_int
private::quick_sort_i8b
private::shell_sort_int_int
private::shell_sort_real_rank2
private::shell_sort_real_rank1_int_rank1_int_rank1
private::shell_sort_real_rank1_ik_vec_rank1_ik_vec_rank1
interface sort
procedure::shell_sort_int
procedure::shell_sort_simple_int
procedure::shell_sort_int_int
procedure::shell_sort_real_rank2
procedure::shell_sort_real_rank1_int_rank1_int_rank1
procedure::shell_sort_real_rank1_ik_vec_rank1_ik_vec_rank1
end interface
interface quick_sort
procedure::quick_sort_int
procedure::quick_sort_i8b
end interface
interface sort_by_first_argument
procedure::sort_by_first_argument_int_int
procedure::sort_by_first_argument_ik
procedure::sort_by_first_argument_int_rk
procedure::sort_by_first_argument_rk_int
end interface
contains
subroutine sort_by_first_argument_int_int(n,list_int_sort_by,list_int)
integer(4),intent(in)::n
integer(4),intent(inout)::list_int_sort_by(1_8:int(n,kind=8))
integer(4),intent(inout)::list_int(1_8:int(n,kind=8))
end
subroutine sort_by_first_argument_int_rk(n,list_rk_sort_by,list_rk)
integer(4),intent(in)::n
integer(4),intent(inout)::list_rk_sort_by(1_8:int(n,kind=8))
real(8),intent(inout)::list_rk(1_8:int(n,kind=8))
end
subroutine sort_by_first_argument_rk_int(n,list_rk_sort_by,list_rk)
integer(4),intent(in)::n
real(8),intent(inout)::list_rk_sort_by(1_8:int(n,kind=8))
integer(4),intent(inout)::list_rk(1_8:int(n,kind=8))
end
subroutine sort_by_first_argument_ik(n,list_int_sort_by,list_int)
integer(4),intent(in)::n
integer(16),intent(inout)::list_int_sort_by(1_8:int(n,kind=8))
integer(4),intent(inout)::list_int(1_8:int(n,kind=8))
end
subroutine shell_sort_simple_int(n,list_int)
integer(4),intent(in)::n
integer(4),intent(inout)::list_int(1_8:int(n,kind=8))
end
subroutine shell_sort_int(n,list_int,num_ops)
integer(4),intent(in)::n
integer(4),intent(inout)::list_int(1_8:int(n,kind=8))
integer(4),intent(out)::num_ops
end
recursive subroutine quick_sort_int(left_in,right,list_int,list_real)
integer(4),intent(in),optional::left_in
integer(4),intent(in)::right
integer(4),intent(inout)::list_int(:)
real(8),intent(inout)::list_real(:)
end
recursive subroutine quick_sort_i8b(left_in,right,list_int,list_real)
integer(8),intent(in),optional::left_in
integer(8),intent(in)::right
integer(8),intent(inout)::list_int(:)
real(8),intent(inout)::list_real(:)
end
subroutine shell_sort_int_int(list_int1,list_int2)
integer(16),intent(inout)::list_int1(:)
integer(16),intent(inout)::list_int2(:)
end
subroutine shell_sort_real_rank2(list_real)
real(8),intent(inout)::list_real(:,:)
end
subroutine shell_sort_real_rank1_int_rank1_int_rank1(list_real,list_int1,list_int2,consider_sign)
real(8),intent(inout)::list_real(:)
integer(16),intent(inout)::list_int1(:)
integer(16),intent(inout)::list_int2(:)
logical(4),optional::consider_sign
end
subroutine shell_sort_real_rank1_int_rank1(list_real,list_int1,consider_sign)
real(8),intent(inout)::list_real(:)
integer(4),intent(inout)::list_int1(:)
logical(4),optional::consider_sign
end
subroutine shell_sort_real_rank1_ik_vec_rank1_ik_vec_rank1(list_real,list_int1,list_int2,consider_sign)
real(8),intent(inout)::list_real(:)
type(ik_vec),intent(inout)::list_int1(:)
type(ik_vec),intent(inout)::list_int2(:)
logical(4),optional::consider_sign
end
end
