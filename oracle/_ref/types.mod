﻿!mod$ v1 sum:34bfdfda242a4e50
module types
integer(4),parameter::i1b=1_4
intrinsic::selected_int_kind
private::selected_int_kind
integer(4),parameter::i2b=2_4
integer(4),parameter::i4b=4_4
integer(4),parameter::i8b=8_4
integer(4),parameter::i16b=16_4
integer(4),parameter,private::r4b=4_4
intrinsic::selected_real_kind
private::selected_real_kind
integer(4),parameter,private::r8b=8_4
integer(4),parameter,private::r16b=16_4
integer(4),parameter::ik=16_4
integer(4),parameter::rk=8_4
integer(4),parameter::num_words=1_4
integer(4),parameter::bits_per_word=127_4
type::ik_vec
integer(16)::v(1_8:1_8)
end type
type::optional_integer
logical(4)::is_present=.false._4
integer(4)::instance
end type
type::optional_rk
logical(4)::is_present=.false._4
real(8)::instance
end type
type::int_vec
integer(4),allocatable::list(:)
integer(4)::n=0_4
integer(4)::size=0_4
contains
procedure::append
end type
type::rs_absh
integer(4)::r
integer(4)::s
real(8)::absh
end type
private::append
contains
subroutine append(this,val)
class(int_vec),intent(inout)::this
integer(4),intent(in)::val
end
end
