﻿!mod$ v1 sum:fa1ecce372f190be
!need$ 34bfdfda242a4e50 n types
module constants
use types,only:rk
private::rk
real(8),parameter::pi=3.141592653589793115997963468544185161590576171875_8
intrinsic::atan
private::atan
end
