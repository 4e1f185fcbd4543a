﻿!mod$ v1 sum:cd6bebd1772363e1
!need$ 34bfdfda242a4e50 n types
module common_imp
use types,only:rk
use types,only:ik
use types,only:ik_vec
use types,only:i8b
logical(4)::semistochastic
integer(4)::imp_iters
integer(4),allocatable::norb_imp(:)
integer(4),allocatable::n_imp_initiators(:)
integer(4),allocatable::n_imp_truncate(:)
integer(4)::n_imp
integer(16),allocatable::imp_up(:)
integer(16),allocatable::imp_dn(:)
integer(8),allocatable::minus_tau_h_indices(:)
integer(8),allocatable::minus_tau_h_nonzero_elements(:)
real(8),allocatable::minus_tau_h_values(:)
logical(4)::diff_from_psi_t
integer(4)::size_deterministic
end
