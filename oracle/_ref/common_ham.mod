﻿!mod$ v1 sum:4d3bc4320e30d618
!need$ 34bfdfda242a4e50 n types
module common_ham
use types,only:rk
integer(4)::nelec
integer(4)::nup
integer(4)::ndn
integer(4)::norb
integer(4)::n_core_orb
integer(4)::ndet
integer(4)::diagonalize_ham
real(8)::hf_energy
real(8)::max_energy
real(8)::energy_exact
real(8)::diagonal_ham_lowest
real(8)::diagonal_ham_highest
character(16_8,1)::hamiltonian_type
end
