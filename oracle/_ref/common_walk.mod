﻿!mod$ v1 sum:3c8be1c898d4d369
!need$ 34bfdfda242a4e50 n types
module common_walk
use types,only:rk
use types,only:ik
use types,only:ik_vec
use types,only:i1b
use types,only:i8b
integer(8)::mwalk
integer(4)::nwalk
integer(4)::n_connected_dets
integer(4)::m_connected_dets
integer(4),allocatable::walk_det(:)
integer(4),allocatable::connected_dets(:)
integer(16),allocatable::walk_dets_up(:)
integer(16),allocatable::walk_dets_dn(:)
integer(1),allocatable::imp_distance(:)
integer(4),allocatable::connected_dets_sign(:)
real(8),allocatable::walk_wt(:)
real(8),allocatable::walk_wt_fn(:)
real(8),allocatable::matrix_elements(:)
real(8),allocatable::e_num_walker(:)
real(8),allocatable::e_den_walker(:)
end
