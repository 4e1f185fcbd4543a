!> A Fortran host calling the HIP library the way davidson_sparse would
!> (more_tools.f90:2115: call fast_sparse_matrix_multiply_upper_triangular(n, H_indices, H_nonzero_elements, H_values, v, Hv)):
!> builds a small symmetric matrix in the reference's storage, multiplies on the GPU,
!> and compares with the reference loop written out below.
program example_spmv
  use iso_c_binding
  use sqmc_gpu_mod
  implicit none
  integer, parameter :: n = 2000
  integer(c_int64_t), allocatable :: counts(:), idx(:)
  real(c_double), allocatable :: val(:), x(:), y(:), yref(:)
  integer(c_int64_t) :: nnz, k, i, j, m
  type(c_ptr) :: plan
  real(c_double) :: err

  allocate(counts(n), idx(8*n), val(8*n), x(n), y(n), yref(n))
  nnz = 0
  do i = 1, n
    counts(i) = 0
    nnz = nnz + 1; idx(nnz) = i; val(nnz) = 1._c_double + 0.001_c_double*i; counts(i) = counts(i) + 1   ! diagonal first
    do j = 1, 5
      m = i - j*j
      if (m >= 1) then
        nnz = nnz + 1; idx(nnz) = m; val(nnz) = 1._c_double/(i + m); counts(i) = counts(i) + 1
      endif
    enddo
    x(i) = sin(0.37_c_double*i)
  enddo
  ! reference loop (more_tools.f90:3645-3655)
  yref = 0
  k = 0
  do i = 1, n
    do j = 1, counts(i)
      k = k + 1
      m = idx(k)
      yref(i) = yref(i) + val(k)*x(m)
      if (i /= m) yref(m) = yref(m) + val(k)*x(i)
    enddo
  enddo
  call sqmc_gpu_check(sqmc_gpu_set_device(0_c_int), 'set_device')
  call sqmc_gpu_check(sqmc_gpu_spmv_prepare(int(n, c_int64_t), counts, idx, val, plan), 'spmv_prepare')
  call sqmc_gpu_check(sqmc_gpu_spmv_apply(plan, x, y, 0_c_int), 'spmv_apply')
  call sqmc_gpu_check(sqmc_gpu_spmv_free(plan), 'spmv_free')
  err = maxval(abs(y - yref))
  write(6, '(a,es10.2)') 'fortran host: max |y_gpu - y_ref| =', err
  if (err > 1.e-12_c_double) stop 'MISMATCH'
  write(6, '(a)') 'fortran host: OK'
end program
