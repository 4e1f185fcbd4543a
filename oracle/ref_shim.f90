! ref_shim.f90 -- OUR code (not reference source): bind(C) doors into the reference
! routines that oracle/_ref compiles unmodified from /root/reference/src
! (tools.f90, rannyu.f90).  Used only by tests to pin the CPU oracle.
module ref_shim
  use iso_c_binding
  use types, only : ik
  use tools, only : permutation_factor, permutation_factor2, count_excitations, random_int, merge_sort2_up_dn
  use generic_sort, only : sort
  implicit none
contains

  subroutine ref_setrn(seed) bind(C, name='ref_setrn')
    integer(c_int), intent(in) :: seed(4)
    integer :: s(4)
    s = seed
    call setrn(s)
  end subroutine

  subroutine ref_savern(seed) bind(C, name='ref_savern')
    integer(c_int), intent(out) :: seed(4)
    integer :: s(4)
    call savern(s)
    seed = s
  end subroutine

  subroutine ref_rannyu_fill(n, out) bind(C, name='ref_rannyu_fill')
    integer(c_int), value :: n
    real(c_double), intent(out) :: out(n)
    real(c_double) :: rannyu
    integer :: i
    do i = 1, n
      out(i) = rannyu()
    enddo
  end subroutine

  subroutine ref_random_int_fill(nmax, n, out) bind(C, name='ref_random_int_fill')
    integer(c_int), value :: nmax, n
    integer(c_int), intent(out) :: out(n)
    integer :: i
    do i = 1, n
      out(i) = random_int(nmax)
    enddo
  end subroutine

  ! dets arrive as 64-bit words (hi word = 0), as on the reference's MPI wire
  ! (mpi_routines.f90:671-680)
  subroutine ref_permutation_factor(n, a, b, out) bind(C, name='ref_permutation_factor')
    integer(c_int), value :: n
    integer(c_int64_t), intent(in) :: a(n), b(n)
    integer(c_int), intent(out) :: out(n)
    integer :: i
    do i = 1, n
      out(i) = permutation_factor(int(a(i), ik), int(b(i), ik))
    enddo
  end subroutine

  subroutine ref_permutation_factor2(n, a, b, out) bind(C, name='ref_permutation_factor2')
    integer(c_int), value :: n
    integer(c_int64_t), intent(in) :: a(n), b(n)
    integer(c_int), intent(out) :: out(5, n)
    integer :: i, g, i1, i2, j1, j2
    do i = 1, n
      call permutation_factor2(int(a(i), ik), int(b(i), ik), g, i1, i2, j1, j2)
      out(:, i) = (/ g, i1, i2, j1, j2 /)
    enddo
  end subroutine

  ! stable (up,dn) merge sort used by the walk: returns the permutation (1-based)
  subroutine ref_merge_sort2_up_dn(n, up, dn, iorder) bind(C, name='ref_merge_sort2_up_dn')
    integer(c_int), value :: n
    integer(c_int64_t), intent(inout) :: up(n), dn(n)
    integer(c_int), intent(out) :: iorder(n)
    integer(ik), allocatable :: ku(:), kd(:), tu(:), td(:)
    integer, allocatable :: io(:), ti(:)
    integer :: i
    allocate(ku(n), kd(n), tu((n+1)/2), td((n+1)/2), io(n), ti((n+1)/2))
    do i = 1, n
      ku(i) = int(up(i), ik); kd(i) = int(dn(i), ik); io(i) = i
    enddo
    call merge_sort2_up_dn(ku, kd, io, n, tu, td, ti)
    do i = 1, n
      up(i) = int(ku(i), c_int64_t); dn(i) = int(kd(i), c_int64_t); iorder(i) = io(i)
    enddo
  end subroutine

  ! shell sort of column vectors by magnitude: the routine that orders the HEG k-points
  ! (generic_sort.f90:554-591, called at heg.f90:700)
  subroutine ref_sort_real_rank2(ndim, n, arr) bind(C, name='ref_sort_real_rank2')
    integer(c_int), value :: ndim, n
    real(c_double), intent(inout) :: arr(ndim, n)
    call sort(arr)
  end subroutine

end module ref_shim
