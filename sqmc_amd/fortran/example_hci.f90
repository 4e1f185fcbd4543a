!> A Fortran host for the variational stage of HCI on the GPU path (INTEGRATION.md, patch sketch 3):
!> what perform_hci does per iteration (hci.f90:359-520) with the three heavy pieces behind
!> sqmc_gpu_mod -- sqmc_gpu_hci_connections for find_doubly_excited + dedup (hci.f90:905-931),
!> sqmc_gpu_build_spmv_plan for generate_sparse_ham_chem_upper_triangular, sqmc_gpu_davidson for
!> davidson_sparse (more_tools.f90:2018-2244) -- and the host logic (the list bookkeeping of get_next_det_list) written here.  Tables come from a deck written by sqmc_amd.host.dump_hci_deck.
!>   usage: example_hci <deck> [eps_pt]      with eps_pt: the Epstein-Nesbet correction of state 1 (second_order_pt,
!>   hci.f90:1100-1182) in one call, sqmc_gpu_hci_pt2, on the basis the deck's context works in
module hci_host_tools
  use iso_c_binding
  implicit none
contains

  !> permutation that sorts determinants by (up, dn); bottom-up merge sort on the index
  subroutine argsort_dets(n, up, dn, order)
    integer(c_int64_t), intent(in) :: n, up(n), dn(n)
    integer(c_int64_t), intent(out) :: order(n)
    integer(c_int64_t), allocatable :: tmp(:)
    integer(c_int64_t) :: width, lo, mid, hi, a, b, k
    allocate(tmp(n))
    do k = 1, n
      order(k) = k
    enddo
    width = 1
    do while (width < n)
      lo = 1
      do while (lo <= n)
        mid = min(lo + width, n + 1); hi = min(lo + 2*width, n + 1)
        a = lo; b = mid; k = lo
        do while (a < mid .and. b < hi)
          if (up(order(b)) < up(order(a)) .or. (up(order(b)) == up(order(a)) .and. dn(order(b)) < dn(order(a)))) then
            tmp(k) = order(b); b = b + 1
          else
            tmp(k) = order(a); a = a + 1
          endif
          k = k + 1
        enddo
        do while (a < mid); tmp(k) = order(a); a = a + 1; k = k + 1; enddo
        do while (b < hi);  tmp(k) = order(b); b = b + 1; k = k + 1; enddo
        lo = lo + 2*width
      enddo
      order = tmp
      width = 2*width
    enddo
  end subroutine

  !> is (u, d) in the list sorted through order?
  logical function in_sorted(n, up, dn, order, u, d)
    integer(c_int64_t), intent(in) :: n, up(n), dn(n), order(n), u, d
    integer(c_int64_t) :: lo, hi, mid, m
    lo = 1; hi = n; in_sorted = .false.
    do while (lo <= hi)
      mid = (lo + hi) / 2; m = order(mid)
      if (up(m) == u .and. dn(m) == d) then
        in_sorted = .true.; return
      elseif (up(m) < u .or. (up(m) == u .and. dn(m) < d)) then
        lo = mid + 1
      else
        hi = mid - 1
      endif
    enddo
  end function
end module

program example_hci
  use iso_c_binding
  use sqmc_gpu_mod
  use hci_host_tools
  implicit none
  character(len=512) :: deck
  integer :: u, it, ns, nsched, i
  integer(c_int64_t) :: hdr(18), n, n_old, n_new, n_conn, nnz, q
  real(c_double) :: max_double, eps, tol, eps_pt, delta_e
  character(len=64) :: arg2
  real(c_double), allocatable :: sched(:)
  integer(c_int32_t), allocatable, target :: prod(:), osym(:), c2(:), hb_r(:), hb_s(:), pq_count(:)
  integer(c_int64_t), allocatable :: pq_ind(:)
  real(c_double), allocatable, target :: ints(:), hb_a(:)
  integer(c_int64_t), allocatable :: up(:), dn(:), tu(:), td(:), order(:), us(:), ds(:)
  real(c_double), allocatable :: wts(:,:), coeffs(:), energy(:), old_energy(:), diag(:)
  real(c_double), allocatable, target :: start(:,:)
  real(c_double), allocatable :: w(:,:), low(:)
  integer(c_int32_t) :: n_mv
  integer(c_int64_t), pointer :: p_up(:), p_dn(:)
  type(c_ptr) :: gpu, plan, c_up, c_dn, c_num, c_den
  type(sqmc_chem_cfg) :: cfg

  if (command_argument_count() < 1) stop 'usage: example_hci <deck> [eps_pt]'
  call get_command_argument(1, deck)
  eps_pt = 0
  if (command_argument_count() >= 2) then
    call get_command_argument(2, arg2); read(arg2, *) eps_pt
  endif
  open(newunit=u, file=trim(deck), access='stream', form='unformatted', status='old')
  read(u) hdr
  if (hdr(1) /= int(z'68636930', c_int64_t)) stop 'not an hci deck'
  nsched = int(hdr(15)); ns = int(hdr(16))
  allocate(sched(nsched), prod(hdr(10)), osym(hdr(11)), c2(hdr(12)), ints(hdr(9) + 1))
  allocate(hb_r(hdr(13)), hb_s(hdr(13)), hb_a(hdr(13)), pq_ind(hdr(14)), pq_count(hdr(14)))
  read(u) max_double, sched, prod, osym, c2, ints, hb_r, hb_s, hb_a, pq_ind, pq_count
  close(u)
  cfg%norb = int(hdr(2), c_int32_t); cfg%nup = int(hdr(3), c_int32_t); cfg%ndn = int(hdr(4), c_int32_t)
  cfg%n_core_orb = int(hdr(5), c_int32_t); cfg%time_sym = int(hdr(6), c_int32_t); cfg%z = int(hdr(7), c_int32_t)
  cfg%n_group = int(hdr(8), c_int32_t)
  cfg%product_table = c_loc(prod); cfg%orbital_symmetries = c_loc(osym); cfg%combine_2 = c_loc(c2)
  cfg%n_integrals = hdr(9); cfg%integrals = c_loc(ints)
  cfg%rng_mode = SQMC_RNG_COUNTER; cfg%irand_seed = [1346, 5634, 6635, 4361]; cfg%mwalk = 0
  call sqmc_gpu_check(sqmc_gpu_set_device(0_c_int), 'set_device')
  call sqmc_gpu_check(sqmc_gpu_init_chem(cfg, gpu), 'init_chem')
  call sqmc_gpu_check(sqmc_gpu_set_hb_tables(gpu, hdr(13), hb_r, hb_s, hb_a, int(hdr(14) - 1, c_int32_t), pq_ind, pq_count, max_double), 'set_hb_tables')

  ! the list starts as the HF determinant (hci.f90:300-323)
  n = 1
  allocate(up(1), dn(1), wts(1, ns), energy(ns), old_energy(ns))
  up(1) = hdr(17); dn(1) = hdr(18); wts = 0; wts(1, 1) = 1
  energy = 0
  call sqmc_gpu_check(sqmc_gpu_hamiltonian_batch(gpu, 1_c_int64_t, up, dn, up, dn, energy), 'hamiltonian')
  old_energy = energy
  write(6, '(''Iteration   0 eps1='',es7.1e1,'' ndets='',i9,'' energy='',10f16.9)') sched(1), n, energy
  tol = 1.d-10
  do it = 1, 50
    eps = sched(min(it, nsched))
    allocate(coeffs(n))
    if (it > 1) then
      coeffs = maxval(abs(wts), dim=2)
    else
      coeffs = wts(:, 1)
    endif
    ! connections above eps/|c|, sorted by (up,dn), unique, old determinants included
    call sqmc_gpu_check(sqmc_gpu_hci_connections(gpu, n, up, dn, coeffs, eps, 0_c_int, n_conn, c_up, c_dn, c_num, c_den), 'hci_connections')
    deallocate(coeffs)
    call c_f_pointer(c_up, p_up, [n_conn]); call c_f_pointer(c_dn, p_dn, [n_conn])
    ! append the new ones behind the old list, in sorted order (hci.f90:979-991)
    allocate(order(n)); call argsort_dets(n, up, dn, order)
    n_old = n; n_new = n
    allocate(tu(n_conn), td(n_conn))
    do q = 1, n_conn
      if (.not. in_sorted(n_old, up, dn, order, p_up(q), p_dn(q))) then
        n_new = n_new + 1; tu(n_new - n_old) = p_up(q); td(n_new - n_old) = p_dn(q)
      endif
    enddo
    call sqmc_gpu_free(c_up); call sqmc_gpu_free(c_dn); call sqmc_gpu_free(c_num); call sqmc_gpu_free(c_den)
    deallocate(order)
    if (n_new == n_old) then
      deallocate(tu, td); cycle
    endif
    if (n_new <= int(1.00001d0 * n_old, c_int64_t) .and. it >= nsched) then
      deallocate(tu, td); exit
    endif
    allocate(us(n_new), ds(n_new))
    us(1:n_old) = up; ds(1:n_old) = dn; us(n_old+1:n_new) = tu(1:n_new-n_old); ds(n_old+1:n_new) = td(1:n_new-n_old)
    deallocate(up, dn, tu, td); call move_alloc(us, up); call move_alloc(ds, dn)
    n = n_new
    ! the builder wants sorted labels: permute in, permute the eigenvectors back out
    allocate(order(n), us(n), ds(n), diag(n), start(n, ns))
    call argsort_dets(n, up, dn, order)
    us = up(order); ds = dn(order)
    call sqmc_gpu_check(sqmc_gpu_build_spmv_plan(gpu, n, us, ds, plan, diag, nnz), 'build_spmv_plan')
    start = 0
    if (it == 1) then
      do i = 1, min(ns, int(n)); start(i, i) = 1; enddo           ! list order = sorted order in iteration 1
    else
      do q = 1, n
        if (order(q) <= n_old) start(q, :) = wts(order(q), :)
      enddo
    endif
    ! ---- davidson_sparse (more_tools.f90:2018-2244) in one call: basis, products and corrections stay on the device; start = the last
    !      iteration's vectors on the old determinants (initial_vector), unit vectors on the first rows in iteration 1
    allocate(w(n, ns), low(ns))
    call sqmc_gpu_check(sqmc_gpu_davidson(plan, diag, int(ns, c_int32_t), c_loc(start), tol, low, w, n_mv), 'davidson')
    call sqmc_gpu_check(sqmc_gpu_spmv_free(plan), 'spmv_free')
    deallocate(wts); allocate(wts(n, ns))
    do q = 1, n
      wts(order(q), :) = w(q, :)
    enddo
    energy = low
    deallocate(w, low, order, us, ds, diag, start)
    write(6, '(''Iteration'',i4,'' eps1='',es7.1e1,'' ndets='',i9,'' nnz='',i11,'' energy='',10f16.9)') it, eps, n, nnz, energy
    if (maxval(abs(energy - old_energy)) < 1.d-5 .and. it >= nsched) exit
    old_energy = energy
  enddo
  write(6, '(a,i10,10es26.17)') 'fortran hci:', n, energy
  if (eps_pt > 0) then
    allocate(coeffs(n)); coeffs = wts(:, 1)
    call sqmc_gpu_check(sqmc_gpu_hci_pt2(gpu, n, up, dn, coeffs, energy(1), eps_pt, 1_c_int32_t, delta_e, n_conn), 'hci_pt2')
    write(6, '(a,i12,2es26.17)') 'fortran pt2:', n_conn, delta_e, energy(1) + delta_e
    deallocate(coeffs)
  endif
  call sqmc_gpu_check(sqmc_gpu_finalize(gpu), 'finalize')

end program
