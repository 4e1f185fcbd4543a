// psit_kernels.h -- the step variant hf_to_psit = .true. ("replace the HF state with the trial wave function",
// do_walk.f90:35, 378-386).  Textually included by sqmc_gpu.hip behind walk_kernels.h.
//
// The first basis state is Psi_T instead of its first determinant: moves to and from it are deterministic, everything else stays
// stochastic, and all n_ct determinants of C(T) stay in the walker list whatever their weights.  The reference keeps three segments
// -- C(T), the survivors outside C(T), this step's spawns (do_walk.f90:1267-1300, 5268-5307, 6484-6833).  Here the first two are ONE
// list sorted by  key' = rank + (outside C(T) ? koff : 0):  the C(T) determinants never leave slots [0, n_ct), so every table of the
// variant (Psi_T locations, deterministic-space locations, e_loc_num/den, diag_elems) is addressed by slot, and sort + annihilation
// are the unchanged kernels (k_anneal<.,1>).  What is new:
//
//   k_gate / k_spawn        slot 0 neither draws nor spawns (3574); a child onto the first determinant has weight 0 (3676, 7642)
//   k_psit_ct_terms         first row, first column and extra diagonal of the transformed projector over C(T)
//                           (2285; fast_sparse_matrix_multiply_upper_triangular's degenerate form, more_tools.f90:3656-3662)
//   k_psit_rows_fin         ... the first row's sum; the Psi_T locations with (1 + tau E_T) c (2286; more_tools.f90:3663-3667)
//   k_psit_apply            w_i <- w_i - tau dw_i + tau E_T w_i on C(T), + the Psi_T row, + the deterministic-space product (2313-2323)
//   k_anneal<ITEMS, 1>      merge_my_original_with_spawned3 + reduce_my_walker for everything outside C(T)
//   (k_psit_finish)         T^-1 and its transpose on the Psi_T locations (2394-2442), in front of the finish of every slot
//   k_psit_finish           check_initiator over C(T) without discarding (2444-2462), reweighting (2487), the sums over C(T) and the
//                           energy estimator, a plain loop over the C(T) slots (2701-2722)
//
// Long sums.  Three sums of the step run over thousands of terms left to right in the reference (first row over C(T): n_ct terms;
// first row over Psi_T and T^-1: n_psit - 1).  seq = 1 keeps that order (one lane; the C(T) row then costs ~0.25 ms).  The default
// adds the same terms through a fixed 64-ary tree -- chunks of 64 consecutive terms, each added left to right, level by level --
// a fixed order any implementation can reproduce bit for bit (the tests' CPU restatement does), rounding-level different from the reference.
//
// All HBM/latency bound; n_ct = 7.7e4 for C2 cc-pVDZ with a 100-determinant Psi_T: every kernel here is a few microseconds.

// (struct PsitArgs: sqmc_gpu.hip, in front of the context that keeps one)
#define PSIT_MAXTERMS (64ll * 64 * 64)      // three tree levels
#define PSIT_FB 120           // blocks of k_psit_finish (grid-stride)

// Two tree levels over the 4096 terms [base, base + 4096) of n by ONE wavefront: the 64 terms of chunk c are added left to right
// (level 1), then the chunk sums left to right (level 2).  The terms are evaluated by all lanes at once -- 16 rows of 64 in flight
// per round trip -- and parked in LDS (s_t: 64 x 65 doubles, one padded row per chunk) where lane c adds its row; every lane returns
// the level-2 sum.  term(i) must be callable by any lane for 0 <= i < n.
template <class F>
__device__ __forceinline__ double wave_sum_4096(F term, long long base, long long n, double (*s_t)[65]) {
  const int lane = threadIdx.x & 63;
  const long long left = n - base;                       // > 0
  const int nterm = (int)(left < 4096 ? left : 4096), nrow = (nterm + 63) / 64;
  for (int r0 = 0; r0 < nrow; r0 += 16) {
    double t[16];
#pragma unroll
    for (int q = 0; q < 16; q++) { const int i = 64 * (r0 + q) + lane; t[q] = (r0 + q < nrow && i < nterm) ? term(base + i) : 0.0; }
#pragma unroll
    for (int q = 0; q < 16; q++) if (r0 + q < nrow) s_t[r0 + q][lane] = t[q];
  }
  __builtin_amdgcn_wave_barrier();
  double s = 0.0;
  if (lane < nrow) {
    const int cnt = nterm - 64 * lane < 64 ? nterm - 64 * lane : 64;
    s = s_t[lane][0];
    for (int k = 1; k < cnt; k++) s = s + s_t[lane][k];
  }
  __builtin_amdgcn_wave_barrier();                       // the rows are free again for the caller's next block
  // level 2: every lane the nrow chunk sums in order (handed over through the shuffle network)
  double tot = __shfl(s, 0, 64);
  for (int k = 1; k < nrow; k++) { const double v = __shfl(s, k, 64); tot = tot + v; }
  return tot;
}
// Sum of n terms by ONE wavefront through the fixed 64-ary tree (or left to right by lane 0 if seq); every lane returns it.
// s_t: 64 x 65 doubles of LDS owned by the wavefront; n <= 64^3 (three levels; checked by the host).
template <class F>
__device__ __forceinline__ double wave_tree_sum(F term, long long n, double (*s_t)[65], int seq) {
  const int lane = threadIdx.x & 63;
  if (n <= 0) return 0.0;
  if (seq) {
    double tot = 0.0;
    if (lane == 0) { tot = term(0); for (long long i = 1; i < n; i++) tot = tot + term(i); }
    return __shfl(tot, 0, 64);
  }
  const long long nblk4096 = (n + 4095) / 4096;
  double tot = 0.0;                                       // level 3: the block sums in order
  for (long long b = 0; b < nblk4096; b++) {
    const double v = wave_sum_4096(term, b * 4096, n, s_t);
    tot = b == 0 ? v : tot + v;
  }
  return tot;
}

// term i of the first row over C(T), more_tools.f90:3657-3660: E_num(1)/E_den(1) w_1, then E_num(i) w_i
__device__ __forceinline__ double psit_ct_term(const PsitArgs &a, const double *__restrict__ wt, long long i) {
  return i == 0 ? a.cnum[0] / a.cden[0] * wt[0] : a.cnum[i] * wt[i];
}
// deltaw of the C(T) slots (first column + extra diagonal of the transformed projector), elementwise
__global__ void __launch_bounds__(TPB) k_psit_ct_col(PsitArgs a, const double *__restrict__ wt) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= a.n_ct || i == 0) return;
  double d = 0.0 + a.cnum[i] * wt[0]; d = d + a.diag[i] * wt[i];
  a.dw_ct[i] = d;
}
// One wavefront per 4096 slots of C(T): two tree levels of the first row's sum.
__global__ void __launch_bounds__(64) k_psit_ct_terms(PsitArgs a, const double *__restrict__ wt) {
  __shared__ double s_t[64][65];
  if (a.seq) return;                                   // the row's sum is made left to right by k_psit_rows_fin
  const double tot = wave_sum_4096([&](long long i) { return psit_ct_term(a, wt, i); }, (long long)blockIdx.x * 4096, a.n_ct, s_t);
  if (threadIdx.x == 0) a.p2[blockIdx.x] = tot;
}
// One block: wave 0 finishes the first row over C(T) (level 3 over the blocks' sums); wave 1 makes the first row over the Psi_T
// locations; all threads its first column.
__global__ void __launch_bounds__(TPB) k_psit_rows_fin(PsitArgs a, const double *__restrict__ wt, double tau, double e_trial) {
  __shared__ double s_scr[2][64][65];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double one_plus = 1.0 + tau * e_trial;
  if (wv == 0) {
    double tot = 0.0;
    if (a.seq) tot = wave_tree_sum([&](long long i) { return psit_ct_term(a, wt, i); }, a.n_ct, s_scr[0], 1);
    else if (lane == 0) { const long long nb = (a.n_ct + 4095) / 4096; tot = a.p2[0]; for (long long b = 1; b < nb; b++) tot = tot + a.p2[b]; }
    if (lane == 0) a.dw_ct[0] = 0.0 + tot;
  } else if (wv == 1) {
    double tot = 0.0;
    if (a.n_psit > 1) tot = 0.0 + wave_tree_sum([&](long long i) { return (one_plus * a.cdet[i + 1]) * wt[a.loc_psit[i + 1]]; }, a.n_psit - 1, s_scr[1], a.seq);
    if (lane == 0) a.dw_ps[0] = tot;
  }
  const double v1 = wt[a.loc_psit[0]];
  for (long long k = 1 + threadIdx.x; k < a.n_psit; k += TPB) a.dw_ps[k] = 0.0 + (one_plus * a.cdet[k]) * v1;
}
// do_walk.f90:2313-2323, slot by slot in the reference's order: the C(T) line, then the Psi_T row, then the deterministic-space product
__global__ void __launch_bounds__(TPB) k_psit_apply(PsitArgs a, double *__restrict__ wt, double tau, double e_trial) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= a.n_ct) return;
  double w = wt[i];
  w = w - tau * a.dw_ct[i] + tau * e_trial * w;
  const int k = a.psit_of[i], r = a.imp_of[i];
  if (k >= 0) w = w + a.dw_ps[k];
  if (r >= 0) w = w + a.dw_imp[r];
  wt[i] = w;
}
// y = A x of the deterministic space, rows added in the reference's order (prj_row_product); no E_T term in this variant (2262 without 2290)
__global__ void __launch_bounds__(TPB) k_psit_imp_rows(PrjPre pp) {
  const int row = (int)blockIdx.x * (TPB / 64) + (int)(threadIdx.x >> 6);
  if (row < pp.n_imp) prj_row_product(pp, row);
}

// T^-1 and its transpose on the Psi_T locations after the merge, do_walk.f90:2394-2442 (one block)
// check_initiator over C(T) (nothing is discarded), the reweighting, and everything the step sums over the C(T) slots
// (pipelined steps, go.on: also the next step's gate of the C(T) slots -- k_anneal<., 1> wrote the keys and the gate of everything outside)
__global__ void __launch_bounds__(TPB) k_psit_finish(PsitArgs a, double *__restrict__ wt, u32 *__restrict__ flg, StepP p, double *__restrict__ partials2, GateOut go, u64 seed) {
  // T^-1 and its transpose on the Psi_T locations (2394-2442): w_1 = ((w_1 - sum_{k>1} c_k w_k) / c_1) / c_1, then w_k -= c_k w_1.  Every
  // block forms the sum itself, in the one order, from the copy k_anneal<., 1> left of those weights (go.ps_raw): the slots themselves
  // are being overwritten by other blocks.  One kernel less on the step's critical path.
  __shared__ double s_scr[64][65];
  __shared__ double s_w1;
  if (threadIdx.x < 64) {
    double tmp = 0.0;
    if (a.n_psit > 1) tmp = 0.0 + wave_tree_sum([&](long long i) { return a.cdet[i + 1] * go.ps_raw[i + 1]; }, a.n_psit - 1, s_scr, a.seq);
    if (threadIdx.x == 0) { double w1 = go.ps_raw[0]; w1 = w1 - tmp; w1 = w1 / a.cdet[0]; w1 = w1 / a.cdet[0]; s_w1 = w1; }
  }
  __syncthreads();
  const double w1 = s_w1;
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
  const double r0 = a.cnum[0] / a.cden[0];
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < a.n_ct; i += (long long)gridDim.x * TPB) {
    double w = wt[i]; const u32 f = flg[i];
    { const int kp = a.psit_of[i]; if (kp == 0) w = w1; else if (kp > 0) w = w - a.cdet[kp] * w1; }
    int d = flg_impd(f), ini = flg_init(f); const int ps = flg_psign(f);
    if (!p.cti || i < a.n_perm) {                       // 2447-2461: with c_t_initiator only the first n_permanent_initiator slots are visited
      const int dd = d - p.imind > 0 ? d - p.imind : 0;
      const double thr = p.r_init * ipow_d(dd, p.ipow), aw = fabs(w);
      if (ini == 3 && p.r_init >= 0) { if (w * ps < 1.0) w = (double)ps; }
      else if (ini == 2 && ((aw <= thr && d > 0) || ((aw <= p.r_init && !p.cti) && d == -2))) ini = 1;
      else if (ini < 2 && ((aw > thr && d >= 0) || ((aw > p.r_init || p.cti) && d == -2))) ini = ini + 1;
    }
    w = w * p.rfi;                                      // 2487
    wt[i] = w; flg[i] = pack_flg(d, ini, ps);
    if (go.on) {
      u64 nc; double wc;
      gate_children(w, go.cutoff, seed, go.step_next, go.keys[i] >> 32, nc, wc);
      if (i == 0) { nc = 0; wc = 0.0; }                 // all moves of the first state are deterministic (do_walk.f90:3574)
      go.nchild[i] = nc; go.wchild[i] = wc;
    }
    s[0] += w; s[1] += fabs(w); s[8] += w * w;          // 2590-2598
    if (ini == 3) s[4] += w * ps;
    if (d == 0 || (d == -2 && p.cti)) s[6] += fabs(w);
    double e_num, e_den;                                // 2701-2722
    if (i == 0) { e_num = r0 * w; e_den = w; } else { e_num = a.cnum[i] * w; e_den = a.cden[i] * w; }
    if (fabs(e_den) < 1e-22) e_den = fabs(e_den);
    s[2] += e_den; s[3] += e_num; s[9] += e_num * e_num; s[10] += e_den * e_den;
    s[11] += e_num * copysign(1.0, e_den); s[12] += fabs(e_den); s[5] += e_num * e_den;
  }
  __shared__ double red[TPB / 64][NSTAT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NSTAT; k++) {
    double v = s[k];
    for (int q = 32; q > 0; q >>= 1) v += __shfl_down(v, q, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSTAT) {
    double v = 0.0;
    for (int q = 0; q < TPB / 64; q++) v += red[q][threadIdx.x];
    partials2[(long long)blockIdx.x * NSTAT + threadIdx.x] = v;
  }
}
