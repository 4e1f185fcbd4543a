/*
 * sqmc_oracle_psit.c -- CPU ORACLE (test infrastructure, NOT product code): the step variant
 * hf_to_psit = .true. ("replace the HF state with the trial wave function", do_walk.f90:35, 378-386).
 * Textually included at the end of sqmc_oracle.c (it uses that file's static helpers).
 *
 * The first basis state is Psi_T instead of its first determinant; every move to and from that
 * state is deterministic, everything else stays stochastic.  The walker list keeps its layout
 * of the reference (do_walk.f90:1267-1300, 5268-5307, 6484-6833):
 *   [0, n_ct)                the ndet_psi_t_connected determinants of C(T), in (up,dn) order, never
 *                            discarded, weights may be zero; slot 0 is the first state
 *   [n_ct, n_ct + n_out)     the survivors outside C(T) (my_ndet_outside_ct), in (up,dn) order
 *   [n_ct + n_out, nwalk)    this step's spawns (only these are sorted)
 *
 * PARITY UNPINNED BY THE REFERENCE: it holds no fixture of any walk, do_walk.f90 does not compile
 * with this image's flang, and its own comments mark parts of this path as untested.  Two places
 * of merge_my_original_with_spawned3 cannot be meant as written (tests/golden/README_hf_to_psit.md):
 *   Q1  do_walk.f90:6709-6813, the bookkeeping that takes the discarded determinants out of the outside segment and splices the
 *       new ones in, loses determinants in three ways.  (i) 6720 `last = my_ndet_outside_ct` for the block behind the LAST
 *       discarded determinant: first = leave_out_list(i)+1 is a position in the walker list (> n_ct), my_ndet_outside_ct a
 *       count; with n_ct > 0 the block is not moved down, the count is still reduced by leave_out, and the list loses its
 *       last leave_out determinants while keeping the discarded one (the make-room loop at 6779 has the position,
 *       my_ndet_psi_t_connected+my_ndet_outside_ct).  (ii) The same block test (6736-6737) leaves the insertion point of a
 *       new determinant that is to be APPENDED (6691: one position behind the segment) where it was although the
 *       segment's end has moved.  (iii) A new determinant whose insertion point is a determinant that is itself discarded
 *       later in the same call falls between two blocks (6734: `< first` of the next, `> last` of the one before) and is
 *       not moved at all: it lands behind determinants with larger labels.  (A GPU run against the literal text showed
 *       (iii) at step 41 of the C2 walk: tools/psit_debug.py.)
 *   Q2  no `imp_distance -1 -> 1` (merge_original_with_spawned2: 5985-5986, 6032-6036): a child
 *       of a deterministic-space parent that lands outside C(T) would stay at -1 for ever (never
 *       rounded, never discarded, no death/clone) and its own children would carry 0.
 * `quirks` = 0 (default) restates what the code means -- the outside segment after the merge is the
 * ordered union of its surviving determinants and the new ones that pass the initiator test (Q1),
 * the conversion at Q2; bit 0 / bit 1 restore the literal text of Q1 / Q2 so that tests can show
 * what it does.
 *
 * `sum_order`: the three long sums of the step (first row of the C(T) product 2285, first row of
 * the Psi_T product 2286, T^-1 2407-2411) run left to right in the reference.  0 = that order;
 * 1 = the same terms through a fixed 64-ary tree (chunks of 64 consecutive terms, each added left
 * to right, level by level): the order the GPU kernels use by default, bit for bit.  The two
 * differ by rounding only (tests compare them).
 */

typedef struct {
  int64_t n_ct, n_out, n_psit, n_imp;
  int64_t *loc_psit;          /* my_locations_of_psit: position (0-based) of dets_psi_t(k) in the walker list, do_walk.f90:1849-1886 */
  double  *cdet;              /* cdet_psi_t in the label order of dets_psi_t (do_walk.f90:1258) */
  double  *diag_elems;        /* do_walk.f90:1091-1116: H_ii of the C(T) determinants outside the deterministic space, 0 inside */
  int64_t *loc_imp;           /* my_locations_of_imp_dets: fixed in this mode (do_walk.f90:2188 skips the rescan) */
  int quirks, sum_order;
  /* tests: the list as it stands in front of the merge (residents after death/clone and the projection, then the sorted spawns) */
  int dbg_on; int64_t dbg_n, dbg_n0; det_t *dbg_up, *dbg_dn; double *dbg_wt; int8_t *dbg_d, *dbg_i;
} orc_psit;

orc_psit *orc_psit_new(int64_t n_ct, int64_t n_psit, const int64_t *loc_psit, const double *cdet, const double *diag_elems,
                       int64_t n_imp, const int64_t *loc_imp) {
  orc_psit *q = calloc(1, sizeof(orc_psit));
  q->n_ct = n_ct; q->n_out = 0; q->n_psit = n_psit; q->n_imp = n_imp;       /* my_ndet_outside_ct = 0, do_walk.f90:1637 */
  q->loc_psit = malloc((n_psit + 1) * sizeof(int64_t)); q->cdet = malloc((n_psit + 1) * sizeof(double));
  q->diag_elems = malloc((n_ct + 1) * sizeof(double)); q->loc_imp = malloc((n_imp + 1) * sizeof(int64_t));
  memcpy(q->loc_psit, loc_psit, n_psit * sizeof(int64_t)); memcpy(q->cdet, cdet, n_psit * sizeof(double));
  memcpy(q->diag_elems, diag_elems, n_ct * sizeof(double)); memcpy(q->loc_imp, loc_imp, n_imp * sizeof(int64_t));
  return q;
}
void orc_psit_free(orc_psit *q) { if (!q) return; free(q->loc_psit); free(q->cdet); free(q->diag_elems); free(q->loc_imp);
                                  free(q->dbg_up); free(q->dbg_dn); free(q->dbg_wt); free(q->dbg_d); free(q->dbg_i); free(q); }
void orc_psit_debug(orc_psit *q, int on) { q->dbg_on = on; }
int64_t orc_psit_debug_get(const orc_psit *q, int64_t cap, int64_t *n0, det_t *up, det_t *dn, double *wt, int8_t *d, int8_t *ini) {
  if (cap >= q->dbg_n && q->dbg_n > 0) { memcpy(up, q->dbg_up, q->dbg_n * 8); memcpy(dn, q->dbg_dn, q->dbg_n * 8); memcpy(wt, q->dbg_wt, q->dbg_n * 8);
                                         memcpy(d, q->dbg_d, q->dbg_n); memcpy(ini, q->dbg_i, q->dbg_n); }
  *n0 = q->dbg_n0; return q->dbg_n;
}
void orc_psit_set(orc_psit *q, int quirks, int sum_order) { q->quirks = quirks; q->sum_order = sum_order; }
int64_t orc_psit_n_out(const orc_psit *q) { return q->n_out; }

/* sum of n terms: left to right (mode 0), or through the fixed 64-ary tree (mode 1).  x is overwritten. */
double orc_ordered_sum(double *x, int64_t n, int mode) {
  if (n <= 0) return 0.0;
  if (mode == 0) { double s = x[0]; for (int64_t i = 1; i < n; i++) s = s + x[i]; return s; }
  while (n > 1) {
    int64_t m = (n + 63) / 64;
    for (int64_t k = 0; k < m; k++) {
      int64_t e = 64 * k + 64 < n ? 64 * k + 64 : n;
      double s = x[64 * k];
      for (int64_t i = 64 * k + 1; i < e; i++) s = s + x[i];
      x[k] = s;
    }
    n = m;
  }
  return x[0];
}

static void psit_copy_slot(orc_walk *w, int64_t to, int64_t from) {
  w->up[to] = w->up[from]; w->dn[to] = w->dn[from]; w->wt[to] = w->wt[from]; w->initiator[to] = w->initiator[from];
  w->e_num_walker[to] = w->e_num_walker[from]; w->e_den_walker[to] = w->e_den_walker[from];
  w->imp_distance[to] = w->imp_distance[from]; w->matrix_elements[to] = w->matrix_elements[from];
}
/* walk_xxx(first+shift : last+shift) = walk_xxx(first : last), array-section semantics (the right side is read first) */
static void psit_shift_block(orc_walk *w, int64_t first, int64_t last, int64_t shift) {
  if (last < first || shift == 0) return;
  if (shift < 0) for (int64_t i = first; i <= last; i++) psit_copy_slot(w, i + shift, i);
  else for (int64_t i = last; i >= first; i--) psit_copy_slot(w, i + shift, i);
}

/* sort_my_walkers3_up_dn, do_walk.f90:5268-5307: only the newly spawned walkers [lo, n) are sorted (merge_sort2_up_dn, stable) */
static void psit_sort_spawns(orc_walk *w, int64_t lo, int64_t n) {
  const int64_t m = n - lo;
  if (m <= 1) return;
  int64_t *ord = malloc(m * sizeof(int64_t)), *tmp = malloc((m + 2) * sizeof(int64_t));
  for (int64_t i = 0; i < m; i++) ord[i] = lo + i;
  msort_idx(w->up, w->dn, ord, tmp, m);
#define PERM(T, A) do { T *b = malloc(m * sizeof(T)); for (int64_t i = 0; i < m; i++) b[i] = (A)[ord[i]]; memcpy((A) + lo, b, m * sizeof(T)); free(b); } while (0)
  PERM(det_t, w->up); PERM(det_t, w->dn); PERM(double, w->wt); PERM(int8_t, w->imp_distance);
  PERM(int8_t, w->initiator); PERM(double, w->matrix_elements); PERM(double, w->e_num_walker); PERM(double, w->e_den_walker);
#undef PERM
  free(ord); free(tmp);
}

/* the pairwise rule shared by the three places of merge_my_original_with_spawned3 that fold walker `src` into walker `dst`
 * (6544-6559 onto C(T), 6590-6607 onto an outside survivor): initiator of dst */
static void psit_fold_initiator(orc_walk *w, int64_t dst, int64_t src, const orc_step_params *p) {
  if (w->wt[src] * w->wt[dst] > 0) { if (w->initiator[src] > w->initiator[dst]) w->initiator[dst] = w->initiator[src]; }
  else if (fabs(w->wt[dst]) < fabs(w->wt[src])) { if (w->initiator[dst] != 3 || p->r_initiator == -1.0) w->initiator[dst] = w->initiator[src]; }
  else if (fabs(w->wt[dst]) == fabs(w->wt[src])) { if (w->initiator[dst] != 3 || p->r_initiator == -1.0) w->initiator[dst] = 0; }
}

/* merge_my_original_with_spawned3, do_walk.f90:6484-6833.  0-based.  Returns 0, or 1 ('Need to set MWALK higher!', 6753-6757). */
static int psit_merge3(orc_walk *w, orc_psit *q, int64_t *my_nwalk, const orc_step_params *p) {
  const int64_t n_ct = q->n_ct; int64_t n_out = q->n_out; const int64_t nw = *my_nwalk, M = w->mwalk;
  int i_perm = 0;
  int64_t leave_out = 0, num_to_move = 0; int merge_with_prev = 0;
  int64_t *leave_out_list = malloc((nw + 2) * sizeof(int64_t)), *empty_outside_ct = malloc((nw + 2) * sizeof(int64_t)), *indices_new_dets = malloc((nw + 2) * sizeof(int64_t));
  if (nw == n_ct + n_out) {                                             /* 6515-6532 */
    for (int64_t io = n_ct; io < n_ct + n_out; io++)
      if (check_initiator(w, io, p, &i_perm)) leave_out_list[leave_out++] = io;
  } else {
    int64_t start_ct = 0, start_out = n_ct, io = n_ct;
    for (int64_t iw = n_ct + n_out; iw < nw; iw++) {                     /* 6538: the sorted spawns */
      int found = 0, reorder = 0;
      for (int64_t ic = start_ct; ic < n_ct; ic++) {                     /* 6542-6585 */
        if (w->up[iw] == w->up[ic] && w->dn[iw] == w->dn[ic]) {
          psit_fold_initiator(w, ic, iw, p);
          if (!(w->imp_distance[ic] == 0 && w->imp_distance[iw] == -1)) w->wt[ic] = w->wt[ic] + w->wt[iw];      /* 6563 */
          found = 1; start_ct = ic; break;
        } else if (w->up[iw] > w->up[ic] || (w->up[iw] == w->up[ic] && w->dn[iw] > w->dn[ic])) {
          start_ct = ic + 1;
        } else { start_ct = ic; break; }
      }
      if (found) continue;
      for (io = start_out; io < n_ct + n_out; io++) {                    /* 6590-6638 */
        if (w->up[iw] == w->up[io] && w->dn[iw] == w->dn[io]) {
          const int a = abs(w->imp_distance[iw]);
          psit_fold_initiator(w, io, iw, p);
          if (a < w->imp_distance[io]) w->imp_distance[io] = (int8_t)a;   /* 6596, 6598: min(own, |child's|) in both branches */
          w->wt[io] = w->wt[io] + w->wt[iw];
          found = 1; start_out = io; break;
        } else if (w->up[io] < w->up[iw] || (w->up[io] == w->up[iw] && w->dn[io] < w->dn[iw])) {
          if (check_initiator(w, io, p, &i_perm)) leave_out_list[leave_out++] = io;
          start_out = io + 1;
        } else { start_out = io; reorder = 1; break; }
      }
      if (!found) {
        if (merge_with_prev) {                                           /* 6642-6676: iwalk-1 folded into iwalk */
          const int64_t pv = iw - 1;
          const int same = (w->wt[pv] * w->wt[iw] > 0);
          /* the initiator of iwalk: max if same sign; the previous one's if |w(iwalk)| < |w(iwalk-1)|; 0 if equal */
          if (same) { if (w->initiator[pv] > w->initiator[iw]) w->initiator[iw] = w->initiator[pv]; }
          w->e_num_walker[iw] = DMIN(w->e_num_walker[iw], w->e_num_walker[pv]);
          w->e_den_walker[iw] = DMIN(w->e_den_walker[iw], w->e_den_walker[pv]);
          { const int a = abs(w->imp_distance[pv]);
            if (q->quirks & 2) { if (a < w->imp_distance[iw]) w->imp_distance[iw] = (int8_t)a; }      /* literal: min(imp_distance(iwalk), abs(imp_distance(iwalk-1))) */
            else { const int b = abs(w->imp_distance[iw]); w->imp_distance[iw] = (int8_t)(a < b ? a : b); } }
          w->matrix_elements[iw] = DMIN(w->matrix_elements[iw], w->matrix_elements[pv]);
          if (!same) {
            if (fabs(w->wt[iw]) < fabs(w->wt[pv])) { if (w->initiator[iw] != 3 || p->r_initiator == -1.0) w->initiator[iw] = w->initiator[pv]; }
            else if (fabs(w->wt[iw]) == fabs(w->wt[pv])) { if (w->initiator[iw] != 3 || p->r_initiator == -1.0) w->initiator[iw] = 0; }
          }
          w->wt[iw] = w->wt[iw] + w->wt[pv];
          merge_with_prev = 0;
        }
        if (iw + 1 < nw && w->up[iw] == w->up[iw + 1] && w->dn[iw] == w->dn[iw + 1]) merge_with_prev = 1;      /* 6678-6679 */
        else {
          if (!(q->quirks & 2) && w->imp_distance[iw] == -1) w->imp_distance[iw] = 1;      /* Q2: what merge_original_with_spawned2 does at 5985-5986 / 6032-6036 */
          if (!check_initiator(w, iw, p, &i_perm)) {                     /* 6682-6693 */
            empty_outside_ct[num_to_move] = reorder ? io : n_ct + n_out;
            indices_new_dets[num_to_move] = iw; num_to_move++;
          }
        }
      }
    }
    for (int64_t i2 = start_out; i2 < n_ct + n_out; i2++)                /* 6711-6717 */
      if (check_initiator(w, i2, p, &i_perm)) leave_out_list[leave_out++] = i2;
  }
  int st = 0;
  if (!(q->quirks & 1)) {
    /* What 6709-6813 is for: the outside segment becomes the ordered union of its determinants that were not discarded and the new
     * determinants that passed the initiator test (both lists are in (up,dn) order already). */
    const int64_t n_keep = n_out - leave_out, n_new_total = n_keep + num_to_move;
    if (n_ct + n_new_total > M) st = 1;
    else {
      orc_walk *t = orc_walk_new(n_new_total + 1);
      int64_t a = n_ct, b = 0, lo = 0, k = 0;
#define TAKE(FROM) do { t->up[k] = w->up[FROM]; t->dn[k] = w->dn[FROM]; t->wt[k] = w->wt[FROM]; t->initiator[k] = w->initiator[FROM]; t->imp_distance[k] = w->imp_distance[FROM]; \
                        t->matrix_elements[k] = w->matrix_elements[FROM]; t->e_num_walker[k] = w->e_num_walker[FROM]; t->e_den_walker[k] = w->e_den_walker[FROM]; k++; } while (0)
      while (a < n_ct + n_out || b < num_to_move) {
        if (a < n_ct + n_out && lo < leave_out && leave_out_list[lo] == a) { a++; lo++; continue; }
        if (b >= num_to_move) { TAKE(a); a++; }
        else if (a >= n_ct + n_out) { TAKE(indices_new_dets[b]); b++; }
        else {
          const int64_t nb = indices_new_dets[b];
          if (w->up[a] < w->up[nb] || (w->up[a] == w->up[nb] && w->dn[a] < w->dn[nb])) { TAKE(a); a++; } else { TAKE(nb); b++; }
        }
      }
#undef TAKE
      for (int64_t i = 0; i < k; i++) {
        const int64_t d = n_ct + i;
        w->up[d] = t->up[i]; w->dn[d] = t->dn[i]; w->wt[d] = t->wt[i]; w->initiator[d] = t->initiator[i]; w->imp_distance[d] = t->imp_distance[i];
        w->matrix_elements[d] = t->matrix_elements[i]; w->e_num_walker[d] = t->e_num_walker[i]; w->e_den_walker[d] = t->e_den_walker[i];
      }
      orc_walk_free(t);
      n_out = n_keep;
    }
  } else {
    /* the literal bookkeeping (quirk bit 0), do_walk.f90:6709-6813 */
    if (leave_out > 0) {                                                   /* 6723-6753 */
      int64_t start_empty = 0;
      for (int64_t i = 0; i < leave_out; i++) {
        const int64_t first = leave_out_list[i] + 1;
        int64_t last;
        if (i + 1 < leave_out) last = leave_out_list[i + 1] - 1;
        else last = n_out - 1;                                             /* `last = my_ndet_outside_ct`, a count where a position is meant */
        const int64_t shift = -(i + 1);
        psit_shift_block(w, first, last, shift);
        for (int64_t j = start_empty; j < num_to_move; j++) {
          if (empty_outside_ct[j] < first) continue;
          else if (empty_outside_ct[j] >= first && empty_outside_ct[j] <= last) empty_outside_ct[j] = empty_outside_ct[j] - (i + 1);
          else { start_empty = j; break; }
        }
      }
      n_out -= leave_out;
    }
      if (num_to_move > 0) {                                                 /* 6756-6813 */
      for (int64_t i = num_to_move - 1; i >= 0 && !st; i--) {
        const int64_t from = indices_new_dets[i], to = M - num_to_move + i;
        if (indices_new_dets[i] >= M - num_to_move) { st = 1; break; }
        psit_copy_slot(w, to, from);
      }
      if (!st) {
        for (int64_t i = num_to_move - 1; i >= 0; i--) {
          const int64_t shift = i + 1, first = empty_outside_ct[i];
          const int64_t last = (i == num_to_move - 1) ? n_ct + n_out - 1 : empty_outside_ct[i + 1] - 1;
          psit_shift_block(w, first, last, shift);
        }
        for (int64_t i = 0; i < num_to_move; i++) psit_copy_slot(w, empty_outside_ct[i] + i, M - num_to_move + i);
      }
    }
  }
  n_out += num_to_move;
  *my_nwalk = n_ct + n_out; q->n_out = n_out;
  for (int64_t i = *my_nwalk; i < M && i < nw + num_to_move + 1; i++) { w->e_num_walker[i] = 1e51; w->e_den_walker[i] = 1e51; w->matrix_elements[i] = 1e51; }
  for (int64_t i = M - num_to_move; i < M; i++) if (i >= *my_nwalk) { w->e_num_walker[i] = 1e51; w->e_den_walker[i] = 1e51; w->matrix_elements[i] = 1e51; }
  free(leave_out_list); free(empty_outside_ct); free(indices_new_dets);
  return st;
}

/* reduce_my_walker with hf_to_psit, do_walk.f90:7196-7254: i_start = my_ndet_psi_t_connected+1 */
static int64_t psit_reduce(orc_walk *w, orc_psit *q, int64_t n, const orc_step_params *p) {
  for (int64_t i = q->n_ct; i < n; i++)
    if (w->imp_distance[i] >= 1 && fabs(w->wt[i]) < p->min_wt) {
      orc_rng_seek(&w->rng, 2, w->key_norb ? orc_det_rank(w->key_norb, w->key_ndn, w->up[i], w->dn[i])
                                           : (uint64_t)w->up[i] * 0x9E3779B97F4A7C15ull + (uint64_t)w->dn[i]);
      if (orc_rannyu(&w->rng) < (fabs(w->wt[i]) / p->min_wt)) w->wt[i] = copysign(p->min_wt, w->wt[i]);
      else w->wt[i] = 0.0;
    }
  int64_t nshift = 0;
  for (int64_t i = q->n_ct; i < n; i++) {
    if (w->wt[i] == 0.0 && w->imp_distance[i] >= 1) nshift++;
    else if (nshift) psit_copy_slot(w, i - nshift, i);
  }
  q->n_out -= nshift;
  return n - nshift;
}

/* One MC step with hf_to_psit = .true., do_walk.f90:2186-2790, semistochastic, ncores = 1, run_type 'none'. */
static int walk_step_psit_sys(orc_sys *s, orc_walk *w, orc_psit *q, const orc_step_params *p, double out[16]) {
  const int64_t n_ct = q->n_ct, n_imp = q->n_imp, n_psit = q->n_psit;
  const int64_t n0 = w->nwalk;
  int64_t attempts = 0;
  if (!p->semistochastic || n0 != n_ct + q->n_out || n_psit < 1 || n_ct < 1) return 5;
  s->psit = 1; s->first_up = w->up[q->loc_psit[0]]; s->first_dn = w->dn[q->loc_psit[0]];      /* dets_up/dn_psi_t(1) */
  w->n_spawn_draws = 0;
  /* ---- 2188: no rescan of my_locations_of_imp_dets.  The products below read the weights BEFORE the move: the deterministic-space
   *      and C(T) determinants have imp_distance <= 0, so move_uniform2 leaves their weights alone (3743) */
  for (int64_t i = 0; i < n0; i++) {               /* 2220-2231 */
    int st = move_uniform2(s, w, p, i, &attempts);
    if (st) return st;
  }
  /* ---- deterministic projection, 2255-2325 */
  double *dw = malloc((n_imp + n_ct + n_psit + 3) * sizeof(double)), *x = malloc((n_imp + n_ct + n_psit + 3) * sizeof(double));
  for (int64_t i = 0; i < n_imp; i++) x[i] = w->wt[q->loc_imp[i]];
  orc_spmv_sym_upper(n_imp, w->prj_counts, w->prj_indices, w->prj_values, x, dw);       /* 2262: no E_T term here (2290 is the else branch) */
  {
    /* 2285, more_tools.f90:3656-3662: first row, first column and the extra diagonal of the transformed projector over C(T) */
    double *a = dw + n_imp, *t = x;
    const double den1 = w->ct_den[0];
    t[0] = w->ct_num[0] / den1 * w->wt[0];
    for (int64_t i = 1; i < n_ct; i++) {
      a[i] = 0.0 + w->ct_num[i] * w->wt[0];
      t[i] = w->ct_num[i] * w->wt[i];
      a[i] = a[i] + q->diag_elems[i] * w->wt[i];
    }
    a[0] = 0.0 + orc_ordered_sum(t, n_ct, q->sum_order);
  }
  {
    /* 2286, more_tools.f90:3663-3667: the Psi_T locations with (1 + tau E_T) c */
    double *a = dw + n_imp + n_ct, *t = x;
    const double v1 = w->wt[q->loc_psit[0]];
    for (int64_t i = 1; i < n_psit; i++) {
      const double val = (1.0 + p->tau * p->e_trial) * q->cdet[i];
      a[i] = 0.0 + val * v1;
      t[i - 1] = val * w->wt[q->loc_psit[i]];
    }
    a[0] = (n_psit > 1) ? 0.0 + orc_ordered_sum(t, n_psit - 1, q->sum_order) : 0.0;
  }
  for (int64_t i = 0; i < n_ct; i++) w->wt[i] = w->wt[i] - p->tau * dw[n_imp + i] + p->tau * p->e_trial * w->wt[i];      /* 2313-2315 */
  for (int64_t i = 0; i < n_psit; i++) w->wt[q->loc_psit[i]] = w->wt[q->loc_psit[i]] + dw[n_imp + n_ct + i];            /* 2316-2318 */
  for (int64_t i = 0; i < n_imp; i++) w->wt[q->loc_imp[i]] = w->wt[q->loc_imp[i]] + dw[i];                               /* 2321-2323 */
  free(dw);
  /* ---- 2332-2333 */
  int64_t n = w->nwalk;
  psit_sort_spawns(w, n_ct + q->n_out, n);
  double wabs_before = 0; for (int64_t i = 0; i < n; i++) wabs_before += fabs(w->wt[i]);
  const int64_t nbefore = n;
  if (q->dbg_on) {
    free(q->dbg_up); free(q->dbg_dn); free(q->dbg_wt); free(q->dbg_d); free(q->dbg_i);
    q->dbg_n = n; q->dbg_n0 = n_ct + q->n_out;
    q->dbg_up = malloc(n * 8 + 8); q->dbg_dn = malloc(n * 8 + 8); q->dbg_wt = malloc(n * 8 + 8); q->dbg_d = malloc(n + 8); q->dbg_i = malloc(n + 8);
    memcpy(q->dbg_up, w->up, n * 8); memcpy(q->dbg_dn, w->dn, n * 8); memcpy(q->dbg_wt, w->wt, n * 8); memcpy(q->dbg_d, w->imp_distance, n); memcpy(q->dbg_i, w->initiator, n);
  }
  if (psit_merge3(w, q, &n, p)) { free(x); return 1; }                   /* 2368 */
  /* ---- T^-1 and its transpose on the Psi_T locations, 2394-2442 */
  {
    double tmp = 0.0;
    if (n_psit > 1) {
      for (int64_t i = 1; i < n_psit; i++) x[i - 1] = q->cdet[i] * w->wt[q->loc_psit[i]];
      tmp = 0.0 + orc_ordered_sum(x, n_psit - 1, q->sum_order);
    }
    const int64_t l0 = q->loc_psit[0];
    w->wt[l0] = w->wt[l0] - tmp;
    w->wt[l0] = w->wt[l0] / q->cdet[0];
    w->wt[l0] = w->wt[l0] / q->cdet[0];
    tmp = w->wt[l0];
    for (int64_t i = 1; i < n_psit; i++) w->wt[q->loc_psit[i]] = w->wt[q->loc_psit[i]] - q->cdet[i] * tmp;
  }
  free(x);
  /* ---- 2444-2462: the initiator criterion on C(T), nothing is discarded */
  {
    int i_perm = 0;
    if (!p->c_t_initiator) { for (int64_t i = 0; i < n_ct; i++) check_initiator(w, i, p, &i_perm); }
    else { for (int64_t i = 0; i < w->n_perm; i++) check_initiator(w, i, p, &i_perm); }      /* literally the first n_permanent_initiator SLOTS */
  }
  n = psit_reduce(w, q, n, p);                                           /* 2473 */
  w->nwalk = n;
  for (int64_t i = 0; i < n; i++) w->wt[i] = w->wt[i] * p->reweight_factor_inv;   /* 2487 */
  if (n == 0) return 4;
  double w_gen = 0, w2 = 0, w_abs = 0, w_abs_imp = 0, w_perm = 0; int ip = 0;      /* 2573-2598 */
  for (int64_t i = 0; i < n; i++) {
    if (w->initiator[i] == 3) w_perm += w->wt[i] * w->sign_perm[ip++];
    w_gen += w->wt[i]; w2 += w->wt[i] * w->wt[i]; w_abs += fabs(w->wt[i]);
    if (w->imp_distance[i] == 0 || (w->imp_distance[i] == -2 && p->c_t_initiator)) w_abs_imp += fabs(w->wt[i]);
  }
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t j = 0; j < n_ct; j++) {                                   /* 2701-2722 */
    double en, ed;
    if (j == 0) { en = w->ct_num[0] / w->ct_den[0] * w->wt[0]; ed = w->wt[0]; }
    else { en = w->ct_num[j] * w->wt[j]; ed = w->ct_den[j] * w->wt[j]; }
    if (fabs(ed) < 1e-22) ed = fabs(ed);
    acc[1] += ed; acc[3] += ed * ed; acc[5] += fabs(ed);
    acc[0] += en; acc[2] += en * en; acc[4] += en * copysign(1.0, ed); acc[6] += en * ed;
  }
  out[0] = w_gen; out[1] = w_abs; out[2] = acc[1]; out[3] = acc[0]; out[4] = w_perm; out[5] = (double)n;
  out[6] = w_abs_imp; out[7] = (double)nbefore; out[8] = w2; out[9] = acc[2]; out[10] = acc[3];
  out[11] = acc[4]; out[12] = acc[5]; out[13] = acc[6]; out[14] = wabs_before; out[15] = (double)attempts;
  w->rng.step++;
  return 0;
}
int orc_walk_step_psit(const orc_chem *c, orc_walk *w, orc_psit *q, const orc_step_params *p, double out[16]) {
  orc_sys y = {c, NULL, NULL, NULL, 1, 0, 0};
  return walk_step_psit_sys(&y, w, q, p, out);
}
int orc_walk_step_psit_heg(const orc_heg *h, orc_walk *w, orc_psit *q, const orc_step_params *p, double out[16]) {
  orc_sys y = {NULL, h, NULL, NULL, 1, 0, 0};
  return walk_step_psit_sys(&y, w, q, p, out);
}
