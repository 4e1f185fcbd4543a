/* sqmc_oracle_hb.c -- TEST INFRASTRUCTURE (part of the CPU oracle): the efficient heat-bath proposal of sqmc
 * (Holmes, Changlani, Umrigar), restated from the reference:
 *   setup_efficient_heatbath                       chemistry.f90:872-1230   (run_type /= hci branch, 1002-1225)
 *   off_diagonal_move_chem_efficient_heatbath      chemistry.f90:5086-5347
 *   apply_time_reversal_symmetry                   chemistry.f90:5350-5427
 *   proposal_prob_efficient_heatbath               chemistry.f90:5431-5549
 *   generate_heatbath_single / prob_heatbath_single  5553-5626
 *   generate_heatbath_double / prob_heatbath_double  5629-5787
 *   p_single_excit / p_double_excit                5791-5816
 *   same_index / opposite_index / p_first_hole / p_second_hole / Htot / compute_single_elem
 *   choose_first_hole / choose_second_hole / check_heatbath_unbiased          9154-9375
 *   setup_alias (rk and real), sample_alias, sample_discrete_distribution      more_tools.f90:5603-5780, 4102-4135
 *
 * PINNING.  The reference holds no fixture for this proposal and refuses it on every system it ships (SURVEY finding 4);
 * chemistry.f90 does not compile unmodified here.  Two accumulators of prob_heatbath_double -- sum_of_one_elec_probs
 * (5728) and, in its else branch, sum_of_probs_of_choosing_2_and_other (5742) -- are never initialised in the source and
 * only work because the reference is built with -finit-local-zero (src/Makefile:24-25).  This restatement gives them the
 * mathematically intended value: they start from 0 on every call (tests/golden/README_heatbath.md).
 * What pins it instead: the proposal must be unbiased -- sum_j p(i -> j) = 1 over the connected determinants, p > 0 wherever
 * H_ij /= 0, and weight_j = -tau H_ij / p(i -> j) -- which tests/test_oracle.py checks by exhaustive enumeration. */
#include "sqmc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int trailz64(det_t d) { return __builtin_ctzll(d); }
static inline det_t bit64(int k) { return (det_t)1 << k; }

/* combine_2_indices, chemistry.f90:9138-9150 */
static long c2i(int i, int j) { return i > j ? ((long)i * (i - 1)) / 2 + j : ((long)j * (j - 1)) / 2 + i; }

/* indices: 1-based like the reference; arrays carry an unused 0 row/column */
#define N1 (h->norb + 1)
#define ONE(i) h->one[(i)]
#define TWO(i, j) h->two[(size_t)(i) * (2 * h->norb + 1) + (j)]
#define IX3(i, j, k) (((size_t)(i) * N1 + (j)) * N1 + (k))
#define HSAME(p, k) h->htot_same[(size_t)(p) * N1 + (k)]

/* same_index / opposite_index, chemistry.f90:9154-9193 (spin-orbital arguments are folded to orbitals) */
static int fold(const orc_hb *h, int i) { return i > h->norb ? i - h->norb : i; }
int64_t orc_hb_same_index(const orc_hb *h, int f1, int f2, int t1, int t2) {
  const int i = fold(h, f1), j = fold(h, f2), k = fold(h, t1), l = fold(h, t2), n = h->norb;
  return (int64_t)(c2i(i, j) - 1) * n * n + (int64_t)(k - 1) * n + l;
}
int64_t orc_hb_opposite_index(const orc_hb *h, int f1, int f2, int t1, int t2) {
  const int i = fold(h, f1), j = fold(h, f2), k = fold(h, t1), l = fold(h, t2); const int64_t n = h->norb;
  return (int64_t)(i - 1) * n * n * n + (int64_t)(j - 1) * n * n + (int64_t)(k - 1) * n + l;
}

/* setup_alias_rk / setup_alias_real, more_tools.f90:5603-5722 (1-based J; slices of length K) */
static void setup_alias_d(int K, const double *pdf, int *smaller, int *larger, int *J, double *q) {
  int n_s = 0, n_l = 0;
  for (int i = 1; i <= K; i++) {
    J[i - 1] = i; q[i - 1] = K * pdf[i - 1];
    if (q[i - 1] < 1.0) smaller[n_s++] = i; else larger[n_l++] = i;
  }
  while (n_s > 0 && n_l > 0) {
    const int small = smaller[n_s - 1], large = larger[n_l - 1];
    J[small - 1] = large;
    q[large - 1] = q[large - 1] + q[small - 1] - 1.0;
    if (q[large - 1] < 1.0) { smaller[n_s - 1] = large; n_l--; } else n_s--;
  }
}
static void setup_alias_f(int K, const float *pdf, int *smaller, int *larger, int *J, float *q) {
  int n_s = 0, n_l = 0;
  for (int i = 1; i <= K; i++) {
    J[i - 1] = i; q[i - 1] = (float)K * pdf[i - 1];
    if (q[i - 1] < 1.0f) smaller[n_s++] = i; else larger[n_l++] = i;
  }
  while (n_s > 0 && n_l > 0) {
    const int small = smaller[n_s - 1], large = larger[n_l - 1];
    J[small - 1] = large;
    q[large - 1] = q[large - 1] + q[small - 1] - 1.0f;
    if ((double)q[large - 1] < 1.0) { smaller[n_s - 1] = large; n_l--; } else n_s--;
  }
}
/* sample_alias, more_tools.f90:5727-5778: one random_int, one rannyu */
static int sample_alias_d(orc_rng *g, int K, const int *J, const double *q) { const int i = orc_random_int(g, K); return (orc_rannyu(g) < q[i - 1]) ? i : J[i - 1]; }
static int sample_alias_f(orc_rng *g, int K, const int *J, const float *q) { const int i = orc_random_int(g, K); return (orc_rannyu(g) < (double)q[i - 1]) ? i : J[i - 1]; }
/* sample_discrete_distribution, more_tools.f90:4102-4135 (1-based result) */
static int sample_discrete(orc_rng *g, const double *c_probs, int n) {
  const double r = orc_rannyu(g);
  int lo = 1, hi = n;
  while (lo < hi) { const int mid = (lo + hi) / 2; if (r < c_probs[mid - 1]) hi = mid; else lo = mid + 1; }
  return lo;
}

/* |H| of the double excitation among bare orbitals: hamiltonian_chem(..., 2, H, nosign) on two-electron determinants */
static double habs_same(const orc_chem *c, int i, int j, int k, int l) {
  return fabs(orc_hamiltonian_chem(c, bit64(i - 1) | bit64(j - 1), 0, bit64(k - 1) | bit64(l - 1), 0, 2));
}
static double habs_opp(const orc_chem *c, int i, int j, int k, int l) {       /* up i -> k, dn j -> l */
  return fabs(orc_hamiltonian_chem(c, bit64(i - 1), bit64(j - 1), bit64(k - 1), bit64(l - 1), 2));
}

/* setup_efficient_heatbath, chemistry.f90:1002-1225 */
orc_hb *orc_hb_setup(const orc_chem *c) {
  orc_hb *h = calloc(1, sizeof(orc_hb));
  const int n = c->norb, nc = c->n_core_orb;
  h->norb = n; h->nup = c->nup; h->ndn = c->ndn; h->n_core = nc;
  const size_t n3 = (size_t)(n + 1) * (n + 1) * (n + 1);
  h->one = calloc(n + 1, sizeof(double));
  h->two = calloc((size_t)(2 * n + 1) * (2 * n + 1), sizeof(double));
  h->three_same = calloc(n3, sizeof(double)); h->three_opp = calloc(n3, sizeof(double));
  h->j3_same = calloc(n3, sizeof(int)); h->j3_opp = calloc(n3, sizeof(int));
  h->q3_same = calloc(n3, sizeof(double)); h->q3_opp = calloc(n3, sizeof(double));
  h->size_same = orc_hb_same_index(h, n, n, n, n); h->size_opp = orc_hb_opposite_index(h, n, n, n, n);
  h->four_same = calloc(h->size_same + 1, sizeof(float)); h->four_opp = calloc(h->size_opp + 1, sizeof(float));
  h->j4_same = calloc(h->size_same + 1, sizeof(int)); h->j4_opp = calloc(h->size_opp + 1, sizeof(int));
  h->q4_same = calloc(h->size_same + 1, sizeof(float)); h->q4_opp = calloc(h->size_opp + 1, sizeof(float));
  h->n_pairs = c2i(n, n);
  h->htot_same = calloc((size_t)(h->n_pairs + 1) * (n + 1), sizeof(double)); h->htot_opp = calloc(n3, sizeof(double));
  int *tmp1 = calloc(2 * n + 2, sizeof(int)), *tmp2 = calloc(2 * n + 2, sizeof(int));
  /* same spin, 1054-1075 */
  for (int i = nc + 1; i <= n; i++) for (int j = nc + 1; j <= n; j++) for (int k = nc + 1; k <= n; k++) for (int l = nc + 1; l <= n; l++) {
    if (i == j || k == l || i == k || i == l || j == k || j == l) continue;
    const double a = habs_same(c, i, j, k, l);
    ONE(i) = ONE(i) + a;
    TWO(i, j) = TWO(i, j) + a; TWO(n + i, n + j) = TWO(n + i, n + j) + a;
    h->three_same[IX3(i, j, k)] = h->three_same[IX3(i, j, k)] + a;
    float *f = &h->four_same[orc_hb_same_index(h, i, j, k, l)];
    *f = (float)((double)*f + a);                                  /* real(single + double) */
  }
  /* opposite spin, 1078-1105 */
  for (int i = nc + 1; i <= n; i++) for (int j = nc + 1; j <= n; j++) for (int k = nc + 1; k <= n; k++) for (int l = nc + 1; l <= n; l++) {
    if (!(i == k || j == l)) {
      const double a = habs_opp(c, i, j, k, l);
      ONE(i) = ONE(i) + a;
      TWO(i, n + j) = TWO(i, n + j) + a; TWO(n + i, j) = TWO(n + i, j) + a;
      h->three_opp[IX3(i, j, k)] = h->three_opp[IX3(i, j, k)] + a;
      float *f = &h->four_opp[orc_hb_opposite_index(h, i, j, k, l)];
      *f = (float)((double)*f + a);
    }
    if (!(i == l || j == k)) {
      const double a = habs_opp(c, i, j, l, k);
      ONE(i) = ONE(i) + a;
      TWO(i, n + j) = TWO(i, n + j) + a; TWO(n + i, j) = TWO(n + i, j) + a;
    }
  }
  /* cumulative three-orbital probabilities and Htot, 1110-1145 */
  for (int i = nc + 1; i <= n; i++) for (int j = nc + 1; j <= n; j++) {
    double c_same = 0.0, c_opp = 0.0;
    for (int k = nc + 1; k <= n; k++) {
      for (int l = nc + 1; l <= n; l++) {
        if (i > j) HSAME(c2i(i, j), k) = HSAME(c2i(i, j), k) + (double)h->four_same[orc_hb_same_index(h, i, j, k, l)] * 2.0;
        h->htot_opp[IX3(i, j, k)] = h->htot_opp[IX3(i, j, k)] + (double)h->four_opp[orc_hb_opposite_index(h, i, j, k, l)] * 2.0;
      }
      c_same = (k == nc + 1) ? h->three_same[IX3(i, j, k)] : h->three_same[IX3(i, j, k)] + c_same;
      c_opp = (k == nc + 1) ? h->three_opp[IX3(i, j, k)] : h->three_opp[IX3(i, j, k)] + c_opp;
    }
    if (c_same > 0.0) for (int k = 1; k <= n; k++) h->three_same[IX3(i, j, k)] = h->three_same[IX3(i, j, k)] / c_same;
    if (c_opp > 0.0) for (int k = 1; k <= n; k++) h->three_opp[IX3(i, j, k)] = h->three_opp[IX3(i, j, k)] / c_opp;
  }
  /* alias tables of the first hole, 1147-1152 */
  for (int i = nc + 1; i <= n; i++) for (int j = nc + 1; j <= n; j++) {
    setup_alias_d(n, &h->three_same[IX3(i, j, 1)], tmp1, tmp2, &h->j3_same[IX3(i, j, 1)], &h->q3_same[IX3(i, j, 1)]);
    setup_alias_d(n, &h->three_opp[IX3(i, j, 1)], tmp1, tmp2, &h->j3_opp[IX3(i, j, 1)], &h->q3_opp[IX3(i, j, 1)]);
  }
  /* four-index tensors: normalise every (i,j,k,:) row and set up its alias table, 1157-1187 */
  for (int i = nc + 1; i <= n; i++) for (int j = nc + 1; j <= n; j++) for (int k = nc + 1; k <= n; k++) {
    double cs = 0.0, co = 0.0;
    const int64_t bs = orc_hb_same_index(h, i, j, k, 1), bo = orc_hb_opposite_index(h, i, j, k, 1);
    for (int l = nc + 1; l <= n; l++) cs = (l == nc + 1) ? (double)h->four_same[bs + l - 1] : cs + (double)h->four_same[bs + l - 1];
    if (cs > 0.0) {
      for (int l = 1; l <= n; l++) h->four_same[bs + l - 1] = (float)((double)h->four_same[bs + l - 1] / cs);
      setup_alias_f(n, &h->four_same[bs], tmp1, tmp2, &h->j4_same[bs], &h->q4_same[bs]);
    }
    for (int l = nc + 1; l <= n; l++) co = (l == nc + 1) ? (double)h->four_opp[bo + l - 1] : co + (double)h->four_opp[bo + l - 1];
    if (co > 0.0) {
      for (int l = 1; l <= n; l++) h->four_opp[bo + l - 1] = (float)((double)h->four_opp[bo + l - 1] / co);
      setup_alias_f(n, &h->four_opp[bo], tmp1, tmp2, &h->j4_opp[bo], &h->q4_opp[bo]);
    }
  }
  free(tmp1); free(tmp2);
  /* check_heatbath_unbiased, 9330-9375: max(nup, ndn) > number of orbitals no single excitation connects to another one */
  int n_uniq = 0;
  for (int p = 1; p <= n; p++) {
    int s = 0;
    for (int q = 1; q <= n && !s; q++) { if (p == q) continue; if (fabs(orc_hamiltonian_chem(c, bit64(p - 1), 0, bit64(q - 1), 0, 1)) > 1.e-10) s = 1; }
    if (!s) n_uniq++;
  }
  h->n_orb_uniq_sym = n_uniq;
  h->unbiased = ((c->nup > c->ndn ? c->nup : c->ndn) > n_uniq) ? 1 : 0;
  return h;
}
void orc_hb_free(orc_hb *h) {
  if (!h) return;
  free(h->one); free(h->two); free(h->three_same); free(h->three_opp); free(h->j3_same); free(h->j3_opp); free(h->q3_same); free(h->q3_opp);
  free(h->four_same); free(h->four_opp); free(h->j4_same); free(h->j4_opp); free(h->q4_same); free(h->q4_opp); free(h->htot_same); free(h->htot_opp); free(h);
}

/* p_first_hole / p_second_hole / Htot, chemistry.f90:9195-9260 */
static double p_first_hole(const orc_hb *h, int f1, int f2, int t1) {
  const int n = h->norb;
  if (f1 <= n) return (f2 <= n) ? h->three_same[IX3(f1, f2, t1)] : h->three_opp[IX3(f1, f2 - n, t1)];
  return (f2 <= n) ? h->three_opp[IX3(f1 - n, f2, t1 - n)] : h->three_same[IX3(f1 - n, f2 - n, t1 - n)];
}
static double p_second_hole(const orc_hb *h, int f1, int f2, int t1, int t2) {
  const int n = h->norb;
  if ((f1 <= n && f2 <= n) || (f1 > n && f2 > n)) return (double)h->four_same[orc_hb_same_index(h, f1, f2, t1, t2)];
  return (double)h->four_opp[orc_hb_opposite_index(h, f1, f2, t1, t2)];
}
static double Htot(const orc_hb *h, int f1, int f2, int t1) {
  const int n = h->norb;
  if (f1 <= n) return (f2 <= n) ? HSAME(c2i(f1, f2), t1) : h->htot_opp[IX3(f1, f2 - n, t1)];
  return (f2 <= n) ? h->htot_opp[IX3(f1 - n, f2, t1 - n)] : HSAME(c2i(f1 - n, f2 - n), t1 - n);
}
/* p_single_excit / p_double_excit, 5791-5816 */
static double p_single_excit(double single_elem, double sum_doubles) { double p = fabs(single_elem) / (fabs(single_elem) + sum_doubles); if (p > 0.5) p = 1.0; return p; }
static double p_double_excit(double single_elem, double sum_doubles) { double p = sum_doubles / (fabs(single_elem) + sum_doubles); if (p < 0.5) p = 1.0; return p; }
/* compute_single_elem, 9262-9276 */
static double compute_single_elem(const orc_chem *c, const orc_hb *h, det_t up, det_t dn, int f1, int t1) {
  const int n = h->norb;
  if (f1 <= n) return orc_hamiltonian_chem(c, up, dn, (up & ~bit64(f1 - 1)) | bit64(t1 - 1), dn, 1);
  return orc_hamiltonian_chem(c, up, dn, up, (dn & ~bit64(f1 - n - 1)) | bit64(t1 - n - 1), 1);
}

typedef struct { int occ_up[ORC_MAXORB], occ_dn[ORC_MAXORB]; } occ_t;
static void occ_lists(const orc_hb *h, det_t up, det_t dn, occ_t *o) {
  int k = 0; for (det_t t = up; t; t &= t - 1) o->occ_up[k++] = trailz64(t) + 1;
  k = 0; for (det_t t = dn; t; t &= t - 1) o->occ_dn[k++] = trailz64(t) + 1;
  (void)h;
}

/* prob_heatbath_single, 5580-5626 */
static double prob_heatbath_single(const orc_hb *h, const occ_t *o, int f1, int t1, double matrix_element, double normalization) {
  const int n = h->norb; double pp = 0.0; const double sing_num = fabs(matrix_element);
  if (f1 <= n) {
    for (int i = 0; i < h->nup; i++) { const int e = o->occ_up[i]; if (f1 == e) continue;
      pp = pp + TWO(f1, e) * h->three_same[IX3(f1, e, t1)] * p_single_excit(sing_num, HSAME(c2i(f1, e), t1)); }
    for (int i = 0; i < h->ndn; i++) { const int e = o->occ_dn[i];
      pp = pp + TWO(f1, e + n) * h->three_opp[IX3(f1, e, t1)] * p_single_excit(sing_num, h->htot_opp[IX3(f1, e, t1)]); }
  } else {
    for (int i = 0; i < h->nup; i++) { const int e = o->occ_up[i];
      pp = pp + TWO(f1, e) * h->three_opp[IX3(f1 - n, e, t1 - n)] * p_single_excit(sing_num, h->htot_opp[IX3(f1 - n, e, t1 - n)]); }
    for (int i = 0; i < h->ndn; i++) { const int e = o->occ_dn[i]; if (f1 == e + n) continue;
      pp = pp + TWO(f1, e + n) * h->three_same[IX3(f1 - n, e, t1 - n)] * p_single_excit(sing_num, HSAME(c2i(f1 - n, e), t1 - n)); }
  }
  return pp * normalization;
}

/* prob_heatbath_double, 5693-5787.  sum_of_one_elec_probs and (else branch) sum_of_probs_of_choosing_2_and_other start from 0:
 * the intended value (the source leaves them to -finit-local-zero). */
static double prob_heatbath_double(const orc_chem *c, const orc_hb *h, det_t up, det_t dn, const occ_t *o, int f1, int f2, int t1, int t2,
                                   double matrix_element, double prob_1_then_2, int same_spin) {
  const int n = h->norb;
  double sum_one = 0.0, sum_2_other = 0.0, one_elec_prob_2 = 0.0;
  if (f2 <= n) {
    for (int i = 0; i < h->nup; i++) { const int e = o->occ_up[i]; sum_one = sum_one + ONE(e); if (f2 == e) { one_elec_prob_2 = ONE(e); continue; } sum_2_other = sum_2_other + TWO(f2, e); }
    for (int i = 0; i < h->ndn; i++) { const int e = o->occ_dn[i]; sum_one = sum_one + ONE(e); sum_2_other = sum_2_other + TWO(f2, e + n); }
  } else {
    for (int i = 0; i < h->nup; i++) { const int e = o->occ_up[i]; sum_one = sum_one + ONE(e); sum_2_other = sum_2_other + TWO(f2, e); }
    for (int i = 0; i < h->ndn; i++) { const int e = o->occ_dn[i]; sum_one = sum_one + ONE(e); if (f2 == e + n) { one_elec_prob_2 = ONE(e); continue; } sum_2_other = sum_2_other + TWO(f2, e + n); }
  }
  const double prob_of_1_and_2 = TWO(f1, f2);
  const double prob_2_then_1 = one_elec_prob_2 / sum_one * prob_of_1_and_2 / sum_2_other;
  double term1, term2, term3, term4, sing_num;
  term1 = p_first_hole(h, f1, f2, t1) * p_second_hole(h, f1, f2, t1, t2);
  sing_num = fabs(matrix_element);
  term1 = term1 * p_double_excit(sing_num, Htot(h, f1, f2, t1));
  if (same_spin) {
    term2 = p_first_hole(h, f1, f2, t2) * p_second_hole(h, f1, f2, t2, t1);
    sing_num = fabs(compute_single_elem(c, h, up, dn, f1, t2));
    term2 = term2 * p_double_excit(sing_num, Htot(h, f1, f2, t2));
    term3 = p_first_hole(h, f2, f1, t1) * p_second_hole(h, f2, f1, t1, t2);
    sing_num = fabs(compute_single_elem(c, h, up, dn, f2, t1));
    term3 = term3 * p_double_excit(sing_num, Htot(h, f2, f1, t1));
  } else { term2 = 0.0; term3 = 0.0; }
  term4 = p_first_hole(h, f2, f1, t2) * p_second_hole(h, f2, f1, t2, t1);
  sing_num = fabs(compute_single_elem(c, h, up, dn, f2, t2));
  term4 = term4 * p_double_excit(sing_num, Htot(h, f2, f1, t2));
  return prob_1_then_2 * (term1 + term2) + prob_2_then_1 * (term3 + term4);
}

/* proposal_prob_efficient_heatbath, 5431-5549: probability with which the move det_i -> det_j WOULD be proposed */
double orc_hb_proposal_prob(const orc_chem *c, const orc_hb *h, det_t iu, det_t id, det_t ju, det_t jd, int excite_level, double off_diag_elem) {
  const int n = h->norb; occ_t o; occ_lists(h, iu, id, &o);
  int f1 = 0, f2 = 0, t1 = 0, t2 = 0, excite_spin;
  if (excite_level == 1) {
    if (iu == ju) { excite_spin = -1; f1 = trailz64(id & ~jd) + n + 1; t1 = trailz64(jd & ~id) + n + 1; }
    else { excite_spin = 1; f1 = trailz64(iu & ~ju) + 1; t1 = trailz64(ju & ~iu) + 1; }
  } else {
    if (iu == ju) {
      excite_spin = -1;
      det_t t = id & ~jd; f1 = trailz64(t) + 1; f2 = trailz64(t & ~bit64(f1 - 1)) + 1;
      t = jd & ~id; t1 = trailz64(t) + 1; t2 = trailz64(t & ~bit64(t1 - 1)) + 1;
      f1 += n; f2 += n; t1 += n; t2 += n;
    } else if (id == jd) {
      excite_spin = 1;
      det_t t = iu & ~ju; f1 = trailz64(t) + 1; f2 = trailz64(t & ~bit64(f1 - 1)) + 1;
      t = ju & ~iu; t1 = trailz64(t) + 1; t2 = trailz64(t & ~bit64(t1 - 1)) + 1;
    } else {
      excite_spin = 0;
      f1 = trailz64(iu & ~ju) + 1; f2 = trailz64(id & ~jd) + n + 1; t1 = trailz64(ju & ~iu) + 1; t2 = trailz64(jd & ~id) + n + 1;
    }
  }
  double sum_one = 0.0, sum_1_other = 0.0, one_elec_prob_1 = 0.0;
  if (f1 <= n) {
    for (int i = 0; i < h->nup; i++) { const int e = o.occ_up[i]; sum_one = sum_one + ONE(e); if (f1 == e) { one_elec_prob_1 = ONE(e); continue; } sum_1_other = sum_1_other + TWO(f1, e); }
    for (int i = 0; i < h->ndn; i++) { const int e = o.occ_dn[i]; sum_one = sum_one + ONE(e); sum_1_other = sum_1_other + TWO(f1, e + n); }
  } else {
    for (int i = 0; i < h->nup; i++) { const int e = o.occ_up[i]; sum_one = sum_one + ONE(e); sum_1_other = sum_1_other + TWO(f1, e); }
    for (int i = 0; i < h->ndn; i++) { const int e = o.occ_dn[i]; sum_one = sum_one + ONE(e); if (f1 == e + n) { one_elec_prob_1 = ONE(e); continue; } sum_1_other = sum_1_other + TWO(f1, e + n); }
  }
  if (excite_level == 1) {
    const double normalization = one_elec_prob_1 / sum_one / sum_1_other;
    return prob_heatbath_single(h, &o, f1, t1, off_diag_elem, normalization);
  }
  const double single_elem = compute_single_elem(c, h, iu, id, f1, t1);
  const double prob_1_then_2 = one_elec_prob_1 / sum_one * TWO(f1, f2) / sum_1_other;
  return prob_heatbath_double(c, h, iu, id, &o, f1, f2, t1, t2, single_elem, prob_1_then_2, excite_spin != 0);
}

/* apply_time_reversal_symmetry, 5350-5427.  Returns 0 when the reference returns early with norm_j = 0 (z = -1, j_up = j_dn). */
static void apply_time_reversal(const orc_chem *c, const orc_hb *h, det_t iu, det_t id, det_t *ju, det_t *jd, double *me, double *pp) {
  const double sqrt2 = sqrt(2.0);
  if ((*ju == iu && *jd == id) || (*jd == iu && *ju == id)) { *me = 0.0; return; }
  const double norm_i = (iu == id) ? sqrt2 : 1.0;      /* z = -1 with up = dn never enters as an incoming state */
  double norm_j = 1.0;
  if (*ju == *jd) {
    if (c->z == 1) norm_j = sqrt2; else return;
    *me = (norm_j / norm_i) * *me;
  } else {
    const int lev = orc_excitation_level(iu, id, *jd, *ju);
    if (lev >= 0) {
      const double me2 = orc_hamiltonian_chem(c, iu, id, *jd, *ju, lev);
      if (fabs(me2) > 1.0e-10) {
        const double ps = orc_hb_proposal_prob(c, h, iu, id, *jd, *ju, lev, me2);
        *pp = *pp + ps;
        *me = (norm_j / norm_i) * (*me + c->z * me2);
      } else *me = (norm_j / norm_i) * *me;
    } else *me = (norm_j / norm_i) * (*me);
  }
  if (*ju > *jd) { const det_t t = *ju; *ju = *jd; *jd = t; *me = *me * c->z; }
}

/* off_diagonal_move_chem_efficient_heatbath, 5086-5347.  Returns n_new_dets (0, 1 or 2); weight_j[k] = 0 for a slot that holds no move. */
int orc_off_diagonal_move_chem_heatbath(const orc_chem *c, const orc_hb *h, orc_rng *g, double tau, det_t iu, det_t id,
                                        det_t ju[2], det_t jd[2], double weight_j[2], int excite_level[2], int *n_draws) {
  const int n = h->norb, nup = h->nup, ndn = h->ndn, nelec = nup + ndn;
  int draws = 0;
  ju[0] = ju[1] = iu; jd[0] = jd[1] = id; weight_j[0] = weight_j[1] = 0.0; excite_level[0] = excite_level[1] = -1;
  occ_t o; occ_lists(h, iu, id, &o);
  int elecs[2 * ORC_MAXORB]; double e1_prob[2 * ORC_MAXORB], c_e1[2 * ORC_MAXORB], e2_prob[2 * ORC_MAXORB], c_e2[2 * ORC_MAXORB];
  for (int i = 0; i < nup; i++) { elecs[i] = o.occ_up[i]; e1_prob[i] = ONE(elecs[i]); c_e1[i] = ONE(elecs[i]); if (i > 0) c_e1[i] = c_e1[i] + c_e1[i - 1]; }
  for (int i = 0; i < ndn; i++) { elecs[i + nup] = o.occ_dn[i] + n; e1_prob[i + nup] = ONE(o.occ_dn[i]); c_e1[i + nup] = ONE(o.occ_dn[i]); c_e1[i + nup] = c_e1[i + nup] + c_e1[i + nup - 1]; }
  { const double tot = c_e1[nelec - 1]; for (int i = 0; i < nelec; i++) e1_prob[i] = e1_prob[i] / tot; for (int i = 0; i < nelec; i++) c_e1[i] = c_e1[i] / tot; }
  int i = sample_discrete(g, c_e1, nelec); draws++;
  const int f1 = elecs[i - 1];
  double proposal_prob = e1_prob[i - 1]; const double e1_prob_sav = e1_prob[i - 1];
  for (int k = 0; k < nup; k++) {
    if (elecs[k] == f1) { e2_prob[k] = 0.0; c_e2[k] = 0.0; } else { e2_prob[k] = TWO(f1, o.occ_up[k]); c_e2[k] = TWO(f1, o.occ_up[k]); }
    if (k > 0) c_e2[k] = c_e2[k] + c_e2[k - 1];
  }
  for (int k = 0; k < ndn; k++) {
    if (elecs[k + nup] == f1) { e2_prob[k + nup] = 0.0; c_e2[k + nup] = 0.0; } else { e2_prob[k + nup] = TWO(f1, o.occ_dn[k] + n); c_e2[k + nup] = TWO(f1, o.occ_dn[k] + n); }
    c_e2[k + nup] = c_e2[k + nup] + c_e2[k + nup - 1];
  }
  const double c_e2_sav = c_e2[nelec - 1];
  for (int k = 0; k < nelec; k++) e2_prob[k] = e2_prob[k] / c_e2_sav;
  for (int k = 0; k < nelec; k++) c_e2[k] = c_e2[k] / c_e2_sav;
  i = sample_discrete(g, c_e2, nelec); draws++;
  const int f2 = elecs[i - 1];
  proposal_prob = proposal_prob * e2_prob[i - 1];
  /* choose_first_hole, 9278-9304 */
  int t1, excite_spin;
  if (f1 <= n && f2 <= n) { excite_spin = 1; t1 = sample_alias_d(g, n, &h->j3_same[IX3(f1, f2, 1)], &h->q3_same[IX3(f1, f2, 1)]); }
  else if (f1 > n && f2 > n) { excite_spin = -1; t1 = sample_alias_d(g, n, &h->j3_same[IX3(f1 - n, f2 - n, 1)], &h->q3_same[IX3(f1 - n, f2 - n, 1)]) + n; }
  else {
    excite_spin = 0;
    if (f1 > n) t1 = sample_alias_d(g, n, &h->j3_opp[IX3(f1 - n, f2, 1)], &h->q3_opp[IX3(f1 - n, f2, 1)]) + n;
    else t1 = sample_alias_d(g, n, &h->j3_opp[IX3(f1, f2 - n, 1)], &h->q3_opp[IX3(f1, f2 - n, 1)]);
  }
  draws += 2;
  if (t1 <= n) { if ((iu >> (t1 - 1)) & 1) { if (n_draws) *n_draws = draws; return 0; } }
  else { if ((id >> (t1 - n - 1)) & 1) { if (n_draws) *n_draws = draws; return 0; } }
  double matrix_element = compute_single_elem(c, h, iu, id, f1, t1);
  const int same_spin = (excite_spin != 0);
  const double sing_num = fabs(matrix_element), sing_den = sing_num + Htot(h, f1, f2, t1);
  int n_new;
#define CHOOSE_SECOND_HOLE(T2) do {                                                                            \
    if (!same_spin) { const int64_t b_ = orc_hb_opposite_index(h, f1, f2, t1, 1);                              \
      T2 = sample_alias_f(g, n, &h->j4_opp[b_], &h->q4_opp[b_]); if (f1 <= n) T2 += n; }                       \
    else { const int64_t b_ = orc_hb_same_index(h, f1, f2, t1, 1);                                             \
      T2 = sample_alias_f(g, n, &h->j4_same[b_], &h->q4_same[b_]); if (f1 > n) T2 += n; }                      \
    draws += 2; } while (0)
#define OCCUPIED(T) (((T) <= n) ? (int)((iu >> ((T) - 1)) & 1) : (int)((id >> ((T) - n - 1)) & 1))
#define MAKE_SINGLE(K) do { if (f1 <= n) { ju[K] = (iu & ~bit64(f1 - 1)) | bit64(t1 - 1); jd[K] = id; }        \
                            else { jd[K] = (id & ~bit64(f1 - n - 1)) | bit64(t1 - n - 1); ju[K] = iu; } } while (0)
#define MAKE_DOUBLE(K, T2) do { det_t u_ = iu, d_ = id;                                                        \
    if (f1 <= n) u_ &= ~bit64(f1 - 1); else d_ &= ~bit64(f1 - n - 1);                                          \
    if (f2 <= n) u_ &= ~bit64(f2 - 1); else d_ &= ~bit64(f2 - n - 1);                                          \
    if (t1 <= n) u_ |= bit64(t1 - 1); else d_ |= bit64(t1 - n - 1);                                            \
    if ((T2) <= n) u_ |= bit64((T2) - 1); else d_ |= bit64((T2) - n - 1);                                      \
    ju[K] = u_; jd[K] = d_; } while (0)
  if (sing_num > (sing_den - sing_num)) {              /* a single larger than all its doubles together: propose both */
    const double prob_1_then_2 = proposal_prob;
    n_new = 2;
    const double normalization = e1_prob_sav / c_e2_sav;
    MAKE_SINGLE(0);
    double pp = prob_heatbath_single(h, &o, f1, t1, matrix_element, normalization), me = 0.0;
    /* as in the source, the time-reversal routine updates matrix_element in place, and the double below is then given the
     * updated value as "the single excitation's element" (5266-5296) */
    if (c->time_sym) apply_time_reversal(c, h, iu, id, &ju[0], &jd[0], &matrix_element, &pp);
    weight_j[0] = -tau * matrix_element / pp; excite_level[0] = 1;
    if (Htot(h, f1, f2, t1) == 0.0) { weight_j[1] = 0.0; if (n_draws) *n_draws = draws; return 1; }
    int t2; CHOOSE_SECOND_HOLE(t2);
    if (OCCUPIED(t2)) { weight_j[1] = 0.0; if (n_draws) *n_draws = draws; return 1; }
    MAKE_DOUBLE(1, t2);
    pp = prob_heatbath_double(c, h, iu, id, &o, f1, f2, t1, t2, matrix_element, prob_1_then_2, same_spin);
    me = orc_hamiltonian_chem(c, iu, id, ju[1], jd[1], 2);
    if (c->time_sym) apply_time_reversal(c, h, iu, id, &ju[1], &jd[1], &me, &pp);
    weight_j[1] = -tau * me / pp; excite_level[1] = 2;
  } else {
    n_new = 1;
    const double p_single = p_single_excit(sing_num, sing_den - sing_num);
    double pp, me;
    draws++;
    if (orc_rannyu(g) < p_single) {
      excite_level[0] = 1;
      const double normalization = e1_prob_sav / c_e2_sav;
      MAKE_SINGLE(0);
      pp = prob_heatbath_single(h, &o, f1, t1, matrix_element, normalization); me = matrix_element;
    } else {
      excite_level[0] = 2;
      int t2; CHOOSE_SECOND_HOLE(t2);
      if (OCCUPIED(t2)) { weight_j[0] = 0.0; if (n_draws) *n_draws = draws; return 0; }
      const double prob_1_then_2 = proposal_prob;
      MAKE_DOUBLE(0, t2);
      pp = prob_heatbath_double(c, h, iu, id, &o, f1, f2, t1, t2, matrix_element, prob_1_then_2, same_spin);
      me = orc_hamiltonian_chem(c, iu, id, ju[0], jd[0], 2);
    }
    if (c->time_sym) apply_time_reversal(c, h, iu, id, &ju[0], &jd[0], &me, &pp);
    weight_j[0] = -tau * me / pp;
  }
  if (n_draws) *n_draws = draws;
  return n_new;
}
