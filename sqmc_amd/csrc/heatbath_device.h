// heatbath_device.h -- the efficient heat-bath proposal (Holmes, Changlani, Umrigar) on the device.  gfx950 only.
// Textually included by sqmc_gpu.hip behind chem_device.h.
//
// Reference: off_diagonal_move_chem_efficient_heatbath (chemistry.f90:5086-5347) with prob_heatbath_single / _double
// (5580-5787), p_single_excit / p_double_excit (5791-5816), proposal_prob_efficient_heatbath (5431-5549),
// apply_time_reversal_symmetry (5350-5427), p_first_hole / p_second_hole / Htot / compute_single_elem / choose_first_hole /
// choose_second_hole (9195-9327), sample_alias and sample_discrete_distribution (more_tools.f90:5727-5778, 4102-4135).
// The tables are the reference's (setup_efficient_heatbath, chemistry.f90:1002-1225): the host hands them over as it holds
// them (sqmc_gpu_set_heatbath_tables), the four-index ones in single precision as the reference stores them.
//
// Two accumulators of prob_heatbath_double are never initialised in the source (5728, 5742) and rely on -finit-local-zero:
// here they start from 0 on every call, the intended value (tests/golden/README_heatbath.md).  Arithmetic follows the
// reference statement by statement (no FMA contraction), so weights agree bit for bit with the CPU restatement.
#pragma once

// all orbital arguments 1-based; spin orbitals: 1..norb up, norb+1..2 norb dn (the reference's convention)
#define HB_ONE(i) hb.one[(i) - 1]
#define HB_TWO(i, j) hb.two[(size_t)((i) - 1) * (2 * n) + ((j) - 1)]
#define HB_IX3(i, j, k) ((((size_t)((i) - 1)) * n + ((j) - 1)) * n + ((k) - 1))
__device__ __forceinline__ long long hb_c2i(int i, int j) { return i > j ? ((long long)i * (i - 1)) / 2 + j : ((long long)j * (j - 1)) / 2 + i; }
__device__ __forceinline__ int hb_fold(int i, int n) { return i > n ? i - n : i; }
__device__ __forceinline__ long long hb_same_index(int n, int f1, int f2, int t1, int t2) {          // 1-based result (chemistry.f90:9154-9173)
  const int i = hb_fold(f1, n), j = hb_fold(f2, n), k = hb_fold(t1, n), l = hb_fold(t2, n);
  return (hb_c2i(i, j) - 1) * n * n + (long long)(k - 1) * n + l;
}
__device__ __forceinline__ long long hb_opp_index(int n, int f1, int f2, int t1, int t2) {
  const int i = hb_fold(f1, n), j = hb_fold(f2, n), k = hb_fold(t1, n), l = hb_fold(t2, n);
  return (long long)(i - 1) * n * n * n + (long long)(j - 1) * n * n + (long long)(k - 1) * n + l;
}
#define HB_HSAME(p_, k_) hb.htot_same[(size_t)((p_) - 1) * n + ((k_) - 1)]
__device__ __forceinline__ double hb_p_first_hole(const HbDev &hb, int f1, int f2, int t1) {
  const int n = hb.norb;
  if (f1 <= n) return (f2 <= n) ? hb.three_same[HB_IX3(f1, f2, t1)] : hb.three_opp[HB_IX3(f1, f2 - n, t1)];
  return (f2 <= n) ? hb.three_opp[HB_IX3(f1 - n, f2, t1 - n)] : hb.three_same[HB_IX3(f1 - n, f2 - n, t1 - n)];
}
__device__ __forceinline__ double hb_p_second_hole(const HbDev &hb, int f1, int f2, int t1, int t2) {
  const int n = hb.norb;
  if ((f1 <= n && f2 <= n) || (f1 > n && f2 > n)) return (double)hb.four_same[hb_same_index(n, f1, f2, t1, t2) - 1];
  return (double)hb.four_opp[hb_opp_index(n, f1, f2, t1, t2) - 1];
}
__device__ __forceinline__ double hb_Htot(const HbDev &hb, int f1, int f2, int t1) {
  const int n = hb.norb;
  if (f1 <= n) return (f2 <= n) ? HB_HSAME(hb_c2i(f1, f2), t1) : hb.htot_opp[HB_IX3(f1, f2 - n, t1)];
  return (f2 <= n) ? hb.htot_opp[HB_IX3(f1 - n, f2, t1 - n)] : HB_HSAME(hb_c2i(f1 - n, f2 - n), t1 - n);
}
__device__ __forceinline__ double hb_p_single_excit(double se, double sd) { double p = fabs(se) / (fabs(se) + sd); if (p > 0.5) p = 1.0; return p; }
__device__ __forceinline__ double hb_p_double_excit(double se, double sd) { double p = sd / (fabs(se) + sd); if (p < 0.5) p = 1.0; return p; }
__device__ __forceinline__ double hb_single_elem(const ChemTab &t, const double *__restrict__ ints, int n, u64 up, u64 dn, int f1, int t1) {
  if (f1 <= n) return h_single(t, ints, up, dn, (up & ~bit64(f1 - 1)) | bit64(t1 - 1), dn);
  return h_single(t, ints, up, dn, up, (dn & ~bit64(f1 - n - 1)) | bit64(t1 - n - 1));
}
// sample_alias (one random_int, one rannyu); J is 1-based
__device__ __forceinline__ int hb_alias_d(Rng &g, int K, const int *__restrict__ J, const double *__restrict__ q) { const int i = rng_int(g, K); return (rng_draw(g) < q[i - 1]) ? i : J[i - 1]; }
__device__ __forceinline__ int hb_alias_f(Rng &g, int K, const int *__restrict__ J, const float *__restrict__ q) { const int i = rng_int(g, K); return (rng_draw(g) < (double)q[i - 1]) ? i : J[i - 1]; }

// prob_heatbath_single, chemistry.f90:5580-5626
__device__ inline double hb_prob_single(const HbDev &hb, u64 iu, u64 id, int f1, int t1, double matrix_element, double normalization) {
  const int n = hb.norb; double pp = 0.0; const double sing_num = fabs(matrix_element);
  if (f1 <= n) {
    for (u64 b = iu; b; b &= b - 1) { const int e = ctz64(b) + 1; if (f1 == e) continue;
      pp = pp + HB_TWO(f1, e) * hb.three_same[HB_IX3(f1, e, t1)] * hb_p_single_excit(sing_num, HB_HSAME(hb_c2i(f1, e), t1)); }
    for (u64 b = id; b; b &= b - 1) { const int e = ctz64(b) + 1;
      pp = pp + HB_TWO(f1, e + n) * hb.three_opp[HB_IX3(f1, e, t1)] * hb_p_single_excit(sing_num, hb.htot_opp[HB_IX3(f1, e, t1)]); }
  } else {
    for (u64 b = iu; b; b &= b - 1) { const int e = ctz64(b) + 1;
      pp = pp + HB_TWO(f1, e) * hb.three_opp[HB_IX3(f1 - n, e, t1 - n)] * hb_p_single_excit(sing_num, hb.htot_opp[HB_IX3(f1 - n, e, t1 - n)]); }
    for (u64 b = id; b; b &= b - 1) { const int e = ctz64(b) + 1; if (f1 == e + n) continue;
      pp = pp + HB_TWO(f1, e + n) * hb.three_same[HB_IX3(f1 - n, e, t1 - n)] * hb_p_single_excit(sing_num, HB_HSAME(hb_c2i(f1 - n, e), t1 - n)); }
  }
  return pp * normalization;
}
// sums over the occupied spin orbitals that the probabilities of "electron f first" need (5699-5722 / 5493-5515): the total
// one-electron weight, f's own, and the pair weights of f with every other electron; up electrons first, each ascending
__device__ __forceinline__ void hb_first_electron_sums(const HbDev &hb, u64 iu, u64 id, int f, double &sum_one, double &one_f, double &sum_pair) {
  const int n = hb.norb;
  sum_one = 0.0; one_f = 0.0; sum_pair = 0.0;
  for (u64 b = iu; b; b &= b - 1) { const int e = ctz64(b) + 1; sum_one = sum_one + HB_ONE(e); if (f == e) { one_f = HB_ONE(e); continue; } sum_pair = sum_pair + HB_TWO(f, e); }
  for (u64 b = id; b; b &= b - 1) { const int e = ctz64(b) + 1; sum_one = sum_one + HB_ONE(e); if (f == e + n) { one_f = HB_ONE(e); continue; } sum_pair = sum_pair + HB_TWO(f, e + n); }
}
// prob_heatbath_double, 5693-5787 (accumulators from 0: see the header)
__device__ inline double hb_prob_double(const ChemTab &t, const double *__restrict__ ints, const HbDev &hb, u64 iu, u64 id, int f1, int f2, int t1, int t2,
                                        double matrix_element, double prob_1_then_2, bool same_spin) {
  const int n = hb.norb;
  double sum_one, one_2, sum_2_other;
  hb_first_electron_sums(hb, iu, id, f2, sum_one, one_2, sum_2_other);
  const double prob_of_1_and_2 = HB_TWO(f1, f2);
  const double prob_2_then_1 = one_2 / sum_one * prob_of_1_and_2 / sum_2_other;
  double term1, term2, term3, term4, sing_num;
  term1 = hb_p_first_hole(hb, f1, f2, t1) * hb_p_second_hole(hb, f1, f2, t1, t2);
  sing_num = fabs(matrix_element);
  term1 = term1 * hb_p_double_excit(sing_num, hb_Htot(hb, f1, f2, t1));
  if (same_spin) {
    term2 = hb_p_first_hole(hb, f1, f2, t2) * hb_p_second_hole(hb, f1, f2, t2, t1);
    sing_num = fabs(hb_single_elem(t, ints, n, iu, id, f1, t2));
    term2 = term2 * hb_p_double_excit(sing_num, hb_Htot(hb, f1, f2, t2));
    term3 = hb_p_first_hole(hb, f2, f1, t1) * hb_p_second_hole(hb, f2, f1, t1, t2);
    sing_num = fabs(hb_single_elem(t, ints, n, iu, id, f2, t1));
    term3 = term3 * hb_p_double_excit(sing_num, hb_Htot(hb, f2, f1, t1));
  } else { term2 = 0.0; term3 = 0.0; }
  term4 = hb_p_first_hole(hb, f2, f1, t2) * hb_p_second_hole(hb, f2, f1, t2, t1);
  sing_num = fabs(hb_single_elem(t, ints, n, iu, id, f2, t2));
  term4 = term4 * hb_p_double_excit(sing_num, hb_Htot(hb, f2, f1, t2));
  return prob_1_then_2 * (term1 + term2) + prob_2_then_1 * (term3 + term4);
}
// proposal_prob_efficient_heatbath, 5431-5549
__device__ inline double hb_proposal_prob(const ChemTab &t, const double *__restrict__ ints, const HbDev &hb, u64 iu, u64 id, u64 ju, u64 jd, int level, double off_diag_elem) {
  const int n = hb.norb;
  int f1 = 0, f2 = 0, t1 = 0, t2 = 0, excite_spin;
  if (level == 1) {
    if (iu == ju) { excite_spin = -1; f1 = ctz64(id & ~jd) + n + 1; t1 = ctz64(jd & ~id) + n + 1; }
    else { excite_spin = 1; f1 = ctz64(iu & ~ju) + 1; t1 = ctz64(ju & ~iu) + 1; }
  } else {
    if (iu == ju) {
      excite_spin = -1;
      u64 x = id & ~jd; f1 = ctz64(x) + 1; f2 = ctz64(x & ~bit64(f1 - 1)) + 1;
      x = jd & ~id; t1 = ctz64(x) + 1; t2 = ctz64(x & ~bit64(t1 - 1)) + 1;
      f1 += n; f2 += n; t1 += n; t2 += n;
    } else if (id == jd) {
      excite_spin = 1;
      u64 x = iu & ~ju; f1 = ctz64(x) + 1; f2 = ctz64(x & ~bit64(f1 - 1)) + 1;
      x = ju & ~iu; t1 = ctz64(x) + 1; t2 = ctz64(x & ~bit64(t1 - 1)) + 1;
    } else {
      excite_spin = 0;
      f1 = ctz64(iu & ~ju) + 1; f2 = ctz64(id & ~jd) + n + 1; t1 = ctz64(ju & ~iu) + 1; t2 = ctz64(jd & ~id) + n + 1;
    }
  }
  double sum_one, one_1, sum_1_other;
  hb_first_electron_sums(hb, iu, id, f1, sum_one, one_1, sum_1_other);
  if (level == 1) {
    const double normalization = one_1 / sum_one / sum_1_other;
    return hb_prob_single(hb, iu, id, f1, t1, off_diag_elem, normalization);
  }
  const double single_elem = hb_single_elem(t, ints, n, iu, id, f1, t1);
  const double prob_1_then_2 = one_1 / sum_one * HB_TWO(f1, f2) / sum_1_other;
  return hb_prob_double(t, ints, hb, iu, id, f1, f2, t1, t2, single_elem, prob_1_then_2, excite_spin != 0);
}
// apply_time_reversal_symmetry, 5350-5427
__device__ inline void hb_time_reversal(const ChemTab &t, const double *__restrict__ ints, const HbDev &hb, u64 iu, u64 id, u64 &ju, u64 &jd, double &me, double &pp) {
  const double sqrt2 = sqrt(2.0);
  if ((ju == iu && jd == id) || (jd == iu && ju == id)) { me = 0.0; return; }
  const double norm_i = (iu == id) ? sqrt2 : 1.0;
  double norm_j = 1.0;
  if (ju == jd) {
    if (t.z == 1) norm_j = sqrt2; else return;
    me = (norm_j / norm_i) * me;
  } else {
    const int lev = excitation_level(iu, id, jd, ju);
    if (lev >= 0) {
      const double me2 = h_level(t, ints, iu, id, jd, ju, lev);
      if (fabs(me2) > 1.0e-10) {
        const double ps = hb_proposal_prob(t, ints, hb, iu, id, jd, ju, lev, me2);
        pp = pp + ps;
        me = (norm_j / norm_i) * (me + t.z * me2);
      } else me = (norm_j / norm_i) * me;
    } else me = (norm_j / norm_i) * (me);
  }
  if (ju > jd) { const u64 x = ju; ju = jd; jd = x; me = me * t.z; }
}
// first index (1-based, electrons in the order up ascending then dn ascending) whose cumulative normalised weight exceeds r:
// what sample_discrete_distribution's binary search returns on the non-decreasing c_probs.  weight(e, spin-orbital) comes from W.
#define HB_SAMPLE_ELECTRON(WEIGHT_UP, WEIGHT_DN, TOTAL, R, OUT_ELEC, OUT_PROB)                               \
  do { double run_ = 0.0; int found_ = 0; OUT_ELEC = 0; OUT_PROB = 0.0; double lastw_ = 0.0; int laste_ = 0; \
    for (u64 b_ = iu; b_; b_ &= b_ - 1) { const int e = ctz64(b_) + 1; const int so = e; const double w_ = (WEIGHT_UP); run_ = run_ + w_; lastw_ = w_; laste_ = so; \
      if (!found_ && (R) < run_ / (TOTAL)) { found_ = 1; OUT_ELEC = so; OUT_PROB = w_ / (TOTAL); } }        \
    for (u64 b_ = id; b_; b_ &= b_ - 1) { const int e = ctz64(b_) + 1; const int so = e + n; const double w_ = (WEIGHT_DN); run_ = run_ + w_; lastw_ = w_; laste_ = so; \
      if (!found_ && (R) < run_ / (TOTAL)) { found_ = 1; OUT_ELEC = so; OUT_PROB = w_ / (TOTAL); } }        \
    if (!found_) { OUT_ELEC = laste_; OUT_PROB = lastw_ / (TOTAL); } } while (0)

// off_diagonal_move_chem_efficient_heatbath, 5086-5347.  Returns the number of slots that hold a move (0, 1 or 2); a slot
// without one has weight 0.  wj = -tau H_ij / p(i -> j) already.
__device__ inline int propose_heatbath(const ChemTab &t, const double *__restrict__ ints, const HbDev &hb, Rng &g, double tau, u64 iu, u64 id,
                                       u64 ju[2], u64 jd[2], double wj[2]) {
  const int n = hb.norb;
  ju[0] = ju[1] = iu; jd[0] = jd[1] = id; wj[0] = wj[1] = 0.0;
  // first electron from one_orbital_probabilities, second from two_orbital_probabilities(first, .)
  double tot1 = 0.0;
  for (u64 b = iu; b; b &= b - 1) tot1 = tot1 + HB_ONE(ctz64(b) + 1);
  for (u64 b = id; b; b &= b - 1) tot1 = tot1 + HB_ONE(ctz64(b) + 1);
  int f1, f2; double e1_prob_sav, e2p;
  { const double r = rng_draw(g); HB_SAMPLE_ELECTRON(HB_ONE(e), HB_ONE(e), tot1, r, f1, e1_prob_sav); }
  double proposal_prob = e1_prob_sav;
  double c_e2_sav = 0.0;
  for (u64 b = iu; b; b &= b - 1) { const int e = ctz64(b) + 1; c_e2_sav = c_e2_sav + ((e == f1) ? 0.0 : HB_TWO(f1, e)); }
  for (u64 b = id; b; b &= b - 1) { const int e = ctz64(b) + 1; c_e2_sav = c_e2_sav + ((e + n == f1) ? 0.0 : HB_TWO(f1, e + n)); }
  { const double r = rng_draw(g); HB_SAMPLE_ELECTRON(((so == f1) ? 0.0 : HB_TWO(f1, so)), ((so == f1) ? 0.0 : HB_TWO(f1, so)), c_e2_sav, r, f2, e2p); }
  proposal_prob = proposal_prob * e2p;
  // choose_first_hole, 9278-9304
  int t1, excite_spin;
  if (f1 <= n && f2 <= n) { excite_spin = 1; t1 = hb_alias_d(g, n, hb.j3_same + HB_IX3(f1, f2, 1), hb.q3_same + HB_IX3(f1, f2, 1)); }
  else if (f1 > n && f2 > n) { excite_spin = -1; t1 = hb_alias_d(g, n, hb.j3_same + HB_IX3(f1 - n, f2 - n, 1), hb.q3_same + HB_IX3(f1 - n, f2 - n, 1)) + n; }
  else {
    excite_spin = 0;
    if (f1 > n) t1 = hb_alias_d(g, n, hb.j3_opp + HB_IX3(f1 - n, f2, 1), hb.q3_opp + HB_IX3(f1 - n, f2, 1)) + n;
    else t1 = hb_alias_d(g, n, hb.j3_opp + HB_IX3(f1, f2 - n, 1), hb.q3_opp + HB_IX3(f1, f2 - n, 1));
  }
#define HB_OCC(T) (((T) <= n) ? (int)((iu >> ((T) - 1)) & 1) : (int)((id >> ((T) - n - 1)) & 1))
  if (HB_OCC(t1)) return 0;
  double matrix_element = hb_single_elem(t, ints, n, iu, id, f1, t1);
  const bool same_spin = (excite_spin != 0);
  const double sing_num = fabs(matrix_element), sing_den = sing_num + hb_Htot(hb, f1, f2, t1);
#define HB_SECOND_HOLE(T2) do {                                                                            \
    if (!same_spin) { const long long b_ = hb_opp_index(n, f1, f2, t1, 1) - 1;                             \
      T2 = hb_alias_f(g, n, hb.j4_opp + b_, hb.q4_opp + b_); if (f1 <= n) T2 += n; }                       \
    else { const long long b_ = hb_same_index(n, f1, f2, t1, 1) - 1;                                       \
      T2 = hb_alias_f(g, n, hb.j4_same + b_, hb.q4_same + b_); if (f1 > n) T2 += n; } } while (0)
#define HB_MAKE_SINGLE(K) do { if (f1 <= n) { ju[K] = (iu & ~bit64(f1 - 1)) | bit64(t1 - 1); jd[K] = id; }  \
                               else { jd[K] = (id & ~bit64(f1 - n - 1)) | bit64(t1 - n - 1); ju[K] = iu; } } while (0)
#define HB_MAKE_DOUBLE(K, T2) do { u64 u_ = iu, d_ = id;                                                   \
    if (f1 <= n) u_ &= ~bit64(f1 - 1); else d_ &= ~bit64(f1 - n - 1);                                      \
    if (f2 <= n) u_ &= ~bit64(f2 - 1); else d_ &= ~bit64(f2 - n - 1);                                      \
    if (t1 <= n) u_ |= bit64(t1 - 1); else d_ |= bit64(t1 - n - 1);                                        \
    if ((T2) <= n) u_ |= bit64((T2) - 1); else d_ |= bit64((T2) - n - 1);                                  \
    ju[K] = u_; jd[K] = d_; } while (0)
  int n_new;
  if (sing_num > (sing_den - sing_num)) {              // a single larger than all its doubles together: propose both (5266-5302)
    const double prob_1_then_2 = proposal_prob;
    n_new = 2;
    const double normalization = e1_prob_sav / c_e2_sav;
    HB_MAKE_SINGLE(0);
    double pp = hb_prob_single(hb, iu, id, f1, t1, matrix_element, normalization);
    if (t.time_sym) hb_time_reversal(t, ints, hb, iu, id, ju[0], jd[0], matrix_element, pp);      // updates matrix_element in place, as the source does
    wj[0] = -tau * matrix_element / pp;
    if (hb_Htot(hb, f1, f2, t1) == 0.0) return 1;
    int t2; HB_SECOND_HOLE(t2);
    if (HB_OCC(t2)) return 1;
    HB_MAKE_DOUBLE(1, t2);
    pp = hb_prob_double(t, ints, hb, iu, id, f1, f2, t1, t2, matrix_element, prob_1_then_2, same_spin);
    double me = h_double(t, ints, iu, id, ju[1], jd[1]);
    if (t.time_sym) hb_time_reversal(t, ints, hb, iu, id, ju[1], jd[1], me, pp);
    wj[1] = -tau * me / pp;
  } else {
    n_new = 1;
    const double p_single = hb_p_single_excit(sing_num, sing_den - sing_num);
    double pp, me;
    if (rng_draw(g) < p_single) {
      const double normalization = e1_prob_sav / c_e2_sav;
      HB_MAKE_SINGLE(0);
      pp = hb_prob_single(hb, iu, id, f1, t1, matrix_element, normalization); me = matrix_element;
    } else {
      int t2; HB_SECOND_HOLE(t2);
      if (HB_OCC(t2)) return 0;
      const double prob_1_then_2 = proposal_prob;
      HB_MAKE_DOUBLE(0, t2);
      pp = hb_prob_double(t, ints, hb, iu, id, f1, f2, t1, t2, matrix_element, prob_1_then_2, same_spin);
      me = h_double(t, ints, iu, id, ju[0], jd[0]);
    }
    if (t.time_sym) hb_time_reversal(t, ints, hb, iu, id, ju[0], jd[0], me, pp);
    wj[0] = -tau * me / pp;
  }
  return n_new;
}
