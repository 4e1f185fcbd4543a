// hii_group.h -- the diagonal element of one determinant by a group of lanes, terms in the reference's order (hamiltonian_chem's
// diagonal branch, chemistry.f90:8605-8700 region as restated in h_diag, chem_device.h).  Textually included by sqmc_gpu.hip.
#pragma once
#define BK_AT 512                      // threads of k_anneal_bucket: one block per CU, so the block itself has to keep the memory pipes busy
// ---- H_ii of one determinant by 16 lanes (chemistry, no time symmetry).  h_diag (chem_device.h) is three running sums -- one-body,
// exchange, direct -- of ~50 integrals; a lane on its own pays one L2 round trip per group of four.  Here every term of a sum has its
// place in the reference's order (closed form from the electron indices), the 16 lanes fetch all terms at once into LDS, and one
// lane per sum adds them up in that order: the same value bit for bit, one round trip instead of ~25.
// HG lanes per determinant: 8 (64 determinants per pass) when its terms fit 56 doubles -- up to 4 + 4 electrons --, else 16
#define BK_HG_TERMS(HG) (BK_CAP_T / (BK_AT / (HG)))            // LDS doubles per group: the groups share the weight array (112 at HG = 16)
#define BK_HG_TASKS 6                                           // tasks per lane at the cap of HG = 16; 12 at HG = 8
__device__ __forceinline__ int bk_nth_orb(u64 x, int n) { for (int k = 0; k < n; k++) x &= x - 1; return ctz64(x); }     // 0-based orbital of the n-th electron
__host__ __device__ __forceinline__ int bk_hii_terms_of(int nup, int ndn) { return (nup + ndn) + (nup * (nup - 1) + ndn * (ndn - 1)) + nup * ndn; }      // doubles a group needs for one determinant
__device__ __forceinline__ int bk_hii_terms(const ChemTab &t) { return bk_hii_terms_of(t.nup, t.ndn); }
__device__ __forceinline__ int bk_hii_group_lanes(const ChemTab &t) {          // 0: no group form for this system
  const int nup = t.nup, ndn = t.ndn, nuu = nup * (nup - 1) / 2, ndd = ndn * (ndn - 1) / 2, nud = nup * ndn;
  const int terms = (nup + ndn) + (nuu + ndd) + (nuu + nud + ndd), tasks = (nup + ndn) + nuu + ndd + nud;
  if (t.sys_type != 0 || t.time_sym) return 0;
  if (terms <= BK_HG_TERMS(8) && tasks <= 8 * 2 * BK_HG_TASKS) return 8;
  if (terms <= BK_HG_TERMS(16) && tasks <= 16 * BK_HG_TASKS) return 16;
  return 0;
}
// all threads of the block call this together (two barriers inside); sg: the group's BK_HG_TERMS(HG) doubles; g: lane inside the group
template <int HG>
__device__ __forceinline__ double bk_hii_group(const ChemTab &t, const double *__restrict__ ints, u64 up, u64 dn, bool valid, double *sg, int g) {
  constexpr int NTASK = BK_HG_TASKS * 16 / HG;
  const int nup = t.nup, ndn = t.ndn, n1 = t.norb + 1;
  const bool same = (dn == up);
  const int nuu = nup * (nup - 1) / 2, ndd = ndn * (ndn - 1) / 2, nud = nup * ndn;
  const int L_e1 = same ? nup : nup + ndn, L_ex = nuu + (same ? 0 : ndd), L_di = nuu + nud + ndd;
  const int o_ex = L_e1, o_di = L_e1 + L_ex;
  const int ntask = valid ? (nup + ndn) + nuu + ndd + nud : 0;
  int p0[NTASK], p1[NTASK], x0[NTASK], x1[NTASK];
#pragma unroll
  for (int m = 0; m < NTASK; m++) {
    int k = g + HG * m;
    p0[m] = -1; p1[m] = -1; x0[m] = 0; x1[m] = 0;
    if (k >= ntask) continue;
    if (k < nup) {                                   // one-body, up electron k
      const int i = bk_nth_orb(up, k) + 1;
      p0[m] = k; x0[m] = integral_index(t, i, i, n1, n1);
    } else if (k < nup + ndn) {                      // one-body, dn electron
      const int b = k - nup;
      if (!same) { const int i = bk_nth_orb(dn, b) + 1; p0[m] = nup + b; x0[m] = integral_index(t, i, i, n1, n1); }
    } else if (k < nup + ndn + nuu) {                // up-up pair (a < a2): an exchange and a direct term
      int r = k - nup - ndn, a = 0;
      while (r >= nup - 1 - a) { r -= nup - 1 - a; a++; }
      const int a2 = a + 1 + r, i0 = bk_nth_orb(up, a), j0 = bk_nth_orb(up, a2);
      const int Bc = __popcll(dn & ((1ull << i0) - 1ull));
      const int base = a * (nup - 1 + ndn) - a * (a - 1) / 2 + Bc * (ndn - 1) - Bc * (Bc - 1) / 2;
      p0[m] = o_ex + a * (nup - 1) - a * (a - 1) / 2 + (a2 - a - 1); x0[m] = integral_index(t, i0 + 1, j0 + 1, j0 + 1, i0 + 1);
      p1[m] = o_di + base + (a2 - a - 1);                            x1[m] = integral_index(t, i0 + 1, i0 + 1, j0 + 1, j0 + 1);
    } else if (k < nup + ndn + nuu + ndd) {          // dn-dn pair (b < b2): direct always, exchange unless the strings are equal
      int r = k - nup - ndn - nuu, b = 0;
      while (r >= ndn - 1 - b) { r -= ndn - 1 - b; b++; }
      const int b2 = b + 1 + r, i0 = bk_nth_orb(dn, b), j0 = bk_nth_orb(dn, b2);
      const int A = __popcll(up & ((2ull << i0) - 1ull));              // up electrons at orbitals <= i0 come first
      const int base = A * (nup - 1 + ndn) - A * (A - 1) / 2 + b * (ndn - 1) - b * (b - 1) / 2;
      if (!same) { p0[m] = o_ex + nuu + b * (ndn - 1) - b * (b - 1) / 2 + (b2 - b - 1); x0[m] = integral_index(t, i0 + 1, j0 + 1, j0 + 1, i0 + 1); }
      p1[m] = o_di + base + (b2 - b - 1); x1[m] = integral_index(t, i0 + 1, i0 + 1, j0 + 1, j0 + 1);
    } else {                                         // up-dn pair: one direct term
      const int r = k - nup - ndn - nuu - ndd, a = r / ndn, b = r - a * ndn;
      const int i0 = bk_nth_orb(up, a), j0 = bk_nth_orb(dn, b);
      const int Bc = __popcll(dn & ((1ull << i0) - 1ull));
      const int base = a * (nup - 1 + ndn) - a * (a - 1) / 2 + Bc * (ndn - 1) - Bc * (Bc - 1) / 2;
      p1[m] = o_di + base + (nup - 1 - a) + b; x1[m] = integral_index(t, i0 + 1, i0 + 1, j0 + 1, j0 + 1);
    }
  }
  double v0[NTASK], v1[NTASK];
#pragma unroll
  for (int m = 0; m < NTASK; m++) { v0[m] = (p0[m] >= 0) ? ints[x0[m]] : 0.0; v1[m] = (p1[m] >= 0) ? ints[x1[m]] : 0.0; }
#pragma unroll
  for (int m = 0; m < NTASK; m++) { if (p0[m] >= 0) sg[p0[m]] = v0[m]; if (p1[m] >= 0) sg[p1[m]] = v1[m]; }
  __syncthreads();
  double acc = 0.0;
  if (valid) {
    // one lane per sum, terms in the reference's order; eight LDS reads are requested before the first of them is added (the
    // additions stay sequential, the reads need not wait for one another: 28 dependent read-add pairs were 1.7 of this phase's 7 us)
    const int o = (g == 0) ? 0 : (g == 1 ? o_ex : o_di), L = (g == 0) ? L_e1 : (g == 1 ? L_ex : (g == 2 ? L_di : 0));
    const double sgn = (g == 1) ? -1.0 : 1.0;
    int q = 0;
    for (; q + 8 <= L; q += 8) {
      double a_[8];
#pragma unroll
      for (int z = 0; z < 8; z++) a_[z] = sg[o + q + z];
#pragma unroll
      for (int z = 0; z < 8; z++) acc = acc + sgn * a_[z];
    }
    for (; q < L; q++) acc = acc + sgn * sg[o + q];
    if (same && g < 2) acc = acc * 2.0;
  }
  const double e1 = __shfl(acc, 0, HG), ex = __shfl(acc, 1, HG), di = __shfl(acc, 2, HG);
  __syncthreads();
  return e1 + (ex + di) + t.nuclear;
}

// ---- the same for the electron gas (hamiltonian_heg's diagonal, heg.f90:845-1011 as restated in h_heg): 14 kinetic terms and 42 pair
// terms 4 pi / |k_a - k_b|^2 for 7 + 7 electrons, each a chain of subtractions, a sum of squares and an fp64 divide -- one lane on its
// own works ~2,000 dependent instructions per determinant.  Here the terms are spread over the HG lanes of a group, parked in LDS in the
// reference's order, and two lanes add them up in that order: the same value bit for bit.
__device__ __forceinline__ int heg_hii_terms(const ChemTab &t) { return (t.nup + t.ndn) + t.nup * (t.nup - 1) / 2 + t.ndn * (t.ndn - 1) / 2; }
__device__ __forceinline__ bool heg_hii_group_ok(const ChemTab &t) { return t.sys_type == 1 && heg_hii_terms(t) <= BK_HG_TERMS(16); }
template <int HG>
__device__ __forceinline__ double heg_hii_group(const ChemTab &t, u64 up, u64 dn, bool valid, double *sg, int g) {
  const int nup = t.nup, ndn = t.ndn, nuu = nup * (nup - 1) / 2, ndd = ndn * (ndn - 1) / 2;
  const int L_kin = nup + ndn, L_pot = nuu + ndd, ntask = valid ? L_kin + L_pot : 0;
  for (int k = g; k < ntask; k += HG) {
    double v;
    if (k < L_kin) {
      const int o = (k < nup) ? bk_nth_orb(up, k) : bk_nth_orb(dn, k - nup);
      v = heg_sumsq(t, t.kvec[o + 1]) * 0.5;
    } else {
      int r = k - L_kin; u64 det = up; int n = nup;
      if (r >= nuu) { r -= nuu; det = dn; n = ndn; }
      int a = 0;
      while (r >= n - 1 - a) { r -= n - 1 - a; a++; }            // pair (a, a + 1 + r) in the order of the reference's double loop
      v = heg_inv_k2(t, bk_nth_orb(det, a) + 1, bk_nth_orb(det, a + 1 + r) + 1);
    }
    sg[k] = v;
  }
  __syncthreads();
  double acc = 0.0;
  if (valid && g < 2) {
    const int o = g ? L_kin : 0, L = g ? L_pot : L_kin;
    for (int q = 0; q < L; q++) acc = acc + sg[o + q];
  }
  const double me = __shfl(acc, 0, HG), pot = __shfl(acc, 1, HG);
  __syncthreads();
  const double Lc = t.length_cell;
  return me - pot / (Lc * Lc * Lc);
}

