// chem_device.h -- device-side chemistry: integral lookup, Slater-Condon matrix elements,
// the 48-bit RNGs and the symmetry-aware uniform proposal.  gfx950 only.
//
// Arithmetic follows the reference operation by operation (no FMA contraction: the
// library is compiled with -ffp-contract=off) so that spawn weights agree bit for bit
// with a gfortran -O2 build of the reference:
//   integral_index / integral_value   chemistry.f90:9106-9134, 1234-1256
//   one_body, two_body ("usual way")  chemistry.f90:1382-1437, 1773-1838
//   one/two_body_single, _double      chemistry.f90:1439-1478, 1845-2001
//   permutation_factor(2)             tools.f90:1294-1396
//   off_diagonal_move_chem            chemistry.f90:4237-5084 (time_sym = .false.)
//   rannyu / random_int               rannyu.f90:54-74, tools.f90:129-147
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef unsigned long long u64;

#define SQ_MAXORB 64
#define SQ_MAXSYM 8

// Small tables every chemistry kernel stages into LDS (divergent per-lane lookups).
struct ChemTab {
  int norb, nup, ndn, ncore, nelec, time_sym, z, ngroup;
  u64 orb_mask;                        // bits 0..norb-1
  u64 sym_mask[SQ_MAXSYM + 1];         // orbitals of each irrep (which_orb_by_sym as a bitmask)
  unsigned char prod[SQ_MAXSYM + 1][SQ_MAXSYM + 1];
  unsigned char orbsym[SQ_MAXORB + 1]; // 1-based
  double nuclear;
  // homogeneous electron gas (sys_type 1): plane-wave orbitals k_vectors(:, i), heg.f90:643-749
  int sys_type, n_dim;                 // 0 = 'chem', 1 = 'heg', 2 = 'hubbard2' (real-space Hubbard, hubbard.f90)
  double length_cell;
  double hub_t, hub_U;                 // hubbard2: hopping and on-site repulsion
  union {
    double kvec[SQ_MAXORB + 1][3];     // heg, 1-based
    unsigned char hub_nbr[SQ_MAXORB + 1][4];   // hubbard2: get_nbr(site, LEFT/RIGHT/UP/DOWN), 0 = not allowed (more_tools.f90:223-355)
  };
  signed char krel[SQ_MAXORB + 1][3];  // the same in units of 2 pi / L (k_vectors_rel): exact momentum bookkeeping
  int heg_nmax;                        // max |krel component|
  int c2_stride, c2_pad;               // combine_2 is stored packed: c2[i*c2_stride + j], 1-based.  heg: c2_pad = 1 when the bytes of c2 hold the
                                       // plane-wave lookup (krel + heg_nmax, base 2 heg_nmax + 1) -> orbital, 0 = no such plane wave (heg_lut)
  unsigned short c2[(SQ_MAXORB + 2) * (SQ_MAXORB + 2)];   // only the first c2_stride^2 entries are used/staged
};

// efficient heat-bath proposal tables on the device (heatbath_device.h): 0-based, last index fastest
struct HbDev {
  int on, norb, npairs;
  const double *one;                     // [norb]                      one_orbital_probabilities
  const double *two;                     // [2 norb][2 norb]            two_orbital_probabilities
  const double *three_same, *three_opp;  // [norb][norb][norb], normalised over the last index
  const int *j3_same, *j3_opp; const double *q3_same, *q3_opp;          // alias tables of the first hole (J 1-based as in the reference)
  const float *four_same, *four_opp; const int *j4_same, *j4_opp; const float *q4_same, *q4_opp;     // [same_index - 1] / [opposite_index - 1], single precision as the reference stores them
  const double *htot_same;               // [npairs][norb]
  const double *htot_opp;                // [norb][norb][norb]
};

#define SQ_BINOM_STRIDE 68              // C(c, i) for every i <= 64, + 3 entries the 4-wide rounds of colex_rank may touch
// hf_to_psit = .true. (do_walk.f90:378-386, psit_kernels.h): the walker list is [C(T), fixed | survivors outside C(T)], which is
// the list sorted by  key' = rank + (determinant outside C(T) ? koff : 0),  koff = number of determinants of the space
struct PsitDev {
  u64 koff;                             // 0: off
  const u64 *hkey; u64 hmask;           // C(T) membership: the open-addressed hash of sqmc_gpu_set_ct_table on the determinant's rank
  u64 first_up, first_dn;               // dets_up/dn_psi_t(1) = the first state: no stochastic move from it (3574) or onto it (3676, 7642)
  long long n_ct;
};
struct ChemDev {                        // pointers into HBM, passed by value
  const ChemTab *tab; int tab_words;
  PsitDev ps;
  const u64 *binom;                     // C(c, i) at [c*SQ_BINOM_STRIDE + i], c < 64, i < SQ_BINOM_STRIDE (0 where i > c)
  u64 n_dn_strings;                     // C(norb, ndn)
  const double *integrals;              // 1-based packed
  // HCI heat-bath table (chemistry.f90:900-993)
  const int *hb_r, *hb_s; const double *hb_absH; const long long *pq_ind; const int *pq_count;
  double max_double;
  HbDev hb;                             // proposal_method fast_heatbath (hb.on) instead of uniform2
};

// 32-bit words of a ChemTab that are in use: header + the used part of combine_2 only (norb=26:
// 2.3 KB instead of 9.4 KB per block).  The host computes it once (ChemDev::tab_words) so that a
// kernel can issue the staging loads without first reading c2_stride from HBM.
__host__ __device__ __forceinline__ int tab_words_used(int c2_stride) {
  return (int)((offsetof(ChemTab, c2) + (size_t)(c2_stride * c2_stride) * sizeof(unsigned short) + sizeof(int) - 1) / sizeof(int));
}
__device__ __forceinline__ void stage_tab(ChemTab *dst, const ChemTab *src, int n) {
  const int *s = reinterpret_cast<const int *>(src);
  int *d = reinterpret_cast<int *>(dst);
  // the first four words per thread are loaded before any is stored: one HBM/L2 round trip for
  // tables up to 4*blockDim words (norb <= 42 with 256 threads), not one per loop iteration
  const int i0 = threadIdx.x, nt = blockDim.x;
  const int r0 = (i0 < n) ? s[i0] : 0, r1 = (i0 + nt < n) ? s[i0 + nt] : 0;
  const int r2 = (i0 + 2 * nt < n) ? s[i0 + 2 * nt] : 0, r3 = (i0 + 3 * nt < n) ? s[i0 + 3 * nt] : 0;
  if (i0 < n) d[i0] = r0;
  if (i0 + nt < n) d[i0 + nt] = r1;
  if (i0 + 2 * nt < n) d[i0 + 2 * nt] = r2;
  if (i0 + 3 * nt < n) d[i0 + 3 * nt] = r3;
  for (int i = i0 + 4 * nt; i < n; i += nt) d[i] = s[i];
  __syncthreads();
}

__device__ __forceinline__ int ctz64(u64 x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int popc64(u64 x) { return __popcll(x); }
__device__ __forceinline__ u64 bit64(int k) { return 1ull << k; }
__device__ __forceinline__ u64 maskr64(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

// Sort key of a determinant: colex rank of the up string times C(norb,ndn) plus the rank of
// the dn string.  For fixed electron numbers the colex rank orders bit strings exactly like
// their integer values, so keys sort walkers by (up, then dn) as the reference does
// (do_walk.f90:5411-5475) in ceil(log2(C(norb,nup)*C(norb,ndn))) bits instead of 2*norb
// (28 instead of 52 for C2 cc-pVDZ): fewer radix passes.
__device__ __forceinline__ u64 colex_rank(const u64 *__restrict__ binom, u64 det) {
  u64 r = 0; int i = 1;
  while (det) {     // four electrons per round so that the four table loads are in flight together; C(0,i>=1) = 0 pads
    const u64 d1 = det & (det - 1), d2 = d1 & (d1 - 1), d3 = d2 & (d2 - 1);
    const int c0 = ctz64(det), c1 = d1 ? ctz64(d1) : 0, c2 = d2 ? ctz64(d2) : 0, c3 = d3 ? ctz64(d3) : 0;
    const u64 b0 = binom[c0 * SQ_BINOM_STRIDE + i], b1 = binom[c1 * SQ_BINOM_STRIDE + i + 1];
    const u64 b2 = binom[c2 * SQ_BINOM_STRIDE + i + 2], b3 = binom[c3 * SQ_BINOM_STRIDE + i + 3];
    r += (b0 + b1) + (b2 + b3);
    det = d3 & (d3 - 1); i += 4;
  }
  return r;
}
__device__ __forceinline__ u64 det_key(const ChemDev &dev, u64 up, u64 dn) {
  // both strings advance together: eight table loads per round trip
  const u64 *__restrict__ binom = dev.binom;
  u64 ru = 0, rd = 0; int i = 1;
  while (up | dn) {
    const u64 u1 = up & (up - 1), u2 = u1 & (u1 - 1), u3 = u2 & (u2 - 1);
    const u64 d1 = dn & (dn - 1), d2 = d1 & (d1 - 1), d3 = d2 & (d2 - 1);
    const int a0 = up ? ctz64(up) : 0, a1 = u1 ? ctz64(u1) : 0, a2 = u2 ? ctz64(u2) : 0, a3 = u3 ? ctz64(u3) : 0;
    const int b0 = dn ? ctz64(dn) : 0, b1 = d1 ? ctz64(d1) : 0, b2 = d2 ? ctz64(d2) : 0, b3 = d3 ? ctz64(d3) : 0;
    const u64 x0 = binom[a0 * SQ_BINOM_STRIDE + i], x1 = binom[a1 * SQ_BINOM_STRIDE + i + 1], x2 = binom[a2 * SQ_BINOM_STRIDE + i + 2], x3 = binom[a3 * SQ_BINOM_STRIDE + i + 3];
    const u64 y0 = binom[b0 * SQ_BINOM_STRIDE + i], y1 = binom[b1 * SQ_BINOM_STRIDE + i + 1], y2 = binom[b2 * SQ_BINOM_STRIDE + i + 2], y3 = binom[b3 * SQ_BINOM_STRIDE + i + 3];
    ru += (x0 + x1) + (x2 + x3); rd += (y0 + y1) + (y2 + y3);
    up = u3 & (u3 - 1); dn = d3 & (d3 - 1); i += 4;
  }
  return ru * dev.n_dn_strings + rd;
}

// ----------------------------------------------------------------------------- RNG
#define SQ_LCG_MULT 34522712143931ull          // 11^13 = 502*8^12 + 1521*8^8 + 4071*8^4 + 2107
#define SQ_MASK48 0xFFFFFFFFFFFFull
#define SQ_GOLDEN 0x9E3779B97F4A7C15ull
struct Rng {
  int mode;   // 0 = rannyu LCG state in x (48 bits); 1 = splitmix64 counter state in x
  u64 x;
};
__host__ __device__ __forceinline__ u64 sq_mix64(u64 v) {
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 27; v *= 0x94D049BB133111EBull; v ^= v >> 31;
  return v;
}
// stream key for the COUNTER discipline: (seed, step, stage, entity index).  `seed` is the MIXED input seed
// (sqmc_gpu_ctx::seed64 = sq_mix64(48-bit seed)): per-rank seeds differ only in their low bits (do_walk.f90:234),
// which is where step and stage enter; mixed first, the streams of different ranks never coincide a step apart.
__host__ __device__ __forceinline__ u64 sq_counter_key(u64 seed, u64 step, int stage, u64 idx) {
  return sq_mix64(sq_mix64(seed ^ (step * 4ull + (u64)stage)) + idx);
}
__device__ __forceinline__ double rng_draw(Rng &g) {
  u64 k;
  if (g.mode == 0) { g.x = (g.x * SQ_LCG_MULT) & SQ_MASK48; k = g.x; }
  else { g.x += SQ_GOLDEN; k = sq_mix64(g.x) >> 16; }
  return (double)k * 3.552713678800500929355621337890625e-15;   // 2^-48, exact
}
__device__ __forceinline__ int rng_int(Rng &g, int n) { return (int)((double)n * rng_draw(g)) + 1; }
// LCG skip-ahead: state * M^k mod 2^48
__host__ __device__ __forceinline__ u64 lcg_skip(u64 x, u64 k) {
  u64 m = SQ_LCG_MULT;
  while (k) { if (k & 1) x = (x * m) & SQ_MASK48; m = (m * m) & SQ_MASK48; k >>= 1; }
  return x;
}

__device__ __forceinline__ int kth_set(u64 bits, int k) {   // 1-based orbital of the k-th set bit
  for (int i = 1; i < k; i++) bits &= bits - 1;
  return ctz64(bits) + 1;
}
// the same by halving (no data-dependent loop: lanes of a wave pick very different k among the holes)
__device__ __forceinline__ int kth_set_wide(u64 bits, int k) {
  unsigned int x = (unsigned int)bits; int pos = 0;
  int pc = __popc(x);
  if (k > pc) { k -= pc; x = (unsigned int)(bits >> 32); pos = 32; }
  pc = __popc(x & 0xFFFFu); if (k > pc) { k -= pc; x >>= 16; pos += 16; }
  pc = __popc(x & 0xFFu);   if (k > pc) { k -= pc; x >>= 8;  pos += 8; }
  pc = __popc(x & 0xFu);    if (k > pc) { k -= pc; x >>= 4;  pos += 4; }
  pc = __popc(x & 0x3u);    if (k > pc) { k -= pc; x >>= 2;  pos += 2; }
  pc = (int)(x & 1u);       if (k > pc) { pos += 1; }
  return pos + 1;
}

// ------------------------------------------------------------------------ integrals
__device__ __forceinline__ int integral_index(const ChemTab &t, int i, int j, int k, int l) {
  int a = t.c2[i * t.c2_stride + j], b = t.c2[k * t.c2_stride + l];
  return (a > b) ? (a * (a - 1)) / 2 + b : (b * (b - 1)) / 2 + a;
}
#define IVAL(p, q, r, s) (ints[integral_index(t, (p), (q), (r), (s))])

__device__ __forceinline__ int permutation_factor(u64 a, u64 b) {
  u64 diff = (a > b) ? (a & (a - b)) : (b & (b - a));
  return (popc64(diff) & 1) ? -1 : 1;
}
__device__ __forceinline__ void permutation_factor2(u64 di, u64 dj, int &gamma, int &i1, int &i2, int &j1, int &j2) {
  u64 d = di & ~dj;
  i1 = ctz64(d); i2 = ctz64(d & ~bit64(i1));
  d = dj & ~di;
  j1 = ctz64(d); j2 = ctz64(d & ~bit64(j1));
  d = di & dj & ((maskr64(i1) ^ maskr64(j1)) ^ (maskr64(i2) ^ maskr64(j2)));
  gamma = (popc64(d) & 1) ? -1 : 1;
}
__device__ __forceinline__ int excitation_level(u64 iu, u64 id, u64 ju, u64 jd) {
  int n = popc64(iu & ~ju) + popc64(id & ~jd);
  return n > 2 ? -1 : n;
}

// Sum of ints[idx(j)] over the set bits j of `bits`, ascending, added one by one onto `acc` in that order -- but loaded four at
// a time: the loads of a group are independent and in flight together, only the additions keep the reference's sequence
// (a determinant's H_ii is ~50 integral reads; one L2 round trip each made it a 30 us chain).
#define SQ_ORDERED_SUM4(ACC, BITS, SIGN, IDXEXPR)                                                         \
  for (u64 b_ = (BITS); b_;) {                                                                            \
    int j_[4]; double v_[4]; int n_ = 0;                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) { j_[q_] = 0; if (b_) { j_[q_] = ctz64(b_) + 1; b_ &= b_ - 1; n_ = q_ + 1; } } \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) { const int j = j_[q_]; v_[q_] = (q_ < n_) ? ints[IDXEXPR] : 0.0; }          \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) if (q_ < n_) ACC = ACC SIGN v_[q_];                  \
  }
__device__ inline double h_diag(const ChemTab &t, const double *__restrict__ ints, u64 up, u64 dn) {
  const int n1 = t.norb + 1;
  double e1 = 0.0;
  SQ_ORDERED_SUM4(e1, up, +, integral_index(t, j, j, n1, n1))
  if (dn == up) e1 = e1 * 2.0;
  else { SQ_ORDERED_SUM4(e1, dn, +, integral_index(t, j, j, n1, n1)) }
  double ex = 0.0, di = 0.0;
  for (u64 a = up; a; a &= a - 1) { const int i = ctz64(a) + 1; SQ_ORDERED_SUM4(ex, a & (a - 1), -, integral_index(t, i, j, j, i)) }
  if (dn == up) ex = ex * 2.0;
  else if (dn != 0)
    for (u64 a = dn; a; a &= a - 1) { const int i = ctz64(a) + 1; SQ_ORDERED_SUM4(ex, a & (a - 1), -, integral_index(t, i, j, j, i)) }
  // direct term: orbitals i ascending; for each, up-up (j>i), up-dn (all j), then dn-dn (j>i)
  for (u64 occ = up | dn; occ; occ &= occ - 1) {
    const int i0 = ctz64(occ), i = i0 + 1;
    if ((up >> i0) & 1) {
      SQ_ORDERED_SUM4(di, up & ~maskr64(i), +, integral_index(t, i, i, j, j))
      SQ_ORDERED_SUM4(di, dn, +, integral_index(t, i, i, j, j))
    }
    if ((dn >> i0) & 1) { SQ_ORDERED_SUM4(di, dn & ~maskr64(i), +, integral_index(t, i, i, j, j)) }
  }
  return e1 + (ex + di) + t.nuclear;
}

__device__ inline double h_single(const ChemTab &t, const double *__restrict__ ints, u64 iu, u64 id, u64 ju, u64 jd) {
  const int n1 = t.norb + 1;
  u64 a = iu, b = id, aj = ju;
  if (iu == ju) { a = id; b = iu; aj = jd; }
  int ib = ctz64(a & ~aj) + 1, jb = ctz64(aj & ~a) + 1;
  int pf = permutation_factor(a, aj);
  double one = pf * IVAL(ib, jb, n1, n1);
  // The terms are added one by one in the reference's order (chemistry.f90:1845-1930), but fetched four orbitals at a time: the loads
  // of a group do not depend on the running sum, only the additions do.  (A single excitation is rare -- 1.6 % of the C2 proposals --
  // but two waves in three hold one, and its 20-odd integrals one L2 round trip after the other were the longest chain in k_spawn.)
  double e = 0.0;
  for (u64 d = a & ~bit64(ib - 1); d;) {            // jb is empty in a; ib is skipped as the reference's test does
    int j_[4]; double x_[4], y_[4]; int n_ = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { j_[q] = 0; if (d) { j_[q] = ctz64(d) + 1; d &= d - 1; n_ = q + 1; } }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int i = j_[q]; x_[q] = (q < n_) ? IVAL(ib, i, i, jb) : 0.0; y_[q] = (q < n_) ? IVAL(ib, jb, i, i) : 0.0; }
#pragma unroll
    for (int q = 0; q < 4; q++) if (q < n_) e = e - x_[q] + y_[q];
  }
  for (u64 d = b; d;) {
    int j_[4]; double y_[4]; int n_ = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { j_[q] = 0; if (d) { j_[q] = ctz64(d) + 1; d &= d - 1; n_ = q + 1; } }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int i = j_[q]; y_[q] = (q < n_) ? IVAL(ib, jb, i, i) : 0.0; }
#pragma unroll
    for (int q = 0; q < 4; q++) if (q < n_) e = e + y_[q];
  }
  return one + pf * e;
}

__device__ inline double h_double(const ChemTab &t, const double *__restrict__ ints, u64 iu, u64 id, u64 ju, u64 jd) {
  int g, i1, i2, j1, j2;
  if (iu == ju || id == jd) {            // both electrons in one spin string (dn if up is untouched)
    const bool in_dn = (iu == ju);
    permutation_factor2(in_dn ? id : iu, in_dn ? jd : ju, g, i1, i2, j1, j2);
    return g * (IVAL(i1 + 1, j1 + 1, i2 + 1, j2 + 1) - IVAL(i1 + 1, j2 + 1, i2 + 1, j1 + 1));
  }
  i1 = ctz64(iu & ~ju); j1 = ctz64(ju & ~iu);
  i2 = ctz64(id & ~jd); j2 = ctz64(jd & ~id);
  return permutation_factor(iu, ju) * permutation_factor(id, jd) * IVAL(i1 + 1, j1 + 1, i2 + 1, j2 + 1);
}

__device__ inline double h_level(const ChemTab &t, const double *__restrict__ ints, u64 iu, u64 id, u64 ju, u64 jd, int level) {
  if (level == 0) return h_diag(t, ints, iu, id);
  if (level == 1) return h_single(t, ints, iu, id, ju, jd);
  if (level == 2) return h_double(t, ints, iu, id, ju, jd);
  return 0.0;
}
// chemistry.f90:1323-1377
__device__ inline double h_time_sym(const ChemTab &t, const double *__restrict__ ints, u64 iu, u64 id, u64 ju, u64 jd) {
  const double sqrt2 = sqrt(2.0), sqrt2inv = 1.0 / sqrt2;
  double m1 = 0.0, m2 = 0.0, norm_ketinv = 1.0, norm_bra = 1.0; bool check = true; int lev;
  if (ju == jd) norm_ketinv = sqrt2inv;
  if (iu == id) { norm_bra = sqrt2; check = false; }
  lev = (iu == ju && id == jd) ? 0 : excitation_level(iu, id, ju, jd);
  if (lev >= 0) m1 = h_level(t, ints, iu, id, ju, jd, lev);
  if (check) {
    if (ju != jd) {
      lev = excitation_level(id, iu, ju, jd);
      if (lev >= 0) m2 = h_level(t, ints, id, iu, ju, jd, lev);
    } else m2 = m1;
  }
  return (norm_bra * norm_ketinv) * (m1 + (t.z * m2));
}
// dispatcher semistoch.f90:2234-2302
// ---- HEG matrix elements, heg.f90:845-1011 (same operation order as the reference)
__device__ __forceinline__ double heg_sumsq(const ChemTab &t, const double *v) { double s = 0.0; for (int j = 0; j < t.n_dim; j++) s = s + v[j] * v[j]; return s; }
__device__ __forceinline__ double heg_inv_k2(const ChemTab &t, int p, int q) {
  const double FOUR_PI = 4.0 * (4.0 * atan(1.0));
  double s = 0.0;
  for (int j = 0; j < t.n_dim; j++) { const double d = t.kvec[p][j] - t.kvec[q][j]; s = s + d * d; }
  return FOUR_PI / s;
}
__device__ __forceinline__ int heg_gamma_exp(u64 det, u64 eor) {
  int g = 0;
  for (u64 e = eor & det; e; e &= e - 1) { const int o = ctz64(e); g += popc64(det & maskr64(o)); }
  return g;
}
__device__ inline double h_heg(const ChemTab &t, u64 iu, u64 id, u64 ju, u64 jd) {
  const double L = t.length_cell;
  if (iu == ju && id == jd) {
    double me = 0.0;
    for (u64 d = iu; d; d &= d - 1) me = me + heg_sumsq(t, t.kvec[ctz64(d) + 1]) * 0.5;
    for (u64 d = id; d; d &= d - 1) me = me + heg_sumsq(t, t.kvec[ctz64(d) + 1]) * 0.5;
    double pot = 0.0;
    for (u64 a = iu; a; a &= a - 1) for (u64 b = a & (a - 1); b; b &= b - 1) pot = pot + heg_inv_k2(t, ctz64(a) + 1, ctz64(b) + 1);
    for (u64 a = id; a; a &= a - 1) for (u64 b = a & (a - 1); b; b &= b - 1) pot = pot + heg_inv_k2(t, ctz64(a) + 1, ctz64(b) + 1);
    return me - pot / (L * L * L);
  }
  const u64 eu = iu ^ ju, ed = id ^ jd;
  const int neu = popc64(eu), ned = popc64(ed);
  if (neu + ned != 4) return 0.0;
  double mc[3] = {0.0, 0.0, 0.0}; int op = 0, oq = 0, os = 0;
  for (int sp = 0; sp < 2; sp++) {
    u64 e = sp ? ed : eu; const u64 di = sp ? id : iu;
    for (; e; e &= e - 1) {
      const int o = ctz64(e) + 1;
      if ((di >> (o - 1)) & 1) { for (int j = 0; j < t.n_dim; j++) mc[j] = mc[j] - t.kvec[o][j]; if (!op) op = o; }
      else { for (int j = 0; j < t.n_dim; j++) mc[j] = mc[j] + t.kvec[o][j]; if (!oq) oq = o; else if (!os) os = o; }
    }
  }
  if (heg_sumsq(t, mc) * (L * L) > 1.0e-15) return 0.0;
  double pot = heg_inv_k2(t, op, oq);
  if (neu != 2) pot = pot - heg_inv_k2(t, op, os);
  const int g = heg_gamma_exp(iu, eu) + heg_gamma_exp(ju, eu) + heg_gamma_exp(id, ed) + heg_gamma_exp(jd, ed);
  if (g & 1) pot = -pot;
  return pot / (L * L * L);
}

// ---- real-space Hubbard, hamiltonian_hubbard (hubbard.f90:1536-1644) with the Jordan-Wigner
// phase of fermionic_phase (more_tools.f90:140-176).  The reference "assumes inherent
// connectedness" of the pair it is given; here the hop must also be a lattice bond
// (is_connected_hubbard), so that the all-pairs matrix builder can call it on anything.
__device__ __forceinline__ double h_hubbard(const ChemTab &t, u64 iu, u64 id, u64 ju, u64 jd) {
  if (iu == ju && id == jd) return t.hub_U * popc64(iu & id);
  u64 cfg, cj;
  if (iu == ju) { cfg = id; cj = jd; } else if (id == jd) { cfg = iu; cj = ju; } else return 0.0;
  const u64 a = cfg & ~cj, b = cj & ~cfg;
  if (popc64(a) != 1 || popc64(b) != 1) return 0.0;
  const int p1 = ctz64(a) + 1, p2 = ctz64(b) + 1;
  if (t.hub_nbr[p1][0] != p2 && t.hub_nbr[p1][1] != p2 && t.hub_nbr[p1][2] != p2 && t.hub_nbr[p1][3] != p2) return 0.0;
  const int lo = p1 < p2 ? p1 : p2, hi = p1 < p2 ? p2 : p1;         // sites strictly between: 0-based bits lo .. hi-2
  const int phase = (popc64(cfg & maskr64(hi - 1) & ~maskr64(lo)) & 1) ? -1 : 1;
  return -t.hub_t * phase;
}

__device__ inline double h_any(const ChemTab &t, const double *__restrict__ ints, u64 iu, u64 id, u64 ju, u64 jd) {
  if (t.sys_type == 1) return h_heg(t, iu, id, ju, jd);
  if (t.sys_type == 2) return h_hubbard(t, iu, id, ju, jd);
  if (t.time_sym) return h_time_sym(t, ints, iu, id, ju, jd);
  int lev = excitation_level(iu, id, ju, jd);
  return lev < 0 ? 0.0 : h_level(t, ints, iu, id, ju, jd, lev);
}

// ------------------------------------------------------------------------- proposal
// One uniform symmetry-aware proposal from det_i.  Returns excitation level (1/2) with
// det_j and the generation probability, or 0 when the reference returns with weight 0.
// Consumes exactly the reference's random_int calls, in its order.
__device__ inline int propose_uniform(const ChemTab &t, Rng &g, u64 iu, u64 id, u64 &ju, u64 &jd, double &prob) {
  const int nup = t.nup, ndn = t.ndn, norb = t.norb, nc = t.ncore, nelec = t.nelec;
  const int n_single = (nup - nc) * (norb - nup) + (ndn - nc) * (norb - ndn);
  const int n_double_up = (nup - nc) * (nup - nc - 1) * (norb - nup) * (norb - nup - 1) / 4;
  const int n_double_dn = (ndn - nc) * (ndn - nc - 1) * (norb - ndn) * (norb - ndn - 1) / 4;
  const int n_double_both = (nup - nc) * (norb - nup) * (ndn - nc) * (norb - ndn);
  const int n_double = n_double_up + n_double_dn + n_double_both, n_total = n_single + n_double;
  int level, e1 = 0, e2 = 0, tot_spin = 0;
  prob = 1.0;
  ju = iu; jd = id;
  if (rng_int(g, n_total) > n_single) {
    level = 2; prob = n_double / (double)n_total;
    e1 = rng_int(g, nelec - 2 * nc); e2 = rng_int(g, nelec - 2 * nc - 1);
    if (e2 == e1) e2 = nelec - 2 * nc;
    if (e1 > nup - nc) { tot_spin -= 1; e1 += 2 * nc; } else { tot_spin += 1; e1 += nc; }
    if (e2 > nup - nc) { tot_spin -= 1; e2 += 2 * nc; } else { tot_spin += 1; e2 += nc; }
  } else {
    level = 1; prob = prob * n_single / (n_total * 1.0);
    e1 = rng_int(g, nelec - 2 * nc);
    if (e1 > nup - nc) { tot_spin = -1; e1 += 2 * nc; } else { tot_spin = 1; e1 += nc; }
  }
  { int s = e1 + e2; e1 = s - (e1 > e2 ? e1 : e2); e2 = s - e1; }
  // electron index e (1-based over up electrons then dn electrons, each ascending) -> orbital
  const int n_occ_up = popc64(iu);
  int sym1, o;
  {
    const bool up2 = (e2 <= n_occ_up);
    o = kth_set(up2 ? iu : id, up2 ? e2 : e2 - n_occ_up);
    if (up2) ju &= ~bit64(o - 1); else jd &= ~bit64(o - 1);
    sym1 = t.orbsym[o];
    if (level == 2) {
      const bool up1 = (e1 <= n_occ_up);
      const int o1 = kth_set(up1 ? iu : id, up1 ? e1 : e1 - n_occ_up);
      if (up1) ju &= ~bit64(o1 - 1); else jd &= ~bit64(o1 - 1);
      sym1 = t.prod[t.orbsym[o1]][sym1];
    }
  }
  int i_open, sp1, sym2; double temp1;
  if (level == 1) {
    prob = prob / (nelec - 2 * nc);
    const u64 occdet = (tot_spin == 1) ? iu : id;
    const u64 open = t.sym_mask[sym1] & ~occdet;
    i_open = popc64(open);
    if (i_open == 0) return 0;
    int to1 = rng_int(g, i_open); prob = prob / i_open;
    o = kth_set_wide(open, to1);
    if (tot_spin == 1) ju |= bit64(o - 1); else jd |= bit64(o - 1);
  } else {
    prob = prob * 2.0 / (1.0 * (nelec - 2 * nc) * (nelec - 2 * nc - 1));
    // One code path for the three spin cases (every wave holds all of them): d1 / d2 are the
    // strings the first / second hole is drawn from; for equal spins the second hole may not
    // repeat the first.  chemistry.f90:4716-4985
    const bool same_spin = (tot_spin != 0);
    bool first_up = (tot_spin == 2); u64 d1 = first_up ? iu : id, d2 = d1; int nchoice;
    if (same_spin) { nchoice = norb - (first_up ? nup : ndn); prob = prob / nchoice; }
    else { nchoice = 2 * norb - nup - ndn; prob = prob * 1.0 / (2 * norb - ndn - nup); }
    int to1 = rng_int(g, nchoice);
    if (!same_spin) {
      first_up = (to1 <= norb - nup);
      if (first_up) { d1 = iu; d2 = id; } else { d1 = id; d2 = iu; to1 -= (norb - nup); }
    }
    const bool second_up = same_spin ? first_up : !first_up;
    sp1 = kth_set_wide(t.orb_mask & ~d1, to1);
    if (first_up) ju |= bit64(sp1 - 1); else jd |= bit64(sp1 - 1);
    const int s1 = t.orbsym[sp1];
    sym2 = same_spin ? t.prod[s1][sym1] : t.prod[sym1][s1];
    const bool same = same_spin && (sym2 == s1);
    u64 open = t.sym_mask[sym2] & ~d2; if (same) open &= ~bit64(sp1 - 1);
    i_open = popc64(open);
    if (i_open == 0) return 0;
    const int to2 = rng_int(g, i_open); temp1 = 1.0 / i_open;
    o = kth_set_wide(open, to2);
    if (second_up) ju |= bit64(o - 1); else jd |= bit64(o - 1);
    const int sy = t.prod[sym2][sym1];
    i_open = popc64(t.sym_mask[sy] & ~d1) - (same ? 1 : 0);
    if (i_open == 0) prob = prob * temp1; else prob = prob * (temp1 + (1.0 / i_open));
  }
  return level;
}

// is_connected_chem, chemistry.f90:2003-2200 (uniform-proposal branch): excitation level of the
// pair and the probability with which propose_uniform would have produced det_j from det_i
// (without the n_single/n_total or n_double/n_total level factor).
__device__ inline bool is_connected_prob(const ChemTab &t, u64 iu, u64 id, u64 ju, u64 jd, int &level, double &prob) {
  int upc = 0, dnc = 0, du1 = 0, du2 = 0, du3 = 0, du4 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0;
  level = -1; prob = 0.0;
  if (iu != ju) {
    const u64 a = iu & ~ju, b = ju & ~iu;
    upc = popc64(a);
    if (upc > 2 || upc != popc64(b)) return false;
    du1 = ctz64(a) + 1; du3 = ctz64(b) + 1;
    if (upc == 2) { du2 = ctz64(a & (a - 1)) + 1; du4 = ctz64(b & (b - 1)) + 1; }
  }
  if (id != jd) {
    const u64 a = id & ~jd, b = jd & ~id;
    dnc = popc64(a);
    if (dnc > 2 || dnc != popc64(b)) return false;
    dd1 = ctz64(a) + 1; dd3 = ctz64(b) + 1;
    if (dnc == 2) { dd2 = ctz64(a & (a - 1)) + 1; dd4 = ctz64(b & (b - 1)) + 1; }
  }
  level = upc + dnc;
  if (level > 2) { level = -1; return false; }
  const int ne = t.nelec - 2 * t.ncore;
  if (level == 1) {
    u64 det = id; int d1 = dd1, d2 = dd3;
    if (upc == 1) { det = iu; d1 = du1; d2 = du3; }
    const int sym1 = t.orbsym[d1];
    if (sym1 != t.orbsym[d2]) return false;
    const int i_open = popc64(t.sym_mask[sym1] & ~det);
    prob = 1.0 / ((ne) * (i_open));
    return true;
  }
  if (level == 2) {
    int d1, d2, d3, d4; u64 det1, det2; double tp, tp2;
    if (upc == 2) { d1 = du1; d2 = du2; d3 = du3; d4 = du4; det1 = iu | bit64(d3 - 1); det2 = iu | bit64(d4 - 1); tp = 2.0 / (t.norb - t.nup); tp2 = 2.0 / (t.norb - t.nup); }
    else if (dnc == 2) { d1 = dd1; d2 = dd2; d3 = dd3; d4 = dd4; det1 = id | bit64(d3 - 1); det2 = id | bit64(d4 - 1); tp = 2.0 / (t.norb - t.ndn); tp2 = 2.0 / (t.norb - t.ndn); }
    else { d1 = du1; d2 = dd1; d3 = du3; d4 = dd3; det2 = iu; det1 = id; tp = 1.0 / (t.norb - t.nup); tp2 = 1.0 / (t.norb - t.ndn); }
    const int sym1 = t.prod[t.orbsym[d1]][t.orbsym[d2]];
    if (sym1 != t.prod[t.orbsym[d3]][t.orbsym[d4]]) return false;
    // orbitals i with product(sym(d3), sym(i)) == sym1 form one irrep in an abelian group
    const int i_open = popc64(t.sym_mask[t.prod[t.orbsym[d3]][sym1]] & ~det1);
    const int i_open2 = popc64(t.sym_mask[t.prod[t.orbsym[d4]][sym1]] & ~det2);
    if (i_open == 0 && i_open2 != 0) prob = (1.0 / ((ne) * (ne - 1))) * ((tp2 / (i_open2)));
    if (i_open2 == 0 && i_open != 0) prob = (1.0 / ((ne) * (ne - 1))) * ((tp / (i_open)));
    if (i_open2 != 0 && i_open != 0) prob = (1.0 / ((ne) * (ne - 1))) * ((tp / (i_open)) + (tp2 / (i_open2)));
    return true;
  }
  return true;
}

// weight_j = -tau * H_ij / p_gen of an accepted proposal; with time-reversal symmetry the matrix
// element and the generation probability get the second pathway through the time-reversed
// determinant and det_j is replaced by its representative (chemistry.f90:4988-5069).
__device__ inline double proposal_weight(const ChemTab &t, const double *__restrict__ ints, double tau, u64 iu, u64 id, u64 &ju, u64 &jd,
                                         int level, double prob) {
  if (t.sys_type == 1) {                 // heg.f90:1592-1596
    const double me = h_heg(t, iu, id, ju, jd);
    const double acc = tau * fabs(me) / prob;
    return acc * copysign(1.0, -me);
  }
  if (t.sys_type == 2) {                 // hubbard.f90:3111-3116; prob holds proposal_prob_inv = nelec * ctr
    const double me = h_hubbard(t, iu, id, ju, jd);
    const double acc = fabs(me) * tau * prob;
    return acc * copysign(1.0, -me);
  }
  if (!t.time_sym) return -tau * h_level(t, ints, iu, id, ju, jd, level) / prob;
  const double sqrt2 = sqrt(2.0);
  const double norm_i = (iu == id) ? sqrt2 : 1.0;
  if ((ju == iu && jd == id) || (jd == iu && ju == id)) return 0.0;
  double me;
  if (ju == jd) {
    if (t.z != 1) return 0.0;
    me = h_level(t, ints, iu, id, ju, jd, level);
    me = (sqrt2 / norm_i) * me;
  } else {
    const double m1 = h_level(t, ints, iu, id, ju, jd, level);
    int lsym; double psym;
    if (is_connected_prob(t, iu, id, jd, ju, lsym, psym)) {
      const double m2 = h_level(t, ints, iu, id, jd, ju, lsym);
      const int nup = t.nup, ndn = t.ndn, norb = t.norb, nc = t.ncore;
      const int n_single = (nup - nc) * (norb - nup) + (ndn - nc) * (norb - ndn);
      const int n_double = (nup - nc) * (nup - nc - 1) * (norb - nup) * (norb - nup - 1) / 4 + (ndn - nc) * (ndn - nc - 1) * (norb - ndn) * (norb - ndn - 1) / 4
                         + (nup - nc) * (norb - nup) * (ndn - nc) * (norb - ndn);
      const int n_total = n_single + n_double;
      if (lsym == 1) prob = prob + (psym * (n_single / (double)n_total));
      if (lsym == 2) prob = prob + (psym * (n_double / (double)n_total));
      me = (1.0 / norm_i) * (m1 + t.z * m2);
    } else me = (1.0 / norm_i) * (m1);
  }
  if (ju > jd) { const u64 x = ju; ju = jd; jd = x; me = me * t.z; }
  return -tau * me / prob;
}

// off_diagonal_move_heg, heg.f90:1344-1598: two electrons uniformly (rejection on equal picks),
// one hole uniformly among the orbitals empty in det_i of the right spin, the second hole fixed
// by momentum conservation (first match scanning orbitals upwards).  Returns 2 or 0.
__device__ inline int propose_heg(const ChemTab &t, Rng &g, u64 iu, u64 id, u64 &ju, u64 &jd, double &prob) {
  const int nelec = t.nelec, nup = t.nup, ndn = t.ndn, norb = t.norb, nd = t.n_dim;
  ju = iu; jd = id;
  const int e1 = rng_int(g, nelec);
  int e2;
  do { e2 = rng_int(g, nelec); } while (e1 == e2);
  const int spin = ((e1 > nup) ? -1 : 1) + ((e2 > nup) ? -1 : 1);
  double from[3] = {0.0, 0.0, 0.0}, to1[3] = {0.0, 0.0, 0.0};
  int kf[3] = {0, 0, 0}, i_first = 0;
  // the two electrons in the order the reference's loop over the occupied orbitals meets them (up ascending, then dn ascending):
  // electron e of that order is the e-th set bit of iu, or the (e - nup)-th of id
  {
    const int ea = e1 < e2 ? e1 : e2, eb = e1 < e2 ? e2 : e1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int e = q ? eb : ea; const bool isup = e <= nup;
      const int i = kth_set_wide(isup ? iu : id, isup ? e : e - nup);
      for (int j = 0; j < nd; j++) from[j] = from[j] + t.kvec[i][j];
      for (int j = 0; j < 3; j++) kf[j] += t.krel[i][j];
      if (isup) ju &= ~bit64(i - 1); else jd &= ~bit64(i - 1);
    }
  }
  bool first_up, second_up; int to1n; float denom;
  if (spin == 2) { to1n = rng_int(g, norb - nup); first_up = true; second_up = true; denom = (float)(nelec * (nelec - 1) * (norb - nup)); }
  else if (spin == -2) { to1n = rng_int(g, norb - ndn); first_up = false; second_up = false; denom = (float)(nelec * (nelec - 1) * (norb - ndn)); }
  else {
    to1n = rng_int(g, 2 * norb - nup - ndn); denom = (float)(nelec * (nelec - 1) * (2 * norb - nelec));
    if (to1n <= norb - nup) { first_up = true; second_up = false; } else { to1n -= (norb - nup); first_up = false; second_up = true; }
  }
  {
    const int i = kth_set_wide(t.orb_mask & ~(first_up ? iu : id), to1n);      // (up to 50 free orbitals: no data-dependent loop)
    i_first = i;
    for (int j = 0; j < nd; j++) to1[j] = to1[j] + t.kvec[i][j];
    if (first_up) ju |= bit64(i - 1); else jd |= bit64(i - 1);
  }
  const u64 free2 = t.orb_mask & ~(second_up ? (iu | ju) : (id | jd));
  if (t.c2_pad) {
    // The reference scans the free orbitals upwards for the first one whose plane wave closes the momentum balance to 1e-15 (1451-1467).
    // Plane waves are integer multiples of 2 pi / L: at most ONE orbital can -- the one whose integer vector is the balance -- and any
    // other misses by at least 2 pi / L.  Look that one up, then put it to the reference's own floating-point test: the same decision.
    const int nm = t.heg_nmax, W = 2 * nm + 1;
    int idx = 0; bool inside = true;
    for (int j = 0; j < 3; j++) { const int kj = kf[j] - t.krel[i_first][j]; inside = inside && kj >= -nm && kj <= nm; idx = idx * W + (kj + nm); }
    const int i = inside ? (int)reinterpret_cast<const unsigned char *>(t.c2)[idx] : 0;
    if (i == 0 || !((free2 >> (i - 1)) & 1ull)) return 0;
    int ok = 0;
    for (int j = 0; j < nd; j++) if (fabs(from[j] - (to1[j] + t.kvec[i][j])) < 1.0e-15) ok++;
    if (ok != nd) return 0;
    if (second_up) ju |= bit64(i - 1); else jd |= bit64(i - 1);
    prob = (double)(4.0f / denom);
    return 2;
  }
  for (u64 fr = free2; fr; fr &= fr - 1) {
    const int i = ctz64(fr) + 1;
    int ok = 0;
    for (int j = 0; j < nd; j++) if (fabs(from[j] - (to1[j] + t.kvec[i][j])) < 1.0e-15) ok++;
    if (ok == nd) {
      if (second_up) ju |= bit64(i - 1); else jd |= bit64(i - 1);
      prob = (double)(4.0f / denom);      // the reference evaluates 4./(integer) in single precision
      return 2;
    }
  }
  return 0;
}
// off_diagonal_move_hubbard, hubbard.f90:2992-3120 (vmc off): an electron by rejection over the
// 2*nsites spin-sites (choose_random_electron, 1024-1058), then one of its empty lattice
// neighbours (LEFT, RIGHT, UP, DOWN order).  Returns 1 with prob = the INVERSE proposal
// probability real((nup+ndn)*ctr) the reference multiplies with, or 0 (no empty neighbour).
#define SQ_HUB_MAX_TRIES 4096            // the rejection loop ends with probability 1; the bound keeps a wave from spinning on a corrupt (empty) determinant
__device__ inline int propose_hubbard(const ChemTab &t, Rng &g, u64 iu, u64 id, u64 &ju, u64 &jd, double &prob) {
  ju = iu; jd = id;
  const int ns2 = 2 * t.norb;
  int site = 0, spin = 0; bool found = false;
  for (int it = 0; it < SQ_HUB_MAX_TRIES && !found; it++) {
    const int cs = rng_int(g, ns2);
    if ((cs & 1) == 0) { spin = 0; site = cs / 2; found = (id >> (site - 1)) & 1; }
    else { spin = 1; site = (cs + 1) / 2; found = (iu >> (site - 1)) & 1; }
  }
  if (!found) return 0;
  const u64 det = spin ? iu : id;
  int ctr = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int nb = t.hub_nbr[site][k]; if (nb && !((det >> (nb - 1)) & 1)) ctr++; }
  if (ctr == 0) return 0;
  const int tk = rng_int(g, ctr);
  int target = 0, seen = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int nb = t.hub_nbr[site][k]; if (nb && !((det >> (nb - 1)) & 1)) { seen++; if (seen == tk) target = nb; } }
  const u64 dj = (det | bit64(target - 1)) & ~bit64(site - 1);
  if (spin) ju = dj; else jd = dj;
  prob = (double)(t.nelec * ctr);
  return 1;
}
// proposal of the system at hand (the procedure pointer `move`, do_walk.f90:126-134, 3599-3633)
__device__ __forceinline__ int propose_any(const ChemTab &t, Rng &g, u64 iu, u64 id, u64 &ju, u64 &jd, double &prob) {
  if (t.sys_type == 1) return propose_heg(t, g, iu, id, ju, jd, prob);
  if (t.sys_type == 2) return propose_hubbard(t, g, iu, id, ju, jd, prob);
  return propose_uniform(t, g, iu, id, ju, jd, prob);
}

// active-space masks of find_important_connected_dets_chem (chemistry.f90:6840-6846, 6926-6947, 7087-7108): mode 0 none, 1 only the
// determinants inside the active space (core orbitals all occupied, virtual orbitals all empty), 2 only those outside it
struct ActiveSpace { int mode; u64 core_up, core_dn, virt_up, virt_dn; };
__device__ __forceinline__ bool active_space_skip(const ActiveSpace &a, u64 nu, u64 nd) {
  if (!a.mode) return false;
  const bool outside = ((a.core_up & nu) != a.core_up) || ((a.core_dn & nd) != a.core_dn) || (a.virt_up & nu) != 0 || (a.virt_dn & nd) != 0;
  return a.mode == 1 ? outside : !outside;
}
