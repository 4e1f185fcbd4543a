// hbuild_kernels.h -- candidate generation for the sparse Hamiltonian of a sorted determinant list without testing all pairs.
// replaces: generate_sparse_ham_chem_upper_triangular with get_connected_dets_in_list (chemistry.f90:7639-8010, 9851-9991).
// Textually included by sqmc_gpu.hip; not a standalone header.
//
// Two determinants can have a nonzero element only if their alpha and beta strings differ by (0, <=2), (<=2, 0) or (1, 1)
// excitations.  With S = the sorted distinct strings of the list (alpha and beta strings together), the determinants grouped by
// alpha string (contiguous: the list is sorted by (up, dn)) and by beta string (a permutation), and for every string the
// strings one excitation away (k_str_nbr), row i's candidates are
//   A  the determinants with its alpha string                      -- beta strings within two excitations,
//   B  the determinants with its beta string, other alpha string   -- alpha strings within two excitations,
//   C  the determinants whose alpha string is one excitation away  -- beta string exactly one excitation away,
// which is the set the all-pairs test popc(up_i ^ up_j) + popc(dn_i ^ dn_j) <= 4 finds, at a cost that follows the group sizes
// instead of n.  A time-symmetrised list adds the same three families for the row's spin-flipped partner (dn_i, up_i); what both
// sources find is dropped after the sort.  Candidates are emitted as (row << 32 | column) words, sorted, and every distinct one
// gets its matrix element from its own thread (k_ham_eval): evaluation is balanced however uneven the rows are.
#pragma once

struct HamIdx {
  const u64 *S; int NS;                 // distinct strings, ascending
  const u32 *sid_up, *sid_dn;           // string of every determinant
  const u32 *ulo, *uhi;                 // determinants [ulo, uhi) carry string s as their alpha string
  const u32 *dlo, *dhi, *perm_d;        // perm_d[dlo .. dhi) = the determinants that carry string s as their beta string, ascending
  const u64 *nptr; const u32 *nbr;      // strings one excitation away from s: nbr[nptr[s] .. nptr[s+1]), ascending
};

// strings one excitation away (two bits differ): one thread per string, the others through LDS; pass 0 counts, pass 1 fills
__global__ void __launch_bounds__(TPB) k_str_nbr(const u64 *__restrict__ S, int NS, int pass, u64 *__restrict__ counts, const u64 *__restrict__ offs, u32 *__restrict__ nbr) {
  __shared__ u64 st[TPB];
  const int s = blockIdx.x * TPB + threadIdx.x;
  const bool live = s < NS;
  const u64 mine = live ? S[s] : 0ull;
  u64 cnt = 0; const u64 base = (pass && live) ? offs[s] : 0ull;
  for (int j0 = 0; j0 < NS; j0 += TPB) {
    __syncthreads();
    st[threadIdx.x] = (j0 + (int)threadIdx.x < NS) ? S[j0 + threadIdx.x] : 0ull;
    __syncthreads();
    if (!live) continue;
    const int lim = (NS - j0 < TPB) ? NS - j0 : TPB;
    for (int q = 0; q < lim; q++)
      if (popc64(mine ^ st[q]) == 2) { if (pass) nbr[base + cnt] = (u32)(j0 + q); cnt++; }
  }
  if (!pass && live) counts[s] = cnt;
}

// the candidate columns j < i of row i seen from the source strings (su, sd) with ids (a, b)
template <class F>
__device__ __forceinline__ void ham_row_candidates(const HamIdx &x, const u64 *__restrict__ up, const u64 *__restrict__ dn, u32 i, u64 su, u64 sd, u32 a, u32 b, F emit) {
  for (u32 j = x.ulo[a], e = x.uhi[a]; j < e && j < i; j++)                                  // A
    if (popc64(sd ^ dn[j]) <= 4) emit(j);
  for (u32 q = x.dlo[b], e = x.dhi[b]; q < e; q++) {                                         // B
    const u32 j = x.perm_d[q];
    if (j >= i) break;
    const u64 uj = up[j];
    if (uj != su && popc64(su ^ uj) <= 4) emit(j);
  }
  for (u64 q = x.nptr[a], e = x.nptr[a + 1]; q < e; q++) {                                   // C
    const u32 a2 = x.nbr[q];
    for (u32 j = x.ulo[a2], e2 = x.uhi[a2]; j < e2 && j < i; j++)
      if (popc64(sd ^ dn[j]) == 2) emit(j);
  }
}
// pass 0: candidates per row; pass 1: their words at the scanned offsets
__global__ void __launch_bounds__(TPB) k_ham_candidates(HamIdx x, const u64 *__restrict__ up, const u64 *__restrict__ dn, long long n, int time_sym, int pass,
                                                        u64 *__restrict__ counts, const u64 *__restrict__ offs, u64 *__restrict__ words) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const u64 ui = up[i], di = dn[i];
  const u32 a = x.sid_up[i], b = x.sid_dn[i];
  u64 cnt = 0; const u64 base = pass ? offs[i] : 0ull;
  auto emit = [&](u32 j) { if (pass) words[base + cnt] = ((u64)i << 32) | (u64)j; cnt++; };
  ham_row_candidates(x, up, dn, (u32)i, ui, di, a, b, emit);
  if (time_sym && ui != di) ham_row_candidates(x, up, dn, (u32)i, di, ui, b, a, emit);          // the spin-flipped partner as a second source
  if (!pass) counts[i] = cnt;
}
// one thread per sorted candidate: its element, and whether it is kept (nonzero, and not the copy a second source found)
__global__ void __launch_bounds__(TPB) k_ham_eval(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, const u64 *__restrict__ words, long long m,
                                                  double *__restrict__ hv, u64 *__restrict__ keep) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  __syncthreads();
  const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
  if (c >= m) return;
  const u64 w = words[c];
  u64 k = 0; double h = 0.0;
  if (c == 0 || words[c - 1] != w) {
    const u32 i = (u32)(w >> 32), j = (u32)w;
    h = h_any(t, dev.integrals, up[i], dn[i], up[j], dn[j]);
    k = (h != 0.0) ? 1ull : 0ull;
  }
  hv[c] = h; keep[c] = k;
}
// rows as generate_sparse_ham_chem_upper_triangular stores them: the diagonal first, then the columns j < i ascending.  With
// P = exclusive scan of keep and coff = candidate offsets of the rows: row i has 1 + P[coff[i+1]] - P[coff[i]] entries and begins at
// i + P[coff[i]]
__global__ void __launch_bounds__(TPB) k_ham_rows(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, long long n, const u64 *__restrict__ coff,
                                                  const u64 *__restrict__ P, long long m, u64 ptotal, u64 *__restrict__ rcount, u64 *__restrict__ roff,
                                                  long long *__restrict__ idx, double *__restrict__ val) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  __syncthreads();
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const u64 c0 = coff[i], c1 = (i + 1 < n) ? coff[i + 1] : (u64)m;
  const u64 p0 = (c0 < (u64)m) ? P[c0] : ptotal, p1 = (c1 < (u64)m) ? P[c1] : ptotal;
  rcount[i] = 1ull + (p1 - p0); roff[i] = (u64)i + p0;
  idx[(u64)i + p0] = i + 1; val[(u64)i + p0] = h_any(t, dev.integrals, up[i], dn[i], up[i], dn[i]);
}
__global__ void __launch_bounds__(TPB) k_ham_place(const u64 *__restrict__ words, const double *__restrict__ hv, const u64 *__restrict__ keep, const u64 *__restrict__ P,
                                                   long long m, long long *__restrict__ idx, double *__restrict__ val) {
  const long long c = (long long)blockIdx.x * TPB + threadIdx.x;
  if (c >= m || !keep[c]) return;
  const u64 w = words[c];
  const u64 dest = (w >> 32) + 1ull + P[c];
  idx[dest] = (long long)(u32)w + 1; val[dest] = hv[c];
}
