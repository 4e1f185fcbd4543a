// door_kernels.h -- batch / test door kernels (matrix elements, proposals, reduce, join)
// Textually included by sqmc_gpu.hip (one translation unit: the kernels share the ChemTab LDS
// image, the walker SoA types and the launch helpers defined there); not a standalone header.

// ============================================================ batch / test door kernels
__global__ void __launch_bounds__(TPB) k_ham_batch(ChemDev dev, const u64 *iu, const u64 *id, const u64 *ju, const u64 *jd, double *h, long long n) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) h[i] = h_any(t, dev.integrals, iu[i], id[i], ju[i], jd[i]);
}
// second_order_pt's sum (hci.f90:1140-1175): one term per deduplicated connection that is NOT in the variational space (binary
// search on its sorted determinant ranks), num^2 / (E_var - H_aa); grid-stride, one partial sum per block (fixed tree)
__global__ void __launch_bounds__(TPB) k_pt2_terms(ChemDev dev, const u64 *__restrict__ cu, const u64 *__restrict__ cd, const double *__restrict__ num, long long n,
                                                   const u64 *__restrict__ vkeys, long long nv, double e_var, double *__restrict__ partial) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const u64 u = cu[i], d = cd[i];
    const u64 key = det_key(dev, u, d);
    long long lo = 0, hi = nv;                         // first variational rank >= key
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (vkeys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < nv && vkeys[lo] == key) continue;         // inside the variational space
    const double haa = h_any(t, dev.integrals, u, d, u, d);
    const double x = num[i];
    acc += x * x / (e_var - haa);
  }
  __shared__ double red[TPB / 64];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double v = 0.0; for (int q = 0; q < TPB / 64; q++) v += red[q]; partial[blockIdx.x] = v; }
}
__global__ void __launch_bounds__(TPB) k_ham_chem_batch(ChemDev dev, const u64 *iu, const u64 *id, const u64 *ju, const u64 *jd, double *h, long long n) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int lev = excitation_level(iu[i], id[i], ju[i], jd[i]);
  h[i] = lev < 0 ? 0.0 : h_level(t, dev.integrals, iu[i], id[i], ju[i], jd[i], lev);
}

// Sparse Hamiltonian among a sorted determinant list by brute force over all pairs: a
// popcount filter (<= 2 orbital differences, also against the time-reversed partner when
// time_sym) in front of the Slater-Condon evaluation.  One thread per row, column
// determinants staged through LDS in tiles.  pass 0 counts, pass 1 fills at the scanned
// offsets; each row holds its diagonal first, then columns j < i ascending.
// replaces: generate_sparse_ham_chem_upper_triangular (chemistry.f90:7639-8010)
__global__ void __launch_bounds__(TPB) k_build_ham(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, long long n, int pass,
                                                   u64 *__restrict__ counts, const u64 *__restrict__ offs, long long *__restrict__ idx, double *__restrict__ val) {
  __shared__ ChemTab t;
  __shared__ u64 su[TPB], sd[TPB];
  stage_tab(&t, dev.tab, dev.tab_words);
  const long long r0 = (long long)blockIdx.x * TPB, i = r0 + threadIdx.x;
  const bool live = i < n;
  const u64 ui = live ? up[i] : 0, di = live ? dn[i] : 0;
  u64 cnt = 0; const u64 base = (pass && live) ? offs[i] : 0;
  if (live) {
    if (pass) { idx[base] = i + 1; val[base] = h_any(t, dev.integrals, ui, di, ui, di); }
    cnt = 1;
  }
  const long long jend = (r0 + TPB < n) ? r0 + TPB : n;
  for (long long j0 = 0; j0 < jend; j0 += TPB) {
    __syncthreads();
    { long long j = j0 + threadIdx.x; su[threadIdx.x] = (j < n) ? up[j] : 0; sd[threadIdx.x] = (j < n) ? dn[j] : 0; }
    __syncthreads();
    if (!live) continue;
    const int lim = (int)((i - j0 < TPB) ? (i - j0) : TPB);      // only j < i
    for (int q = 0; q < lim; q++) {
      const u64 uj = su[q], dj = sd[q];
      bool cand = (popc64(ui ^ uj) + popc64(di ^ dj)) <= 4;
      if (!cand && t.time_sym) cand = (popc64(ui ^ dj) + popc64(di ^ uj)) <= 4;
      if (!cand) continue;
      const double h = h_any(t, dev.integrals, ui, di, uj, dj);
      if (h == 0.0) continue;
      if (pass) { idx[base + cnt] = j0 + q + 1; val[base + cnt] = h; }
      cnt++;
    }
  }
  if (!pass && live) counts[i] = cnt;
}

__global__ void __launch_bounds__(TPB) k_propose_heatbath_batch(ChemDev dev, const u64 *up, const u64 *dn, const u64 *state_in, u64 *ju, u64 *jd,
                                                                double *wj, u64 *state_out, long long n, double tau) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  Rng g; g.mode = 0; g.x = state_in[i];
  u64 a[2], b[2]; double w[2];
  propose_heatbath(t, dev.integrals, dev.hb, g, tau, up[i], dn[i], a, b, w);
  for (int k = 0; k < 2; k++) { ju[2 * i + k] = a[k]; jd[2 * i + k] = b[k]; wj[2 * i + k] = w[k]; }
  state_out[i] = g.x;
}
__global__ void __launch_bounds__(TPB) k_propose_batch(ChemDev dev, const u64 *up, const u64 *dn, const u64 *state_in, u64 *ju, u64 *jd,
                                                       double *wj, u64 *state_out, long long n, double tau) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  Rng g; g.mode = 0; g.x = state_in[i];
  u64 a, b; double prob;
  int level = propose_any(t, g, up[i], dn[i], a, b, prob);
  double w = 0.0;
  if (level > 0) w = proposal_weight(t, dev.integrals, tau, up[i], dn[i], a, b, level, prob);
  ju[i] = a; jd[i] = b; wj[i] = w; state_out[i] = g.x;
}

// ---- host helpers of sqmc_gpu_set_heatbath_tables (templates: outside the extern "C" block)
// column-major (i,j,k) of extent n^3 -> [i][j][k] with k fastest
template <typename T>
static std::vector<T> hb_transpose3(const T *src, int n) {
  std::vector<T> out((size_t)n * n * n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) out[((size_t)i * n + j) * n + k] = src[(size_t)i + (size_t)n * ((size_t)j + (size_t)n * k)];
  return out;
}
template <typename T>
static int hb_upload(sqmc_gpu_ctx *c, int slot, const T *host, size_t count, const T **dev) {
  hipFree(c->d_hbt[slot]); c->d_hbt[slot] = nullptr;
  HIPCHK(hipMalloc(&c->d_hbt[slot], (count + 1) * sizeof(T)));
  HIPCHK(hipMemcpy(c->d_hbt[slot], host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev = (const T *)c->d_hbt[slot];
  return SQMC_OK;
}
