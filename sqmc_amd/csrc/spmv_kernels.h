// spmv_kernels.h -- deterministic-core sparse matvec kernels
// Textually included by sqmc_gpu.hip (one translation unit: the kernels share the ChemTab LDS
// image, the walker SoA types and the launch helpers defined there); not a standalone header.

// ============================================================================ SpMV
// Symmetric matrix kept as FULL CSR (int32 columns) so that every row is owned by one
// wavefront and no atomics are needed: 12 B per stored entry + 8 B gathered x.
struct sqmc_spmv_plan { long long n, nnz_full; int *d_ptr, *d_col; double *d_val, *d_x, *d_y; hipStream_t st;
                        int *u_ptr, *u_col; double *u_val; };      // u_*: the stored triangle as it came (only with SQMC_SPMV_UPPER_ATOMIC: the layout measured against the full CSR)
// ---- full CSR of the symmetric matrix on the device, from the upper-triangular storage that
// k_build_ham leaves in HBM (row i: diagonal first, then columns j < i ascending).  Row j of the
// full matrix = its stored part followed by the entries (i, j), i > j, in increasing i: the order
// comes from a STABLE sort on the column index, never from atomics, so the matvec sums in the
// same order run after run (the HCI selection thresholds see the same eigenvector bits).
__global__ void __launch_bounds__(TPB) k_csr_keys(const u64 *__restrict__ cnt, const u64 *__restrict__ off, const long long *__restrict__ idx,
                                                  u64 *__restrict__ keys, u32 *__restrict__ vals, u32 *__restrict__ rowof, u32 *__restrict__ colcount,
                                                  long long n) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const long long b = (long long)off[i], c = (long long)cnt[i];
  keys[b] = (u64)n; vals[b] = (u32)b; rowof[b] = (u32)i;                         // the diagonal sorts last and is not transposed
  for (long long k = 1; k < c; k++) {
    const long long j = idx[b + k] - 1;
    keys[b + k] = (u64)j; vals[b + k] = (u32)(b + k); rowof[b + k] = (u32)i;
    atomicAdd(&colcount[j], 1u);                                                   // a count: order-independent
  }
}
__global__ void __launch_bounds__(TPB) k_csr_rowlen(const u64 *__restrict__ cnt, const u32 *__restrict__ colcount, u64 *__restrict__ rowlen, u64 *__restrict__ colc64, long long n) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) { rowlen[i] = cnt[i] + colcount[i]; colc64[i] = colcount[i]; }
}
__global__ void __launch_bounds__(TPB) k_csr_fill_stored(const u64 *__restrict__ cnt, const u64 *__restrict__ off, const long long *__restrict__ idx, const double *__restrict__ val,
                                                         const u64 *__restrict__ ptr64, int *__restrict__ ptr, int *__restrict__ col, double *__restrict__ v,
                                                         double *__restrict__ diag, long long n, long long nnz_full) {
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i > n) return;
  if (i == n) { ptr[n] = (int)nnz_full; return; }
  const long long b = (long long)off[i], c = (long long)cnt[i], d = (long long)ptr64[i];
  ptr[i] = (int)d; diag[i] = val[b];
  for (long long k = 0; k < c; k++) { col[d + k] = (int)(idx[b + k] - 1); v[d + k] = val[b + k]; }
}
__global__ void __launch_bounds__(TPB) k_csr_fill_transposed(const u64 *__restrict__ skeys, const u32 *__restrict__ sperm, const u32 *__restrict__ rowof,
                                                             const double *__restrict__ val, const u64 *__restrict__ cnt, const u64 *__restrict__ ptr64,
                                                             const u64 *__restrict__ colstart, int *__restrict__ col, double *__restrict__ v,
                                                             long long n, long long n_strict) {
  const long long q = (long long)blockIdx.x * TPB + threadIdx.x;
  if (q >= n_strict) return;
  const long long j = (long long)skeys[q]; const u32 k = sperm[q];
  const long long dst = (long long)ptr64[j] + (long long)cnt[j] + (q - (long long)colstart[j]);
  col[dst] = (int)rowof[k]; v[dst] = val[k];
}

#define SPMV_ROWS_PER_BLOCK 4
// PROBE (measurement only, tools/bench_hci.py): every lane gathers from the same 512 bytes of x -- what the product would cost if the gather were free
template <int PROBE>
__global__ void __launch_bounds__(64 * SPMV_ROWS_PER_BLOCK) k_spmv_wave(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                                        const double *__restrict__ x, double *__restrict__ y, long long n) {
  const long long row = (long long)blockIdx.x * SPMV_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const int b = ptr[row], e = ptr[row + 1];
  double s = 0.0;
  for (int k = b + lane; k < e; k += 64) s += val[k] * x[PROBE ? (col[k] & 63) : col[k]];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) y[row] = s;
}

// The other layout SURVEY section 7 asks to be measured: the stored triangle only (12 B per stored entry instead of 24), the transposed
// half applied with native fp64 atomics, y zeroed in front.  Sums in an order that changes from run to run -- HCI's selection thresholds
// would see different eigenvector bits --, and 1.8 10^7 scattered atomics cost far more than the bytes they save:
// kept only as the measured alternative, behind SQMC_SPMV_UPPER_ATOMIC=1.
__global__ void __launch_bounds__(64 * SPMV_ROWS_PER_BLOCK) k_spmv_upper_atomic(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                                                const double *__restrict__ x, double *__restrict__ y, long long n) {
  const long long row = (long long)blockIdx.x * SPMV_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const int b = ptr[row], e = ptr[row + 1];
  const double xr = x[row];
  double s = 0.0;
  for (int k = b + lane; k < e; k += 64) {
    const int m = col[k]; const double a = val[k];
    s += a * x[m];
    if (m != (int)row) unsafeAtomicAdd(&y[m], a * xr);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) unsafeAtomicAdd(&y[row], s);
}

static void expand_full_csr(long long n, const int64_t *rc, const int64_t *idx, const double *val,
                            std::vector<int> &ptr, std::vector<int> &col, std::vector<double> &v) {
  std::vector<long long> cnt(n + 1, 0);
  long long k = 0;
  for (long long i = 0; i < n; i++) for (long long j = 0; j < rc[i]; j++, k++) { long long m = idx[k] - 1; cnt[i]++; if (m != i) cnt[m]++; }
  ptr.assign(n + 1, 0);
  for (long long i = 0; i < n; i++) ptr[i + 1] = ptr[i] + (int)cnt[i];
  col.resize(ptr[n]); v.resize(ptr[n]);
  std::vector<int> fill(ptr.begin(), ptr.end() - 1);
  k = 0;
  for (long long i = 0; i < n; i++) for (long long j = 0; j < rc[i]; j++, k++) {   // k ascending == reference accumulation order
    long long m = idx[k] - 1;
    col[fill[i]] = (int)m; v[fill[i]++] = val[k];
    if (m != i) { col[fill[m]] = (int)i; v[fill[m]++] = val[k]; }
  }
}
