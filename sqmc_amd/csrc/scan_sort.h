// scan_sort.h -- device-wide exclusive scan (u64 lanes, used with packed 2x32-bit counters)
// and a stable LSD radix sort of (u64 key, u32 value) pairs, written for 64-wide
// wavefronts on gfx950.
//
// The sort replaces merge_sort2_up_dn (do_walk.f90:5411-5614): any stable sort on
// (up, dn) yields the reference's order.  One wavefront owns one tile; inside a tile the
// rank of a key among equal digits comes from 8 wave ballots (match-any) + popcount, so
// the scatter is stable without any cross-wave hand-off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef unsigned int u32;

#define SCAN_BLOCK 256
#ifndef SCAN_ITEMS_SMALL
#define SCAN_ITEMS_SMALL 8                 // elements per thread: small inputs want many tiles (latency) ...
#endif
#define SCAN_ITEMS_LARGE 16                // ... large ones few (the look-back chain grows with the number of tiles)
#define SCAN_LARGE_N (1ll << 19)
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS_SMALL)     // state words are provisioned for the small tile

__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u64 o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix,
// total in *total (valid for all threads)
__device__ __forceinline__ u64 block_excl_scan_u64(u64 v, u64 *total) {
  __shared__ u64 wsum[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  u64 inc = wave_incl_scan_u64(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  u64 off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_BLOCK / 64; i++) { if (i < w) off += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

// Single-pass exclusive scan with decoupled look-back.  Tiles are handed out by an atomic
// ticket, so every predecessor of a tile is already running or finished (forward progress
// without co-residency assumptions).  One 64-bit word per tile carries {status:2, value:62};
// it is published and polled with device-scope atomics only, so no fence pairs with the
// payload and the non-coherent per-XCD L2s are never in the way.  Values must stay < 2^62
// (two packed counters < 2^30 each).  state[] and ticket must be zero on entry; the caller
// re-zeroes them (the walker step does it inside k_finish for the next step).
#define SCAN_ST_AGG (1ull << 62)
#define SCAN_ST_INC (2ull << 62)
#define SCAN_VAL_MASK ((1ull << 62) - 1ull)
// Called by the 64 lanes of ONE wavefront of the block that owns `tile`: publishes the tile's
// aggregate, sums the aggregates of all earlier tiles (64 predecessors per poll, stopping at the
// first one that already carries an inclusive value), publishes the tile's inclusive value and
// returns the exclusive prefix in every lane.
__device__ __forceinline__ u64 lookback_exclusive(u64 *__restrict__ state, u32 tile, u64 tot, int lane) {
  u64 excl = 0;
  if (tile == 0) { if (lane == 0) atomicExch((unsigned long long *)&state[0], SCAN_ST_INC | tot); return 0; }
  if (lane == 0) atomicExch((unsigned long long *)&state[tile], SCAN_ST_AGG | tot);
  long long p = (long long)tile - 1;
  while (true) {
    const long long idx = p - lane;
    const u64 w = (idx >= 0) ? atomicAdd((unsigned long long *)&state[idx], 0ull) : SCAN_ST_INC;
    const u64 st = w >> 62;
    const u64 inc_mask = __ballot(st == 2), zero_mask = __ballot(st == 0);
    const int first_inc = inc_mask ? __builtin_ctzll(inc_mask) : 64;
    const u64 need = (first_inc >= 63) ? ~0ull : ((2ull << first_inc) - 1ull);
    if (zero_mask & need) continue;           // some needed predecessor has not published yet
    u64 val = (lane <= first_inc) ? (w & SCAN_VAL_MASK) : 0;
    for (int o = 32; o > 0; o >>= 1) val += __shfl_down(val, o, 64);
    excl += __shfl(val, 0, 64);
    if (first_inc < 64) break;
    p -= 64;
  }
  if (lane == 0) atomicExch((unsigned long long *)&state[tile], SCAN_ST_INC | (excl + tot));
  return excl;
}
// Extra: work of one additional block (the last of the grid) that has nothing to do with the scan and only wants to
// run beside it; ScanNoExtra: none.
struct ScanNoExtra { static constexpr bool on = false; __device__ void operator()() const {} };
template <int SCAN_ITEMS, class Extra = ScanNoExtra>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_lookback_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, long long n_arg,
                                                                   u64 *__restrict__ state, u32 *__restrict__ ticket, u64 *__restrict__ total_out,
                                                                   const u64 *__restrict__ n_dev, Extra extra, const int *__restrict__ skip) {
  constexpr int TILE_ELEMS = SCAN_BLOCK * SCAN_ITEMS;
  if (Extra::on && blockIdx.x == gridDim.x - 1) { extra(); return; }
  if (skip && *skip) return;                    // the caller's "do nothing" switch (set before this kernel was enqueued behind its producer)
  // n_dev: the length lives in device memory (the launch was sized for an upper bound n_arg by a host
  // that does not know it yet); tiles past the end scan zeros and repeat the total
  const long long n = n_dev ? (long long)(*n_dev & 0xFFFFFFFFull) : n_arg;
  __shared__ u32 s_tile; __shared__ u64 s_excl;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  // Every wavefront owns a contiguous run of 64*ITEMS elements and reads it row by row (lane l takes
  // element r*64 + l: 512 contiguous bytes per load instruction); rows are scanned with wave
  // shuffles and chained through the row totals, waves through LDS.
  __shared__ u64 s_wsum[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long base = (long long)tile * TILE_ELEMS + (long long)wv * (64 * SCAN_ITEMS) + lane;
  u64 v[SCAN_ITEMS], inc[SCAN_ITEMS], carry = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { long long i = base + (long long)k * 64; v[k] = (i < n) ? in[i] : 0; }
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const u64 x = wave_incl_scan_u64(v[k], lane);
    inc[k] = x + carry;
    carry += __shfl(x, 63, 64);
  }
  if (lane == 0) s_wsum[wv] = carry;
  __syncthreads();
  u64 ex = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < SCAN_BLOCK / 64; q++) { if (q < wv) ex += s_wsum[q]; tot += s_wsum[q]; }
  if (threadIdx.x < 64) {                       // wave 0 publishes and looks back, 64 predecessors per poll
    const u64 excl = lookback_exclusive(state, tile, tot, threadIdx.x);
    if (threadIdx.x == 0) {
      s_excl = excl;
      if (total_out && (long long)(tile + 1) * TILE_ELEMS >= n) *total_out = excl + tot;
    }
  }
  __syncthreads();
  ex += s_excl;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { long long i = base + (long long)k * 64; if (i < n) out[i] = ex + inc[k] - v[k]; }
}
__global__ void scan_clear_kernel(u64 *state, u32 *ticket, int ntiles) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ntiles) state[i] = 0;
  if (i == 0) *ticket = 0;
}

struct ScanWork { u64 *state; u32 *ticket; long long cap_tiles; bool self_clear; };

// exclusive scan of n u64 values (< 2^62 in total); total (optional) is written on device
// returns the number of look-back words (tiles) the launch may touch
template <class Extra = ScanNoExtra>
static inline int device_excl_scan_u64(const u64 *in, u64 *out, long long n, u64 *total_out, ScanWork &w, hipStream_t st, const u64 *n_dev = nullptr, Extra extra = Extra(),
                                       const int *skip = nullptr) {
  if (n <= 0 && !Extra::on) { if (total_out) hipMemsetAsync(total_out, 0, sizeof(u64), st); return 0; }
  if (n <= 0) n = 1;
  const bool large = n >= SCAN_LARGE_N;
  const long long tile = (long long)SCAN_BLOCK * (large ? SCAN_ITEMS_LARGE : SCAN_ITEMS_SMALL);
  int ntiles = (int)((n + tile - 1) / tile);
  if (w.self_clear) hipLaunchKernelGGL(scan_clear_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, st, w.state, w.ticket, ntiles);
  const int grid = ntiles + (Extra::on ? 1 : 0);
  if (large) hipLaunchKernelGGL((scan_lookback_kernel<SCAN_ITEMS_LARGE, Extra>), dim3(grid), dim3(SCAN_BLOCK), 0, st, in, out, n, w.state, w.ticket, total_out, n_dev, extra, skip);
  else hipLaunchKernelGGL((scan_lookback_kernel<SCAN_ITEMS_SMALL, Extra>), dim3(grid), dim3(SCAN_BLOCK), 0, st, in, out, n, w.state, w.ticket, total_out, n_dev, extra, skip);
  return ntiles;
}

// ------------------------------------------------------------------------ radix sort
#define RS_MAX_RADIX 1024                   // 10-bit digits at most
#define RS_LARGE_N (1ll << 20)
#ifndef RS_TILE
#define RS_TILE 1024                        // keys per 256-thread block (4 waves x 4 rounds of 64)
#endif

// histogram: hist[digit * ntiles + tile]; one 256-thread block (4 waves) per tile of 256 * ROUNDS keys (1024; 4096 for long lists: a
// quarter of the tiles, so that the per-tile histogram matrix of a 10-bit digit is no larger than an 8-bit digit's over small tiles)
template <int BITS, int ROUNDS>
__global__ void __launch_bounds__(256) rs_hist_kernel(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int RS_RADIX = 1 << BITS, TILE = 256 * ROUNDS;
  __shared__ u32 cnt[RS_RADIX];
  for (int d = threadIdx.x; d < RS_RADIX; d += 256) cnt[d] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * TILE;
  u64 kreg[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) { long long i = base + r * 256 + threadIdx.x; kreg[r] = (i < n) ? keys[i] : ~0ull; }
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) { long long i = base + r * 256 + threadIdx.x; if (i < n) atomicAdd(&cnt[(kreg[r] >> shift) & (RS_RADIX - 1)], 1u); }
  __syncthreads();
  for (int d = threadIdx.x; d < RS_RADIX; d += 256) hist[(long long)d * ntiles + blockIdx.x] = cnt[d];
}

// one block per digit: in-place exclusive scan of that digit's row (ntiles entries) and the
// row total; the scatter kernel turns the 256 row totals into digit bases itself.
__global__ void __launch_bounds__(SCAN_BLOCK) rs_scan_kernel(u32 *__restrict__ hist, u32 *__restrict__ rowtot, int ntiles) {
  u32 *row = hist + (long long)blockIdx.x * ntiles;
  u64 carry = 0;
  for (int base = 0; base < ntiles; base += SCAN_BLOCK * 4) {
    int i0 = base + threadIdx.x * 4;
    u32 v[4]; u64 s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (i0 + k < ntiles) ? row[i0 + k] : 0; s += v[k]; }
    u64 tot; u64 ex = carry + block_excl_scan_u64(s, &tot);
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k < ntiles) row[i0 + k] = (u32)ex; ex += v[k]; }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) rowtot[blockIdx.x] = (u32)carry;
}
// the same for short rows (ntiles <= 64 * RS_ROW_PER_LANE): one WAVE per digit row, a lane takes consecutive entries (their loads in
// flight together), one shuffle scan of the lanes' sums -- no barrier, a quarter of the blocks
#define RS_ROW_PER_LANE 16
__global__ void __launch_bounds__(SCAN_BLOCK) rs_scan_rows_wave(u32 *__restrict__ hist, u32 *__restrict__ rowtot, int ntiles, int nrows) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * (SCAN_BLOCK / 64) + (threadIdx.x >> 6);
  if (r >= nrows) return;
  u32 *row = hist + (long long)r * ntiles;
  const int per = (ntiles + 63) / 64, i0 = lane * per;
  u32 v[RS_ROW_PER_LANE]; u32 s = 0;
#pragma unroll
  for (int k = 0; k < RS_ROW_PER_LANE; k++) { v[k] = (k < per && i0 + k < ntiles) ? row[i0 + k] : 0u; s += v[k]; }
  u32 inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
  u32 ex = inc - s;
#pragma unroll
  for (int k = 0; k < RS_ROW_PER_LANE; k++) { if (k < per && i0 + k < ntiles) row[i0 + k] = ex; ex += v[k]; }
  if (lane == 63) rowtot[r] = inc;
}

// Stable scatter: a 256-thread block owns a 1024-key tile; wave w owns keys [256w, 256w+256)
// in 4 rounds of 64.  Rank among equal digits inside a round = popc(match & lanemask_lt) from
// BITS ballots; per-wave digit counters live in LDS (private to the wave, in-order LDS queue),
// then the four waves' counters are prefix-added once.  Only 4 sequential rounds per wave.
template <int BITS, bool VALS, int ROUNDS>
__global__ void __launch_bounds__(256) rs_scatter_kernel(const u64 *__restrict__ kin, const u32 *__restrict__ vin,
                                                         u64 *__restrict__ kout, u32 *__restrict__ vout,
                                                         const u32 *__restrict__ hist, const u32 *__restrict__ rowtot,
                                                         long long n, int ntiles, int shift) {
  constexpr int RS_RADIX = 1 << BITS, PER = RS_RADIX / 256, TILE = 256 * ROUNDS;
  __shared__ u32 off[RS_RADIX];            // global base of each digit for this tile
  __shared__ u32 wcnt[4][RS_RADIX];        // per-wave digit counts -> per-wave bases
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the tile's keys are requested first: their latency hides behind the digit-base prologue, which has its own
  // chain of dependent loads (row totals -> block scan -> this tile's row prefixes)
  const long long base = (long long)blockIdx.x * TILE + wv * (64 * ROUNDS);
  u64 kreg[ROUNDS]; u32 vreg[ROUNDS], rnk[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) { long long i = base + r * 64 + lane; kreg[r] = (i < n) ? kin[i] : 0; vreg[r] = (VALS && i < n) ? vin[i] : 0; }
  {   // digit bases = exclusive scan of the row totals (PER per thread + block scan) + this tile's row prefix
    u32 t[PER]; u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { t[q] = rowtot[tid * PER + q]; sum += t[q]; }
    u64 tot; u32 ex = (u32)block_excl_scan_u64(sum, &tot);
#pragma unroll
    for (int q = 0; q < PER; q++) { off[tid * PER + q] = ex + hist[(long long)(tid * PER + q) * ntiles + blockIdx.x]; ex += t[q]; }
  }
  for (int d = tid; d < 4 * RS_RADIX; d += 256) (&wcnt[0][0])[d] = 0;
  __syncthreads();
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long long i = base + r * 64 + lane;
    const bool valid = i < n;
    const u32 dig = (u32)((kreg[r] >> shift) & (RS_RADIX - 1));
    u64 same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) { u64 m = __ballot((dig >> b) & 1); same &= ((dig >> b) & 1) ? m : ~m; }
    const u32 rank = (u32)__popcll(same & lt), cnt = (u32)__popcll(same);
    u32 prev = 0;
    if (valid) prev = wcnt[wv][dig];
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) wcnt[wv][dig] = prev + cnt;
    __builtin_amdgcn_wave_barrier();
    rnk[r] = prev + rank;
  }
  __syncthreads();
  for (int d = tid; d < RS_RADIX; d += 256) {   // turn per-wave counts into per-wave bases
    u32 b = off[d];
#pragma unroll
    for (int q = 0; q < 4; q++) { u32 c = wcnt[q][d]; wcnt[q][d] = b; b += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const long long i = base + r * 64 + lane;
    if (i < n) {
      const u32 dig = (u32)((kreg[r] >> shift) & (RS_RADIX - 1));
      const u32 dst = wcnt[wv][dig] + rnk[r];
      kout[dst] = kreg[r]; if (VALS) vout[dst] = vreg[r];
    }
  }
}

struct SortWork { u64 *k_alt; u32 *v_alt; u32 *hist; u32 *rowtot; long long cap; };

// Sorts keys[0..n) (with values) on bits [0, nbits); result ends up in the returned buffers
// (either the inputs or the alternates).  Stable.
template <int BITS, int ROUNDS>
static inline void radix_pass(u64 *ka, u32 *va, u64 *kb, u32 *vb, long long n, int shift, SortWork &w, hipStream_t st) {
  const int ntiles = (int)((n + 256 * ROUNDS - 1) / (256 * ROUNDS));
  hipLaunchKernelGGL((rs_hist_kernel<BITS, ROUNDS>), dim3(ntiles), dim3(256), 0, st, ka, w.hist, n, ntiles, shift);
  static const bool rows_by_wave = !(getenv("SQMC_RS_SCAN_BLOCKS") && getenv("SQMC_RS_SCAN_BLOCKS")[0] == '1');
  if (rows_by_wave && ntiles <= 64 * RS_ROW_PER_LANE)
    hipLaunchKernelGGL(rs_scan_rows_wave, dim3(((1 << BITS) + SCAN_BLOCK / 64 - 1) / (SCAN_BLOCK / 64)), dim3(SCAN_BLOCK), 0, st, w.hist, w.rowtot, ntiles, 1 << BITS);
  else
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1 << BITS), dim3(SCAN_BLOCK), 0, st, w.hist, w.rowtot, ntiles);
  if (va) hipLaunchKernelGGL((rs_scatter_kernel<BITS, true, ROUNDS>), dim3(ntiles), dim3(256), 0, st, ka, va, kb, vb, w.hist, w.rowtot, n, ntiles, shift);
  else hipLaunchKernelGGL((rs_scatter_kernel<BITS, false, ROUNDS>), dim3(ntiles), dim3(256), 0, st, ka, va, kb, vb, w.hist, w.rowtot, n, ntiles, shift);
}
// digit width: the fewest passes of at most 10 bits, then the narrowest digit that still
// covers the key in that many passes (28-bit C2 keys: 3 passes of 10 bits).  vals may be null
// (keys only: the caller packed its payload into the bits below shift0, which are not sorted on
// and ride along -- one scattered 8-byte write per element and pass instead of an 8- and a 4-byte
// one).
// counts_wanted: the caller reads the digit totals of the (single) pass from w.rowtot afterwards, so a list of one
// element goes through the pass as well instead of returning untouched with the totals of some earlier sort there.
static inline void device_radix_sort(u64 *&keys, u32 *&vals, long long n, int nbits, SortWork &w, hipStream_t st, int shift0 = 0, bool counts_wanted = false) {
  if (n <= 0 || (n == 1 && !counts_wanted)) return;
  u64 *ka = keys, *kb = w.k_alt; u32 *va = vals, *vb = vals ? w.v_alt : nullptr;
  // Small inputs are launch bound: the fewest passes (10-bit digits).  Large ones are bound by the
  // per-tile histogram matrix, whose column accesses cost a whole memory sector per 4-byte count
  // and which grows with 2^bits: 8-bit digits there.
  static const int maxbits_env = getenv("SQMC_SORT_MAXBITS") ? atoi(getenv("SQMC_SORT_MAXBITS")) : 0;
  // long lists: tiles of 4096 keys (a quarter of the rows in the histogram matrix), so the digits can stay 10 bits wide there too:
  // 28-bit keys in three passes instead of four (SQMC_SORT_BIG_TILE=0: the 8-bit digits over 1024-key tiles of before)
  static const int big_tile_env = getenv("SQMC_SORT_BIG_TILE") ? atoi(getenv("SQMC_SORT_BIG_TILE")) : 1;      // (2: the long lists' tiles for every list -- tests)
  const bool big = big_tile_env == 2 || (n >= RS_LARGE_N && big_tile_env);
  const int maxbits = maxbits_env ? maxbits_env : ((n >= RS_LARGE_N && !big) ? 8 : 10);
  const int npass = (nbits + maxbits - 1) / maxbits;
  // digit widths as even as the templates allow (8, 9 or 10 bits), the wide ones first: 28 bits in three passes
  // are 10 + 9 + 9, not 10 + 10 + 10 -- every kernel of a pass is a little cheaper with half the bins
  int left = nbits;
  for (int pss = 0, shift = shift0; pss < npass; pss++) {
    int bits = (left + (npass - pss) - 1) / (npass - pss); if (bits < 8) bits = 8;
    left -= bits; if (left < 0) left = 0;
    const int used = bits;
    if (big) {
      if (bits == 8) radix_pass<8, 16>(ka, va, kb, vb, n, shift, w, st);
      else if (bits == 9) radix_pass<9, 16>(ka, va, kb, vb, n, shift, w, st);
      else radix_pass<10, 16>(ka, va, kb, vb, n, shift, w, st);
    } else {
      if (bits == 8) radix_pass<8, 4>(ka, va, kb, vb, n, shift, w, st);
      else if (bits == 9) radix_pass<9, 4>(ka, va, kb, vb, n, shift, w, st);
      else radix_pass<10, 4>(ka, va, kb, vb, n, shift, w, st);
    }
    shift += used;
    u64 *tk = ka; ka = kb; kb = tk; u32 *tv = va; va = vb; vb = tv;
  }
  w.k_alt = kb; if (vals) { w.v_alt = vb; vals = va; } keys = ka;
}

// ------------------------------------------------------------------------ merge of two sorted runs
// Stable merge on (word >> shift), A before B on equal keys (the walkers of a step are already in
// order: only the spawns need sorting, then one merge).  Merge path: a 256-thread block per tile
// of MP_TILE outputs finds the tile's split of A and B by a binary search on its diagonal, stages
// both segments in LDS, every thread finds its own diagonal there, merges MP_ITEMS outputs
// sequentially, and the tile leaves through LDS in coalesced rows.
#define MP_ITEMS 8
#define MP_TILE (256 * MP_ITEMS)
// number of A elements among the first d outputs of the merge
template <typename P>
__device__ __forceinline__ long long merge_path_split(P A, long long nA, P B, long long nB, long long d, int shift) {
  long long lo = d > nB ? d - nB : 0, hi = d < nA ? d : nA;
  while (lo < hi) {
    const long long i = (lo + hi) >> 1;
    if ((A[i] >> shift) <= (B[d - i - 1] >> shift)) lo = i + 1; else hi = i;      // A[i] precedes that B element: it is inside
  }
  return lo;
}
// VALS: keys and their 32-bit payloads are two arrays (keys wider than 32 bits); the tile is half as long so that the
// staged payloads fit beside the keys.
template <bool VALS>
__global__ void __launch_bounds__(256) merge_path_kernel(const u64 *__restrict__ A, const u32 *__restrict__ Av, long long nA,
                                                         const u64 *__restrict__ B, const u32 *__restrict__ Bv, long long nB,
                                                         u64 *__restrict__ out, u32 *__restrict__ outv, int shift) {
  constexpr int ITEMS = VALS ? MP_ITEMS / 2 : MP_ITEMS, TILE = 256 * ITEMS;
  __shared__ u64 sA[TILE], sB[TILE], sO[TILE];
  __shared__ u32 vA[VALS ? TILE : 1], vB[VALS ? TILE : 1], vO[VALS ? TILE : 1];
  __shared__ long long s_split[2];
  const long long n = nA + nB, d0 = (long long)blockIdx.x * TILE, d1 = (d0 + TILE < n) ? d0 + TILE : n;
  if (threadIdx.x == 0) s_split[0] = merge_path_split(A, nA, B, nB, d0, shift);
  if (threadIdx.x == 64) s_split[1] = merge_path_split(A, nA, B, nB, d1, shift);
  __syncthreads();
  const long long a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
  const int la = (int)(a1 - a0), lb = (int)(b1 - b0), lt = (int)(d1 - d0);
  for (int k = threadIdx.x; k < la; k += 256) { sA[k] = A[a0 + k]; if (VALS) vA[k] = Av[a0 + k]; }
  for (int k = threadIdx.x; k < lb; k += 256) { sB[k] = B[b0 + k]; if (VALS) vB[k] = Bv[b0 + k]; }
  __syncthreads();
  const int dd = threadIdx.x * ITEMS;
  if (dd < lt) {
    int i = (int)merge_path_split((const u64 *)sA, (long long)la, (const u64 *)sB, (long long)lb, (long long)dd, shift), j = dd - i;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (dd + k >= lt) break;
      const bool takeA = (j >= lb) || (i < la && (sA[i] >> shift) <= (sB[j] >> shift));
      sO[dd + k] = takeA ? sA[i] : sB[j];
      if (VALS) vO[dd + k] = takeA ? vA[i] : vB[j];
      if (takeA) i++; else j++;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < lt; k += 256) { out[d0 + k] = sO[k]; if (VALS) outv[d0 + k] = vO[k]; }
}
static inline void device_merge_sorted(const u64 *A, const u32 *Av, long long nA, const u64 *B, const u32 *Bv, long long nB, u64 *out, u32 *outv, int shift, hipStream_t st) {
  const long long n = nA + nB;
  if (n <= 0) return;
  if (outv) hipLaunchKernelGGL(merge_path_kernel<true>, dim3((unsigned)((n + MP_TILE / 2 - 1) / (MP_TILE / 2))), dim3(256), 0, st, A, Av, nA, B, Bv, nB, out, outv, shift);
  else hipLaunchKernelGGL(merge_path_kernel<false>, dim3((unsigned)((n + MP_TILE - 1) / MP_TILE)), dim3(256), 0, st, A, Av, nA, B, Bv, nB, out, outv, shift);
}
