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

typedef unsigned long long u64;
typedef unsigned int u32;

#define SCAN_BLOCK 256
#define SCAN_ITEMS 8                       // elements per thread
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

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

// phase 1: per-tile sums
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(const u64 *__restrict__ in, u64 *__restrict__ tile_sums, long long n) {
  long long base = (long long)blockIdx.x * SCAN_TILE;
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    long long i = base + (long long)k * SCAN_BLOCK + threadIdx.x;
    if (i < n) s += in[i];
  }
  u64 tot; block_excl_scan_u64(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
// phase 2: single block scans the tile sums in place (exclusive), writes grand total
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles_kernel(u64 *__restrict__ tile_sums, int ntiles, u64 *__restrict__ total_out) {
  u64 carry = 0;
  for (int base = 0; base < ntiles; base += SCAN_BLOCK) {
    int i = base + threadIdx.x;
    u64 v = (i < ntiles) ? tile_sums[i] : 0, tot;
    u64 ex = block_excl_scan_u64(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}
// phase 3: per-tile exclusive scan + tile offset.  Thread-contiguous items keep order.
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, const u64 *__restrict__ tile_sums, long long n) {
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  u64 v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { long long i = base + k; v[k] = (i < n) ? in[i] : 0; s += v[k]; }
  u64 tot; u64 ex = block_excl_scan_u64(s, &tot) + tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { long long i = base + k; if (i < n) out[i] = ex; ex += v[k]; }
}

struct ScanWork { u64 *tile_sums; long long cap_tiles; };

// exclusive scan of n u64 values; total (optional) is written on device
static inline void device_excl_scan_u64(const u64 *in, u64 *out, long long n, u64 *total_out, ScanWork &w, hipStream_t st) {
  if (n <= 0) { if (total_out) hipMemsetAsync(total_out, 0, sizeof(u64), st); return; }
  int ntiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(ntiles), dim3(SCAN_BLOCK), 0, st, in, w.tile_sums, n);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, w.tile_sums, ntiles, total_out);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(ntiles), dim3(SCAN_BLOCK), 0, st, in, out, w.tile_sums, n);
}

// ------------------------------------------------------------------------ radix sort
#define RS_MAX_RADIX 1024                   // 10-bit digits at most
#define RS_WAVE_ITEMS 16                    // rounds of 64 keys per wave
#define RS_TILE (64 * RS_WAVE_ITEMS)        // 1024 keys per wavefront

// histogram: hist[digit * ntiles + tile]
template <int BITS>
__global__ void __launch_bounds__(64) rs_hist_kernel(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int RS_RADIX = 1 << BITS;
  __shared__ u32 cnt[RS_RADIX];
  const int lane = threadIdx.x;
  for (int d = lane; d < RS_RADIX; d += 64) cnt[d] = 0;
  __syncthreads();
  long long base = (long long)blockIdx.x * RS_TILE;
  u64 kreg[RS_WAVE_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_WAVE_ITEMS; r++) {          // all loads of the tile in flight at once
    long long i = base + (long long)r * 64 + lane;
    kreg[r] = (i < n) ? keys[i] : ~0ull;
  }
#pragma unroll
  for (int r = 0; r < RS_WAVE_ITEMS; r++) {
    long long i = base + (long long)r * 64 + lane;
    if (i < n) atomicAdd(&cnt[(kreg[r] >> shift) & (RS_RADIX - 1)], 1u);
  }
  __syncthreads();
  for (int d = lane; d < RS_RADIX; d += 64) hist[(long long)d * ntiles + blockIdx.x] = cnt[d];
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

template <int BITS>
__global__ void __launch_bounds__(64) rs_scatter_kernel(const u64 *__restrict__ kin, const u32 *__restrict__ vin,
                                                        u64 *__restrict__ kout, u32 *__restrict__ vout,
                                                        const u32 *__restrict__ hist, const u32 *__restrict__ rowtot,
                                                        long long n, int ntiles, int shift) {
  constexpr int RS_RADIX = 1 << BITS, RS_BITS = BITS, PER = RS_RADIX / 64;
  __shared__ u32 off[RS_RADIX];
  const int lane = threadIdx.x;
  {   // digit bases = exclusive scan of the row totals (PER per lane + wave scan)
    u32 t[PER]; u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { t[q] = rowtot[lane * PER + q]; sum += t[q]; }
    u32 ex = (u32)(wave_incl_scan_u64(sum, lane) - sum);
#pragma unroll
    for (int q = 0; q < PER; q++) { off[lane * PER + q] = ex; ex += t[q]; }
  }
  __syncthreads();
  for (int d = lane; d < RS_RADIX; d += 64) off[d] += hist[(long long)d * ntiles + blockIdx.x];
  __syncthreads();
  long long base = (long long)blockIdx.x * RS_TILE;
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  u64 kreg[RS_WAVE_ITEMS]; u32 vreg[RS_WAVE_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_WAVE_ITEMS; r++) {          // the whole tile goes to registers first
    long long i = base + (long long)r * 64 + lane;
    kreg[r] = (i < n) ? kin[i] : 0; vreg[r] = (i < n) ? vin[i] : 0;
  }
#pragma unroll
  for (int r = 0; r < RS_WAVE_ITEMS; r++) {
    long long i = base + (long long)r * 64 + lane;
    bool valid = i < n;
    u64 key = kreg[r]; u32 val = vreg[r];
    u32 dig = (u32)((key >> shift) & (RS_RADIX - 1));
    u64 same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; b++) {
      u64 m = __ballot((dig >> b) & 1);
      same &= ((dig >> b) & 1) ? m : ~m;
    }
    u32 rank = (u32)__popcll(same & lt), cnt = (u32)__popcll(same);
    u32 dst = 0;
    if (valid) dst = off[dig] + rank;
    __syncthreads();                     // all lanes have read off[] before leaders bump it
    if (valid && rank == 0) off[dig] += cnt;
    __syncthreads();
    if (valid) { kout[dst] = key; vout[dst] = val; }
  }
}

struct SortWork { u64 *k_alt; u32 *v_alt; u32 *hist; u32 *rowtot; long long cap; };

// Sorts keys[0..n) (with values) on bits [0, nbits); result ends up in the returned buffers
// (either the inputs or the alternates).  Stable.
template <int BITS>
static inline void radix_pass(u64 *ka, u32 *va, u64 *kb, u32 *vb, long long n, int ntiles, int shift, SortWork &w, hipStream_t st) {
  hipLaunchKernelGGL(rs_hist_kernel<BITS>, dim3(ntiles), dim3(64), 0, st, ka, w.hist, n, ntiles, shift);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1 << BITS), dim3(SCAN_BLOCK), 0, st, w.hist, w.rowtot, ntiles);
  hipLaunchKernelGGL(rs_scatter_kernel<BITS>, dim3(ntiles), dim3(64), 0, st, ka, va, kb, vb, w.hist, w.rowtot, n, ntiles, shift);
}
// digit width: the fewest passes of at most 10 bits, then the narrowest digit that still
// covers the key in that many passes (28-bit C2 keys: 3 passes of 10 bits)
static inline void device_radix_sort(u64 *&keys, u32 *&vals, long long n, int nbits, SortWork &w, hipStream_t st) {
  if (n <= 1) return;
  int ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
  u64 *ka = keys, *kb = w.k_alt; u32 *va = vals, *vb = w.v_alt;
  const int npass = (nbits + 9) / 10;
  int bits = (nbits + npass - 1) / npass; if (bits < 8) bits = 8;
  for (int pss = 0, shift = 0; pss < npass; pss++, shift += bits) {
    if (bits == 8) radix_pass<8>(ka, va, kb, vb, n, ntiles, shift, w, st);
    else if (bits == 9) radix_pass<9>(ka, va, kb, vb, n, ntiles, shift, w, st);
    else radix_pass<10>(ka, va, kb, vb, n, ntiles, shift, w, st);
    u64 *tk = ka; ka = kb; kb = tk; u32 *tv = va; va = vb; vb = tv;
  }
  w.k_alt = kb; w.v_alt = vb; keys = ka; vals = va;
}
