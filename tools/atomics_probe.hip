// Probe for DESIGN section 10 item 0b: what do the per-tile digit histograms of a radix pass cost when they are counted with global
// atomics (no return value) by the kernel that already holds the keys, against the LDS histogram kernel of its own?
//   hipcc --offload-arch=gfx950 -O3 tools/atomics_probe.hip -o /tmp/atomics_probe && /tmp/atomics_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (a) the stand-alone kernel: a block per tile of 4096 keys, LDS counters, strided write (what rs_hist_kernel does)
template <int BITS>
__global__ void __launch_bounds__(256) hist_lds(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  __shared__ u32 cnt[R];
  for (int d = threadIdx.x; d < R; d += 256) cnt[d] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * 4096;
  u64 k[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; if (i < n) atomicAdd(&cnt[(k[r] >> shift) & (R - 1)], 1u); }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += 256) hist[(long long)d * ntiles + blockIdx.x] = cnt[d];
}
// (a') the same tile by 1024 threads, four keys each: a shorter chain per thread
template <int BITS>
__global__ void __launch_bounds__(1024) hist_lds_wide(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  __shared__ u32 cnt[R];
  const long long base = (long long)blockIdx.x * 4096;
  u64 k[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { const long long i = base + r * 1024 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
  for (int d = threadIdx.x; d < R; d += 1024) cnt[d] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) { const long long i = base + r * 1024 + threadIdx.x; if (i < n) atomicAdd(&cnt[(k[r] >> shift) & (R - 1)], 1u); }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += 1024) hist[(long long)d * ntiles + blockIdx.x] = cnt[d];
}
// (a'') 256 threads, the loads issued before the counters are cleared
template <int BITS>
__global__ void __launch_bounds__(256) hist_lds_early(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  __shared__ u32 cnt[R];
  const long long base = (long long)blockIdx.x * 4096;
  u64 k[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
  for (int d = threadIdx.x; d < R; d += 256) cnt[d] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; if (i < n) atomicAdd(&cnt[(k[r] >> shift) & (R - 1)], 1u); }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += 256) hist[(long long)d * ntiles + blockIdx.x] = cnt[d];
}
// (a''') what the strided write costs: the same kernel writing its counters as one row (hist[tile][digit]) -- not a layout the scan
//        and the scatter can use as they are, a bound
template <int BITS>
__global__ void __launch_bounds__(256) hist_lds_row(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  __shared__ u32 cnt[R];
  const long long base = (long long)blockIdx.x * 4096;
  u64 k[16];
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
  for (int d = threadIdx.x; d < R; d += 256) cnt[d] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; r++) { const long long i = base + r * 256 + threadIdx.x; if (i < n) atomicAdd(&cnt[(k[r] >> shift) & (R - 1)], 1u); }
  __syncthreads();
  for (int d = threadIdx.x; d < R; d += 256) hist[(long long)blockIdx.x * R + d] = cnt[d];
}
// (a4) G consecutive tiles per block of 1024 threads, counters of all G in LDS, written G at a time: hist[digit][tile .. tile+G) is one
//      piece of 4 G bytes instead of G pieces a stride apart -- the layout of the scan and the scatter stays
template <int BITS, int G>
__global__ void __launch_bounds__(1024) hist_lds_group(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  __shared__ u32 cnt[R * G];                                   // [digit][g]
  const int t0 = blockIdx.x * G;
  for (int d = threadIdx.x; d < R * G; d += 1024) cnt[d] = 0;
  __syncthreads();
  constexpr int B = G < 4 ? G : 4;                             // tiles whose loads are in flight together (4 keys a thread and tile)
  for (int g0 = 0; g0 < G; g0 += B) {
    u64 k[B][4];
#pragma unroll
    for (int g = 0; g < B; g++)
#pragma unroll
      for (int r = 0; r < 4; r++) { const long long i = (long long)(t0 + g0 + g) * 4096 + r * 1024 + threadIdx.x; k[g][r] = i < n ? keys[i] : ~0ull; }
#pragma unroll
    for (int g = 0; g < B; g++)
#pragma unroll
      for (int r = 0; r < 4; r++) { const long long i = (long long)(t0 + g0 + g) * 4096 + r * 1024 + threadIdx.x; if (i < n) atomicAdd(&cnt[((k[g][r] >> shift) & (R - 1)) * G + g0 + g], 1u); }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * G; e += 1024) { const int d = e / G, g = e % G; if (t0 + g < ntiles) hist[(long long)d * ntiles + t0 + g] = cnt[e]; }
}
// (b) the keys are in registers anyway (a scatter pass, k_spawn): one global atomic per key on hist[digit][tile of dst]
//     dst = a random permutation target, as after a scatter on other bits
template <int BITS>
__global__ void __launch_bounds__(256) hist_global(const u64 *__restrict__ keys, const u32 *__restrict__ dst, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  const long long base = (long long)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const long long i = base + r * 256 + threadIdx.x;
    if (i < n) { const u64 k = keys[i]; const u32 t = dst[i] >> 12; atomicAdd(&hist[(long long)((k >> shift) & (R - 1)) * ntiles + t], 1u); }
  }
}
// (c) the same with the tile of the key's own position (k_spawn's case: children of one block land in one tile)
template <int BITS>
__global__ void __launch_bounds__(256) hist_global_own(const u64 *__restrict__ keys, u32 *__restrict__ hist, long long n, int ntiles, int shift) {
  constexpr int R = 1 << BITS;
  const long long base = (long long)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const long long i = base + r * 256 + threadIdx.x;
    if (i < n) { const u64 k = keys[i]; atomicAdd(&hist[(long long)((k >> shift) & (R - 1)) * ntiles + blockIdx.x], 1u); }
  }
}

int main() {
  for (long long n : {1800000ll, 18000000ll}) {
    const int ntiles = (int)((n + 4095) / 4096);
    std::vector<u64> h(n); std::vector<u32> hd(n);
    u64 x = 88172645463325252ull;
    for (long long i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x & ((1ull << 28) - 1); hd[i] = (u32)((x >> 30) % (u64)n); }
    u64 *dk; u32 *dd, *dh;
    CHK(hipMalloc(&dk, n * 8)); CHK(hipMalloc(&dd, n * 4)); CHK(hipMalloc(&dh, (size_t)1024 * ntiles * 4));
    CHK(hipMemcpy(dk, h.data(), n * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dd, hd.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
      for (int w = 0; w < 3; w++) launch();
      CHK(hipDeviceSynchronize());
      float best = 1e9f;
      for (int rep = 0; rep < 10; rep++) {
        CHK(hipMemsetAsync(dh, 0, (size_t)1024 * ntiles * 4, 0));
        CHK(hipEventRecord(e0, 0)); launch(); CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("n = %lld  %-44s %8.2f us\n", n, name, best * 1e3);
    };
    timeit("LDS histogram kernel, 10 bits", [&] { hipLaunchKernelGGL(hist_lds<10>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram kernel, 9 bits", [&] { hipLaunchKernelGGL(hist_lds<9>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 10); });
    timeit("LDS histogram, 1024 threads, 10 bits", [&] { hipLaunchKernelGGL(hist_lds_wide<10>, dim3(ntiles), dim3(1024), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 1024 threads, 9 bits", [&] { hipLaunchKernelGGL(hist_lds_wide<9>, dim3(ntiles), dim3(1024), 0, 0, dk, dh, n, ntiles, 10); });
    timeit("LDS histogram, loads first, 10 bits", [&] { hipLaunchKernelGGL(hist_lds_early<10>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, loads first, 9 bits", [&] { hipLaunchKernelGGL(hist_lds_early<9>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 10); });
    timeit("LDS histogram, loads first, row write, 10 bits", [&] { hipLaunchKernelGGL(hist_lds_row<10>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 4 tiles a block, 10 bits", [&] { hipLaunchKernelGGL((hist_lds_group<10, 4>), dim3((ntiles + 3) / 4), dim3(1024), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 4 tiles a block, 9 bits", [&] { hipLaunchKernelGGL((hist_lds_group<9, 4>), dim3((ntiles + 3) / 4), dim3(1024), 0, 0, dk, dh, n, ntiles, 10); });
    timeit("LDS histogram, 2 tiles a block, 10 bits", [&] { hipLaunchKernelGGL((hist_lds_group<10, 2>), dim3((ntiles + 1) / 2), dim3(1024), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 8 tiles a block, 10 bits", [&] { hipLaunchKernelGGL((hist_lds_group<10, 8>), dim3((ntiles + 7) / 8), dim3(1024), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 16 tiles a block, 10 bits", [&] { hipLaunchKernelGGL((hist_lds_group<10, 16>), dim3((ntiles + 15) / 16), dim3(1024), 0, 0, dk, dh, n, ntiles, 0); });
    timeit("LDS histogram, 16 tiles a block, 9 bits", [&] { hipLaunchKernelGGL((hist_lds_group<9, 16>), dim3((ntiles + 15) / 16), dim3(1024), 0, 0, dk, dh, n, ntiles, 10); });
    timeit("global atomics, tile of a random destination", [&] { hipLaunchKernelGGL(hist_global<9>, dim3(ntiles), dim3(256), 0, 0, dk, dd, dh, n, ntiles, 10); });
    timeit("global atomics, tile of the own position", [&] { hipLaunchKernelGGL(hist_global_own<10>, dim3(ntiles), dim3(256), 0, 0, dk, dh, n, ntiles, 0); });
    CHK(hipFree(dk)); CHK(hipFree(dd)); CHK(hipFree(dh));
  }
  return 0;
}
