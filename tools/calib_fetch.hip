// calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the walker step.
// MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced streaming reads; "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Each kernel below moves a KNOWN
// number of bytes; run under `rocprofv3 --pmc FETCH_SIZE` and divide (tools/profile_bench.sh does, -> profiles/*_traffic.json).
//   k_stream16   16 B per lane, coalesced            (the guide's reference pattern: expect bytes / FETCH = 2)
//   k_stream8     8 B per lane, coalesced            (keys, weights, determinants of resident walkers)
//   k_stream4     4 B per lane, coalesced            (flag words)
//   k_gather32   one 32-byte record per lane at a random index (spawn records gathered through the sort permutation)
//   k_gather8    one  8-byte word per lane at a random index
// build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <random>
typedef unsigned long long u64;
struct __attribute__((aligned(32))) Rec { u64 a, b, c, d; };
__global__ void k_stream16(const ulonglong2 *in, u64 *out, long long n) { long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) { ulonglong2 v = in[i]; if ((v.x ^ v.y) == 0x123456789ull) out[0] = v.x; } }
__global__ void k_stream8(const u64 *in, u64 *out, long long n) { long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) { u64 v = in[i]; if (v == 0x123456789ull) out[0] = v; } }
__global__ void k_stream4(const unsigned *in, u64 *out, long long n) { long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) { unsigned v = in[i]; if (v == 0x12345678u) out[0] = v; } }
__global__ void k_gather32(const Rec *tab, const unsigned *idx, u64 *out, long long n) { long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) { Rec r = tab[idx[i]]; if ((r.a ^ r.b ^ r.c ^ r.d) == 0x123456789ull) out[0] = r.a; } }
__global__ void k_gather8(const u64 *tab, const unsigned *idx, u64 *out, long long n) { long long i = (long long)blockIdx.x * 256 + threadIdx.x; if (i < n) { u64 v = tab[idx[i]]; if (v == 0x123456789ull) out[0] = v; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const long long NB = 1ll << 28;                 // 256 MiB table
  const long long NG = 1ll << 22;                 // 4 Mi gathers / stream elements per launch
  void *tab; unsigned *idx; u64 *out;
  CK(hipMalloc(&tab, NB)); CK(hipMemset(tab, 1, NB)); CK(hipMalloc(&idx, NG * 4)); CK(hipMalloc(&out, 64));
  std::vector<unsigned> h(NG); std::mt19937_64 g(5);
  const unsigned nrec = (unsigned)(NB / 32);
  std::vector<unsigned> perm(nrec); for (unsigned i = 0; i < nrec; i++) perm[i] = i;
  std::shuffle(perm.begin(), perm.end(), g);
  for (long long i = 0; i < NG; i++) h[i] = perm[i];      // distinct records: no gather hits a line another one brought in (beyond chance neighbours)
  CK(hipMemcpy(idx, h.data(), NG * 4, hipMemcpyHostToDevice));
  const int nb = (int)(NG / 256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream16, dim3(nb), dim3(256), 0, 0, (const ulonglong2 *)tab, out, NG);
    hipLaunchKernelGGL(k_stream8, dim3(nb), dim3(256), 0, 0, (const u64 *)tab, out, NG);
    hipLaunchKernelGGL(k_stream4, dim3(nb), dim3(256), 0, 0, (const unsigned *)tab, out, NG);
    hipLaunchKernelGGL(k_gather32, dim3(nb), dim3(256), 0, 0, (const Rec *)tab, idx, out, NG);
    hipLaunchKernelGGL(k_gather8, dim3(nb), dim3(256), 0, 0, (const u64 *)tab, idx, out, NG);
    CK(hipDeviceSynchronize());
  }
  // known bytes per launch (index array of the gathers included: 4 B per lane, coalesced)
  printf("known_bytes k_stream16 %lld\nknown_bytes k_stream8 %lld\nknown_bytes k_stream4 %lld\nknown_bytes k_gather32 %lld (+%lld index)\nknown_bytes k_gather8 %lld (+%lld index)\n",
         NG * 16, NG * 8, NG * 4, NG * 32, NG * 4, NG * 8, NG * 4);
  return 0;
}
