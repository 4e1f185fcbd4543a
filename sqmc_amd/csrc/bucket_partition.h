// bucket_partition.h -- block-local partition of a step's children into key ranges ("buckets"): the first half of the
// short-list annihilation tail described in bucket_kernels.h.  Textually included by sqmc_gpu.hip in front of walk_kernels.h
// (k_spawn partitions its own children as it emits them); not a standalone header.
#pragma once

#define BK_T 256
#define BK_PART_LDS ((1 + BK_T / 64) * BK_MAXB * 4)     // dynamic LDS of a kernel that partitions its block's children: splitters + per-wave counters
#define BK_CAP_S 2560                 // spawns of one bucket
#define BK_CAP_R 1536                 // residents of one bucket
#define BK_CAP_T 3584                 // both
#define BK_CAP_ROWS 2560              // partition blocks (256 children each)
#define BK_MAXB 1024
#define BK_TARGET 1150                // slots per bucket the host aims at (B = nall / BK_TARGET, at most one block per CU while that holds)
#define BK_STOP 0x80000000u           // in the merged-order array (source index: residents [0, R), sorted spawns [R, R + S)): this slot starts a run

// What the bucket tail does itself instead of the side-stream kernels when `on`: death/clone of every resident outside the
// deterministic space (k_diag's multiplication; the H_ii are all cached by then) and the last line of the deterministic projection
// of the residents inside it, w += (A x)(row) + E_T tau x(row) (k_prj_apply's), with A x already there.  x_in = the deterministic-space weights by row as the
// LAST step left them (written by that step's bucket tail into x_out, with every such walker's row in WalkArr::irk).
struct FusedSide { int on; const double *y; const double *x_in; double *x_out; };      // y = A x_in, row by row, from the spare blocks of k_spawn (PrjPre)
// the matrix-vector part of the projection, which needs nothing the host still has to decide: computed by spare blocks of k_spawn
struct PrjPre { int n_imp; const int *ptr, *col; const double *val; const double *x; double *y; };

struct BucketArgs {
  int B, nsb;                          // buckets, partition blocks
  u64 *words;                          // nsb x 256 sort words of the children, grouped by bucket inside each block
  unsigned short *segoff;              // nsb rows of B+1 group offsets
  u64 *state; u32 *ticket;             // look-back over the buckets
  int force_retry;                     // tests: behave as if a bucket did not fit
  // Where the buckets begin in the sorted resident list.  frac == null: at b n0 / B (equal numbers of residents).  Otherwise at
  // frac[b] n0 / 2^32: boundaries that equalise residents + spawns per bucket as measured two steps back (the spawns crowd on
  // the heavy determinants: with equal residents the fullest bucket held 3x the mean, and every bucket waits for it).
  const u32 *frac;
  u32 *scount;                         // out, B + 1 words: spawns of every bucket, then n0 -- what the next boundaries are made from
  const u32 *frac_prev; u32 *frac_out; // k_spawn's spare block: boundaries for the step after this one from scount and the boundaries it was counted with
};
#define BK_REBAL_MAXB 256
#define BK_REBAL_UNIFORM 0.125         // share of the equal-residents boundaries in the blend (no bucket narrower than 1/8 of its equal share)
__device__ __forceinline__ long long bk_bound(const BucketArgs &ba, int b, long long n0) {
  if (b <= 0) return 0;
  if (b >= ba.B) return n0;
  return ba.frac ? (long long)(((u64)ba.frac[b] * (u64)n0) >> 32) : ((long long)b * n0) / ba.B;
}
// one block of BK_T threads: new boundaries such that every bucket holds the same number of residents + spawns, if the spawns
// fall as they did when scount was taken (piecewise-constant density inside the old buckets); reads everything before it writes
__device__ __forceinline__ void bk_rebalance_block(const u32 *__restrict__ fprev, const u32 *__restrict__ scount, int B, u32 *__restrict__ fout) {
  __shared__ double s_fr[BK_REBAL_MAXB + 1], s_pre[BK_REBAL_MAXB + 1];
  const int tid = threadIdx.x;
  const double n0 = (double)scount[B];
  for (int b = tid; b <= B; b += BK_T) s_fr[b] = (b == 0) ? 0.0 : (b == B ? 1.0 : (fprev ? (double)fprev[b] * (1.0 / 4294967296.0) : (double)b / (double)B));
  __syncthreads();
  if (tid == 0) {                      // 256 additions beside a kernel that runs for tens of microseconds
    double acc = 0.0;
    for (int b = 0; b < B; b++) { s_pre[b] = acc; acc += (s_fr[b + 1] - s_fr[b]) * n0 + (double)scount[b]; }
    s_pre[B] = acc;
  }
  __syncthreads();
  const double total = s_pre[B];
  for (int j = tid; j < B; j += BK_T) {
    if (j == 0) { fout[0] = 0u; continue; }
    const double target = total * (double)j / (double)B;
    int lo = 0, hi = B;                 // s_pre[lo] <= target < s_pre[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= target) lo = mid; else hi = mid; }
    const double width = s_pre[lo + 1] - s_pre[lo];
    const double x = width > 0.0 ? (target - s_pre[lo]) / width : 0.0;
    const double bal = s_fr[lo] + x * (s_fr[lo + 1] - s_fr[lo]);
    double v = (1.0 - BK_REBAL_UNIFORM) * bal + BK_REBAL_UNIFORM * (double)j / (double)B;
    v = v * 4294967296.0;
    fout[j] = v >= 4294967295.0 ? 4294967295u : (u32)v;
  }
}

// ------------------------------------------------------------------------------------------------ partition
// The 256 children of one block, grouped by bucket (stable) behind the block's row of group offsets.  spl[b] (b >= 1) = first
// key of bucket b, staged by the caller; wcnt = BK_T/64 x BK_MAXB zeroed counters.  Every thread of the block calls it.
__device__ __forceinline__ void bucket_partition_block(const u32 *__restrict__ spl, u32 (*__restrict__ wcnt)[BK_MAXB], bool valid, u32 key, u64 word, long long blk,
                                                       const BucketArgs &ba) {
  const int B = ba.B, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int bkt = 0;
  if (valid) {                         // the last bucket whose first key is <= key (bucket 0 starts at -infinity)
    int lo = 0, hi = B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (spl[mid] <= key) lo = mid; else hi = mid; }
    bkt = lo;
  }
  // stable rank among the children of the block that go to the same bucket: ballots inside the wave, counters across waves
  u64 same = __ballot(valid);
  int nbit = 1; while ((1 << nbit) < B) nbit++;
  for (int q = 0; q < nbit; q++) { const u64 m = __ballot((bkt >> q) & 1); same &= ((bkt >> q) & 1) ? m : ~m; }
  const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const u32 rank = (u32)__popcll(same & lt);
  if (valid && rank == 0) wcnt[wv][bkt] = (u32)__popcll(same);
  __syncthreads();
  // group offsets: exclusive scan over the buckets of the four waves' counts; wcnt becomes the base of each wave's share
  constexpr int PER = BK_MAXB / BK_T;
  u32 t4[PER]; u64 sum = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) { const int b = tid * PER + q; u32 s = 0; if (b < B) for (int v = 0; v < BK_T / 64; v++) s += wcnt[v][b]; t4[q] = s; sum += s; }
  u64 tot; u32 ex = (u32)block_excl_scan_u64(sum, &tot);
  unsigned short *row = ba.segoff + blk * (B + 1);
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int b = tid * PER + q;
    if (b < B) { row[b] = (unsigned short)ex; u32 a = ex; for (int v = 0; v < BK_T / 64; v++) { const u32 cn = wcnt[v][b]; wcnt[v][b] = a; a += cn; } }
    ex += t4[q];
  }
  if (tid == 0) row[B] = (unsigned short)tot;
  __syncthreads();
  if (valid) ba.words[blk * BK_T + wcnt[wv][bkt] + rank] = word;
}
// splitters and zeroed counters of a partition block (no barrier inside: the caller synchronises once before partitioning)
__device__ __forceinline__ void bucket_partition_stage(u32 *__restrict__ spl, u32 (*__restrict__ wcnt)[BK_MAXB], const u64 *__restrict__ rkeys, long long n0, const BucketArgs &ba) {
  const int B = ba.B;
  for (int b = threadIdx.x; b < B; b += BK_T) spl[b] = b ? (u32)(rkeys[bk_bound(ba, b, n0)] >> 32) : 0u;      // first key of bucket b
  for (int d = threadIdx.x; d < (BK_T / 64) * BK_MAXB; d += BK_T) (&wcnt[0][0])[d] = 0;
}
