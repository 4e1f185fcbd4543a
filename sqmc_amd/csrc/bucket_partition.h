// bucket_partition.h -- block-local partition of a step's children into key ranges ("buckets"): the first half of the
// short-list annihilation tail described in bucket_kernels.h.  Textually included by sqmc_gpu.hip in front of walk_kernels.h
// (k_spawn partitions its own children as it emits them); not a standalone header.
#pragma once

#define BK_T 256
#define BK_PART_STRIDE(B) (((B) + 63) & ~63)             // counters per wave of a partitioning block (ws below)
#define BK_PART_LDS(B) ((1 + BK_T / 64) * BK_PART_STRIDE(B) * 4)     // dynamic LDS of a kernel that partitions its block's children: splitters + per-wave counters, sized by the buckets there are
// Two sizes of bucket.  Default: one block of k_anneal_bucket per CU with the whole LDS.  -DBK_HALF=1: a bucket is half a CU's LDS and two
// blocks share a CU.  Measured in round 3: a bucket's chain is as long at 575 slots as at 1150 (it is barriers and
// dependent round trips, not elements), so twice as many buckets took as long each -- 42 us for the launch against 40 -- while the
// fullest bucket moved from 1.25x to 2.2x the mean and the partition rows doubled.  Kept as a switch for other machines.
#ifndef BK_HALF
#define BK_HALF 0
#endif
#if BK_HALF
#define BK_CAP_S 1024                 // spawns of one bucket
#define BK_CAP_R 768                  // residents of one bucket
#define BK_CAP_T 1536                 // both
#define BK_TARGET 575                 // slots per bucket the host aims at (B = nall / BK_TARGET, at most two blocks per CU while that holds)
#define BK_PER_CU 2
typedef unsigned short bk_cnt_t;      // digit counters of the in-LDS sort (a bucket's spawns fit 16 bits)
#else
#define BK_CAP_S 2560
#define BK_CAP_R 1536
#define BK_CAP_T 3584
#define BK_TARGET 1150                // one block per CU
#define BK_PER_CU 1
typedef u32 bk_cnt_t;
#endif
#define BK_RESIDENTS_MAX (BK_HALF ? BK_CAP_R * 80 / 100 : 1000)     // boundaries that would put more residents into a bucket are not used
#define BK_HQ_DEFER 64                // deferred H_ii positions per bucket (13 on average at the bench size; the rest is done in the tail)
#define BK_CAP_ROWS 2560              // partition blocks (256 children each)
#define BK_MAXB 1024
#define BK_STOP 0x80000000u           // in the merged-order array (source index: residents [0, R), sorted spawns [R, R + S)): this slot starts a run

// What the bucket tail does itself instead of the side-stream kernels when `on`: death/clone of every resident outside the
// deterministic space (k_diag's multiplication; the H_ii are all cached by then) and the last line of the deterministic projection
// of the residents inside it, w += (A x)(row) + E_T tau x(row) (k_prj_apply's), with A x already there.  x_in = the deterministic-space weights by row as the
// LAST step left them (written by that step's bucket tail into x_out, with every such walker's row in WalkArr::irk).
struct FusedSide { int on; const double *y; const double *x_in; double *x_out; };      // y = A x_in, row by row, from the spare blocks of k_spawn (PrjPre)
// the matrix-vector part of the projection, which needs nothing the host still has to decide: computed by spare blocks of k_spawn
struct PrjPre { int n_imp; const int *ptr, *col; const double *val; const double *x; double *y;
                const int *grow; };      // grow != null (sharded steps): y[i] = row grow[i] of A times x, for the n_imp rows this rank owns

struct BucketArgs {
  int B, nsb;                          // buckets, partition blocks
  u64 *words;                          // nsb x 256 sort words of the children, grouped by bucket inside each block
  unsigned short *segoff;              // nsb rows of B+1 group offsets
  u64 *state; u32 *ticket;             // look-back over the buckets
  int force_retry;                     // tests: behave as if a bucket did not fit
  // H_ii of the determinants a bucket creates, deferred to spare blocks of the NEXT step's k_spawn (which runs beside its spawning
  // blocks instead of at the end of every bucket's chain): bucket b leaves hq_cnt[b] <= BK_HQ_DEFER positions in hq_pos[b * BK_HQ_DEFER ..]
  u32 *hq_cnt, *hq_pos; int hq_nblk;   // (k_spawn: hq_nblk spare blocks work through the B = hq_B queues)
  int hq_B;
  // Where the buckets begin.  kb == null: at the residents at positions b n0 / B (equal numbers of residents).  Otherwise bucket b
  // holds the keys [kb[b], kb[b+1]): boundaries that equalise the cost of a bucket, residents + 1.7 spawns, as measured one bucket
  // step back (the spawns crowd on the heavy determinants: with equal residents the fullest bucket held 3x the mean, and every
  // bucket waits for it).  Keys, not positions: a determinant keeps its key, while its position moves with every birth and death
  // in front of it.
  const u32 *kb;
  u32 *pos;                            // positions of kb in this step's resident list (B + 1): a spare block of k_spawn finds them, the tail reads them
  u32 *scount;                         // out, B + 1 words: spawns of every bucket, then n0 -- what the next boundaries are made from
  const u32 *kb_prev; u32 *kb_out;     // k_spawn's spare block: the next boundaries from scount and the boundaries it was counted with (null: equal residents)
  const u32 *pos_prev;                 // where kb_prev lay in the list scount was taken from (the pos of that step)
  const u32 *hint; u32 *hint_out;      // about where kb lies in today's list (found when it was made) / the same for kb_out
  // k_spawn: parent of the first child of every spawning block (written by the kernel that wrote the child offsets; null: the block
  // searches the offsets itself, three dependent round trips at 10^6 walkers where this is one)
  const u32 *parent_hint;
};
#define BK_REBAL_MAXB (256 * BK_PER_CU)
#define BK_REBAL_SPAWN_COST 1.2        // a spawn against a resident in a bucket's cost.  Its time to publish fits 7.1 + 0.0101 S + 0.0059 R us over the
                                       // buckets of a step, i.e. 1.7; the smaller weight keeps the residents of spawn-poor ranges below the gap sort's 1023
__device__ __forceinline__ long long bk_bound(const BucketArgs &ba, int b, long long n0) {
  if (b <= 0) return 0;
  if (b >= ba.B) return n0;
  return ba.kb ? (long long)ba.pos[b] : ((long long)b * n0) / ba.B;
}
// first resident position whose key is >= k (rkeys: key << 32 | index, ascending), looked for around `hint` first: the list
// changes by a fraction of a percent from step to step, so a boundary found yesterday is within a few hundred places today
__device__ __forceinline__ long long bk_lower_bound_near(const u64 *__restrict__ rkeys, long long n0, u32 k, long long hint) {
  const long long W = 384;
  long long lo = hint - W < 0 ? 0 : hint - W, hi = hint + W > n0 ? n0 : hint + W;
  if (lo > hi) lo = hi;
  const u64 vlo = (lo > 0 && lo < n0) ? rkeys[lo] : 0ull, vhi = (hi > 0 && hi < n0) ? rkeys[hi - 1] : 0ull;      // both in flight
  if (lo > 0 && (lo >= n0 || (u32)(vlo >> 32) >= k)) lo = 0;
  if (hi < n0 && hi > 0 && (u32)(vhi >> 32) < k) hi = n0;
  if (hi == 0) hi = n0;
  while (lo < hi) { const long long m = (lo + hi) >> 1; if ((u32)(rkeys[m] >> 32) < k) lo = m + 1; else hi = m; }
  return lo;
}
// One block of BK_T threads beside the spawning blocks of k_spawn: (1) where this step's boundaries lie in this step's resident
// list; (2) the next boundaries: every bucket the same cost, if residents and spawns fall as they did when scount was taken,
// evenly over the KEY range of each bucket of then.  (Evenly over its residents does not work: most spawns land on
// determinants nobody occupies, whole key ranges of them, and die there by the initiator rule -- a bucket can hold 2000 spawns
// in a range its 650 residents only line the edges of.)  If the new boundaries would overfill a block with residents, or leave
// the first or the last bucket without any, equal-residents boundaries are written instead.
__device__ __forceinline__ void bk_rebalance_block(const BucketArgs &ba, const u64 *__restrict__ rkeys, long long n0) {
  __shared__ double s_pre[BK_REBAL_MAXB + 1]; __shared__ u32 s_pp[BK_REBAL_MAXB + 1], s_kk[BK_REBAL_MAXB + 1], s_kn[BK_REBAL_MAXB + 1], s_pn[BK_REBAL_MAXB + 1];
  __shared__ int s_bad;
  const int tid = threadIdx.x, B = ba.B;
  if (tid == 0) s_bad = 0;
  for (int j = tid; j <= B; j += BK_T) {
    const bool inner = j > 0 && j < B;
    if (ba.kb) ba.pos[j] = (j == 0) ? 0u : (j == B ? (u32)n0 : (u32)bk_lower_bound_near(rkeys, n0, ba.kb[j], ba.hint ? (long long)ba.hint[j] : n0 / 2));
    if (ba.kb_out) {
      // the boundaries of then, where they lay then: the residents they held are the ones the spawns were counted beside
      long long pp = (j == 0) ? 0 : (j == B ? n0 : ((ba.kb_prev && ba.pos_prev) ? (long long)ba.pos_prev[j] : ((long long)j * n0) / B));
      if (pp > n0) pp = n0;
      s_pp[j] = (u32)pp;
      s_kk[j] = (j == 0) ? 0u : (j == B ? (u32)(rkeys[n0 - 1] >> 32) + 1u : ((ba.kb_prev && ba.pos_prev) ? ba.kb_prev[j] : (u32)(rkeys[pp < n0 ? pp : n0 - 1] >> 32)));
    }
    (void)inner;
  }
  if (!ba.kb_out) return;
  __syncthreads();
  if (tid == 0) {                      // 256 additions beside a kernel that runs for tens of microseconds
    double acc = 0.0;
    for (int b = 0; b < B; b++) { s_pre[b] = acc; acc += (double)(s_pp[b + 1] - s_pp[b]) + BK_REBAL_SPAWN_COST * (double)ba.scount[b]; }
    s_pre[B] = acc;
  }
  __syncthreads();
  const double total = s_pre[B];
  for (int j = tid; j <= B; j += BK_T) {
    u32 kn = (j == 0) ? 0u : s_kk[B];
    if (j > 0 && j < B) {
      const double target = total * (double)j / (double)B;
      int lo = 0, hi = B;               // s_pre[lo] <= target < s_pre[hi]
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= target) lo = mid; else hi = mid; }
      // inside bucket lo of then: its residents where they are today, its spawns evenly over its key range.  Cost up to resident p
      // (exclusive) = (p - p_lo) + d (key(p) - K_lo); the boundary goes behind the last resident that cost still admits, then on
      // into the gap behind it as far as the spawn density allows
      const double r = target - s_pre[lo];
      const u32 p_lo = s_pp[lo], p_hi = s_pp[lo + 1], K_lo = s_kk[lo], K_hi = s_kk[lo + 1];
      const double d = (K_hi > K_lo) ? BK_REBAL_SPAWN_COST * (double)ba.scount[lo] / (double)(K_hi - K_lo) : 0.0;
      u32 a = p_lo, e = p_hi;           // residents [p_lo, a) are admitted, [e, p_hi) are not
      while (a < e) {
        const u32 m = (a + e) >> 1;
        const u32 km = (u32)(rkeys[m] >> 32);
        const double f = (double)(m - p_lo) + d * (double)(km > K_lo ? km - K_lo : 0u);
        if (f <= r) a = m + 1; else e = m;
      }
      if (a == p_lo) kn = K_lo + (d > 0.0 ? (u32)(r / d) : 0u);                      // in front of the bucket's first resident
      else {
        const u32 kp = (u32)(rkeys[a - 1] >> 32);
        const double f = (double)(a - 1 - p_lo) + d * (double)(kp > K_lo ? kp - K_lo : 0u);
        kn = kp + 1u + (d > 0.0 ? (u32)((r - f) / d) : 0u);
      }
      const u32 cap_k = (a < p_hi) ? (u32)(rkeys[a] >> 32) : K_hi;                   // not past the first resident that was not admitted
      if (kn > cap_k) kn = cap_k;
      s_pn[j] = a;                       // kn lies in (key(a - 1), key(a)]: a is its place in today's list
    } else s_pn[j] = (j == 0) ? 0u : (u32)n0;
    s_kn[j] = kn;
  }
  __syncthreads();
  for (int j = tid + 1; j < B; j += BK_T) if (s_kn[j] <= s_kn[j - 1] || s_pn[j] < s_pn[j - 1]) s_bad = 1;
  __syncthreads();
  for (int j = tid; j < B; j += BK_T) {
    const u32 r = s_pn[j + 1] - s_pn[j];
    if (r > (u32)BK_RESIDENTS_MAX || ((j == 0 || j == B - 1) && r == 0u)) s_bad = 1;
  }
  __syncthreads();
  const bool bad = s_bad != 0;
  for (int j = tid; j < B; j += BK_T) {
    ba.kb_out[j] = (j == 0) ? 0u : (bad ? (u32)(rkeys[((long long)j * n0) / B] >> 32) : s_kn[j]);
    if (ba.hint_out) ba.hint_out[j] = bad ? (u32)(((long long)j * n0) / B) : s_pn[j];
  }
}

// ------------------------------------------------------------------------------------------------ partition
// The 256 children of one block, grouped by bucket (stable) behind the block's row of group offsets.  spl[b] (b >= 1) = first
// key of bucket b, staged by the caller; wcnt = BK_T/64 rows of ws >= B zeroed counters.  Every thread of the block calls it.
__device__ __forceinline__ void bucket_partition_block(const u32 *__restrict__ spl, u32 *__restrict__ wcnt, int ws, bool valid, u32 key, u64 word, long long blk,
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
  if (valid && rank == 0) wcnt[wv * ws + bkt] = (u32)__popcll(same);
  __syncthreads();
  // group offsets: exclusive scan over the buckets of the four waves' counts; wcnt becomes the base of each wave's share
  constexpr int PER = BK_MAXB / BK_T;
  u32 t4[PER]; u64 sum = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) { const int b = tid * PER + q; u32 s = 0; if (b < B) for (int v = 0; v < BK_T / 64; v++) s += wcnt[v * ws + b]; t4[q] = s; sum += s; }
  u64 tot; u32 ex = (u32)block_excl_scan_u64(sum, &tot);
  unsigned short *row = ba.segoff + blk * (B + 1);
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int b = tid * PER + q;
    if (b < B) { row[b] = (unsigned short)ex; u32 a = ex; for (int v = 0; v < BK_T / 64; v++) { const u32 cn = wcnt[v * ws + b]; wcnt[v * ws + b] = a; a += cn; } }
    ex += t4[q];
  }
  if (tid == 0) row[B] = (unsigned short)tot;
  __syncthreads();
  if (valid) ba.words[blk * BK_T + wcnt[wv * ws + bkt] + rank] = word;
}
// splitters and zeroed counters of a partition block (no barrier inside: the caller synchronises once before partitioning)
__device__ __forceinline__ void bucket_partition_stage(u32 *__restrict__ spl, u32 *__restrict__ wcnt, int ws, const u64 *__restrict__ rkeys, long long n0, const BucketArgs &ba) {
  const int B = ba.B;
  for (int b = threadIdx.x; b < B; b += BK_T) spl[b] = b ? (ba.kb ? ba.kb[b] : (u32)(rkeys[((long long)b * n0) / B] >> 32)) : 0u;      // first key of bucket b
  for (int d = threadIdx.x; d < (BK_T / 64) * ws; d += BK_T) wcnt[d] = 0;
}
