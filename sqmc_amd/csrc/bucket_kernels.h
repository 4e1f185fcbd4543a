// bucket_kernels.h -- the annihilation tail of a semistochastic step for SHORT lists, in one launch instead of ten.
// Textually included by sqmc_gpu.hip behind walk_kernels.h (same translation unit, same types).
//
// At 10^5 walkers a step is a chain of dependent launches, not bandwidth: the stable radix sort of the (walkers + spawns)
// list alone was nine kernels (54 us of a 122 us step) in front of the one-kernel annihilation.  Here the sort never
// leaves the chip:
//
//   partition            one block per 256 children (inside k_spawn as it emits them, or k_bucket_partition for spawn lists that
//                        came from elsewhere): every child finds its bucket among B key ranges and the block writes its children
//                        grouped by bucket (stable counting sort in LDS) with the B+1 group offsets.  No counts cross blocks,
//                        no atomics, no scan kernel.  The ranges: until boundaries have been learnt, the keys of the resident
//                        walkers at positions b n0 / B (every step leaves the residents sorted); then boundary keys that
//                        equalise residents + spawns per bucket, remade every step from the last tail's counts
//                        (bk_rebalance_block, bucket_partition.h).
//   k_anneal_bucket      one block per bucket: gathers its children from every partition block (two dependent loads), sorts
//                        them in LDS -- one stable counting pass on the GAP between residents a child falls into, then a
//                        ranking inside the gap: the order of merge_sort2_up_dn (do_walk.f90:5411-5614), residents first on
//                        equal keys, spawns in creation order --, folds every run of equal determinants with the rules of
//                        merge_original_with_spawned2 (5866-6083), rounds small weights (reduce_my_walker, 7196-7254), and
//                        compacts into the other walker buffer through one decoupled look-back over the buckets --
//                        reweighting, C(T) lookup, estimator sums as in k_anneal; the next step's gate AND child offsets (no
//                        scan launch); death/clone and the projection's last line when the host fused them in (FusedSide);
//                        H_ii of the determinants it creates.
//
// A bucket that does not fit the LDS of its block (a population that moved more than the head-room in one step) raises
// DevScalars::retry and nothing else: the kernel only ever writes the OTHER walker buffer and scratch, so the host re-runs
// the tail of that step through the radix path and keeps the bucket path off for a while.  COUNTER discipline, packed keys.
#pragma once

// the boundary block as a kernel of its own (sharded steps: on the side stream, beside the exchange)
__global__ void __launch_bounds__(BK_T) k_bucket_boundaries(const u64 *__restrict__ keys, long long n0, BucketArgs ba) { bk_rebalance_block(ba, keys, n0); }
// stand-alone: for spawn lists that did not come out of k_spawn (the annihilation door) or heads launched without it
__global__ void __launch_bounds__(BK_T) k_bucket_partition(const u64 *__restrict__ keys, long long n0, long long nch, u64 invalid_key, BucketArgs ba, int n_extra) {
  if ((int)blockIdx.x < n_extra) { bk_rebalance_block(ba, keys, n0); return; }      // the boundary block, in front (as in k_spawn)
  __shared__ u32 spl[BK_MAXB];
  __shared__ u32 wcnt[BK_T / 64][BK_MAXB];
  const long long blk = (long long)blockIdx.x - n_extra;
  const long long c = blk * BK_T + threadIdx.x;
  u64 word = 0; u32 key = 0; bool valid = false;
  if (c < nch) word = keys[n0 + c];
  bucket_partition_stage(spl, &wcnt[0][0], BK_MAXB, keys, n0, ba);
  if (c < nch) { key = (u32)(word >> 32); valid = (u64)key != invalid_key; }
  __syncthreads();
  bucket_partition_block(spl, &wcnt[0][0], BK_MAXB, valid, key, word, blk, ba);
}

// ------------------------------------------------------------------------------------------------ annihilation per bucket
// merge_original_with_spawned2's pairwise combination (do_walk.f90:5897-5950) of a follower (w2, flags fs) into the
// running walker of its determinant: the same statement as MERGE_FOLD of fold_slot
__device__ __forceinline__ void bk_fold(double &wt, int &ini, int &d, double w2, u32 fs, const StepP &p) {
  const int i2 = flg_init(fs), d2 = flg_impd(fs);
  const bool same_sign = (w2 * wt > 0);
  if (same_sign) { if (i2 > ini) ini = i2; }
  if (d == -2) { if (d2 == 0) d = 0; }
  else if (d2 == -2) { if (d != 0) d = -2; }
  else if (d != 0 && d != -2) { const int a_ = d2 < 0 ? -d2 : d2; if (a_ < d) d = a_; }
  if (!same_sign) {
    if (fabs(wt) < fabs(w2)) { if (ini != 3 || p.r_init == -1.0) ini = i2; }
    else if (fabs(wt) == fabs(w2)) { if (ini != 3 || p.r_init == -1.0) ini = 0; }
  }
  if (!(d == 0 && d2 == -1)) wt = wt + w2;
}

#ifdef BUCKET_PROF
// build with -DBUCKET_PROF (tools/bucket_prof.py): wall-clock stamps (100 MHz) of every bucket at the phase boundaries
__device__ unsigned long long g_bprof[16 * 1024];
#define BPROF(K) do { if (threadIdx.x == 0 && b < 1024) g_bprof[b * 16 + (K)] = wall_clock64(); } while (0)
#define BPROF_VAL(K, V) do { if (threadIdx.x == 0 && b < 1024) g_bprof[b * 16 + (K)] = (unsigned long long)(V); } while (0)
extern "C" int sqmc_gpu_debug_bprof(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bprof), sizeof(g_bprof)); }
#else
#define BPROF(K)
#define BPROF_VAL(K, V)
#endif
// block-wide exclusive scan for the BK_AT threads of the annihilation kernel (total in every thread)
__device__ __forceinline__ u64 bk_block_excl_scan(u64 v, u64 *total) {
  __shared__ u64 wsum[BK_AT / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 inc = wave_incl_scan_u64(v, lane);
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  u64 off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < BK_AT / 64; i++) { if (i < wv) off += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}
// Look-back over the buckets with the whole block: every thread polls its own predecessors (two at most: B <= BK_MAXB), so the
// running value of all earlier buckets arrives in ONE round trip once they have published, instead of 64 predecessors per
// poll.  Same words and protocol as lookback_exclusive (status in the top two bits: 1 = this bucket's own total, 2 = running total
// up to and including it).  All BK_AT threads call it; returns the exclusive running value in every thread.
__device__ __forceinline__ u64 bk_lookback_wide(u64 *__restrict__ state, int b, u64 tot) {
  __shared__ int s_pstar[BK_AT / 64]; __shared__ u64 s_part[BK_AT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (b == 0) { if (tid == 0) atomicExch((unsigned long long *)&state[0], SCAN_ST_INC | tot); return 0; }
  if (tid == 0) atomicExch((unsigned long long *)&state[b], SCAN_ST_AGG | tot);
  u64 w_[BK_MAXB / BK_AT]; int ix_[BK_MAXB / BK_AT];
  int pstar = -1;
#pragma unroll
  for (int q = 0; q < BK_MAXB / BK_AT; q++) {
    const int idx = b - 1 - tid - q * BK_AT; ix_[q] = idx; w_[q] = 0;
    if (idx >= 0) {
      u64 w;
      do { w = atomicAdd((unsigned long long *)&state[idx], 0ull); } while ((w >> 62) == 0);
      w_[q] = w;
      if ((w >> 62) == 2 && idx > pstar) pstar = idx;
    }
  }
  // the nearest predecessor that already carries a running total
  for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(pstar, o, 64); pstar = x > pstar ? x : pstar; }
  if (lane == 0) s_pstar[wv] = pstar;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < BK_AT / 64; v++) pstar = s_pstar[v] > pstar ? s_pstar[v] : pstar;
  u64 sum = 0;
#pragma unroll
  for (int q = 0; q < BK_MAXB / BK_AT; q++) if (ix_[q] >= 0 && ix_[q] >= pstar) sum += w_[q] & SCAN_VAL_MASK;
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) s_part[wv] = sum;
  __syncthreads();
  u64 excl = 0;
#pragma unroll
  for (int v = 0; v < BK_AT / 64; v++) excl += s_part[v];
  if (tid == 0) atomicExch((unsigned long long *)&state[b], SCAN_ST_INC | (excl + tot));
  return excl;
}
// (H_ii of one determinant by a group of lanes: hii_group.h -- k_spawn's spare blocks use it too)
#define BK_LONG_RUN 24                 // followers a head folds itself
#define BK_LONG_MAX 32                 // longer runs a bucket hands to wavefronts (more than that: their heads fold them alone)
#define BK_HQ_DETS (BK_CAP_T * 2 / 16)                       // determinants whose (up, dn) wait in LDS for the H_ii phase
// sum over the wavefront by DPP row shifts and row broadcasts (lane 63 holds it): 6 steps of two 32-bit moves and an add
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double bk_dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bk_wave_sum_lane63(double v) {
  v += bk_dpp_f64<0x111, 0xf>(v); v += bk_dpp_f64<0x112, 0xf>(v); v += bk_dpp_f64<0x114, 0xf>(v); v += bk_dpp_f64<0x118, 0xf>(v);     // row_shr 1, 2, 4, 8: prefix inside rows of 16
  v += bk_dpp_f64<0x142, 0xa>(v);                                                                                                    // row_bcast 15 into rows 1 and 3
  v += bk_dpp_f64<0x143, 0xc>(v);                                                                                                    // row_bcast 31 into rows 2 and 3
  return v;
}
// elements one thread handles per phase at the caps (loops are unrolled to these so that all loads of a phase are in flight)
#define BK_PER_ROWS ((BK_CAP_ROWS + BK_AT - 1) / BK_AT)
#define BK_PER_S ((BK_CAP_S + BK_AT - 1) / BK_AT)
#define BK_PER_R ((BK_CAP_R + BK_AT - 1) / BK_AT)
#define BK_PER_T ((BK_CAP_T + BK_AT - 1) / BK_AT)

__global__ void __launch_bounds__(BK_AT, BK_PER_CU * (BK_AT / 64) / 4) k_anneal_bucket(WalkArr w, WalkArr o, const u64 *__restrict__ rkeys, int *__restrict__ loc_imp,
                                                         const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 hmask,
                                                         const double *__restrict__ cnum, const double *__restrict__ cden,
                                                         double *__restrict__ partials, double *__restrict__ wabs_part, long long n0, long long nch, StepP p,
                                                         u64 invalid_key, u64 seed, u64 step, DevScalars *sc, BucketArgs ba, GateOut go, FusedSide fs, ChemDev dev) {
  // ---- LDS: the whole bucket lives here
  __shared__ u64 sw[BK_CAP_S], sw2[BK_CAP_S];            // sort words of the spawns (double buffer)
  __shared__ u32 rk[BK_CAP_R];                            // keys of the residents
  __shared__ double s_w[BK_CAP_T]; __shared__ u32 s_f[BK_CAP_T];   // weight / flags by SOURCE: residents [0, R), sorted spawns [R, R + S); later the merged walker of a run, at its head
  __shared__ u32 m2s[BK_CAP_T];                           // merged order -> source (| BK_STOP at the first slot of a run)
  __shared__ __align__(16) unsigned short rnk[BK_CAP_T];               // sort: rank inside its digit; later: keep code of a merged slot
  // rows of the gather (offset u16 + base u32), then the digit counters of the sort, then child counts [0, T) + the Slater-Condon tables [BK_SCR_TAB, ...)
  constexpr int BK_SCR_TAB = (BK_CAP_T + 3) & ~3;
  constexpr int BK_SCR_ROWS = (BK_CAP_ROWS / 2 + 4) + BK_CAP_ROWS + 1, BK_SCR_CNT = (BK_AT / 64) * 1024 * (int)sizeof(bk_cnt_t) / 4, BK_SCR_TABEND = BK_SCR_TAB + ((int)sizeof(ChemTab) + 3) / 4;
  constexpr int BK_SCR_WORDS = ((BK_SCR_ROWS > BK_SCR_CNT ? (BK_SCR_ROWS > BK_SCR_TABEND ? BK_SCR_ROWS : BK_SCR_TABEND) : (BK_SCR_CNT > BK_SCR_TABEND ? BK_SCR_CNT : BK_SCR_TABEND)) + 3) & ~3;
  __shared__ __align__(16) u32 scratch[BK_SCR_WORDS];
  __shared__ unsigned short s_hq[BK_CAP_T]; __shared__ int s_hqn;      // kept walkers of this bucket that have no H_ii yet (position inside the bucket's output)
  static_assert(BK_CAP_T <= 4096 && (sizeof(bk_cnt_t) == 4 || BK_CAP_S <= 65535), "ranks inside a bucket are 12 bits; the counters hold a bucket's spawns");
  ChemTab *s_tab = (ChemTab *)(scratch + BK_SCR_TAB);
  u64 *s_hdet = (u64 *)rnk;                               // (up, dn) of the first BK_HQ_DETS queued determinants: the keep codes are idle once the ranks exist
  __shared__ u32 s_tile; __shared__ u32 s_kmin, s_kmax;
  __shared__ double s_red[BK_AT / 64][NSTAT + 2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { s_tile = atomicAdd(ba.ticket, 1u); s_hqn = 0; }
  __syncthreads();
  const int b = (int)s_tile, B = ba.B, nsb = ba.nsb;
  const long long r_lo = bk_bound(ba, b, n0), r_hi = bk_bound(ba, b + 1, n0);
  const int R = (int)(r_hi - r_lo);
  BPROF(0);
  // key range of the bucket (the sort only looks at the bits that vary inside it)
  if (tid == 0) s_kmin = b ? (ba.kb ? ba.kb[b] : (u32)(rkeys[r_lo] >> 32)) : 0u;
  if (tid == 64) s_kmax = (b + 1 < B) ? (ba.kb ? ba.kb[b + 1] : (u32)(rkeys[r_hi] >> 32)) - 1u : (u32)invalid_key;
  unsigned short *seg_lo = (unsigned short *)scratch; u32 *seg_base = scratch + (BK_CAP_ROWS / 2 + 4);     // u16 x ROWS, then u32 x (ROWS + 1)
  // (key boundaries: the first and the last walker of the whole list get special treatment below and are looked for in the first
  //  and the last bucket -- which equal-residents boundaries never leave empty; here an empty one sends the step to the radix tail)
  bool fits = nsb <= BK_CAP_ROWS && R <= BK_CAP_R && !(ba.kb && (b == 0 || b == B - 1) && R == 0);
  // ---- rows: where the bucket's children lie in every partition block.  Thread t owns rows [t C, (t+1) C): all loads first.
  //      The residents' keys are requested in the same round trip.
  u32 rk_reg[BK_PER_R];
#pragma unroll
  for (int q = 0; q < BK_PER_R; q++) { const int i = tid + q * BK_AT; rk_reg[q] = (fits && i < R) ? (u32)(rkeys[r_lo + i] >> 32) : 0u; }
  int S = 0;
  const bool rows_ok = nsb <= BK_CAP_ROWS;             // (the host checks this before it launches)
  if (rows_ok) {                                       // also for a bucket that will give up: the host learns the next boundaries from every bucket's count
    const int C = (nsb + BK_AT - 1) / BK_AT;
    unsigned short lo_[BK_PER_ROWS], hi_[BK_PER_ROWS];
#pragma unroll
    for (int q = 0; q < BK_PER_ROWS; q++) {
      const int t = tid * C + q;
      lo_[q] = 0; hi_[q] = 0;
      if (q < C && t < nsb) { const unsigned short *rp = ba.segoff + (long long)t * (B + 1) + b; lo_[q] = rp[0]; hi_[q] = rp[1]; }
    }
    u64 mine = 0;
#pragma unroll
    for (int q = 0; q < BK_PER_ROWS; q++) mine += (u64)(hi_[q] - lo_[q]);
    u64 tot; u32 ex = (u32)bk_block_excl_scan(mine, &tot);
#pragma unroll
    for (int q = 0; q < BK_PER_ROWS; q++) {
      const int t = tid * C + q;
      if (q < C && t < nsb) { seg_lo[t] = lo_[q]; seg_base[t] = ex; ex += (u32)(hi_[q] - lo_[q]); }
    }
    if (tid == 0) seg_base[nsb] = (u32)tot;
    S = (int)tot;
    fits = fits && S <= BK_CAP_S && R + S <= BK_CAP_T && !ba.force_retry;
  }
  const int T = R + S;
  if (!fits) {
    // nothing was written that the radix path would miss; the look-back must still see this bucket
    if (tid == 0) atomicExch((int *)&sc->retry, 1);
    if (tid == 0 && ba.scount && rows_ok) { ba.scount[b] = (u32)S; if (b == 0) ba.scount[B] = (u32)n0; }
#ifdef BUCKET_PROF
    if (tid == 0) printf("bucket %d of %d does not fit: rows %d, R %d, S %d (r_lo %lld)\n", b, B, nsb, R, S, r_lo);
#endif
    if (tid < 64) lookback_exclusive(ba.state, (u32)b, 0ull, tid);
    return;
  }
#pragma unroll
  for (int q = 0; q < BK_PER_R; q++) { const int i = tid + q * BK_AT; if (i < R) rk[i] = rk_reg[q]; }
  __syncthreads();
  BPROF(1);
  // ---- gather the children's sort words, partition block by partition block (= creation order): word j of the bucket lies in
  //      the row whose base is the last one <= j
  {
    u64 wv_[BK_PER_S];
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) {
      const int j = tid + q * BK_AT;
      wv_[q] = 0;
      if (j < S) {
        int lo = 0, hi = nsb;                        // seg_base[lo] <= j < seg_base[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= (u32)j) lo = mid; else hi = mid; }
        wv_[q] = ba.words[(long long)lo * BK_T + seg_lo[lo] + ((u32)j - seg_base[lo])];
      }
    }
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) { const int j = tid + q * BK_AT; if (j < S) sw[j] = wv_[q]; }
  }
  __syncthreads();
  BPROF(2); BPROF_VAL(12, S); BPROF_VAL(13, R);
  if (tid == 0 && ba.scount) { ba.scount[b] = (u32)S; if (b == 0) ba.scount[B] = (u32)n0; }
  // ---- sort of the spawn words.  Short resident lists (always, at the sizes the host picks): ONE stable counting pass on the GAP of
  //      a spawn -- the number of residents with key <= its own, found by binary search -- then every spawn ranks itself among the
  //      spawns of its gap by (key, creation order).  A gap holds a handful of spawns (those between two neighbouring residents), or
  //      many with one key (the heavy determinants: the loop below reads LDS at one address per step, a broadcast).  The time does
  //      not depend on how wide the bucket's key range is (the key-digit sort took 2-4 passes), and the gap IS the merge position.
  u64 *sa = sw, *sb = sw2;
  const bool gap_sort = R <= 1023;
  unsigned short *gs = (unsigned short *)m2s, *gst = gs + BK_CAP_S;     // gap of the spawn at a sorted position; first sorted position of a gap (R + 2 entries) -- m2s is idle until the merged order is written
  static_assert((BK_CAP_S + 1026) * 2 <= BK_CAP_T * 4, "gap tables fit the merged-order array");
  if (gap_sort) {
    unsigned short *gp = s_hq;                            // gap by creation order (the H_ii queue is idle until the compaction)
    bk_cnt_t(*wcnt)[1024] = (bk_cnt_t(*)[1024])scratch;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int chunk = ((S + BK_AT - 1) / BK_AT) * 64;
    for (int d = tid; d < BK_SCR_CNT; d += BK_AT) scratch[d] = 0;
    for (int j = tid; j < S; j += BK_AT) {
      const u32 k = (u32)(sa[j] >> 32);
      int lo = 0, hi = R;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (rk[mid] <= k) lo = mid + 1; else hi = mid; }
      gp[j] = (unsigned short)lo;
    }
    __syncthreads();
    {
      const int beg = wv * chunk, end = (beg + chunk < S) ? beg + chunk : S;
      for (int base = beg; base < end; base += 64) {
        const int idx = base + lane; const bool valid = idx < end;
        const u32 dig = valid ? (u32)gp[idx] : 0u;
        u64 same = __ballot(valid);
#pragma unroll
        for (int q = 0; q < 10; q++) { const u64 m = __ballot((dig >> q) & 1); same &= ((dig >> q) & 1) ? m : ~m; }
        const u32 rank = (u32)__popcll(same & lt), cnt = (u32)__popcll(same);
        u32 prev = 0;
        if (valid) prev = wcnt[wv][dig];
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wcnt[wv][dig] = (bk_cnt_t)(prev + cnt);
        __builtin_amdgcn_wave_barrier();
        if (valid) rnk[idx] = (unsigned short)(prev + rank);
      }
    }
    __syncthreads();
    {
      u32 t2[2]; u64 sum = 0;
#pragma unroll
      for (int q = 0; q < 2; q++) { const int d = tid * 2 + q; u32 s2 = 0; for (int v = 0; v < BK_AT / 64; v++) s2 += wcnt[v][d]; t2[q] = s2; sum += s2; }
      u64 tt; u32 ex = (u32)bk_block_excl_scan(sum, &tt);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int d = tid * 2 + q; u32 a2 = ex;
        if (d <= R) gst[d] = (unsigned short)ex;
        for (int v = 0; v < BK_AT / 64; v++) { const u32 cn = wcnt[v][d]; wcnt[v][d] = (bk_cnt_t)a2; a2 += cn; }
        ex += t2[q];
      }
      if (tid == 0) gst[R + 1] = (unsigned short)S;
    }
    __syncthreads();
    for (int idx = tid; idx < S; idx += BK_AT) {
      const u32 dig = (u32)gp[idx];
      const u32 pos = wcnt[idx / chunk][dig] + rnk[idx];
      sb[pos] = sa[idx]; gs[pos] = (unsigned short)dig;
    }
    __syncthreads();
    for (int j = tid; j < S; j += BK_AT) {              // inside the gap: by key, then by creation order (the word's low half)
      const u64 mine = sb[j];
      const int g0 = (int)gs[j], a2 = (int)gst[g0], e2 = (int)gst[g0 + 1];
      int less = 0;
      if (e2 - a2 > 1) for (int i = a2; i < e2; i++) less += (sb[i] < mine) ? 1 : 0;
      sa[a2 + less] = mine;
    }
    __syncthreads();
  } else {
  // wide resident lists: stable LDS radix sort of the spawn words on key - kmin
  {
    const u32 kmin = s_kmin, span = s_kmax - kmin;
    int nbits = 0; while (nbits < 32 && (span >> nbits)) nbits++;
    const int npass = (nbits + 9) / 10, dbits = npass ? (nbits + npass - 1) / npass : 0;
    const int chunk = ((S + BK_AT - 1) / BK_AT) * 64;             // consecutive elements one wave ranks, in rounds of 64
    bk_cnt_t(*wcnt)[1024] = (bk_cnt_t(*)[1024])scratch;
    const u64 lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int ps = 0, shift = 0; ps < npass && S > 1; ps++, shift += dbits) {
      const u32 mask = (1u << dbits) - 1u;
      for (int d = tid; d < BK_SCR_CNT; d += BK_AT) scratch[d] = 0;
      __syncthreads();
      const int beg = wv * chunk, end = (beg + chunk < S) ? beg + chunk : S;
      for (int base = beg; base < end; base += 64) {
        const int idx = base + lane; const bool valid = idx < end;
        const u32 dig = valid ? ((((u32)(sa[idx] >> 32) - kmin) >> shift) & mask) : 0u;
        u64 same = __ballot(valid);
        for (int q = 0; q < dbits; q++) { const u64 m = __ballot((dig >> q) & 1); same &= ((dig >> q) & 1) ? m : ~m; }
        const u32 rank = (u32)__popcll(same & lt), cnt = (u32)__popcll(same);
        u32 prev = 0;
        if (valid) prev = wcnt[wv][dig];
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) wcnt[wv][dig] = (bk_cnt_t)(prev + cnt);
        __builtin_amdgcn_wave_barrier();
        if (valid) rnk[idx] = (unsigned short)(prev + rank);
      }
      __syncthreads();
      {   // digit bases: exclusive scan over the digits of the waves' counts (two digits per thread)
        u32 t2[2]; u64 sum = 0;
#pragma unroll
        for (int q = 0; q < 2; q++) { const int d = tid * 2 + q; u32 s2 = 0; for (int v = 0; v < BK_AT / 64; v++) s2 += wcnt[v][d]; t2[q] = s2; sum += s2; }
        u64 tt; u32 ex = (u32)bk_block_excl_scan(sum, &tt);
#pragma unroll
        for (int q = 0; q < 2; q++) { const int d = tid * 2 + q; u32 a = ex; for (int v = 0; v < BK_AT / 64; v++) { const u32 cn = wcnt[v][d]; wcnt[v][d] = (bk_cnt_t)a; a += cn; } ex += t2[q]; }
      }
      __syncthreads();
      for (int idx = tid; idx < S; idx += BK_AT) {
        const u64 x = sa[idx];
        const u32 dig = ((((u32)(x >> 32)) - kmin) >> shift) & mask;
        sb[wcnt[idx / chunk][dig] + rnk[idx]] = x;
      }
      __syncthreads();
      u64 *tp = sa; sa = sb; sb = tp;
    }
  }
  }
  BPROF(3);
  // ---- records by source, all requested before any is used.  With fs.on the block also does what the side-stream kernels
  //      did: death/clone (do_walk.f90:3743-3793) of the residents outside the deterministic space as their weights arrive ...
  {
    double rw[BK_PER_R]; u32 rf[BK_PER_R]; double rm[BK_PER_R]; u32 rr[BK_PER_R]; double sw_[BK_PER_S]; u64 sf_[BK_PER_S];
#pragma unroll
    for (int q = 0; q < BK_PER_R; q++) {
      const int i = tid + q * BK_AT; rw[q] = 0.0; rf[q] = 0; rm[q] = 0.0; rr[q] = 0;
      if (i < R) { rw[q] = w.wt[r_lo + i]; rf[q] = w.flg[r_lo + i]; if (fs.on) { rm[q] = w.me[r_lo + i]; rr[q] = w.irk[r_lo + i]; } }
    }
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) {
      const int j = tid + q * BK_AT; sw_[q] = 0.0; sf_[q] = 0;
      if (j < S) { const SpawnRec *rp = w.sp + ((long long)(u32)sa[j] - n0); sw_[q] = rp->wt; sf_[q] = rp->flg; }     // the word's low half is the walker slot n0 + child
    }
#pragma unroll
    for (int q = 0; q < BK_PER_R; q++) {
      const int i = tid + q * BK_AT;
      if (i < R) {
        double x = rw[q];
        if (fs.on) {
          const int impd = flg_impd(rf[q]);
          if (!(p.semi && impd < 1)) {
            double f = 1.0 + p.tau * (p.e_trial - rm[q]);
            if (f < 0) { if (p.reached > 1) sc->err = SQMC_ERR_NEG_DIAG; f = 0; }
            x = x * f;
          } else if (impd == 0) {                       // ... and the projection's last line for those inside it: w(loc) += (A x)(row) + E_T tau x(row), k_prj_apply's statements
            double y = fs.y[rr[q]];
            y = y + p.e_trial * p.tau * fs.x_in[rr[q]];
            x = x + y;
          }
        }
        s_w[i] = x; s_f[i] = rf[q];
      }
    }
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) { const int j = tid + q * BK_AT; if (j < S) { s_w[R + j] = sw_[q]; s_f[R + j] = (u32)sf_[q]; } }
  }
  __syncthreads();
  BPROF(4);
  // ---- sums over the pre-merge list (do_walk.f90:2347-2349), after death/clone and projection as the reference takes them
  double wabs = 0.0, cnt = 0.0;
  for (int x = tid; x < T; x += BK_AT) { wabs += fabs(s_w[x]); cnt += 1.0; }
  // ---- merged order: a resident goes behind the spawns with smaller keys, a spawn behind the residents with keys <= its own
  if (gap_sort) {
    // both are known from the sort: gst[i + 1] spawns lie in the gaps 0..i, i.e. in front of resident i; the spawn at sorted
    // position j lies behind gs[j] residents.  (The tables live in the array the merged order goes to: read first, then write.)
    u32 rp_[BK_PER_R], cp_[BK_PER_S]; u32 ce_[BK_PER_S];
#pragma unroll
    for (int q = 0; q < BK_PER_R; q++) { const int i = tid + q * BK_AT; rp_[q] = (i < R) ? (u32)i + (u32)gst[i + 1] : 0u; }
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) {
      const int j = tid + q * BK_AT; cp_[q] = 0; ce_[q] = 0;
      if (j < S) {
        const u32 k = (u32)(sa[j] >> 32); const int g0 = (int)gs[j];
        const bool on_resident = g0 > 0 && rk[g0 - 1] == k;
        const bool head = !on_resident && (j == 0 || (u32)(sa[j - 1] >> 32) != k);
        cp_[q] = (u32)(j + g0); ce_[q] = (head ? BK_STOP : 0u) | (u32)(R + j);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BK_PER_R; q++) { const int i = tid + q * BK_AT; if (i < R) m2s[rp_[q]] = BK_STOP | (u32)i; }
#pragma unroll
    for (int q = 0; q < BK_PER_S; q++) { const int j = tid + q * BK_AT; if (j < S) m2s[cp_[q]] = ce_[q]; }
  } else {
  for (int i = tid; i < R; i += BK_AT) {
    const u32 k = rk[i];
    int lo = 0, hi = S;                                                    // spawns with key < k
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((u32)(sa[mid] >> 32) < k) lo = mid + 1; else hi = mid; }
    m2s[i + lo] = BK_STOP | (u32)i;
  }
  for (int j = tid; j < S; j += BK_AT) {
    const u32 k = (u32)(sa[j] >> 32);
    int lo = 0, hi = R;                                                    // residents with key <= k
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (rk[mid] <= k) lo = mid + 1; else hi = mid; }
    const bool on_resident = lo > 0 && rk[lo - 1] == k;
    const bool head = !on_resident && (j == 0 || (u32)(sa[j - 1] >> 32) != k);
    m2s[j + lo] = (head ? BK_STOP : 0u) | (u32)(R + j);
  }
  }
  __syncthreads();
  BPROF(5);
  u32 *s_nc = scratch;                                  // the sort's counters are idle from here on: child count of every kept walker, then its prefix
  // ---- runs: the head folds its followers in merged order, then check_initiator, the discard rule and the rounding;
  //      the merged walker stays at the head's source slot (weight, packed flags); rnk[q] = 1 kept, 0x101 kept in the deterministic space
  // what follows the fold of a run, for its head at merged slot q (source slot si) and the slot qq behind its last follower
  auto finish_head = [&](int q, int si, double wt, int ini, int d, int psg, int qq) -> unsigned short {
    unsigned short keep = 0;
    {   // check_initiator, do_walk.f90:6838-6872
      const int dd = d - p.imind > 0 ? d - p.imind : 0;
      const double thr = p.r_init * ipow_d(dd, p.ipow), aw = fabs(wt);
      if (ini == 3 && p.r_init >= 0) { if (wt * psg < 1.0) wt = (double)psg; }
      else if (ini == 2 && ((aw <= thr && d > 0) || ((aw <= p.r_init && !p.cti) && d == -2))) ini = 1;
      else if (ini < 2 && ((aw > thr && d >= 0) || ((aw > p.r_init || p.cti) && d == -2))) ini = ini + 1;
    }
    int dtest = d;
    if (d == -1) { if (b == B - 1 && qq >= T) dtest = 1; d = 1; }          // 6032-6036 then the last-det test at 6038
    const bool discard = (((wt == 0.0 && (ini != 3 || p.r_init < 0)) || ini == 0) && dtest >= 1);
    if (!discard) {
      if (d >= 1 && fabs(wt) < p.min_wt) {                                // reduce_my_walker, 7196-7254: the draw is keyed by the determinant's rank (its sort key)
        const u32 kk = (si < R) ? rk[si] : (u32)(sa[si - R] >> 32);
        Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, (u64)kk);
        if (rng_draw(g) < (fabs(wt) / p.min_wt)) wt = copysign(p.min_wt, wt); else wt = 0.0;
      }
      if (!(wt == 0.0 && d >= 1)) keep = (d == 0) ? 0x101 : 0x1;        // zero weights outside the deterministic space are dropped (7222-7249)
    }
    s_w[si] = wt; s_f[si] = pack_flg(d, ini, psg);
    if (go.child_off && keep) {                                           // the next step's spawn gate: only the child count is needed before the positions are
      const u32 kk = (si < R) ? rk[si] : (u32)(sa[si - R] >> 32);
      u64 nc; double wc;
      gate_children(wt * p.rfi, go.cutoff, seed, go.step_next, (u64)kk, nc, wc);
      s_nc[q] = (u32)nc;
    }
    return keep;
  };
  // A head folds its first BK_LONG_RUN followers itself; what is left of a longer run (a heavy determinant: the Neel state of the
  // 4x4 Hubbard lattice collects ~10^3 spawns of both signs a step, and one thread folding them -- 100 cycles apiece -- kept every
  // other bucket waiting for 45 us) is handed to a whole wavefront below.
  __shared__ int s_lq[BK_LONG_MAX], s_lqq[BK_LONG_MAX], s_lini[BK_LONG_MAX], s_ld[BK_LONG_MAX], s_nlong;
  __shared__ double s_lwt[BK_LONG_MAX];
  if (tid == 0) s_nlong = 0;
  __syncthreads();
  for (int q = tid; q < T; q += BK_AT) {
    const u32 e = m2s[q];
    unsigned short keep = 0;
    if (e & BK_STOP) {
      const int si = (int)(e & ~BK_STOP);
      double wt = s_w[si]; const u32 f0 = s_f[si];
      int ini = flg_init(f0), d = flg_impd(f0); const int psg = flg_psign(f0);
      if (d == -1 && !(b == 0 && q == 0)) d = 1;                          // 5985-5986 (the very first walker keeps -1 until the end)
      int qq = q + 1;
      bool more = false;
      for (; qq < T; qq++) {
        const u32 e2 = m2s[qq];
        if (e2 & BK_STOP) break;
        if (qq - q > BK_LONG_RUN) { more = true; break; }
        bk_fold(wt, ini, d, s_w[e2], s_f[e2], p);
      }
      int slot = -1;
      if (more) { slot = atomicAdd(&s_nlong, 1); if (slot >= BK_LONG_MAX) { slot = -1; for (; qq < T; qq++) { const u32 e2 = m2s[qq]; if (e2 & BK_STOP) break; bk_fold(wt, ini, d, s_w[e2], s_f[e2], p); } } }
      if (slot >= 0) { s_lq[slot] = q; s_lqq[slot] = qq; s_lwt[slot] = wt; s_lini[slot] = ini; s_ld[slot] = d; }
      else keep = finish_head(q, si, wt, ini, d, psg, qq);
    }
    rnk[q] = keep;
  }
  __syncthreads();
  {
    // The rest of a long run, 64 followers at a time by one wavefront.  The reference's pairwise rule is sequential in the weight
    // (floating-point additions in storage order) and in the initiator flag (which looks at the running weight); both are kept:
    //   * lane 0 adds the weights one by one and leaves the running weight BEFORE every follower where that follower's own weight
    //     stood -- 8 cycles apiece, not 100;
    //   * every lane then knows what its follower does to the initiator flag, a map of {0,1,2,3} to itself (5905-5925): the maximum
    //     with its own flag if the signs agree, its own flag / 0 / nothing as the running weight is smaller / equal / larger if they
    //     do not (a permanent initiator keeps its 3 unless r_initiator = -1); the maps are composed in order by a shuffle tree;
    //   * the distance flag of a run does not depend on the order once no follower can carry 0 (spawns never do): 0 and -2 stay,
    //     any -2 follower makes -2, else the smallest |d2| (a head still at -1 keeps it).  The one place where the order of the
    //     flags and the weights meet -- a follower with -1 is not added to a head with 0 (5946) -- depends on the HEAD's 0 only.
    const int nlong = s_nlong < BK_LONG_MAX ? s_nlong : BK_LONG_MAX;
    for (int k = wv; k < nlong; k += BK_AT / 64) {
      const int q = s_lq[k]; int qq = s_lqq[k];
      double wt = s_lwt[k]; int ini = s_lini[k], d = s_ld[k];
      const int si = (int)(m2s[q] & ~BK_STOP);
      const int psg = flg_psign(s_f[si]);
      const bool keep3 = !(p.r_init == -1.0);                         // a permanent initiator's flag survives a sign change
      for (;;) {
        const int idx = qq + lane;
        const u32 e2 = (idx < T) ? m2s[idx] : BK_STOP;
        const u64 stops = __ballot((e2 & BK_STOP) != 0);
        const int n = stops ? (int)__builtin_ctzll(stops) : 64;       // followers in this round: consecutive sorted spawns
        if (n > 0) {
          const bool in = lane < n;
          const int e0 = (int)(__shfl((int)(e2 & ~BK_STOP), 0, 64));
          const double w2 = in ? s_w[e0 + lane] : 0.0; const u32 f2 = in ? s_f[e0 + lane] : 0u;
          const int i2 = flg_init(f2), d2 = flg_impd(f2);
          const u64 skipm = __ballot(in && d == 0 && d2 == -1);          // not added (d stays 0 once it is 0)
          const bool anym2 = __ballot(in && d2 == -2) != 0ull;
          int am = (in && d2 != -2) ? (d2 < 0 ? -d2 : d2) : 0x7FFFFFFF;
          __builtin_amdgcn_wave_barrier();
          if (lane == 0) {                                           // running weight in front of every follower, in storage order
            double run = wt;
            int z = 0;
            for (; z + 8 <= n; z += 8) {                             // eight reads in flight, then eight additions in order
              double a_[8];
#pragma unroll
              for (int y = 0; y < 8; y++) a_[y] = s_w[e0 + z + y];
#pragma unroll
              for (int y = 0; y < 8; y++) { s_w[e0 + z + y] = run; if (!((skipm >> (z + y)) & 1ull)) run = run + a_[y]; }
            }
            for (; z < n; z++) { const double wz = s_w[e0 + z]; s_w[e0 + z] = run; if (!((skipm >> z) & 1ull)) run = run + wz; }
            wt = run;
          }
          __builtin_amdgcn_wave_barrier();
          const double wb = in ? s_w[e0 + lane] : 0.0;               // the weight this follower met
          // its map of the initiator flag, two bits per entry (entry x in bits 2x, 2x+1); identity for the idle lanes
          u32 tab = 0xE4u;                                           // 3,2,1,0
          if (in) {
            if (w2 * wb > 0.0) { tab = 0; for (int x = 0; x < 4; x++) tab |= (u32)(x > i2 ? x : i2) << (2 * x); }
            else if (fabs(wb) < fabs(w2)) { tab = 0; for (int x = 0; x < 4; x++) tab |= (u32)((x != 3 || !keep3) ? i2 : x) << (2 * x); }
            else if (fabs(wb) == fabs(w2)) { tab = 0; for (int x = 0; x < 4; x++) tab |= (u32)((x != 3 || !keep3) ? 0 : x) << (2 * x); }
          }
          for (int o = 1; o < 64; o <<= 1) {                         // lane l: followers l .. l + 2 o - 1 after this round, applied in order
            const u32 nxt = (u32)__shfl_down((int)tab, o, 64);
            const int am2 = __shfl_down(am, o, 64);
            if (lane + o < 64) {
              u32 c2 = 0;
              for (int x = 0; x < 4; x++) c2 |= ((nxt >> (2 * ((tab >> (2 * x)) & 3u))) & 3u) << (2 * x);      // nxt after tab
              tab = c2;
              am = am2 < am ? am2 : am;
            }
          }
          tab = (u32)__shfl((int)tab, 0, 64); am = __shfl(am, 0, 64);
          ini = (int)((tab >> (2 * ini)) & 3u);
          if (d != 0 && d != -2) { if (anym2) d = -2; else if (d != -1 && am < d) d = am; }
          wt = __shfl(wt, 0, 64);
          qq += n;
        }
        if (stops) break;
      }
      if (lane == 0) rnk[q] = finish_head(q, si, wt, ini, d, psg, qq);
    }
  }
  __syncthreads();
  // ---- rank of every kept walker inside the bucket (lo 16 bits: position, hi: index among the deterministic-space walkers):
  //      each thread scans a contiguous share of the merged order; then one look-back over the buckets
  BPROF(6);
  u32 *s_rank = (u32 *)sb;                              // the idle sort buffer
  // packed running value of the look-back: position (20 bits), index among the deterministic-space walkers (18), children (24)
  u64 ex_glob;
  {
    const int C = (T + BK_AT - 1) / BK_AT, beg = tid * C, end = (beg + C < T) ? beg + C : T;
    const bool ch = go.child_off != nullptr;
    u64 mine = 0;
    for (int q = beg; q < end; q++) { const unsigned short k = rnk[q]; if (k & 1) mine += 1ull | ((u64)(k >> 8) << 20) | (ch ? ((u64)s_nc[q] << 38) : 0ull); }
    u64 tot; u64 ex = bk_block_excl_scan(mine, &tot);
    for (int q = beg; q < end; q++) {
      const unsigned short k = rnk[q];
      if (k & 1) {
        s_rank[q] = (u32)(ex & 0xFFFull) | ((u32)((ex >> 20) & 0x3FFFFull) << 12);       // inside the bucket: position < 4096, deterministic index < 2^18
        const u64 add = 1ull | ((u64)(k >> 8) << 20) | (ch ? ((u64)s_nc[q] << 38) : 0ull);
        if (ch) s_nc[q] = (u32)(ex >> 38);
        ex += add;
      } else s_rank[q] = 0xFFFFFFFFu;
    }
    __syncthreads();                                 // ranks and child prefixes are read row-wise below
    // the determinants this step creates get their H_ii at the end of this kernel (no kernel of its own on a side stream, nothing to
    // join): the tables travel to the idle counters while the look-back waits for its predecessors
    stage_tab(s_tab, dev.tab, dev.tab_words);
    ex_glob = bk_lookback_wide(ba.state, b, tot);
    if (tid == 0 && b == B - 1) {
      const u64 all = ex_glob + tot, npos = all & 0xFFFFFull, ndet = (all >> 20) & 0x3FFFFull;
      sc->tot2 = npos | (ndet << 32); sc->nwalk = npos;
      if (ch) sc->n_children = all >> 38;
    }
    BPROF(9);
  }
  BPROF(7);
  // ---- compaction into the other buffer, reweighting (2487), estimator pieces (2573-2684, more_tools.f90:4041-4098), next gate.
  //      Two slots per thread and round: their loads (record, then C(T) probe) are in flight together.
  double st[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) st[k] = 0.0;
  constexpr int CB = 2;
  for (int q0b = tid; q0b < T; q0b += CB * BK_AT) {
    int si_[CB]; u32 r_[CB]; u64 up_[CB], dn_[CB], key_[CB]; double me_[CB], en_[CB], ed_[CB]; long long hh_[CB];
#pragma unroll
    for (int z = 0; z < CB; z++) {
      const int q = q0b + z * BK_AT;
      r_[z] = (q < T) ? s_rank[q] : 0xFFFFFFFFu;
      si_[z] = 0; up_[z] = 0; dn_[z] = 0; key_[z] = 0; me_[z] = 1e51; en_[z] = 1e51; ed_[z] = 1e51; hh_[z] = -2;
      if (r_[z] != 0xFFFFFFFFu) {
        const int si = (int)(m2s[q] & ~BK_STOP); si_[z] = si;
        if (si < R) { const long long ix = r_lo + si; up_[z] = w.up[ix]; dn_[z] = w.dn[ix]; me_[z] = w.me[ix]; en_[z] = w.en[ix]; ed_[z] = w.ed[ix]; key_[z] = rk[si]; }
        else { const u64 x = sa[si - R]; const SpawnRec *rp = w.sp + ((long long)(u32)x - n0); up_[z] = rp->up; dn_[z] = rp->dn; key_[z] = x >> 32; }
      }
    }
#pragma unroll
    for (int z = 0; z < CB; z++) if (r_[z] != 0xFFFFFFFFu && en_[z] > 1e50) hh_[z] = ct_lookup(hkey, hidx, hmask, key_[z]);
#pragma unroll
    for (int z = 0; z < CB; z++) if (hh_[z] != -2) { if (hh_[z] < 0) { en_[z] = 0.0; ed_[z] = 0.0; } else { en_[z] = cnum[hh_[z]]; ed_[z] = cden[hh_[z]]; } }
#pragma unroll
    for (int z = 0; z < CB; z++) {
      if (r_[z] == 0xFFFFFFFFu) continue;
      const int si = si_[z];
      const long long q0 = (long long)(ex_glob & 0xFFFFFull) + (long long)(r_[z] & 0xFFFu);
      const long long qd = (long long)((ex_glob >> 20) & 0x3FFFFull) + (long long)(r_[z] >> 12);
      const double wt = s_w[si] * p.rfi;
      const u32 fl = s_f[si];
      const int d = flg_impd(fl), ini = flg_init(fl), psg = flg_psign(fl);
      const double en = en_[z], ed = ed_[z];
      o.up[q0] = up_[z]; o.dn[q0] = dn_[z]; o.wt[q0] = wt; o.flg[q0] = fl;
      o.me[q0] = me_[z]; o.en[q0] = en; o.ed[q0] = ed;
      if (me_[z] > 1e50 && !(p.semi && d < 1)) {                   // k_diag's rule (walk_kernels.h): death/clone will want H_ii
        const int slot = atomicAdd(&s_hqn, 1);
        s_hq[slot] = (unsigned short)(r_[z] & 0xFFFu);
        if (slot < BK_HQ_DETS) { s_hdet[2 * slot] = up_[z]; s_hdet[2 * slot + 1] = dn_[z]; }
      }
      if (go.on) {
        u64 nc; double wc;
        gate_children(wt, go.cutoff, seed, go.step_next, key_[z], nc, wc);
        go.keys[q0] = (key_[z] << 32) | (u64)q0; go.wchild[q0] = wc;
        if (go.child_off) { const u64 off = (ex_glob >> 38) + (u64)s_nc[q0b + z * BK_AT]; go.child_off[q0] = off; gate_block_parents(go.bpar, off, nc, q0); } else go.nchild[q0] = nc;
      }
      if (d == 0 && p.semi && qd < p.nimp_cap) { loc_imp[qd] = (int)q0; o.irk[q0] = (u32)qd; if (fs.x_out) fs.x_out[qd] = wt; }
      st[0] += wt; st[1] += fabs(wt); st[8] += wt * wt;
      if (ini == 3) st[4] += wt * psg;
      if (d == 0 || (d == -2 && p.cti)) st[6] += fabs(wt);
      double e_num = en * wt, e_den = ed * wt;
      if (e_num != 0.0) {
        if (fabs(e_den) < 1e-22) e_den = fabs(e_den);
        st[2] += e_den; st[3] += e_num; st[9] += e_num * e_num; st[10] += e_den * e_den;
        st[11] += e_num * copysign(1.0, e_den); st[12] += fabs(e_den); st[5] += e_num * e_den;
      }
    }
  }
  BPROF(8);
  // ---- H_ii of the new determinants: 8 or 16 lanes per determinant (chemistry); any other system, one thread per determinant
  __syncthreads();
  BPROF(14);
  {
    const int nq = s_hqn;
    BPROF_VAL(15, nq);
    const long long base = (long long)(ex_glob & 0xFFFFFull);
    const int hg = bk_hii_group_lanes(*s_tab);
    // a pipelined step: the first BK_HQ_DEFER of them are left to spare blocks of the k_spawn that follows (they run beside its spawning
    // blocks; here they were 5.5 of a bucket's 35 us, at the end of its chain).  Every bucket writes its count, also a zero.
    int n_def = 0;
    if (ba.hq_cnt) {
      n_def = nq < BK_HQ_DEFER ? nq : BK_HQ_DEFER;
      for (int k = tid; k < n_def; k += BK_AT) ba.hq_pos[b * BK_HQ_DEFER + k] = (u32)(base + (long long)s_hq[k]);
      if (tid == 0) ba.hq_cnt[b] = (u32)n_def;
    }
#define BK_HII_PASSES(HG)                                                                                          \
    {                                                                                                              \
      const int G = tid / HG, g = tid % HG;                                                                        \
      double *sg = s_w + G * BK_HG_TERMS(HG);            /* the weights were last read by the compaction */        \
      for (int k0 = n_def; k0 < nq; k0 += BK_AT / HG) {                                                            \
        const int k = k0 + G; const bool valid = k < nq;                                                           \
        const long long q0 = valid ? base + (long long)s_hq[k] : 0;                                                \
        u64 u = 0, dd = 0;                                                                                         \
        if (valid) { if (k < BK_HQ_DETS) { u = s_hdet[2 * k]; dd = s_hdet[2 * k + 1]; } else { u = o.up[q0]; dd = o.dn[q0]; } }   \
        const double v = bk_hii_group<HG>(*s_tab, dev.integrals, u, dd, valid, sg, g);                             \
        if (valid && g == 0) o.me[q0] = v;                                                                         \
      }                                                                                                            \
    }
    // few determinants (a bucket creates 13 on average at the bench size, 28 at most): more lanes each -- the phase is the latency
    // of one lane's decode + fetch + sum chain, and a lane with 2 tasks is through sooner than one with 5
    if (nq == n_def) { }
    else if ((hg == 8 || hg == 16) && nq - n_def <= BK_AT / 32 && bk_hii_terms(*s_tab) <= BK_HG_TERMS(32)) BK_HII_PASSES(32)
    else if (hg == 8 && nq - n_def <= BK_AT / 16) BK_HII_PASSES(16)
    else if (hg == 8) BK_HII_PASSES(8)
    else if (hg == 16) BK_HII_PASSES(16)
    else {
      for (int k = n_def + tid; k < nq; k += BK_AT) {
        const long long q0 = base + (long long)s_hq[k];
        const u64 u = o.up[q0], dd = o.dn[q0];
        o.me[q0] = h_any(*s_tab, dev.integrals, u, dd, u, dd);
      }
    }
#undef BK_HII_PASSES
  }
  BPROF(11);
  // block sums: the 13 estimator pieces and the two pre-merge sums
#pragma unroll
  for (int k = 0; k < NSTAT + 2; k++) {
    double v = (k < NSTAT) ? st[k] : (k == NSTAT ? wabs : cnt);
    v = bk_wave_sum_lane63(v);
    if (lane == 63) s_red[wv][k] = v;
  }
  __syncthreads();
  if (tid < NSTAT + 2) {
    double v = 0.0;
    for (int q = 0; q < BK_AT / 64; q++) v += s_red[q][tid];
    if (tid < NSTAT) partials[(long long)b * NSTAT + tid] = v;
    else wabs_part[2 * b + (tid - NSTAT)] = v;
  }
  // how full the fullest bucket was (per mille of the caps): the host keeps the bucket path off while the head-room is thin
  BPROF(10);
  if (tid == 0) { const int fs = (1000 * S) / BK_CAP_S, ft = (1000 * T) / BK_CAP_T; atomicMax((unsigned int *)&sc->bk_fill, (unsigned int)(fs > ft ? fs : ft)); }
}
