// walk_kernels.h -- the kernels of one walker step: gate, scan inputs, spawn, annihilation, finish
// Textually included by sqmc_gpu.hip (one translation unit: the kernels share the ChemTab LDS
// image, the walker SoA types and the launch helpers defined there); not a standalone header.

// ===================================================================== step kernels

// C(T) lookup: open-addressed hash (linear probing, load <= 1/2) from the determinant's sort
// key to its row in the C(T) arrays.  Replaces the binary search of
// binary_search_list_and_update (more_tools.f90:4041-4098): one or two dependent reads
// instead of log2(n_ct) ~ 17; the 2 MB table is L2-resident.
#define CT_EMPTY (~0ull)
__device__ __forceinline__ u64 ct_hash(u64 k) { return sq_mix64(k); }
__device__ __forceinline__ long long ct_lookup(const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 mask, u64 key) {
  u64 h = ct_hash(key) & mask;
  while (true) {
    const u64 k = hkey[h];
    if (k == key) return (long long)hidx[h];
    if (k == CT_EMPTY) return -1;
    h = (h + 1) & mask;
  }
}
// hf_to_psit: sort key of a determinant that is not known to be a resident of the C(T) segment (a child, a caller's spawn)
__device__ __forceinline__ u64 psit_key(const PsitDev &ps, u64 rank) {
  u64 h = ct_hash(rank) & ps.hmask;
  while (true) {
    const u64 k = ps.hkey[h];
    if (k == rank) return rank;
    if (k == CT_EMPTY) return rank + ps.koff;
    h = (h + 1) & ps.hmask;
  }
}

// Everything k_finish does, as arguments: in the pipelined head the first block of the NEXT step's gate
// kernel does it (one launch less on the critical path).
// (struct FinArgs: sqmc_gpu.hip, in front of the context that keeps one)
__device__ void finish_all(const FinArgs &f, DevScalars *sc);
// spawn gate and child count of one walker (COUNTER discipline: the draw is keyed by step and the determinant's rank in
// (up, dn) order = its sort key `i`, so that it can be taken before the walker's position in the new list is known).  do_walk.f90:3577-3589
__device__ __forceinline__ void gate_children(double w, double cutoff, u64 seed, u64 step, u64 i, u64 &nchild, double &wchild) {
  bool spawn, use_wt;
  if (fabs(w) < cutoff) {
    Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 0, i);
    spawn = rng_draw(g) < fabs(w / cutoff); use_wt = false;
  } else { spawn = true; use_wt = true; }
  long long nc = 0; double wc = 0.0;
  if (spawn) {
    if (use_wt) { nc = llround(fabs(w)); if (nc < 1) nc = 1; wc = w / (double)nc; }
    else { nc = 1; wc = copysign(cutoff, w); }
  }
  nchild = (u64)nc; wchild = wc;
}
// What k_gate would compute for the NEXT step, written by k_anneal as it places a walker (pipelined steps: the
// parameters of the gate do not change any more, and the walker's key is at hand): one kernel less on the critical path.
struct GateOut { u64 *keys; u64 *nchild; double *wchild; double cutoff; u64 step_next; int on;
                 u64 *child_off;
                 u32 *vals; int pack;       // keys wider than 32 bits: the slot index travels in vals (put_key)
                 u32 *bpar;                 // with child_off: parent of the first child of every block of 256 children (k_spawn's parent_hint)
                 // hf_to_psit (set whether or not the gate is fused): slot of C(T) -> index in Psi_T or -1, and where k_anneal<., 1> leaves the
                 // merged weights of the Psi_T determinants -- k_psit_finish sums them for T^-1 while other blocks already overwrite the slots
                 const int *ps_of; double *ps_raw; };      // child_off != null (bucket tail only): the kernel also writes the next step's child offsets and total -- no scan launch
// the walker at q0 owns the children [off, off + nc): it is the parent of the first child of every block of 256 children that starts inside
__device__ __forceinline__ void gate_block_parents(u32 *__restrict__ bpar, u64 off, u64 nc, long long q0) {
  if (!bpar || nc == 0) return;
  for (u64 b = (off + 255) >> 8; (b << 8) < off + nc; b++) bpar[b] = (u32)q0;
}
// gate + child count.
__global__ void __launch_bounds__(TPB) k_gate(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, const double *__restrict__ wt,
                                              u64 *__restrict__ nchild, double *__restrict__ wchild, u64 *__restrict__ keys, u32 *__restrict__ vals,
                                              long long n_arg, StepP p, u64 seed, u64 step, DevScalars *sc, int pack, int n_on_device, FinArgs fin) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  const long long n = n_on_device ? (long long)sc->nwalk : n_arg;      // pipelined head: the grid covers an upper bound
  if (fin.on && blockIdx.x == 0) finish_all(fin, sc);                  // the last step's final sums and mail, before this step clears the scalars
  if (n_on_device && sc->retry) return;                                // pipelined head of a step whose predecessor's bucket tail gave up: the host re-runs that tail first
  if (i == 0) { sc->n_invalid = 0; sc->tot1 = 0; sc->tot2 = 0; sc->err = 0; }   // every writer of these runs after this kernel
  if (i >= n) return;
  const u64 kk = det_key(dev, up[i], dn[i]);
  put_key(keys, vals, i, kk + ((p.koff && i >= p.nct) ? p.koff : 0ull), pack);      // sort key of the walker itself (hf_to_psit: the survivors outside C(T) sort behind C(T))
  u64 nc; double wc;
  gate_children(wt[i], p.cutoff, seed, step, kk, nc, wc);
  if (p.koff && i == 0) { nc = 0; wc = 0.0; }      // hf_to_psit: all moves of the first state are deterministic (do_walk.f90:3574)
  nchild[i] = nc; wchild[i] = wc;
}

// REPLAY discipline: one lane walks the walkers in order, consuming the single rannyu
// stream exactly as the reference does, and records where every child starts in it.
// (proposal_method fast_heatbath: the same walk with propose_heatbath -- how many draws a proposal takes depends on the tables and on the
//  matrix elements it computes, so the lane runs the whole proposal, not just its control flow)
__global__ void __launch_bounds__(64) k_replay_prepass(const ChemTab *__restrict__ gtab, const u64 *__restrict__ up, const u64 *__restrict__ dn,
                                                       const double *__restrict__ wt, u64 *__restrict__ nchild, double *__restrict__ wchild,
                                                       u64 *__restrict__ child_off, u64 *__restrict__ child_state, long long n,
                                                       long long cap_children, StepP p, DevScalars *sc, ChemDev dev) {
  __shared__ ChemTab t;
  stage_tab(&t, gtab, tab_words_used(gtab->c2_stride));
  if (threadIdx.x != 0) return;
  sc->n_invalid = 0; sc->tot1 = 0; sc->tot2 = 0; sc->err = 0;
  Rng g; g.mode = 0; g.x = sc->lcg;
  u64 c = 0;
  for (long long i = 0; i < n; i++) {
    double w = wt[i]; bool spawn, use_wt;
    if (p.koff && i == 0) { nchild[i] = 0; wchild[i] = 0.0; child_off[i] = c; continue; }      // hf_to_psit: the first state neither draws nor spawns (do_walk.f90:3574)
    if (fabs(w) < p.cutoff) { spawn = rng_draw(g) < fabs(w / p.cutoff); use_wt = false; }
    else { spawn = true; use_wt = true; }
    long long nc = 0; double wc = 0.0;
    if (spawn) {
      if (use_wt) { nc = llround(fabs(w)); if (nc < 1) nc = 1; wc = w / (double)nc; }
      else { nc = 1; wc = copysign(p.cutoff, w); }
    }
    nchild[i] = (u64)nc; wchild[i] = wc; child_off[i] = c;
    u64 iu = up[i], id = dn[i];
    for (long long k = 0; k < nc; k++) {
      if ((long long)c < cap_children) child_state[c] = g.x;
      if (dev.hb.on) { u64 ju2[2], jd2[2]; double wj2[2]; propose_heatbath(t, dev.integrals, dev.hb, g, p.tau, iu, id, ju2, jd2, wj2); }
      else { u64 ju, jd; double pr; propose_any(t, g, iu, id, ju, jd, pr); }
      c++;
    }
  }
  child_off[n] = c;
  sc->n_children = c; sc->lcg = g.x;
}

// diagonal death/clone, do_walk.f90:3743-3793, of the 256 walkers of block `blk` (t: the ChemTab image in LDS, staged by the caller)
// fill_only: only the missing H_ii are computed (pipelined head: weights are not touched)
__device__ __forceinline__ void diag_block(const ChemTab &t, const ChemDev &dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, double *__restrict__ wt,
                                           const u32 *__restrict__ flg, double *__restrict__ me, long long n, const StepP &p, DevScalars *sc, int fill_only, long long blk) {
  // Only determinants first occupied in the last step lack H_ii (the 1e51 sentinel), and they sit anywhere in the
  // sorted list: computed in place every wavefront would pay the whole Slater-Condon sum for a few lanes.  The block
  // queues them in LDS and its first threads work the queue off densely.
  __shared__ int q[TPB];
  __shared__ double hq[TPB];
  __shared__ int wcnt[TPB / 64];
  const long long i = blk * TPB + threadIdx.x;
  const bool live = i < n && !(p.semi && flg_impd(flg[i]) < 1);
  double hii = live ? me[i] : 0.0;
  const bool need = live && hii > 1e50;
  const u64 bal = __ballot(need);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wcnt[wv] = __popcll(bal);
  __syncthreads();
  int base = 0, qn = 0;
  for (int k = 0; k < TPB / 64; k++) { if (k < wv) base += wcnt[k]; qn += wcnt[k]; }
  if (need) q[base + __popcll(bal & ((1ull << lane) - 1ull))] = threadIdx.x;
  __syncthreads();
  if (qn) {
    __shared__ double s_hg[(TPB / 16) * BK_HG_TERMS(16)];      // the groups' terms (chemistry or electron gas)
    if (bk_hii_group_lanes(t)) {
      // chemistry without time-reversal symmetry: 16 lanes per determinant fetch all of h_diag's ~50 integrals in one round trip and one
      // lane per sum adds them in the reference's order (hii_group.h: the same bits as the serial sum, without its ~13 dependent round trips)
      const int G = (int)threadIdx.x / 16, g = (int)threadIdx.x % 16;
      for (int k0 = 0; k0 < qn; k0 += TPB / 16) {
        const int k = k0 + G; const bool valid = k < qn;
        const long long j = valid ? blk * TPB + q[k] : 0;
        const u64 u = valid ? up[j] : 0ull, d = valid ? dn[j] : 0ull;
        const double v = bk_hii_group<16>(t, dev.integrals, u, d, valid, s_hg + G * BK_HG_TERMS(16), g);
        if (valid && g == 0) { me[j] = v; hq[q[k]] = v; }
      }
    } else if (heg_hii_group_ok(t)) {          // electron gas: the pair terms of a determinant spread over 16 lanes (hii_group.h)
      const int G = (int)threadIdx.x / 16, g = (int)threadIdx.x % 16;
      for (int k0 = 0; k0 < qn; k0 += TPB / 16) {
        const int k = k0 + G; const bool valid = k < qn;
        const long long j = valid ? blk * TPB + q[k] : 0;
        const u64 u = valid ? up[j] : 0ull, d = valid ? dn[j] : 0ull;
        const double v = heg_hii_group<16>(t, u, d, valid, s_hg + G * BK_HG_TERMS(16), g);
        if (valid && g == 0) { me[j] = v; hq[q[k]] = v; }
      }
    } else {
      for (int k = threadIdx.x; k < qn; k += TPB) {
        const long long j = blk * TPB + q[k];
        me[j] = hq[q[k]] = h_any(t, dev.integrals, up[j], dn[j], up[j], dn[j]);
      }
    }
    __syncthreads();
    if (need) hii = hq[threadIdx.x];
  }
  if (!live || fill_only) return;
  double f = 1.0 + p.tau * (p.e_trial - hii);
  if (f < 0) { if (p.reached > 1) sc->err = SQMC_ERR_NEG_DIAG; f = 0; }
  wt[i] = wt[i] * f;
}
__global__ void __launch_bounds__(TPB) k_diag(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, double *__restrict__ wt,
                                              const u32 *__restrict__ flg, double *__restrict__ me, long long n_arg, StepP p, DevScalars *sc, int fill_only) {
  if (fill_only && sc->retry) return;
  const long long n = fill_only ? (long long)sc->nwalk : n_arg;
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  diag_block(t, dev, up, dn, wt, flg, me, n, p, sc, fill_only, (long long)blockIdx.x);
}

// rank that owns a determinant (get_det_owner, mpi_routines.f90:419-445: any hash of the determinant mod the number of ranks)
__host__ __device__ __forceinline__ int det_owner(u64 key, int nranks) { return (int)((sq_mix64(key ^ 0xA5A5A5A5A5A5A5A5ull) >> 17) % (u64)nranks); }
// The reference's own ownership function, bit for bit (SQMC_OWNER_DJB): get_det_owner -> hash -> djb_hash,
// mpi_routines.f90:419-445, 257-289, 354-379, on the reference's 128-bit determinants (here: high halves zero).
//   hash = PRIME; for word in (det_up, det_dn + OFFSET): test = IEOR(word, X'5555555555555555') * PRIME;
//                 for j = 0, 8, ..., 120: hash = (ishft(hash,5) + hash) + ishft(test, -j)
//   owner = abs(mod(hash, ncores))                   wrapping INTEGER(16) arithmetic, logical shifts
// PRIME = 2^88 + 315 (the FNV-128 prime): x * PRIME = (x << 88) + 315 x.
struct U128 { u64 lo, hi; };
__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b) { U128 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull); return r; }
__host__ __device__ __forceinline__ U128 shr128(U128 a, int j) {
  U128 r;
  if (j == 0) return a;
  if (j < 64) { r.lo = (a.lo >> j) | (a.hi << (64 - j)); r.hi = a.hi >> j; } else { r.lo = a.hi >> (j - 64); r.hi = 0; }
  return r;
}
__host__ __device__ __forceinline__ U128 mul_prime128(U128 x) {
#ifdef __HIP_DEVICE_COMPILE__
  const u64 carry = __umul64hi(x.lo, 315ull);
#else
  const u64 carry = (u64)(((unsigned __int128)x.lo * 315u) >> 64);
#endif
  U128 r; r.lo = x.lo * 315ull; r.hi = x.hi * 315ull + carry + (x.lo << 24);
  return r;
}
__host__ __device__ __forceinline__ U128 djb_hash128(u64 up, u64 dn) {
  U128 hash; hash.lo = 0x13Bull; hash.hi = 0x1000000ull;
  for (int i = 0; i < 2; i++) {
    U128 tmp; tmp.lo = i ? dn : up; tmp.hi = 0;
    if (i) { U128 off; off.lo = 0x62B821756295C58Dull; off.hi = 0x6C62272E07BB0142ull; tmp = add128(tmp, off); }
    tmp.lo ^= 0x5555555555555555ull;
    const U128 test = mul_prime128(tmp);
    for (int j = 0; j < 128; j += 8) {
      U128 h32; h32.lo = hash.lo << 5; h32.hi = (hash.hi << 5) | (hash.lo >> 59);
      hash = add128(add128(h32, hash), shr128(test, j));
    }
  }
  return hash;
}
__host__ __device__ __forceinline__ int det_owner_djb(u64 up, u64 dn, int nranks) {
  if (nranks == 1) return 0;
  U128 h = djb_hash128(up, dn);
  if (h.hi >> 63) { h.lo = ~h.lo; h.hi = ~h.hi; h.lo += 1; if (h.lo == 0) h.hi += 1; }      // abs(mod(x, n)) = |x| mod n
  const u64 n = (u64)nranks, two64 = ((~0ull) % n + 1ull) % n;
  return (int)(((h.hi % n) * two64 + h.lo % n) % n);
}
#define SQMC_OWNER_MIX 0
#define SQMC_OWNER_DJB 1
__host__ __device__ __forceinline__ int det_owner_any(int mode, u64 key, u64 up, u64 dn, int nranks) {
  return mode == SQMC_OWNER_DJB ? det_owner_djb(up, dn, nranks) : det_owner(key, nranks);
}
// sharded steps: k_spawn also notes the destination rank of every child (nranks for a child that made no walker), the key of the bucketing pass
struct OwnerOut { u64 *okey; u32 *oval; int nranks; int mode; };      // okey == nullptr: off
// a spawned walker (or the "no walker" marker) into slot n0 + c.  do_walk.f90:3700-3731
__device__ __forceinline__ u64 spawn_emit(const ChemDev &dev, const WalkArr &w, u64 *__restrict__ keys, u32 *__restrict__ vals, long long n0, long long c,
                                          u32 pf, u64 ju, u64 jd, double wj, const StepP &p, u64 invalid_key, int pack, const OwnerOut &oo) {
  const long long k = n0 + c;
  if (p.koff && ju == dev.ps.first_up && jd == dev.ps.first_dn) wj = 0.0;      // hf_to_psit: no stochastic spawning onto the first state (do_walk.f90:3676, 7642)
  if (wj != 0.0) {
    const int pd = flg_impd(pf), pi = flg_init(pf);
    int d;
    if (pd == -2) d = p.cti ? 1 : 2; else d = (pd < 126 ? pd : 126) + 1;
    if (p.semi && pd == 0) d = -1;
    int ini = (pi >= 2) ? 1 : 0;
    if (p.cti && pd == -2) ini = 1;
    if (p.semi && pd == 0) ini = 1;
    // matrix_elements / e_num / e_den of a spawn are the 1e51 sentinel (do_walk.f90:3728-3730):
    // not stored, k_merge supplies them for every slot >= n0
    SpawnRec r; r.up = ju; r.dn = jd; r.wt = wj; r.flg = pack_flg(d, ini, 0);
    w.sp[c] = r;
    u64 key = det_key(dev, ju, jd);
    if (p.koff) key = psit_key(dev.ps, key);
    put_key(keys, vals, k, key, pack);
    if (oo.okey) { oo.okey[c] = (u64)det_owner_any(oo.mode, key, ju, jd, oo.nranks); oo.oval[c] = (u32)c; }
    return key;
  } else {
    w.sp[c].wt = 0.0; put_key(keys, vals, k, invalid_key, pack);     // sorts behind every real determinant
    if (oo.okey) { oo.okey[c] = (u64)oo.nranks; oo.oval[c] = (u32)c; }
    return invalid_key;
  }
}

// (A x)(row) with the row's products added in storage order (fast_sparse_matrix_multiply_upper_triangular, more_tools.f90:3622-3670,
// as k_prj_apply adds them): one wavefront, 64 products per round trip
__device__ __forceinline__ void prj_row_product(const PrjPre &pp, int row) {
  __shared__ double s_prod[TPB / 64][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int arow = pp.grow ? pp.grow[row] : row;
  const int b = pp.ptr[arow], e = pp.ptr[arow + 1];
  double y = 0.0;
  for (int base = b; base < e; base += 64) {
    const int k = base + lane;
    s_prod[wv][lane] = (k < e) ? pp.val[k] * pp.x[pp.col[k]] : 0.0;
    __builtin_amdgcn_wave_barrier();
    const int cnt = (e - base < 64) ? (e - base) : 64;
    if (cnt == 64) {
#pragma unroll
      for (int l = 0; l < 64; l++) y = y + s_prod[wv][l];
    } else {
      for (int l = 0; l < cnt; l++) y = y + s_prod[wv][l];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) pp.y[row] = y;
}
#ifdef SPAWN_PROF
__device__ unsigned long long g_prof[8 * 8192];
#define PROF(K) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (K)] = wall_clock64(); } while (0)
extern "C" int sqmc_gpu_debug_prof(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)); }
#else
#define PROF(K)
#endif
// one thread per child proposal; parent found by binary search in the child offsets
// HB: proposal_method fast_heatbath (two walker slots per child) -- a kernel of its own, so that the uniform proposal keeps its registers
// FUSE: the short-list extras (children grouped by key range for the bucket tail, the block with the final sums, the projector rows);
//       large populations run FUSE = 0, whose register and LDS budget is the plain spawn's (7 waves per SIMD)
template <int HB, int FUSE>
__global__ void __launch_bounds__(TPB) k_spawn(ChemDev dev, WalkArr w, const u64 *__restrict__ child_off, const double *__restrict__ wchild,
                                               const u64 *__restrict__ child_state, u64 *__restrict__ keys, u32 *__restrict__ vals,
                                               long long n0_arg, long long cap_all, StepP p, int mode, u64 seed, u64 step, u64 invalid_key, const DevScalars *sc,
                                               HostMail *mail, u64 cnt_seq, int pack, int n_on_device, OwnerOut oo, BucketArgs ba, FinArgs fin, PrjPre pp, int n_extra) {
  // The first n_extra blocks do not spawn (they come first in the grid so that they run beside the spawning blocks from the start:
  // behind them they waited for the thousands of empty blocks a capacity-sized grid has, and the kernel for them -- 32 against
  // 20 us).  Steps whose child offsets came out of the bucket tail have no scan launch to carry the last step's final sums:
  // block 0 does them (nothing it touches is read by the spawning blocks).  The blocks behind it multiply the deterministic
  // projector into last step's deterministic weights, one wavefront per row: the part of the projection that needs nothing
  // the host still has to decide (E_T enters in the tail, bucket_kernels.h).  The last of them makes the next bucket boundaries.
  __shared__ ChemTab t;
  extern __shared__ u32 s_part[];      // BK_PART_LDS(B) bytes when the launch partitions (at least 16 x terms doubles when its spare blocks compute H_ii), none otherwise (large populations keep their occupancy)
  if (FUSE && (int)blockIdx.x < n_extra) {
    const int xb = (int)blockIdx.x;
    if (fin.on && xb == 0) { finish_all(fin, const_cast<DevScalars *>(sc)); return; }
    if (n_on_device && sc->retry) return;
    int hb0 = xb - (fin.on ? 1 : 0);
    if ((ba.kb || ba.kb_out) && xb == n_extra - 1) {      // the last spare block: bucket boundaries (bucket_partition.h)
      bk_rebalance_block(ba, keys, n_on_device ? (long long)sc->nwalk : n0_arg);
      return;
    }
    if (hb0 < ba.hq_nblk) {
      // H_ii of the determinants the last step's short-list tail created: the buckets left their positions in the new list
      // (bucket_kernels.h).  One block per bucket, 16 lanes per determinant, terms in the reference's order -- a bucket creates 13 on
      // average at the bench size, so a block is through after one pass, well inside the spawning blocks' time.  Death/clone of THIS
      // step's tail reads the values.
      stage_tab(&t, dev.tab, dev.tab_words);
      __syncthreads();
      const int bkt = hb0, cnt = (int)ba.hq_cnt[bkt];
      if (bk_hii_group_lanes(t)) {
        const int G = (int)threadIdx.x / 16, g = (int)threadIdx.x % 16;
        double *sg = (double *)s_part + G * bk_hii_terms(t);
        for (int k0 = 0; k0 < cnt; k0 += TPB / 16) {
          const int k = k0 + G; const bool valid = k < cnt;
          const long long q0 = valid ? (long long)ba.hq_pos[bkt * BK_HQ_DEFER + k] : 0;
          u64 u = 0, dd = 0;
          if (valid) { u = w.up[q0]; dd = w.dn[q0]; }
          const double v = bk_hii_group<16>(t, dev.integrals, u, dd, valid, sg, g);
          if (valid && g == 0) w.me[q0] = v;
        }
      } else {
        for (int k = threadIdx.x; k < cnt; k += TPB) {
          const long long q0 = (long long)ba.hq_pos[bkt * BK_HQ_DEFER + k];
          const u64 u = w.up[q0], dd = w.dn[q0];
          w.me[q0] = h_any(t, dev.integrals, u, dd, u, dd);
        }
      }
      return;
    }
    hb0 -= ba.hq_nblk;
    const int row = hb0 * (TPB / 64) + (int)(threadIdx.x >> 6);
    if (pp.n_imp > 0 && row < pp.n_imp) prj_row_product(pp, row);
    return;
  }
  if (n_on_device && sc->retry) return;                                 // see k_gate
  const unsigned bx = blockIdx.x - (FUSE ? (unsigned)n_extra : 0u);      // index among the spawning blocks
  const long long n0 = n_on_device ? (long long)sc->nwalk : n0_arg;     // pipelined head: launched before the host learnt the walker count
  // the grid covers the free capacity of the walker arrays; the number of children is read from
  // device memory so that the launch does not wait for the host to learn it
  PROF(0);
  __shared__ u64 s_win[SPAWN_WIN];
  // short lists (ba.B > 0): the block also groups its children by key range for the bucket tail (bucket_kernels.h)
  const int pws = BK_PART_STRIDE(ba.B);
  u32 *s_spl = s_part; u32 *s_wcnt = s_part + pws;
  // Parent of child c = largest i with child_off[i] <= c.  The 256 children of a block have
  // neighbouring parents, so the block narrows [0,n0) for its first child with 256-way splits
  // (one round trip per level instead of log2(n0) dependent loads), then every thread finishes
  // inside a 1024-entry LDS window (global search only if it runs past it).  The first probe, the
  // table staging and the child count do not depend on each other: they are issued together.
  const long long c0 = (long long)bx * TPB;
  long long wlo = 0, whi = n0;                       // invariant: child_off[wlo] <= c0, answer for c0 in [wlo, whi)
  u64 pv = 0;
  if (ba.parent_hint) {                              // the kernel that wrote the offsets left the parent of this block's first child
    const long long h = (long long)ba.parent_hint[bx];
    wlo = (h < n0) ? h : (n0 > 0 ? n0 - 1 : 0); whi = wlo;      // (blocks past the last child read a stale word and return below)
  }
  if (whi - wlo > SPAWN_WIN) {
    const long long stepw = (whi - wlo + TPB - 1) / TPB, probe = wlo + (long long)threadIdx.x * stepw;
    pv = (probe < whi) ? child_off[probe] : ~0ull;
  }
  stage_tab(&t, dev.tab, dev.tab_words);
  const long long nchildren = (long long)sc->n_children;
  if (mail && bx == 0 && threadIdx.x == 0) {      // the host sizes the sort from this while the kernel runs
    mail->n_children = (u64)nchildren; __threadfence_system(); mail->cnt_seq = cnt_seq;
  }
  const long long spc = HB ? 2 : 1;             // walker slots per child: the heat-bath proposal may return a single AND a double
  if (c0 >= nchildren || n0 + spc * nchildren > cap_all) return;
  const bool part = FUSE && ba.B > 0 && (int)bx < ba.nsb;                  // rows beyond the room the host provided: the tail will see that and use its own partition
  if (part) bucket_partition_stage(s_spl, s_wcnt, pws, keys, n0, ba);     // resident keys [0, n0) of the same array the children's keys go to; a barrier follows below
  PROF(1);
  while (whi - wlo > SPAWN_WIN) {
    const long long stepw = (whi - wlo + TPB - 1) / TPB, probe = wlo + (long long)threadIdx.x * stepw;
    const int le = (probe < whi && pv <= (u64)c0) ? 1 : 0;
    const int cnt = __syncthreads_count(le);           // probes are sorted: the first cnt of them are <= c0
    const long long nlo = wlo + (long long)(cnt - 1) * stepw;
    whi = (nlo + stepw < whi) ? nlo + stepw : whi; wlo = nlo;
    if (whi - wlo > SPAWN_WIN) {
      const long long stepw2 = (whi - wlo + TPB - 1) / TPB, probe2 = wlo + (long long)threadIdx.x * stepw2;
      pv = (probe2 < whi) ? child_off[probe2] : ~0ull;
    }
  }
  for (int k = threadIdx.x; k < SPAWN_WIN; k += TPB) {  // window keeps going past whi: later children of the block live there
    const long long i = wlo + k;
    s_win[k] = (i < n0) ? child_off[i] : ~0ull;
  }
  __syncthreads();
  PROF(2);
  const long long c = c0 + threadIdx.x;
  const bool active = c < nchildren;
  u64 ckey = invalid_key;
  if (active) {
    long long ip;
    if (s_win[SPAWN_WIN - 1] <= (u64)c) {                 // beyond the window (many childless parents in between)
      long long lo = wlo + SPAWN_WIN - 1, hi = n0;
      while (hi - lo > 1) { long long mid = (lo + hi) >> 1; if (child_off[mid] <= (u64)c) lo = mid; else hi = mid; }
      ip = lo;
    } else {
      int lo = 0, hi = SPAWN_WIN - 1;                     // s_win[lo] <= c < s_win[hi]
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_win[mid] <= (u64)c) lo = mid; else hi = mid; }
      ip = wlo + lo;
    }
    Rng g; g.mode = mode;
    g.x = (mode == 0) ? child_state[c] : sq_counter_key(seed, step, 1, (u64)c);
    const u64 iu = w.up[ip], id = w.dn[ip];
    const u32 pflg = w.flg[ip]; const double wch = wchild[ip];     // needed at the end: fetched in the same round trip
    u64 ju, jd; double prob;
    PROF(3);
    if (HB) {                    // proposal_method fast_heatbath: both slots of the child are written (weight 0: no walker), do_walk.f90:3604-3611
      u64 ju2[2], jd2[2]; double wj2[2];
      propose_heatbath(t, dev.integrals, dev.hb, g, p.tau, iu, id, ju2, jd2, wj2);
      spawn_emit(dev, w, keys, vals, n0, 2 * c, pflg, ju2[0], jd2[0], wch * wj2[0], p, invalid_key, pack, oo);
      spawn_emit(dev, w, keys, vals, n0, 2 * c + 1, pflg, ju2[1], jd2[1], wch * wj2[1], p, invalid_key, pack, oo);
    } else {
    const int level = propose_any(t, g, iu, id, ju, jd, prob);
    PROF(4);
    double wj = 0.0;
    if (level > 0) {
      wj = proposal_weight(t, dev.integrals, p.tau, iu, id, ju, jd, level, prob);
      wj = wch * wj;
    }
    ckey = spawn_emit(dev, w, keys, vals, n0, c, pflg, ju, jd, wj, p, invalid_key, pack, oo);
    }
  }
  if (part) bucket_partition_block(s_spl, s_wcnt, pws, active && ckey != invalid_key, (u32)ckey, (ckey << 32) | (u64)(n0 + c), (long long)bx, ba);
  PROF(5);
}

__global__ void __launch_bounds__(TPB) k_main_keys(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, u64 *__restrict__ keys,
                                                   u32 *__restrict__ vals, long long n, int pack) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) put_key(keys, vals, i, det_key(dev, up[i], dn[i]) + ((dev.ps.koff && i >= dev.ps.n_ct) ? dev.ps.koff : 0ull), pack);
}

// deterministic projection: x = w(loc); y = A x (rows summed in the reference's order);
// w(loc) += y + (E_T*tau)*x.   do_walk.f90:2262, 2290, 2321-2323
__global__ void __launch_bounds__(TPB) k_prj_gather(const double *__restrict__ wt, const int *__restrict__ loc, double *__restrict__ x, long long n) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) x[i] = wt[loc[i]];
}
// one wavefront per row: the 64 products of a chunk are formed in parallel (coalesced loads)
// and parked in LDS, then added in storage order (LDS broadcast reads, only the fp64 adds
// are on the dependent chain), so y is bit-identical to the reference's sequential
// accumulation even for the HF row that touches the whole deterministic space.
__global__ void __launch_bounds__(TPB) k_prj_apply(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                   const double *__restrict__ x, const int *__restrict__ loc, double *__restrict__ wt,
                                                   long long n, double e_trial, double tau) {
  __shared__ double sprod[TPB / 64][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * (TPB / 64) + wv;
  if (i >= n) return;
  const int b = ptr[i], e = ptr[i + 1];
  double y = 0.0;
  for (int base = b; base < e; base += 64) {
    const int k = base + lane;
    sprod[wv][lane] = (k < e) ? val[k] * x[col[k]] : 0.0;
    __builtin_amdgcn_wave_barrier();
    const int cnt = (e - base < 64) ? (e - base) : 64;
    if (cnt == 64) {
#pragma unroll
      for (int l = 0; l < 64; l++) y = y + sprod[wv][l];
    } else {
      for (int l = 0; l < cnt; l++) y = y + sprod[wv][l];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    y = y + e_trial * tau * x[i];
    wt[loc[i]] = wt[loc[i]] + y;
  }
}
__global__ void __launch_bounds__(TPB) k_scale(double *__restrict__ v, long long n, double r) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) v[i] = v[i] * r;
}

// integer ** integer of the reference (0**0 = 1)
__device__ __forceinline__ double ipow_d(int b, int e) { double r = 1.0; for (int i = 0; i < e; i++) r *= (double)b; return r; }

// Annihilation + initiator rules: one thread per run of equal determinants in the sorted
// order.  Within a run the original walker comes first and spawns keep creation order
// (stable sort), so the pairwise combination below is the reference's left-to-right scan.
// do_walk.f90:5866-6083, check_initiator 6838-6872.
struct MergedRec { u64 up, dn; double wt, me, en, ed; u32 flg; int d; u64 f; };   // f: bit 0 = kept after the merge, bit 32 = small weight, to be rounded
// block sum of the two pre-merge partials into row `tile` (all 256 threads)
__device__ __forceinline__ void store_wabs(double *__restrict__ wabs_part, long long tile, double wabs, double cnt) {
  __shared__ double red[2][TPB / 64];
  double v = wabs, q = cnt;
  for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); q += __shfl_down(q, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) { wabs_part[2 * tile] = red[0][0] + red[0][1] + red[0][2] + red[0][3]; wabs_part[2 * tile + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3]; }
}
// The merged walker of sorted slot j (f = 0 for slots that are not the head of a run or are
// discarded).  The 64 lanes of a wavefront call it together on 64 CONSECUTIVE slots (lane l: slot
// j0 + l).  Every lane fetches the record of its own slot -- one gather for the whole row, no
// dependent chain per follower -- and parks weight and flags in LDS; the head of a run then folds
// its followers in storage order out of LDS (the reference's left-to-right scan).  Only a run
// that leaves the row needs more loads: the wavefront fetches it 64 records per round trip.
// load_slot also adds the slot's share of the sums over the pre-merge list
// (my_w_abs_before_merge_cum, nwalk_before_merge; do_walk.f90:2347-2349) to wabs / cnt.
struct SlotIn { u64 key; SpawnRec h; double me, en, ed; bool head, real; };
// all global loads of one slot (independent of every other slot: a thread issues those of its ITEMS slots together)
__device__ __forceinline__ SlotIn load_slot(const WalkArr &w, const u64 *__restrict__ skey, const u32 *__restrict__ perm,
                                            long long j, long long n0, long long n_all, u64 invalid_key, int pack, double &wabs, double &cnt) {
  SlotIn in; in.key = invalid_key; in.h.up = 0; in.h.dn = 0; in.h.wt = 0.0; in.h.flg = 0; in.me = 1e51; in.en = 1e51; in.ed = 1e51;
  const int lane = threadIdx.x & 63;
  const bool valid = j < n_all;
  u32 t = 0;
  if (valid) { in.key = get_key(skey, j, pack); t = get_perm(skey, perm, j, pack); }
  u64 kprev = __shfl_up(in.key, 1, 64);                  // key of the slot before: the neighbouring lane has it
  if (lane == 0) kprev = (valid && j > 0) ? get_key(skey, j - 1, pack) : invalid_key;
  in.real = valid && in.key != invalid_key;              // children that produced no walker sort last
  in.head = in.real && !(j > 0 && kprev == in.key);
  // the head of a run is the resident walker if there is one (stable sort), else the first spawn;
  // every later walker of a run is a spawn (walkers are unique): its cached values are the 1e51
  // sentinel, so the reference's min() merges leave me / en / ed unchanged
  if (in.real) {
    if ((long long)t >= n0) in.h = w.sp[t - n0];
    else { in.h.up = w.up[t]; in.h.dn = w.dn[t]; in.h.wt = w.wt[t]; in.h.flg = w.flg[t]; in.me = w.me[t]; in.en = w.en[t]; in.ed = w.ed[t]; }
    wabs += fabs(in.h.wt); cnt += 1.0;
  }
  return in;
}
#define MERGE_AHEAD 4
#define MERGE_SELF 8
#define SLOT_STOP 0x80000000u     // in the staged flag word: this slot starts a run or holds no walker
// stage weight and flags of a slot at its place in the tile (LDS); the block synchronises before folding
__device__ __forceinline__ void stage_slot(const SlotIn &in, double *__restrict__ s_w, u32 *__restrict__ s_f, int idx) {
  s_w[idx] = in.h.wt; s_f[idx] = (u32)in.h.flg | ((in.head || !in.real) ? SLOT_STOP : 0u);
}
// fold the run that starts at in-tile slot idx (if `in` is a head) out of the staged tile; a run that
// leaves the tile is continued from HBM by the head's wavefront, 64 records per round trip
// PSIT (hf_to_psit, merge_my_original_with_spawned3, do_walk.f90:6484-6833): a run whose key lies below p.koff belongs to a determinant
// of C(T) -- always a resident.  Spawns fold into it by the same pairwise rule (6544-6563 = 5897-5950 with the resident's
// imp_distance, -2 or 0, left alone by the table), but it is never discarded and its initiator test waits until T^-1 has been
// applied (2394-2462: k_psit_tinv / k_psit_finish).  Everything else -- a survivor outside C(T) (6590-6607), a new determinant
// (6642-6693) -- folds, is tested and discarded exactly as merge_original_with_spawned2 does it.
template <int PSIT>
__device__ __forceinline__ MergedRec fold_slot(const SlotIn &in, const double *__restrict__ s_w, const u32 *__restrict__ s_f, int idx, int tile_slots,
                                               const WalkArr &w, const u64 *__restrict__ skey, const u32 *__restrict__ perm,
                                               long long j, long long n0, long long n_all, const StepP &p, u64 invalid_key, int pack) {
  const long long n = n_all;
  MergedRec out; out.up = 0; out.dn = 0; out.wt = 0.0; out.me = 1e51; out.en = 1e51; out.ed = 1e51; out.flg = 0; out.d = 0; out.f = 0;
  __shared__ double s_w2[TPB / 64][64];
  __shared__ u32 s_f2[TPB / 64][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u64 key = in.key; const bool head = in.head;
  const SpawnRec h = in.h; const double me = in.me, en = in.en, ed = in.ed;
  double wt = h.wt;
  int ini = 0, d = 0, ps = 0;
  long long jj = j + 1;
#define MERGE_FOLD(W2, FS) do {                                                                     \
    const double w2_ = (W2); const u32 fs_ = (FS); const int i2 = flg_init(fs_), d2 = flg_impd(fs_); \
    const bool same_sign = (w2_ * wt > 0);                                                          \
    if (same_sign) { if (i2 > ini) ini = i2; }                                                      \
    if (d == -2) { if (d2 == 0) d = 0; }                                                            \
    else if (d2 == -2) { if (d != 0) d = -2; }                                                      \
    else if (d != 0 && d != -2) { int a_ = d2 < 0 ? -d2 : d2; if (a_ < d) d = a_; }                 \
    if (!same_sign) {                                                                               \
      if (fabs(wt) < fabs(w2_)) { if (ini != 3 || p.r_init == -1.0) ini = i2; }                     \
      else if (fabs(wt) == fabs(w2_)) { if (ini != 3 || p.r_init == -1.0) ini = 0; }                \
    }                                                                                               \
    if (!(d == 0 && d2 == -1)) wt = wt + w2_;                                                       \
  } while (0)
  // A head folds the first MERGE_SELF followers of its run itself (most runs end there).  What is
  // left of a long run (a heavy determinant whose children land on a few neighbours: hundreds of
  // equal keys) is folded by the whole wavefront, 64 records at a time: out of the staged tile as
  // far as it reaches, then out of HBM.
  int lt = idx + 1;                              // next in-tile slot of this lane's run
  bool pending = false;
  if (head) {
    const u32 ft = (u32)h.flg;
    ini = flg_init(ft); d = flg_impd(ft); ps = flg_psign(ft);
    if (d == -1 && j > 0) d = 1;                 // 5985-5986 (the very first walker keeps -1 until the end)
    bool open = true;
    for (int k = 0; k < MERGE_SELF && lt < tile_slots; k++, lt++) { const u32 fl = s_f[lt]; if (fl & SLOT_STOP) { open = false; break; } MERGE_FOLD(s_w[lt], fl); }
    jj = j + (lt - idx);
    pending = open && (lt < tile_slots ? !(s_f[lt] & SLOT_STOP) : (j - idx + tile_slots < n && get_key(skey, j - idx + tile_slots, pack) == key));
  }
  // one chunk of up to 64 followers held one per lane (valid lanes form a prefix); returns its length
  auto fold_chunk = [&](bool valid, double w2, u32 f2, int leader) -> int {
    const u64 vb = __ballot(valid);
    const int cnt = (vb == ~0ull) ? 64 : __ffsll((long long)~vb) - 1;
    // A chunk whose weights all carry the sign of the running sum (the usual case: children of one
    // parent) needs no sign logic: the initiator flag is a maximum, imp_distance a minimum, and only
    // the additions stay in order (skipped terms become -0.0, which leaves a non-zero sum unchanged).
    const double wt_l = __shfl(wt, leader, 64); const int d_l = __shfl(d, leader, 64);
    const bool use = valid && lane < cnt;
    const int i2 = flg_init(f2), d2 = flg_impd(f2);
    const bool plain = !use || (((w2 > 0) == (wt_l > 0)) && fabs(w2) > 1e-150 && d2 != 0 && d2 != -2);
    if (fabs(wt_l) > 1e-150 && __ballot(plain) == ~0ull) {
      int im = use ? i2 : 0, dm = use ? (d2 < 0 ? -d2 : d2) : 1 << 20;
      for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(im, o, 64), b = __shfl_xor(dm, o, 64); im = a > im ? a : im; dm = b < dm ? b : dm; }
      s_w2[wv][lane] = (use && !(d_l == 0 && d2 == -1)) ? w2 : -0.0;
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) {
        if (im > ini) ini = im;
        if (d >= 1 && dm < d) d = dm;
        if (cnt == 64) {
#pragma unroll
          for (int l = 0; l < 64; l++) wt = wt + s_w2[wv][l];
        } else for (int l = 0; l < cnt; l++) wt = wt + s_w2[wv][l];
        jj += cnt;
      }
    } else {
      s_w2[wv][lane] = w2; s_f2[wv][lane] = f2;
      __builtin_amdgcn_wave_barrier();
      if (lane == leader) { for (int l = 0; l < cnt; l++) MERGE_FOLD(s_w2[wv][l], s_f2[wv][l]); jj += cnt; }
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
  };
  const long long jn = (j - idx) + tile_slots;   // first slot after the tile
  for (u64 pend = __ballot(pending); pend; pend &= pend - 1) {
    const int leader = __ffsll((long long)pend) - 1;
    int ltl = __shfl(lt, leader, 64);
    const u64 lkey = __shfl(key, leader, 64);
    // ---- the part of the run that lies in the staged tile
    while (ltl < tile_slots) {
      const int li = ltl + lane;
      const u32 fl = (li < tile_slots) ? s_f[li] : SLOT_STOP;
      const bool valid = !(fl & SLOT_STOP);
      const int cnt = fold_chunk(valid, valid ? s_w[li] : 0.0, fl, leader);
      ltl += cnt;
      if (cnt < 64) break;
    }
    if (ltl < tile_slots) continue;              // the run ended inside the tile
    // ---- the run reaches the end of the tile: the rest, if any, comes from HBM
    long long base = jn;
    for (bool more = true; more;) {
      // MERGE_AHEAD rows of 64 records are requested together (one latency for 256 records), then folded row by row
      double w2q[MERGE_AHEAD]; u32 f2q[MERGE_AHEAD]; bool vq[MERGE_AHEAD];
#pragma unroll
      for (int q = 0; q < MERGE_AHEAD; q++) {
        const long long ix = base + (long long)q * 64 + lane;
        vq[q] = ix < n && get_key(skey, ix, pack) == lkey;
        w2q[q] = 0.0; f2q[q] = 0;
        if (vq[q]) { const u32 sx = get_perm(skey, perm, ix, pack); const SpawnRec r2 = w.sp[sx - n0]; w2q[q] = r2.wt; f2q[q] = (u32)r2.flg; }
      }
#pragma unroll
      for (int q = 0; q < MERGE_AHEAD; q++) {
        if (!more) break;
        if (fold_chunk(vq[q], w2q[q], f2q[q], leader) < 64) more = false;
      }
      base += 64 * MERGE_AHEAD;
    }
  }
#undef MERGE_FOLD
  if (!head) return out;
  const bool ct_head = PSIT && key < p.koff;
  // check_initiator
  if (!ct_head) {
    const int dd = d - p.imind > 0 ? d - p.imind : 0;
    const double thr = p.r_init * ipow_d(dd, p.ipow), aw = fabs(wt);
    if (ini == 3 && p.r_init >= 0) { if (wt * ps < 1.0) wt = (double)ps; }
    else if (ini == 2 && ((aw <= thr && d > 0) || ((aw <= p.r_init && !p.cti) && d == -2))) ini = 1;
    else if (ini < 2 && ((aw > thr && d >= 0) || ((aw > p.r_init || p.cti) && d == -2))) ini = ini + 1;
  }
  int dtest = d;
  if (d == -1) { if (jj >= n || get_key(skey, jj, pack) == invalid_key) dtest = 1; d = 1; }   // 6032-6036 then the last-det test at 6038
  const bool discard = !ct_head && (((wt == 0.0 && (ini != 3 || p.r_init < 0)) || ini == 0) && dtest >= 1);
  out.up = h.up; out.dn = h.dn; out.wt = wt; out.flg = pack_flg(d, ini, ps); out.d = d;
  out.me = me; out.en = en; out.ed = ed;
  if (!discard) { out.f = 1ull; if (p.semi && d >= 1 && fabs(wt) < p.min_wt) out.f |= (1ull << 32); }
  return out;
}
__global__ void __launch_bounds__(TPB) k_merge(WalkArr w, WalkArr m, const u64 *__restrict__ skey, const u32 *__restrict__ perm,
                                               u64 *__restrict__ flags, double *__restrict__ wabs_part, long long n0, long long n_all, StepP p, u64 invalid_key,
                                               int pack) {
  const long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  double wabs = 0.0, cnt = 0.0;
  __shared__ double s_w[TPB]; __shared__ u32 s_f[TPB];
  const SlotIn in = load_slot(w, skey, perm, j, n0, n_all, invalid_key, pack, wabs, cnt);
  stage_slot(in, s_w, s_f, threadIdx.x);
  __syncthreads();
  const MergedRec r = fold_slot<0>(in, s_w, s_f, threadIdx.x, TPB, w, skey, perm, j, n0, n_all, p, invalid_key, pack);
  store_wabs(wabs_part, blockIdx.x, wabs, cnt);
  if (j >= n_all) return;
  flags[j] = r.f;
  if (!(r.f & 1ull)) return;                   // not the head of a run, or discarded: nothing to store
  m.up[j] = r.up; m.dn[j] = r.dn; m.wt[j] = r.wt; m.flg[j] = r.flg;
  m.me[j] = r.me; m.en[j] = r.en; m.ed[j] = r.ed;
}

// stochastic rounding of small weights (reduce_my_walker, do_walk.f90:7196-7254); RNG draws
// are taken in merged-walker order: REPLAY = skip-ahead of the rannyu LCG by the rank of the
// draw, COUNTER = stream keyed by the determinant's rank in (up, dn) order (its sort key).
__global__ void __launch_bounds__(TPB) k_round(WalkArr m, const u64 *__restrict__ flags, const u64 *__restrict__ pos, u64 *__restrict__ flags2,
                                               long long n_all, StepP p, int mode, u64 seed, u64 step, const DevScalars *sc, const u64 *__restrict__ skey, int pack) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j >= n_all) return;
  const u64 f = flags[j];
  if (!(f & 1ull)) { flags2[j] = 0; return; }
  double wt = m.wt[j];
  if (f >> 32) {
    const u64 ps = pos[j];
    double r;
    if (mode == 0) r = (double)lcg_skip(sc->lcg, (ps >> 32) + 1) * 3.552713678800500929355621337890625e-15;
    else { Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, get_key(skey, j, pack)); r = rng_draw(g); }   // keyed by the determinant's rank = its sort key
    if (r < (fabs(wt) / p.min_wt)) wt = copysign(p.min_wt, wt); else wt = 0.0;
    m.wt[j] = wt;
  }
  const int d = flg_impd(m.flg[j]);
  u64 f2 = 0;
  // reduce_my_walker drops zero weights outside the deterministic space (7222-7249), join_walker2 every zero weight (7071-7092)
  const bool drop = p.semi ? (wt == 0.0 && d >= 1) : (wt == 0.0);
  if (!drop) { f2 = 1ull; if (d == 0) f2 |= (1ull << 32); }
  flags2[j] = f2;
}

// join_walker2 (do_walk.f90:6990-7103), the non-semistochastic counterpart of the rounding: the
// small walkers of one sign are joined along the list -- the pair's weight goes to one of the two
// with probability proportional to its own weight, one draw per join -- until the running weight
// exceeds min_wt; positive walkers first, then negative ones.  Which walkers end a chain depends on
// the running sum, so the chain is followed by ONE lane; the 256 threads of the block only stream
// the merged list through LDS in 1024-walker tiles (coalesced) ahead of it.  Draws: REPLAY = the
// rannyu stream in join order, COUNTER = stream keyed by the later walker's rank among the merged walkers (the low word of pos),
// the key k_join_par uses too.
#define JOIN_TILE 1024
__global__ void __launch_bounds__(TPB) k_join(WalkArr m, const u64 *__restrict__ flags, const u64 *__restrict__ pos, long long n_all, StepP p,
                                              int mode, u64 seed, u64 step, DevScalars *sc) {
  __shared__ double s_wt[JOIN_TILE]; __shared__ unsigned int s_rank[JOIN_TILE]; __shared__ unsigned char s_cand[JOIN_TILE];
  u64 lcg = sc->lcg;
  for (int pass = 0; pass < 2; pass++) {
    bool ipair = false; long long j2 = 0; double w2 = 0.0;           // the walker currently carrying the chain and its weight (lane 0)
    for (long long base = 0; base < n_all; base += JOIN_TILE) {
      for (int k = threadIdx.x; k < JOIN_TILE; k += TPB) {
        const long long j = base + k;
        unsigned char cand = 0; double wt = 0.0; unsigned int rk = 0;
        if (j < n_all && (flags[j] & 1ull)) {
          wt = m.wt[j];
          cand = ((pass == 0 ? wt > 0.0 : wt < 0.0) && fabs(wt) < p.min_wt && flg_init(m.flg[j]) < 3) ? 1 : 0;
          rk = (unsigned int)(pos[j] & 0xFFFFFFFFull);
        }
        s_wt[k] = wt; s_rank[k] = rk; s_cand[k] = cand;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        const int lim = (n_all - base < JOIN_TILE) ? (int)(n_all - base) : JOIN_TILE;
        for (int k = 0; k < lim; k++) {
          if (!s_cand[k]) continue;
          const long long j = base + k; const double wi = s_wt[k];
          if (!ipair) { ipair = true; j2 = j; w2 = wi; continue; }
          const double wttot = fabs(wi) + fabs(w2);
          double r;
          if (mode == 0) { lcg = (lcg * SQ_LCG_MULT) & SQ_MASK48; r = (double)lcg * 3.552713678800500929355621337890625e-15; }
          else { Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, (u64)s_rank[k]); r = rng_draw(g); }
          if (r > (fabs(wi) / wttot)) { w2 = copysign(wttot, w2); m.wt[j2] = w2; m.wt[j] = 0.0; }
          else { m.wt[j2] = 0.0; w2 = copysign(wttot, wi); m.wt[j] = w2; j2 = j; }
          if (wttot > p.min_wt) ipair = false;
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0 && mode == 0) sc->lcg = lcg;
}

// The same joins in the COUNTER discipline, where the draw of a join is keyed by the later walker's rank and so needs no stream:
// WHERE the chains of joined walkers begin and end depends on the running sums alone, not on the draws.  One block per sign
// (positive and negative candidates never meet) streams the merged list in tiles of JP_TILE walkers and, per tile,
//   1. compacts the candidates in order into LDS; slot 0 stands for a chain left open by the tile before (its running weight);
//   2. lets EVERY candidate follow the chain that would begin at it, to where the running sum first exceeds min_wt (the sums in
//      the reference's order, |w_i| + running): where the next chain would begin.  A few steps each, all in parallel;
//   3. finds the chains that really begin -- the candidates reachable from the first one over those pointers -- by pointer
//      doubling (log2 levels of jump tables, marked from the longest jump down);
//   4. lets the first candidate of every real chain walk it once more, with the draws: who carries the weight in the end.  The
//      carrier gets the chain's weight, every other member zero -- the state the reference's one-by-one stores leave behind.
// One lane following every join of the list took 41 ms a step at 10^5 walkers (the whole semistochastic step takes 0.07).
#define JP_TILE 2048
#define JP_LV 12                       // 2^JP_LV > JP_TILE + 2
#define JP_T 1024                      // threads: every phase is a few dependent LDS reads per candidate, so the block wants all the waves a CU can hold
#define JP_CW 1024                     // chunk counts held in LDS at a time
// step 1 over the whole chip: the candidates of every chunk of JP_TILE walkers, compacted inside the chunk's own slots -- the positive
// ones from the front, the negative ones from the back (both in list order) -- and their two counts
__global__ void __launch_bounds__(TPB) k_join_gather(WalkArr m, const u64 *__restrict__ flags, const u64 *__restrict__ pos, long long n_all, StepP p,
                                                     double *__restrict__ ca, u64 *__restrict__ cb, u32 *__restrict__ counts) {
  constexpr int PER = JP_TILE / TPB;
  const int tid = (int)threadIdx.x;
  const long long cbase = (long long)blockIdx.x * JP_TILE;
  const long long len = (n_all - cbase < JP_TILE) ? n_all - cbase : JP_TILE;
  u64 f_[PER], ps_[PER]; double w_[PER]; u32 fl_[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const long long j = cbase + (long long)tid * PER + q;
    const bool in = j < n_all;
    f_[q] = in ? flags[j] : 0ull; w_[q] = in ? m.wt[j] : 0.0; fl_[q] = in ? m.flg[j] : 0u; ps_[q] = in ? pos[j] : 0ull;
  }
  unsigned mp = 0, mn = 0; int np_ = 0, nn_ = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const double wt = w_[q];
    if ((f_[q] & 1ull) && wt != 0.0 && fabs(wt) < p.min_wt && flg_init(fl_[q]) < 3) { if (wt > 0.0) { mp |= 1u << q; np_++; } else { mn |= 1u << q; nn_++; } }
  }
  u64 tot; const u64 ex = block_excl_scan_u64((u64)np_ | ((u64)nn_ << 32), &tot);
  int op = (int)(ex & 0xFFFFFFFFull), on = (int)(ex >> 32);
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const u64 rec = ((ps_[q] & 0xFFFFFFFFull) << 32) | (u64)(u32)(cbase + (long long)tid * PER + q);       // draw key (rank) | index in the list
    if (mp & (1u << q)) { ca[cbase + op] = fabs(w_[q]); cb[cbase + op] = rec; op++; }
    if (mn & (1u << q)) { ca[cbase + len - 1 - on] = fabs(w_[q]); cb[cbase + len - 1 - on] = rec; on++; }
  }
  if (tid == 0) counts[blockIdx.x] = (u32)(tot & 0xFFFFull) | ((u32)((tot >> 32) & 0xFFFFull) << 16);
}
// steps 2-4: one block per sign works through the chunks' candidates, as many chunks at a time as fit a tile (one round trip a tile)
__global__ void __launch_bounds__(JP_T) k_join_par(WalkArr m, const double *__restrict__ ca, const u64 *__restrict__ cb, const u32 *__restrict__ counts, long long n_all,
                                                   StepP p, u64 seed, u64 step) {
  __shared__ double s_a[JP_TILE + 2];
  __shared__ u32 s_rk[JP_TILE + 2], s_g[JP_TILE + 2];
  __shared__ unsigned short s_J[JP_LV][JP_TILE + 2];
  __shared__ unsigned char s_mark[JP_TILE + 2];
  __shared__ unsigned short s_cn[JP_CW];
  __shared__ u32 s_cg; __shared__ double s_ctot; __shared__ int s_copen;
  const int pass = (int)blockIdx.x, tid = (int)threadIdx.x;
  const double sgn = pass == 0 ? 1.0 : -1.0;
  const long long nchunks = (n_all + JP_TILE - 1) / JP_TILE;
  if (tid == 0) { s_copen = 0; s_cg = 0; s_ctot = 0.0; }
  __syncthreads();
  // phases 2-4 on the candidates 1 .. nc (slot 0: the chain the tile before left open)
  auto process = [&](int nc) {
    const int copen = s_copen;
    if (tid == 0) { s_a[0] = s_ctot; s_g[0] = s_cg; s_rk[0] = 0; }
    const int first = copen ? 0 : 1, END = nc + 1;
    __syncthreads();
    if (first <= nc) {
      // ---- 2. where the chain that would begin at i ends
      for (int i = first + tid; i <= nc; i += JP_T) {
        double run = s_a[i]; int k = i + 1;
        for (; k <= nc; k++) { run = s_a[k] + run; if (run > p.min_wt) break; }
        s_J[0][i] = (unsigned short)(k <= nc ? k + 1 : END);
        s_mark[i] = 0;
      }
      if (tid == 0) { s_J[0][END] = (unsigned short)END; s_mark[END] = 0; }
      __syncthreads();
      // ---- 3. the chains that really begin: reachable from `first`
      int nlv = 1; while ((1 << nlv) < END + 1 && nlv < JP_LV) nlv++;
      for (int l = 1; l < nlv; l++) {
        for (int i = first + tid; i <= END; i += JP_T) s_J[l][i] = s_J[l - 1][s_J[l - 1][i]];
        __syncthreads();
      }
      if (tid == 0) s_mark[first] = 1;
      __syncthreads();
      // (a mark set at this level may or may not be seen by another thread at this level: either way only candidates on the path
      //  are ever marked -- any jump from one of them lands on it -- and the binary expansion of a candidate's distance reaches it)
      for (int l = nlv - 1; l >= 0; l--) {
        for (int i = first + tid; i <= nc; i += JP_T) if (s_mark[i]) { const int t = s_J[l][i]; if (t <= nc) s_mark[t] = 1; }
        __syncthreads();
      }
      // ---- 4. every real chain once more, with the draws
      for (int i = first + tid; i <= nc; i += JP_T) {
        if (!s_mark[i]) continue;
        double run = s_a[i]; int holder = i, k = i + 1; bool closed = false;
        for (; k <= nc; k++) {
          const double ak = s_a[k], t = ak + run;
          Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, (u64)s_rk[k]);
          const double r = rng_draw(g);
          if (!(r > (ak / t))) holder = k;
          run = t;
          if (t > p.min_wt) { closed = true; break; }
        }
        const int last = closed ? k : nc;
        for (int q = i; q <= last; q++) m.wt[s_g[q]] = (q == holder) ? copysign(run, sgn) : 0.0;
        if (!closed) { s_cg = s_g[holder]; s_ctot = run; s_copen = 1; }       // the tile's last chain: carried on
        else if (s_J[0][i] == END) s_copen = 0;                                // ... or closed with the tile's last candidate
      }
    }
    __syncthreads();
  };
  long long ci = 0, wbase = -JP_CW;
  while (ci < nchunks) {
    if (ci >= wbase + JP_CW) {                           // the next JP_CW chunks' counts of this sign
      wbase = ci;
      for (int k = tid; k < JP_CW; k += JP_T) { const long long c = wbase + k; const u32 v = (c < nchunks) ? counts[c] : 0u; s_cn[k] = (unsigned short)(pass == 0 ? (v & 0xFFFFu) : (v >> 16)); }
      __syncthreads();
    }
    // as many chunks as fit the tile (a chunk alone always does); every thread finds the same cj
    long long cj = ci; int nc = 0;
    while (cj < nchunks && cj < wbase + JP_CW) { const int v = s_cn[cj - wbase]; if (nc + v > JP_TILE) break; nc += v; cj++; }
    // ---- 1. their candidates, in list order, in one round trip
    for (int o = tid; o < nc; o += JP_T) {
      long long c = ci; int k = o;
      for (;; c++) { const int v = s_cn[c - wbase]; if (k < v) break; k -= v; }
      const long long cb0 = c * JP_TILE;
      const long long len = (n_all - cb0 < JP_TILE) ? n_all - cb0 : JP_TILE;
      const long long slot = pass == 0 ? cb0 + k : cb0 + len - 1 - k;
      const u64 rec = cb[slot];
      s_a[1 + o] = ca[slot]; s_rk[1 + o] = (u32)(rec >> 32); s_g[1 + o] = (u32)rec;
    }
    process(nc);                                       // (its first barrier also publishes the tile)
    ci = cj;
  }
}

// builds the C(T) hash (ct_hash / ct_lookup at the top of this file)
__global__ void __launch_bounds__(TPB) k_ct_build(ChemDev dev, const u64 *__restrict__ cu, const u64 *__restrict__ cd, long long n,
                                                  u64 *__restrict__ hkey, u32 *__restrict__ hidx, u64 mask) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const u64 key = det_key(dev, cu[i], cd[i]);
  u64 h = ct_hash(key) & mask;
  while (true) {
    u64 prev = atomicCAS((unsigned long long *)&hkey[h], CT_EMPTY, key);
    if (prev == CT_EMPTY) { hidx[h] = (u32)i; return; }
    h = (h + 1) & mask;
  }
}
#define NSTAT 13
__device__ void finish_step(const double *__restrict__ partials, int nblocks, const double *__restrict__ wabs_part, int nwabs,
                            int mode, DevScalars *sc, u64 *__restrict__ scan_state, u32 *__restrict__ scan_ticket, int n_scan_words, int n_tickets, long long n_children, HostMail *mail, long long expect_nimp,
                            const double *__restrict__ partials2 = nullptr, int nblocks2 = 0);
// compaction into the walker arrays + reweight (2487) + estimator pieces (2573-2684 and
// binary_search_list_and_update, more_tools.f90:4041-4098) + per-block partial sums
__global__ void __launch_bounds__(TPB) k_compact(WalkArr m, WalkArr w, const u64 *__restrict__ flags2, const u64 *__restrict__ pos2,
                                                 int *__restrict__ loc_imp, const u64 *__restrict__ skey, const u64 *__restrict__ hkey,
                                                 const u32 *__restrict__ hidx, u64 hmask,
                                                 const double *__restrict__ cnum, const double *__restrict__ cden,
                                                 long long n_all, StepP p, double *__restrict__ partials, int pack) {
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
  for (long long j = (long long)blockIdx.x * TPB + threadIdx.x; j < n_all; j += (long long)gridDim.x * TPB) {
    if (!(flags2[j] & 1ull)) continue;
    const u64 ps = pos2[j]; const long long o = (long long)(ps & 0xFFFFFFFFull);
    const u64 u = m.up[j], dd = m.dn[j];
    const double wt = m.wt[j] * p.rfi;
    const u32 fj = m.flg[j]; const int d = flg_impd(fj), ini = flg_init(fj), psg = flg_psign(fj);
    double en = m.en[j], ed = m.ed[j];
    if (en > 1e50) {
      long long q = ct_lookup(hkey, hidx, hmask, get_key(skey, j, pack));
      if (q < 0) { en = 0.0; ed = 0.0; } else { en = cnum[q]; ed = cden[q]; }
    }
    w.up[o] = u; w.dn[o] = dd; w.wt[o] = wt; w.flg[o] = fj;
    w.me[o] = m.me[j]; w.en[o] = en; w.ed[o] = ed;
    if (d == 0 && p.semi && (long long)(ps >> 32) < p.nimp_cap) loc_imp[ps >> 32] = (int)o;
    s[0] += wt; s[1] += fabs(wt); s[8] += wt * wt;
    if (ini == 3) s[4] += wt * psg;
    if (d == 0 || (d == -2 && p.cti)) s[6] += fabs(wt);
    double e_num = en * wt, e_den = ed * wt;
    if (e_num != 0.0) {
      if (fabs(e_den) < 1e-22) e_den = fabs(e_den);
      s[2] += e_den; s[3] += e_num; s[9] += e_num * e_num; s[10] += e_den * e_den;
      s[11] += e_num * copysign(1.0, e_den); s[12] += fabs(e_den); s[5] += e_num * e_den;
    }
  }
  // deterministic block reduction (wave shuffles, then 4 wave sums in LDS)
  __shared__ double red[TPB / 64][NSTAT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NSTAT; k++) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSTAT) {
    double v = 0.0;
    for (int q = 0; q < TPB / 64; q++) v += red[q][threadIdx.x];
    partials[(long long)blockIdx.x * NSTAT + threadIdx.x] = v;
  }
}
// The whole annihilation tail of a semistochastic step in ONE kernel: merge of the sorted list
// (merge_slot), rank among the kept walkers by a decoupled look-back across tiles, stochastic
// rounding (reduce_my_walker, do_walk.f90:7196-7254; draws exactly as k_round takes them), rank
// among the survivors by a second look-back, then compaction into the OTHER walker buffer with the
// reweighting, the C(T) lookup of first-visit determinants and the per-tile estimator sums
// (k_compact).  The merged walkers never leave the registers: the intermediate arrays, the two
// flag/position arrays and four launches of the unfused path (k_merge, scan, k_round, scan,
// k_compact) are gone.  Tiles are handed out by an atomic ticket (forward progress without
// co-residency assumptions, as in scan_lookback_kernel); both look-backs use the same tile order.
#ifdef ANNEAL_PROF
__device__ unsigned long long g_aprof[8 * 16384];
#define APROF(K) do { if (threadIdx.x == 0 && tile < 16384) g_aprof[tile * 8 + (K)] = wall_clock64(); } while (0)
extern "C" int sqmc_gpu_debug_aprof(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_aprof), sizeof(g_aprof)); }
#else
#define APROF(K)
#endif
// PSIT: the hf_to_psit step (fold_slot<1>).  The C(T) segment comes out where it was (its determinants sort first and none is dropped) with
// the merged weights and flags only: T^-1, the initiator test, the reweighting and every sum over C(T) -- the whole energy estimator,
// do_walk.f90:2701-2722 -- are k_psit_tinv / k_psit_finish's; the walkers outside C(T) are finished here, without an estimator lookup.
// SPLIT (long lists, COUNTER discipline, pipelined): no tile waits for the tiles in front.  The kernel stops behind the rounding: the kept
// walkers of a tile go, compacted inside the tile, to a staging buffer with their counts (k_anneal_split_scan adds the counts up,
// k_anneal_place moves every tile to its place and does what needs the place: the estimator sums, the next gate, the row table).
// (struct AnnealStage: sqmc_gpu.hip, in front of the context that keeps one)
template <int ITEMS, int PSIT, int SPLIT>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) k_anneal(WalkArr w, WalkArr o, const u64 *__restrict__ skey, const u32 *__restrict__ perm, int *__restrict__ loc_imp,
                                                const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 hmask,
                                                const double *__restrict__ cnum, const double *__restrict__ cden,
                                                double *__restrict__ partials, double *__restrict__ wabs_part, long long n0, long long n_all, StepP p,
                                                u64 invalid_key, int pack, int mode, u64 seed, u64 step, DevScalars *sc,
                                                u64 *__restrict__ state1, u64 *__restrict__ state2, u32 *__restrict__ ticket, GateOut go, AnnealStage sg) {
  constexpr int TILE = TPB * ITEMS;
  __shared__ u32 s_tile; __shared__ u64 s_ex[2]; __shared__ u64 s_wsum[2][TPB / 64];
  __shared__ double s_w[TILE]; __shared__ u32 s_f[TILE];          // weight and flags of every slot of the tile
  if (threadIdx.x == 0) s_tile = SPLIT ? (u32)blockIdx.x : atomicAdd(ticket, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // every wavefront owns 64*ITEMS consecutive slots and takes them row by row (lane l: slot r*64 + l)
  const long long base = (long long)tile * TILE + (long long)wv * (64 * ITEMS) + lane;
  const bool last_tile = (long long)(tile + 1) * TILE >= n_all;
  APROF(0);
  u64 key[ITEMS]; double wabs = 0.0, cnt = 0.0;
  MergedRec r[ITEMS];
  // Three or more slots per thread do not fit the 128 registers of 4 waves per SIMD.  The cached H_ii and the two estimator
  // pieces of a walker pass through the merge untouched (every later walker of a run is a spawn without them) and are only
  // needed again at the compaction: they wait in LDS, not in scratch memory (which is HBM traffic: the scratch of all
  // resident waves is larger than the L2).
  constexpr bool PARK = ITEMS >= 3;
  __shared__ double s_park[PARK ? 3 : 1][PARK ? TILE : 1];
  {
    SlotIn in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { in[k] = load_slot(w, skey, perm, base + (long long)k * 64, n0, n_all, invalid_key, pack, wabs, cnt); key[k] = in[k].key; }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) stage_slot(in[k], s_w, s_f, wv * (64 * ITEMS) + k * 64 + lane);
    if (PARK) {
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const int q = wv * (64 * ITEMS) + k * 64 + lane;
        s_park[0][q] = in[k].me; s_park[PARK ? 1 : 0][q] = in[k].en; s_park[PARK ? 2 : 0][q] = in[k].ed;
        in[k].me = 1e51; in[k].en = 1e51; in[k].ed = 1e51;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) r[k] = fold_slot<PSIT>(in[k], s_w, s_f, wv * (64 * ITEMS) + k * 64 + lane, TILE, w, skey, perm, base + (long long)k * 64, n0, n_all, p, invalid_key, pack);
  }
  APROF(1);
  store_wabs(wabs_part, tile, wabs, cnt);
  APROF(2);
  // ---- REPLAY discipline only: rank among the rounding draws of the one rannyu stream (hi word;
  //      lo = rank among the kept walkers).  The COUNTER discipline keys a draw by its determinant,
  //      needs no rank, and so spares every tile the wait for the slowest earlier tile.
  u64 inc[ITEMS], carry = 0, ex = 0, tot = 0;
  if (mode == 0) {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { const u64 x = wave_incl_scan_u64(r[k].f, lane); inc[k] = x + carry; carry += __shfl(x, 63, 64); }
    if (lane == 0) s_wsum[0][wv] = carry;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPB / 64; q++) { if (q < wv) ex += s_wsum[0][q]; tot += s_wsum[0][q]; }
    if (threadIdx.x < 64) {
      const u64 e = lookback_exclusive(state1, tile, tot, threadIdx.x);
      if (threadIdx.x == 0) { s_ex[0] = e; if (last_tile) sc->tot1 = e + tot; }
    }
    __syncthreads();
    ex += s_ex[0];
  }
  APROF(3);
  u64 f2[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    f2[k] = 0;
    if (r[k].f & 1ull) {
      if (r[k].f >> 32) {
        double rr;
        if (mode == 0) { const u64 ex1 = ex + inc[k] - r[k].f; rr = (double)lcg_skip(sc->lcg, (ex1 >> 32) + 1) * 3.552713678800500929355621337890625e-15; }
        else { Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, key[k] - (PSIT ? p.koff : 0ull)); rr = rng_draw(g); }      // keyed by the determinant's rank = its sort key (PSIT: only determinants outside C(T) are rounded)
        if (rr < (fabs(r[k].wt) / p.min_wt)) r[k].wt = copysign(p.min_wt, r[k].wt); else r[k].wt = 0.0;
      }
      // reduce_my_walker drops zero weights outside the deterministic space (7222-7249)
      const bool drop = p.semi ? (r[k].wt == 0.0 && r[k].d >= 1) : (r[k].wt == 0.0);
      if (!drop) { f2[k] = 1ull; if (r[k].d == 0) f2[k] |= (1ull << 32); }
    }
  }
  // ---- pipelined steps of long lists (go.child_off): the next step's child counts are known as soon as the weights are final, so their
  //      running sum travels through a look-back of its own -- the words of the REPLAY rank, idle in the COUNTER discipline -- beside
  //      the positions': the next head needs no scan launch (26 us at 10^6 walkers, 140 at 10^7)
  const bool choff = go.on && go.child_off != nullptr;
  u32 ncv[ITEMS], cpre[ITEMS]; u64 ccarry = 0;
  if (choff) {
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      u64 nc = 0; double wc;
      if (f2[k] & 1ull) gate_children(r[k].wt * p.rfi, go.cutoff, seed, go.step_next, key[k], nc, wc);
      ncv[k] = (u32)nc;
      const u64 x = wave_incl_scan_u64(nc, lane);
      cpre[k] = (u32)(x + ccarry - nc); ccarry += __shfl(x, 63, 64);
    }
    if (lane == 0) s_wsum[0][wv] = ccarry;
  }
  // ---- final position (lo) and rank among the deterministic-space walkers (hi)
  carry = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) { const u64 x = wave_incl_scan_u64(f2[k], lane); inc[k] = x + carry; carry += __shfl(x, 63, 64); }
  if (lane == 0) s_wsum[1][wv] = carry;
  __syncthreads();
  ex = 0; tot = 0;
  u64 cex = 0, ctot = 0;
#pragma unroll
  for (int q = 0; q < TPB / 64; q++) { if (q < wv) ex += s_wsum[1][q]; tot += s_wsum[1][q]; }
  if (choff) {
#pragma unroll
    for (int q = 0; q < TPB / 64; q++) { if (q < wv) cex += s_wsum[0][q]; ctot += s_wsum[0][q]; }
  }
  if (SPLIT) {
    // the tile's own counts, and its kept walkers at their places inside the tile: nothing here depends on another tile
    if (threadIdx.x == 0) { sg.cnt_a[tile] = tot; sg.cnt_b[tile] = ctot; }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      if (!(f2[k] & 1ull)) continue;
      const u64 ex2 = ex + inc[k] - f2[k];
      const long long q = (long long)tile * TILE + (long long)(ex2 & 0xFFFFFFFFull);
      double me = r[k].me, en = r[k].en, ed = r[k].ed;
      if (PARK) { const int qq = wv * (64 * ITEMS) + k * 64 + lane; me = s_park[0][qq]; en = s_park[PARK ? 1 : 0][qq]; ed = s_park[PARK ? 2 : 0][qq]; }
      sg.up[q] = r[k].up; sg.dn[q] = r[k].dn; sg.key[q] = key[k]; sg.wt[q] = r[k].wt * p.rfi; sg.flg[q] = r[k].flg;
      sg.me[q] = me; sg.en[q] = en; sg.ed[q] = ed;
      sg.nc[q] = ncv[k]; sg.lch[q] = (u32)(cex + (u64)cpre[k]); sg.ldet[q] = (u32)(ex2 >> 32);
    }
    APROF(4); APROF(5);
    return;
  }
  if (threadIdx.x < 64) {
    const u64 e = lookback_exclusive(state2, tile, tot, threadIdx.x);
    if (threadIdx.x == 0) { s_ex[1] = e; if (last_tile) { sc->tot2 = e + tot; sc->nwalk = (e + tot) & 0xFFFFFFFFull; } }
  } else if (choff && threadIdx.x < 128) {        // the second wavefront, at the same time
    const u64 e = lookback_exclusive(state1, tile, ctot, threadIdx.x - 64);
    if (threadIdx.x == 64) { s_ex[0] = e; if (last_tile) sc->n_children = e + ctot; }
  }
  __syncthreads();
  ex += s_ex[1];
  if (choff) cex += s_ex[0];
  APROF(4);
  // ---- compaction, reweighting (2487), estimator pieces (2573-2684, more_tools.f90:4041-4098)
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    if (!(f2[k] & 1ull)) continue;
    const u64 ex2 = ex + inc[k] - f2[k];
    const long long q0 = (long long)(ex2 & 0xFFFFFFFFull);
    const bool ct_head = PSIT && key[k] < p.koff;
    const double wt = ct_head ? r[k].wt : r[k].wt * p.rfi;
    const int d = r[k].d, ini = flg_init(r[k].flg), psg = flg_psign(r[k].flg);
    double me = r[k].me, en = r[k].en, ed = r[k].ed;
    if (PARK) { const int q = wv * (64 * ITEMS) + k * 64 + lane; me = s_park[0][q]; en = s_park[PARK ? 1 : 0][q]; ed = s_park[PARK ? 2 : 0][q]; }
    if (!PSIT && en > 1e50) {
      const long long q = ct_lookup(hkey, hidx, hmask, key[k]);
      if (q < 0) { en = 0.0; ed = 0.0; } else { en = cnum[q]; ed = cden[q]; }
    }
    o.up[q0] = r[k].up; o.dn[q0] = r[k].dn; o.wt[q0] = wt; o.flg[q0] = r[k].flg;
    o.me[q0] = me; o.en[q0] = en; o.ed[q0] = ed;
    if (PSIT && ct_head) { const int kp = go.ps_of[q0]; if (kp >= 0) go.ps_raw[kp] = wt; }
    if (choff) {        // the child weight follows from the weight and the count (gate_children: w / n above the cutoff, +-cutoff for the one child below it)
      const u32 nc = ncv[k];
      put_key(go.keys, go.vals, q0, key[k], go.pack); go.child_off[q0] = cex + (u64)cpre[k]; gate_block_parents(go.bpar, cex + (u64)cpre[k], nc, q0);
      go.wchild[q0] = nc == 0 ? 0.0 : (fabs(wt) < go.cutoff ? copysign(go.cutoff, wt) : wt / (double)nc);
    } else if (go.on) {
      put_key(go.keys, go.vals, q0, key[k], go.pack);
      if (!ct_head) {          // hf_to_psit: the weights of the C(T) segment are not final yet -- k_psit_finish writes their gate; the draw is keyed by the determinant, not by where it sorts
        u64 nc; double wc;
        gate_children(wt, go.cutoff, seed, go.step_next, PSIT ? key[k] - p.koff : key[k], nc, wc);
        go.nchild[q0] = nc; go.wchild[q0] = wc;
      }
    }
    if (d == 0 && p.semi && (long long)(ex2 >> 32) < p.nimp_cap) loc_imp[ex2 >> 32] = (int)q0;
    if (ct_head) continue;
    s[0] += wt; s[1] += fabs(wt); s[8] += wt * wt;
    if (ini == 3) s[4] += wt * psg;
    if (d == 0 || (d == -2 && p.cti)) s[6] += fabs(wt);
    double e_num = en * wt, e_den = ed * wt;
    if (!PSIT && e_num != 0.0) {
      if (fabs(e_den) < 1e-22) e_den = fabs(e_den);
      s[2] += e_den; s[3] += e_num; s[9] += e_num * e_num; s[10] += e_den * e_den;
      s[11] += e_num * copysign(1.0, e_den); s[12] += fabs(e_den); s[5] += e_num * e_den;
    }
  }
  __shared__ double red[TPB / 64][NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) {
    double v = s[k];
    for (int q = 32; q > 0; q >>= 1) v += __shfl_down(v, q, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSTAT) {
    double v = 0.0;
    for (int q = 0; q < TPB / 64; q++) v += red[q][threadIdx.x];
    partials[(long long)tile * NSTAT + threadIdx.x] = v;
  }
  APROF(5);
}
// exclusive sums of the tiles' counts (a few thousand to a few ten thousand tiles), 4 TPB tiles a block: a block first adds up
// every count in front of its own (coalesced, all blocks at once -- no chain from block to block), then scans its own; the last
// block's totals are the step's.  Counts in, sums out: separate arrays, the blocks read each other's counts.
__global__ void __launch_bounds__(TPB) k_anneal_split_scan(const u64 *__restrict__ cnt_a, const u64 *__restrict__ cnt_b, u64 *__restrict__ off_a, u64 *__restrict__ off_b,
                                                           int ntiles, DevScalars *sc) {
  const int b0 = (int)blockIdx.x * 4 * TPB;
  u64 pa = 0, pb = 0;
  for (int i = (int)threadIdx.x; i < b0; i += 4 * TPB) {      // (b0 is a multiple of 4 TPB)
    u64 xa[4], xb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { xa[k] = cnt_a[i + k * TPB]; xb[k] = cnt_b[i + k * TPB]; }
#pragma unroll
    for (int k = 0; k < 4; k++) { pa += xa[k]; pb += xb[k]; }
  }
  const int i0 = b0 + (int)threadIdx.x * 4;
  u64 va[4], vb[4], sa = 0, sb = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { va[k] = (i0 + k < ntiles) ? cnt_a[i0 + k] : 0ull; vb[k] = (i0 + k < ntiles) ? cnt_b[i0 + k] : 0ull; sa += va[k]; sb += vb[k]; }
  u64 front_a, front_b, own_a, own_b;
  (void)block_excl_scan_u64(pa, &front_a);
  (void)block_excl_scan_u64(pb, &front_b);
  u64 ea = front_a + block_excl_scan_u64(sa, &own_a);
  u64 eb = front_b + block_excl_scan_u64(sb, &own_b);
#pragma unroll
  for (int k = 0; k < 4; k++) { if (i0 + k < ntiles) { off_a[i0 + k] = ea; off_b[i0 + k] = eb; } ea += va[k]; eb += vb[k]; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const u64 tot_a = front_a + own_a, tot_b = front_b + own_b;
    sc->tot2 = tot_a; sc->nwalk = tot_a & 0xFFFFFFFFull; sc->n_children = tot_b;
  }
}
// every tile's kept walkers from the staging buffer to their places in the new list, with everything that needed the place
// (the second half of k_anneal: reweighted weights are staged already; estimator pieces, row table, next gate and child offsets)
template <int ITEMS>
__global__ void __launch_bounds__(TPB) k_anneal_place(AnnealStage sg, WalkArr o, int *__restrict__ loc_imp, const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 hmask,
                                                      const double *__restrict__ cnum, const double *__restrict__ cden, double *__restrict__ partials, StepP p, int ntiles, GateOut go,
                                                      const DevScalars *__restrict__ sc) {
  constexpr int TILE = TPB * ITEMS;
  const int tile = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 base_a = sg.off_a[tile], base_b = sg.off_b[tile];
  const u64 end_a = (tile + 1 < ntiles) ? sg.off_a[tile + 1] : (u64)sc->tot2;      // (the scan left exclusive sums; the total is the step's walker count)
  double s[NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) s[k] = 0.0;
  const long long pos0 = (long long)(base_a & 0xFFFFFFFFull), det0 = (long long)(base_a >> 32);
  const u32 cnt = (u32)((end_a & 0xFFFFFFFFull) - (base_a & 0xFFFFFFFFull));
  for (u32 j = threadIdx.x; j < cnt; j += TPB) {
    const long long q = (long long)tile * TILE + j, q0 = pos0 + j;
    const u64 up = sg.up[q], dn = sg.dn[q], key = sg.key[q]; const double wt = sg.wt[q]; const u32 fl = sg.flg[q];
    double me = sg.me[q], en = sg.en[q], ed = sg.ed[q]; const u32 nc = sg.nc[q];
    const int d = flg_impd(fl), ini = flg_init(fl), psg = flg_psign(fl);
    if (en > 1e50) { const long long h = ct_lookup(hkey, hidx, hmask, key); if (h < 0) { en = 0.0; ed = 0.0; } else { en = cnum[h]; ed = cden[h]; } }
    o.up[q0] = up; o.dn[q0] = dn; o.wt[q0] = wt; o.flg[q0] = fl; o.me[q0] = me; o.en[q0] = en; o.ed[q0] = ed;
    put_key(go.keys, go.vals, q0, key, go.pack); go.child_off[q0] = base_b + (u64)sg.lch[q]; gate_block_parents(go.bpar, base_b + (u64)sg.lch[q], nc, q0);
    go.wchild[q0] = nc == 0 ? 0.0 : (fabs(wt) < go.cutoff ? copysign(go.cutoff, wt) : wt / (double)nc);
    const long long qd = det0 + (long long)sg.ldet[q];
    if (d == 0 && p.semi && qd < p.nimp_cap) loc_imp[qd] = (int)q0;
    s[0] += wt; s[1] += fabs(wt); s[8] += wt * wt;
    if (ini == 3) s[4] += wt * psg;
    if (d == 0 || (d == -2 && p.cti)) s[6] += fabs(wt);
    double e_num = en * wt, e_den = ed * wt;
    if (e_num != 0.0) {
      if (fabs(e_den) < 1e-22) e_den = fabs(e_den);
      s[2] += e_den; s[3] += e_num; s[9] += e_num * e_num; s[10] += e_den * e_den;
      s[11] += e_num * copysign(1.0, e_den); s[12] += fabs(e_den); s[5] += e_num * e_den;
    }
  }
  __shared__ double red[TPB / 64][NSTAT];
#pragma unroll
  for (int k = 0; k < NSTAT; k++) {
    double v = s[k];
    for (int q = 32; q > 0; q >>= 1) v += __shfl_down(v, q, 64);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSTAT) {
    double v = 0.0;
    for (int q = 0; q < TPB / 64; q++) v += red[q][threadIdx.x];
    partials[(long long)tile * NSTAT + threadIdx.x] = v;
  }
}
// posts the (all-reduced) scalars of a sharded step to the host mailbox
__device__ __forceinline__ void post_reduced(DevScalars *sc, HostMail *mail, u64 seq, const double *red_at = nullptr) {
  const double *red = red_at ? red_at : sc->red;
  for (int i = 0; i < 7; i++) sc->stats[i] = red[i];              // the global sums replace the local ones
  for (int i = 0; i < 16; i++) mail->stats[i] = sc->stats[i];
  mail->tot2 = sc->tot2; mail->err = err_decode(red[7]);          // the highest status any rank raised: every rank stops with it
  mail->retry = (u64)((sc->retry ? 1 : 0) | (fmod(red[7], 256.0) >= 1.0 ? 2 : 0));      // bit 0: this rank's bucket tail gave up, bit 1: some rank's did
  mail->bk_fill = (u64)sc->bk_fill;
  __threadfence_system();
  mail->seq = seq;
}
__global__ void k_post_mail(DevScalars *sc, HostMail *mail, u64 seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  post_reduced(sc, mail, seq);
}
// The final reduction stays a kernel of its own: folding it into the last-arriving block of
// k_compact needs an agent-scope release in every block and cost more than this launch.
__global__ void __launch_bounds__(TPB) k_finish(FinArgs f, DevScalars *sc) { finish_all(f, sc); }
// Steps whose gate was computed by k_anneal have no gate kernel to carry the last step's final sums: one extra block of
// the child-offset scan does them (and clears the step scalars, as the gate kernel would), next to the scan's own tiles.
// It must not touch the look-back words of the scan it rides on: f.scan_state / f.scan_ticket name the OTHER set.
struct FinExtra {
  static constexpr bool on = true;
  FinArgs f; DevScalars *sc;
  __device__ void operator()() const {
    if (f.on) finish_all(f, sc);
    if (threadIdx.x == 0) { sc->n_invalid = 0; sc->tot1 = 0; sc->tot2 = 0; sc->err = 0; }
  }
};
__device__ void finish_all(const FinArgs &f, DevScalars *sc) {
  if (f.on == 3) {            // sharded step: the sums were finished and all-reduced by kernels before this one; only the mail is left
    if (threadIdx.x == 0) post_reduced(sc, f.mail, f.seq, f.red);
    __syncthreads();
    return;
  }
  // the look-back words k_anneal used this step (two arrays, n_ftiles words each) are zero again for the next one
  for (int i = threadIdx.x; i < f.n_ftiles; i += TPB) { f.fstate[i] = 0; f.fstate[f.cap_ftiles + i] = 0; }
  if (threadIdx.x == 0 && f.n_ftiles > 0) *f.fticket = 0;
  finish_step(f.partials, f.nblocks, f.wabs_part, f.nwabs, f.mode, sc, f.scan_state, f.scan_ticket, f.n_scan_words, f.n_tickets, f.n_children, f.mail, f.expect_nimp, f.partials2, f.nblocks2);
  if (f.mail && threadIdx.x == 0) {
    __threadfence_system();
    f.mail->seq = f.seq;
  }
  __syncthreads();
}

// final reduction: sums block partials
// (fixed strided order + fixed tree: reproducible run to run), publishes the step's sums,
// advances the REPLAY stream and re-zeroes the look-back scan states for the next step
__device__ void finish_step(const double *__restrict__ partials, int nblocks, const double *__restrict__ wabs_part, int nwabs,
                            int mode, DevScalars *sc, u64 *__restrict__ scan_state, u32 *__restrict__ scan_ticket, int n_scan_words, int n_tickets, long long n_children, HostMail *mail, long long expect_nimp,
                            const double *__restrict__ partials2, int nblocks2) {
  __shared__ double red2[TPB / 64][NSTAT + 2];
  __shared__ double tot[NSTAT + 2];
  for (int i = threadIdx.x; i < n_scan_words; i += TPB) scan_state[i] = 0;
  if ((int)threadIdx.x < n_tickets) scan_ticket[threadIdx.x] = 0;
  // the scalars the last thread-0 section needs are requested now, with the partials
  u64 tot2 = 0, nch = 0; int err = 0;
  if (threadIdx.x == 0) { tot2 = sc->tot2; err = atomicExch(&sc->err, 0); nch = n_children >= 0 ? (u64)n_children : sc->n_children; }      // read and clear: the next step's death/clone kernel may already be running
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // every thread first adds up its rows (a row's 13 loads, and several rows, are in flight
  // together), then one shuffle tree per statistic
  double acc[NSTAT + 2];
#pragma unroll
  for (int k = 0; k < NSTAT + 2; k++) acc[k] = 0.0;
#pragma unroll 4
  for (int b = threadIdx.x; b < nblocks; b += TPB) {
#pragma unroll
    for (int k = 0; k < NSTAT; k++) acc[k] += partials[(long long)b * NSTAT + k];
  }
  for (int b = threadIdx.x; b < nblocks2; b += TPB) {
#pragma unroll
    for (int k = 0; k < NSTAT; k++) acc[k] += partials2[(long long)b * NSTAT + k];
  }
#pragma unroll 4
  for (int b = threadIdx.x; b < nwabs; b += TPB) { acc[NSTAT] += wabs_part[2 * b]; acc[NSTAT + 1] += wabs_part[2 * b + 1]; }
#pragma unroll
  for (int k = 0; k < NSTAT + 2; k++) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red2[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSTAT + 2) { double v = 0.0; for (int q = 0; q < TPB / 64; q++) v += red2[q][threadIdx.x]; tot[threadIdx.x] = v; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double o[16];
    o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2]; o[3] = tot[3]; o[4] = tot[4];
    o[5] = (double)(tot2 & 0xFFFFFFFFull); o[6] = tot[6];
    o[7] = tot[NSTAT + 1]; o[8] = tot[8]; o[9] = tot[9]; o[10] = tot[10];
    o[11] = tot[11]; o[12] = tot[12]; o[13] = tot[5]; o[14] = tot[NSTAT]; o[15] = (double)nch;
#pragma unroll
    for (int i = 0; i < 16; i++) sc->stats[i] = o[i];
    sc->nwalk = tot2 & 0xFFFFFFFFull;
    if (expect_nimp >= 0 || expect_nimp == -2) {      // sharded step (-2: a plain walk, no deterministic space to check): sums and status go through the all-reduce before anything is posted
      if (expect_nimp >= 0 && !err && !sc->retry && (long long)(tot2 >> 32) != expect_nimp) { err = SQMC_ERR_IMP_BROKEN; sc->err = err; }      // (a tail that gave up has counted nothing)
#pragma unroll
      for (int i = 0; i < 7; i++) sc->redl[i] = o[i];
      sc->redl[7] = err_encode(err) + (sc->retry ? 1.0 : 0.0);        // a bucket tail that gave up is a status too: every rank has to learn of it
    }
    if (mail) {          // the host's copy goes out from the same registers (finish_all fences and posts the sequence word)
#pragma unroll
      for (int i = 0; i < 16; i++) mail->stats[i] = o[i];
      mail->tot2 = tot2; mail->err = err;
      mail->retry = (u64)sc->retry; mail->bk_fill = (u64)sc->bk_fill;
    }
    sc->bk_fill = 0;
    if (mode == 0) sc->lcg = lcg_skip(sc->lcg, sc->tot1 >> 32);
  }
}
