// sqmc_gpu.hip -- libsqmc_gpu.so: HIP kernels (gfx950 / MI355X) + the C ABI of
// include/sqmc_gpu.h for sqmc's semistochastic walker step.
//
// One MC step (do_walk.f90:2171-2934, semistochastic chem, ncores=1) runs as this
// kernel pipeline on one HIP stream; all walker data stays in HBM as SoA arrays:
//
//   gate      per walker: low-weight spawn gate, nwalk_child, child weight   (3577-3589)
//   scan      child offsets (device-wide exclusive scan)
//   diag      per walker: death/clone factor 1+tau(E_T-H_ii), H_ii cached     (3743-3793)
//   spawn     per CHILD (load-balanced): uniform proposal + H_ij -> appended  (3599-3731)
//   project   deterministic core: gather, CSR matvec, scatter-add             (2255-2325)
//   sort      stable LSD radix sort of (up,dn) keys                           (5169-5197)
//   merge     one thread per determinant segment: annihilation + initiator    (5866-6083)
//   round     stochastic rounding of small weights, compaction (2 scans)      (7196-7254)
//   estimate  C(T) lookup, reweight, 13 block-reduced sums                    (2487-2790)
//
// Everything is HBM/latency bound integer + fp64 work: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include "../../include/sqmc_gpu.h"
#include "chem_device.h"
#include "scan_sort.h"

#define TPB 256
#define SPAWN_WIN 1024
static thread_local std::string g_err;
static int fail(int code, const std::string &m) { g_err = m; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(SQMC_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
static inline int nblk(long long n, int tpb = TPB) { return (int)((n + tpb - 1) / tpb); }

// ------------------------------------------------------------------ walker SoA in HBM
// A spawned walker as ONE 32-byte record (the wire record of the sharded exchange as well):
// the annihilation kernel gathers spawns in sorted order, i.e. at random, and a record costs one
// memory sector where four SoA fields cost four.  Resident walkers stay SoA (they are streamed).
struct __attribute__((aligned(32))) SpawnRec { u64 up, dn; double wt; u64 flg; };
struct WalkArr {
  SpawnRec *sp;            // spawn c of the step lives in sp[c]; walker slots >= nwalk of the SoA arrays are unused during a step
  u64 *up, *dn; double *wt; u32 *flg; double *me, *en, *ed;
};
// imp_distance / initiator / perm_sign packed in one word: a gather through the sort
// permutation costs one access instead of three
__host__ __device__ __forceinline__ u32 pack_flg(int impd, int init, int psign) {
  return (u32)(impd & 0xFF) | ((u32)(init & 0xFF) << 8) | ((u32)(psign & 0xFF) << 16);
}
__host__ __device__ __forceinline__ int flg_impd(u32 f) { return (int)(int8_t)(f & 0xFF); }
__host__ __device__ __forceinline__ int flg_init(u32 f) { return (int)(int8_t)((f >> 8) & 0xFF); }
__host__ __device__ __forceinline__ int flg_psign(u32 f) { return (int)(int8_t)((f >> 16) & 0xFF); }
// Sort records.  When the determinant key fits 32 bits (C2 cc-pVDZ: 28) the walker index rides in
// the low half of the same 64-bit word, so every radix pass moves ONE word per element (the scatter
// is one 8-byte write instead of an 8- and a 4-byte one to two places); otherwise keys and
// indices are two arrays.
__device__ __forceinline__ void put_key(u64 *__restrict__ keys, u32 *__restrict__ vals, long long k, u64 key, int pack) {
  if (pack) keys[k] = (key << 32) | (u64)k; else { keys[k] = key; vals[k] = (u32)k; }
}
__device__ __forceinline__ u64 get_key(const u64 *__restrict__ skey, long long j, int pack) { return pack ? (skey[j] >> 32) : skey[j]; }
__device__ __forceinline__ u32 get_perm(const u64 *__restrict__ skey, const u32 *__restrict__ perm, long long j, int pack) {
  return pack ? (u32)skey[j] : perm[j];
}
static int alloc_walk(WalkArr &a, long long n, bool with_spawn_records) {
  a.sp = nullptr;
  if (with_spawn_records) HIPCHK(hipMalloc(&a.sp, n * sizeof(SpawnRec)));
  HIPCHK(hipMalloc(&a.up, n * 8)); HIPCHK(hipMalloc(&a.dn, n * 8)); HIPCHK(hipMalloc(&a.wt, n * 8));
  HIPCHK(hipMalloc(&a.flg, n * 4));
  HIPCHK(hipMalloc(&a.me, n * 8)); HIPCHK(hipMalloc(&a.en, n * 8)); HIPCHK(hipMalloc(&a.ed, n * 8));
  return 0;
}
static void free_walk(WalkArr &a) {
  hipFree(a.sp); hipFree(a.up); hipFree(a.dn); hipFree(a.wt); hipFree(a.flg);
  hipFree(a.me); hipFree(a.en); hipFree(a.ed);
}

struct StepP {       // device copy of sqmc_step_params + derived values
  double tau, e_trial, rfi, r_init, min_wt, cutoff;
  int ipow, imind, cti, semi, reached;
  int nimp_cap;            // entries of the loc_imp array (set by step_tail): spawn records with made-up flags cannot push an index past it
};

// device-side scalars of a step
struct DevScalars {
  u64 lcg;                 // REPLAY stream state (48 bits)
  u64 n_children;          // total child proposals this step
  u64 n_invalid;           // children that produced no walker (weight 0)
  u64 tot1;                // packed scan total 1: lo = kept after merge, hi = rounding draws
  u64 tot2;                // packed scan total 2: lo = final walkers, hi = det-space walkers
  u64 nwalk;               // walkers after the last finished step (k_finish); the next step's head kernels read it when the host does not know it yet
  int err;                 // SQMC_ERR_* raised on device
  int pad;
  double stats[16];
};

// Mailbox in pinned host memory that the GPU writes directly (no copy kernel, no interrupt): the
// child count as soon as k_spawn starts, the step's sums at the end of k_finish.  The host spins
// on the sequence words.  Data first, system-scope fence, then the sequence word.
struct HostMail {
  volatile u64 seq; u64 tot2; long long err; double stats[16];
  volatile u64 cnt_seq; u64 n_children;
};

#define NTIMERS 32
struct sqmc_gpu_ctx {
  hipStream_t st;
  ChemTab htab; ChemTab *d_tab; double *d_ints; ChemDev dev;
  int *d_hb_r, *d_hb_s; double *d_hb_absH; long long *d_pq_ind; int *d_pq_count;
  long long mwalk, nwalk;
  WalkArr w, m;                        // walkers (main + appended spawns), merge results
  u64 *d_nchild; u64 *d_child_off; double *d_wchild; u64 *d_child_state;
  u64 *d_keys, *d_keys_alt; u32 *d_vals, *d_vals_alt; u32 *d_hist, *d_rowtot;
  u64 *d_flags, *d_pos, *d_flags2, *d_pos2; u64 *d_scan_state; u32 *d_scan_ticket; long long cap_tiles;   // 3 look-back scans per step
  u64 *d_fstate; u32 *d_fticket; long long cap_ftiles;          // k_anneal: two look-backs over 256-slot tiles
  // projector (full CSR, rows in the reference's accumulation order)
  long long n_imp, prj_nnz; int *d_prj_ptr, *d_prj_col; double *d_prj_val; int *d_loc_imp, *d_loc_imp_new; double *d_prj_x;
  // C(T)
  long long n_ct; u64 *d_ct_up, *d_ct_dn; double *d_ct_num, *d_ct_den; u64 *d_ct_hkey; u32 *d_ct_hidx; u64 ct_mask;
  int rng_mode; u64 seed64; u64 step_no;
  DevScalars *d_sc; DevScalars *h_sc;   // h_sc pinned
  HostMail *h_mail, *d_mail; u64 mail_seq, cnt_seq;      // the same pinned words seen from host and device
  bool timers_pending;
  double *d_partials; int n_partial_blocks; double *d_wabs_part; u32 *d_done;
  int key_bits; int pack; u64 invalid_key; u64 *d_binom;
  // multi-rank sharding (owner = hash(det) mod shard_n)
  int shard_rank, shard_n; int *d_grow; long long n_imp_local; long long shard_n0, shard_nch;
  // in-library exchange over RCCL (sqmc_gpu_comm_init): communicator + device staging
  ncclComm_t comm, comm2; double *d_xg; u64 *d_send, *d_recv; long long xch_cap; u32 *d_cnt_mine, *d_cnt_all; u32 *h_cnt_all, *d_cnt_mail;
  u64 cntall_seq;      // comm2: second communicator (ncclCommSplit) for the all-reduce that runs on the side stream
  // timing
  int timing; hipEvent_t ev0[NTIMERS], ev1[NTIMERS]; const char *tname[NTIMERS]; int nt; float tms[NTIMERS];
  double tsum[NTIMERS]; long long tsteps;         // accumulated over the steps since sqmc_gpu_set_timing
  hipStream_t st2; hipEvent_t e_fork, e_join, e_cnt;    // second stream: death + deterministic projection beside spawn + sort
  // pipelined head (sqmc_gpu_run, COUNTER discipline, target population reached): gate + scan + spawn of step n+1 are
  // enqueued right behind k_finish of step n, before the host has read step n's sums
  bool pipeline_next, head_ready; StepP head_p; u64 head_cseq; hipEvent_t hev[4];
  bool owner_ready;           // this step's k_spawn already wrote the owner key of every child (sharded steps)
  bool residents_sorted;      // the walker arrays are known to be in (up, dn) order: true after every finished step, false after an upload
};

// ===================================================================== step kernels

// Everything k_finish does, as arguments: in the pipelined head the first block of the NEXT step's gate
// kernel does it (one launch less on the critical path).
struct FinArgs {
  const double *partials; int nblocks; const double *wabs_part; int nwabs; int mode; u64 *scan_state; u32 *scan_ticket; int n_scan_words;
  HostMail *mail; u64 seq; u64 *fstate; u32 *fticket; long long cap_ftiles; int n_ftiles; int on;
};
__device__ void finish_all(const FinArgs &f, DevScalars *sc);
// gate + child count (COUNTER discipline).  do_walk.f90:3577-3589
__global__ void __launch_bounds__(TPB) k_gate(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, const double *__restrict__ wt,
                                              u64 *__restrict__ nchild, double *__restrict__ wchild, u64 *__restrict__ keys, u32 *__restrict__ vals,
                                              long long n_arg, StepP p, u64 seed, u64 step, DevScalars *sc, int pack, int n_on_device, FinArgs fin) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  const long long n = n_on_device ? (long long)sc->nwalk : n_arg;      // pipelined head: the grid covers an upper bound
  if (fin.on && blockIdx.x == 0) finish_all(fin, sc);                  // the last step's final sums and mail, before this step clears the scalars
  if (i == 0) { sc->n_invalid = 0; sc->tot1 = 0; sc->tot2 = 0; sc->err = 0; }   // every writer of these runs after this kernel
  if (i >= n) return;
  put_key(keys, vals, i, det_key(dev, up[i], dn[i]), pack);      // sort key of the walker itself
  double w = wt[i]; bool spawn, use_wt;
  if (fabs(w) < p.cutoff) {
    Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 0, (u64)i);
    spawn = rng_draw(g) < fabs(w / p.cutoff); use_wt = false;
  } else { spawn = true; use_wt = true; }
  long long nc = 0; double wc = 0.0;
  if (spawn) {
    if (use_wt) { nc = llround(fabs(w)); if (nc < 1) nc = 1; wc = w / (double)nc; }
    else { nc = 1; wc = copysign(p.cutoff, w); }
  }
  nchild[i] = (u64)nc; wchild[i] = wc;
}

// REPLAY discipline: one lane walks the walkers in order, consuming the single rannyu
// stream exactly as the reference does, and records where every child starts in it.
__global__ void __launch_bounds__(64) k_replay_prepass(const ChemTab *__restrict__ gtab, const u64 *__restrict__ up, const u64 *__restrict__ dn,
                                                       const double *__restrict__ wt, u64 *__restrict__ nchild, double *__restrict__ wchild,
                                                       u64 *__restrict__ child_off, u64 *__restrict__ child_state, long long n,
                                                       long long cap_children, StepP p, DevScalars *sc) {
  __shared__ ChemTab t;
  stage_tab(&t, gtab, tab_words_used(gtab->c2_stride));
  if (threadIdx.x != 0) return;
  sc->n_invalid = 0; sc->tot1 = 0; sc->tot2 = 0; sc->err = 0;
  Rng g; g.mode = 0; g.x = sc->lcg;
  u64 c = 0;
  for (long long i = 0; i < n; i++) {
    double w = wt[i]; bool spawn, use_wt;
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
      u64 ju, jd; double pr;
      propose_any(t, g, iu, id, ju, jd, pr);
      c++;
    }
  }
  child_off[n] = c;
  sc->n_children = c; sc->lcg = g.x;
}

// diagonal death/clone, do_walk.f90:3743-3793
__global__ void __launch_bounds__(TPB) k_diag(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, double *__restrict__ wt,
                                              const u32 *__restrict__ flg, double *__restrict__ me, long long n, StepP p, DevScalars *sc) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  if (p.semi && flg_impd(flg[i]) < 1) return;
  double hii = me[i];
  if (hii > 1e50) { hii = h_any(t, dev.integrals, up[i], dn[i], up[i], dn[i]); me[i] = hii; }
  double f = 1.0 + p.tau * (p.e_trial - hii);
  if (f < 0) { if (p.reached > 1) sc->err = SQMC_ERR_NEG_DIAG; f = 0; }
  wt[i] = wt[i] * f;
}

// rank that owns a determinant (get_det_owner, mpi_routines.f90:419-445: any hash of the determinant mod the number of ranks)
__host__ __device__ __forceinline__ int det_owner(u64 key, int nranks) { return (int)((sq_mix64(key ^ 0xA5A5A5A5A5A5A5A5ull) >> 17) % (u64)nranks); }
// sharded steps: k_spawn also notes the destination rank of every child (nranks for a child that made no walker), the key of the bucketing pass
struct OwnerOut { u64 *okey; u32 *oval; int nranks; };      // okey == nullptr: off
// a spawned walker (or the "no walker" marker) into slot n0 + c.  do_walk.f90:3700-3731
__device__ __forceinline__ void spawn_emit(const ChemDev &dev, const WalkArr &w, u64 *__restrict__ keys, u32 *__restrict__ vals, long long n0, long long c,
                                           u32 pf, u64 ju, u64 jd, double wj, const StepP &p, u64 invalid_key, int pack, const OwnerOut &oo) {
  const long long k = n0 + c;
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
    const u64 key = det_key(dev, ju, jd);
    put_key(keys, vals, k, key, pack);
    if (oo.okey) { oo.okey[c] = (u64)det_owner(key, oo.nranks); oo.oval[c] = (u32)c; }
  } else {
    w.sp[c].wt = 0.0; put_key(keys, vals, k, invalid_key, pack);     // sorts behind every real determinant
    if (oo.okey) { oo.okey[c] = (u64)oo.nranks; oo.oval[c] = (u32)c; }
  }
}

#ifdef SPAWN_PROF
__device__ unsigned long long g_prof[8 * 8192];
#define PROF(K) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (K)] = wall_clock64(); } while (0)
extern "C" int sqmc_gpu_debug_prof(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)); }
#else
#define PROF(K)
#endif
// one thread per child proposal; parent found by binary search in the child offsets
__global__ void __launch_bounds__(TPB) k_spawn(ChemDev dev, WalkArr w, const u64 *__restrict__ child_off, const double *__restrict__ wchild,
                                               const u64 *__restrict__ child_state, u64 *__restrict__ keys, u32 *__restrict__ vals,
                                               long long n0_arg, long long cap_all, StepP p, int mode, u64 seed, u64 step, u64 invalid_key, const DevScalars *sc,
                                               HostMail *mail, u64 cnt_seq, int pack, int n_on_device, OwnerOut oo) {
  const long long n0 = n_on_device ? (long long)sc->nwalk : n0_arg;     // pipelined head: launched before the host learnt the walker count
  // the grid covers the free capacity of the walker arrays; the number of children is read from
  // device memory so that the launch does not wait for the host to learn it
  PROF(0);
  __shared__ ChemTab t;
  __shared__ u64 s_win[SPAWN_WIN];
  // Parent of child c = largest i with child_off[i] <= c.  The 256 children of a block have
  // neighbouring parents, so the block narrows [0,n0) for its first child with 256-way splits
  // (one round trip per level instead of log2(n0) dependent loads), then every thread finishes
  // inside a 1024-entry LDS window (global search only if it runs past it).  The first probe, the
  // table staging and the child count do not depend on each other: they are issued together.
  const long long c0 = (long long)blockIdx.x * TPB;
  long long wlo = 0, whi = n0;                       // invariant: child_off[wlo] <= c0, answer for c0 in [wlo, whi)
  u64 pv = 0;
  if (whi - wlo > SPAWN_WIN) {
    const long long stepw = (whi - wlo + TPB - 1) / TPB, probe = wlo + (long long)threadIdx.x * stepw;
    pv = (probe < whi) ? child_off[probe] : ~0ull;
  }
  stage_tab(&t, dev.tab, dev.tab_words);
  const long long nchildren = (long long)sc->n_children;
  if (mail && blockIdx.x == 0 && threadIdx.x == 0) {      // the host sizes the sort from this while the kernel runs
    mail->n_children = (u64)nchildren; __threadfence_system(); mail->cnt_seq = cnt_seq;
  }
  if (c0 >= nchildren || n0 + nchildren > cap_all) return;
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
    const int level = propose_any(t, g, iu, id, ju, jd, prob);
    PROF(4);
    double wj = 0.0;
    if (level > 0) {
      wj = proposal_weight(t, dev.integrals, p.tau, iu, id, ju, jd, level, prob);
      wj = wch * wj;
    }
    spawn_emit(dev, w, keys, vals, n0, c, pflg, ju, jd, wj, p, invalid_key, pack, oo);
  }
  PROF(5);
}

__global__ void __launch_bounds__(TPB) k_main_keys(ChemDev dev, const u64 *__restrict__ up, const u64 *__restrict__ dn, u64 *__restrict__ keys,
                                                   u32 *__restrict__ vals, long long n, int pack) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) put_key(keys, vals, i, det_key(dev, up[i], dn[i]), pack);
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
  // check_initiator
  {
    const int dd = d - p.imind > 0 ? d - p.imind : 0;
    const double thr = p.r_init * ipow_d(dd, p.ipow), aw = fabs(wt);
    if (ini == 3 && p.r_init >= 0) { if (wt * ps < 1.0) wt = (double)ps; }
    else if (ini == 2 && ((aw <= thr && d > 0) || ((aw <= p.r_init && !p.cti) && d == -2))) ini = 1;
    else if (ini < 2 && ((aw > thr && d >= 0) || ((aw > p.r_init || p.cti) && d == -2))) ini = ini + 1;
  }
  int dtest = d;
  if (d == -1) { if (jj >= n || get_key(skey, jj, pack) == invalid_key) dtest = 1; d = 1; }   // 6032-6036 then the last-det test at 6038
  const bool discard = (((wt == 0.0 && (ini != 3 || p.r_init < 0)) || ini == 0) && dtest >= 1);
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
  const MergedRec r = fold_slot(in, s_w, s_f, threadIdx.x, TPB, w, skey, perm, j, n0, n_all, p, invalid_key, pack);
  store_wabs(wabs_part, blockIdx.x, wabs, cnt);
  if (j >= n_all) return;
  flags[j] = r.f;
  if (!(r.f & 1ull)) return;                   // not the head of a run, or discarded: nothing to store
  m.up[j] = r.up; m.dn[j] = r.dn; m.wt[j] = r.wt; m.flg[j] = r.flg;
  m.me[j] = r.me; m.en[j] = r.en; m.ed[j] = r.ed;
}

// stochastic rounding of small weights (reduce_my_walker, do_walk.f90:7196-7254); RNG draws
// are taken in merged-walker order: REPLAY = skip-ahead of the rannyu LCG by the rank of the
// draw, COUNTER = stream keyed by the merged index.
__global__ void __launch_bounds__(TPB) k_round(WalkArr m, const u64 *__restrict__ flags, const u64 *__restrict__ pos, u64 *__restrict__ flags2,
                                               long long n_all, StepP p, int mode, u64 seed, u64 step, const DevScalars *sc) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j >= n_all) return;
  const u64 f = flags[j];
  if (!(f & 1ull)) { flags2[j] = 0; return; }
  double wt = m.wt[j];
  if (f >> 32) {
    const u64 ps = pos[j];
    double r;
    if (mode == 0) r = (double)lcg_skip(sc->lcg, (ps >> 32) + 1) * 3.552713678800500929355621337890625e-15;
    else { Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, m.up[j] * SQ_GOLDEN + m.dn[j]); r = rng_draw(g); }   // keyed by the determinant
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
// rannyu stream in join order, COUNTER = stream keyed by the merged index of the later walker.
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

// C(T) lookup: open-addressed hash (linear probing, load <= 1/2) from the determinant's sort
// key to its row in the C(T) arrays.  Replaces the binary search of
// binary_search_list_and_update (more_tools.f90:4041-4098): one or two dependent reads
// instead of log2(n_ct) ~ 17; the 2 MB table is L2-resident.
#define CT_EMPTY (~0ull)
__device__ __forceinline__ u64 ct_hash(u64 k) { return sq_mix64(k); }
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
__device__ __forceinline__ long long ct_lookup(const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 mask, u64 key) {
  u64 h = ct_hash(key) & mask;
  while (true) {
    const u64 k = hkey[h];
    if (k == key) return (long long)hidx[h];
    if (k == CT_EMPTY) return -1;
    h = (h + 1) & mask;
  }
}

#define NSTAT 13
__device__ void finish_step(const double *__restrict__ partials, int nblocks, const double *__restrict__ wabs_part, int nwabs,
                            int mode, DevScalars *sc, u64 *__restrict__ scan_state, u32 *__restrict__ scan_ticket, int n_scan_words);
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
template <int ITEMS>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) k_anneal(WalkArr w, WalkArr o, const u64 *__restrict__ skey, const u32 *__restrict__ perm, int *__restrict__ loc_imp,
                                                const u64 *__restrict__ hkey, const u32 *__restrict__ hidx, u64 hmask,
                                                const double *__restrict__ cnum, const double *__restrict__ cden,
                                                double *__restrict__ partials, double *__restrict__ wabs_part, long long n0, long long n_all, StepP p,
                                                u64 invalid_key, int pack, int mode, u64 seed, u64 step, DevScalars *sc,
                                                u64 *__restrict__ state1, u64 *__restrict__ state2, u32 *__restrict__ ticket) {
  constexpr int TILE = TPB * ITEMS;
  __shared__ u32 s_tile; __shared__ u64 s_ex[2]; __shared__ u64 s_wsum[2][TPB / 64];
  __shared__ double s_w[TILE]; __shared__ u32 s_f[TILE];          // weight and flags of every slot of the tile
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  const u32 tile = s_tile;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // every wavefront owns 64*ITEMS consecutive slots and takes them row by row (lane l: slot r*64 + l)
  const long long base = (long long)tile * TILE + (long long)wv * (64 * ITEMS) + lane;
  const bool last_tile = (long long)(tile + 1) * TILE >= n_all;
  APROF(0);
  u64 key[ITEMS]; double wabs = 0.0, cnt = 0.0;
  MergedRec r[ITEMS];
  {
    SlotIn in[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) { in[k] = load_slot(w, skey, perm, base + (long long)k * 64, n0, n_all, invalid_key, pack, wabs, cnt); key[k] = in[k].key; }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) stage_slot(in[k], s_w, s_f, wv * (64 * ITEMS) + k * 64 + lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) r[k] = fold_slot(in[k], s_w, s_f, wv * (64 * ITEMS) + k * 64 + lane, TILE, w, skey, perm, base + (long long)k * 64, n0, n_all, p, invalid_key, pack);
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
        else { Rng g; g.mode = 1; g.x = sq_counter_key(seed, step, 2, r[k].up * SQ_GOLDEN + r[k].dn); rr = rng_draw(g); }
        if (rr < (fabs(r[k].wt) / p.min_wt)) r[k].wt = copysign(p.min_wt, r[k].wt); else r[k].wt = 0.0;
      }
      // reduce_my_walker drops zero weights outside the deterministic space (7222-7249)
      const bool drop = p.semi ? (r[k].wt == 0.0 && r[k].d >= 1) : (r[k].wt == 0.0);
      if (!drop) { f2[k] = 1ull; if (r[k].d == 0) f2[k] |= (1ull << 32); }
    }
  }
  // ---- final position (lo) and rank among the deterministic-space walkers (hi)
  carry = 0;
#pragma unroll
  for (int k = 0; k < ITEMS; k++) { const u64 x = wave_incl_scan_u64(f2[k], lane); inc[k] = x + carry; carry += __shfl(x, 63, 64); }
  if (lane == 0) s_wsum[1][wv] = carry;
  __syncthreads();
  ex = 0; tot = 0;
#pragma unroll
  for (int q = 0; q < TPB / 64; q++) { if (q < wv) ex += s_wsum[1][q]; tot += s_wsum[1][q]; }
  if (threadIdx.x < 64) {
    const u64 e = lookback_exclusive(state2, tile, tot, threadIdx.x);
    if (threadIdx.x == 0) { s_ex[1] = e; if (last_tile) { sc->tot2 = e + tot; sc->nwalk = (e + tot) & 0xFFFFFFFFull; } }
  }
  __syncthreads();
  ex += s_ex[1];
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
    const double wt = r[k].wt * p.rfi;
    const int d = r[k].d, ini = flg_init(r[k].flg), psg = flg_psign(r[k].flg);
    double en = r[k].en, ed = r[k].ed;
    if (en > 1e50) {
      const long long q = ct_lookup(hkey, hidx, hmask, key[k]);
      if (q < 0) { en = 0.0; ed = 0.0; } else { en = cnum[q]; ed = cden[q]; }
    }
    o.up[q0] = r[k].up; o.dn[q0] = r[k].dn; o.wt[q0] = wt; o.flg[q0] = r[k].flg;
    o.me[q0] = r[k].me; o.en[q0] = en; o.ed[q0] = ed;
    if (d == 0 && p.semi && (long long)(ex2 >> 32) < p.nimp_cap) loc_imp[ex2 >> 32] = (int)q0;
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
  APROF(5);
}
// posts the (all-reduced) scalars of a sharded step to the host mailbox
__global__ void k_post_mail(const DevScalars *sc, HostMail *mail, u64 seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 16; i++) mail->stats[i] = sc->stats[i];
  mail->tot2 = sc->tot2; mail->err = sc->err;
  __threadfence_system();
  mail->seq = seq;
}
// The final reduction stays a kernel of its own: folding it into the last-arriving block of
// k_compact needs an agent-scope release in every block and cost more than this launch.
__global__ void __launch_bounds__(TPB) k_finish(FinArgs f, DevScalars *sc) { finish_all(f, sc); }
__device__ void finish_all(const FinArgs &f, DevScalars *sc) {
  if (f.on == 3) {            // sharded step: the sums were finished and all-reduced by kernels before this one; only the mail is left
    if (threadIdx.x == 0) {
      for (int i = 0; i < 16; i++) f.mail->stats[i] = sc->stats[i];
      f.mail->tot2 = sc->tot2; f.mail->err = sc->err;
      __threadfence_system();
      f.mail->seq = f.seq;
    }
    __syncthreads();
    return;
  }
  // the look-back words k_anneal used this step (two arrays, n_ftiles words each) are zero again for the next one
  for (int i = threadIdx.x; i < f.n_ftiles; i += TPB) { f.fstate[i] = 0; f.fstate[f.cap_ftiles + i] = 0; }
  if (threadIdx.x == 0 && f.n_ftiles > 0) *f.fticket = 0;
  finish_step(f.partials, f.nblocks, f.wabs_part, f.nwabs, f.mode, sc, f.scan_state, f.scan_ticket, f.n_scan_words);
  if (f.mail && threadIdx.x == 0) {
    for (int i = 0; i < 16; i++) f.mail->stats[i] = sc->stats[i];
    f.mail->tot2 = sc->tot2; f.mail->err = sc->err;
    __threadfence_system();
    f.mail->seq = f.seq;
  }
  __syncthreads();
}

// final reduction: sums block partials
// (fixed strided order + fixed tree: reproducible run to run), publishes the step's sums,
// advances the REPLAY stream and re-zeroes the look-back scan states for the next step
__device__ void finish_step(const double *__restrict__ partials, int nblocks, const double *__restrict__ wabs_part, int nwabs,
                            int mode, DevScalars *sc, u64 *__restrict__ scan_state, u32 *__restrict__ scan_ticket, int n_scan_words) {
  __shared__ double red2[TPB / 64][NSTAT + 2];
  __shared__ double tot[NSTAT + 2];
  for (int i = threadIdx.x; i < n_scan_words; i += TPB) scan_state[i] = 0;
  if (threadIdx.x < 3) scan_ticket[threadIdx.x] = 0;
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
    double *o = sc->stats;
    o[0] = tot[0]; o[1] = tot[1]; o[2] = tot[2]; o[3] = tot[3]; o[4] = tot[4];
    o[5] = (double)(sc->tot2 & 0xFFFFFFFFull); o[6] = tot[6];
    sc->nwalk = sc->tot2 & 0xFFFFFFFFull;
    o[7] = tot[NSTAT + 1]; o[8] = tot[8]; o[9] = tot[9]; o[10] = tot[10];
    o[11] = tot[11]; o[12] = tot[12]; o[13] = tot[5]; o[14] = tot[NSTAT]; o[15] = (double)sc->n_children;
    if (mode == 0) sc->lcg = lcg_skip(sc->lcg, sc->tot1 >> 32);
  }
}
// ============================================================ batch / test door kernels
__global__ void __launch_bounds__(TPB) k_ham_batch(ChemDev dev, const u64 *iu, const u64 *id, const u64 *ju, const u64 *jd, double *h, long long n) {
  __shared__ ChemTab t;
  stage_tab(&t, dev.tab, dev.tab_words);
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) h[i] = h_any(t, dev.integrals, iu[i], id[i], ju[i], jd[i]);
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

#define HEG_LUT_MAX 729            // (2*4+1)^3: plane-wave indices up to +-4 per direction
// ================================================================ HCI connections
// find_important_connected_dets_chem, chemistry.f90:6819-7159: one thread per reference
// determinant; pass 0 counts, pass 1 writes at the scanned offsets.  Emits (up, dn,
// H_ij*c_j, e_mix_den) with the reference determinant itself in slot 0.
__global__ void __launch_bounds__(TPB) k_hci_gen(ChemDev dev, const u64 *__restrict__ rup, const u64 *__restrict__ rdn, const double *__restrict__ coef,
                                                 double eps_var, int diag_mode, long long n_ref, int pass, u64 *__restrict__ counts,
                                                 const u64 *__restrict__ offs, u64 *__restrict__ ou, u64 *__restrict__ od,
                                                 double *__restrict__ onum, double *__restrict__ oden, u64 key_lo, u64 key_hi) {
  __shared__ ChemTab t;
  __shared__ unsigned char s_lut[HEG_LUT_MAX];        // plane wave (kx,ky,kz) -> orbital id, 0 = not in the basis (find_orb_id, heg.f90:752-771)
  stage_tab(&t, dev.tab, dev.tab_words);
  if (t.sys_type == 1) {
    const int W = 2 * t.heg_nmax + 1;
    for (int k = threadIdx.x; k < W * W * W; k += TPB) s_lut[k] = 0;
    __syncthreads();
    for (int o = 1 + threadIdx.x; o <= t.norb; o += TPB)
      s_lut[((t.krel[o][0] + t.heg_nmax) * W + (t.krel[o][1] + t.heg_nmax)) * W + (t.krel[o][2] + t.heg_nmax)] = (unsigned char)o;
    __syncthreads();
  }
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n_ref) return;
  const double c = coef[i];
  if (c == 0.0) { if (!pass) counts[i] = 0; return; }
  const double eps = eps_var / fabs(c);
  const u64 up = rup[i], dn = rdn[i];
  const int n = t.norb;
  const double sqrt2 = sqrt(2.0), sqrt2inv = 1.0 / sqrt2;
  u64 cnt = 0; const u64 base = pass ? offs[i] : 0;
  // diag_mode 2 ("raw", for the semistochastic PT): e_mix_den carries the index of the reference determinant instead
  // [key_lo, key_hi): only connections whose determinant key falls in this slice are kept -- the PT stage
  // of a large space is done in slices of the connected space, each with exact sums (the role of
  // n_energy_batch, hci.f90:642); the full range keeps everything without computing keys
  const bool sliced = !(key_lo == 0 && key_hi == ~0ull);
#define EMIT(U, D, M, DEN) do { bool in_ = true; if (sliced) { const u64 kk_ = det_key(dev, (U), (D)); in_ = (kk_ >= key_lo && kk_ < key_hi); } \
    if (in_) { if (pass) { ou[base + cnt] = (U); od[base + cnt] = (D); onum[base + cnt] = (M) * c; oden[base + cnt] = (diag_mode == 2) ? (double)i : (DEN); } cnt++; } } while (0)
  { double hd = (diag_mode == 1) ? h_any(t, dev.integrals, up, dn, up, dn) : 0.0; EMIT(up, dn, hd, c); }
  if (t.sys_type == 1) {
    // find_important_connected_dets_heg, heg.f90:2475-2727: no single excitations (momentum); every
    // double p,q -> r,s with k_p + k_q = k_r + k_s whose |H| exceeds eps/|c|.  The reference walks
    // |H|-sorted translation-invariant lists and stops at absH <= eps (:2608, :2629); here each
    // candidate's element is evaluated and screened -- the same set, 91 pairs x norb holes for 14
    // electrons.  Same-spin pairs take r < s (:2618).
    const int nm = t.heg_nmax, W = 2 * nm + 1;
    for (int cls = 0; cls < 3; cls++) {
      const u64 A = (cls == 1) ? dn : up, B = (cls == 0) ? up : dn;     // strings of the first / second electron
      for (u64 ea = A; ea; ea &= ea - 1) {
        const int pa = ctz64(ea) + 1;
        for (u64 eb = (cls == 2) ? B : (ea & (ea - 1)); eb; eb &= eb - 1) {
          const int qb = ctz64(eb) + 1;
          const int sx = t.krel[pa][0] + t.krel[qb][0], sy = t.krel[pa][1] + t.krel[qb][1], sz = t.krel[pa][2] + t.krel[qb][2];
          for (u64 hr = t.orb_mask & ~A; hr; hr &= hr - 1) {
            const int r = ctz64(hr) + 1;
            const int kx = sx - t.krel[r][0], ky = sy - t.krel[r][1], kz = sz - t.krel[r][2];
            if (kx < -nm || kx > nm || ky < -nm || ky > nm || kz < -nm || kz > nm) continue;
            const int s_ = s_lut[((kx + nm) * W + (ky + nm)) * W + (kz + nm)];
            if (!s_) continue;
            if (cls != 2 && s_ <= r) continue;
            if ((B >> (s_ - 1)) & 1) continue;
            u64 nu = up, nd = dn;
            if (cls == 0) nu = (up & ~bit64(pa - 1) & ~bit64(qb - 1)) | bit64(r - 1) | bit64(s_ - 1);
            else if (cls == 1) nd = (dn & ~bit64(pa - 1) & ~bit64(qb - 1)) | bit64(r - 1) | bit64(s_ - 1);
            else { nu = (up & ~bit64(pa - 1)) | bit64(r - 1); nd = (dn & ~bit64(qb - 1)) | bit64(s_ - 1); }
            const double mel = h_heg(t, up, dn, nu, nd);
            if (!(fabs(mel) > eps)) continue;
            EMIT(nu, nd, mel, 0.0);
          }
        }
      }
    }
    if (!pass) counts[i] = cnt;
    return;
  }
  // singles
  for (int sp = 0; sp < 2; sp++) {
    const u64 occ = sp ? dn : up;
    for (u64 e = occ; e; e &= e - 1) {
      const int pe = ctz64(e) + 1;
      for (u64 h = t.sym_mask[t.orbsym[pe]] & ~occ; h; h &= h - 1) {
        const int r = ctz64(h) + 1;
        u64 nu = up, nd = dn;
        if (!sp) nu = (up & ~bit64(pe - 1)) | bit64(r - 1); else nd = (dn & ~bit64(pe - 1)) | bit64(r - 1);
        if (t.time_sym) { if (nu == nd && t.z < 0) continue; if (up == nd && dn == nu) continue; }
        double mel = h_single(t, dev.integrals, up, dn, nu, nd);
        if (fabs(mel) < eps) continue;
        if (t.time_sym) {
          if (up == dn && nu != nd) mel = sqrt2inv * mel;
          if (nu == nd && up != dn) mel = sqrt2 * mel;
          if (nu > nd) { u64 x = nu; nu = nd; nd = x; mel = t.z * mel; }
        }
        EMIT(nu, nd, mel, 0.0);
      }
    }
  }
  if (!(eps > dev.max_double)) {
    // occupied pairs: up-up, dn-dn, up-dn (chemistry.f90:7000-7021)
    for (int cls = 0; cls < 3; cls++) {
      const u64 A = (cls == 1) ? dn : up, B = (cls == 0) ? up : dn;
      for (u64 ea = A; ea; ea &= ea - 1) {
        const int pa = ctz64(ea) + 1;
        for (u64 eb = (cls == 2) ? B : (ea & (ea - 1)); eb; eb &= eb - 1) {
          const int qb = ctz64(eb) + 1;
          int p = pa + (cls == 1 ? n : 0), q = qb + (cls == 0 ? 0 : n);
          int p2 = p, q2 = q;
          const bool both_dn = (cls == 1), swapped = (cls == 2 && p > q - n);
          if (both_dn) { p2 = p - n; q2 = q - n; }
          if (swapped) { p2 = q - n; q2 = p + n; }
          const long long e = (p2 > q2) ? ((long long)p2 * (p2 - 1)) / 2 + q2 : ((long long)q2 * (q2 - 1)) / 2 + p2;
          const long long k0 = dev.pq_ind[e] - 1; const int kc = dev.pq_count[e];
          for (int hh = 0; hh < kc; hh++) {
            if (dev.hb_absH[k0 + hh] <= eps) break;
            int r = dev.hb_r[k0 + hh], s = dev.hb_s[k0 + hh];
            if (both_dn) { r += n; s += n; }
            if (swapped) { int rt = s - n; s = r + n; r = rt; }
            if (r <= n ? ((up >> (r - 1)) & 1) : ((dn >> (r - n - 1)) & 1)) continue;
            if (s <= n ? ((up >> (s - 1)) & 1) : ((dn >> (s - n - 1)) & 1)) continue;
            u64 nu = up, nd = dn;
            if (p <= n) nu &= ~bit64(p - 1); else nd &= ~bit64(p - n - 1);
            if (q <= n) nu &= ~bit64(q - 1); else nd &= ~bit64(q - n - 1);
            if (r <= n) nu |= bit64(r - 1); else nd |= bit64(r - n - 1);
            if (s <= n) nu |= bit64(s - 1); else nd |= bit64(s - n - 1);
            if (t.time_sym) { if (nu == nd && t.z < 0) continue; if (up == nd && dn == nu) continue; }
            double mel = 0.0;
            if (pass) {
              mel = h_double(t, dev.integrals, up, dn, nu, nd);
              if (t.time_sym) {
                if (up == dn && nu != nd) mel = sqrt2inv * mel;
                if (nu == nd && up != dn) mel = sqrt2 * mel;
              }
            }
            if (t.time_sym && nu > nd) { u64 x = nu; nu = nd; nd = x; mel = t.z * mel; }
            EMIT(nu, nd, mel, 0.0);
          }
        }
      }
    }
  }
#undef EMIT
  if (!pass) counts[i] = cnt;
}
// dedup of the sorted connection list: sums e_mix_num / e_mix_den of equal determinants
// left to right (merge_original_with_spawned3, tools.f90:577-660)
__global__ void __launch_bounds__(TPB) k_hci_heads(const u64 *__restrict__ skey, u64 *__restrict__ flags, long long n) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j < n) flags[j] = (j == 0 || skey[j] != skey[j - 1]) ? 1ull : 0ull;
}
__global__ void __launch_bounds__(TPB) k_hci_dedup(const u64 *__restrict__ skey, const u32 *__restrict__ perm, const u64 *__restrict__ flags,
                                                   const u64 *__restrict__ pos, const u64 *__restrict__ iu, const u64 *__restrict__ id,
                                                   const double *__restrict__ inum, const double *__restrict__ iden,
                                                   u64 *__restrict__ ou, u64 *__restrict__ od, double *__restrict__ onum, double *__restrict__ oden, long long n) {
  long long j = (long long)blockIdx.x * TPB + threadIdx.x;
  if (j >= n || !flags[j]) return;
  const u64 key = skey[j]; u32 t = perm[j];
  double a = inum[t], b = iden[t];
  for (long long jj = j + 1; jj < n && skey[jj] == key; jj++) { a = a + inum[perm[jj]]; b = b + iden[perm[jj]]; }
  const u64 o = pos[j];
  ou[o] = iu[t]; od[o] = id[t]; onum[o] = a; oden[o] = b;
}


// ============================================================================ SpMV
// Symmetric matrix kept as FULL CSR (int32 columns) so that every row is owned by one
// wavefront and no atomics are needed: 12 B per stored entry + 8 B gathered x.
struct sqmc_spmv_plan { long long n, nnz_full; int *d_ptr, *d_col; double *d_val, *d_x, *d_y; hipStream_t st; };
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
__global__ void __launch_bounds__(64 * SPMV_ROWS_PER_BLOCK) k_spmv_wave(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                                        const double *__restrict__ x, double *__restrict__ y, long long n) {
  const long long row = (long long)blockIdx.x * SPMV_ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const int b = ptr[row], e = ptr[row + 1];
  double s = 0.0;
  for (int k = b + lane; k < e; k += 64) s += val[k] * x[col[k]];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) y[row] = s;
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

// ================================================================================ ABI
extern "C" {

const char *sqmc_gpu_last_error(void) { return g_err.c_str(); }
int sqmc_gpu_set_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SQMC_ERR_HIP, "no HIP device: libsqmc_gpu has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(SQMC_ERR_BAD_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  return SQMC_OK;
}
void sqmc_gpu_free(void *p) { free(p); }

// everything that does not depend on the kind of system: device copy of the tables, sort-key
// width, RNG, walker arrays and work buffers
static int init_common(sqmc_gpu_ctx *c, int norb, int nup, int ndn, int rng_mode, const int32_t seed_in[4], long long mwalk_in, sqmc_gpu_ctx **out) {
  const ChemTab &t = c->htab;
  struct { int norb, nup, ndn, rng_mode; int32_t irand_seed[4]; long long mwalk; } cfgv = {norb, nup, ndn, rng_mode, {seed_in[0], seed_in[1], seed_in[2], seed_in[3]}, mwalk_in};
  auto *cfg = &cfgv;
  (void)t;
  HIPCHK(hipMalloc(&c->d_tab, sizeof(ChemTab)));
  HIPCHK(hipMemcpy(c->d_tab, &c->htab, sizeof(ChemTab), hipMemcpyHostToDevice));
  c->dev.tab = c->d_tab; c->dev.tab_words = tab_words_used(c->htab.c2_stride); c->dev.integrals = c->d_ints; c->dev.max_double = 0.0;
  {   // binomial table + width of the colex sort key
    std::vector<u64> bn(64 * SQ_BINOM_STRIDE, 0);
    for (int a = 0; a < 64; a++) { bn[a * SQ_BINOM_STRIDE] = 1; for (int b = 1; b <= 32 && b <= a; b++) bn[a * SQ_BINOM_STRIDE + b] = (b == a) ? 1 : bn[(a - 1) * SQ_BINOM_STRIDE + b - 1] + bn[(a - 1) * SQ_BINOM_STRIDE + b]; }
    auto choose = [&](int n_, int k_) -> long double { long double r = 1; for (int q = 1; q <= k_; q++) r = r * (n_ - k_ + q) / q; return r; };
    long double tot = choose(cfg->norb, cfg->nup) * choose(cfg->norb, cfg->ndn);
    if (tot >= 9.0e18L) { delete c; return fail(SQMC_ERR_UNSUPPORTED, "determinant space needs more than 63 key bits"); }
    u64 nd = (u64)(choose(cfg->norb, cfg->ndn) + 0.5L), total = (u64)(tot + 0.5L);
    int bits = 1; while (bits < 63 && ((1ull << bits) - 1ull) < total) bits++;
    c->key_bits = bits; c->invalid_key = (1ull << bits) - 1ull; c->pack = (bits <= 32) ? 1 : 0;
    HIPCHK(hipMalloc(&c->d_binom, bn.size() * 8));
    HIPCHK(hipMemcpy(c->d_binom, bn.data(), bn.size() * 8, hipMemcpyHostToDevice));
    c->dev.binom = c->d_binom; c->dev.n_dn_strings = nd;
  }
  c->rng_mode = cfg->rng_mode;
  // limbs of the input seed may exceed 12 bits ('(4i4,x,4i4)' reads 4 decimal digits each):
  // rannyu's limb products treat them as coefficients of powers of 2^12, so the state is the SUM
  u64 s48 = (((u64)cfg->irand_seed[0] << 36) + ((u64)cfg->irand_seed[1] << 24) + ((u64)cfg->irand_seed[2] << 12) + (u64)(2 * (cfg->irand_seed[3] / 2) + 1)) & SQ_MASK48;
  c->seed64 = s48; c->step_no = 0;
  c->mwalk = cfg->mwalk > 0 ? cfg->mwalk : 0;
  HIPCHK(hipMalloc(&c->d_sc, sizeof(DevScalars)));
  HIPCHK(hipMemset(c->d_sc, 0, sizeof(DevScalars)));
  HIPCHK(hipHostMalloc(&c->h_sc, sizeof(DevScalars)));
  memset(c->h_sc, 0, sizeof(DevScalars));
  HIPCHK(hipHostMalloc(&c->h_mail, sizeof(HostMail), hipHostMallocMapped));
  memset((void *)c->h_mail, 0, sizeof(HostMail));
  HIPCHK(hipHostGetDevicePointer((void **)&c->d_mail, (void *)c->h_mail, 0));
  c->h_sc->lcg = s48;
  HIPCHK(hipMemcpy(&c->d_sc->lcg, &s48, 8, hipMemcpyHostToDevice));
  if (c->mwalk > 0) {
    const long long M = c->mwalk;
    if (M >= (1ll << 30)) { delete c; return fail(SQMC_ERR_UNSUPPORTED, "MWALK must be < 2^30"); }
    if (alloc_walk(c->w, M, true) || alloc_walk(c->m, M, false)) return SQMC_ERR_HIP;
    HIPCHK(hipMalloc(&c->d_nchild, (M + 1) * 8)); HIPCHK(hipMalloc(&c->d_child_off, (M + 1) * 8));
    HIPCHK(hipMalloc(&c->d_wchild, M * 8)); HIPCHK(hipMalloc(&c->d_child_state, M * 8));
    HIPCHK(hipMalloc(&c->d_keys, M * 8)); HIPCHK(hipMalloc(&c->d_keys_alt, M * 8));
    HIPCHK(hipMalloc(&c->d_vals, M * 4)); HIPCHK(hipMalloc(&c->d_vals_alt, M * 4));
    long long ntiles = (M + RS_TILE - 1) / RS_TILE;
    HIPCHK(hipMalloc(&c->d_hist, ntiles * RS_MAX_RADIX * 4)); HIPCHK(hipMalloc(&c->d_rowtot, RS_MAX_RADIX * 4));
    HIPCHK(hipMalloc(&c->d_flags, M * 8)); HIPCHK(hipMalloc(&c->d_pos, M * 8));
    HIPCHK(hipMalloc(&c->d_flags2, M * 8)); HIPCHK(hipMalloc(&c->d_pos2, M * 8));
    c->cap_tiles = (M + SCAN_TILE - 1) / SCAN_TILE + 1;
    HIPCHK(hipMalloc(&c->d_scan_state, 3 * c->cap_tiles * 8)); HIPCHK(hipMalloc(&c->d_scan_ticket, 3 * 4));
    HIPCHK(hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8)); HIPCHK(hipMemset(c->d_scan_ticket, 0, 3 * 4));
    c->cap_ftiles = nblk(M) + 1;
    HIPCHK(hipMalloc(&c->d_fstate, 2 * c->cap_ftiles * 8)); HIPCHK(hipMalloc(&c->d_fticket, 4));
    HIPCHK(hipMemset(c->d_fstate, 0, 2 * c->cap_ftiles * 8)); HIPCHK(hipMemset(c->d_fticket, 0, 4));
    c->n_partial_blocks = nblk(M);
    HIPCHK(hipMalloc(&c->d_partials, ((long long)c->n_partial_blocks * NSTAT + 128) * 8));
    HIPCHK(hipMalloc(&c->d_wabs_part, ((long long)c->n_partial_blocks * 2 + 2) * 8));
    HIPCHK(hipMalloc(&c->d_done, 4)); HIPCHK(hipMemset(c->d_done, 0, 4));
  }
  for (int i = 0; i < NTIMERS; i++) { HIPCHK(hipEventCreate(&c->ev0[i])); HIPCHK(hipEventCreate(&c->ev1[i])); }
  HIPCHK(hipStreamCreate(&c->st2));
  HIPCHK(hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->e_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->e_cnt, hipEventDisableTiming));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c->hev[i]));
  *out = c;
  return SQMC_OK;
}

int sqmc_gpu_init_chem(const sqmc_chem_cfg *cfg, sqmc_gpu_ctx **out) {
  if (!cfg || !out) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (cfg->norb < 1 || cfg->norb > SQ_MAXORB) return fail(SQMC_ERR_UNSUPPORTED, "norb must be in 1..64 (one 64-bit word per spin)");
  if (cfg->n_group < 1 || cfg->n_group > SQ_MAXSYM) return fail(SQMC_ERR_UNSUPPORTED, "point group order must be <= 8");
  if (cfg->nup < 0 || cfg->ndn < 0 || cfg->nup > cfg->norb || cfg->ndn > cfg->norb || cfg->n_core_orb < 0 || cfg->n_core_orb > cfg->ndn || cfg->n_core_orb > cfg->nup ||
      cfg->nup + cfg->ndn - 2 * cfg->n_core_orb < 2)
    return fail(SQMC_ERR_BAD_ARG, "nup / ndn / n_core_orb out of range (at least two active electrons: the proposal draws a second one)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SQMC_ERR_HIP, "no HIP device: libsqmc_gpu has no CPU fallback");
  sqmc_gpu_ctx *c = new sqmc_gpu_ctx();
  memset((void *)c, 0, sizeof(*c));
  HIPCHK(hipStreamCreate(&c->st));
  ChemTab &t = c->htab;
  t.norb = cfg->norb; t.nup = cfg->nup; t.ndn = cfg->ndn; t.ncore = cfg->n_core_orb; t.nelec = cfg->nup + cfg->ndn;
  t.time_sym = cfg->time_sym; t.z = cfg->z; t.ngroup = cfg->n_group;
  t.orb_mask = (cfg->norb >= 64) ? ~0ull : ((1ull << cfg->norb) - 1ull);
  for (int i = 1; i <= cfg->n_group; i++) for (int j = 1; j <= cfg->n_group; j++) t.prod[i][j] = (unsigned char)cfg->product_table[i * 9 + j];
  for (int i = 1; i <= cfg->norb; i++) {
    int s = cfg->orbital_symmetries[i];
    if (s < 1 || s > cfg->n_group) { delete c; return fail(SQMC_ERR_BAD_ARG, "orbital symmetry out of range"); }
    t.orbsym[i] = (unsigned char)s; t.sym_mask[s] |= 1ull << (i - 1);
  }
  const int n2 = cfg->norb + 2;
  t.c2_stride = n2;
  for (int i = 1; i <= cfg->norb + 1; i++) for (int j = 1; j <= cfg->norb + 1; j++) t.c2[i * n2 + j] = (unsigned short)cfg->combine_2[i * n2 + j];
  {
    int a = t.c2[(cfg->norb + 1) * n2 + cfg->norb + 1]; long long ix = ((long long)a * (a - 1)) / 2 + a;
    if (ix > cfg->n_integrals) { delete c; return fail(SQMC_ERR_BAD_ARG, "integral table shorter than integral_index(norb+1,...)"); }
    t.nuclear = cfg->integrals[ix];
  }
  HIPCHK(hipMalloc(&c->d_ints, (cfg->n_integrals + 1) * sizeof(double)));
  HIPCHK(hipMemcpy(c->d_ints, cfg->integrals, (cfg->n_integrals + 1) * sizeof(double), hipMemcpyHostToDevice));
  return init_common(c, cfg->norb, cfg->nup, cfg->ndn, cfg->rng_mode, cfg->irand_seed, cfg->mwalk, out);
}

int sqmc_gpu_init_heg(const sqmc_heg_cfg *cfg, sqmc_gpu_ctx **out) {
  if (!cfg || !out || !cfg->k_vectors) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (cfg->norb < 1 || cfg->norb > SQ_MAXORB) return fail(SQMC_ERR_UNSUPPORTED, "norb must be in 1..64 (one 64-bit word per spin)");
  if (cfg->n_dim != 2 && cfg->n_dim != 3) return fail(SQMC_ERR_BAD_ARG, "n_dim must be 2 or 3");
  if (cfg->nup < 0 || cfg->ndn < 0 || cfg->nup > cfg->norb || cfg->ndn > cfg->norb || cfg->nup + cfg->ndn < 2)
    return fail(SQMC_ERR_BAD_ARG, "the electron gas needs at least two electrons (off_diagonal_move_heg draws a pair) and at most norb per spin");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SQMC_ERR_HIP, "no HIP device: libsqmc_gpu has no CPU fallback");
  sqmc_gpu_ctx *c = new sqmc_gpu_ctx();
  memset((void *)c, 0, sizeof(*c));
  HIPCHK(hipStreamCreate(&c->st));
  ChemTab &t = c->htab;
  t.sys_type = 1; t.n_dim = cfg->n_dim; t.length_cell = cfg->length_cell;
  t.norb = cfg->norb; t.nup = cfg->nup; t.ndn = cfg->ndn; t.ncore = 0; t.nelec = cfg->nup + cfg->ndn; t.time_sym = 0; t.z = 1; t.ngroup = 1;
  t.orb_mask = (cfg->norb >= 64) ? ~0ull : ((1ull << cfg->norb) - 1ull);
  t.c2_stride = 0;
  for (int i = 1; i <= cfg->norb; i++) for (int j = 0; j < 3; j++) t.kvec[i][j] = (j < cfg->n_dim) ? cfg->k_vectors[(i - 1) * cfg->n_dim + j] : 0.0;
  t.heg_nmax = 0;
  for (int i = 1; i <= cfg->norb; i++) for (int j = 0; j < 3; j++) {
    const long long kr = llround(t.kvec[i][j] * cfg->length_cell / (2.0 * 3.14159265358979323846264338327950288));
    if (kr < -127 || kr > 127) { delete c; return fail(SQMC_ERR_UNSUPPORTED, "plane-wave index beyond +-127"); }
    t.krel[i][j] = (signed char)kr;
    if (llabs(kr) > t.heg_nmax) t.heg_nmax = (int)llabs(kr);
  }
  return init_common(c, cfg->norb, cfg->nup, cfg->ndn, cfg->rng_mode, cfg->irand_seed, cfg->mwalk, out);
}

// get_nbr, more_tools.f90:223-355: neighbour `type` (0 LEFT, 1 RIGHT, 2 UP, 3 DOWN) of a 1-based site, 0 if not allowed
static int hubbard_nbr(int lx, int ly, int pbc, int site, int type) {
  const int y1 = (site - 1) / lx + 1, x1 = site - (y1 - 1) * lx;
  int x2 = x1, y2 = y1; bool ok = true;
  if (type == 0) { x2 = x1 - 1; if (!pbc) ok = x2 > 0; else { if (x2 == 0) x2 = lx; if (x2 == x1) ok = false; } }
  if (type == 1) { x2 = x1 + 1; if (!pbc) ok = x2 <= lx; else { if (x2 == lx + 1) x2 = 1; if (x2 == x1) ok = false; } }
  if (type == 2) { y2 = y1 + 1; if (!pbc) ok = y2 <= ly; else { if (y2 == ly + 1) y2 = 1; if (y2 == y1) ok = false; } }
  if (type == 3) { y2 = y1 - 1; if (!pbc) ok = y2 > 0; else { if (y2 == 0) y2 = ly; if (y2 == y1) ok = false; } }
  return ok ? (y2 - 1) * lx + x2 : 0;
}

int sqmc_gpu_init_hubbard(const sqmc_hubbard_cfg *cfg, sqmc_gpu_ctx **out) {
  if (!cfg || !out) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (cfg->l_x < 1 || cfg->l_y < 1) return fail(SQMC_ERR_BAD_ARG, "l_x and l_y must be positive");
  const long long ns = (long long)cfg->l_x * cfg->l_y;
  if (ns > SQ_MAXORB) return fail(SQMC_ERR_UNSUPPORTED, "l_x*l_y must be <= 64 (one 64-bit word per spin)");
  if (cfg->nup < 0 || cfg->ndn < 0 || cfg->nup > ns || cfg->ndn > ns || cfg->nup + cfg->ndn < 1) return fail(SQMC_ERR_BAD_ARG, "nup / ndn out of range");
  if (cfg->pbc && (cfg->l_x == 2 || cfg->l_y == 2))
    return fail(SQMC_ERR_UNSUPPORTED, "a periodic direction of length 2 doubles a bond in the reference's connected list but not in hamiltonian_hubbard");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SQMC_ERR_HIP, "no HIP device: libsqmc_gpu has no CPU fallback");
  sqmc_gpu_ctx *c = new sqmc_gpu_ctx();
  memset((void *)c, 0, sizeof(*c));
  HIPCHK(hipStreamCreate(&c->st));
  ChemTab &t = c->htab;
  t.sys_type = 2; t.n_dim = 2; t.hub_t = cfg->t; t.hub_U = cfg->U;
  t.norb = (int)ns; t.nup = cfg->nup; t.ndn = cfg->ndn; t.ncore = 0; t.nelec = cfg->nup + cfg->ndn; t.time_sym = 0; t.z = 1; t.ngroup = 1;
  t.orb_mask = (ns >= 64) ? ~0ull : ((1ull << ns) - 1ull);
  t.c2_stride = 0;
  for (int s = 1; s <= (int)ns; s++) for (int k = 0; k < 4; k++) t.hub_nbr[s][k] = (unsigned char)hubbard_nbr(cfg->l_x, cfg->l_y, cfg->pbc, s, k);
  return init_common(c, (int)ns, cfg->nup, cfg->ndn, cfg->rng_mode, cfg->irand_seed, cfg->mwalk, out);
}

static void comm_release(sqmc_gpu_ctx *c);
int sqmc_gpu_finalize(sqmc_gpu_ctx *c) {
  if (!c) return SQMC_OK;
  hipStreamSynchronize(c->st);
  if (c->mwalk > 0) {
    free_walk(c->w); free_walk(c->m);
    hipFree(c->d_nchild); hipFree(c->d_child_off); hipFree(c->d_wchild); hipFree(c->d_child_state);
    hipFree(c->d_keys); hipFree(c->d_keys_alt); hipFree(c->d_vals); hipFree(c->d_vals_alt); hipFree(c->d_hist); hipFree(c->d_rowtot);
    hipFree(c->d_flags); hipFree(c->d_pos); hipFree(c->d_flags2); hipFree(c->d_pos2); hipFree(c->d_scan_state); hipFree(c->d_scan_ticket); hipFree(c->d_fstate); hipFree(c->d_fticket); hipFree(c->d_partials); hipFree(c->d_wabs_part); hipFree(c->d_done);
  }
  hipFree(c->d_binom); hipFree(c->d_grow);
  comm_release(c);
  hipFree(c->d_tab); hipFree(c->d_ints); hipFree(c->d_hb_r); hipFree(c->d_hb_s); hipFree(c->d_hb_absH); hipFree(c->d_pq_ind); hipFree(c->d_pq_count);
  hipFree(c->d_prj_ptr); hipFree(c->d_prj_col); hipFree(c->d_prj_val); hipFree(c->d_loc_imp); hipFree(c->d_prj_x);
  hipFree(c->d_ct_up); hipFree(c->d_ct_dn); hipFree(c->d_ct_num); hipFree(c->d_ct_den); hipFree(c->d_ct_hkey); hipFree(c->d_ct_hidx);
  hipFree(c->d_sc); hipHostFree(c->h_sc); if (c->h_mail) hipHostFree((void *)c->h_mail);
  for (int i = 0; i < NTIMERS; i++) { hipEventDestroy(c->ev0[i]); hipEventDestroy(c->ev1[i]); }
  hipEventDestroy(c->e_fork); hipEventDestroy(c->e_join); hipEventDestroy(c->e_cnt);
  for (int i = 0; i < 4; i++) hipEventDestroy(c->hev[i]);
  hipStreamDestroy(c->st2); hipStreamDestroy(c->st);
  delete c;
  return SQMC_OK;
}

int sqmc_gpu_set_hb_tables(sqmc_gpu_ctx *c, int64_t n_hb, const int32_t *r, const int32_t *s, const double *a, int32_t n_pq,
                           const int64_t *pq_ind, const int32_t *pq_count, double max_double) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  HIPCHK(hipMalloc(&c->d_hb_r, (n_hb + 1) * 4)); HIPCHK(hipMalloc(&c->d_hb_s, (n_hb + 1) * 4)); HIPCHK(hipMalloc(&c->d_hb_absH, (n_hb + 1) * 8));
  HIPCHK(hipMalloc(&c->d_pq_ind, (n_pq + 1) * 8)); HIPCHK(hipMalloc(&c->d_pq_count, (n_pq + 1) * 4));
  HIPCHK(hipMemcpy(c->d_hb_r, r, n_hb * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->d_hb_s, s, n_hb * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_hb_absH, a, n_hb * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_pq_ind, pq_ind, (n_pq + 1) * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->d_pq_count, pq_count, (n_pq + 1) * 4, hipMemcpyHostToDevice));
  c->dev.hb_r = c->d_hb_r; c->dev.hb_s = c->d_hb_s; c->dev.hb_absH = c->d_hb_absH; c->dev.pq_ind = c->d_pq_ind; c->dev.pq_count = c->d_pq_count;
  c->dev.max_double = max_double;
  return SQMC_OK;
}

int sqmc_gpu_set_projector(sqmc_gpu_ctx *c, int64_t n_imp, int64_t nnz, const int64_t *rc, const int64_t *idx, const double *val) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  long long chk = 0; for (long long i = 0; i < n_imp; i++) chk += rc[i];
  if (chk != nnz) return fail(SQMC_ERR_BAD_ARG, "sum(row_counts) != nnz");
  for (long long k = 0; k < nnz; k++) if (idx[k] < 1 || idx[k] > n_imp) return fail(SQMC_ERR_BAD_ARG, "column index out of range");
  std::vector<int> ptr, col; std::vector<double> v;
  expand_full_csr(n_imp, rc, idx, val, ptr, col, v);
  hipFree(c->d_prj_ptr); hipFree(c->d_prj_col); hipFree(c->d_prj_val); hipFree(c->d_loc_imp); hipFree(c->d_prj_x);
  c->n_imp = n_imp; c->prj_nnz = (long long)col.size();
  HIPCHK(hipMalloc(&c->d_prj_ptr, (n_imp + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_col, (col.size() + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_val, (v.size() + 1) * 8));
  HIPCHK(hipMalloc(&c->d_loc_imp, (n_imp + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_x, (n_imp + 1) * 8));
  HIPCHK(hipMemcpy(c->d_prj_ptr, ptr.data(), (n_imp + 1) * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_prj_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_prj_val, v.data(), v.size() * 8, hipMemcpyHostToDevice));
  return SQMC_OK;
}
int sqmc_gpu_scale_projector(sqmc_gpu_ctx *c, double ratio) {
  if (!c || !c->d_prj_val) return fail(SQMC_ERR_BAD_ARG, "no projector");
  hipLaunchKernelGGL(k_scale, dim3(nblk(c->prj_nnz)), dim3(TPB), 0, c->st, c->d_prj_val, c->prj_nnz, ratio);
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}

int sqmc_gpu_set_ct_table(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, const double *num, const double *den) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  for (long long i = 1; i < n; i++)
    if (!(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "C(T) list must be strictly sorted by (up,dn)");
  hipFree(c->d_ct_up); hipFree(c->d_ct_dn); hipFree(c->d_ct_num); hipFree(c->d_ct_den);
  c->n_ct = n;
  HIPCHK(hipMalloc(&c->d_ct_up, (n + 1) * 8)); HIPCHK(hipMalloc(&c->d_ct_dn, (n + 1) * 8)); HIPCHK(hipMalloc(&c->d_ct_num, (n + 1) * 8)); HIPCHK(hipMalloc(&c->d_ct_den, (n + 1) * 8));
  HIPCHK(hipMemcpy(c->d_ct_up, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->d_ct_dn, dn, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_ct_num, num, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->d_ct_den, den, n * 8, hipMemcpyHostToDevice));
  {
    const u64 lim = c->htab.orb_mask;
    for (long long i = 0; i < n; i++) {
      if ((up[i] & ~lim) || (dn[i] & ~lim) || __builtin_popcountll(up[i]) != c->htab.nup || __builtin_popcountll(dn[i]) != c->htab.ndn)
        return fail(SQMC_ERR_BAD_ARG, "C(T) determinant with the wrong number of electrons or orbitals beyond norb");
    }
    u64 cap = 64; while (cap < 2ull * (u64)n) cap <<= 1;
    hipFree(c->d_ct_hkey); hipFree(c->d_ct_hidx);
    HIPCHK(hipMalloc(&c->d_ct_hkey, cap * 8)); HIPCHK(hipMalloc(&c->d_ct_hidx, cap * 4));
    HIPCHK(hipMemset(c->d_ct_hkey, 0xFF, cap * 8));
    c->ct_mask = cap - 1;
    if (n > 0) hipLaunchKernelGGL(k_ct_build, dim3(nblk(n)), dim3(TPB), 0, c->st, c->dev, c->d_ct_up, c->d_ct_dn, (long long)n, c->d_ct_hkey, c->d_ct_hidx, c->ct_mask);
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->st));
  }
  return SQMC_OK;
}

int sqmc_gpu_upload_walkers(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, const double *wt, const int8_t *impd,
                            const int8_t *init, const int8_t *psign, const double *me, const double *en, const double *ed) {
  if (!c || c->mwalk <= 0) return fail(SQMC_ERR_BAD_ARG, "context has no walker arrays (mwalk=0)");
  if (n > c->mwalk) return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  const u64 lim = c->htab.orb_mask;
  for (long long i = 0; i < n; i++) {
    if ((up[i] & ~lim) || (dn[i] & ~lim)) return fail(SQMC_ERR_BAD_ARG, "determinant has bits beyond norb");
    if (__builtin_popcountll(up[i]) != c->htab.nup || __builtin_popcountll(dn[i]) != c->htab.ndn)
      return fail(SQMC_ERR_BAD_ARG, "determinant does not hold nup / ndn electrons");
    if (i && !(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "walkers must be sorted by (up,dn) and unique");
  }
  HIPCHK(hipMemcpy(c->w.up, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->w.dn, dn, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->w.wt, wt, n * 8, hipMemcpyHostToDevice));
  { std::vector<u32> f(n); for (long long i = 0; i < n; i++) f[i] = pack_flg(impd[i], init[i], psign[i]);
    HIPCHK(hipMemcpy(c->w.flg, f.data(), n * 4, hipMemcpyHostToDevice)); }
  HIPCHK(hipMemcpy(c->w.me, me, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->w.en, en, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->w.ed, ed, n * 8, hipMemcpyHostToDevice));
  c->nwalk = n; c->residents_sorted = false;      // the host's order is taken as it comes: the next step sorts everything
  if (c->n_imp > 0) {        // my_locations_of_imp_dets, do_walk.f90:2188-2212
    const long long expect = c->d_grow ? c->n_imp_local : c->n_imp;
    std::vector<int> loc; loc.reserve(expect);
    for (long long i = 0; i < n; i++) if (impd[i] == 0) loc.push_back((int)i);
    if ((long long)loc.size() != expect) return fail(SQMC_ERR_IMP_BROKEN, "number of imp_distance==0 walkers != n_imp (of this rank)");
    if (!loc.empty()) HIPCHK(hipMemcpy(c->d_loc_imp, loc.data(), loc.size() * 4, hipMemcpyHostToDevice));
  }
  return SQMC_OK;
}
int sqmc_gpu_num_walkers(sqmc_gpu_ctx *c, int64_t *n) { if (!c || !n) return SQMC_ERR_BAD_ARG; *n = c->nwalk; return SQMC_OK; }
int sqmc_gpu_download_walkers(sqmc_gpu_ctx *c, int64_t cap, int64_t *n, uint64_t *up, uint64_t *dn, double *wt, int8_t *impd, int8_t *init,
                              double *me, double *en, double *ed) {
  if (!c || !n) return fail(SQMC_ERR_BAD_ARG, "null");
  HIPCHK(hipStreamSynchronize(c->st));
  *n = c->nwalk;
  if (cap < c->nwalk) return fail(SQMC_ERR_BAD_ARG, "download buffer too small");
  const long long k = c->nwalk;
  if (up) HIPCHK(hipMemcpy(up, c->w.up, k * 8, hipMemcpyDeviceToHost));
  if (dn) HIPCHK(hipMemcpy(dn, c->w.dn, k * 8, hipMemcpyDeviceToHost));
  if (wt) HIPCHK(hipMemcpy(wt, c->w.wt, k * 8, hipMemcpyDeviceToHost));
  if (impd || init) {
    std::vector<u32> f(k);
    HIPCHK(hipMemcpy(f.data(), c->w.flg, k * 4, hipMemcpyDeviceToHost));
    for (long long i = 0; i < k; i++) { if (impd) impd[i] = (int8_t)flg_impd(f[i]); if (init) init[i] = (int8_t)flg_init(f[i]); }
  }
  if (me) HIPCHK(hipMemcpy(me, c->w.me, k * 8, hipMemcpyDeviceToHost));
  if (en) HIPCHK(hipMemcpy(en, c->w.en, k * 8, hipMemcpyDeviceToHost));
  if (ed) HIPCHK(hipMemcpy(ed, c->w.ed, k * 8, hipMemcpyDeviceToHost));
  return SQMC_OK;
}

int sqmc_gpu_get_rng(sqmc_gpu_ctx *c, int32_t seed[4]) {
  if (!c) return SQMC_ERR_BAD_ARG;
  u64 x; HIPCHK(hipStreamSynchronize(c->st)); HIPCHK(hipMemcpy(&x, &c->d_sc->lcg, 8, hipMemcpyDeviceToHost));
  seed[0] = (int)((x >> 36) & 4095); seed[1] = (int)((x >> 24) & 4095); seed[2] = (int)((x >> 12) & 4095); seed[3] = (int)(x & 4095);
  return SQMC_OK;
}
int sqmc_gpu_set_rng(sqmc_gpu_ctx *c, const int32_t seed[4]) {
  if (!c) return SQMC_ERR_BAD_ARG;
  u64 x = (((u64)seed[0] << 36) + ((u64)seed[1] << 24) + ((u64)seed[2] << 12) + (u64)(2 * (seed[3] / 2) + 1)) & SQ_MASK48;
  HIPCHK(hipStreamSynchronize(c->st)); HIPCHK(hipMemcpy(&c->d_sc->lcg, &x, 8, hipMemcpyHostToDevice));
  c->seed64 = x;
  return SQMC_OK;
}

static void collect_timers(sqmc_gpu_ctx *c);
int sqmc_gpu_set_timing(sqmc_gpu_ctx *c, int on) {
  if (!c) return SQMC_ERR_BAD_ARG;
  hipStreamSynchronize(c->st); hipStreamSynchronize(c->st2); c->timers_pending = false;
  c->timing = on; c->tsteps = 0; c->nt = 0;
  for (int i = 0; i < NTIMERS; i++) c->tsum[i] = 0.0;
  return SQMC_OK;
}
int sqmc_gpu_get_timing(sqmc_gpu_ctx *c, int32_t *n, const char **names, float *ms) {
  if (!c) return SQMC_ERR_BAD_ARG;
  collect_timers(c);
  *n = c->tsteps > 0 ? c->nt : 0;
  for (int i = 0; i < *n; i++) { names[i] = c->tname[i]; ms[i] = (float)(c->tsum[i] / (double)c->tsteps); }
  return SQMC_OK;
}
// stage timers: a pair of HIP events on the stream the stage runs on
#define TBEG(NAME, STREAM) int t_##NAME = -1; do { if ((c->timing >= 2 || (c->timing == 1 && !strcmp(#NAME, "spawn") && c->step_no % 8 == 0)) && c->nt < NTIMERS) { t_##NAME = c->nt++; c->tname[t_##NAME] = #NAME; hipEventRecord(c->ev0[t_##NAME], STREAM); } } while (0)
#define TEND(NAME, STREAM) do { if (t_##NAME >= 0) hipEventRecord(c->ev1[t_##NAME], STREAM); } while (0)
// Timing level 1 takes the kernel-exact start/stop events of k_spawn and k_anneal on every SQMC_TIMING_STRIDE-th step only:
// each timed launch costs about 5 us of the step it sits in, which a throughput measurement should not pay on every step.
#define SQMC_TIMING_STRIDE 8
static inline bool kernel_events_on(const sqmc_gpu_ctx *c, u64 step) { return c->timing >= 2 || (c->timing == 1 && step % SQMC_TIMING_STRIDE == 0); }

static int comm_allreduce_stats(sqmc_gpu_ctx *c);
// Spin on a mailbox word the GPU writes into pinned host memory.  Returns 0 when it arrived, -1
// if the stream drained without it, a hipError_t > 0 if the stream reports an error.
static int wait_mail(volatile u64 *flag, u64 expect, hipStream_t st) {
  for (unsigned long it = 1;; it++) {
    if (*flag == expect) return 0;
    if ((it & 0x3FFF) == 0) {
      hipError_t e = hipStreamQuery(st);
      if (e != hipErrorNotReady) { if (*flag == expect) return 0; return e == hipSuccess ? -1 : (int)e; }
    }
    __builtin_ia32_pause();
  }
}
// stage timers of the last step are read when the next step starts (or when they are asked for):
// by then their events have completed and no extra synchronisation is paid inside the step
static void collect_timers(sqmc_gpu_ctx *c) {
  if (!c->timers_pending) return;
  c->timers_pending = false;
  for (int i = 0; i < c->nt; i++) {
    if (hipEventSynchronize(c->ev1[i]) != hipSuccess) continue;
    if (hipEventElapsedTime(&c->tms[i], c->ev0[i], c->ev1[i]) == hipSuccess) c->tsum[i] += c->tms[i];
  }
  c->tsteps++;
}
// sharded contexts: k_spawn fills the owner keys of the bucketing pass (arrays that are free until the annihilation)
static OwnerOut shard_owner_out(sqmc_gpu_ctx *c) {
  if (!c->d_grow) return OwnerOut{nullptr, nullptr, 0};
  c->owner_ready = true;
  return OwnerOut{c->d_flags, (u32 *)c->d_flags2, c->shard_n};
}
// The head of a step: spawn gate + child offsets + k_spawn (COUNTER discipline).  With dev_n the
// walker count is read on the device (sc->nwalk, written by k_finish of the step before) and n0 is
// only an upper bound that sizes the grids: the pipelined launch behind k_finish of the previous
// step, before the host has read that step's sums.  g0/g1 and s0/s1 (may be null) time gate+scan
// and k_spawn; the child count goes to the host mailbox under sequence number *cseq.
static int enqueue_head(sqmc_gpu_ctx *c, const StepP &p, u64 step, long long n0, bool dev_n, hipEvent_t g0, hipEvent_t g1, hipEvent_t s0, hipEvent_t s1, u64 *cseq,
                        const FinArgs *fin = nullptr) {
  hipStream_t st = c->st;
  const long long M = c->mwalk;
  ScanWork sw0; sw0.state = c->d_scan_state; sw0.ticket = c->d_scan_ticket; sw0.cap_tiles = c->cap_tiles; sw0.self_clear = false;
  FinArgs fa; memset(&fa, 0, sizeof(fa)); if (fin) fa = *fin;
  if (g0) hipEventRecord(g0, st);
  hipLaunchKernelGGL(k_gate, dim3(nblk(n0)), dim3(TPB), 0, st, c->dev, c->w.up, c->w.dn, c->w.wt, c->d_nchild, c->d_wchild, c->d_keys, c->d_vals,
                     n0, p, c->seed64, step, c->d_sc, c->pack, dev_n ? 1 : 0, fa);
  device_excl_scan_u64(c->d_nchild, c->d_child_off, n0, &c->d_sc->n_children, sw0, st, dev_n ? &c->d_sc->nwalk : nullptr);
  if (g1) hipEventRecord(g1, st);
  // ---- spawn goes out first: the host is the slower side at the start of a step, and k_spawn
  //      is on the critical path (exactly one k_spawn launch inside this timer: the per-launch time
  //      bench.py reports, taken from the kernel's own start/stop timestamps).  It is launched
  //      over the whole free capacity with a device-side child count and posts that count to the
  //      host mailbox as soon as it starts.
  HIPCHK(hipEventRecord(c->e_fork, st));
  *cseq = ++c->cnt_seq;
  const OwnerOut oo = shard_owner_out(c);
  const long long nfree = dev_n ? M : M - n0;          // dev_n: nothing is known about the count but that it is >= 0
  if (nfree > 0) {
    if (s0)
      hipExtLaunchKernelGGL(k_spawn, dim3(nblk(nfree)), dim3(TPB), 0, st, s0, s1, 0, c->dev, c->w, c->d_child_off, c->d_wchild,
                            c->d_child_state, c->d_keys, c->d_vals, n0, M, p, c->rng_mode, c->seed64, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, *cseq, c->pack, dev_n ? 1 : 0, oo);
    else
      hipLaunchKernelGGL(k_spawn, dim3(nblk(nfree)), dim3(TPB), 0, st, c->dev, c->w, c->d_child_off, c->d_wchild, c->d_child_state, c->d_keys, c->d_vals,
                         n0, M, p, c->rng_mode, c->seed64, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, *cseq, c->pack, dev_n ? 1 : 0, oo);
  } else if (s0) { hipEventRecord(s0, st); hipEventRecord(s1, st); }
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}
// a pipelined head whose step will not run (the step before it failed): drain it and reset what it touched
static void drop_head(sqmc_gpu_ctx *c) {
  if (!c->head_ready) return;
  c->head_ready = false;
  hipStreamSynchronize(c->st);
  hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
}
// sort -> merge -> round -> compact/estimate -> readback; shared by the single-rank step and
// the sharded step (where the spawns behind slot n0 arrived from other ranks)
static int step_tail(sqmc_gpu_ctx *c, const StepP &p_in, long long n0, long long nall, bool join_side_stream, double out[16]) {
  StepP p = p_in;
  p.nimp_cap = c->d_loc_imp ? (int)std::max<long long>(c->n_imp_local, c->n_imp) : 0;     // what set_projector / shard_config allocated
  hipStream_t st = c->st;
  const long long M = c->mwalk;
  const int mode = c->rng_mode; const u64 seed = c->seed64, step = c->step_no;
  ScanWork sw[3];
  for (int q = 0; q < 3; q++) { sw[q].state = c->d_scan_state + q * c->cap_tiles; sw[q].ticket = c->d_scan_ticket + q; sw[q].cap_tiles = c->cap_tiles; sw[q].self_clear = false; }
  // ---- sort
  TBEG(sort, st);
  if (mode == SQMC_RNG_REPLAY)
    hipLaunchKernelGGL(k_main_keys, dim3(nblk(n0)), dim3(TPB), 0, st, c->dev, c->w.up, c->w.dn, c->d_keys, c->d_vals, n0, c->pack);
  SortWork so; so.k_alt = c->d_keys_alt; so.v_alt = c->d_vals_alt; so.hist = c->d_hist; so.rowtot = c->d_rowtot; so.cap = M;
  u64 *skey = c->d_keys; u32 *perm = c->pack ? (u32 *)nullptr : c->d_vals;
  static const long long merge_min = getenv("SQMC_MERGE_SORT_MIN") ? atoll(getenv("SQMC_MERGE_SORT_MIN")) : (1ll << 20);
  if (c->pack && p.semi && c->residents_sorted && nall >= merge_min) {
    // Large lists: the walkers [0, n0) are in order already (every step leaves them so), so only the spawns
    // [n0, nall) are sorted and one stable merge (walker before spawns on equal keys, spawns in creation order)
    // gives the order the full sort would.  The merged list lands in the flag array the fused tail does not use.
    const long long nch = nall - n0;
    if (nch > 0) {
      u64 *sk = c->d_keys + n0; u32 *nov = nullptr;
      so.k_alt = c->d_keys_alt + n0;
      device_radix_sort(sk, nov, nch, c->key_bits, so, st, 32);
      device_merge_sorted(c->d_keys, n0, sk, nch, c->d_flags, 32, st);
      skey = c->d_flags;
    }
  } else {
    device_radix_sort(skey, perm, nall, c->key_bits, so, st, c->pack ? 32 : 0);
    if (skey != c->d_keys) { c->d_keys_alt = c->d_keys; c->d_keys = skey; if (!c->pack) { c->d_vals_alt = c->d_vals; c->d_vals = perm; } }
  }
  TEND(sort, st);
  // ---- join: from here on weights are read
  if (join_side_stream) HIPCHK(hipStreamWaitEvent(st, c->e_join, 0));
  const int nbm = nblk(nall);
  const bool use_mail = (c->comm == nullptr);          // with a communicator the sums are all-reduced on the device first
  const u64 seq = ++c->mail_seq;
  int nb, n_ft = 0;
  if (p.semi) {
    // one kernel from the sorted list to the new walker arrays; the buffers swap roles afterwards
    // timed by the kernel's own start/stop timestamps (hipExtLaunchKernelGGL events) at every timing level: the
    // per-launch time bench.py reports for the roofline of this, the longest kernel on the critical path
    int t_anneal = -1;
    if (kernel_events_on(c, step) && c->nt < NTIMERS) { t_anneal = c->nt++; c->tname[t_anneal] = "anneal"; }
    static const int items_env = getenv("SQMC_ANNEAL_ITEMS") ? atoi(getenv("SQMC_ANNEAL_ITEMS")) : 0;
    const int items = items_env ? items_env : (nall < (1ll << 20) ? 2 : 4);     // small lists want many tiles, large ones short look-back chains
    nb = n_ft = (int)((nall + (long long)TPB * items - 1) / ((long long)TPB * items));
#define ANNEAL_ARGS c->w, c->m, skey, perm, c->d_loc_imp, c->d_ct_hkey, c->d_ct_hidx, c->ct_mask, c->d_ct_num, c->d_ct_den, c->d_partials, c->d_wabs_part, n0, nall, p,  \
                    c->invalid_key, c->pack, mode, seed, step, c->d_sc, c->d_fstate, c->d_fstate + c->cap_ftiles, c->d_fticket
#define ANNEAL_LAUNCH(I) do { if (t_anneal >= 0) hipExtLaunchKernelGGL(k_anneal<I>, dim3(nb), dim3(TPB), 0, st, c->ev0[t_anneal], c->ev1[t_anneal], 0, ANNEAL_ARGS); \
                              else hipLaunchKernelGGL(k_anneal<I>, dim3(nb), dim3(TPB), 0, st, ANNEAL_ARGS); } while (0)
    if (items == 1) ANNEAL_LAUNCH(1); else if (items == 2) ANNEAL_LAUNCH(2); else if (items == 8) ANNEAL_LAUNCH(8); else ANNEAL_LAUNCH(4);
#undef ANNEAL_LAUNCH
#undef ANNEAL_ARGS
    std::swap(c->w.up, c->m.up); std::swap(c->w.dn, c->m.dn); std::swap(c->w.wt, c->m.wt); std::swap(c->w.flg, c->m.flg);
    std::swap(c->w.me, c->m.me); std::swap(c->w.en, c->m.en); std::swap(c->w.ed, c->m.ed);
  } else {
    TBEG(merge, st);
    hipLaunchKernelGGL(k_merge, dim3(nbm), dim3(TPB), 0, st, c->w, c->m, skey, perm, c->d_flags, c->d_wabs_part, n0, nall, p, c->invalid_key, c->pack);
    device_excl_scan_u64(c->d_flags, c->d_pos, nall, &c->d_sc->tot1, sw[1], st);
    TEND(merge, st);
    TBEG(round, st);
    hipLaunchKernelGGL(k_join, dim3(1), dim3(TPB), 0, st, c->m, c->d_flags, c->d_pos, nall, p, mode, seed, step, c->d_sc);
    hipLaunchKernelGGL(k_round, dim3(nblk(nall)), dim3(TPB), 0, st, c->m, c->d_flags, c->d_pos, c->d_flags2, nall, p, mode, seed, step, c->d_sc);
    device_excl_scan_u64(c->d_flags2, c->d_pos2, nall, &c->d_sc->tot2, sw[2], st);
    TEND(round, st);
    nb = std::min(nblk(nall), 2048);
    TBEG(compact, st);
    hipLaunchKernelGGL(k_compact, dim3(nb), dim3(TPB), 0, st, c->m, c->w, c->d_flags2, c->d_pos2, c->d_loc_imp, skey, c->d_ct_hkey, c->d_ct_hidx, c->ct_mask,
                       c->d_ct_num, c->d_ct_den, nall, p, c->d_partials, c->pack);
    TEND(compact, st);
  }
  FinArgs fa;
  fa.partials = c->d_partials; fa.nblocks = nb; fa.wabs_part = c->d_wabs_part; fa.nwabs = p.semi ? nb : nbm; fa.mode = mode;
  fa.scan_state = c->d_scan_state; fa.scan_ticket = c->d_scan_ticket; fa.n_scan_words = (int)(3 * c->cap_tiles);
  fa.mail = use_mail ? c->d_mail : (HostMail *)nullptr; fa.seq = seq; fa.fstate = c->d_fstate; fa.fticket = c->d_fticket; fa.cap_ftiles = c->cap_ftiles;
  fa.n_ftiles = n_ft; fa.on = 1;
  const bool fin_in_gate = c->pipeline_next && p.semi && use_mail;     // the next step's gate kernel does the final sums in its first block
  TBEG(estimate, st);
  if (!fin_in_gate) hipLaunchKernelGGL(k_finish, dim3(1), dim3(TPB), 0, st, fa, c->d_sc);
  TEND(estimate, st);
  HIPCHK(hipGetLastError());
  bool mail_in_gate = false;
  if (!use_mail) {
    int rr = comm_allreduce_stats(c); if (rr) return rr;      // do_walk.f90:2778-2790: the sums every rank needs
    mail_in_gate = c->pipeline_next && p.semi;                 // the next step's gate kernel posts them (one launch less)
    if (!mail_in_gate) hipLaunchKernelGGL(k_post_mail, dim3(1), dim3(64), 0, st, (const DevScalars *)c->d_sc, c->d_mail, seq);
    else { memset(&fa, 0, sizeof(fa)); fa.on = 3; fa.mail = c->d_mail; fa.seq = seq; }
  }
  if (c->pipeline_next) {
    // the next step's gate + scan + spawn go out now, behind k_finish: the GPU runs on while the host
    // reads this step's sums and does its population control.  nall bounds the new walker count.
    c->pipeline_next = false;
    int rh = enqueue_head(c, p, step + 1, nall, true, c->timing >= 2 ? c->hev[0] : nullptr, c->timing >= 2 ? c->hev[1] : nullptr,
                          kernel_events_on(c, step + 1) ? c->hev[2] : nullptr, kernel_events_on(c, step + 1) ? c->hev[3] : nullptr, &c->head_cseq, (fin_in_gate || mail_in_gate) ? &fa : nullptr);
    if (rh) return rh;
    c->head_ready = true; c->head_p = p;
  }
  {
    int wr = wait_mail(&c->h_mail->seq, seq, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) {            // stream drained without the mail: read the scalars the slow way
      HIPCHK(hipMemcpy(c->h_sc, c->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost));
    } else { c->h_sc->tot2 = c->h_mail->tot2; c->h_sc->err = (int)c->h_mail->err; for (int i = 0; i < 16; i++) c->h_sc->stats[i] = c->h_mail->stats[i]; }
  }
  c->timers_pending = kernel_events_on(c, step);
  c->step_no++;
  if (c->h_sc->err) { drop_head(c); return fail(c->h_sc->err, "diagonal_factor<0 after target population has been reached"); }
  const long long nfinal = (long long)(c->h_sc->tot2 & 0xFFFFFFFFull), nimp = (long long)(c->h_sc->tot2 >> 32);
  c->nwalk = nfinal; c->residents_sorted = true;
  for (int i = 0; i < 16; i++) out[i] = c->h_sc->stats[i];
  if (nfinal == 0) { drop_head(c); return fail(SQMC_ERR_NO_WALKERS, "my_nwalk=0"); }
  if (p.semi && nimp != (c->shard_n > 1 || c->d_grow ? c->n_imp_local : c->n_imp)) { drop_head(c); return fail(SQMC_ERR_IMP_BROKEN, "locations of my imp broken"); }
  return SQMC_OK;
}

int sqmc_gpu_step(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double out[16]) {
  if (!c || !sp || !out) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (c->mwalk <= 0 || c->nwalk <= 0) return fail(SQMC_ERR_NO_WALKERS, "my_nwalk=0");
  if (c->d_grow) return fail(SQMC_ERR_BAD_ARG, "context is configured for sharded steps: use sqmc_gpu_shard_begin/pack/finish");
  if (sp->semistochastic && (c->n_imp <= 0 || !c->d_prj_ptr)) return fail(SQMC_ERR_BAD_ARG, "semistochastic step without projector");
  if (!c->d_ct_up) return fail(SQMC_ERR_BAD_ARG, "C(T) table not set");
  hipStream_t st = c->st;
  StepP p; p.tau = sp->tau; p.e_trial = sp->e_trial; p.rfi = sp->reweight_factor_inv; p.r_init = sp->r_initiator; p.min_wt = sp->min_wt;
  p.cutoff = sp->always_spawn_cutoff_wt; p.ipow = sp->initiator_power; p.imind = sp->initiator_min_distance; p.cti = sp->c_t_initiator;
  p.semi = sp->semistochastic; p.reached = sp->reached_w_abs_gen;
  const long long n0 = c->nwalk, M = c->mwalk;
  const int mode = c->rng_mode; const u64 seed = c->seed64, step = c->step_no;
  collect_timers(c);
  c->nt = 0;
  hipStream_t st2 = c->st2;
  int t_gate_scan = -1, t_spawn = -1;
  if (c->timing >= 2 && c->nt < NTIMERS) { t_gate_scan = c->nt++; c->tname[t_gate_scan] = "gate_scan"; }
  if (kernel_events_on(c, step) && c->nt < NTIMERS) { t_spawn = c->nt++; c->tname[t_spawn] = "spawn"; }
  u64 cseq;
  if (c->head_ready) {
    // gate + scan + spawn of this step already run behind k_finish of the last one (pipelined head):
    // they only depend on parameters that are constant once the target population has been reached
    c->head_ready = false;
    const StepP &h = c->head_p;
    if (h.tau != p.tau || h.cutoff != p.cutoff || h.semi != p.semi || h.cti != p.cti || mode == SQMC_RNG_REPLAY) {
      hipStreamSynchronize(st);
      return fail(SQMC_ERR_BAD_ARG, "internal: the pipelined head of this step was launched with other parameters");
    }
    cseq = c->head_cseq;
    if (t_gate_scan >= 0) { std::swap(c->ev0[t_gate_scan], c->hev[0]); std::swap(c->ev1[t_gate_scan], c->hev[1]); }
    if (t_spawn >= 0) { std::swap(c->ev0[t_spawn], c->hev[2]); std::swap(c->ev1[t_spawn], c->hev[3]); }
  } else if (mode == SQMC_RNG_REPLAY) {
    // ---- gate / child offsets (the gate kernel also clears the step's device scalars)
    if (t_gate_scan >= 0) hipEventRecord(c->ev0[t_gate_scan], st);
    hipLaunchKernelGGL(k_replay_prepass, dim3(1), dim3(64), 0, st, c->d_tab, c->w.up, c->w.dn, c->w.wt, c->d_nchild, c->d_wchild, c->d_child_off,
                       c->d_child_state, n0, M - n0, p, c->d_sc);
    if (t_gate_scan >= 0) hipEventRecord(c->ev1[t_gate_scan], st);
    HIPCHK(hipEventRecord(c->e_fork, st));
    cseq = ++c->cnt_seq;
    if (M > n0) {
      if (t_spawn >= 0)
        hipExtLaunchKernelGGL(k_spawn, dim3(nblk(M - n0)), dim3(TPB), 0, st, c->ev0[t_spawn], c->ev1[t_spawn], 0, c->dev, c->w, c->d_child_off, c->d_wchild,
                              c->d_child_state, c->d_keys, c->d_vals, n0, M, p, mode, seed, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, cseq, c->pack, 0, OwnerOut{nullptr, nullptr, 0});
      else
        hipLaunchKernelGGL(k_spawn, dim3(nblk(M - n0)), dim3(TPB), 0, st, c->dev, c->w, c->d_child_off, c->d_wchild, c->d_child_state, c->d_keys, c->d_vals,
                           n0, M, p, mode, seed, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, cseq, c->pack, 0, OwnerOut{nullptr, nullptr, 0});
    } else if (t_spawn >= 0) { hipEventRecord(c->ev0[t_spawn], st); hipEventRecord(c->ev1[t_spawn], st); }
  } else {
    int r = enqueue_head(c, p, step, n0, false, t_gate_scan >= 0 ? c->ev0[t_gate_scan] : nullptr, t_gate_scan >= 0 ? c->ev1[t_gate_scan] : nullptr,
                         t_spawn >= 0 ? c->ev0[t_spawn] : nullptr, t_spawn >= 0 ? c->ev1[t_spawn] : nullptr, &cseq);
    if (r) return r;
  }
  // ---- fork: death/clone and the deterministic projection only touch weights, which neither
  //      the spawn kernel (it uses the child weights of the gate) nor the sort reads
  HIPCHK(hipStreamWaitEvent(st2, c->e_fork, 0));
  TBEG(diag, st2);
  hipLaunchKernelGGL(k_diag, dim3(nblk(n0)), dim3(TPB), 0, st2, c->dev, c->w.up, c->w.dn, c->w.wt, c->w.flg, c->w.me, n0, p, c->d_sc);
  TEND(diag, st2);
  TBEG(project, st2);
  if (p.semi) {
    hipLaunchKernelGGL(k_prj_gather, dim3(nblk(c->n_imp)), dim3(TPB), 0, st2, c->w.wt, c->d_loc_imp, c->d_prj_x, c->n_imp);
    hipLaunchKernelGGL(k_prj_apply, dim3(nblk(c->n_imp, TPB / 64)), dim3(TPB), 0, st2, c->d_prj_ptr, c->d_prj_col, c->d_prj_val, c->d_prj_x, c->d_loc_imp, c->w.wt,
                       c->n_imp, p.e_trial, p.tau);
  }
  TEND(project, st2);
  HIPCHK(hipEventRecord(c->e_join, st2));
  HIPCHK(hipGetLastError());
  // ---- the child count, from the mailbox (or the slow way when there was no k_spawn launch)
  long long nch;
  if (M > n0) {
    int wr = wait_mail(&c->h_mail->cnt_seq, cseq, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) { u64 v; HIPCHK(hipMemcpy(&v, &c->d_sc->n_children, 8, hipMemcpyDeviceToHost)); nch = (long long)v; }
    else nch = (long long)c->h_mail->n_children;
  } else { u64 v; HIPCHK(hipMemcpyAsync(&v, &c->d_sc->n_children, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); nch = (long long)v; }
  if (n0 + nch > M) {
    hipStreamSynchronize(st); hipStreamSynchronize(st2);
    hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
    return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  }
  const long long nall = n0 + nch;
  return step_tail(c, p, n0, nall, true, out);
}

int sqmc_gpu_shard_step(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double out[16]);
typedef int (*step_fn)(sqmc_gpu_ctx *, const sqmc_step_params *, double *);
static int run_steps(sqmc_gpu_ctx *c, sqmc_popctl *pc, int64_t nsteps, double *stats, double totals[16], step_fn one_step) {
  if (!c || !pc || !totals || nsteps < 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  for (int k = 0; k < 16; k++) totals[k] = 0.0;
  for (int64_t it = 0; it < nsteps; it++) {
    // do_walk.f90:2175-2184
    if (pc->reached_w_abs_gen == 0) {
      const double f = 1.0 + log(pc->w_abs_gen_target / pc->w_abs_gen);
      pc->tau = pc->tau_sav * f;
      pc->r_initiator = pc->r_initiator_sav * pow(f, pc->initiator_rescale_power);
      const double ratio = pc->tau / pc->tau_prev;
      if (ratio != 1.0 && pc->semistochastic) { int r = sqmc_gpu_scale_projector(c, ratio); if (r) return r; }
    }
    sqmc_step_params sp;
    sp.tau = pc->tau; sp.e_trial = pc->e_trial; sp.reweight_factor_inv = pc->reweight_factor_inv; sp.r_initiator = pc->r_initiator;
    sp.min_wt = pc->min_wt; sp.always_spawn_cutoff_wt = pc->always_spawn_cutoff_wt; sp.initiator_power = pc->initiator_power;
    sp.initiator_min_distance = pc->initiator_min_distance; sp.c_t_initiator = pc->c_t_initiator; sp.semistochastic = pc->semistochastic;
    sp.reached_w_abs_gen = pc->reached_w_abs_gen; sp.reserved = 0;
    // pipelined head: once the target population has been reached tau and r_initiator stay put, and the head of a step
    // (gate, scan, spawn) depends on nothing else that this step's sums could change
    c->pipeline_next = ((one_step == (step_fn)sqmc_gpu_step || (one_step == (step_fn)sqmc_gpu_shard_step && c->comm2 != nullptr)) && it + 1 < nsteps &&
                        pc->reached_w_abs_gen == 2 && c->rng_mode != SQMC_RNG_REPLAY && !getenv("SQMC_NO_PIPELINE"));
    double out[16];
    int r = one_step(c, &sp, out);
    c->pipeline_next = false;
    if (r) { drop_head(c); return r; }
    if (stats) memcpy(stats + it * 16, out, sizeof(out));
    for (int k = 0; k < 16; k++) totals[k] += out[k];
    // do_walk.f90:2880-2923
    pc->istep++;
    const double w_abs_gen = out[1], e_den_gen = out[2], e_num_gen = out[3];
    if (e_den_gen != 0.0) pc->e_num_cum += e_num_gen * (e_den_gen > 0 ? 1.0 : -1.0);
    pc->e_den_cum += fabs(e_den_gen);
    if (pc->e_den_cum != 0.0) pc->e_est = pc->e_num_cum / pc->e_den_cum;
    const double pw = fmin(1.0, pc->tau * pc->population_control_exponent);
    if (pc->istep <= pc->n_equil) {
      const double d = pc->e_est - pc->e_trial;
      pc->e_trial = pc->e_trial + (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0)) * fmin(fabs(d), 1.0);
      pc->reweight_factor_inv = fmin(2.0, fmax(0.5, pow(pc->w_abs_gen_target / w_abs_gen, pw)));
    } else {
      pc->reweight_factor_inv = fmin(2.0, fmax(0.5, (1.0 / (1.0 + pc->tau * (pc->e_trial - pc->e_est))) * pow(pc->w_abs_gen_target / w_abs_gen, pw)));
    }
    pc->reweight_factor_inv = fmin(pc->reweight_factor_inv, pc->reweight_factor_inv_max);
    if (pc->reached_w_abs_gen == 0 && w_abs_gen >= pc->w_abs_gen_target) {
      pc->reached_w_abs_gen = 2;
      const double ratio = pc->tau_sav / pc->tau;
      pc->tau = pc->tau_sav; pc->r_initiator = pc->r_initiator_sav;
      if (ratio != 1.0 && pc->semistochastic) { int r2 = sqmc_gpu_scale_projector(c, ratio); if (r2) return r2; }
    }
    pc->tau_prev = pc->tau; pc->w_abs_gen = w_abs_gen;
  }
  return SQMC_OK;
}
int sqmc_gpu_run(sqmc_gpu_ctx *c, sqmc_popctl *pc, int64_t nsteps, double *stats, double totals[16]) {
  return run_steps(c, pc, nsteps, stats, totals, sqmc_gpu_step);
}


// keys of caller-supplied spawns (the "no walker" marker for weight 0, as k_spawn writes it)
__global__ void __launch_bounds__(TPB) k_spawn_keys(ChemDev dev, WalkArr w, u64 *__restrict__ keys, u32 *__restrict__ vals, long long n0, long long nall, u64 invalid_key,
                                                    int pack) {
  long long k = n0 + (long long)blockIdx.x * TPB + threadIdx.x;
  if (k >= nall) return;
  const SpawnRec r = w.sp[k - n0];
  put_key(keys, vals, k, (r.wt != 0.0) ? det_key(dev, r.up, r.dn) : invalid_key, pack);
}

// The second half of a step on its own: the caller's spawned walkers (creation order) are appended
// behind the resident walkers, then sort -> merge_original_with_spawned2 -> reduce_my_walker ->
// estimator sums run as in sqmc_gpu_step.  do_walk.f90:2364-2487 as one call; also the door the
// parity tests use to put hand-built collision cases through k_merge.
int sqmc_gpu_annihilate(sqmc_gpu_ctx *c, const sqmc_step_params *sp, int64_t n_spawn, const uint64_t *up, const uint64_t *dn, const double *wt,
                        const int8_t *impd, const int8_t *init, double out[16]) {
  if (!c || !sp || !out || n_spawn < 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  if (n_spawn > 0 && (!up || !dn || !wt || !impd || !init)) return fail(SQMC_ERR_BAD_ARG, "null spawn array");
  if (c->mwalk <= 0 || c->nwalk <= 0) return fail(SQMC_ERR_NO_WALKERS, "my_nwalk=0");
  if (c->d_grow) return fail(SQMC_ERR_BAD_ARG, "context is configured for sharded steps");
  if (!c->d_ct_up) return fail(SQMC_ERR_BAD_ARG, "C(T) table not set");
  const long long n0 = c->nwalk, nall = n0 + n_spawn;
  if (nall > c->mwalk) return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  const u64 lim = c->htab.orb_mask;
  for (long long i = 0; i < n_spawn; i++) if ((up[i] & ~lim) || (dn[i] & ~lim)) return fail(SQMC_ERR_BAD_ARG, "determinant has bits beyond norb");
  hipStream_t st = c->st;
  StepP p; p.tau = sp->tau; p.e_trial = sp->e_trial; p.rfi = sp->reweight_factor_inv; p.r_init = sp->r_initiator; p.min_wt = sp->min_wt;
  p.cutoff = sp->always_spawn_cutoff_wt; p.ipow = sp->initiator_power; p.imind = sp->initiator_min_distance; p.cti = sp->c_t_initiator;
  p.semi = sp->semistochastic; p.reached = sp->reached_w_abs_gen;
  collect_timers(c);
  c->nt = 0;
  HIPCHK(hipStreamSynchronize(st));
  if (n_spawn > 0) {
    std::vector<SpawnRec> recs(n_spawn);
    for (long long i = 0; i < n_spawn; i++) { recs[i].up = up[i]; recs[i].dn = dn[i]; recs[i].wt = wt[i]; recs[i].flg = pack_flg(impd[i], init[i], 0); }
    HIPCHK(hipMemcpy(c->w.sp, recs.data(), n_spawn * sizeof(SpawnRec), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemsetAsync(&c->d_sc->n_children, 0, 4 * sizeof(u64) + 2 * sizeof(int), st));
  hipLaunchKernelGGL(k_main_keys, dim3(nblk(n0)), dim3(TPB), 0, st, c->dev, c->w.up, c->w.dn, c->d_keys, c->d_vals, n0, c->pack);
  if (n_spawn > 0) hipLaunchKernelGGL(k_spawn_keys, dim3(nblk(n_spawn)), dim3(TPB), 0, st, c->dev, c->w, c->d_keys, c->d_vals, n0, nall, c->invalid_key, c->pack);
  return step_tail(c, p, n0, nall, false, out);
}

// ------------------------------------------------------------------ multi-rank sharding
// owner of a determinant (the role of get_det_owner, mpi_routines.f90:419-445; any hash will
// do for ownership)
__global__ void __launch_bounds__(TPB) k_owner_batch(ChemDev dev, const u64 *up, const u64 *dn, int *owner, long long n, int nranks) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) owner[i] = det_owner(det_key(dev, up[i], dn[i]), nranks);
}
// x_global(grow(k)) = w(loc(k)) for the deterministic-space walkers this rank owns
__global__ void __launch_bounds__(TPB) k_prj_gather_rows(const double *__restrict__ wt, const int *__restrict__ loc, const int *__restrict__ grow,
                                                         double *__restrict__ xg, long long n) {
  long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) xg[grow[i]] = wt[loc[i]];
}
// rows owned by this rank of y = A x_global, same ordered accumulation as k_prj_apply
__global__ void __launch_bounds__(TPB) k_prj_apply_rows(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ xg, const int *__restrict__ loc, const int *__restrict__ grow,
                                                        double *__restrict__ wt, long long n, double e_trial, double tau) {
  __shared__ double sprod[TPB / 64][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * (TPB / 64) + wv;
  if (i >= n) return;
  const int row = grow[i];
  const int b = ptr[row], e = ptr[row + 1];
  double y = 0.0;
  for (int base = b; base < e; base += 64) {
    const int k = base + lane;
    sprod[wv][lane] = (k < e) ? val[k] * xg[col[k]] : 0.0;
    __builtin_amdgcn_wave_barrier();
    const int cnt = (e - base < 64) ? (e - base) : 64;
    for (int l = 0; l < cnt; l++) y = y + sprod[wv][l];
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) { y = y + e_trial * tau * xg[row]; wt[loc[i]] = wt[loc[i]] + y; }
}
// destination rank of every child (nranks = "no walker": weight 0), as an 8-bit sort key
__global__ void __launch_bounds__(TPB) k_child_owner(const u64 *__restrict__ keys, u64 *__restrict__ okey, u32 *__restrict__ oval,
                                                     long long n0, long long nch, u64 invalid_key, int nranks, int pack) {
  long long c = (long long)blockIdx.x * TPB + threadIdx.x;
  if (c >= nch) return;
  const u64 k = get_key(keys, n0 + c, pack);
  okey[c] = (k == invalid_key) ? (u64)nranks : (u64)det_owner(k, nranks);
  oval[c] = (u32)c;
}
// 32-byte wire record {up, dn, weight bits, flags}: the t_walk of mpi_routines.f90:29-34 without
// the fields that are sentinels for fresh spawns
__global__ void __launch_bounds__(TPB) k_pack_send(WalkArr w, const u32 *__restrict__ order, u64 *__restrict__ rec, long long n0, long long nsend) {
  long long q = (long long)blockIdx.x * TPB + threadIdx.x;
  if (q >= nsend) return;
  const long long k = n0 + order[q];
  const SpawnRec r = w.sp[k - n0];
  rec[4 * q] = r.up; rec[4 * q + 1] = r.dn; rec[4 * q + 2] = (u64)__double_as_longlong(r.wt); rec[4 * q + 3] = r.flg;
}
__global__ void __launch_bounds__(TPB) k_unpack_recv(ChemDev dev, WalkArr w, const u64 *__restrict__ rec, u64 *__restrict__ keys, u32 *__restrict__ vals,
                                                     long long n0, long long nrecv, int pack, const u64 *__restrict__ self_rec, long long self_lo, long long self_hi) {
  long long q = (long long)blockIdx.x * TPB + threadIdx.x;
  if (q >= nrecv) return;
  const long long k = n0 + q;
  // records [self_lo, self_hi) are this rank's own bucket: they are read where k_pack_send left them (no copy into the receive buffer)
  const u64 *src = (self_rec && q >= self_lo && q < self_hi) ? self_rec + 4 * (q - self_lo) : rec + 4 * q;
  const u64 u = src[0], d = src[1];
  SpawnRec r; r.up = u; r.dn = d; r.wt = __longlong_as_double((long long)src[2]); r.flg = src[3] & 0xFFFFFFFFull;
  w.sp[q] = r;
  put_key(keys, vals, k, det_key(dev, u, d), pack);
}

int sqmc_gpu_det_owner(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, int32_t nranks, int32_t *owner) {
  if (!c || nranks < 1 || nranks > 255) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  if (n <= 0) return SQMC_OK;
  u64 *du, *dd; int *dout;
  HIPCHK(hipMalloc(&du, n * 8)); HIPCHK(hipMalloc(&dd, n * 8)); HIPCHK(hipMalloc(&dout, n * 4));
  HIPCHK(hipMemcpy(du, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd, dn, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_owner_batch, dim3(nblk(n)), dim3(TPB), 0, c->st, c->dev, du, dd, dout, (long long)n, (int)nranks);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipMemcpy(owner, dout, n * 4, hipMemcpyDeviceToHost));
  hipFree(du); hipFree(dd); hipFree(dout);
  return SQMC_OK;
}

int sqmc_gpu_shard_config(sqmc_gpu_ctx *c, int32_t rank, int32_t nranks, int64_t n_imp_local, const int32_t *global_row) {
  if (!c || nranks < 1 || nranks > 255 || rank < 0 || rank >= nranks || n_imp_local < 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  if (n_imp_local > 0 && (!global_row || !c->d_prj_ptr)) return fail(SQMC_ERR_BAD_ARG, "set the (global) projector before shard_config");
  for (long long i = 0; i < n_imp_local; i++) if (global_row[i] < 0 || global_row[i] >= c->n_imp) return fail(SQMC_ERR_BAD_ARG, "global row out of range");
  c->shard_rank = rank; c->shard_n = nranks; c->n_imp_local = n_imp_local;
  hipFree(c->d_grow); hipFree(c->d_loc_imp);
  HIPCHK(hipMalloc(&c->d_grow, (n_imp_local + 1) * 4)); HIPCHK(hipMalloc(&c->d_loc_imp, (std::max<long long>(n_imp_local, c->n_imp) + 1) * 4));
  if (n_imp_local > 0) HIPCHK(hipMemcpy(c->d_grow, global_row, n_imp_local * 4, hipMemcpyHostToDevice));
  return SQMC_OK;
}

// phase 1 of a sharded step: gate, child offsets, death/clone, spawn (into local slots), and the
// owned entries of the deterministic-space weight vector written into x_global (device pointer,
// n_imp doubles, zeroed here) for the caller's all-reduce (do_walk.f90:2259-2260).
static int shard_begin_impl(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double *x_global_dev, int64_t *n_children, bool full_sync, bool side = false) {
  if (!c || !sp || !n_children) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (c->shard_n < 1 || !c->d_grow) return fail(SQMC_ERR_BAD_ARG, "sqmc_gpu_shard_config not called");
  if (c->rng_mode != SQMC_RNG_COUNTER) return fail(SQMC_ERR_UNSUPPORTED, "sharded steps need the COUNTER RNG discipline");
  if (!sp->semistochastic || !c->d_ct_up) return fail(SQMC_ERR_UNSUPPORTED, "sharded step: semistochastic walk with the C(T) table set");
  if (c->mwalk <= 0) return fail(SQMC_ERR_BAD_ARG, "no walker arrays");
  hipStream_t st = c->st;
  StepP p; p.tau = sp->tau; p.e_trial = sp->e_trial; p.rfi = sp->reweight_factor_inv; p.r_init = sp->r_initiator; p.min_wt = sp->min_wt;
  p.cutoff = sp->always_spawn_cutoff_wt; p.ipow = sp->initiator_power; p.imind = sp->initiator_min_distance; p.cti = sp->c_t_initiator;
  p.semi = sp->semistochastic; p.reached = sp->reached_w_abs_gen;
  const long long n0 = c->nwalk, M = c->mwalk;
  ScanWork sw0; sw0.state = c->d_scan_state; sw0.ticket = c->d_scan_ticket; sw0.cap_tiles = c->cap_tiles; sw0.self_clear = false;
  collect_timers(c);
  c->nt = 0;
  if (n0 == 0) HIPCHK(hipMemsetAsync(&c->d_sc->n_children, 0, 4 * sizeof(u64) + 2 * sizeof(int), st));   // otherwise k_gate clears them
  // side = the in-library step with a second communicator: death/clone, the gather of the owned
  // deterministic weights (and later their all-reduce and the projection) run on the side stream
  // beside k_spawn, the bucketing of the spawns and their exchange
  hipStream_t sx = side ? c->st2 : st;
  if (!side && x_global_dev && c->n_imp > 0) HIPCHK(hipMemsetAsync(x_global_dev, 0, c->n_imp * 8, st));
  const bool mail = (n0 > 0 && M > n0);
  u64 cseq;
  if (c->head_ready) {
    // gate + scan + spawn of this step already run behind the last step's mail (pipelined head of sqmc_gpu_shard_run)
    c->head_ready = false;
    const StepP &h = c->head_p;
    if (!side || n0 <= 0 || M <= n0 || h.tau != p.tau || h.cutoff != p.cutoff || h.semi != p.semi || h.cti != p.cti) {
      hipStreamSynchronize(st);
      return fail(SQMC_ERR_BAD_ARG, "internal: the pipelined head of this sharded step does not fit it");
    }
    cseq = c->head_cseq;
    if (kernel_events_on(c, c->step_no) && c->nt < NTIMERS) { const int t = c->nt++; c->tname[t] = "spawn"; std::swap(c->ev0[t], c->hev[2]); std::swap(c->ev1[t], c->hev[3]); }
  } else {
    cseq = ++c->cnt_seq;
    if (n0 > 0) {
      hipLaunchKernelGGL(k_gate, dim3(nblk(n0)), dim3(TPB), 0, st, c->dev, c->w.up, c->w.dn, c->w.wt, c->d_nchild, c->d_wchild, c->d_keys, c->d_vals,
                         n0, p, c->seed64, c->step_no, c->d_sc, c->pack, 0, FinArgs{});
      device_excl_scan_u64(c->d_nchild, c->d_child_off, n0, &c->d_sc->n_children, sw0, st);
    }
    if (side) HIPCHK(hipEventRecord(c->e_fork, st));
    if (n0 > 0) {
      TBEG(spawn, st);
      if (M > n0)      // first: it posts the child count to the host mailbox as soon as it starts
        hipLaunchKernelGGL(k_spawn, dim3(nblk(M - n0)), dim3(TPB), 0, st, c->dev, c->w, c->d_child_off, c->d_wchild, c->d_child_state, c->d_keys, c->d_vals,
                           n0, M, p, c->rng_mode, c->seed64, c->step_no, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, cseq, c->pack, 0, shard_owner_out(c));
      TEND(spawn, st);
    }
  }
  if (side) {
    HIPCHK(hipStreamWaitEvent(sx, c->e_fork, 0));
    if (x_global_dev && c->n_imp > 0) HIPCHK(hipMemsetAsync(x_global_dev, 0, c->n_imp * 8, sx));
  }
  if (n0 > 0) {
    hipLaunchKernelGGL(k_diag, dim3(nblk(n0)), dim3(TPB), 0, sx, c->dev, c->w.up, c->w.dn, c->w.wt, c->w.flg, c->w.me, n0, p, c->d_sc);
    if (c->n_imp_local > 0)
      hipLaunchKernelGGL(k_prj_gather_rows, dim3(nblk(c->n_imp_local)), dim3(TPB), 0, sx, c->w.wt, c->d_loc_imp, c->d_grow, x_global_dev, c->n_imp_local);
  }
  HIPCHK(hipGetLastError());
  long long nch = 0;
  int wr = mail ? wait_mail(&c->h_mail->cnt_seq, cseq, st) : -1;
  if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
  if (wr == 0) nch = (long long)c->h_mail->n_children;
  else if (n0 > 0) { u64 v; HIPCHK(hipMemcpyAsync(&v, &c->d_sc->n_children, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); nch = (long long)v; }
  if (full_sync) HIPCHK(hipStreamSynchronize(st));      // the caller's collective library reads x_global on its own stream
  if (n0 + nch > M) {
    hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
    return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  }
  c->shard_n0 = n0; c->shard_nch = nch;
  *n_children = nch;
  return SQMC_OK;
}

int sqmc_gpu_shard_begin(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double *x_global_dev, int64_t *n_children) {
  return shard_begin_impl(c, sp, x_global_dev, n_children, true);
}

// phase 2: apply the owned rows of the deterministic projection with the all-reduced x_global, then
// bucket this step's children by owner rank (stable) into 32-byte records: send_counts[r] records
// for rank r, contiguous in rank order in send_dev (capacity cap_records).
//
// device part of phase 2; leaves the per-destination counts in d_rowtot[0..P) (valid when nch > 0)
// and the permutation of the children by destination in *order.  Children that produced no
// walker sort behind the last rank.
static int shard_bucket(sqmc_gpu_ctx *c, const sqmc_step_params *sp, const double *x_global_dev, u32 **order, bool apply_rows = true) {
  hipStream_t st = c->st;
  const long long n0 = c->shard_n0, nch = c->shard_nch; const int P = c->shard_n;
  if (apply_rows && c->n_imp_local > 0)
    hipLaunchKernelGGL(k_prj_apply_rows, dim3(nblk(c->n_imp_local, TPB / 64)), dim3(TPB), 0, st, c->d_prj_ptr, c->d_prj_col, c->d_prj_val, x_global_dev,
                       c->d_loc_imp, c->d_grow, c->w.wt, c->n_imp_local, sp->e_trial, sp->tau);
  *order = nullptr;
  if (nch > 0) {
    u64 *okey = c->d_flags, *okey_alt = c->d_pos; u32 *oval = (u32 *)c->d_flags2, *oval_alt = (u32 *)c->d_pos2;
    if (!c->owner_ready) hipLaunchKernelGGL(k_child_owner, dim3(nblk(nch)), dim3(TPB), 0, st, c->d_keys, okey, oval, n0, nch, c->invalid_key, P, c->pack);
    SortWork so; so.k_alt = okey_alt; so.v_alt = oval_alt; so.hist = c->d_hist; so.rowtot = c->d_rowtot; so.cap = c->mwalk;
    u64 *sk = okey; u32 *sv = oval;
    device_radix_sort(sk, sv, nch, 8, so, st);               // one stable 8-bit pass; rowtot[d] = children per destination
    *order = sv;
  }
  c->owner_ready = false;
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}

int sqmc_gpu_shard_pack(sqmc_gpu_ctx *c, const sqmc_step_params *sp, const double *x_global_dev, uint64_t *send_dev, int64_t cap_records,
                        int64_t *send_counts) {
  if (!c || !sp || !send_counts) return fail(SQMC_ERR_BAD_ARG, "null argument");
  hipStream_t st = c->st;
  const long long n0 = c->shard_n0, nch = c->shard_nch; const int P = c->shard_n;
  u32 *order;
  int r = shard_bucket(c, sp, x_global_dev, &order); if (r) return r;
  for (int q = 0; q < P; q++) send_counts[q] = 0;
  if (nch > 0) {
    u32 cnt[256];
    HIPCHK(hipMemcpyAsync(cnt, c->d_rowtot, 256 * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    long long nsend = 0;
    for (int q = 0; q < P; q++) { send_counts[q] = cnt[q]; nsend += cnt[q]; }
    if (nsend > cap_records) return fail(SQMC_ERR_SPAWN_OVERFLOW, "send buffer too small for this step's spawns");
    if (nsend > 0) hipLaunchKernelGGL(k_pack_send, dim3(nblk(nsend)), dim3(TPB), 0, st, c->w, order, (u64 *)send_dev, n0, nsend);
  }
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
  return SQMC_OK;
}

// phase 3: the records received from all ranks (rank order, creation order inside a rank) become
// the spawned walkers behind the occupied slots; then the usual sort / merge / round / estimate.
static int shard_finish_impl(sqmc_gpu_ctx *c, const sqmc_step_params *sp, const uint64_t *recv_dev, int64_t n_recv, double out[16], bool join,
                             const u64 *self_rec = nullptr, long long self_lo = 0, long long self_hi = 0) {
  if (!c || !sp || !out || n_recv < 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  hipStream_t st = c->st;
  StepP p; p.tau = sp->tau; p.e_trial = sp->e_trial; p.rfi = sp->reweight_factor_inv; p.r_init = sp->r_initiator; p.min_wt = sp->min_wt;
  p.cutoff = sp->always_spawn_cutoff_wt; p.ipow = sp->initiator_power; p.imind = sp->initiator_min_distance; p.cti = sp->c_t_initiator;
  p.semi = sp->semistochastic; p.reached = sp->reached_w_abs_gen;
  const long long n0 = c->shard_n0;
  if (n0 + n_recv > c->mwalk) {
    hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
    return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  }
  if (n_recv > 0)
    hipLaunchKernelGGL(k_unpack_recv, dim3(nblk(n_recv)), dim3(TPB), 0, st, c->dev, c->w, (const u64 *)recv_dev, c->d_keys, c->d_vals, n0, (long long)n_recv, c->pack, self_rec, self_lo, self_hi);
  if (n0 + n_recv == 0) {           // an empty shard stays empty this step
    if (join) HIPCHK(hipStreamWaitEvent(st, c->e_join, 0));
    for (int i = 0; i < 16; i++) out[i] = 0.0;
    hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
    c->step_no++;
    if (c->comm) {                  // still a party to the all-reduce of the sums
      HIPCHK(hipMemsetAsync(c->d_sc->stats, 0, 16 * 8, st));
      int rr = comm_allreduce_stats(c); if (rr) return rr;
      HIPCHK(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      c->mail_seq++;
      for (int i = 0; i < 7; i++) out[i] = c->h_sc->stats[i];
    }
    return SQMC_OK;
  }
  int r = step_tail(c, p, n0, n0 + n_recv, join, out);
  if (r == SQMC_ERR_NO_WALKERS) r = SQMC_OK;      // a shard may legitimately own nothing
  return r;
}
int sqmc_gpu_shard_finish(sqmc_gpu_ctx *c, const sqmc_step_params *sp, const uint64_t *recv_dev, int64_t n_recv, double out[16]) {
  return shard_finish_impl(c, sp, recv_dev, n_recv, out, false);
}


// ------------------------------------------------------------------ in-library exchange (RCCL over xGMI)
// The three exchanges of a sharded step issued from the library on its own stream, so a step
// costs no Python and no extra host round trips: all-reduce of the deterministic-space vector
// (do_walk.f90:2259-2260), all-to-all of the spawned walkers (mpi_snd_list, mpi_routines.f90:
// 1147-1270 -- here grouped ncclSend/ncclRecv of 32-byte records straight between HBMs), and
// the all-reduce of the seven sums (do_walk.f90:2778-2790).  RCCL is bound at run time from
// the copy already in the process (torch's) or the ROCm one, so single-GPU users never load it.
struct RcclApi {
  void *lib;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *);     // optional (NCCL >= 2.18)
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  const char *(*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl;
static int rccl_bind() {
  if (g_rccl.lib) return SQMC_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  if (const char *over = getenv("SQMC_RCCL_LIB")) h = dlopen(over, RTLD_NOW | RTLD_LOCAL);       // another library with the same entry points (tests: a transport double)
  else for (const char *nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) return fail(SQMC_ERR_UNSUPPORTED, std::string("cannot load RCCL: ") + dlerror());
#define BIND(F) do { *(void **)(&g_rccl.F) = dlsym(h, "nccl" #F); if (!g_rccl.F) return fail(SQMC_ERR_UNSUPPORTED, "RCCL lacks nccl" #F); } while (0)
  BIND(GetUniqueId); BIND(CommInitRank); BIND(CommDestroy); BIND(AllReduce); BIND(AllGather); BIND(Send); BIND(Recv); BIND(GroupStart); BIND(GroupEnd);
  BIND(GetErrorString);
#undef BIND
  *(void **)(&g_rccl.CommSplit) = dlsym(h, "ncclCommSplit");
  g_rccl.lib = h;
  return SQMC_OK;
}
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(SQMC_ERR_HIP, std::string(#x) + ": " + g_rccl.GetErrorString(r_)); } while (0)

static void comm_release(sqmc_gpu_ctx *c) {
  if (c->comm2 && g_rccl.lib) g_rccl.CommDestroy(c->comm2);
  if (c->comm && g_rccl.lib) g_rccl.CommDestroy(c->comm);
  c->comm = nullptr; c->comm2 = nullptr;
  hipFree(c->d_xg); hipFree(c->d_send); hipFree(c->d_recv); hipFree(c->d_cnt_mine); hipFree(c->d_cnt_all);
  if (c->h_cnt_all) hipHostFree(c->h_cnt_all);
  c->d_xg = nullptr; c->d_send = c->d_recv = nullptr; c->d_cnt_mine = c->d_cnt_all = nullptr; c->h_cnt_all = nullptr;
}
static int comm_allreduce_stats(sqmc_gpu_ctx *c) {
  NCCLCHK(g_rccl.AllReduce(c->d_sc->stats, c->d_sc->stats, 7, ncclDouble, ncclSum, c->comm, c->st));
  return SQMC_OK;
}

int sqmc_gpu_comm_unique_id(uint8_t id[SQMC_COMM_ID_BYTES]) {
  if (!id) return fail(SQMC_ERR_BAD_ARG, "null argument");
  static_assert(sizeof(ncclUniqueId) <= SQMC_COMM_ID_BYTES, "unique id size");
  int r = rccl_bind(); if (r) return r;
  ncclUniqueId u;
  NCCLCHK(g_rccl.GetUniqueId(&u));
  memset(id, 0, SQMC_COMM_ID_BYTES); memcpy(id, &u, sizeof(u));
  return SQMC_OK;
}

int sqmc_gpu_comm_init(sqmc_gpu_ctx *c, const uint8_t id[SQMC_COMM_ID_BYTES]) {
  if (!c || !id) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (c->shard_n < 1 || !c->d_grow) return fail(SQMC_ERR_BAD_ARG, "sqmc_gpu_shard_config not called");
  if (c->mwalk <= 0) return fail(SQMC_ERR_BAD_ARG, "no walker arrays");
  int r = rccl_bind(); if (r) return r;
  comm_release(c);
  ncclUniqueId u; memcpy(&u, id, sizeof(u));
  NCCLCHK(g_rccl.CommInitRank(&c->comm, c->shard_n, u, c->shard_rank));
  const int P = c->shard_n;
  c->xch_cap = c->mwalk;             // a rank can neither spawn nor hold more than MWALK walkers
  HIPCHK(hipMalloc(&c->d_xg, (c->n_imp + 1) * 8));
  HIPCHK(hipMalloc(&c->d_send, c->xch_cap * 32)); HIPCHK(hipMalloc(&c->d_recv, c->xch_cap * 32));
  HIPCHK(hipMalloc(&c->d_cnt_mine, P * 4)); HIPCHK(hipMalloc(&c->d_cnt_all, (size_t)P * P * 4));
  const size_t cnt_bytes = ((size_t)P * P + 32) * 4 + 16;
  HIPCHK(hipHostMalloc(&c->h_cnt_all, cnt_bytes, hipHostMallocMapped));     // the P x P send counts + a sequence word: written by the GPU
  memset(c->h_cnt_all, 0, cnt_bytes);
  HIPCHK(hipHostGetDevicePointer((void **)&c->d_cnt_mail, c->h_cnt_all, 0));
  c->cntall_seq = 0;
  // a second communicator for the all-reduce of the deterministic weights, so that it can run on the side
  // stream beside the spawn path (SQMC_SHARD_OVERLAP=0 keeps everything on one stream and one communicator)
  c->comm2 = nullptr;
  const char *ov = getenv("SQMC_SHARD_OVERLAP");
  if (g_rccl.CommSplit && !(ov && ov[0] == '0')) {
    if (g_rccl.CommSplit(c->comm, 0, c->shard_rank, &c->comm2, nullptr) != ncclSuccess) c->comm2 = nullptr;
  }
  return SQMC_OK;
}

// the P x P counts of the all-gather, posted to pinned host memory (counts, system fence, sequence word)
__global__ void k_post_counts(const u32 *__restrict__ cnt, int n, u32 *mail, u64 seq) {
  for (int k = threadIdx.x; k < n; k += blockDim.x) mail[k] = cnt[k];
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *(volatile u64 *)(mail + ((n + 15) / 16) * 16) = seq; }
}

// One sharded MC step with the exchanges inside: out[0..6] are the global sums, out[7..15] local.
int sqmc_gpu_shard_step(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double out[16]) {
  if (!c || !sp || !out) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (!c->comm) return fail(SQMC_ERR_BAD_ARG, "sqmc_gpu_comm_init not called");
  hipStream_t st = c->st;
  const int P = c->shard_n, me = c->shard_rank;
  int64_t nch = 0;
  const bool side = (c->comm2 != nullptr);
  int r = shard_begin_impl(c, sp, c->d_xg, &nch, false, side);
  if (r) return r;
  {   // deterministic projection: all-reduce of the weights, then the rows this rank owns
    hipStream_t sx = side ? c->st2 : st;
    if (c->n_imp > 0) NCCLCHK(g_rccl.AllReduce(c->d_xg, c->d_xg, (size_t)c->n_imp, ncclDouble, ncclSum, side ? c->comm2 : c->comm, sx));
    if (side) {
      if (c->n_imp_local > 0)
        hipLaunchKernelGGL(k_prj_apply_rows, dim3(nblk(c->n_imp_local, TPB / 64)), dim3(TPB), 0, sx, c->d_prj_ptr, c->d_prj_col, c->d_prj_val, c->d_xg,
                           c->d_loc_imp, c->d_grow, c->w.wt, c->n_imp_local, sp->e_trial, sp->tau);
      HIPCHK(hipEventRecord(c->e_join, sx));
    }
  }
  // bucket + pack without a host round trip: every child is packed in destination order (the
  // ones that made no walker sort last and are never sent), the counts stay on the device and go
  // straight into the all-gather that tells every rank who sends how much to whom
  u32 *order;
  r = shard_bucket(c, sp, c->d_xg, &order, !side); if (r) return r;
  const long long nch_l = c->shard_nch;
  if (nch_l > c->xch_cap) return fail(SQMC_ERR_SPAWN_OVERFLOW, "send buffer too small for this step's spawns");
  if (nch_l > 0) {
    hipLaunchKernelGGL(k_pack_send, dim3(nblk(nch_l)), dim3(TPB), 0, st, c->w, order, c->d_send, c->shard_n0, nch_l);
  } else HIPCHK(hipMemsetAsync(c->d_cnt_mine, 0, P * 4, st));
  // the per-destination counts are the digit totals the bucketing pass left in d_rowtot: they go into the all-gather from there
  NCCLCHK(g_rccl.AllGather(nch_l > 0 ? c->d_rowtot : c->d_cnt_mine, c->d_cnt_all, (size_t)P, ncclUint32, c->comm, st));
  {
    const u64 qs = ++c->cntall_seq;
    hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(256), 0, st, (const u32 *)c->d_cnt_all, P * P, c->d_cnt_mail, qs);
    HIPCHK(hipGetLastError());
    volatile u64 *flag = (volatile u64 *)(c->h_cnt_all + ((P * P + 15) / 16) * 16);
    int wr = wait_mail(flag, qs, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) HIPCHK(hipMemcpy(c->h_cnt_all, c->d_cnt_all, (size_t)P * P * 4, hipMemcpyDeviceToHost));
  }
  long long scnt[256];
  for (int q = 0; q < P; q++) scnt[q] = c->h_cnt_all[(size_t)me * P + q];
  long long soff[257], roff[257]; soff[0] = roff[0] = 0;
  for (int q = 0; q < P; q++) { soff[q + 1] = soff[q] + scnt[q]; roff[q + 1] = roff[q] + c->h_cnt_all[(size_t)q * P + me]; }
  const long long n_recv = roff[P];
  if (n_recv > c->xch_cap || c->shard_n0 + n_recv > c->mwalk) return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  // the walkers themselves, HBM to HBM; records from rank q land at roff[q] (rank order,
  // creation order inside a rank: the order the merge rules see, do_walk.f90:5866-6083)
  NCCLCHK(g_rccl.GroupStart());
  for (int q = 0; q < P; q++) {
    if (q == me) continue;
    if (scnt[q] > 0) NCCLCHK(g_rccl.Send(c->d_send + 4 * soff[q], (size_t)(4 * scnt[q]), ncclUint64, q, c->comm, st));
    const long long rc = roff[q + 1] - roff[q];
    if (rc > 0) NCCLCHK(g_rccl.Recv(c->d_recv + 4 * roff[q], (size_t)(4 * rc), ncclUint64, q, c->comm, st));
  }
  NCCLCHK(g_rccl.GroupEnd());
  return shard_finish_impl(c, sp, (const uint64_t *)c->d_recv, n_recv, out, side, c->d_send + 4 * soff[me], roff[me], roff[me] + scnt[me]);
}

int sqmc_gpu_shard_run(sqmc_gpu_ctx *c, sqmc_popctl *pc, int64_t nsteps, double *stats, double totals[16]) {
  return run_steps(c, pc, nsteps, stats, totals, sqmc_gpu_shard_step);
}

// ---------------------------------------------------------------- batch doors
int sqmc_gpu_hamiltonian_batch(sqmc_gpu_ctx *c, int64_t n, const uint64_t *iu, const uint64_t *id, const uint64_t *ju, const uint64_t *jd, double *h) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  if (n <= 0) return SQMC_OK;
  u64 *d[4]; double *dh;
  const uint64_t *src[4] = {iu, id, ju, jd};
  for (int k = 0; k < 4; k++) { HIPCHK(hipMalloc(&d[k], n * 8)); HIPCHK(hipMemcpy(d[k], src[k], n * 8, hipMemcpyHostToDevice)); }
  HIPCHK(hipMalloc(&dh, n * 8));
  hipLaunchKernelGGL(k_ham_batch, dim3(nblk(n)), dim3(TPB), 0, c->st, c->dev, d[0], d[1], d[2], d[3], dh, (long long)n);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipMemcpy(h, dh, n * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; k++) hipFree(d[k]);
  hipFree(dh);
  return SQMC_OK;
}

int sqmc_gpu_hamiltonian_chem_batch(sqmc_gpu_ctx *c, int64_t n, const uint64_t *iu, const uint64_t *id, const uint64_t *ju, const uint64_t *jd, double *h) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  if (n <= 0) return SQMC_OK;
  u64 *d[4]; double *dh;
  const uint64_t *src[4] = {iu, id, ju, jd};
  for (int k = 0; k < 4; k++) { HIPCHK(hipMalloc(&d[k], n * 8)); HIPCHK(hipMemcpy(d[k], src[k], n * 8, hipMemcpyHostToDevice)); }
  HIPCHK(hipMalloc(&dh, n * 8));
  hipLaunchKernelGGL(k_ham_chem_batch, dim3(nblk(n)), dim3(TPB), 0, c->st, c->dev, d[0], d[1], d[2], d[3], dh, (long long)n);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipMemcpy(h, dh, n * 8, hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; k++) hipFree(d[k]);
  hipFree(dh);
  return SQMC_OK;
}

int sqmc_gpu_build_sparse_ham(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, int64_t *out_nnz,
                              int64_t **out_row_counts, int64_t **out_indices, double **out_values) {
  if (!c || !out_nnz || !out_row_counts || !out_indices || !out_values || n <= 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  for (long long i = 1; i < n; i++)
    if (!(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "determinant list must be strictly sorted by (up,dn)");
  hipStream_t st = c->st;
  u64 *du, *dd, *dcnt, *doff, *dtot, *dts;
  HIPCHK(hipMalloc(&du, n * 8)); HIPCHK(hipMalloc(&dd, n * 8)); HIPCHK(hipMalloc(&dcnt, n * 8)); HIPCHK(hipMalloc(&doff, n * 8)); HIPCHK(hipMalloc(&dtot, 8));
  long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
  HIPCHK(hipMalloc(&dts, (tiles + 1) * 8));
  HIPCHK(hipMemcpy(du, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd, dn, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_build_ham, dim3(nblk(n)), dim3(TPB), 0, st, c->dev, du, dd, (long long)n, 0, dcnt, doff, (long long *)nullptr, (double *)nullptr);
  ScanWork sw; sw.state = dts; sw.ticket = (u32 *)(dts + tiles); sw.cap_tiles = tiles; sw.self_clear = true;
  device_excl_scan_u64(dcnt, doff, n, dtot, sw, st);
  u64 total = 0;
  HIPCHK(hipMemcpyAsync(&total, dtot, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
  long long *didx; double *dval;
  HIPCHK(hipMalloc(&didx, (total + 1) * 8)); HIPCHK(hipMalloc(&dval, (total + 1) * 8));
  hipLaunchKernelGGL(k_build_ham, dim3(nblk(n)), dim3(TPB), 0, st, c->dev, du, dd, (long long)n, 1, dcnt, doff, didx, dval);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
  int64_t *rc = (int64_t *)malloc(n * 8), *ix = (int64_t *)malloc((total + 1) * 8); double *vl = (double *)malloc((total + 1) * 8);
  HIPCHK(hipMemcpy(rc, dcnt, n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(ix, didx, total * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(vl, dval, total * 8, hipMemcpyDeviceToHost));
  *out_nnz = (int64_t)total; *out_row_counts = rc; *out_indices = ix; *out_values = vl;
  void *fr[] = {du, dd, dcnt, doff, dtot, dts, didx, dval};
  for (void *q : fr) hipFree(q);
  return SQMC_OK;
}


// The sparse Hamiltonian of a sorted determinant list built on the device AND left there as a
// matvec plan: what generate_sparse_ham_chem_upper_triangular + davidson_sparse's matvec need,
// without the matrix crossing PCIe twice or being expanded on the host.  diag[n] (the Davidson
// preconditioner) and the number of stored (upper-triangular) nonzeros come back to the host.
int sqmc_gpu_build_spmv_plan(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, sqmc_spmv_plan **plan, double *diag, int64_t *out_nnz) {
  if (!c || !plan || !diag || !out_nnz || n <= 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  for (long long i = 1; i < n; i++)
    if (!(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "determinant list must be strictly sorted by (up,dn)");
  hipStream_t st = c->st;
  u64 *du, *dd, *dcnt, *doff, *dtot, *dts;
  HIPCHK(hipMalloc(&du, n * 8)); HIPCHK(hipMalloc(&dd, n * 8)); HIPCHK(hipMalloc(&dcnt, n * 8)); HIPCHK(hipMalloc(&doff, n * 8)); HIPCHK(hipMalloc(&dtot, 16));
  long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
  HIPCHK(hipMalloc(&dts, (tiles + 1) * 8));
  HIPCHK(hipMemcpy(du, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd, dn, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_build_ham, dim3(nblk(n)), dim3(TPB), 0, st, c->dev, du, dd, (long long)n, 0, dcnt, doff, (long long *)nullptr, (double *)nullptr);
  ScanWork sw; sw.state = dts; sw.ticket = (u32 *)(dts + tiles); sw.cap_tiles = tiles; sw.self_clear = true;
  device_excl_scan_u64(dcnt, doff, n, dtot, sw, st);
  u64 total = 0;
  HIPCHK(hipMemcpyAsync(&total, dtot, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
  const long long n_strict = (long long)total - n, nnz_full = (long long)total + n_strict;
  if (nnz_full >= (1ll << 31)) return fail(SQMC_ERR_UNSUPPORTED, "more than 2^31 expanded nonzeros");
  long long *didx; double *dval;
  HIPCHK(hipMalloc(&didx, (total + 1) * 8)); HIPCHK(hipMalloc(&dval, (total + 1) * 8));
  hipLaunchKernelGGL(k_build_ham, dim3(nblk(n)), dim3(TPB), 0, st, c->dev, du, dd, (long long)n, 1, dcnt, doff, didx, dval);
  // transpose bookkeeping
  u64 *dkeys, *dkeys_alt, *drowlen, *dptr64, *dcolc, *dcolstart; u32 *dvals, *dvals_alt, *drowof, *dcolcount, *dhist, *drowtot;
  const long long ntile_sort = (long long)((total + RS_TILE - 1) / RS_TILE);
  HIPCHK(hipMalloc(&dkeys, total * 8)); HIPCHK(hipMalloc(&dkeys_alt, total * 8)); HIPCHK(hipMalloc(&dvals, total * 4)); HIPCHK(hipMalloc(&dvals_alt, total * 4));
  HIPCHK(hipMalloc(&drowof, total * 4)); HIPCHK(hipMalloc(&dcolcount, (n + 1) * 4)); HIPCHK(hipMalloc(&drowlen, n * 8)); HIPCHK(hipMalloc(&dptr64, n * 8));
  HIPCHK(hipMalloc(&dcolc, n * 8)); HIPCHK(hipMalloc(&dcolstart, n * 8));
  HIPCHK(hipMalloc(&dhist, (size_t)RS_MAX_RADIX * (ntile_sort + 1) * 4)); HIPCHK(hipMalloc(&drowtot, RS_MAX_RADIX * 4));
  HIPCHK(hipMemsetAsync(dcolcount, 0, (n + 1) * 4, st));
  hipLaunchKernelGGL(k_csr_keys, dim3(nblk(n)), dim3(TPB), 0, st, dcnt, doff, didx, dkeys, dvals, drowof, dcolcount, (long long)n);
  hipLaunchKernelGGL(k_csr_rowlen, dim3(nblk(n)), dim3(TPB), 0, st, dcnt, dcolcount, drowlen, dcolc, (long long)n);
  device_excl_scan_u64(drowlen, dptr64, n, dtot, sw, st);
  device_excl_scan_u64(dcolc, dcolstart, n, dtot + 1, sw, st);
  int kb = 1; while ((1ll << kb) <= n) kb++;                      // keys are column indices 0..n (n = the diagonal marker)
  SortWork so; so.k_alt = dkeys_alt; so.v_alt = dvals_alt; so.hist = dhist; so.rowtot = drowtot; so.cap = (long long)total;
  u64 *sk = dkeys; u32 *sv = dvals;
  device_radix_sort(sk, sv, (long long)total, kb, so, st);
  sqmc_spmv_plan *p = new sqmc_spmv_plan(); p->n = n; p->nnz_full = nnz_full;
  HIPCHK(hipStreamCreate(&p->st));
  HIPCHK(hipMalloc(&p->d_ptr, (n + 1) * 4)); HIPCHK(hipMalloc(&p->d_col, (nnz_full + 1) * 4)); HIPCHK(hipMalloc(&p->d_val, (nnz_full + 1) * 8));
  HIPCHK(hipMalloc(&p->d_x, n * 8)); HIPCHK(hipMalloc(&p->d_y, n * 8));
  double *ddiag; HIPCHK(hipMalloc(&ddiag, n * 8));
  hipLaunchKernelGGL(k_csr_fill_stored, dim3(nblk(n + 1)), dim3(TPB), 0, st, dcnt, doff, didx, dval, dptr64, p->d_ptr, p->d_col, p->d_val, ddiag, (long long)n, nnz_full);
  if (n_strict > 0)
    hipLaunchKernelGGL(k_csr_fill_transposed, dim3(nblk(n_strict)), dim3(TPB), 0, st, sk, sv, drowof, dval, dcnt, dptr64, dcolstart, p->d_col, p->d_val, (long long)n, n_strict);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(diag, ddiag, n * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
  *out_nnz = (int64_t)total; *plan = p;
  void *fr[] = {du, dd, dcnt, doff, dtot, dts, didx, dval, dkeys, dkeys_alt, dvals, dvals_alt, drowof, dcolcount, drowlen, dptr64, dcolc, dcolstart, dhist, drowtot, ddiag};
  for (void *q : fr) hipFree(q);
  return SQMC_OK;
}

int sqmc_gpu_propose_batch(sqmc_gpu_ctx *c, int64_t n, double tau, const uint64_t *up, const uint64_t *dn, const int32_t *seeds,
                           uint64_t *ju, uint64_t *jd, double *wj, int32_t *seeds_after) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  if (n <= 0) return SQMC_OK;
  std::vector<u64> s(n);
  for (long long i = 0; i < n; i++) s[i] = (((u64)seeds[4 * i] << 36) + ((u64)seeds[4 * i + 1] << 24) + ((u64)seeds[4 * i + 2] << 12) + (u64)seeds[4 * i + 3]) & SQ_MASK48;
  u64 *du, *dd, *ds, *dju, *djd, *dso; double *dw;
  HIPCHK(hipMalloc(&du, n * 8)); HIPCHK(hipMalloc(&dd, n * 8)); HIPCHK(hipMalloc(&ds, n * 8)); HIPCHK(hipMalloc(&dju, n * 8));
  HIPCHK(hipMalloc(&djd, n * 8)); HIPCHK(hipMalloc(&dso, n * 8)); HIPCHK(hipMalloc(&dw, n * 8));
  HIPCHK(hipMemcpy(du, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd, dn, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds, s.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_propose_batch, dim3(nblk(n)), dim3(TPB), 0, c->st, c->dev, du, dd, ds, dju, djd, dw, dso, (long long)n, tau);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(c->st));
  HIPCHK(hipMemcpy(ju, dju, n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(jd, djd, n * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(wj, dw, n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(s.data(), dso, n * 8, hipMemcpyDeviceToHost));
  for (long long i = 0; i < n; i++) { u64 x = s[i]; seeds_after[4 * i] = (int)((x >> 36) & 4095); seeds_after[4 * i + 1] = (int)((x >> 24) & 4095);
    seeds_after[4 * i + 2] = (int)((x >> 12) & 4095); seeds_after[4 * i + 3] = (int)(x & 4095); }
  hipFree(du); hipFree(dd); hipFree(ds); hipFree(dju); hipFree(djd); hipFree(dso); hipFree(dw);
  return SQMC_OK;
}

int sqmc_gpu_hci_connections(sqmc_gpu_ctx *c, int64_t n_ref, const uint64_t *ref_up, const uint64_t *ref_dn, const double *coeffs, double eps,
                             int diag_mode, int64_t *out_n, uint64_t **out_up, uint64_t **out_dn, double **out_num, double **out_den) {
  return sqmc_gpu_hci_connections_slice(c, n_ref, ref_up, ref_dn, coeffs, eps, diag_mode, 0, 1, out_n, out_up, out_dn, out_num, out_den);
}

int sqmc_gpu_hci_connections_slice(sqmc_gpu_ctx *c, int64_t n_ref, const uint64_t *ref_up, const uint64_t *ref_dn, const double *coeffs, double eps,
                                   int diag_mode, int32_t slice, int32_t n_slices, int64_t *out_n, uint64_t **out_up, uint64_t **out_dn,
                                   double **out_num, double **out_den) {
  if (!c || !out_n) return fail(SQMC_ERR_BAD_ARG, "null argument");
  if (n_slices < 1 || slice < 0 || slice >= n_slices) return fail(SQMC_ERR_BAD_ARG, "slice out of range");
  u64 key_lo = 0, key_hi = ~0ull;
  if (n_slices > 1) {                 // equal parts of the key range [0, invalid_key]
    const long double span = ((long double)c->invalid_key + 1.0L) / (long double)n_slices;
    key_lo = (u64)(span * slice); key_hi = (slice == n_slices - 1) ? (~0ull - 1ull) : (u64)(span * (slice + 1));
  }
  if (c->htab.sys_type == 0 && !c->dev.hb_r) return fail(SQMC_ERR_BAD_ARG, "heat-bath tables not set (sqmc_gpu_set_hb_tables)");
  if (c->htab.sys_type == 2) return fail(SQMC_ERR_UNSUPPORTED, "connection generation for hubbard2 is the host's 4*nelec neighbour list (find_connected_dets_hubbard); no heat-bath screening applies");
  if (c->htab.sys_type == 1 && c->htab.heg_nmax > 4) return fail(SQMC_ERR_UNSUPPORTED, "HEG connections: plane-wave index beyond +-4");
  *out_n = 0;
  if (n_ref <= 0) return SQMC_OK;
  hipStream_t st = c->st;
  u64 *dru, *drd, *dcnt, *doff, *dtot, *dts; double *dco;
  HIPCHK(hipMalloc(&dru, n_ref * 8)); HIPCHK(hipMalloc(&drd, n_ref * 8)); HIPCHK(hipMalloc(&dco, n_ref * 8));
  HIPCHK(hipMalloc(&dcnt, n_ref * 8)); HIPCHK(hipMalloc(&doff, n_ref * 8)); HIPCHK(hipMalloc(&dtot, 8));
  long long tiles = (n_ref + SCAN_TILE - 1) / SCAN_TILE + 1;
  HIPCHK(hipMalloc(&dts, (tiles + 1) * 8));
  HIPCHK(hipMemcpy(dru, ref_up, n_ref * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(drd, ref_dn, n_ref * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dco, coeffs, n_ref * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_hci_gen, dim3(nblk(n_ref)), dim3(TPB), 0, st, c->dev, dru, drd, dco, eps, diag_mode, (long long)n_ref, 0, dcnt, doff,
                     (u64 *)nullptr, (u64 *)nullptr, (double *)nullptr, (double *)nullptr, key_lo, key_hi);
  ScanWork sw; sw.state = dts; sw.ticket = (u32 *)(dts + tiles); sw.cap_tiles = tiles; sw.self_clear = true;
  device_excl_scan_u64(dcnt, doff, n_ref, dtot, sw, st);
  u64 total = 0;
  HIPCHK(hipMemcpyAsync(&total, dtot, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
  if (total >= (1ull << 31)) return fail(SQMC_ERR_UNSUPPORTED, "more than 2^31 connections in one call: use more slices (sqmc_gpu_hci_connections_slice)");
  if (total == 0) { void *fz[] = {dru, drd, dco, dcnt, doff, dtot, dts}; for (void *q : fz) hipFree(q); return SQMC_OK; }
  const long long T = (long long)total;
  u64 *du, *dd, *keys, *kalt, *flags, *pos, *ou, *od, *dts2, *dtot2; u32 *vals, *valt, *hist, *rowtot; double *dnum, *dden, *onum, *oden;
  HIPCHK(hipMalloc(&du, T * 8)); HIPCHK(hipMalloc(&dd, T * 8)); HIPCHK(hipMalloc(&dnum, T * 8)); HIPCHK(hipMalloc(&dden, T * 8));
  hipLaunchKernelGGL(k_hci_gen, dim3(nblk(n_ref)), dim3(TPB), 0, st, c->dev, dru, drd, dco, eps, diag_mode, (long long)n_ref, 1, dcnt, doff, du, dd, dnum, dden, key_lo, key_hi);
  if (diag_mode == 2) {               // the unmerged list, in generation order
    HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
    *out_n = T;
    uint64_t *hu = (uint64_t *)malloc(T * 8 + 8), *hd = (uint64_t *)malloc(T * 8 + 8);
    double *hn = (double *)malloc(T * 8 + 8), *hden = (double *)malloc(T * 8 + 8);
    HIPCHK(hipMemcpy(hu, du, T * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hd, dd, T * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hn, dnum, T * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hden, dden, T * 8, hipMemcpyDeviceToHost));
    if (out_up) *out_up = hu; else free(hu);
    if (out_dn) *out_dn = hd; else free(hd);
    if (out_num) *out_num = hn; else free(hn);
    if (out_den) *out_den = hden; else free(hden);
    void *fr0[] = {dru, drd, dco, dcnt, doff, dtot, dts, du, dd, dnum, dden};
    for (void *q : fr0) hipFree(q);
    return SQMC_OK;
  }
  HIPCHK(hipMalloc(&keys, T * 8)); HIPCHK(hipMalloc(&kalt, T * 8)); HIPCHK(hipMalloc(&vals, T * 4)); HIPCHK(hipMalloc(&valt, T * 4));
  long long ntiles = (T + RS_TILE - 1) / RS_TILE;
  HIPCHK(hipMalloc(&hist, ntiles * RS_MAX_RADIX * 4)); HIPCHK(hipMalloc(&rowtot, RS_MAX_RADIX * 4));
  hipLaunchKernelGGL(k_main_keys, dim3(nblk(T)), dim3(TPB), 0, st, c->dev, du, dd, keys, vals, T, 0);
  SortWork so; so.k_alt = kalt; so.v_alt = valt; so.hist = hist; so.rowtot = rowtot; so.cap = T;
  u64 *skey = keys; u32 *perm = vals;
  device_radix_sort(skey, perm, T, c->key_bits, so, st);
  HIPCHK(hipMalloc(&flags, T * 8)); HIPCHK(hipMalloc(&pos, T * 8));
  long long tiles2 = (T + SCAN_TILE - 1) / SCAN_TILE + 1;
  HIPCHK(hipMalloc(&dts2, (tiles2 + 1) * 8)); HIPCHK(hipMalloc(&dtot2, 8));
  hipLaunchKernelGGL(k_hci_heads, dim3(nblk(T)), dim3(TPB), 0, st, skey, flags, T);
  ScanWork sw2; sw2.state = dts2; sw2.ticket = (u32 *)(dts2 + tiles2); sw2.cap_tiles = tiles2; sw2.self_clear = true;
  device_excl_scan_u64(flags, pos, T, dtot2, sw2, st);
  u64 nuniq = 0;
  HIPCHK(hipMemcpyAsync(&nuniq, dtot2, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
  const long long U = (long long)nuniq;
  HIPCHK(hipMalloc(&ou, U * 8)); HIPCHK(hipMalloc(&od, U * 8)); HIPCHK(hipMalloc(&onum, U * 8)); HIPCHK(hipMalloc(&oden, U * 8));
  hipLaunchKernelGGL(k_hci_dedup, dim3(nblk(T)), dim3(TPB), 0, st, skey, perm, flags, pos, du, dd, dnum, dden, ou, od, onum, oden, T);
  HIPCHK(hipGetLastError()); HIPCHK(hipStreamSynchronize(st));
  *out_n = U;
  uint64_t *hu = (uint64_t *)malloc(U * 8 + 8), *hd = (uint64_t *)malloc(U * 8 + 8);
  double *hn = (double *)malloc(U * 8 + 8), *hden = (double *)malloc(U * 8 + 8);
  HIPCHK(hipMemcpy(hu, ou, U * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hd, od, U * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hn, onum, U * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hden, oden, U * 8, hipMemcpyDeviceToHost));
  if (out_up) *out_up = hu; else free(hu);
  if (out_dn) *out_dn = hd; else free(hd);
  if (out_num) *out_num = hn; else free(hn);
  if (out_den) *out_den = hden; else free(hden);
  void *fr[] = {dru, drd, dco, dcnt, doff, dtot, dts, du, dd, dnum, dden, keys, kalt, vals, valt, hist, rowtot, flags, pos, dts2, dtot2, ou, od, onum, oden};
  for (void *q : fr) hipFree(q);
  return SQMC_OK;
}

// ----------------------------------------------------------------------- SpMV
int sqmc_gpu_spmv_prepare(int64_t n, const int64_t *rc, const int64_t *idx, const double *val, sqmc_spmv_plan **plan) {
  if (!rc || !idx || !val || !plan || n <= 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SQMC_ERR_HIP, "no HIP device: libsqmc_gpu has no CPU fallback");
  long long nnz = 0; for (long long i = 0; i < n; i++) nnz += rc[i];
  for (long long k = 0; k < nnz; k++) if (idx[k] < 1 || idx[k] > n) return fail(SQMC_ERR_BAD_ARG, "column index out of range");
  if (2 * nnz >= (1ll << 31)) return fail(SQMC_ERR_UNSUPPORTED, "more than 2^31 expanded nonzeros");
  std::vector<int> ptr, col; std::vector<double> v;
  expand_full_csr(n, rc, idx, val, ptr, col, v);
  sqmc_spmv_plan *p = new sqmc_spmv_plan(); p->n = n; p->nnz_full = (long long)col.size();
  HIPCHK(hipStreamCreate(&p->st));
  HIPCHK(hipMalloc(&p->d_ptr, (n + 1) * 4)); HIPCHK(hipMalloc(&p->d_col, (col.size() + 1) * 4)); HIPCHK(hipMalloc(&p->d_val, (v.size() + 1) * 8));
  HIPCHK(hipMalloc(&p->d_x, n * 8)); HIPCHK(hipMalloc(&p->d_y, n * 8));
  HIPCHK(hipMemcpy(p->d_ptr, ptr.data(), (n + 1) * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(p->d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(p->d_val, v.data(), v.size() * 8, hipMemcpyHostToDevice));
  *plan = p;
  return SQMC_OK;
}
int sqmc_gpu_spmv_apply(sqmc_spmv_plan *p, const double *x, double *y, int on_device) {
  if (!p || !x || !y) return fail(SQMC_ERR_BAD_ARG, "null");
  const double *dx = x; double *dy = y;
  if (!on_device) { HIPCHK(hipMemcpyAsync(p->d_x, x, p->n * 8, hipMemcpyHostToDevice, p->st)); dx = p->d_x; dy = p->d_y; }
  hipLaunchKernelGGL(k_spmv_wave, dim3((unsigned)((p->n + SPMV_ROWS_PER_BLOCK - 1) / SPMV_ROWS_PER_BLOCK)), dim3(64 * SPMV_ROWS_PER_BLOCK), 0, p->st,
                     p->d_ptr, p->d_col, p->d_val, dx, dy, p->n);
  HIPCHK(hipGetLastError());
  if (!on_device) { HIPCHK(hipMemcpyAsync(y, p->d_y, p->n * 8, hipMemcpyDeviceToHost, p->st)); }
  HIPCHK(hipStreamSynchronize(p->st));
  return SQMC_OK;
}
int sqmc_gpu_spmv_free(sqmc_spmv_plan *p) {
  if (!p) return SQMC_OK;
  hipFree(p->d_ptr); hipFree(p->d_col); hipFree(p->d_val); hipFree(p->d_x); hipFree(p->d_y); hipStreamDestroy(p->st); delete p;
  return SQMC_OK;
}
int sqmc_gpu_spmv_sym_upper(int64_t n, const int64_t *rc, const int64_t *idx, const double *val, const double *x, double *y) {
  sqmc_spmv_plan *p = nullptr;
  int r = sqmc_gpu_spmv_prepare(n, rc, idx, val, &p);
  if (r) return r;
  r = sqmc_gpu_spmv_apply(p, x, y, 0);
  sqmc_gpu_spmv_free(p);
  return r;
}

}  // extern "C"
