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
#include <time.h>
#include <vector>
#include <string>
#include <algorithm>
#include "../../include/sqmc_gpu.h"
#include "chem_device.h"
#include "heatbath_device.h"
#include "scan_sort.h"
#include "bucket_partition.h"
#include "hii_group.h"

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
  u32 *irk;                // for a deterministic-space walker (imp_distance 0): its row in the projector = its rank among them (written by the bucket tail)
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
  HIPCHK(hipMalloc(&a.irk, n * 4));
  return 0;
}
static void free_walk(WalkArr &a) {
  hipFree(a.sp); hipFree(a.up); hipFree(a.dn); hipFree(a.wt); hipFree(a.flg);
  hipFree(a.me); hipFree(a.en); hipFree(a.ed); hipFree(a.irk);
}

struct StepP {       // device copy of sqmc_step_params + derived values
  double tau, e_trial, rfi, r_init, min_wt, cutoff;
  int ipow, imind, cti, semi, reached;
  int nimp_cap;            // entries of the loc_imp array (set by step_tail): spawn records with made-up flags cannot push an index past it
  u64 koff = 0; long long nct = 0; // hf_to_psit (0: off): offset of the sort key of a determinant outside C(T), and the length of the C(T) segment
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
  int retry;               // the bucket tail met a bucket that does not fit its block: nothing of this step's tail counts, the host re-runs it (sticky: only the host clears it)
  double stats[16];
  double red[8];           // sharded steps: the all-reduced seven global sums and the collective status (256^code summed over ranks, + 1 per rank whose bucket tail gave up)
  double redl[8];          // this rank's contribution to red (kept: a step whose tail some rank has to re-run is all-reduced twice)
  unsigned int bk_fill, pad2;   // bucket tail: fill of the fullest bucket this step, per mille of the LDS caps
};
// status of a step summed over ranks as 256^code: the highest code any rank raised (codes 1..5, at most 255 ranks: exact in a double)
__host__ __device__ __forceinline__ double err_encode(int code) { double v = 0.0; if (code > 0) { v = 1.0; for (int k = 0; k < code; k++) v *= 256.0; } return v; }
__host__ __device__ __forceinline__ int err_decode(double v) { int code = 0; double lim = 256.0; for (int k = 1; k <= 5; k++) { if (v >= lim) code = k; lim *= 256.0; } return code; }

// Mailbox in pinned host memory that the GPU writes directly (no copy kernel, no interrupt): the
// child count as soon as k_spawn starts, the step's sums at the end of k_finish.  The host spins
// on the sequence words.  Data first, system-scope fence, then the sequence word.
struct HostMail {
  volatile u64 seq; u64 tot2; long long err; double stats[16];
  volatile u64 cnt_seq; u64 n_children;
  u64 retry, bk_fill;      // bucket tail: see DevScalars
};

#define NTIMERS 32
// what the block that finishes a step needs (finish_all, walk_kernels.h)
struct FinArgs {
  const double *partials; int nblocks; const double *wabs_part; int nwabs; int mode; u64 *scan_state; u32 *scan_ticket; int n_scan_words;
  HostMail *mail; u64 seq; u64 *fstate; u32 *fticket; long long cap_ftiles; int n_ftiles; int on;
  int n_tickets;      // scan tickets behind scan_ticket to reset with the n_scan_words state words (3: all of them)
  long long n_children;   // >= 0: the step's child count from the host (a finish that rides on the NEXT step's scan must not read the scalar that scan writes)
  const double *red;      // sharded steps: where the all-reduced sums lie when they travelled behind the deterministic weights (null: DevScalars::red)
  long long expect_nimp;  // >= 0: deterministic-space walkers this rank must still hold; anything else raises SQMC_ERR_IMP_BROKEN on the device (sharded steps: the status is all-reduced)
  const double *partials2; int nblocks2;      // hf_to_psit: the sums over the C(T) segment (k_psit_finish), added to the tiles' partials
};

// device tables of the hf_to_psit step variant (psit_kernels.h)
struct PsitArgs {
  long long n_ct, n_psit, n_imp;
  const int *loc_psit;        // [n_psit] slot of dets_psi_t(k) (my_locations_of_psit, do_walk.f90:1849-1886); loc_psit[0] = 0
  const double *cdet;         // [n_psit] cdet_psi_t in label order
  const double *diag;         // [n_ct]   diag_elems (do_walk.f90:1091-1116)
  const int *psit_of;         // [n_ct]   k with loc_psit[k] = slot, or -1
  const int *imp_of;          // [n_ct]   row of the slot in the deterministic-space matrix, or -1
  const double *cnum, *cden;  // psi_t_connected_e_loc_num / _den
  double *dw_ct, *dw_ps, *dw_imp;     // deltaw(n_imp+1 : n_imp+n_ct), deltaw(n_imp+n_ct+1 : ...), deltaw(1 : n_imp)
  double *p2;                 // first-row sum over C(T): one partial per 4096 terms (two tree levels)
  int n_perm;                 // n_permanent_initiator (0 or 1: only the first state can be one, do_walk.f90:1276-1292)
  int seq;
};

// staging buffer of the two-kernel annihilation of long lists (walk_kernels.h, k_anneal<., 0, 1>): the kept walkers of every tile, compacted
// inside the tile, and the tiles' counts (kept | deterministic-space << 32; children)
struct AnnealStage { u64 *up, *dn, *key; double *wt, *me, *en, *ed; u32 *flg, *nc, *lch, *ldet; u64 *cnt_a, *cnt_b, *off_a, *off_b; };
struct HbHost;            // heatbath_setup.inc: host copies of the efficient heat-bath tables the library built itself
struct sqmc_gpu_ctx {
  hipStream_t st;
  ChemTab htab; ChemTab *d_tab; double *d_ints; ChemDev dev;
  int *d_hb_r, *d_hb_s; double *d_hb_absH; long long *d_pq_ind; int *d_pq_count;
  void *d_hbt[16];                     // device copies of the efficient heat-bath tables (sqmc_gpu_set_heatbath_tables)
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
  int rng_mode; u64 seed64; u64 step_no;      // seed64: sq_mix64 of the 48-bit input seed, the root of every COUNTER stream key
  DevScalars *d_sc; DevScalars *h_sc;   // h_sc pinned
  HostMail *h_mail, *d_mail; u64 mail_seq, cnt_seq;      // the same pinned words seen from host and device
  bool timers_pending;
  double *d_partials; int n_partial_blocks; double *d_wabs_part; u32 *d_done;
  int key_bits; int pack; u64 invalid_key; u64 *d_binom;
  // multi-rank sharding (owner = hash(det) mod shard_n)
  int owner_mode;             // SQMC_OWNER_MIX (default) or SQMC_OWNER_DJB (the reference's get_det_owner, bit for bit)
  int shard_rank, shard_n; int *d_grow; int *d_ginv; long long n_imp_local; long long shard_n0, shard_nch;
  // in-library exchange over RCCL (sqmc_gpu_comm_init): communicator + device staging
  ncclComm_t comm, comm2; double *d_xg; u64 *d_send, *d_recv; long long xch_cap; u32 *d_cnt_mine, *d_cnt_all; u32 *h_cnt_all, *d_cnt_mail;
  u64 cntall_seq;      // comm2: second communicator (ncclCommSplit) for the all-reduce that runs on the side stream
  // timing
  int timing; hipEvent_t ev0[NTIMERS], ev1[NTIMERS]; const char *tname[NTIMERS]; int nt; float tms[NTIMERS];
  double tsum[NTIMERS]; long long tsteps;         // accumulated over the steps since sqmc_gpu_set_timing
  hipStream_t st2; hipEvent_t e_fork, e_join, e_cnt, e_spawned;    // second stream: death/clone beside spawn + sort (long lists: behind k_spawn, beside the sort)
  bool spawned_valid;
  hipStream_t st3; hipEvent_t e_join3;                   // third stream: the deterministic projection (it touches the deterministic-space walkers only, death/clone all the others)
  // pipelined head (sqmc_gpu_run, COUNTER discipline, target population reached): gate + scan + spawn of step n+1 are
  // enqueued right behind k_finish of step n, before the host has read step n's sums
  bool in_run;                // inside sqmc_gpu_run / sqmc_gpu_shard_run (they decide about the pipelined head themselves)
  bool chained_runs;          // sqmc_gpu_set_chained_runs: the last step of a run call enqueues the head of the first step of the next call
  bool pipeline_next, head_ready; StepP head_p; u64 head_cseq; hipEvent_t hev[4];
  bool owner_ready;           // this step's k_spawn already wrote the owner key of every child (sharded steps)
  int scan_flip, scan_used[2];   // gate-fused heads: look-back set of the next head scan, and how many words of each set its last scan may have touched
  bool residents_sorted;      // the walker arrays are known to be in (up, dn) order: true after every finished step and after an upload (which refuses unsorted lists)
  unsigned short *d_segoff; long long segoff_cap;      // bucket tail: group offsets of the partition blocks
  u32 *d_bhint; int pos_flip, scount_pos;      // where each boundary set lay when it was made (3 x BK_MAXB + 1); which of the two d_bpos halves this step writes / the counts were taken with
  u32 *d_jcnt;                       // plain walks: candidates of join_walker2 per chunk of 2048 walkers (k_join_gather)
  u32 *d_hq_cnt, *d_hq_pos; int hii_deferred_B;      // H_ii queues of the buckets (BK_MAXB counts, BK_MAXB x BK_HQ_DEFER positions); > 0: the last tail filled them for the head that follows
  u32 *d_bkb, *d_bpos, *d_bscount;   // bucket boundaries (three sets of BK_MAXB + 1 keys: in use, counted with, being made), their positions in this step's list, the spawns the last bucket tail counted per bucket
  int kb_B[3], scount_B;      // the bucket count each set was made for / the counts were taken with (0: not valid)
  bool shard_x_ready, shard_x_used;   // the deterministic weights of the COMING step are all-reduced already (every rank did that behind this step's sums, whether or not it enqueued a head) / this step is using them
  bool shard_y_used;          // ... and this step is using it
  bool head_sums_ride;        // the all-reduce of the step's sums is the one in front of the pipelined head (they lie behind the weights)
  FinArgs head_fin;           // ... and the final sums themselves are made by the kernel that gathers the weights
  bool shard_y_ok;            // in-library sharded step: the pipelined head all-reduced the deterministic weights and its spare blocks multiplied the projector into them (d_prj_y): the step only adds the last line
  BucketArgs shard_ba;        // sharded steps: the boundaries chosen at the start of the step (their block runs on the side stream)
  int kb_next, scount_buf, head_kb_use;   // set the next bucket head partitions with; set the counts were taken with; set the enqueued head uses (-1: equal-residents boundaries)
  double *d_prj_y; const double *head_prj_x; bool head_y_done;      // A x of the pipelined head's spare k_spawn blocks, the x it used
  double *d_prj_xs[2]; int xs_cur; bool xs_valid;      // snapshots of the deterministic-space weights by row, written by the bucket tail for the NEXT step's projection (two: one is read while the other is written)
  bool side_pending;          // death/clone and the projection of this step have not been launched as kernels: the bucket tail does them itself, any other tail must launch them first
  double slow_us[4]; long long slow_step[4];      // the slowest steps of the last run_steps call (wall clock; host jitter shows up here)
  ActiveSpace as;             // masks of the HCI generator (sqmc_gpu_hci_set_active_space); mode 0 = none
  bool tail_fills_hii;        // the tail that enqueues the next head is a bucket tail: it computes the H_ii of the determinants it creates itself
  bool fork_valid;            // e_fork was recorded behind the last tail (a head behind a bucket tail forks nothing and skips it)
  bool head_hii, head_hii_joined;     // the pipelined head fills the missing H_ii of this step's walkers (joined: inside k_spawn itself, nothing to wait for)
  bool head_offsets_done;     // the tail of the step before wrote this step's child offsets and total (no scan launch in the head)
  bool head_offsets_bucket;   // ... and it was a bucket tail (its head's k_spawn carries the final sums in a spare block; behind a radix tail they are a launch of their own)
  double last_wabs;           // sum |w| after the last step (bounds the next step's child count)
  BucketArgs head_ba; long long last_nall;      // partition already done by the head's k_spawn (B > 0), and the length of the last sorted list (sizes the next one)
  int bk_holdoff;             // steps for which the bucket tail stays off (after a bucket overflowed or came close)
  long long bk_steps, bk_retries;
  u32 *d_bpar; bool head_bpar_ok;      // parent of the first child of every block of 256 children, written beside the child offsets; valid for the head that follows
  AnnealStage stage; void *stage_mem; long long stage_cap;      // long lists: staging buffer of the two-kernel annihilation (k_anneal<., 0, 1> + k_anneal_place), allocated at first use
  double sh_us[4]; long long sh_steps;      // host wall clock of the in-library sharded steps: head, exchange, tail, of which waiting for the GPU's mail (sqmc_gpu_shard_time_split)
  // hf_to_psit (psit_kernels.h)
  HbHost *hb_host;
  long long dbg_n0, dbg_nall;          // sizes of the last step's list in front of the merge (sqmc_gpu_debug_premerge)
  bool psit_on; int base_key_bits; PsitArgs psit; int *d_ps_loc, *d_ps_of, *d_ps_impof; double *d_ps_c, *d_ps_diag, *d_ps_dwct, *d_ps_dwps, *d_ps_dwimp, *d_ps_p2, *d_ps_part, *d_ps_raw;
};
// a head enqueued for a step that is not going to be the next thing that happens (chained runs): forget it
static void abandon_head(sqmc_gpu_ctx *c);
static void psit_off(sqmc_gpu_ctx *c);
static int shard_head_project(sqmc_gpu_ctx *c, bool with_sums, bool empty, const FinArgs *fin = nullptr);      // abi_shard.inc


#include "walk_kernels.h"
#include "psit_kernels.h"
#define SPAWN_LAUNCH(HB_, FUSE_, ...) do { if (HB_) hipLaunchKernelGGL((k_spawn<1, 1>), __VA_ARGS__); else if (FUSE_) hipLaunchKernelGGL((k_spawn<0, 1>), __VA_ARGS__); else hipLaunchKernelGGL((k_spawn<0, 0>), __VA_ARGS__); } while (0)
#define SPAWN_LAUNCH_EXT(HB_, FUSE_, ...) do { if (HB_) hipExtLaunchKernelGGL((k_spawn<1, 1>), __VA_ARGS__); else if (FUSE_) hipExtLaunchKernelGGL((k_spawn<0, 1>), __VA_ARGS__); else hipExtLaunchKernelGGL((k_spawn<0, 0>), __VA_ARGS__); } while (0)
#include "bucket_kernels.h"
#include "door_kernels.h"
#include "hci_kernels.h"
#include "hbuild_kernels.h"
#include "spmv_kernels.h"

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
    for (int a = 0; a < 64; a++) { bn[a * SQ_BINOM_STRIDE] = 1; for (int b = 1; b <= a; b++) bn[a * SQ_BINOM_STRIDE + b] = (b == a) ? 1 : bn[(a - 1) * SQ_BINOM_STRIDE + b - 1] + bn[(a - 1) * SQ_BINOM_STRIDE + b]; }
    auto choose = [&](int n_, int k_) -> long double { long double r = 1; for (int q = 1; q <= k_; q++) r = r * (n_ - k_ + q) / q; return r; };
    long double tot = choose(cfg->norb, cfg->nup) * choose(cfg->norb, cfg->ndn);
    if (tot >= 9.0e18L) { delete c; return fail(SQMC_ERR_UNSUPPORTED, "determinant space needs more than 63 key bits"); }
    u64 nd = (u64)(choose(cfg->norb, cfg->ndn) + 0.5L), total = (u64)(tot + 0.5L);
    int bits = 1; while (bits < 63 && ((1ull << bits) - 1ull) < total) bits++;
    c->key_bits = bits; c->invalid_key = (1ull << bits) - 1ull; c->pack = (bits <= 32 && !getenv("SQMC_FORCE_UNPACKED")) ? 1 : 0;      // SQMC_FORCE_UNPACKED: the two-array key layout of wide keys on a system whose keys would pack (tests)
    HIPCHK(hipMalloc(&c->d_binom, bn.size() * 8));
    HIPCHK(hipMemcpy(c->d_binom, bn.data(), bn.size() * 8, hipMemcpyHostToDevice));
    c->dev.binom = c->d_binom; c->dev.n_dn_strings = nd;
  }
  c->rng_mode = cfg->rng_mode;
  // limbs of the input seed may exceed 12 bits ('(4i4,x,4i4)' reads 4 decimal digits each):
  // rannyu's limb products treat them as coefficients of powers of 2^12, so the state is the SUM
  u64 s48 = (((u64)cfg->irand_seed[0] << 36) + ((u64)cfg->irand_seed[1] << 24) + ((u64)cfg->irand_seed[2] << 12) + (u64)(2 * (cfg->irand_seed[3] / 2) + 1)) & SQ_MASK48;
  c->seed64 = sq_mix64(s48); c->step_no = 0;
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
    HIPCHK(hipMalloc(&c->d_bpar, (M / TPB + 4) * 4)); HIPCHK(hipMemset(c->d_bpar, 0, (M / TPB + 4) * 4));
    HIPCHK(hipMalloc(&c->d_flags, M * 8)); HIPCHK(hipMalloc(&c->d_pos, M * 8));
    HIPCHK(hipMalloc(&c->d_flags2, M * 8)); HIPCHK(hipMalloc(&c->d_pos2, M * 8)); HIPCHK(hipMalloc(&c->d_jcnt, (M / 2048 + 2) * 4));
    c->cap_tiles = (M + SCAN_TILE - 1) / SCAN_TILE + 1;
    HIPCHK(hipMalloc(&c->d_scan_state, 3 * c->cap_tiles * 8)); HIPCHK(hipMalloc(&c->d_scan_ticket, 3 * 4));
    HIPCHK(hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8)); HIPCHK(hipMemset(c->d_scan_ticket, 0, 3 * 4));
    c->cap_ftiles = std::max<long long>(nblk(M) + 1, BK_MAXB + 1);
    c->segoff_cap = ((std::min<long long>(M, 1ll << 20) + BK_T - 1) / BK_T + 1) * (BK_MAXB + 1);
    HIPCHK(hipMalloc(&c->d_segoff, c->segoff_cap * sizeof(unsigned short)));
    HIPCHK(hipMalloc(&c->d_bkb, 3 * (BK_MAXB + 1) * sizeof(u32))); HIPCHK(hipMalloc(&c->d_bpos, 2 * (BK_MAXB + 1) * sizeof(u32))); HIPCHK(hipMalloc(&c->d_bhint, 3 * (BK_MAXB + 1) * sizeof(u32))); HIPCHK(hipMemset(c->d_bhint, 0, 3 * (BK_MAXB + 1) * sizeof(u32))); c->pos_flip = 0; c->scount_pos = 0; HIPCHK(hipMalloc(&c->d_bscount, (BK_MAXB + 1) * sizeof(u32)));
    HIPCHK(hipMalloc(&c->d_hq_cnt, BK_MAXB * sizeof(u32))); HIPCHK(hipMalloc(&c->d_hq_pos, (size_t)BK_MAXB * BK_HQ_DEFER * sizeof(u32))); HIPCHK(hipMemset(c->d_hq_cnt, 0, BK_MAXB * sizeof(u32)));
    HIPCHK(hipMemset(c->d_bkb, 0, 3 * (BK_MAXB + 1) * sizeof(u32))); HIPCHK(hipMemset(c->d_bpos, 0, 2 * (BK_MAXB + 1) * sizeof(u32))); HIPCHK(hipMemset(c->d_bscount, 0, (BK_MAXB + 1) * sizeof(u32)));
    c->kb_B[0] = c->kb_B[1] = c->kb_B[2] = 0; c->scount_B = 0; c->kb_next = c->scount_buf = c->head_kb_use = -1;
    HIPCHK(hipMalloc(&c->d_fstate, 2 * c->cap_ftiles * 8)); HIPCHK(hipMalloc(&c->d_fticket, 4));
    HIPCHK(hipMemset(c->d_fstate, 0, 2 * c->cap_ftiles * 8)); HIPCHK(hipMemset(c->d_fticket, 0, 4));
    c->n_partial_blocks = std::max(nblk(M), BK_MAXB);
    HIPCHK(hipMalloc(&c->d_partials, ((long long)c->n_partial_blocks * NSTAT + 128) * 8));
    HIPCHK(hipMalloc(&c->d_wabs_part, ((long long)c->n_partial_blocks * 2 + 2) * 8));
    HIPCHK(hipMalloc(&c->d_done, 4)); HIPCHK(hipMemset(c->d_done, 0, 4));
  }
  for (int i = 0; i < NTIMERS; i++) { HIPCHK(hipEventCreate(&c->ev0[i])); HIPCHK(hipEventCreate(&c->ev1[i])); }
  if (getenv("SQMC_ONE_STREAM")) { c->st2 = c->st; c->st3 = c->st; }      // experiment: no side streams (their fork/join costs event latencies)
  else { HIPCHK(hipStreamCreate(&c->st2)); HIPCHK(hipStreamCreate(&c->st3)); }
  HIPCHK(hipEventCreateWithFlags(&c->e_join3, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->e_join, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->e_spawned, hipEventDisableTiming));
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
  // plane-wave lookup for the proposal's momentum balance (propose_heg): integer vector -> orbital, in the bytes combine_2 has no use for here
  t.c2_pad = 0;
  { const int W = 2 * t.heg_nmax + 1;
    if ((size_t)W * W * W <= sizeof(t.c2) && !getenv("SQMC_HEG_NO_LUT")) {
      unsigned char *lut = reinterpret_cast<unsigned char *>(t.c2);
      memset(lut, 0, (size_t)W * W * W);
      for (int i = 1; i <= cfg->norb; i++) lut[((t.krel[i][0] + t.heg_nmax) * W + (t.krel[i][1] + t.heg_nmax)) * W + (t.krel[i][2] + t.heg_nmax)] = (unsigned char)i;
      int stride = 1; while ((size_t)stride * stride * sizeof(unsigned short) < (size_t)W * W * W) stride++;
      t.c2_stride = stride; t.c2_pad = 1;          // (c2_stride only sizes the staging of the table here)
    } }
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
static void hb_host_release(sqmc_gpu_ctx *c);
int sqmc_gpu_finalize(sqmc_gpu_ctx *c) {
  abandon_head(c);
  if (!c) return SQMC_OK;
  hb_host_release(c);
  hipStreamSynchronize(c->st);
  if (c->mwalk > 0) {
    free_walk(c->w); free_walk(c->m);
    hipFree(c->d_nchild); hipFree(c->d_child_off); hipFree(c->d_wchild); hipFree(c->d_child_state);
    hipFree(c->d_keys); hipFree(c->d_keys_alt); hipFree(c->d_vals); hipFree(c->d_vals_alt); hipFree(c->d_hist); hipFree(c->d_rowtot);
    hipFree(c->d_flags); hipFree(c->d_pos); hipFree(c->d_flags2); hipFree(c->d_pos2); hipFree(c->d_scan_state); hipFree(c->d_scan_ticket); hipFree(c->d_fstate); hipFree(c->d_fticket); hipFree(c->d_partials); hipFree(c->d_wabs_part); hipFree(c->d_done); hipFree(c->d_segoff); hipFree(c->d_bkb); hipFree(c->d_bhint); hipFree(c->d_bpos); hipFree(c->d_bscount); hipFree(c->d_hq_cnt); hipFree(c->d_hq_pos); hipFree(c->d_jcnt);
  }
  hipFree(c->d_binom); hipFree(c->d_grow); hipFree(c->d_ginv);
  for (int q = 0; q < 16; q++) hipFree(c->d_hbt[q]);
  comm_release(c);
  hipFree(c->d_tab); hipFree(c->d_ints); hipFree(c->d_hb_r); hipFree(c->d_hb_s); hipFree(c->d_hb_absH); hipFree(c->d_pq_ind); hipFree(c->d_pq_count);
  hipFree(c->d_prj_ptr); hipFree(c->d_prj_col); hipFree(c->d_prj_val); hipFree(c->d_loc_imp); hipFree(c->d_prj_x); hipFree(c->d_prj_xs[0]); hipFree(c->d_prj_xs[1]); hipFree(c->d_prj_y);
  hipFree(c->d_ct_up); hipFree(c->d_ct_dn); hipFree(c->d_ct_num); hipFree(c->d_ct_den); hipFree(c->d_ct_hkey); hipFree(c->d_ct_hidx);
  hipFree(c->d_ps_loc); hipFree(c->d_ps_of); hipFree(c->d_ps_impof); hipFree(c->d_ps_c); hipFree(c->d_ps_diag); hipFree(c->d_ps_dwct); hipFree(c->d_ps_dwps);
  hipFree(c->d_ps_dwimp); hipFree(c->d_ps_p2); hipFree(c->d_ps_part); hipFree(c->d_ps_raw); hipFree(c->stage_mem); hipFree(c->d_bpar);
  hipFree(c->d_sc); hipHostFree(c->h_sc); if (c->h_mail) hipHostFree((void *)c->h_mail);
  for (int i = 0; i < NTIMERS; i++) { hipEventDestroy(c->ev0[i]); hipEventDestroy(c->ev1[i]); }
  hipEventDestroy(c->e_fork); hipEventDestroy(c->e_join); hipEventDestroy(c->e_cnt); hipEventDestroy(c->e_spawned);
  for (int i = 0; i < 4; i++) hipEventDestroy(c->hev[i]);
  hipEventDestroy(c->e_join3); if (c->st3 != c->st) hipStreamDestroy(c->st3);
  if (c->st2 != c->st) hipStreamDestroy(c->st2);
  hipStreamDestroy(c->st);
  delete c;
  return SQMC_OK;
}

int sqmc_gpu_set_hb_tables(sqmc_gpu_ctx *c, int64_t n_hb, const int32_t *r, const int32_t *s, const double *a, int32_t n_pq,
                           const int64_t *pq_ind, const int32_t *pq_count, double max_double) {
  abandon_head(c);
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
  abandon_head(c);
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  long long chk = 0; for (long long i = 0; i < n_imp; i++) chk += rc[i];
  if (chk != nnz) return fail(SQMC_ERR_BAD_ARG, "sum(row_counts) != nnz");
  psit_off(c);
  for (long long k = 0; k < nnz; k++) if (idx[k] < 1 || idx[k] > n_imp) return fail(SQMC_ERR_BAD_ARG, "column index out of range");
  std::vector<int> ptr, col; std::vector<double> v;
  expand_full_csr(n_imp, rc, idx, val, ptr, col, v);
  hipFree(c->d_prj_ptr); hipFree(c->d_prj_col); hipFree(c->d_prj_val); hipFree(c->d_loc_imp); hipFree(c->d_prj_x);
  c->n_imp = n_imp; c->prj_nnz = (long long)col.size();
  HIPCHK(hipMalloc(&c->d_prj_ptr, (n_imp + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_col, (col.size() + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_val, (v.size() + 1) * 8));
  HIPCHK(hipMalloc(&c->d_loc_imp, (n_imp + 1) * 4)); HIPCHK(hipMalloc(&c->d_prj_x, (n_imp + 1) * 8));
  hipFree(c->d_prj_xs[0]); hipFree(c->d_prj_xs[1]); hipFree(c->d_prj_y);
  HIPCHK(hipMalloc(&c->d_prj_xs[0], (n_imp + 1) * 8)); HIPCHK(hipMalloc(&c->d_prj_xs[1], (n_imp + 1) * 8)); HIPCHK(hipMalloc(&c->d_prj_y, (n_imp + 1) * 8));
  c->xs_valid = false; c->xs_cur = 0;
  HIPCHK(hipMemcpy(c->d_prj_ptr, ptr.data(), (n_imp + 1) * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_prj_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_prj_val, v.data(), v.size() * 8, hipMemcpyHostToDevice));
  return SQMC_OK;
}
int sqmc_gpu_scale_projector(sqmc_gpu_ctx *c, double ratio) {
  abandon_head(c);
  if (!c || !c->d_prj_val) return fail(SQMC_ERR_BAD_ARG, "no projector");
  hipLaunchKernelGGL(k_scale, dim3(nblk(c->prj_nnz)), dim3(TPB), 0, c->st, c->d_prj_val, c->prj_nnz, ratio);
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}

int sqmc_gpu_set_ct_table(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, const double *num, const double *den) {
  abandon_head(c);
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null ctx");
  for (long long i = 1; i < n; i++)
    if (!(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "C(T) list must be strictly sorted by (up,dn)");
  psit_off(c);
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

// hf_to_psit off: the tables it was set up with are about to change
static void psit_off(sqmc_gpu_ctx *c) {
  if (!c->psit_on) return;
  c->psit_on = false; c->dev.ps.koff = 0;
  c->key_bits = c->base_key_bits; c->invalid_key = (1ull << c->key_bits) - 1ull;
  c->pack = (c->key_bits <= 32 && !getenv("SQMC_FORCE_UNPACKED")) ? 1 : 0;
}
int sqmc_gpu_set_hf_to_psit(sqmc_gpu_ctx *c, int64_t n_psit, const int64_t *psit_ct_index, const double *cdet_psi_t, const double *diag_elems, int32_t sum_order) {
  abandon_head(c);
  if (!c || n_psit < 1 || !psit_ct_index || !cdet_psi_t || !diag_elems) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  if (c->mwalk <= 0) return fail(SQMC_ERR_BAD_ARG, "context has no walker arrays (mwalk=0)");
  if (c->d_grow || c->comm) return fail(SQMC_ERR_UNSUPPORTED, "hf_to_psit is built for one rank only");
  if (!c->d_ct_up || c->n_ct < 1) return fail(SQMC_ERR_BAD_ARG, "set the C(T) table first");
  if (!c->d_prj_ptr || c->n_imp < 1) return fail(SQMC_ERR_BAD_ARG, "set the deterministic-space matrix first");
  if (sum_order != 0 && sum_order != 1) return fail(SQMC_ERR_BAD_ARG, "sum_order must be 0 (left to right) or 1 (64-ary tree)");
  if (n_psit - 1 > PSIT_MAXTERMS || c->n_ct > PSIT_MAXTERMS) return fail(SQMC_ERR_UNSUPPORTED, "trial wave function or C(T) longer than 64^3 determinants: the tree sums have three levels");
  if (c->n_ct >= (1ll << 31)) return fail(SQMC_ERR_UNSUPPORTED, "C(T) too long");
  psit_off(c);
  const long long n_ct = c->n_ct;
  std::vector<int> loc(n_psit), of(n_ct, -1);
  for (long long k = 0; k < n_psit; k++) {
    const long long q = psit_ct_index[k] - 1;
    if (q < 0 || q >= n_ct || (k && q <= loc[k - 1])) return fail(SQMC_ERR_BAD_ARG, "psit_ct_index must be increasing 1-based positions in the C(T) list (Psi_T in label order)");
    loc[k] = (int)q; of[q] = (int)k;
  }
  if (loc[0] != 0) return fail(SQMC_ERR_UNSUPPORTED, "hf_to_psit: the first determinant of Psi_T must be the first determinant of C(T) (the reference assumes it: do_walk.f90:2701-2706, 3574)");
  if (cdet_psi_t[0] == 0.0) return fail(SQMC_ERR_BAD_ARG, "cdet_psi_t(1) = 0");
  // sort keys get one more bit: a determinant outside C(T) sorts behind every determinant of C(T)
  u64 total = 0;
  { auto choose = [&](int n_, int k_) -> long double { long double r = 1; for (int q = 1; q <= k_; q++) r = r * (n_ - k_ + q) / q; return r; };
    total = (u64)(choose(c->htab.norb, c->htab.nup) * choose(c->htab.norb, c->htab.ndn) + 0.5L); }
  if (!c->base_key_bits) c->base_key_bits = c->key_bits;
  if (c->base_key_bits + 1 > 62) return fail(SQMC_ERR_UNSUPPORTED, "determinant space too large for the hf_to_psit sort key");
  int *d_loc, *d_of, *d_impof; double *d_c, *d_diag, *d_dwct, *d_dwps, *d_dwimp, *d_p2, *d_part;
  hipFree(c->d_ps_loc); hipFree(c->d_ps_of); hipFree(c->d_ps_impof); hipFree(c->d_ps_c); hipFree(c->d_ps_diag); hipFree(c->d_ps_dwct); hipFree(c->d_ps_dwps);
  hipFree(c->d_ps_dwimp); hipFree(c->d_ps_p2); hipFree(c->d_ps_part); hipFree(c->d_ps_raw); c->d_ps_raw = nullptr;
  HIPCHK(hipMalloc(&c->d_ps_raw, n_psit * 8));
  HIPCHK(hipMalloc(&d_loc, n_psit * 4)); HIPCHK(hipMalloc(&d_of, n_ct * 4)); HIPCHK(hipMalloc(&d_impof, n_ct * 4));
  HIPCHK(hipMalloc(&d_c, n_psit * 8)); HIPCHK(hipMalloc(&d_diag, n_ct * 8)); HIPCHK(hipMalloc(&d_dwct, n_ct * 8)); HIPCHK(hipMalloc(&d_dwps, n_psit * 8));
  HIPCHK(hipMalloc(&d_dwimp, (c->n_imp + 1) * 8)); HIPCHK(hipMalloc(&d_p2, ((n_ct + 4095) / 4096 + 1) * 8)); HIPCHK(hipMalloc(&d_part, (size_t)PSIT_FB * NSTAT * 8));
  HIPCHK(hipMemcpy(d_loc, loc.data(), n_psit * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_of, of.data(), n_ct * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_impof, 0xFF, n_ct * 4));                  // filled by the upload (the deterministic-space slots are the walkers with imp_distance 0)
  HIPCHK(hipMemcpy(d_c, cdet_psi_t, n_psit * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_diag, diag_elems, n_ct * 8, hipMemcpyHostToDevice));
  c->d_ps_loc = d_loc; c->d_ps_of = d_of; c->d_ps_impof = d_impof; c->d_ps_c = d_c; c->d_ps_diag = d_diag; c->d_ps_dwct = d_dwct; c->d_ps_dwps = d_dwps;
  c->d_ps_dwimp = d_dwimp; c->d_ps_p2 = d_p2; c->d_ps_part = d_part;
  PsitArgs &a = c->psit; memset(&a, 0, sizeof(a));
  a.n_ct = n_ct; a.n_psit = n_psit; a.n_imp = c->n_imp; a.loc_psit = d_loc; a.cdet = d_c; a.diag = d_diag; a.psit_of = d_of; a.imp_of = d_impof;
  a.cnum = c->d_ct_num; a.cden = c->d_ct_den; a.dw_ct = d_dwct; a.dw_ps = d_dwps; a.dw_imp = d_dwimp; a.p2 = d_p2; a.n_perm = 0; a.seq = sum_order == 0 ? 1 : 0;
  u64 first[2];
  HIPCHK(hipMemcpy(&first[0], c->d_ct_up, 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&first[1], c->d_ct_dn, 8, hipMemcpyDeviceToHost));
  c->dev.ps.koff = total; c->dev.ps.hkey = c->d_ct_hkey; c->dev.ps.hmask = c->ct_mask; c->dev.ps.first_up = first[0]; c->dev.ps.first_dn = first[1]; c->dev.ps.n_ct = n_ct;
  c->key_bits = c->base_key_bits + 1; c->invalid_key = (1ull << c->key_bits) - 1ull;
  c->pack = (c->key_bits <= 32 && !getenv("SQMC_FORCE_UNPACKED")) ? 1 : 0;
  c->psit_on = true; c->nwalk = 0;           // walkers are uploaded anew, in the layout of this variant
  return SQMC_OK;
}

int sqmc_gpu_upload_walkers(sqmc_gpu_ctx *c, int64_t n, const uint64_t *up, const uint64_t *dn, const double *wt, const int8_t *impd,
                            const int8_t *init, const int8_t *psign, const double *me, const double *en, const double *ed) {
  abandon_head(c);
  if (!c || c->mwalk <= 0) return fail(SQMC_ERR_BAD_ARG, "context has no walker arrays (mwalk=0)");
  if (n > c->mwalk) return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  const u64 lim = c->htab.orb_mask;
  const long long seg = c->psit_on ? c->dev.ps.n_ct : 0;       // hf_to_psit: [C(T) | survivors outside C(T)], each segment in order (do_walk.f90:1267-1300, 6484-6833)
  for (long long i = 0; i < n; i++) {
    if ((up[i] & ~lim) || (dn[i] & ~lim)) return fail(SQMC_ERR_BAD_ARG, "determinant has bits beyond norb");
    if (__builtin_popcountll(up[i]) != c->htab.nup || __builtin_popcountll(dn[i]) != c->htab.ndn)
      return fail(SQMC_ERR_BAD_ARG, "determinant does not hold nup / ndn electrons");
    if (i && i != seg && !(up[i - 1] < up[i] || (up[i - 1] == up[i] && dn[i - 1] < dn[i]))) return fail(SQMC_ERR_BAD_ARG, "walkers must be sorted by (up,dn) and unique");
  }
  if (c->psit_on) {
    if (n < seg) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit: the walker list begins with ALL determinants of C(T)");
    std::vector<u64> cu(seg), cd(seg);
    HIPCHK(hipMemcpy(cu.data(), c->d_ct_up, seg * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(cd.data(), c->d_ct_dn, seg * 8, hipMemcpyDeviceToHost));
    std::vector<int> impof(seg, -1); int row = 0, n_perm = 0;
    for (long long i = 0; i < seg; i++) {
      if (up[i] != cu[i] || dn[i] != cd[i]) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit: the first n_ct walkers must be the C(T) list");
      if (impd[i] != 0 && impd[i] != -2) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit: a C(T) walker carries imp_distance 0 or -2");
      if (impd[i] == 0) impof[i] = row++;
      if (init[i] == 3) { if (i != 0) return fail(SQMC_ERR_UNSUPPORTED, "hf_to_psit: only the first state can be a permanent initiator (do_walk.f90:1276-1292)"); n_perm = 1; }
    }
    for (long long i = seg; i < n; i++) {
      if (impd[i] < 1) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit: a walker outside C(T) carries imp_distance >= 1 (the deterministic space lies inside C(T))");
      if (init[i] == 3) return fail(SQMC_ERR_UNSUPPORTED, "hf_to_psit: only the first state can be a permanent initiator");
    }
    if (row != c->n_imp) return fail(SQMC_ERR_IMP_BROKEN, "number of imp_distance==0 walkers != n_imp");
    HIPCHK(hipMemcpy(c->d_ps_impof, impof.data(), seg * 4, hipMemcpyHostToDevice));
    c->psit.n_perm = n_perm;
  }
  HIPCHK(hipMemcpy(c->w.up, up, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->w.dn, dn, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->w.wt, wt, n * 8, hipMemcpyHostToDevice));
  { std::vector<u32> f(n); for (long long i = 0; i < n; i++) f[i] = pack_flg(impd[i], init[i], psign[i]);
    HIPCHK(hipMemcpy(c->w.flg, f.data(), n * 4, hipMemcpyHostToDevice)); }
  HIPCHK(hipMemcpy(c->w.me, me, n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(c->w.en, en, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->w.ed, ed, n * 8, hipMemcpyHostToDevice));
  c->nwalk = n; c->residents_sorted = true;       // checked above: sorted and unique
  c->xs_valid = false;
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
  abandon_head(c);
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

int sqmc_gpu_tail_stats(sqmc_gpu_ctx *c, int64_t *bucket_steps, int64_t *bucket_retries) {
  if (!c || !bucket_steps || !bucket_retries) return fail(SQMC_ERR_BAD_ARG, "null argument");
  *bucket_steps = c->bk_steps; *bucket_retries = c->bk_retries;
  return SQMC_OK;
}
int sqmc_gpu_get_rng(sqmc_gpu_ctx *c, int32_t seed[4]) {
  if (!c) return SQMC_ERR_BAD_ARG;
  u64 x; HIPCHK(hipStreamSynchronize(c->st)); HIPCHK(hipMemcpy(&x, &c->d_sc->lcg, 8, hipMemcpyDeviceToHost));
  seed[0] = (int)((x >> 36) & 4095); seed[1] = (int)((x >> 24) & 4095); seed[2] = (int)((x >> 12) & 4095); seed[3] = (int)(x & 4095);
  return SQMC_OK;
}
int sqmc_gpu_set_rng(sqmc_gpu_ctx *c, const int32_t seed[4]) {
  abandon_head(c);
  if (!c) return SQMC_ERR_BAD_ARG;
  u64 x = (((u64)seed[0] << 36) + ((u64)seed[1] << 24) + ((u64)seed[2] << 12) + (u64)(2 * (seed[3] / 2) + 1)) & SQ_MASK48;
  HIPCHK(hipStreamSynchronize(c->st)); HIPCHK(hipMemcpy(&c->d_sc->lcg, &x, 8, hipMemcpyHostToDevice));
  c->seed64 = sq_mix64(x);
  return SQMC_OK;
}

static void collect_timers(sqmc_gpu_ctx *c);
int sqmc_gpu_set_timing(sqmc_gpu_ctx *c, int on) {
  abandon_head(c);
  if (!c) return SQMC_ERR_BAD_ARG;
  hipStreamSynchronize(c->st); hipStreamSynchronize(c->st2); hipStreamSynchronize(c->st3); c->timers_pending = false;
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
// Timing level 1 takes the kernel-exact start/stop events of k_anneal (plain walks: k_spawn) on every SQMC_TIMING_STRIDE-th step only:
// each timed launch costs about 5 us of the step it sits in, which a throughput measurement should not pay on every step.
#define SQMC_TIMING_STRIDE 8
static inline bool kernel_events_on(const sqmc_gpu_ctx *c, u64 step) { return c->timing >= 2 || (c->timing == 1 && step % SQMC_TIMING_STRIDE == 0); }
// level 1 of a semistochastic walk times the annihilation kernel only (the longest on the critical path, the one the roofline is
// quoted on): a timed launch cannot overlap its neighbours, and a second timed kernel on the same step costs the step another 2 us
static inline bool spawn_events_on(const sqmc_gpu_ctx *c, u64 step, int semi) { return c->timing >= 2 || (!semi && kernel_events_on(c, step)); }

static int comm_allreduce_stats(sqmc_gpu_ctx *c);
// Spin on a mailbox word the GPU writes into pinned host memory.  Returns 0 when it arrived, -1
// if the stream drained without it, a hipError_t > 0 if the stream reports an error.
// A stream that neither drains nor delivers within SQMC_MAIL_TIMEOUT_S seconds (default 300; a step takes milliseconds) is
// reported as hipErrorLaunchTimeOut instead of spinning for ever: a peer rank that never enters its collective must not
// turn into a silent hang of the whole job.
static inline double now_us() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec; }
static double *g_mail_wait_us = nullptr;      // sqmc_gpu_shard_step: where the time spent in here is added up (sqmc_gpu_shard_time_split)
static int wait_mail_(volatile u64 *flag, u64 expect, hipStream_t st);
static int wait_mail(volatile u64 *flag, u64 expect, hipStream_t st) {
  if (!g_mail_wait_us) return wait_mail_(flag, expect, st);
  const double t0 = now_us(); const int r = wait_mail_(flag, expect, st); *g_mail_wait_us += now_us() - t0;
  return r;
}
static int wait_mail_(volatile u64 *flag, u64 expect, hipStream_t st) {
  static const double limit_s = getenv("SQMC_MAIL_TIMEOUT_S") ? atof(getenv("SQMC_MAIL_TIMEOUT_S")) : 300.0;
  struct timespec t0; bool timed = false;
  for (unsigned long it = 1;; it++) {
    if (*flag == expect) return 0;
    if ((it & 0x3FFF) == 0) {
      hipError_t e = hipStreamQuery(st);
      if (e != hipErrorNotReady) { if (*flag == expect) return 0; return e == hipSuccess ? -1 : (int)e; }
      if ((it & 0xFFFFF) == 0) {
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        if (!timed) { t0 = t1; timed = true; }
        else if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit_s) return (int)hipErrorLaunchTimeOut;
      }
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
  if (!c->d_grow) return OwnerOut{nullptr, nullptr, 0, 0};
  c->owner_ready = true;
  return OwnerOut{c->d_flags, (u32 *)c->d_flags2, c->shard_n, c->owner_mode};
}
// The head of a step: spawn gate + child offsets + k_spawn (COUNTER discipline).  With dev_n the
// walker count is read on the device (sc->nwalk, written by k_finish of the step before) and n0 is
// only an upper bound that sizes the grids: the pipelined launch behind k_finish of the previous
// step, before the host has read that step's sums.  g0/g1 and s0/s1 (may be null) time gate+scan
// and k_spawn; the child count goes to the host mailbox under sequence number *cseq.
// what does not change from step to step about the short-list (bucket) tail
static inline bool bucket_static_ok(const sqmc_gpu_ctx *c, const StepP &p) {
  static const int bucket_env = getenv("SQMC_ANNEAL_ITEMS") ? 0 : (getenv("SQMC_BUCKET") ? atoi(getenv("SQMC_BUCKET")) : 1);      // a forced tile shape asks for the radix tail's kernel
  static const bool shard_bucket = !(getenv("SQMC_SHARD_BUCKET") && getenv("SQMC_SHARD_BUCKET")[0] == '0');
  return bucket_env && c->pack && p.semi && c->rng_mode == SQMC_RNG_COUNTER && (shard_bucket || (!c->d_grow && c->comm == nullptr)) && !c->dev.hb.on && !c->psit_on;      // heat-bath children take two slots each: radix tail; hf_to_psit: radix tail (psit_kernels.h)
}
// Boundaries that follow the spawns (bucket_partition.h): the partition about to be launched uses the set the last one's boundary
// block made, if it was made for B buckets; its own boundary block makes the next set from the counts of the last bucket tail.
// Returns the set in use (-1: equal-residents boundaries).
static int choose_boundaries(sqmc_gpu_ctx *c, long long B, long long n_known, BucketArgs &hb) {
  static const bool no_rebal = getenv("SQMC_BUCKET_UNIFORM") != nullptr;
  int use = -1;
  if (!no_rebal && B <= BK_REBAL_MAXB && n_known >= 16 * B) {
    use = (c->kb_next >= 0 && c->kb_B[c->kb_next] == (int)B) ? c->kb_next : -1;
    if (use >= 0) { c->pos_flip ^= 1; hb.kb = c->d_bkb + use * (BK_MAXB + 1); hb.pos = c->d_bpos + c->pos_flip * (BK_MAXB + 1); hb.hint = c->d_bhint + use * (BK_MAXB + 1); }
    if (c->scount_B == (int)B) {
      const int prev = (c->scount_buf >= 0 && c->kb_B[c->scount_buf] == (int)B) ? c->scount_buf : -1;
      int out = 0; while (out == use || out == prev) out++;
      hb.kb_prev = prev >= 0 ? c->d_bkb + prev * (BK_MAXB + 1) : (const u32 *)nullptr;
      hb.kb_out = c->d_bkb + out * (BK_MAXB + 1); hb.hint_out = c->d_bhint + out * (BK_MAXB + 1);
      hb.pos_prev = prev >= 0 ? c->d_bpos + c->scount_pos * (BK_MAXB + 1) : (const u32 *)nullptr;
      c->kb_B[out] = (int)B; c->kb_next = out;
    }
  }
  return use;
}
static inline long long bucket_count(long long nall) {
  static const long long bk_target = getenv("SQMC_BUCKET_TARGET") ? atoll(getenv("SQMC_BUCKET_TARGET")) : BK_TARGET;
  long long B = (nall + bk_target - 1) / bk_target;
  // BK_PER_CU blocks per CU (their LDS is the bucket): up to 256 BK_PER_CU blocks run at once; a few blocks more would wait for a whole second round
  const long long conc = 256 * BK_PER_CU;
  if (B > conc && nall <= conc * (long long)(bk_target + bk_target / 4)) B = conc;
  return B < 1 ? 1 : (B > BK_MAXB ? BK_MAXB : B);
}
static int enqueue_head(sqmc_gpu_ctx *c, const StepP &p, u64 step, long long n0, bool dev_n, hipEvent_t g0, hipEvent_t g1, hipEvent_t s0, hipEvent_t s1, u64 *cseq,
                        const FinArgs *fin = nullptr, bool gate_done = false) {
  hipStream_t st = c->st;
  const long long M = c->mwalk;
  ScanWork sw0; sw0.state = c->d_scan_state; sw0.ticket = c->d_scan_ticket; sw0.cap_tiles = c->cap_tiles; sw0.self_clear = false;
  FinArgs fa; memset(&fa, 0, sizeof(fa)); if (fin) fa = *fin;
  if (g0) hipEventRecord(g0, st);
  FinArgs spawn_fin; memset(&spawn_fin, 0, sizeof(spawn_fin));
  PrjPre pp; memset(&pp, 0, sizeof(pp));
  c->head_y_done = false;
  c->shard_y_ok = false;
  if (dev_n && c->d_grow && c->comm && !c->comm2 && c->n_imp > 0 && p.semi) {
    // In-library sharded step (one communicator): the all-reduce of the deterministic weights needs nothing the host still has to
    // decide either.  It goes in front of k_spawn, whose spare blocks then multiply this rank's rows of the projector into the
    // result; the step itself only adds the last line.  (Decided by quantities every rank shares: the collectives keep their order.)
    int rp = shard_head_project(c, c->head_sums_ride, false, c->head_sums_ride ? &c->head_fin : nullptr); if (rp) return rp;
    pp.n_imp = (int)c->n_imp_local; pp.ptr = c->d_prj_ptr; pp.col = c->d_prj_col; pp.val = c->d_prj_val; pp.x = c->d_xg; pp.y = c->d_prj_y; pp.grow = c->d_grow;
    c->shard_y_ok = true; c->shard_x_ready = true;
  }
  if (dev_n && c->head_prj_x && c->n_imp > 0 && !c->d_grow) {       // the tail that enqueues this head is a bucket tail: its deterministic weights, row by row, are this step's x
    pp.n_imp = (int)c->n_imp; pp.ptr = c->d_prj_ptr; pp.col = c->d_prj_col; pp.val = c->d_prj_val; pp.x = c->head_prj_x; pp.y = c->d_prj_y;
    c->head_y_done = true;
  }
  c->head_prj_x = nullptr;
  if (gate_done && c->head_offsets_done) {
    // the bucket tail of the step before wrote keys, child weights, child OFFSETS and their total: nothing to scan.  That
    // step's final sums ride on k_spawn as one extra block.
    c->head_offsets_done = false;
    if (c->head_offsets_bucket) spawn_fin = fa;
    else if (fa.on) {
      // long lists: k_spawn keeps the registers and LDS of its plain form, and the one block that adds up thousands of tile partials
      // (20 us at 10^6 walkers, 80 at 10^7) runs BESIDE it on the side stream, as it ran beside the scan; the next tail joins that
      // stream before its annihilation kernel needs the look-back words this block re-zeroes
      if (c->st2 != st) { HIPCHK(hipEventRecord(c->e_cnt, st)); HIPCHK(hipStreamWaitEvent(c->st2, c->e_cnt, 0)); }
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(TPB), 0, c->st2, fa, c->d_sc);
    }
  } else if (gate_done) {
    // k_anneal of the step before wrote keys, child counts and child weights; the final sums of that step ride on the
    // scan as one extra block.  The scan works on look-back set scan_flip while that block re-zeroes the other set
    // (the one the head scan before this one used).
    const int f = c->scan_flip;
    sw0.state = c->d_scan_state + (long long)f * c->cap_tiles; sw0.ticket = c->d_scan_ticket + f;
    c->scan_used[f] = device_excl_scan_u64<FinExtra>(c->d_nchild, c->d_child_off, n0, &c->d_sc->n_children, sw0, st, dev_n ? &c->d_sc->nwalk : nullptr, FinExtra{fa, c->d_sc},
                                                             dev_n ? &c->d_sc->retry : nullptr);
    c->scan_flip = f ^ 1;
  } else {
    hipLaunchKernelGGL(k_gate, dim3(nblk(n0)), dim3(TPB), 0, st, c->dev, c->w.up, c->w.dn, c->w.wt, c->d_nchild, c->d_wchild, c->d_keys, c->d_vals,
                       n0, p, c->seed64, step, c->d_sc, c->pack, dev_n ? 1 : 0, fa);
    c->scan_used[0] = device_excl_scan_u64(c->d_nchild, c->d_child_off, n0, &c->d_sc->n_children, sw0, st, dev_n ? &c->d_sc->nwalk : nullptr, ScanNoExtra(), dev_n ? &c->d_sc->retry : nullptr);
    c->scan_flip = 1;          // set 0 stays dirty until a finish re-zeroes it: a gate-fused head that follows works on set 1
  }
  if (g1) hipEventRecord(g1, st);
  // ---- spawn goes out first: the host is the slower side at the start of a step, and k_spawn
  //      is on the critical path (exactly one k_spawn launch inside this timer: the per-launch time
  //      bench.py reports, taken from the kernel's own start/stop timestamps).  It is launched
  //      over the whole free capacity with a device-side child count and posts that count to the
  //      host mailbox as soon as it starts.
  const bool tail_fills = dev_n && c->tail_fills_hii && !c->d_grow;
  c->tail_fills_hii = false;
  if (!tail_fills) HIPCHK(hipEventRecord(c->e_fork, st));
  c->fork_valid = !tail_fills;
  // pipelined head: the diagonal elements of the determinants the last step created depend on nothing the host still has to
  // decide: a kernel on the side stream computes them now, beside the scan and k_spawn; death/clone later finds them cached.
  // (Doing this in spare blocks of k_spawn instead saves the cross-stream join but costs k_spawn its occupancy through the extra
  // LDS: 122 against 110 us per step at 10^5 walkers, and 35 % more spawn time at 10^6-10^7 -- taken out again.)
  static const bool no_early = getenv("SQMC_NO_EARLY_HII") != nullptr;
  // Behind a bucket tail there is nothing to fill: that kernel computes the H_ii of the determinants it creates, and the step
  // runs on one stream (no fork, no join: every cross-stream wait costs 5-12 us on the critical path at 10^5 walkers).
  // (long lists: k_spawn and the sort are longer than both passes of k_diag together, and the early pass only takes resources from
  //  k_spawn -- 0.449 -> 0.444 ms at 10^6 walkers, 3.15 -> 3.10 ms at 10^7 without it)
  const bool early = dev_n && !c->d_grow && !no_early && !tail_fills && c->last_nall < (1ll << 20);
  if (early) {
    HIPCHK(hipStreamWaitEvent(c->st2, c->e_fork, 0));
    hipLaunchKernelGGL(k_diag, dim3(nblk(n0)), dim3(TPB), 0, c->st2, c->dev, c->w.up, c->w.dn, c->w.wt, c->w.flg, c->w.me, n0, p, c->d_sc, 1);
    HIPCHK(hipEventRecord(c->e_join, c->st2));          // a tail that does death/clone itself still has to wait for these
  }
  c->head_hii = early || tail_fills; c->head_hii_joined = tail_fills;
  *cseq = ++c->cnt_seq;
  const OwnerOut oo = shard_owner_out(c);
  // short lists: k_spawn groups its children by key range as it emits them (the bucket tail then needs no partition kernel).
  // The list's length is not known yet: the last step's sizes it; the tail checks that everything fitted before it relies on it.
  BucketArgs hb; memset(&hb, 0, sizeof(hb));
  {
    static const long long merge_min = getenv("SQMC_MERGE_SORT_MIN") ? atoll(getenv("SQMC_MERGE_SORT_MIN")) : (1ll << 20);
    static const bool no_fuse = getenv("SQMC_BUCKET_NO_SPAWN_FUSION") != nullptr;
    const long long n_known = dev_n ? c->nwalk : n0;                   // dev_n: the walkers of the step that is finishing, not of this one -- close
    const long long est = c->last_nall > 0 ? c->last_nall : 0;
    if (!no_fuse && !c->d_grow && bucket_static_ok(c, p) && c->residents_sorted && c->bk_holdoff == 0 && est > 0 && est < merge_min && est < (1ll << 20) && n_known >= 256) {      // sharded steps: the children a rank anneals are not the ones it spawned
      long long B = bucket_count(est); if (B > n_known / 2) B = n_known / 2;
      if (B >= 1) {
        hb.B = (int)B; hb.words = c->d_flags; hb.segoff = c->d_segoff; hb.state = c->d_fstate; hb.ticket = c->d_fticket;
        hb.scount = c->d_bscount;
        hb.nsb = (int)std::min<long long>(std::min<long long>(c->segoff_cap / (B + 1), M / BK_T), BK_CAP_ROWS);      // rows there is room for
        c->head_kb_use = choose_boundaries(c, B, n_known, hb);
      }
    }
    c->head_ba = hb;
  }
  if (gate_done && c->head_bpar_ok && dev_n) hb.parent_hint = c->d_bpar;      // the tail that enqueues this head wrote the child offsets and, beside them, every block's first parent
  c->head_bpar_ok = false;
  int hq_blk = 0;
  if (dev_n && tail_fills && c->hii_deferred_B > 0) {      // the tail left the new determinants' H_ii to this kernel: one spare block per bucket's queue
    hb.hq_cnt = c->d_hq_cnt; hb.hq_pos = c->d_hq_pos; hb.hq_B = c->hii_deferred_B; hb.hq_nblk = hq_blk = c->hii_deferred_B;
    c->head_ba.hq_cnt = nullptr; c->head_ba.hq_pos = nullptr; c->head_ba.hq_nblk = 0;      // (the copy the tail takes over describes the partition only)
  }
  c->hii_deferred_B = 0;
  const long long nfree = dev_n ? M : M - n0;          // dev_n: nothing is known about the count but that it is >= 0
  const int spawn_fuse = (hb.B > 0 || spawn_fin.on || pp.n_imp > 0 || c->shard_y_ok || hq_blk > 0) ? 1 : 0;
  const size_t spawn_lds = std::max<size_t>(hb.B > 0 ? (size_t)BK_PART_LDS(hb.B) : 0, hq_blk > 0 ? (size_t)(TPB / 16) * bk_hii_terms_of(c->htab.nup, c->htab.ndn) * 8 : 0);
  if (nfree > 0) {
    if (s0)
      SPAWN_LAUNCH_EXT(c->dev.hb.on, spawn_fuse, dim3(nblk(nfree) + (spawn_fin.on ? 1 : 0) + hq_blk + (pp.n_imp > 0 ? nblk(pp.n_imp, TPB / 64) : 0) + ((hb.kb || hb.kb_out) ? 1 : 0)), dim3(TPB), spawn_lds, st, s0, s1, 0, c->dev, c->w, c->d_child_off, c->d_wchild,
                            c->d_child_state, c->d_keys, c->d_vals, n0, M, p, c->rng_mode, c->seed64, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, *cseq, c->pack, dev_n ? 1 : 0, oo, hb, spawn_fin, pp, (spawn_fin.on ? 1 : 0) + hq_blk + (pp.n_imp > 0 ? nblk(pp.n_imp, TPB / 64) : 0) + ((hb.kb || hb.kb_out) ? 1 : 0));
    else
      SPAWN_LAUNCH(c->dev.hb.on, spawn_fuse, dim3(nblk(nfree) + (spawn_fin.on ? 1 : 0) + hq_blk + (pp.n_imp > 0 ? nblk(pp.n_imp, TPB / 64) : 0) + ((hb.kb || hb.kb_out) ? 1 : 0)), dim3(TPB), spawn_lds, st, c->dev, c->w, c->d_child_off, c->d_wchild, c->d_child_state, c->d_keys, c->d_vals,
                         n0, M, p, c->rng_mode, c->seed64, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, *cseq, c->pack, dev_n ? 1 : 0, oo, hb, spawn_fin, pp, (spawn_fin.on ? 1 : 0) + hq_blk + (pp.n_imp > 0 ? nblk(pp.n_imp, TPB / 64) : 0) + ((hb.kb || hb.kb_out) ? 1 : 0));
  } else if (s0) { hipEventRecord(s0, st); hipEventRecord(s1, st); }
  HIPCHK(hipEventRecord(c->e_spawned, st)); c->spawned_valid = true;
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}
// a pipelined head whose step will not run (the step before it failed): drain it and reset what it touched
static void drop_head(sqmc_gpu_ctx *c) {
  if (!c->head_ready) return;
  c->head_ready = false;
  hipStreamSynchronize(c->st); hipStreamSynchronize(c->st2);
  hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
  c->scan_used[0] = c->scan_used[1] = 0;
}
static void abandon_head(sqmc_gpu_ctx *c) {
  if (c) c->shard_x_ready = false;       // the walkers are about to change under the caller's hands (every rank's: uploads are collective in a sharded walk)
  if (!c || !c->head_ready) return;
  drop_head(c);                          // waits for its kernels (they wrote scratch only: keys, child offsets, spawn records, partition rows) and re-zeroes the scan words
  c->head_ba.B = 0; c->head_offsets_done = false; c->head_y_done = false; c->head_hii = false; c->head_hii_joined = false; c->head_prj_x = nullptr;
  c->side_pending = false; c->xs_valid = false; c->tail_fills_hii = false; c->fork_valid = false;
}
// sort -> merge -> round -> compact/estimate -> readback; shared by the single-rank step and
// the sharded step (where the spawns behind slot n0 arrived from other ranks)
// death/clone (k_diag) and the deterministic projection as kernels of their own: on the two side streams, forked at e_fork and
// joined by e_join / e_join3 -- or, `serial`, on the main stream (a tail that found them still pending)
static int launch_side_kernels(sqmc_gpu_ctx *c, const StepP &p, long long n0, bool serial) {
  if (!serial && !c->fork_valid) { HIPCHK(hipEventRecord(c->e_fork, c->st)); c->fork_valid = true; }      // the head forked nothing: the side streams start behind what is enqueued so far
  hipStream_t st2 = serial ? c->st : c->st2;
  // long lists: death/clone starts behind k_spawn and runs beside the sort, whose kernels leave most of the chip idle -- beside k_spawn it
  // took its wave slots (k_spawn alone: 80 us at 10^6 walkers, 93 with death/clone beside it)
  static const bool diag_late_env = !(getenv("SQMC_DIAG_BESIDE_SPAWN") != nullptr);
  const bool diag_late = diag_late_env && !serial && c->spawned_valid && c->last_nall >= (1ll << 20);
  if (!serial) HIPCHK(hipStreamWaitEvent(st2, diag_late ? c->e_spawned : c->e_fork, 0));
  c->spawned_valid = false;
  TBEG(diag, st2);
  hipLaunchKernelGGL(k_diag, dim3(nblk(n0)), dim3(TPB), 0, st2, c->dev, c->w.up, c->w.dn, c->w.wt, c->w.flg, c->w.me, n0, p, c->d_sc, 0);
  TEND(diag, st2);
  // the projection reads and writes the weights of the deterministic-space walkers only (imp_distance 0), death/clone skips
  // exactly those: the two run side by side
  static const bool no_st3 = getenv("SQMC_NO_ST3") != nullptr;
  hipStream_t st3 = (p.semi && !no_st3 && !serial) ? c->st3 : st2;
  if (st3 != st2) HIPCHK(hipStreamWaitEvent(st3, c->e_fork, 0));
  TBEG(project, st3);
  if (p.semi && c->psit_on) {
    // hf_to_psit, do_walk.f90:2262-2323: the deterministic-space product without the E_T term, the first row / column / extra diagonal of
    // the transformed projector over C(T), the Psi_T locations; then the three updates slot by slot in the reference's order
    PsitArgs &a = c->psit;
    hipLaunchKernelGGL(k_prj_gather, dim3(nblk(c->n_imp)), dim3(TPB), 0, st3, c->w.wt, c->d_loc_imp, c->d_prj_x, c->n_imp);
    PrjPre pp; memset(&pp, 0, sizeof(pp));
    pp.n_imp = (int)c->n_imp; pp.ptr = c->d_prj_ptr; pp.col = c->d_prj_col; pp.val = c->d_prj_val; pp.x = c->d_prj_x; pp.y = a.dw_imp;
    hipLaunchKernelGGL(k_psit_imp_rows, dim3(nblk(c->n_imp, TPB / 64)), dim3(TPB), 0, st3, pp);
    hipLaunchKernelGGL(k_psit_ct_col, dim3(nblk(a.n_ct)), dim3(TPB), 0, st3, a, (const double *)c->w.wt);
    hipLaunchKernelGGL(k_psit_ct_terms, dim3((unsigned)((a.n_ct + 4095) / 4096)), dim3(64), 0, st3, a, (const double *)c->w.wt);
    hipLaunchKernelGGL(k_psit_rows_fin, dim3(1), dim3(TPB), 0, st3, a, (const double *)c->w.wt, p.tau, p.e_trial);
    hipLaunchKernelGGL(k_psit_apply, dim3(nblk(a.n_ct)), dim3(TPB), 0, st3, a, c->w.wt, p.tau, p.e_trial);
    if (st3 != st2) HIPCHK(hipEventRecord(c->e_join3, st3));
  } else if (p.semi) {
    hipLaunchKernelGGL(k_prj_gather, dim3(nblk(c->n_imp)), dim3(TPB), 0, st3, c->w.wt, c->d_loc_imp, c->d_prj_x, c->n_imp);
    hipLaunchKernelGGL(k_prj_apply, dim3(nblk(c->n_imp, TPB / 64)), dim3(TPB), 0, st3, c->d_prj_ptr, c->d_prj_col, c->d_prj_val, c->d_prj_x, c->d_loc_imp, c->w.wt,
                       c->n_imp, p.e_trial, p.tau);
    if (st3 != st2) HIPCHK(hipEventRecord(c->e_join3, st3));
  }
  TEND(project, st3);
  if (!serial) HIPCHK(hipEventRecord(c->e_join, st2));
  HIPCHK(hipGetLastError());
  return SQMC_OK;
}
// the two-kernel annihilation of long lists: pipelined COUNTER steps of one GPU whose tail carries the child offsets; allocates the
// staging buffer (72 bytes per walker slot) the first time
static bool anneal_split_ok(sqmc_gpu_ctx *c, const StepP &p, int mode, long long nall, int items, bool child_off, bool use_mail) {
  static const int env = getenv("SQMC_ANNEAL_SPLIT") ? atoi(getenv("SQMC_ANNEAL_SPLIT")) : 1;
  const char *mn = getenv("SQMC_ANNEAL_SPLIT_MIN");          // tests: the two-kernel form on short lists too (read per step)
  const long long min_slots = mn ? atoll(mn) : (1ll << 20);
  if (!env || !p.semi || c->psit_on || mode != SQMC_RNG_COUNTER || !child_off || !use_mail || c->d_grow || nall < min_slots || items < 3) return false;
  const long long M = c->mwalk;
  if (!c->stage_mem || c->stage_cap < M) {
    hipFree(c->stage_mem); c->stage_mem = nullptr; c->stage_cap = 0;
    const long long nt = M / (TPB * 3) + 2;
    const size_t bytes = (size_t)M * (7 * 8 + 4 * 4) + (size_t)nt * 32 + 256;
    if (hipMalloc(&c->stage_mem, bytes) != hipSuccess) { (void)hipGetLastError(); c->stage_mem = nullptr; return false; }
    char *b = (char *)c->stage_mem; AnnealStage &g = c->stage;
    g.up = (u64 *)b; b += M * 8; g.dn = (u64 *)b; b += M * 8; g.key = (u64 *)b; b += M * 8;
    g.wt = (double *)b; b += M * 8; g.me = (double *)b; b += M * 8; g.en = (double *)b; b += M * 8; g.ed = (double *)b; b += M * 8;
    g.flg = (u32 *)b; b += M * 4; g.nc = (u32 *)b; b += M * 4; g.lch = (u32 *)b; b += M * 4; g.ldet = (u32 *)b; b += M * 4;
    g.cnt_a = (u64 *)b; b += nt * 8; g.cnt_b = (u64 *)b; b += nt * 8; g.off_a = (u64 *)b; b += nt * 8; g.off_b = (u64 *)b;
    c->stage_cap = M;
  }
  return true;
}
#define SQMC_INTERNAL_RETRY 1000      // step_tail_impl: the bucket tail gave up, nothing of it counts; run the radix tail
static int step_tail_impl(sqmc_gpu_ctx *c, const StepP &p_in, long long n0, long long nall, bool join_side_stream, double out[16], bool allow_bucket) {
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
  // Short lists (the launch-bound regime): no global sort at all.  The spawns are partitioned block-locally into B key
  // ranges whose splitters are resident walkers (sorted since the last step), and one block per range sorts, merges and
  // annihilates its share in LDS (bucket_kernels.h).  Needs packed keys, the COUNTER discipline, ordered residents.
  static const long long bucket_max = getenv("SQMC_BUCKET_MAX") ? atoll(getenv("SQMC_BUCKET_MAX")) : (1ll << 20);
  BucketArgs ba; memset(&ba, 0, sizeof(ba));
  bool bucket = false;
  const BucketArgs head_ba = c->head_ba; c->head_ba.B = 0;             // consumed (or ignored) by this tail
  c->last_nall = nall; c->dbg_n0 = n0; c->dbg_nall = nall;
  if (allow_bucket && bucket_static_ok(c, p) && c->residents_sorted && nall > n0 && nall < bucket_max && nall < merge_min && n0 >= 64) {
    if (c->bk_holdoff > 0) c->bk_holdoff--;
    else {
      const long long nch = nall - n0, nsb = (nch + BK_T - 1) / BK_T;
      static const int force_every = getenv("SQMC_BUCKET_FORCE_RETRY") ? atoi(getenv("SQMC_BUCKET_FORCE_RETRY")) : 0;      // tests: every n-th bucket step gives up
      if (head_ba.B > 0 && head_ba.B <= n0 && nsb <= head_ba.nsb && (n0 + head_ba.B - 1) / head_ba.B <= BK_CAP_R) {
        bucket = true; ba = head_ba; ba.nsb = (int)nsb;                  // k_spawn partitioned its children already
      } else {
        long long B = bucket_count(nall); if (B > n0) B = n0;
        if (nsb <= BK_CAP_ROWS && (n0 + B - 1) / B <= BK_CAP_R && nsb * (B + 1) <= c->segoff_cap && nsb * BK_T <= M) {
          bucket = true;
          ba.B = (int)B; ba.nsb = (int)nsb; ba.words = c->d_flags; ba.segoff = c->d_segoff; ba.state = c->d_fstate; ba.ticket = c->d_fticket;
          ba.scount = c->d_bscount;
          int n_extra = 0;
          if (c->d_grow && c->shard_ba.B == (int)B) {          // sharded step: chosen, and computed on the side stream, when the step began
            ba.kb = c->shard_ba.kb; ba.pos = c->shard_ba.pos; ba.hint = c->shard_ba.hint;
          } else if (!c->d_grow) {
            c->head_kb_use = choose_boundaries(c, B, n0, ba);
            n_extra = (ba.kb || ba.kb_out) ? 1 : 0;
          } else c->head_kb_use = -1;
          c->shard_ba.B = 0;
          hipLaunchKernelGGL(k_bucket_partition, dim3((unsigned)(nsb + n_extra)), dim3(BK_T), 0, st, (const u64 *)c->d_keys, n0, nch, c->invalid_key, ba, n_extra);
        }
      }
      // pipelined, unsharded: the H_ii of the determinants this tail creates are left to spare blocks of the head's k_spawn (enqueued below)
      static const bool no_defer = getenv("SQMC_BUCKET_NO_HII_DEFER") != nullptr;
      c->hii_deferred_B = 0;
      if (bucket && c->pipeline_next && !c->d_grow && !no_defer && c->rng_mode != SQMC_RNG_REPLAY) { ba.hq_cnt = c->d_hq_cnt; ba.hq_pos = c->d_hq_pos; c->hii_deferred_B = ba.B; }
      static const int force_rank = getenv("SQMC_BUCKET_FORCE_RETRY_RANK") ? atoi(getenv("SQMC_BUCKET_FORCE_RETRY_RANK")) : -1;      // tests: only this rank of a sharded walk
      if (bucket) ba.force_retry = (force_every > 0 && (force_rank < 0 || force_rank == c->shard_rank) && (c->bk_steps % force_every) == force_every - 1) ? 1 : 0;
    }
  }
  if (!bucket && c->side_pending) {          // the head counted on the bucket tail for death/clone and the projection: do them now, in line
    c->side_pending = false;
    int rs = launch_side_kernels(c, p, n0, true); if (rs) return rs;
  }
  if (bucket) {
  } else if (p.semi && c->residents_sorted && nall >= merge_min) {
    // Large lists: the walkers [0, n0) are in order already (every step leaves them so), so only the spawns
    // [n0, nall) are sorted and one stable merge (walker before spawns on equal keys, spawns in creation order)
    // gives the order the full sort would.  The merged list lands in the flag arrays the fused tail does not use.
    const long long nch = nall - n0;
    if (nch > 0) {
      u64 *sk = c->d_keys + n0; u32 *sv = c->pack ? (u32 *)nullptr : c->d_vals + n0;
      so.k_alt = c->d_keys_alt + n0; so.v_alt = c->d_vals_alt + n0;
      device_radix_sort(sk, sv, nch, c->key_bits, so, st, c->pack ? 32 : 0);
      if (c->pack) device_merge_sorted(c->d_keys, nullptr, n0, sk, nullptr, nch, c->d_flags, nullptr, 32, st);
      else device_merge_sorted(c->d_keys, c->d_vals, n0, sk, sv, nch, c->d_flags, (u32 *)c->d_flags2, 0, st);
      skey = c->d_flags; if (!c->pack) perm = (u32 *)c->d_flags2;
    }
  } else {
    device_radix_sort(skey, perm, nall, c->key_bits, so, st, c->pack ? 32 : 0);
    if (skey != c->d_keys) { c->d_keys_alt = c->d_keys; c->d_keys = skey; if (!c->pack) { c->d_vals_alt = c->d_vals; c->d_vals = perm; } }
  }
  TEND(sort, st);
  // ---- join: from here on weights are read
  if (join_side_stream) { HIPCHK(hipStreamWaitEvent(st, c->e_join, 0)); if (p.semi && !c->d_grow && !getenv("SQMC_NO_ST3")) HIPCHK(hipStreamWaitEvent(st, c->e_join3, 0)); }
  const int nbm = nblk(nall);
  const bool use_mail = (c->comm == nullptr);          // with a communicator the sums are all-reduced on the device first
  const u64 seq = ++c->mail_seq;
  int nb, n_ft = 0;
  bool fuse_gate = false;
  if (p.semi) {
    // one kernel from the sorted list to the new walker arrays; the buffers swap roles afterwards
    // timed by the kernel's own start/stop timestamps (hipExtLaunchKernelGGL events) at every timing level: the
    // per-launch time bench.py reports for the roofline of this, the longest kernel on the critical path
    int t_anneal = -1;
    if (kernel_events_on(c, step) && c->nt < NTIMERS) { t_anneal = c->nt++; c->tname[t_anneal] = "anneal"; }
    static const int items_env = getenv("SQMC_ANNEAL_ITEMS") ? atoi(getenv("SQMC_ANNEAL_ITEMS")) : 0;
    // small lists want many tiles, large ones short look-back chains; 4 slots per thread spill 31 registers at the 4 waves per SIMD
    // the kernel wants (3: 8), which only pays from ~10^7 slots on (measured: 0.449 against 0.458 ms/step at 2.5e6 slots, 3.44 against 3.35 at 1.7e7)
    int items = items_env ? items_env : (nall < (1ll << 20) ? 2 : (nall < (1ll << 23) ? 3 : 4));
    if (items < 3 && getenv("SQMC_ANNEAL_SPLIT_MIN") && nall >= atoll(getenv("SQMC_ANNEAL_SPLIT_MIN"))) items = 3;      // (tests: the two-kernel form exists for 3 and 4 slots per thread)
    if (c->psit_on) items = items <= 2 ? 2 : 3;          // the two shapes k_anneal<., 1> is instantiated with
    nb = n_ft = bucket ? ba.B : (int)((nall + (long long)TPB * items - 1) / ((long long)TPB * items));
    // pipelined steps: the kernel also does the next step's gate (keys, child counts, child weights) as it places a walker
    static const bool no_fuse = getenv("SQMC_NO_GATE_FUSION") != nullptr;
    static const bool no_psit_fuse = getenv("SQMC_PSIT_NO_GATE_FUSION") != nullptr;
    static const bool no_wide_fuse = getenv("SQMC_NO_WIDE_KEY_GATE_FUSION") != nullptr;      // keys wider than 32 bits (the electron gas): fused as well, the slot index in its own array
    fuse_gate = c->pipeline_next && (c->pack || (!c->psit_on && !no_wide_fuse && !bucket)) && !no_fuse && !(c->psit_on && no_psit_fuse);
    GateOut go; memset(&go, 0, sizeof(go));
    if (c->psit_on) { go.ps_of = c->d_ps_of; go.ps_raw = c->d_ps_raw; }
    if (fuse_gate) {
      go.on = 1; go.keys = (skey == c->d_keys) ? c->d_keys_alt : c->d_keys;      // never the buffer the kernel reads its sorted words from
      go.vals = (skey == c->d_keys) ? c->d_vals_alt : c->d_vals; go.pack = c->pack;
      go.nchild = c->d_nchild; go.wchild = c->d_wchild; go.cutoff = p.cutoff; go.step_next = step + 1;
      static const bool no_off = getenv("SQMC_BUCKET_NO_OFFSETS") != nullptr;
      static const bool no_shard_off = getenv("SQMC_SHARD_NO_OFFSETS") != nullptr;
      if (bucket && (use_mail || !no_shard_off) && !no_off && c->n_imp < (1ll << 18) && c->last_wabs > 0 && c->last_wabs < 4.0e6) go.child_off = c->d_child_off;     // 24 bits of the look-back word hold the children
      // the radix tail of an unsharded pipelined step carries the child offsets too, in a look-back of their own (walk_kernels.h)
      static const bool no_roff = getenv("SQMC_RADIX_NO_OFFSETS") != nullptr;
      if (!bucket && use_mail && !c->d_grow && mode == SQMC_RNG_COUNTER && !no_roff && !c->psit_on) go.child_off = c->d_child_off;      // (hf_to_psit: the C(T) segment's counts come from a later kernel: the head scans)
      static const bool no_hint = getenv("SQMC_NO_PARENT_HINT") != nullptr;
      if (go.child_off && use_mail && !c->d_grow && !no_hint) go.bpar = c->d_bpar;
    }
    c->head_bpar_ok = go.bpar != nullptr;
#define ANNEAL_ARGS c->w, c->m, skey, perm, c->d_loc_imp, c->d_ct_hkey, c->d_ct_hidx, c->ct_mask, c->d_ct_num, c->d_ct_den, c->d_partials, c->d_wabs_part, n0, nall, p,  \
                    c->invalid_key, c->pack, mode, seed, step, c->d_sc, c->d_fstate, c->d_fstate + c->cap_ftiles, c->d_fticket, go
#define ANNEAL_LAUNCH_(I, P) do { if (t_anneal >= 0) hipExtLaunchKernelGGL((k_anneal<I, P, 0>), dim3(nb), dim3(TPB), 0, st, c->ev0[t_anneal], c->ev1[t_anneal], 0, ANNEAL_ARGS, AnnealStage{}); \
                                  else hipLaunchKernelGGL((k_anneal<I, P, 0>), dim3(nb), dim3(TPB), 0, st, ANNEAL_ARGS, AnnealStage{}); } while (0)
#define ANNEAL_LAUNCH(I) ANNEAL_LAUNCH_(I, 0)
    if (bucket) {
      FusedSide fs; memset(&fs, 0, sizeof(fs));
      fs.on = c->side_pending ? 1 : 0; fs.y = c->d_prj_y;
      fs.x_in = c->d_prj_xs[c->xs_cur]; fs.x_out = c->d_prj_xs[c->xs_cur ^ 1];
      static const bool no_fs = getenv("SQMC_NO_FUSED_SIDE") != nullptr;
      if (c->pipeline_next && p.semi && !no_fs) c->head_prj_x = fs.x_out;      // the head enqueued below may multiply the projector into it
#define BUCKET_ARGS c->w, c->m, (const u64 *)c->d_keys, c->d_loc_imp, c->d_ct_hkey, c->d_ct_hidx, c->ct_mask, c->d_ct_num, c->d_ct_den, c->d_partials, c->d_wabs_part, \
                    n0, nall - n0, p, c->invalid_key, seed, step, c->d_sc, ba, go, fs, c->dev
      if (t_anneal >= 0) hipExtLaunchKernelGGL(k_anneal_bucket, dim3(nb), dim3(BK_AT), 0, st, c->ev0[t_anneal], c->ev1[t_anneal], 0, BUCKET_ARGS);
      else hipLaunchKernelGGL(k_anneal_bucket, dim3(nb), dim3(BK_AT), 0, st, BUCKET_ARGS);
#undef BUCKET_ARGS
      c->bk_steps++;
      c->head_offsets_done = (go.child_off != nullptr); c->head_offsets_bucket = true;
      c->scount_B = ba.B; c->scount_buf = ba.kb ? c->head_kb_use : -1; c->scount_pos = c->pos_flip;
    } else if (c->psit_on) { c->scount_B = 0; if (items <= 2) ANNEAL_LAUNCH_(2, 1); else ANNEAL_LAUNCH_(3, 1); }      // hf_to_psit: everything outside C(T); the C(T) segment is finished below
    else if (anneal_split_ok(c, p, mode, nall, items, go.child_off != nullptr, use_mail)) {
      // long lists: fold + round + a compaction inside every tile, a scan of the tiles' counts, then every tile to its place: no tile waits
      // for the slowest tile in front of it (18.7 of a tile's 44 us at 10^6 walkers, tools/anneal_prof.py)
      c->scount_B = 0;
      const AnnealStage sg = c->stage;
      if (t_anneal >= 0) hipEventRecord(c->ev0[t_anneal], st);
      if (items == 3) hipLaunchKernelGGL((k_anneal<3, 0, 1>), dim3(nb), dim3(TPB), 0, st, ANNEAL_ARGS, sg);
      else hipLaunchKernelGGL((k_anneal<4, 0, 1>), dim3(nb), dim3(TPB), 0, st, ANNEAL_ARGS, sg);
      hipLaunchKernelGGL(k_anneal_split_scan, dim3((nb + 4 * TPB - 1) / (4 * TPB)), dim3(TPB), 0, st, (const u64 *)sg.cnt_a, (const u64 *)sg.cnt_b, sg.off_a, sg.off_b, (int)nb, c->d_sc);
      if (items == 3) hipLaunchKernelGGL(k_anneal_place<3>, dim3(nb), dim3(TPB), 0, st, sg, c->m, c->d_loc_imp, (const u64 *)c->d_ct_hkey, (const u32 *)c->d_ct_hidx, c->ct_mask, (const double *)c->d_ct_num, (const double *)c->d_ct_den, c->d_partials, p, nb, go, (const DevScalars *)c->d_sc);
      else hipLaunchKernelGGL(k_anneal_place<4>, dim3(nb), dim3(TPB), 0, st, sg, c->m, c->d_loc_imp, (const u64 *)c->d_ct_hkey, (const u32 *)c->d_ct_hidx, c->ct_mask, (const double *)c->d_ct_num, (const double *)c->d_ct_den, c->d_partials, p, nb, go, (const DevScalars *)c->d_sc);
      if (t_anneal >= 0) hipEventRecord(c->ev1[t_anneal], st);
      c->head_offsets_done = true; c->head_offsets_bucket = false;
    }
    else { c->scount_B = 0; if (items == 1) ANNEAL_LAUNCH(1); else if (items == 2) ANNEAL_LAUNCH(2); else if (items == 3) ANNEAL_LAUNCH(3); else ANNEAL_LAUNCH(4);
           c->head_offsets_done = (go.child_off != nullptr); c->head_offsets_bucket = false; }
#undef ANNEAL_LAUNCH_
#undef ANNEAL_LAUNCH
#undef ANNEAL_ARGS
    if (fuse_gate && go.keys == c->d_keys_alt) { std::swap(c->d_keys, c->d_keys_alt); if (!c->pack) std::swap(c->d_vals, c->d_vals_alt); }      // the next step's spawn kernel appends its keys behind the walkers'
    std::swap(c->w.up, c->m.up); std::swap(c->w.dn, c->m.dn); std::swap(c->w.wt, c->m.wt); std::swap(c->w.flg, c->m.flg);
    std::swap(c->w.me, c->m.me); std::swap(c->w.en, c->m.en); std::swap(c->w.ed, c->m.ed); std::swap(c->w.irk, c->m.irk);
    if (c->psit_on) {          // do_walk.f90:2394-2462, 2487, 2590-2598, 2701-2722 on the C(T) segment of the NEW list
      TBEG(psit_fin, st);
      hipLaunchKernelGGL(k_psit_finish, dim3(PSIT_FB), dim3(TPB), 0, st, c->psit, c->w.wt, c->w.flg, p, c->d_ps_part, go, seed);
      TEND(psit_fin, st);
    }
  } else {
    TBEG(merge, st);
    hipLaunchKernelGGL(k_merge, dim3(nbm), dim3(TPB), 0, st, c->w, c->m, skey, perm, c->d_flags, c->d_wabs_part, n0, nall, p, c->invalid_key, c->pack);
    device_excl_scan_u64(c->d_flags, c->d_pos, nall, &c->d_sc->tot1, sw[1], st);
    TEND(merge, st);
    TBEG(round, st);
    const bool serial_join = getenv("SQMC_SERIAL_JOIN") != nullptr;      // read per call: the one-lane k_join (REPLAY's kernel) on a COUNTER walk, for tests
    if (mode == SQMC_RNG_COUNTER && !serial_join) {
      // the arrays of the rounding pass behind it are idle: the chunks' candidates (|w|, draw key | index) and their counts go there
      const unsigned nchunks = (unsigned)((nall + JP_TILE - 1) / JP_TILE);
      hipLaunchKernelGGL(k_join_gather, dim3(nchunks), dim3(TPB), 0, st, c->m, c->d_flags, c->d_pos, nall, p, (double *)c->d_flags2, c->d_pos2, c->d_jcnt);
      hipLaunchKernelGGL(k_join_par, dim3(2), dim3(JP_T), 0, st, c->m, (const double *)c->d_flags2, (const u64 *)c->d_pos2, (const u32 *)c->d_jcnt, nall, p, seed, step);
    }
    else hipLaunchKernelGGL(k_join, dim3(1), dim3(TPB), 0, st, c->m, c->d_flags, c->d_pos, nall, p, mode, seed, step, c->d_sc);
    hipLaunchKernelGGL(k_round, dim3(nblk(nall)), dim3(TPB), 0, st, c->m, c->d_flags, c->d_pos, c->d_flags2, nall, p, mode, seed, step, c->d_sc, skey, c->pack);
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
  fa.n_ftiles = n_ft; fa.on = 1; fa.n_tickets = 3; fa.n_children = -1; fa.red = nullptr;
  fa.partials2 = (c->psit_on && p.semi) ? c->d_ps_part : (const double *)nullptr; fa.nblocks2 = (c->psit_on && p.semi) ? PSIT_FB : 0;
  fa.expect_nimp = !use_mail ? (p.semi ? c->n_imp_local : -2) : -1;      // sharded in-library step: 'locations of my imp broken' must reach every rank
  if (fuse_gate && use_mail) {       // the finishing block runs beside the next head's scan (look-back set scan_flip): it re-zeroes the other set only
    const int other = c->scan_flip ^ 1;
    fa.scan_state = c->d_scan_state + (long long)other * c->cap_tiles; fa.scan_ticket = c->d_scan_ticket + other;
    fa.n_scan_words = c->scan_used[other]; fa.n_tickets = 1; c->scan_used[other] = 0;
    fa.n_children = (nall - n0) / (c->dev.hb.on ? 2 : 1);      // children, not walker slots
  }
  if (!use_mail && c->head_offsets_done) fa.n_children = c->shard_nch;      // the bucket tail has written the NEXT step's child count over this step's already
  const bool fin_in_gate = c->pipeline_next && p.semi && use_mail;     // the next step's gate kernel does the final sums in its first block
  // Sharded in-library step: do_walk.f90:2778-2790, the sums every rank needs.  In a pipelined run with one communicator the next
  // step's deterministic weights are all-reduced straight behind them (enqueue_head): one call carries both, and the kernel that
  // gathers the weights does the final sums in its first block (decided by quantities all ranks share)
  static const bool two_calls = getenv("SQMC_SHARD_SPLIT_ALLREDUCE") != nullptr;
  c->head_sums_ride = !use_mail && c->pipeline_next && c->d_grow && !c->comm2 && c->n_imp > 0 && p.semi && !two_calls;
  TBEG(estimate, st);
  if (c->head_sums_ride) c->head_fin = fa;
  else if (!fin_in_gate) hipLaunchKernelGGL(k_finish, dim3(1), dim3(TPB), 0, st, fa, c->d_sc);
  TEND(estimate, st);
  HIPCHK(hipGetLastError());
  bool mail_in_gate = false;
  if (!use_mail) {
    if (!c->head_sums_ride) { int rr = comm_allreduce_stats(c); if (rr) return rr; }
    mail_in_gate = c->pipeline_next && p.semi;                 // the next step's gate kernel posts them (one launch less)
    if (!mail_in_gate) hipLaunchKernelGGL(k_post_mail, dim3(1), dim3(64), 0, st, c->d_sc, c->d_mail, seq);
    else { memset(&fa, 0, sizeof(fa)); fa.on = 3; fa.mail = c->d_mail; fa.seq = seq; fa.expect_nimp = -1; if (c->head_sums_ride) fa.red = c->d_xg + c->n_imp; }
  }
  c->tail_fills_hii = bucket && c->pipeline_next;
  if (c->pipeline_next) {
    // the next step's gate + scan + spawn go out now, behind k_finish: the GPU runs on while the host
    // reads this step's sums and does its population control.  nall bounds the new walker count.
    c->pipeline_next = false;
    int rh = enqueue_head(c, p, step + 1, nall, true, c->timing >= 2 ? c->hev[0] : nullptr, c->timing >= 2 ? c->hev[1] : nullptr,
                          spawn_events_on(c, step + 1, p.semi) ? c->hev[2] : nullptr, spawn_events_on(c, step + 1, p.semi) ? c->hev[3] : nullptr, &c->head_cseq, (fin_in_gate || mail_in_gate) ? &fa : nullptr, fuse_gate);
    if (rh) return rh;
    c->head_ready = true; c->head_p = p;
  }
  {
    int wr = wait_mail(&c->h_mail->seq, seq, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) {            // stream drained without the mail: read the scalars the slow way
      HIPCHK(hipMemcpy(c->h_sc, c->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost));
    } else { c->h_sc->tot2 = c->h_mail->tot2; c->h_sc->err = (int)c->h_mail->err; for (int i = 0; i < 16; i++) c->h_sc->stats[i] = c->h_mail->stats[i];
             c->h_sc->retry = (int)c->h_mail->retry; c->h_sc->bk_fill = (unsigned int)c->h_mail->bk_fill; }
  }
  const bool stop_now = !use_mail && c->h_sc->err != 0;      // a collective stop outranks a re-run: every rank returns the status, none reduces again
  if (bucket && !use_mail && !stop_now && (c->h_sc->retry & 2) && !(c->h_sc->retry & 1)) {
    // In-library sharded step: ANOTHER rank's bucket tail gave up.  The sums that were just all-reduced contain its unfinished ones:
    // every rank reduces once more after that rank has re-run its tail -- this rank contributes the same local sums again (redl).
    const u64 tot2_first = c->h_sc->tot2; double loc[16]; for (int i = 0; i < 16; i++) loc[i] = c->h_sc->stats[i];
    int rr = comm_allreduce_stats(c); if (rr) return rr;
    const u64 seq2 = ++c->mail_seq;
    hipLaunchKernelGGL(k_post_mail, dim3(1), dim3(64), 0, st, c->d_sc, c->d_mail, seq2);
    int wr = wait_mail(&c->h_mail->seq, seq2, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) HIPCHK(hipMemcpy(c->h_sc, c->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost));
    else { c->h_sc->err = (int)c->h_mail->err; for (int i = 0; i < 7; i++) c->h_sc->stats[i] = c->h_mail->stats[i]; }
    for (int i = 7; i < 16; i++) c->h_sc->stats[i] = loc[i];          // the local figures and the walker counts are this rank's own, from its own (valid) tail:
    c->h_sc->tot2 = tot2_first; c->h_sc->retry = 0;                    // the head enqueued behind it has cleared the device copies since
    c->shard_y_ok = false; c->shard_x_ready = false;                   // ... and its projection used weights the other rank had not finished: the step redoes it
    { static const int hold = getenv("SQMC_BUCKET_HOLDOFF") ? atoi(getenv("SQMC_BUCKET_HOLDOFF")) : 8; c->bk_holdoff = hold; }
  }
  if (bucket && !stop_now) {
    if (c->h_sc->retry & 1) {
      // A bucket outgrew its block.  The kernel wrote only the other walker buffer and scratch; the head of the next step,
      // if it was enqueued, saw the flag and did nothing.  Undo the host's bookkeeping and let the caller run the radix tail.
      drop_head(c);
      hipStreamSynchronize(st);
      hipMemset(c->d_fstate, 0, 2 * c->cap_ftiles * 8); hipMemset(c->d_fticket, 0, 4);
      hipMemset(&c->d_sc->retry, 0, sizeof(int)); hipMemset(&c->d_sc->bk_fill, 0, sizeof(unsigned int));
      c->h_sc->retry = 0;
      std::swap(c->w.up, c->m.up); std::swap(c->w.dn, c->m.dn); std::swap(c->w.wt, c->m.wt); std::swap(c->w.flg, c->m.flg);
      std::swap(c->w.me, c->m.me); std::swap(c->w.en, c->m.en); std::swap(c->w.ed, c->m.ed); std::swap(c->w.irk, c->m.irk);
      if (fuse_gate) std::swap(c->d_keys, c->d_keys_alt);
      c->pipeline_next = false; c->bk_retries++; c->head_offsets_done = false; c->shard_y_ok = false; c->shard_x_ready = false;
      // The boundaries are learnt anew -- from this attempt: every bucket left its count of spawns, also the ones that gave up (with
      // equal-residents boundaries the spawn-rich key ranges hold 4x the mean), and the list they were counted in is the input again.
      static const bool no_learn = getenv("SQMC_BUCKET_NO_RETRY_LEARN") != nullptr;
      const int B_failed = c->scount_B, set_failed = c->scount_buf;
      bool learnt = false;
      if (!no_learn && !c->d_grow && c->comm == nullptr && B_failed > 0 && B_failed <= BK_REBAL_MAXB && n0 >= 16ll * B_failed && !ba.force_retry) {
        BucketArgs rb; memset(&rb, 0, sizeof(rb));
        int out = 0; while (out == set_failed) out++;
        rb.B = B_failed; rb.scount = c->d_bscount;
        rb.kb_prev = set_failed >= 0 ? c->d_bkb + set_failed * (BK_MAXB + 1) : (const u32 *)nullptr;
        rb.pos_prev = set_failed >= 0 ? c->d_bpos + c->scount_pos * (BK_MAXB + 1) : (const u32 *)nullptr;
        rb.kb_out = c->d_bkb + out * (BK_MAXB + 1); rb.hint_out = c->d_bhint + out * (BK_MAXB + 1);
        hipLaunchKernelGGL(k_bucket_boundaries, dim3(1), dim3(BK_T), 0, st, (const u64 *)c->d_keys, n0, rb);
        c->kb_B[0] = c->kb_B[1] = c->kb_B[2] = 0; c->kb_B[out] = B_failed; c->kb_next = out; learnt = true;
      } else { c->kb_B[0] = c->kb_B[1] = c->kb_B[2] = 0; c->kb_next = -1; }
      c->scount_B = 0; c->scount_buf = -1;
      { static const int hold = getenv("SQMC_BUCKET_HOLDOFF") ? atoi(getenv("SQMC_BUCKET_HOLDOFF")) : 8; c->bk_holdoff = (learnt && set_failed < 0) ? 1 : hold; }
      return SQMC_INTERNAL_RETRY;
    }
    if (c->h_sc->bk_fill > 850) c->bk_holdoff = 4;         // thin head-room: radix tail for a few steps
  }
  c->timers_pending = kernel_events_on(c, step);
  c->step_no++;
  if (c->h_sc->err) {          // raised on the device (sharded in-library steps: by any rank, all-reduced)
    drop_head(c);
    const int e = c->h_sc->err;
    return fail(e, e == SQMC_ERR_NEG_DIAG ? "diagonal_factor<0 after target population has been reached" : e == SQMC_ERR_IMP_BROKEN ? "locations of my imp broken" :
                   e == SQMC_ERR_MWALK ? "nwalk>MWALK" : "step stopped on the device");
  }
  const long long nfinal = (long long)(c->h_sc->tot2 & 0xFFFFFFFFull), nimp = (long long)(c->h_sc->tot2 >> 32);
  c->nwalk = nfinal; c->residents_sorted = true;
  if (bucket && p.semi) { c->xs_cur ^= 1; c->xs_valid = true; } else c->xs_valid = false;
  c->side_pending = false;
  for (int i = 0; i < 16; i++) out[i] = c->h_sc->stats[i];
  c->last_wabs = out[1];
  if (nfinal == 0) { drop_head(c); return fail(SQMC_ERR_NO_WALKERS, "my_nwalk=0"); }
  if (p.semi && nimp != (c->shard_n > 1 || c->d_grow ? c->n_imp_local : c->n_imp)) { drop_head(c); return fail(SQMC_ERR_IMP_BROKEN, "locations of my imp broken"); }
  return SQMC_OK;
}

static int step_tail(sqmc_gpu_ctx *c, const StepP &p, long long n0, long long nall, bool join_side_stream, double out[16]) {
  int r = step_tail_impl(c, p, n0, nall, join_side_stream, out, true);
  if (r == SQMC_INTERNAL_RETRY) r = step_tail_impl(c, p, n0, nall, false, out, false);
  return r;
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
  if (c->psit_on) {
    if (!sp->semistochastic) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit needs a semistochastic step");
    if (c->dev.hb.on) return fail(SQMC_ERR_UNSUPPORTED, "hf_to_psit with proposal_method fast_heatbath is not built");
    p.koff = c->dev.ps.koff; p.nct = c->dev.ps.n_ct;
    if (c->nwalk < p.nct) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit: the walker list does not hold the C(T) segment");
  }
  const long long n0 = c->nwalk, M = c->mwalk;
  const int mode = c->rng_mode; const u64 seed = c->seed64, step = c->step_no;
  // a host that calls step by step (the reference's own loop does work between steps) and has promised to come back with the
  // same tau and cutoff (sqmc_gpu_set_chained_runs): this step enqueues the next one's head, as the steps of sqmc_gpu_run do
  if (!c->in_run) c->pipeline_next = c->chained_runs && sp->reached_w_abs_gen == 2 && mode != SQMC_RNG_REPLAY && !getenv("SQMC_NO_PIPELINE");
  { static const bool psit_no_pipe = getenv("SQMC_PSIT_NO_PIPELINE") != nullptr;      // hf_to_psit: the head (gate, scan, spawn) behind the tail like any other step's; the gate stays a kernel of its own
    if (c->psit_on && psit_no_pipe) c->pipeline_next = false; }
  collect_timers(c);
  c->nt = 0;
  hipStream_t st2 = c->st2;
  int t_gate_scan = -1, t_spawn = -1;
  if (c->timing >= 2 && c->nt < NTIMERS) { t_gate_scan = c->nt++; c->tname[t_gate_scan] = "gate_scan"; }
  if (spawn_events_on(c, step, p.semi) && c->nt < NTIMERS) { t_spawn = c->nt++; c->tname[t_spawn] = "spawn"; }
  u64 cseq;
  if (c->head_ready) {                 // chained runs: the caller came back with other parameters than it left with
    const StepP &h0 = c->head_p;
    if (h0.tau != p.tau || h0.cutoff != p.cutoff || h0.semi != p.semi || h0.cti != p.cti || mode == SQMC_RNG_REPLAY) abandon_head(c);
  }
  const bool from_head = c->head_ready;
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
                       c->d_child_state, n0, (M - n0) / (c->dev.hb.on ? 2 : 1), p, c->d_sc, c->dev);
    if (t_gate_scan >= 0) hipEventRecord(c->ev1[t_gate_scan], st);
    HIPCHK(hipEventRecord(c->e_fork, st)); c->fork_valid = true;
    cseq = ++c->cnt_seq;
    if (M > n0) {
      if (t_spawn >= 0)
        SPAWN_LAUNCH_EXT(c->dev.hb.on, 0, dim3(nblk(M - n0)), dim3(TPB), 0, st, c->ev0[t_spawn], c->ev1[t_spawn], 0, c->dev, c->w, c->d_child_off, c->d_wchild,
                              c->d_child_state, c->d_keys, c->d_vals, n0, M, p, mode, seed, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, cseq, c->pack, 0, OwnerOut{nullptr, nullptr, 0, 0}, BucketArgs{}, FinArgs{}, PrjPre{}, 0);
      else
        SPAWN_LAUNCH(c->dev.hb.on, 0, dim3(nblk(M - n0)), dim3(TPB), 0, st, c->dev, c->w, c->d_child_off, c->d_wchild, c->d_child_state, c->d_keys, c->d_vals,
                           n0, M, p, mode, seed, step, c->invalid_key, (const DevScalars *)c->d_sc, c->d_mail, cseq, c->pack, 0, OwnerOut{nullptr, nullptr, 0, 0}, BucketArgs{}, FinArgs{}, PrjPre{}, 0);
    } else if (t_spawn >= 0) { hipEventRecord(c->ev0[t_spawn], st); hipEventRecord(c->ev1[t_spawn], st); }
  } else {
    int r = enqueue_head(c, p, step, n0, false, t_gate_scan >= 0 ? c->ev0[t_gate_scan] : nullptr, t_gate_scan >= 0 ? c->ev1[t_gate_scan] : nullptr,
                         t_spawn >= 0 ? c->ev0[t_spawn] : nullptr, t_spawn >= 0 ? c->ev1[t_spawn] : nullptr, &cseq);
    if (r) return r;
  }
  // ---- death/clone and the deterministic projection: side streams beside spawn + sort -- or nothing at all here when the
  //      bucket tail is going to do them itself (pipelined steps whose missing H_ii the head already filled and whose
  //      deterministic weights the last bucket tail left row by row)
  {
    static const bool no_fused_side = getenv("SQMC_NO_FUSED_SIDE") != nullptr;
    c->side_pending = !no_fused_side && from_head && c->head_hii && c->head_y_done && c->xs_valid && p.semi && bucket_static_ok(c, p) && c->bk_holdoff == 0 && c->head_ba.B > 0;
    c->head_hii = false;
    if (!c->side_pending) c->head_hii_joined = false;
    if (!c->side_pending) { int r = launch_side_kernels(c, p, n0, false); if (r) return r; }
  }
  // ---- the child count, from the mailbox (or the slow way when there was no k_spawn launch)
  long long nch;
  if (M > n0) {
    int wr = wait_mail(&c->h_mail->cnt_seq, cseq, st);
    if (wr > 0) return fail(SQMC_ERR_HIP, std::string("step failed on the device: ") + hipGetErrorString((hipError_t)wr));
    if (wr < 0) { u64 v; HIPCHK(hipMemcpy(&v, &c->d_sc->n_children, 8, hipMemcpyDeviceToHost)); nch = (long long)v; }
    else nch = (long long)c->h_mail->n_children;
  } else { u64 v; HIPCHK(hipMemcpyAsync(&v, &c->d_sc->n_children, 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); nch = (long long)v; }
  const long long spc = c->dev.hb.on ? 2 : 1;          // walker slots per child (fast_heatbath: two)
  if (n0 + spc * nch > M) {
    hipStreamSynchronize(st); hipStreamSynchronize(st2); hipStreamSynchronize(c->st3);
    hipMemset(c->d_scan_state, 0, 3 * c->cap_tiles * 8); hipMemset(c->d_scan_ticket, 0, 3 * 4);
    return fail(SQMC_ERR_MWALK, "nwalk>MWALK");
  }
  const long long nall = n0 + spc * nch;
  return step_tail(c, p, n0, nall, !(c->side_pending && c->head_hii_joined), out);       // nothing ran on the side streams when the tail does that work itself and the H_ii came out of k_spawn
}

int sqmc_gpu_shard_step(sqmc_gpu_ctx *c, const sqmc_step_params *sp, double out[16]);
typedef int (*step_fn)(sqmc_gpu_ctx *, const sqmc_step_params *, double *);
static int run_steps(sqmc_gpu_ctx *c, sqmc_popctl *pc, int64_t nsteps, double *stats, double totals[16], step_fn one_step) {
  if (!c || !pc || !totals || nsteps < 0) return fail(SQMC_ERR_BAD_ARG, "bad argument");
  for (int k = 0; k < 16; k++) totals[k] = 0.0;
  for (int k = 0; k < 4; k++) { c->slow_us[k] = 0.0; c->slow_step[k] = -1; }
  struct InRun { sqmc_gpu_ctx *c; InRun(sqmc_gpu_ctx *x) : c(x) { c->in_run = true; } ~InRun() { c->in_run = false; } } in_run_guard(c);
  struct timespec ts_prev; clock_gettime(CLOCK_MONOTONIC, &ts_prev);
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
    c->pipeline_next = ((one_step == (step_fn)sqmc_gpu_step || (one_step == (step_fn)sqmc_gpu_shard_step && c->comm != nullptr && !getenv("SQMC_SHARD_NO_PIPELINE"))) && (it + 1 < nsteps || (c->chained_runs && one_step == (step_fn)sqmc_gpu_step)) &&
                        pc->reached_w_abs_gen == 2 && c->rng_mode != SQMC_RNG_REPLAY && !getenv("SQMC_NO_PIPELINE"));
    double out[16];
    int r = one_step(c, &sp, out);
    c->pipeline_next = false;
    if (r) { drop_head(c); return r; }
    if (stats) memcpy(stats + it * 16, out, sizeof(out));
    for (int k = 0; k < 16; k++) totals[k] += out[k];
    {
      struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
      double us = 1e6 * (double)(ts.tv_sec - ts_prev.tv_sec) + 1e-3 * (double)(ts.tv_nsec - ts_prev.tv_nsec); long long wh = it; ts_prev = ts;
      for (int k = 0; k < 4; k++) if (us > c->slow_us[k]) { std::swap(us, c->slow_us[k]); std::swap(wh, c->slow_step[k]); }
    }
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
#ifdef BUCKET_PROF
extern "C" int sqmc_gpu_debug_buckets(sqmc_gpu_ctx *c, unsigned int *kb3, unsigned int *pos, unsigned int *scount, int *state) {
  hipStreamSynchronize(c->st);
  hipMemcpy(kb3, c->d_bkb, 3 * (BK_MAXB + 1) * 4, hipMemcpyDeviceToHost); hipMemcpy(pos, c->d_bpos, (BK_MAXB + 1) * 4, hipMemcpyDeviceToHost);
  hipMemcpy(scount, c->d_bscount, (BK_MAXB + 1) * 4, hipMemcpyDeviceToHost);
  state[0] = c->kb_next; state[1] = c->scount_buf; state[2] = c->head_kb_use; state[3] = c->scount_B; state[4] = c->kb_B[0]; state[5] = c->kb_B[1]; state[6] = c->kb_B[2];
  return 0;
}
#endif
int sqmc_gpu_set_chained_runs(sqmc_gpu_ctx *c, int32_t on) {
  if (!c) return fail(SQMC_ERR_BAD_ARG, "null argument");
  c->chained_runs = on != 0;
  if (!on) abandon_head(c);
  return SQMC_OK;
}
int sqmc_gpu_slowest_steps(sqmc_gpu_ctx *c, double us[4], int64_t step[4]) {
  if (!c || !us || !step) return fail(SQMC_ERR_BAD_ARG, "null argument");
  for (int k = 0; k < 4; k++) { us[k] = c->slow_us[k]; step[k] = c->slow_step[k]; }
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
  u64 key = invalid_key;
  if (r.wt != 0.0) { key = det_key(dev, r.up, r.dn); if (dev.ps.koff) key = psit_key(dev.ps, key); }
  put_key(keys, vals, k, key, pack);
}

// The second half of a step on its own: the caller's spawned walkers (creation order) are appended
// behind the resident walkers, then sort -> merge_original_with_spawned2 -> reduce_my_walker ->
// estimator sums run as in sqmc_gpu_step.  do_walk.f90:2364-2487 as one call; also the door the
// parity tests use to put hand-built collision cases through k_merge.
int sqmc_gpu_annihilate(sqmc_gpu_ctx *c, const sqmc_step_params *sp, int64_t n_spawn, const uint64_t *up, const uint64_t *dn, const double *wt,
                        const int8_t *impd, const int8_t *init, double out[16]) {
  abandon_head(c);
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
  if (c->psit_on) { if (!sp->semistochastic) return fail(SQMC_ERR_BAD_ARG, "hf_to_psit needs a semistochastic step"); p.koff = c->dev.ps.koff; p.nct = c->dev.ps.n_ct; }
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

// Debugging aid, not part of the ABI in include/: the list the last semistochastic step (radix tail) stood on in front of its merge.
// Resident weights after death/clone and the projection (the buffer the annihilation kernel read: the OTHER one now) and the step's
// spawn records in creation order (weight 0: no walker).
int sqmc_gpu_debug_premerge(sqmc_gpu_ctx *c, int64_t cap, int64_t *n0, int64_t *n_spawn, double *res_wt, uint64_t *sp_up, uint64_t *sp_dn, double *sp_wt, int8_t *sp_impd, int8_t *sp_init) {
  if (!c || !n0 || !n_spawn) return SQMC_ERR_BAD_ARG;
  hipStreamSynchronize(c->st);
  *n0 = c->dbg_n0; *n_spawn = c->dbg_nall - c->dbg_n0;
  if (cap < c->dbg_nall) return SQMC_OK;
  if (hipMemcpy(res_wt, c->m.wt, c->dbg_n0 * 8, hipMemcpyDeviceToHost) != hipSuccess) return SQMC_ERR_HIP;
  std::vector<SpawnRec> r(*n_spawn);
  if (*n_spawn > 0 && hipMemcpy(r.data(), c->w.sp, *n_spawn * sizeof(SpawnRec), hipMemcpyDeviceToHost) != hipSuccess) return SQMC_ERR_HIP;
  for (long long i = 0; i < *n_spawn; i++) { sp_up[i] = r[i].up; sp_dn[i] = r[i].dn; sp_wt[i] = r[i].wt; sp_impd[i] = (int8_t)flg_impd((u32)r[i].flg); sp_init[i] = (int8_t)flg_init((u32)r[i].flg); }
  return SQMC_OK;
}

#include "abi_shard.inc"
#include "abi_doors.inc"
#include "davidson.inc"
#include "heatbath_setup.inc"

}  // extern "C"
