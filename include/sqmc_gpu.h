/*
 * sqmc_gpu.h -- C ABI of libsqmc_gpu.so: the MI355X (gfx950) implementation of sqmc's
 * semistochastic walker step, deterministic-core matvec and HCI connection generation.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A Fortran
 * host binds it with iso_c_binding (sqmc_amd/fortran/sqmc_gpu_mod.f90; INTEGRATION.md shows
 * the call sites to patch in the reference).  Each entry point cites the reference
 * interface it replaces, file:line relative to the reference's src/.
 *
 * Conventions (identical to the reference):
 *   - determinants: one 64-bit word per spin (n_det_words = 1, norb <= 64 this round); bit k
 *     <-> orbital k+1; walkers sorted by (up, then dn) as unsigned integers.  The reference
 *     splits its 128-bit dets into 2 x int64 on the wire (mpi_routines.f90:671-680); the
 *     low word is what is passed here, the high word must be zero.
 *   - CSR of the symmetric projector/Hamiltonian: "upper triangular" storage of
 *     more_tools.f90:3622-3670 -- row_counts(n), 1-based int64 column indices, fp64 values;
 *     every stored (i,m) with i != m is applied to both y(i) and y(m).
 *   - weights fp64; initiator in {0,1,2,3}; imp_distance in {-2,-1,0,1..127} as int8.
 *   - every function returns 0 on success; >0 are the reference's own stop conditions
 *     (see SQMC_ERR_*), <0 are library/HIP failures (text via sqmc_gpu_last_error()).
 *   - one context per process/GPU, driven by one host thread (the reference is
 *     single-threaded per MPI rank).
 */
#ifndef SQMC_GPU_H
#define SQMC_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sqmc_gpu_ctx sqmc_gpu_ctx;

/* status codes > 0 mirror the reference's stop statements */
#define SQMC_OK 0
#define SQMC_ERR_MWALK 1          /* 'nwalk>MWALK'                    do_walk.f90:3684-3690 */
#define SQMC_ERR_SPAWN_OVERFLOW 2 /* spawn buffer full                mpi_routines.f90:2494 */
#define SQMC_ERR_NEG_DIAG 3       /* diagonal_factor<0 after equil.   do_walk.f90:3788     */
#define SQMC_ERR_NO_WALKERS 4     /* 'my_nwalk=0'                     do_walk.f90:2489     */
#define SQMC_ERR_IMP_BROKEN 5     /* 'locations of my imp broken'     do_walk.f90:2204     */
#define SQMC_ERR_BAD_ARG -1
#define SQMC_ERR_HIP -2
#define SQMC_ERR_UNSUPPORTED -3

/* RNG discipline.  REPLAY reproduces the reference's single rannyu stream draw for draw
 * (rannyu.f90:54-74 consumed in walker order, do_walk.f90:3577-3583, chemistry.f90:4391-4439,
 * do_walk.f90:7222-7230) and is meant for fixed-seed parity runs; COUNTER keys an
 * independent 48-bit stream on (seed, step, stage, entity) -- the walker index for the spawn
 * gate, the child index for a proposal, the determinant itself for the rounding draw -- so
 * that every stage is fully parallel and none needs a rank among the others.  Both give
 * reals k/2^48 like rannyu. */
#define SQMC_RNG_REPLAY 0
#define SQMC_RNG_COUNTER 1

/* Tables the reference builds in system_setup_chem / read_integrals / setup_orb_by_symm
 * (chemistry.f90:299-537, 538-869, 2461-2527) and keeps in module chemistry. */
typedef struct {
  int32_t norb, nup, ndn, n_core_orb;
  int32_t time_sym, z;              /* chemistry.f90:163-181 (walk kernels need time_sym=0) */
  int32_t n_group;                  /* order of the abelian point group (<= 8)             */
  const int32_t *product_table;     /* [9*9], 1-based: product_table(i,j) at [i*9+j]       */
  const int32_t *orbital_symmetries;/* [norb+1], 1-based, after orbital reordering         */
  const int32_t *combine_2;         /* [(norb+2)*(norb+2)], 1-based, chemistry.f90:856-866 */
  int64_t n_integrals;              /* integral_index(norb+1,...,norb+1)                   */
  const double *integrals;          /* [n_integrals+1], 1-based packed array               */
  int32_t rng_mode;                 /* SQMC_RNG_*                                          */
  int32_t irand_seed[4];            /* stream-2 seed of input line 1, do_walk.f90:231,646  */
  int64_t mwalk;                    /* MWALK: capacity of the walker arrays                */
} sqmc_chem_cfg;

/* one process per GPU: select the device of this rank before creating a context (the
 * reference's cluster_init, mpi_routines.f90:766, has no device notion). */
int sqmc_gpu_set_device(int device);

/* replaces: system_setup_chem + init_move table setup.  Copies everything to HBM. */
int sqmc_gpu_init_chem(const sqmc_chem_cfg *cfg, sqmc_gpu_ctx **out);
/* Homogeneous electron gas in a plane-wave basis (hamiltonian_type 'heg'): the tables
 * system_setup_heg / generate_k_vectors build (heg.f90:168-215, 643-749).  k_vectors is the
 * reference's k_vectors(n_dim, norb) in its own (column-major) memory order, orbitals sorted by
 * |k| exactly as the reference sorts them; the walk then uses off_diagonal_move_heg
 * (heg.f90:1344-1598) and hamiltonian_heg (heg.f90:845-1011) in place of the chemistry pair. */
typedef struct {
  int32_t n_dim, norb, nup, ndn;
  double length_cell;
  const double *k_vectors;          /* [norb * n_dim] */
  int32_t rng_mode;
  int32_t irand_seed[4];
  int64_t mwalk;
} sqmc_heg_cfg;
int sqmc_gpu_init_heg(const sqmc_heg_cfg *cfg, sqmc_gpu_ctx **out);
/* Real-space Hubbard model on an l_x by l_y square lattice (hamiltonian_type 'hubbard2'): the
 * scalars read_hubbard takes from the deck (hubbard.f90:138-382: l_x, l_y, pbc, t, U, nup, ndn).
 * Site s (1-based) sits at x = mod(s-1, l_x)+1, y = (s-1)/l_x+1 and bit s-1 of a determinant
 * word; neighbours follow get_nbr (more_tools.f90:223-355).  The walk then uses
 * off_diagonal_move_hubbard (hubbard.f90:2992-3120) and hamiltonian_hubbard (1536-1644) as the
 * operator pair.  The energy estimator is the library's usual one over a determinant-list trial
 * wavefunction (the 'else' branch of energy_pieces_hubbard, 4514-4527, as a C(T) table); the
 * Gutzwiller/Slater trial functions of hubbard.f90 are host-side code outside this path.
 * A periodic direction of length 2 is refused: the reference's connected list counts that bond
 * twice while hamiltonian_hubbard counts it once. */
typedef struct {
  int32_t l_x, l_y, pbc, nup, ndn;
  double t, U;
  int32_t rng_mode;
  int32_t irand_seed[4];
  int64_t mwalk;
} sqmc_hubbard_cfg;
int sqmc_gpu_init_hubbard(const sqmc_hubbard_cfg *cfg, sqmc_gpu_ctx **out);
int sqmc_gpu_finalize(sqmc_gpu_ctx *ctx);
const char *sqmc_gpu_last_error(void);

/* proposal_method 'fast_heatbath': the tables setup_efficient_heatbath builds when run_type /= hci (chemistry.f90:1002-1225),
 * handed over exactly as the reference holds them (module variables of chemistry.f90:52-75; Fortran column-major order, the
 * four-index arrays and their alias tables 1-D and single precision, indexed by same_index / opposite_index, 9154-9193).
 * After this call sqmc_gpu_step spawns with off_diagonal_move_chem_efficient_heatbath (5086-5347) instead of
 * off_diagonal_move_chem: every child may add TWO walkers (a single excitation larger than all its doubles together comes
 * back with a double, do_walk.f90:3604-3611 -> add_walker 7584-7697), so a step needs room for 2 x children spawned walkers.
 * Both RNG disciplines (REPLAY: one lane runs every proposal to follow the stream).  Systems for which check_heatbath_unbiased (9330-9375) fails are the caller's to refuse, as the
 * reference does (1215-1223).  tests/golden/README_heatbath.md records what parity is defined against. */
typedef struct {
  int32_t norb, reserved;
  const double *one_orbital_probabilities;                 /* (norb) */
  const double *two_orbital_probabilities;                 /* (2 norb, 2 norb) */
  const double *three_orbital_probabilities_same_spin;     /* (norb, norb, norb), normalised over the last index */
  const double *three_orbital_probabilities_opposite_spin;
  const int32_t *J_three_orbital_probabilities_same_spin, *J_three_orbital_probabilities_opposite_spin;   /* alias tables, (norb, norb, norb) */
  const double *q_three_orbital_probabilities_same_spin, *q_three_orbital_probabilities_opposite_spin;
  int64_t size_same, size_opposite;                        /* same_index(norb,norb,norb,norb), opposite_index(norb,norb,norb,norb) */
  const float *four_orbital_probabilities_same_spin, *four_orbital_probabilities_opposite_spin;
  const int32_t *J_four_orbital_probabilities_same_spin, *J_four_orbital_probabilities_opposite_spin;
  const float *q_four_orbital_probabilities_same_spin, *q_four_orbital_probabilities_opposite_spin;
  const double *Htot_same;                                 /* (combine_2_indices(norb, norb), norb) */
  const double *Htot_opposite;                             /* (norb, norb, norb) */
} sqmc_heatbath_tables;
int sqmc_gpu_set_heatbath_tables(sqmc_gpu_ctx *ctx, const sqmc_heatbath_tables *t);
/* replaces: setup_efficient_heatbath for run_type /= hci (chemistry.f90:1002-1225) with setup_alias (more_tools.f90:5603-5722) and
 * check_heatbath_unbiased (9330-9375): the library builds the tables itself -- |H_ijkl| of all orbital quadruples on the device, the
 * partial sums, normalisations and alias tables in the reference's loop order and precisions -- and installs them as
 * sqmc_gpu_set_heatbath_tables would.  *is_heatbath_unbiased = (max(nup, ndn) > number of orbitals of unique symmetry); when it is 0
 * the reference stops ("Heatbath may be biased for this system!", 1219) and nothing is installed.  sqmc_gpu_get_heatbath_tables hands
 * out the library's host copies (the reference's layout; valid until the next setup call or sqmc_gpu_finalize). */
int sqmc_gpu_setup_efficient_heatbath(sqmc_gpu_ctx *ctx, int32_t *is_heatbath_unbiased);
int sqmc_gpu_get_heatbath_tables(sqmc_gpu_ctx *ctx, sqmc_heatbath_tables *t, int32_t *n_orb_uniq_sym);
/* n proposals of off_diagonal_move_chem_efficient_heatbath, each from its own rannyu state seeds[4 i .. 4 i + 3] (test door, like
 * sqmc_gpu_propose_batch): det_j and weight_j = -tau H_ij / p of the two slots of proposal i at [2 i] and [2 i + 1] (weight 0: no move). */
int sqmc_gpu_propose_heatbath_batch(sqmc_gpu_ctx *ctx, int64_t n, double tau, const uint64_t *up, const uint64_t *dn, const int32_t *seeds,
                                    uint64_t *det_j_up, uint64_t *det_j_dn, double *weight_j, int32_t *seeds_after);

/* replaces: dtm_hb / pq_ind / pq_count built by setup_efficient_heatbath (chemistry.f90:900-993).
 * hb_r/hb_s/hb_absH: n_hb records sorted by descending absH inside each (p,q) class;
 * pq_ind (1-based start) / pq_count indexed by combine_2_indices(p,q) in [1, n_pq]. */
int sqmc_gpu_set_hb_tables(sqmc_gpu_ctx *ctx, int64_t n_hb, const int32_t *hb_r, const int32_t *hb_s,
                           const double *hb_absH, int32_t n_pq, const int64_t *pq_ind,
                           const int32_t *pq_count, double max_double);

/* replaces: minus_tau_H_indices / _nonzero_elements / _values (common_imp.f90:14-15) as
 * consumed at do_walk.f90:2262; scale = the tau_ratio rescale of do_walk.f90:2179,2919. */
int sqmc_gpu_set_projector(sqmc_gpu_ctx *ctx, int64_t n_imp, int64_t nnz, const int64_t *row_counts,
                           const int64_t *indices, const double *values);
int sqmc_gpu_scale_projector(sqmc_gpu_ctx *ctx, double ratio);

/* replaces: psi_t_connected_dets_up/dn, psi_t_connected_e_loc_num/den (common_psi_t.f90:20-32),
 * sorted by (up,dn), searched by binary_search_list_and_update (more_tools.f90:4041-4098). */
int sqmc_gpu_set_ct_table(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *up, const uint64_t *dn,
                          const double *e_num, const double *e_den);

/* hf_to_psit = .true. (input line do_walk.f90:378; "replace HF with psi_T for 1st state"): the step variant in which the first basis
 * state is the trial wave function, moves to and from it are deterministic and all determinants of C(T) stay in the walker list
 * (do_walk.f90:1056-1116, 1267-1300, 2272-2320, 2394-2462, 2701-2722, 3574, 3676, 5268-5307, 6484-6833, 7206-7254).
 * Call after sqmc_gpu_set_ct_table and sqmc_gpu_set_projector, before sqmc_gpu_upload_walkers:
 *   - the projector is the deterministic-space matrix as generate_sparse_ham_*_upper_triangular builds it with hf_to_psit
 *     (chemistry.f90:7885-7897, 7926-7933: no first row and column, the (1,1) element stored as 0), times -tau;
 *   - psit_ct_index[k] (1-based, increasing): where dets_up/dn_psi_t(k), Psi_T in label order (do_walk.f90:1258), lies in the C(T)
 *     list -- my_locations_of_psit (1849-1886); the first must be 1;  cdet_psi_t[k] its coefficient;
 *   - diag_elems[n_ct]: H_ii of the C(T) determinants outside the deterministic space, 0 inside (1091-1116);
 *   - the walker list is [the n_ct determinants of C(T) in order | the survivors outside C(T) in order]: upload takes it so
 *     (at the start: C(T) alone, do_walk.f90:1267-1300, imp_distance 0 inside the deterministic space and -2 elsewhere) and
 *     download returns it so.  The deterministic space must lie inside C(T) and only the first state may be a permanent initiator,
 *     as in the reference's own set-up.
 * sum_order: 1 = the step's three long sums (first row over C(T), first row over Psi_T, T^-1) through a fixed 64-ary tree;
 * 0 = left to right as the reference's loops run (one lane: slow).  One rank, COUNTER or REPLAY discipline, uniform proposal.
 * tests/golden/README_hf_to_psit.md records what parity is defined against (the reference's merge for this variant cannot run as written). */
int sqmc_gpu_set_hf_to_psit(sqmc_gpu_ctx *ctx, int64_t n_psit, const int64_t *psit_ct_index, const double *cdet_psi_t,
                            const double *diag_elems, int32_t sum_order);

/* walker SoA of common_walk.f90:5-18.  perm_sign(i) = sign_permanent_initiator of walker i
 * if initiator(i)==3 else 0 (the reference keeps the signs in sorted-walker order,
 * do_walk.f90:1150,2593-2594; here the sign travels with its walker). */
int sqmc_gpu_upload_walkers(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *up, const uint64_t *dn,
                            const double *wt, const int8_t *imp_distance, const int8_t *initiator,
                            const int8_t *perm_sign, const double *matrix_elements,
                            const double *e_num, const double *e_den);
int sqmc_gpu_num_walkers(sqmc_gpu_ctx *ctx, int64_t *n);
int sqmc_gpu_download_walkers(sqmc_gpu_ctx *ctx, int64_t cap, int64_t *n, uint64_t *up, uint64_t *dn,
                              double *wt, int8_t *imp_distance, int8_t *initiator,
                              double *matrix_elements, double *e_num, double *e_den);

/* by-value scalars of one MC step, do_walk.f90:2171-2934 */
typedef struct {
  double tau, e_trial, reweight_factor_inv, r_initiator, min_wt, always_spawn_cutoff_wt;
  int32_t initiator_power, initiator_min_distance, c_t_initiator, semistochastic, reached_w_abs_gen;
  int32_t reserved;
} sqmc_step_params;

/* out_stats[16]: 0 w_gen  1 w_abs_gen  2 e_den_gen  3 e_num_gen  4 w_perm_initiator_gen
 *   5 nwalk  6 w_abs_gen_imp  (the 7 reduced values of do_walk.f90:2689-2725)
 *   7 nwalk_before_merge  8 w2_gen  9 e_num2  10 e_den2  11 e_num_abs  12 e_den_abs
 *   13 e_num_e_den (the my_*_cum increments, 2670-2679)  14 w_abs_before_merge
 *   15 number of off-diagonal proposals made.
 * replaces: the move loop, deterministic projection, sort, merge, reduce, reweight and
 * estimator sums of do_walk.f90:2216-2487 and 2573-2790 (semistochastic chem, ncores=1). */
int sqmc_gpu_step(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, double out_stats[16]);
/* Population control around the step, as the reference's walk loop does it on the host:
 * tau / r_initiator ramp until w_abs_gen_target is first reached (do_walk.f90:2175-2184,
 * 2913-2923, with the projector rescaled by tau_ratio), e_est from the cumulated sums,
 * e_trial and reweight_factor_inv (do_walk.f90:2880-2901).  The first n_equil steps count as
 * equilibration (iblkk <= ntimes_nblk_eq*nblk_eq in the reference). */
typedef struct {
  double tau_sav, tau, tau_prev;          /* input tau; current (ramped) tau; tau of the previous step */
  double e_trial, e_est;
  double w_abs_gen_target, w_abs_gen;     /* target; w_abs_gen of the last step (set to the initial sum|w| before the first) */
  double r_initiator_sav, r_initiator, initiator_rescale_power;
  double population_control_exponent;
  double reweight_factor_inv, reweight_factor_inv_max;
  double e_num_cum, e_den_cum;            /* sums of e_num_gen*sign(e_den_gen), |e_den_gen| */
  double min_wt, always_spawn_cutoff_wt;
  int32_t reached_w_abs_gen, initiator_power, initiator_min_distance, c_t_initiator, semistochastic, reserved;
  int64_t istep, n_equil;
} sqmc_popctl;

/* replaces: the body of "do istep=1,nstep" (do_walk.f90:2171-2934) for nsteps consecutive steps
 * without returning to the caller in between: sqmc_gpu_step + the scalar updates above.
 * stats (may be NULL) receives the 16 per-step values of every step (nsteps*16 doubles);
 * totals[16] their sums over the nsteps steps.  pc is updated in place. */
int sqmc_gpu_run(sqmc_gpu_ctx *ctx, sqmc_popctl *pc, int64_t nsteps, double *stats, double totals[16]);

/* replaces: merge_sort2_up_dn + merge_original_with_spawned2 + reduce_my_walker and the estimator
 * sums, do_walk.f90:2364-2487, 2573-2790, as one call for a host that produces its spawns itself:
 * the n_spawn walkers (in creation order; weight 0 = no walker; imp_distance / initiator as
 * move_uniform2 sets them, do_walk.f90:3700-3727) are appended behind the resident walkers, then
 * sort, annihilation with the initiator rules, stochastic rounding, reweighting and the sums of
 * sqmc_gpu_step.  out_stats as in sqmc_gpu_step (entry 15 = 0). */
int sqmc_gpu_annihilate(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, int64_t n_spawn, const uint64_t *up, const uint64_t *dn,
                        const double *wt, const int8_t *imp_distance, const int8_t *initiator, double out_stats[16]);

/* ---- multi-rank sharding: one process per GPU, walkers owned by hash(det) mod nranks, as the
 * reference shards them over MPI ranks (get_det_owner, mpi_routines.f90:419-445).  The library
 * does no communication itself: the step is cut at the reference's two exchange points and the
 * host moves device buffers with its collective library (RCCL through torch.distributed in
 * sqmc_amd/host.py; MPI in the reference):
 *   shard_begin   gate, death/clone, spawn; owned deterministic-space weights -> x_global
 *     [host: all-reduce SUM of x_global]                      (mpi_redscatt, do_walk.f90:2259-2260)
 *   shard_pack    owned rows of the projection; spawns bucketed by owner into 32-byte records
 *     [host: all-to-all of the counts, then of the records]   (mpi_sendnewwalks, do_walk.f90:2237)
 *   shard_finish  received records appended; sort, annihilation, rounding, estimator sums
 *     [host: all-reduce SUM of out_stats[0..6]]               (mpi_allred, do_walk.f90:2778)
 * set_projector takes the GLOBAL projector on every rank; shard_config gives, for the k-th
 * deterministic-space walker this rank owns (in its sorted order), its row in that matrix. */
int sqmc_gpu_det_owner(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *up, const uint64_t *dn, int32_t nranks, int32_t *owner);
/* Which hash decides ownership.  0 (default): a mix of the determinant's sort key (cheapest).  1: the reference's own
 * get_det_owner -> hash -> djb_hash (mpi_routines.f90:419-445, 257-289, 354-379) bit for bit, so that ranks running the
 * reference and ranks running this library agree on who owns a determinant.  Call before distributing walkers. */
int sqmc_gpu_set_owner_hash(sqmc_gpu_ctx *ctx, int32_t mode);
int sqmc_gpu_shard_config(sqmc_gpu_ctx *ctx, int32_t rank, int32_t nranks, int64_t n_imp_local, const int32_t *global_row);
int sqmc_gpu_shard_begin(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, double *x_global_dev, int64_t *n_children);
int sqmc_gpu_shard_pack(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, const double *x_global_dev, uint64_t *send_dev,
                        int64_t cap_records, int64_t *send_counts);
int sqmc_gpu_shard_finish(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, const uint64_t *recv_dev, int64_t n_recv, double out_stats[16]);

/* The same sharded step with the three exchanges issued by the library itself over RCCL/xGMI
 * (replaces mpi_allred of the deterministic weights do_walk.f90:2259-2260, mpi_snd_list /
 * mpi_sendnewwalks mpi_routines.f90:1147-1270, and the mpi_allred of the sums
 * do_walk.f90:2778-2790).  Rank 0 obtains an id with sqmc_gpu_comm_unique_id and hands it to the
 * other ranks by any means (MPI_Bcast in the reference's build); every rank then calls
 * sqmc_gpu_comm_init after sqmc_gpu_shard_config.  sqmc_gpu_shard_step = begin + all-reduce +
 * pack + all-to-all + finish + all-reduce: out_stats[0..6] are sums over all ranks,
 * out_stats[7..15] stay local.  sqmc_gpu_shard_run is sqmc_gpu_run over sharded steps (every
 * rank carries the same population-control state, as the reference's ranks do).  Once a
 * communicator is attached, sqmc_gpu_shard_finish also all-reduces; use either the three-phase
 * calls without a communicator or shard_step with one. */
#define SQMC_COMM_ID_BYTES 128
int sqmc_gpu_comm_unique_id(uint8_t id[SQMC_COMM_ID_BYTES]);
int sqmc_gpu_comm_init(sqmc_gpu_ctx *ctx, const uint8_t id[SQMC_COMM_ID_BYTES]);
/* number of ranks RCCL itself reports for the communicator (ncclCommCount; the reference's ncores, mpi_routines.f90:310) */
int sqmc_gpu_comm_size(sqmc_gpu_ctx *ctx, int32_t *nranks);
int sqmc_gpu_shard_step(sqmc_gpu_ctx *ctx, const sqmc_step_params *p, double out_stats[16]);
int sqmc_gpu_shard_run(sqmc_gpu_ctx *ctx, sqmc_popctl *pc, int64_t nsteps, double *stats /* nsteps*16 or NULL */, double totals[16]);
/* diagnostics: where the HOST spent the wall clock of the sqmc_gpu_shard_step calls since the last reset, in microseconds summed over
 * *steps completed steps: us[0] head (gate, spawn and death/clone enqueued or taken over from the pipelined head, until the child count is
 * known), us[1] exchange (projection all-reduce, bucketing by owner, pack, all-gather of the send counts until the host has read them,
 * grouped send/receive enqueued), us[2] tail (unpack, sort and annihilation enqueued, until the all-reduced sums are read; the next
 * step's head is enqueued in here), us[3] the part of all three spent waiting for a word the GPU writes.  What a scaling run needs to
 * explain itself: kernel time hides inside the waits, link latency inside us[1] and us[2]. */
int sqmc_gpu_shard_time_split(sqmc_gpu_ctx *ctx, double us[4], int64_t *steps, int32_t reset);

/* RNG state of the REPLAY stream (savern / setrn, rannyu.f90:11-21,77-87) */
/* How many steps took the short-list tail (block-local partition + one annihilation kernel per key range, no global sort) and
 * how many of those had to be re-run through the radix tail because a key range outgrew its block (diagnostics, tests). */
int sqmc_gpu_tail_stats(sqmc_gpu_ctx *ctx, int64_t *bucket_steps, int64_t *bucket_retries);
/* A host that runs the walk in blocks (nstep steps, then its block statistics: do_walk.f90:2171 inside the iblk loop, 2086-3300)
 * calls sqmc_gpu_run once per block.  With chained runs on, the last step of a call enqueues the head (gate, child offsets, spawn)
 * of the first step of the NEXT call behind its own tail, as every other step of the call does for its successor, so the GPU keeps
 * working while the host does its block bookkeeping (the first step of a call costs 0.135 instead of 0.075 ms otherwise).  The
 * head is forgotten, at the cost of one stream synchronisation, if anything but a step with the same tau / cutoff / mode comes
 * next (walkers uploaded or downloaded, projector rescaled, tables of either heat-bath kind set, hf_to_psit set, RNG reset, chaining switched off, finalize).  The same holds for a host that
 * calls sqmc_gpu_step itself, step by step (its own work between steps, as in the reference's loop): with chaining on, every step past the
 * target population enqueues its successor's head.  Single-GPU steps only. */
int sqmc_gpu_set_chained_runs(sqmc_gpu_ctx *ctx, int32_t on);
/* diagnostics: wall-clock time (microseconds) and index of the four slowest steps of the last sqmc_gpu_run / sqmc_gpu_shard_run
 * call -- a step that waited for the host (scheduling, a rerun through the radix tail) stands out here */
int sqmc_gpu_slowest_steps(sqmc_gpu_ctx *ctx, double us[4], int64_t step[4]);
int sqmc_gpu_get_rng(sqmc_gpu_ctx *ctx, int32_t seed[4]);
int sqmc_gpu_set_rng(sqmc_gpu_ctx *ctx, const int32_t seed[4]);

/* replaces: fast_sparse_matrix_multiply_upper_triangular (more_tools.f90:3622-3670) as
 * called from davidson_sparse (more_tools.f90:2115,2188).  prepare once per matrix
 * (converts to int32 full CSR in HBM), apply per matvec.  x/y are host pointers unless
 * on_device != 0. */
typedef struct sqmc_spmv_plan sqmc_spmv_plan;
int sqmc_gpu_spmv_prepare(int64_t n, const int64_t *row_counts, const int64_t *indices,
                          const double *values, sqmc_spmv_plan **plan);
int sqmc_gpu_spmv_apply(sqmc_spmv_plan *plan, const double *x, double *y, int on_device);
int sqmc_gpu_spmv_free(sqmc_spmv_plan *plan);
/* The lowest n_states eigenpairs of the plan's matrix by the diagonally preconditioned Davidson of davidson_sparse
 * (more_tools.f90:2018-2244; davidson_sparse_single :3056-3230), every vector resident on the device.  diag[n]: the diagonal (the
 * preconditioner, as sqmc_gpu_build_spmv_plan returns it); v0: n x n_states start vectors, column-major (initial_vector), or NULL for
 * unit vectors on the first rows (more_tools.f90:3113-3114: the HF determinant when it is listed first); tol: the reference's
 * epsilon = 1e-10 on the eigenvalues.  Out: evals[n_states], evecs[n x n_states] column-major (largest component of the small
 * problem's eigenvector positive), *n_matvec (may be NULL) the products it took. */
int sqmc_gpu_davidson(sqmc_spmv_plan *plan, const double *diag, int32_t n_states, const double *v0, double tol, double *evals, double *evecs, int32_t *n_matvec);
/* generate_sparse_ham_chem_upper_triangular (chemistry.f90:7639-8010) and the plan of the Davidson
 * matvec in one call, with the matrix never leaving the GPU: for a determinant list sorted by
 * (up,dn) the Hamiltonian is built and expanded to the full symmetric CSR on the device; diag[n]
 * (Davidson's preconditioner, more_tools.f90:2099-2110) and the number of stored upper-triangular
 * nonzeros come back.  Use the plan with sqmc_gpu_spmv_apply / _free. */
int sqmc_gpu_build_spmv_plan(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *up, const uint64_t *dn, sqmc_spmv_plan **plan, double *diag,
                             int64_t *out_nnz);
int sqmc_gpu_spmv_sym_upper(int64_t n, const int64_t *row_counts, const int64_t *indices,
                            const double *values, const double *x, double *y);

/* replaces: hamiltonian_chem / hamiltonian_chem_time_sym through the dispatcher
 * semistoch.f90:2234-2302, for n (bra,ket) pairs; 0 where not connected. */
int sqmc_gpu_hamiltonian_batch(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *iu, const uint64_t *id,
                               const uint64_t *ju, const uint64_t *jd, double *h);

/* replaces: hamiltonian_chem (chemistry.f90:1260-1320) with the excitation level found from
 * the pair (no time-reversal symmetrisation even when time_sym is set); used to build
 * dtm_hb exactly as double_excitation_matrix_element_no_ref does (chemistry.f90:9615-9646). */
int sqmc_gpu_hamiltonian_chem_batch(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *iu, const uint64_t *id,
                                    const uint64_t *ju, const uint64_t *jd, double *h);

/* replaces: generate_sparse_ham_chem_upper_triangular (chemistry.f90:7639-8010) for a list
 * sorted by (up,dn): the symmetric Hamiltonian in the "upper triangular" storage above
 * (each row: its diagonal first, then columns j < i ascending; zero elements dropped).
 * Output arrays are allocated by the library; release each with sqmc_gpu_free. */
int sqmc_gpu_build_sparse_ham(sqmc_gpu_ctx *ctx, int64_t n, const uint64_t *up, const uint64_t *dn,
                              int64_t *out_nnz, int64_t **out_row_counts, int64_t **out_indices,
                              double **out_values);

/* test door onto the proposal kernel: off_diagonal_move_chem (chemistry.f90:4237-5084) for n
 * parents, child k of the batch drawing from rannyu state seeds[4k..4k+3]; returns det_j,
 * weight_j (= -tau*H/p) and the state after. */
int sqmc_gpu_propose_batch(sqmc_gpu_ctx *ctx, int64_t n, double tau, const uint64_t *up,
                           const uint64_t *dn, const int32_t *seeds, uint64_t *ju, uint64_t *jd,
                           double *weight_j, int32_t *seeds_after);

/* replaces: find_doubly_excited + find_important_connected_dets_chem + sort/dedup
 * (semistoch.f90:1750-2131, chemistry.f90:6819-7159, tools.f90:577-660) as used by
 * get_next_det_list (hci.f90:865) and generate_psi_t_connected_e_loc (semistoch.f90:27):
 * for n_ref reference dets with coefficients coeffs, all connections with
 * |H| >= eps/|c| (the reference det itself included), sorted by (up,dn), duplicates merged:
 * e_mix_num = sum_j H_ij c_j, e_mix_den = c_i on the reference determinants, else 0
 * (semistoch.f90:2039-2063).  diag_mode 0: the self slot carries H=0 (HCI, chemistry.f90:6896);
 * 1: it carries H_ii (find_connected_dets_chem, chemistry.f90:6574-6576, for C(T));
 * 2: "raw" -- no sort, no merge: every generated connection in generation order with
 *    e_mix_num = H_ki c_i and e_mix_den = i (0-based index of its reference determinant), what the
 *    weighted sums of second_order_pt_alias (hci.f90:1314-1660: term1, term2) are formed from.
 * out arrays are allocated by the library, release each with sqmc_gpu_free. */
int sqmc_gpu_hci_connections(sqmc_gpu_ctx *ctx, int64_t n_ref, const uint64_t *ref_up,
                             const uint64_t *ref_dn, const double *coeffs, double eps, int diag_mode,
                             int64_t *out_n, uint64_t **out_up, uint64_t **out_dn,
                             double **out_e_mix_num, double **out_e_mix_den);
/* The same restricted to slice `slice` of `n_slices` equal parts of the determinant-key range: every
 * connection belongs to exactly one slice and its merged sums are exact within it, so a PT stage whose
 * connected space does not fit one call (2^31 connections) is done slice by slice (the role of
 * n_energy_batch, hci.f90:642).  slice 0 of 1 = sqmc_gpu_hci_connections. */
int sqmc_gpu_hci_connections_slice(sqmc_gpu_ctx *ctx, int64_t n_ref, const uint64_t *ref_up, const uint64_t *ref_dn,
                                   const double *coeffs, double eps, int diag_mode, int32_t slice, int32_t n_slices, int64_t *out_n,
                                   uint64_t **out_up, uint64_t **out_dn, double **out_e_mix_num, double **out_e_mix_den);
/* replaces: the optional arguments core_up/dn, virt_up/dn, active_only of find_important_connected_dets_chem
 * (chemistry.f90:6840-6846, 6926-6947, 7087-7108) as get_next_det_list, second_order_pt and do_pt pass them down
 * (hci.f90:150-182, 386-389, 786-798, 914-920).  mode 0: no masks; 1: only determinants inside the active space (every core
 * orbital occupied, every virtual orbital empty); 2: only determinants outside it.  State of the context: applies to
 * sqmc_gpu_hci_connections(_slice) and sqmc_gpu_hci_pt2 until changed.  Chemistry only. */
int sqmc_gpu_hci_set_active_space(sqmc_gpu_ctx *ctx, uint64_t core_up, uint64_t core_dn, uint64_t virt_up, uint64_t virt_dn, int32_t mode);
/* replaces: second_order_pt (hci.f90:1100-1182), the deterministic Epstein-Nesbet correction of a variational wavefunction:
 * delta_e = sum_a (sum_i H_ai c_i)^2 / (E_var - H_aa) over the connected determinants outside the variational space, the inner
 * sum screened by |H_ai c_i| >= eps_pt; n_connections = connected determinants visited (the reference prints it).  Everything
 * stays on the device; n_slices > 1 does the connected space in parts of the determinant range.  Needs sqmc_gpu_set_hb_tables
 * (chem) like the generator.  For a time-symmetrised wavefunction pass the determinant-basis expansion, as the reference does
 * (convert_time_symmetrized_to_dets, hci.f90:4365-4564) on a context initialised with time_sym = 0. */
int sqmc_gpu_hci_pt2(sqmc_gpu_ctx *ctx, int64_t n_var, const uint64_t *var_up, const uint64_t *var_dn, const double *coeffs, double e_var,
                     double eps_pt, int32_t n_slices, double *delta_e, int64_t *n_connections);
void sqmc_gpu_free(void *p);

/* HIP-event timing on the library's streams.  level 0 off; 1 = only the longest kernel on the
 * critical path, on every 8th step (the annihilation kernel of a semistochastic walk, k_spawn of
 * a plain one: a timed launch cannot overlap its neighbours); 2 = every stage of every step.  get_timing returns the mean
 * milliseconds per step of each timer over the steps run since set_timing (n <= 32). */
int sqmc_gpu_set_timing(sqmc_gpu_ctx *ctx, int level);
int sqmc_gpu_get_timing(sqmc_gpu_ctx *ctx, int32_t *n, const char **names, float *ms);

#ifdef __cplusplus
}
#endif
#endif
