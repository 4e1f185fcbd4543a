/*
 * sqmc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the sqmc reference algorithms on the walker hot
 * path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker.
 * Every function cites the reference file:line (relative to /root/reference/src)
 * whose behaviour it restates.  Determinants are one 64-bit word per spin
 * (norb <= 64); bit k <-> orbital k+1, as in the reference (types.f90:14-44).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - rannyu / random_int / permutation_factor / merge sort: checked against the
 *     real reference objects compiled unmodified into oracle/_ref (tools.f90,
 *     rannyu.f90).
 *   - hamiltonian_chem + HCI generator + sym. matvec: checked against the HCI
 *     energies and determinant counts the reference produced on
 *     C2_v2z_curve/r1.24253.
 *   - walker step (move/merge/reduce/estimators): the reference holds no fixture
 *     for it ("parity unpinned" beyond the component pins above).
 */
#ifndef SQMC_ORACLE_H
#define SQMC_ORACLE_H
#include <stdint.h>

typedef uint64_t det_t;

/* ---- RNG: rannyu.f90:11-87, tools.f90:129-147 ---- */
/* mode 0: the reference's rannyu stream.  mode 1 ("counter"): an independent 48-bit
 * splitmix64 stream per (seed, step, stage, entity) so that a parallel implementation can
 * be checked draw for draw at any size; not in the reference (DESIGN.md "RNG"). */
typedef struct { int l[4]; int mode; int pad; uint64_t ctr, seed, step; } orc_rng;
void   orc_rng_set_mode(orc_rng *g, int mode);
void   orc_rng_seek(orc_rng *g, int stage, uint64_t idx);
void   orc_setrn(orc_rng *g, const int seed[4]);
void   orc_savern(const orc_rng *g, int seed[4]);
double orc_rannyu(orc_rng *g);
int    orc_random_int(orc_rng *g, int n);

/* ---- chemistry system: chemistry.f90:121-869 ---- */
#define ORC_MAXORB 64
#define ORC_MAXSYM 8
typedef struct {
  int norb, nelec, nup, ndn, n_core_orb;
  int time_sym, z;
  int n_group;                       /* point group order (d2h = 8)            */
  int prod[ORC_MAXSYM + 1][ORC_MAXSYM + 1];       /* product_table, 1-based     */
  int orbsym[ORC_MAXORB + 1];        /* orbital_symmetries after reordering     */
  int orb_order[ORC_MAXORB + 2];     /* new index -> FCIDUMP label (1-based)    */
  int orb_order_inv[ORC_MAXORB + 2];
  int combine_2[ORC_MAXORB + 2][ORC_MAXORB + 2];  /* chemistry.f90:384-394,856  */
  int64_t n_int;
  double *integrals;                 /* 1-based packed array (index 0 unused)   */
  double nuclear;                    /* nuclear_nuclear_energy                  */
  double orbital_energies[ORC_MAXORB + 1];
  det_t hf_up, hf_dn;
  int num_orb_by_sym[ORC_MAXSYM + 1];
  int which_orb_by_sym[ORC_MAXSYM + 1][ORC_MAXORB + 1];
  /* HCI heat-bath double-excitation table: chemistry.f90:900-993 */
  int64_t n_hb;
  int *hb_r, *hb_s; double *hb_absH;
  int64_t *pq_ind; int *pq_count; int n_pq;
  double max_double;
  /* active-space masks of find_important_connected_dets_chem (chemistry.f90:6840-6846, 6926-6947): as_mode 0 = none,
   * 1 = only determinants inside the active space (core fully occupied, virtuals empty), 2 = only those outside it */
  int as_mode; det_t as_core_up, as_core_dn, as_virt_up, as_virt_dn;
} orc_chem;

/* hf_mode: 0 = first nup/ndn orbitals (walk decks), 1 = auto HF of symmetry
 * hf_symmetry (HCI decks, chemistry.f90:10359-10454). */
orc_chem *orc_chem_load(const char *fcidump, int nelec, int nup, const char *point_group,
                        int time_sym, int z, int n_core_orb, int hf_mode, int hf_symmetry);
void   orc_chem_free(orc_chem *s);
void   orc_chem_setup_hb(orc_chem *s);
int64_t orc_integral_index(const orc_chem *s, int i, int j, int k, int l);
double orc_integral_value(const orc_chem *s, int p, int q, int r, int t);

int    orc_permutation_factor(det_t a, det_t b);
void   orc_permutation_factor2(det_t di, det_t dj, int *gamma, int *i1, int *i2, int *j1, int *j2);
int    orc_excitation_level(det_t iu, det_t id, det_t ju, det_t jd);
double orc_hamiltonian_chem(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd, int level);
double orc_hamiltonian_chem_time_sym(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd);
/* dispatcher semistoch.f90:2234: time_sym aware, any pair; returns 0 if not connected */
double orc_hamiltonian(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd);

/* off_diagonal_move_chem, chemistry.f90:4237-5084 (time_sym = false branch) */
void orc_off_diagonal_move_chem(const orc_chem *s, orc_rng *g, double tau, det_t iu, det_t id,
                                det_t *ju, det_t *jd, double *weight_j, int *n_draws);

/* ---- efficient heat-bath proposal (oracle/sqmc_oracle_hb.c): chemistry.f90:872-1230, 5086-5816, 9154-9375 ---- */
typedef struct {
  int norb, nup, ndn, n_core, n_orb_uniq_sym, unbiased;
  int64_t size_same, size_opp, n_pairs;
  double *one, *two;                                  /* one_orbital_probabilities(norb), two_orbital_probabilities(2norb,2norb); 1-based with an unused 0 row */
  double *three_same, *three_opp; int *j3_same, *j3_opp; double *q3_same, *q3_opp;      /* (norb+1)^3, [i][j][k] */
  float *four_same, *four_opp; int *j4_same, *j4_opp; float *q4_same, *q4_opp;          /* single precision, as the reference stores them; [same_index] / [opposite_index] */
  double *htot_same, *htot_opp;                       /* Htot_same(combine_2(i,j), k), Htot_opposite(i,j,k) */
} orc_hb;
orc_hb *orc_hb_setup(const orc_chem *c);
void    orc_hb_free(orc_hb *h);
int64_t orc_hb_same_index(const orc_hb *h, int f1, int f2, int t1, int t2);
int64_t orc_hb_opposite_index(const orc_hb *h, int f1, int f2, int t1, int t2);
/* returns n_new_dets (0, 1 or 2: a single larger than all its doubles together comes back WITH a double, do_walk.f90:3604-3611) */
int orc_off_diagonal_move_chem_heatbath(const orc_chem *c, const orc_hb *h, orc_rng *g, double tau, det_t iu, det_t id,
                                        det_t ju[2], det_t jd[2], double weight_j[2], int excite_level[2], int *n_draws);
/* probability with which det_i -> det_j would be proposed (proposal_prob_efficient_heatbath, 5431-5549) */
double orc_hb_proposal_prob(const orc_chem *c, const orc_hb *h, det_t iu, det_t id, det_t ju, det_t jd, int excite_level, double off_diag_elem);

/* connections */
int orc_find_connected_dets_chem(const orc_chem *s, det_t up, det_t dn, det_t *cu, det_t *cd,
                                 double *elems, int cap);
void orc_set_active_space(orc_chem *s, det_t core_up, det_t core_dn, det_t virt_up, det_t virt_dn, int mode);
int orc_find_important_connected_dets_chem(const orc_chem *s, det_t up, det_t dn, double eps,
                                           det_t *cu, det_t *cd, double *elems, int cap);

/* symmetric "upper triangular" CSR matvec, more_tools.f90:3622-3670 */
void orc_spmv_sym_upper(int64_t n, const int64_t *row_counts, const int64_t *indices,
                        const double *values, const double *x, double *y);

/* sparse Hamiltonian among a sorted determinant list (lower triangle, diagonal first
 * in each row); pure function of the list -- any construction algorithm gives the same
 * matrix (reference: chemistry.f90:7639-8010). Returns nnz; arrays malloc'ed. */
int64_t orc_build_sparse_ham(const orc_chem *s, int64_t n, const det_t *up, const det_t *dn,
                             int64_t **row_counts, int64_t **indices, double **values);
void orc_free(void *p);

/* ---- walker state + one MC step: do_walk.f90:2171-2934 ---- */
typedef struct {
  int64_t nwalk, mwalk;
  det_t *up, *dn;
  double *wt;
  int8_t *imp_distance, *initiator;
  double *matrix_elements, *e_num_walker, *e_den_walker;
  /* deterministic projector -tau*H, common_imp.f90:4-17 */
  int64_t n_imp, nnz;
  int64_t *prj_counts, *prj_indices; double *prj_values;
  /* C(T): common_psi_t.f90:20-32 (sorted by up,dn) */
  int64_t n_ct; det_t *ct_up, *ct_dn; double *ct_num, *ct_den;
  /* permanent initiators (sorted order) do_walk.f90:1150 */
  int n_perm; int8_t *sign_perm;
  orc_rng rng;
  int64_t n_spawn_draws;            /* RNG draws consumed in the last step      */
  /* COUNTER discipline only: norb and the number of dn electrons, from which the rounding draw's key -- the determinant's
   * rank in (up, dn) order, colex_rank(up) C(norb, ndn) + colex_rank(dn) -- is computed (0: not set, keyed by up*phi + dn) */
  int key_norb, key_ndn;
} orc_walk;
uint64_t orc_det_rank(int norb, int ndn, det_t up, det_t dn);

typedef struct {
  double tau, e_trial, reweight_factor_inv, r_initiator, min_wt, always_spawn_cutoff_wt;
  int initiator_power, initiator_min_distance, c_t_initiator, semistochastic, reached_w_abs_gen;
} orc_step_params;

/* out[16]: 0 w_gen 1 w_abs_gen 2 e_den_gen 3 e_num_gen 4 w_perm_initiator_gen 5 nwalk
 *          6 w_abs_gen_imp 7 nwalk_before_merge 8 w2_gen 9 e_num2 10 e_den2 11 e_num_abs
 *          12 e_den_abs 13 e_num_e_den 14 w_abs_before_merge 15 n_spawn_attempts */
orc_walk *orc_walk_new(int64_t mwalk);
void orc_walk_free(orc_walk *w);
int  orc_walk_step(const orc_chem *s, orc_walk *w, const orc_step_params *p, double out[16]);
/* the same with proposal_method fast_heatbath: every child may add TWO walkers (do_walk.f90:3604-3611, add_walker 7584-7697) */
int  orc_walk_step_heatbath(const orc_chem *c, const orc_hb *hb, orc_walk *w, const orc_step_params *p, double out[16]);
/* n > 1: the COUNTER-discipline step runs its spawn loop, sort and permutations on n OpenMP threads (bit-identical results):
 * the all-host-cores CPU baseline of SURVEY section 8d(ii) */
void orc_set_threads(int n);
int  orc_get_threads(void);

/* pieces exposed for component tests */
void orc_merge_sort_walkers(orc_walk *w, int64_t n);                       /* do_walk.f90:5169-5197 */
/* get_det_owner (mpi_routines.f90:419-445) -> hash (257-289) -> djb_hash (354-379): the rank that owns a determinant.
 * Determinants are the reference's 128-bit integers, passed as (low, high) 64-bit halves (conv_128_to_64, 671-680).
 * PARITY UNPINNED: mpi_routines.f90 does not compile unmodified with this image's flang and the reference holds no
 * fixture of owners; the restatement follows the source text (wrapping INTEGER(16) arithmetic, logical shifts). */
void orc_djb_hash(uint64_t up_lo, uint64_t up_hi, uint64_t dn_lo, uint64_t dn_hi, uint64_t hash_out[2]);
int  orc_get_det_owner(uint64_t up_lo, uint64_t up_hi, uint64_t dn_lo, uint64_t dn_hi, int ncores);
int64_t orc_merge_original_with_spawned2(orc_walk *w, int64_t n, const orc_step_params *p); /* 5866-6083 */
int64_t orc_reduce_my_walker(orc_walk *w, int64_t n, const orc_step_params *p);            /* 7196-7254 */
int64_t orc_join_walker2(orc_walk *w, int64_t n, const orc_step_params *p);                /* 6990-7103 */

#endif

/* ---- homogeneous electron gas in k-space: heg.f90 (read_heg 119-166, system_setup_heg 168-215,
 * generate_k_vectors 643-749, hamiltonian_heg 845-1011, off_diagonal_move_heg 1344-1598) ---- */
#ifndef SQMC_ORACLE_HEG_H
#define SQMC_ORACLE_HEG_H
typedef struct {
  int n_dim, nelec, nup, ndn, norb, n_max;
  double r_s, length_cell;
  double k[ORC_MAXORB + 1][3];        /* k_vectors(:, i), 1-based orbital index */
  int krel[ORC_MAXORB + 1][3];        /* k in units of 2 pi / L */
} orc_heg;
orc_heg *orc_heg_new(int n_dim, double r_s, int nelec, int nup, double cutoff_radius);
void   orc_heg_free(orc_heg *h);
double orc_hamiltonian_heg(const orc_heg *h, det_t iu, det_t id, det_t ju, det_t jd);
void   orc_off_diagonal_move_heg(const orc_heg *h, orc_rng *g, double tau, det_t iu, det_t id,
                                 det_t *ju, det_t *jd, double *weight_j, int *n_draws);
/* the determinant itself + every momentum-conserving double excitation (any order) */
int    orc_connected_heg(const orc_heg *h, det_t up, det_t dn, det_t *cu, det_t *cd, double *elems, int cap);
int64_t orc_build_sparse_ham_heg(const orc_heg *h, int64_t n, const det_t *up, const det_t *dn,
                                 int64_t **row_counts, int64_t **indices, double **values);
/* one MC step with the HEG as the system (same step logic as orc_walk_step) */
int  orc_walk_step_heg(const orc_heg *h, orc_walk *w, const orc_step_params *p, double out[16]);
#endif

/* ---- real-space Hubbard model on a square lattice: hubbard.f90 (read_hubbard 138-382,
 * choose_random_electron 1024-1058, hamiltonian_hubbard 1536-1644, off_diagonal_move_hubbard
 * 2992-3120, find_connected_dets_hubbard 5306-5457), more_tools.f90 (fermionic_phase 140-176,
 * get_nbr 223-355).  PARITY UNPINNED: the reference holds no fixture for this model and
 * hubbard.f90 does not compile unmodified with the flang of this image; the restatement follows
 * the source text line by line. ---- */
#ifndef SQMC_ORACLE_HUB_H
#define SQMC_ORACLE_HUB_H
typedef struct {
  int l_x, l_y, pbc, nsites, nup, ndn;
  double t, U;
} orc_hub;
orc_hub *orc_hub_new(int l_x, int l_y, int pbc, int nup, int ndn, double t, double U);
void   orc_hub_free(orc_hub *h);
/* get_nbr: nbr_type 1 LEFT, 2 RIGHT, 3 UP, 4 DOWN; returns the neighbour (1-based) or -1 if not allowed */
int    orc_get_nbr(int l_x, int l_y, int pbc, int site, int nbr_type);
int    orc_fermionic_phase(det_t config, int site_1, int site_2);
/* hamiltonian_hubbard as written: "NOT checking for neighbors, assumes inherent connectedness" */
double orc_hamiltonian_hubbard(const orc_hub *h, det_t iu, det_t id, det_t ju, det_t jd);
/* the same with the lattice-bond test of is_connected_hubbard (5876-5977) in front: what a matrix builder may call on any pair */
double orc_hamiltonian_hubbard_checked(const orc_hub *h, det_t iu, det_t id, det_t ju, det_t jd);
void   orc_off_diagonal_move_hubbard(const orc_hub *h, orc_rng *g, double tau, det_t iu, det_t id,
                                     det_t *ju, det_t *jd, double *weight_j, int *n_draws);
/* find_connected_dets_hubbard: the hops in the reference's order (electron by electron, L R U D), the determinant itself LAST */
int    orc_connected_hubbard(const orc_hub *h, det_t up, det_t dn, det_t *cu, det_t *cd, double *elems, int cap);
int64_t orc_build_sparse_ham_hubbard(const orc_hub *h, int64_t n, const det_t *up, const det_t *dn,
                                     int64_t **row_counts, int64_t **indices, double **values);
int  orc_walk_step_hubbard(const orc_hub *h, orc_walk *w, const orc_step_params *p, double out[16]);
#endif
