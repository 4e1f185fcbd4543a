/*
 * sqmc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See sqmc_oracle.h for scope, rules of use and pinning status.
 * Plain C restatement of the reference algorithms; file:line citations are
 * relative to /root/reference/src.  Written to be read next to the reference,
 * not to be fast.  Compile with -ffp-contract=off (the reference Makefile's
 * gfortran -O2 on x86-64 issues no fused multiply-adds).
 */
#include "sqmc_oracle.h"
#include <math.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define BIT(k) ((det_t)1 << (k))
static inline int btest(det_t d, int k) { return (int)((d >> k) & 1u); }
static inline int trailz(det_t d) { return __builtin_ctzll(d); }
static inline int popcnt(det_t d) { return __builtin_popcountll(d); }

/* ===================================================================== RNG */
/* rannyu.f90:11-21 : seed copy, last limb forced odd */
void orc_setrn(orc_rng *g, const int seed[4]) {
  for (int i = 0; i < 4; i++) g->l[i] = seed[i];
  g->l[3] = 2 * (g->l[3] / 2) + 1;
  /* 48-bit value of the state; limbs read with '(4i4)' may exceed 12 bits, rannyu's limb
   * products treat them as coefficients of 2^12 powers, hence the sum */
  g->seed = ((((uint64_t)g->l[0] << 36) + ((uint64_t)g->l[1] << 24) + ((uint64_t)g->l[2] << 12) + (uint64_t)g->l[3])) & 0xFFFFFFFFFFFFull;
}
static uint64_t mix64(uint64_t v) {
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull; v ^= v >> 27; v *= 0x94D049BB133111EBull; v ^= v >> 31;
  return v;
}
void orc_rng_set_mode(orc_rng *g, int mode) { g->mode = mode; g->step = 0; }
void orc_rng_seek(orc_rng *g, int stage, uint64_t idx) {
  /* the seed is mixed BEFORE step and stage enter: per-rank seeds differ in their low bits only (do_walk.f90:234 adds the
   * rank to the last limb), where the step/stage field lives -- unmixed, rank r at step s would reuse rank 0's streams of
   * another step */
  if (g->mode == 1) g->ctr = mix64(mix64(mix64(g->seed) ^ (g->step * 4ull + (uint64_t)stage)) + idx);
}
/* rannyu.f90:77-87 */
void orc_savern(const orc_rng *g, int seed[4]) { for (int i = 0; i < 4; i++) seed[i] = g->l[i]; }
/* rannyu.f90:54-74 : l <- l * 11^13 mod 2^48 on four 12-bit limbs; m = 502,1521,4071,2107 */
double orc_rannyu(orc_rng *g) {
  if (g->mode == 1) {
    g->ctr += 0x9E3779B97F4A7C15ull;
    return (double)(mix64(g->ctr) >> 16) * 3.552713678800500929355621337890625e-15;
  }
  const int m1 = 502, m2 = 1521, m3 = 4071, m4 = 2107;
  int l1 = g->l[0], l2 = g->l[1], l3 = g->l[2], l4 = g->l[3];
  int i1 = l1 * m4 + l2 * m3 + l3 * m2 + l4 * m1;
  int i2 = l2 * m4 + l3 * m3 + l4 * m2;
  int i3 = l3 * m4 + l4 * m3;
  int i4 = l4 * m4;
  l4 = i4 % 4096; i3 += i4 / 4096;
  l3 = i3 % 4096; i2 += i3 / 4096;
  l2 = i2 % 4096;
  l1 = (i1 + i2 / 4096) % 4096;
  g->l[0] = l1; g->l[1] = l2; g->l[2] = l3; g->l[3] = l4;
  const double t = 2.44140625e-4;
  return t * ((double)l1 + t * ((double)l2 + t * ((double)l3 + t * ((double)l4))));
}
/* tools.f90:129-147 */
int orc_random_int(orc_rng *g, int n) { return (int)((double)n * orc_rannyu(g)) + 1; }

/* ============================================================== integrals */
/* chemistry.f90:9106-9134 */
int64_t orc_integral_index(const orc_chem *s, int i, int j, int k, int l) {
  int64_t a = s->combine_2[i][j], b = s->combine_2[k][l];
  return (a > b) ? (a * (a - 1)) / 2 + b : (b * (b - 1)) / 2 + a;
}
/* chemistry.f90:1234-1256 */
double orc_integral_value(const orc_chem *s, int p, int q, int r, int t) {
  return s->integrals[orc_integral_index(s, p, q, r, t)];
}
#define IV(p, q, r, t) orc_integral_value(s, (p), (q), (r), (t))

/* chemistry.f90:7232-7343 (c1, cs, c2v, c2h, d2h; 'dih' is not restated) */
static int init_point_group(orc_chem *s, const char *pg) {
  static const int d2h[8][8] = {{1,2,3,4,5,6,7,8},{2,1,4,3,6,5,8,7},{3,4,1,2,7,8,5,6},{4,3,2,1,8,7,6,5},
                                {5,6,7,8,1,2,3,4},{6,5,8,7,2,1,4,3},{7,8,5,6,3,4,1,2},{8,7,6,5,4,3,2,1}};
  int n;
  if (!strcmp(pg, "c1")) n = 1; else if (!strcmp(pg, "cs")) n = 2;
  else if (!strcmp(pg, "c2v") || !strcmp(pg, "c2h")) n = 4; else if (!strcmp(pg, "d2h")) n = 8;
  else return -1;
  s->n_group = n;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) s->prod[i + 1][j + 1] = d2h[i][j];
  return 0;
}

/* tools.f90:1294-1340 */
int orc_permutation_factor(det_t a, det_t b) {
  det_t diff = (a > b) ? (a & (a - b)) : (b & (b - a));
  return (popcnt(diff) & 1) ? -1 : 1;
}
static inline det_t maskr(int n) { return n >= 64 ? ~(det_t)0 : (BIT(n) - 1); }
/* tools.f90:1345-1396 */
void orc_permutation_factor2(det_t di, det_t dj, int *gamma, int *i1, int *i2, int *j1, int *j2) {
  det_t d = di & ~dj;
  *i1 = trailz(d); *i2 = trailz(d & ~BIT(*i1));
  d = dj & ~di;
  *j1 = trailz(d); *j2 = trailz(d & ~BIT(*j1));
  d = di & dj & ((maskr(*i1) ^ maskr(*j1)) ^ (maskr(*i2) ^ maskr(*j2)));
  *gamma = (popcnt(d) & 1) ? -1 : 1;
}
/* chemistry.f90:7162-7227 : 0/1/2 or -1 if more than a double apart */
int orc_excitation_level(det_t iu, det_t id, det_t ju, det_t jd) {
  int n = popcnt(iu & ~ju) + popcnt(id & ~jd);
  return n > 2 ? -1 : n;
}

/* chemistry.f90:1382-1437 */
static double one_body(const orc_chem *s, det_t up, det_t dn) {
  double e = 0; int n1 = s->norb + 1;
  for (det_t d = up; d; d &= d - 1) { int i = trailz(d) + 1; e = e + IV(i, i, n1, n1); }
  if (dn == up) e = e * 2.0;
  else for (det_t d = dn; d; d &= d - 1) { int i = trailz(d) + 1; e = e + IV(i, i, n1, n1); }
  return e;
}
/* chemistry.f90:1609-1841, the non-incremental ("usual way") branch 1773-1838.  The
 * reference also has a stateful incremental branch (1642-1772) whose round-off depends
 * on call history; it is not restated (tolerance 1e-12 Ha on H_ii). */
static double two_body(const orc_chem *s, det_t up, det_t dn) {
  double ex = 0, di = 0; int n = s->norb;
  for (int i = 1; i <= n; i++) if (btest(up, i - 1))
    for (int j = i + 1; j <= n; j++) if (btest(up, j - 1)) ex = ex - IV(i, j, j, i);
  if (dn == up) ex = ex * 2.0;
  else if (dn != 0)
    for (int i = 1; i <= n; i++) if (btest(dn, i - 1))
      for (int j = i + 1; j <= n; j++) if (btest(dn, j - 1)) ex = ex - IV(i, j, j, i);
  for (int i = 1; i <= n; i++) {
    if (btest(up, i - 1)) {
      for (int j = i + 1; j <= n; j++) if (btest(up, j - 1)) di = di + IV(i, i, j, j);
      for (int j = 1; j <= n; j++) if (btest(dn, j - 1)) di = di + IV(i, i, j, j);
    }
    if (btest(dn, i - 1))
      for (int j = i + 1; j <= n; j++) if (btest(dn, j - 1)) di = di + IV(i, i, j, j);
  }
  return ex + di;
}
/* chemistry.f90:1439-1478 */
static double one_body_single(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd) {
  int n1 = s->norb + 1;
  if (iu != ju) {
    int ib = trailz(iu & ~ju), jb = trailz(ju & ~iu);
    return orc_permutation_factor(iu, ju) * IV(ib + 1, jb + 1, n1, n1);
  }
  int ib = trailz(id & ~jd), jb = trailz(jd & ~id);
  return orc_permutation_factor(id, jd) * IV(ib + 1, jb + 1, n1, n1);
}
/* chemistry.f90:1845-1930 */
static double two_body_single(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd) {
  double e = 0;
  det_t a = iu, b = id, aj = ju;           /* a = spin channel that changes */
  if (iu == ju) { a = id; b = iu; aj = jd; }
  int ib = trailz(a & ~aj) + 1, jb = trailz(aj & ~a) + 1;
  for (det_t d = a; d; d &= d - 1) {
    int i = trailz(d) + 1;
    if (i != ib && i != jb) e = e - IV(ib, i, i, jb) + IV(ib, jb, i, i);
  }
  for (det_t d = b; d; d &= d - 1) { int i = trailz(d) + 1; e = e + IV(ib, jb, i, i); }
  return orc_permutation_factor(a, aj) * e;
}
/* chemistry.f90:1934-2001 */
static double two_body_double(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd) {
  int g, i1, i2, j1, j2;
  if (iu == ju) {
    orc_permutation_factor2(id, jd, &g, &i1, &i2, &j1, &j2);
    return g * (IV(i1 + 1, j1 + 1, i2 + 1, j2 + 1) - IV(i1 + 1, j2 + 1, i2 + 1, j1 + 1));
  } else if (id == jd) {
    orc_permutation_factor2(iu, ju, &g, &i1, &i2, &j1, &j2);
    return g * (IV(i1 + 1, j1 + 1, i2 + 1, j2 + 1) - IV(i1 + 1, j2 + 1, i2 + 1, j1 + 1));
  }
  i1 = trailz(iu & ~ju); j1 = trailz(ju & ~iu);
  i2 = trailz(id & ~jd); j2 = trailz(jd & ~id);
  return orc_permutation_factor(iu, ju) * orc_permutation_factor(id, jd) * IV(i1 + 1, j1 + 1, i2 + 1, j2 + 1);
}
/* chemistry.f90:1260-1320 */
double orc_hamiltonian_chem(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd, int level) {
  if (level == 0) return one_body(s, iu, id) + two_body(s, iu, id) + s->nuclear;
  if (level == 1) return one_body_single(s, iu, id, ju, jd) + two_body_single(s, iu, id, ju, jd);
  if (level == 2) return two_body_double(s, iu, id, ju, jd);
  return 0.0;
}
/* chemistry.f90:1323-1377 */
double orc_hamiltonian_chem_time_sym(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd) {
  const double sqrt2 = sqrt(2.0), sqrt2inv = 1.0 / sqrt2;
  double m1 = 0, m2 = 0, norm_ketinv = 1, norm_bra = 1; int check = 1, lev;
  if (ju == jd) norm_ketinv = sqrt2inv;
  if (iu == id) { norm_bra = sqrt2; check = 0; }
  lev = (iu == ju && id == jd) ? 0 : orc_excitation_level(iu, id, ju, jd);
  if (lev >= 0) m1 = orc_hamiltonian_chem(s, iu, id, ju, jd, lev);
  if (check) {
    if (ju != jd) {
      lev = orc_excitation_level(id, iu, ju, jd);
      if (lev >= 0) m2 = orc_hamiltonian_chem(s, id, iu, ju, jd, lev);
    } else m2 = m1;
  }
  return (norm_bra * norm_ketinv) * (m1 + (s->z * m2));
}
/* semistoch.f90:2234-2302 (chem branch) */
double orc_hamiltonian(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd) {
  if (s->time_sym) return orc_hamiltonian_chem_time_sym(s, iu, id, ju, jd);
  int lev = orc_excitation_level(iu, id, ju, jd);
  return lev < 0 ? 0.0 : orc_hamiltonian_chem(s, iu, id, ju, jd, lev);
}

/* chemistry.f90:10525-10560 */
static int det_sym(const orc_chem *s, det_t up, det_t dn) {
  int sym = 1;
  for (det_t d = up; d; d &= d - 1) sym = s->prod[sym][s->orbsym[trailz(d) + 1]];
  for (det_t d = dn; d; d &= d - 1) sym = s->prod[sym][s->orbsym[trailz(d) + 1]];
  return sym;
}

/* chemistry.f90:6471-6815 : every symmetry-allowed single and double (element may be 0).
 * Order: the det itself, up-up, dn-dn, up-dn doubles, up singles, dn singles.  With
 * sym_filter>0 keep dets of that total symmetry instead of the pairwise symmetry test. */
static int connected_all(const orc_chem *s, det_t up, det_t dn, det_t *cu, det_t *cd, double *el,
                         int cap, int sym_filter) {
  int fu[ORC_MAXORB], eu[ORC_MAXORB], fd[ORC_MAXORB], ed[ORC_MAXORB], nfu = 0, neu = 0, nfd = 0, ned = 0;
  const int *os = s->orbsym; int n = 0, nc = s->n_core_orb;
  for (int i = 0; i < s->norb; i++) {
    if (btest(up, i)) fu[nfu++] = i; else eu[neu++] = i;
    if (btest(dn, i)) fd[nfd++] = i; else ed[ned++] = i;
  }
#define PUSH(U, D, LEV) do { if (n < cap) { cu[n] = (U); cd[n] = (D); \
    if (el) el[n] = s->time_sym ? orc_hamiltonian_chem_time_sym(s, up, dn, (U), (D)) \
                                : orc_hamiltonian_chem(s, up, dn, (U), (D), (LEV)); } n++; } while (0)
  PUSH(up, dn, 0);
  for (int i = nc; i < nfu - 1; i++) for (int j = i + 1; j < nfu; j++)
    for (int k = 0; k < neu - 1; k++) for (int l = k + 1; l < neu; l++) {
      det_t t = (up & ~BIT(fu[i]) & ~BIT(fu[j])) | BIT(eu[k]) | BIT(eu[l]);
      int keep = sym_filter ? det_sym(s, t, dn) == sym_filter
                            : s->prod[os[fu[i] + 1]][os[fu[j] + 1]] == s->prod[os[eu[k] + 1]][os[eu[l] + 1]];
      if (keep) PUSH(t, dn, 2);
    }
  for (int i = nc; i < nfd - 1; i++) for (int j = i + 1; j < nfd; j++)
    for (int k = 0; k < ned - 1; k++) for (int l = k + 1; l < ned; l++) {
      det_t t = (dn & ~BIT(fd[i]) & ~BIT(fd[j])) | BIT(ed[k]) | BIT(ed[l]);
      int keep = sym_filter ? det_sym(s, up, t) == sym_filter
                            : s->prod[os[fd[i] + 1]][os[fd[j] + 1]] == s->prod[os[ed[k] + 1]][os[ed[l] + 1]];
      if (keep) PUSH(up, t, 2);
    }
  for (int i = nc; i < nfu; i++) for (int j = nc; j < nfd; j++)
    for (int k = 0; k < neu; k++) for (int l = 0; l < ned; l++) {
      det_t tu = (up & ~BIT(fu[i])) | BIT(eu[k]), td = (dn & ~BIT(fd[j])) | BIT(ed[l]);
      int keep = sym_filter ? det_sym(s, tu, td) == sym_filter
                            : s->prod[os[fu[i] + 1]][os[fd[j] + 1]] == s->prod[os[eu[k] + 1]][os[ed[l] + 1]];
      if (keep) PUSH(tu, td, 2);
    }
  for (int i = nc; i < nfu; i++) for (int k = 0; k < neu; k++) {
    det_t t = (up & ~BIT(fu[i])) | BIT(eu[k]);
    int keep = sym_filter ? det_sym(s, t, dn) == sym_filter : os[fu[i] + 1] == os[eu[k] + 1];
    if (keep) PUSH(t, dn, 1);
  }
  for (int i = nc; i < nfd; i++) for (int k = 0; k < ned; k++) {
    det_t t = (dn & ~BIT(fd[i])) | BIT(ed[k]);
    int keep = sym_filter ? det_sym(s, up, t) == sym_filter : os[fd[i] + 1] == os[ed[k] + 1];
    if (keep) PUSH(up, t, 1);
  }
#undef PUSH
  return n;
}
int orc_find_connected_dets_chem(const orc_chem *s, det_t up, det_t dn, det_t *cu, det_t *cd,
                                 double *elems, int cap) {
  return connected_all(s, up, dn, cu, cd, elems, cap, 0);
}

/* chemistry.f90:9378-9442 */
static void compute_orbital_energies(const orc_chem *s, det_t hu, det_t hd, double *oe) {
  int n = s->norb, n1 = n + 1;
  for (int i = 1; i <= n; i++) {
    double ex = 0, di = 0;
    for (int j = 1; j <= n; j++) {
      if (j != i && btest(hu, j - 1)) ex = ex - IV(i, j, j, i);
      if (j != i && btest(hd, j - 1)) ex = ex - IV(i, j, j, i);
    }
    for (int j = 1; j <= n; j++) if (j != i && btest(hu, j - 1)) di = di + IV(i, i, j, j);
    for (int j = 1; j <= n; j++) if (btest(hd, j - 1)) di = di + IV(i, i, j, j);
    for (int j = 1; j <= n; j++) if (j != i && btest(hd, j - 1)) di = di + IV(i, i, j, j);
    for (int j = 1; j <= n; j++) if (btest(hu, j - 1)) di = di + IV(i, i, j, j);
    oe[i] = IV(i, i, n1, n1) + .5 * (ex + di);
  }
}

/* chemistry.f90:538-869 (FCIDUMP branch), 8921-9022 sort_integrals, 10359-10522 auto HF */
orc_chem *orc_chem_load(const char *path, int nelec, int nup, const char *pg, int time_sym, int z,
                        int n_core_orb, int hf_mode, int hf_symmetry) {
  FILE *f = fopen(path, "r");
  if (!f) return NULL;
  orc_chem *s = (orc_chem *)calloc(1, sizeof(orc_chem));
  s->nelec = nelec; s->nup = nup; s->ndn = nelec - nup; s->time_sym = time_sym; s->z = z;
  s->n_core_orb = n_core_orb;
  if (init_point_group(s, pg)) { free(s); fclose(f); return NULL; }
  /* header: NORB= and ORBSYM= up to the line holding &END or / */
  char line[4096]; char hdr[16384] = "";
  while (fgets(line, sizeof line, f)) {
    strncat(hdr, line, sizeof hdr - strlen(hdr) - 1);
    if (strstr(line, "&END") || strstr(line, "&end") || strchr(line, '/')) break;
  }
  char *p = strstr(hdr, "NORB=");
  if (!p) { free(s); fclose(f); return NULL; }
  s->norb = atoi(p + 5);
  int n = s->norb, n1 = n + 1;
  if (n > ORC_MAXORB) { free(s); fclose(f); return NULL; }
  p = strstr(hdr, "ORBSYM=");
  if (p) { p += 7; for (int i = 1; i <= n; i++) { s->orbsym[i] = (int)strtol(p, &p, 10); while (*p == ',' || isspace((unsigned char)*p)) p++; } }
  else for (int i = 1; i <= n; i++) s->orbsym[i] = 1;
  /* chemistry.f90:384-394 : identity-order combine_2 while reading */
  for (int i = 1; i <= n1; i++) { s->orb_order[i] = i; s->orb_order_inv[i] = i; }
  for (int i = 1; i <= n; i++) for (int j = 1; j <= n; j++)
    s->combine_2[i][j] = (i > j) ? (i * (i - 1)) / 2 + j : (j * (j - 1)) / 2 + i;
  s->combine_2[n1][n1] = (n1 * n) / 2 + n1;
  s->n_int = orc_integral_index(s, n1, n1, n1, n1);
  s->integrals = (double *)calloc((size_t)s->n_int + 1, sizeof(double));
  double v; int a, b, c, d;
  while (fscanf(f, "%lf %d %d %d %d", &v, &a, &b, &c, &d) == 5) {
    if (a == 0) a = n1; if (b == 0) b = n1; if (c == 0) c = n1; if (d == 0) d = n1;
    /* one-body entries come as (p,q,0,0): combine_2(p,q) then (n1,n1): fine; guard (p,n1) */
    if (fabs(v) > 1e-9) s->integrals[orc_integral_index(s, a, b, c, d)] = v;
  }
  fclose(f);
  s->nuclear = s->integrals[orc_integral_index(s, n1, n1, n1, n1)];
  /* starting det: first orbitals (chemistry.f90:700-712) */
  det_t hu = maskr(s->nup), hd = maskr(s->ndn);
  if (hf_mode == 1) {   /* auto_assign_hci0_occs with a nonzero input det: CISD descent */
    int cap = 200000; det_t *cu = malloc(cap * sizeof(det_t)), *cd = malloc(cap * sizeof(det_t));
    det_t du = 0, dd = 0;
    while (!(hu == du && hd == dd)) {
      if (du != 0) { hu = du; hd = dd; }
      int nc = connected_all(s, hu, hd, cu, cd, NULL, cap, hf_symmetry);
      double best = 1e50;
      for (int i = 0; i < nc; i++) {
        if (det_sym(s, cu[i], cd[i]) != hf_symmetry) continue;
        if (time_sym && z < 0 && cu[i] == cd[i]) continue;
        double e = time_sym ? orc_hamiltonian_chem_time_sym(s, cu[i], cd[i], cu[i], cd[i])
                            : orc_hamiltonian_chem(s, cu[i], cd[i], cu[i], cd[i], 0);
        if (e < best) { best = e; du = cu[i]; dd = cd[i]; }
      }
    }
    free(cu); free(cd);
  }
  /* sort_integrals */
  double oe[ORC_MAXORB + 1], tmp[ORC_MAXORB + 1];
  compute_orbital_energies(s, hu, hd, oe);
  for (int i = 1; i <= n; i++) {
    tmp[i] = oe[i];
    if (btest(hu, i - 1)) tmp[i] = tmp[i] - 1.e9;
    if (btest(hd, i - 1)) tmp[i] = tmp[i] - 1.e9;
  }
  for (int i = 1; i <= n; i++) {
    double mn = tmp[1]; for (int j = 2; j <= n; j++) if (tmp[j] < mn) mn = tmp[j];
    for (int j = 1; j <= n; j++) if (tmp[j] == mn) { s->orb_order[i] = j; s->orb_order_inv[j] = i; tmp[j] = 1.e99; break; }
  }
  int os_new[ORC_MAXORB + 1];
  for (int i = 1; i <= n; i++) { os_new[i] = s->orbsym[s->orb_order[i]]; s->orbital_energies[i] = oe[s->orb_order[i]]; }
  for (int i = 1; i <= n; i++) s->orbsym[i] = os_new[i];
  det_t nu = 0, nd = 0;
  for (det_t t = hu; t; t &= t - 1) nu |= BIT(s->orb_order_inv[trailz(t) + 1] - 1);
  for (det_t t = hd; t; t &= t - 1) nd |= BIT(s->orb_order_inv[trailz(t) + 1] - 1);
  if (time_sym && nd < nu) { det_t t = nu; nu = nd; nd = t; }
  s->hf_up = nu; s->hf_dn = nd;
  /* chemistry.f90:856-866 : combine_2 through orb_order */
  for (int i = 1; i <= n; i++) { int aa = s->orb_order[i];
    for (int j = 1; j <= n; j++) { int bb = s->orb_order[j];
      s->combine_2[i][j] = (aa > bb) ? (aa * (aa - 1)) / 2 + bb : (bb * (bb - 1)) / 2 + aa; } }
  /* setup_orb_by_symm chemistry.f90:2500-2504 */
  for (int i = 1; i <= n; i++) { int sy = s->orbsym[i]; s->which_orb_by_sym[sy][++s->num_orb_by_sym[sy]] = i; }
  return s;
}
void orc_chem_free(orc_chem *s) {
  if (!s) return;
  free(s->integrals); free(s->hb_r); free(s->hb_s); free(s->hb_absH); free(s->pq_ind); free(s->pq_count); free(s);
}
void orc_free(void *p) { free(p); }

/* ============================================================ HCI tables */
/* chemistry.f90:9137-9151 */
static int64_t combine_2_indices(int i, int j) { return i > j ? ((int64_t)i * (i - 1)) / 2 + j : ((int64_t)j * (j - 1)) / 2 + i; }
/* chemistry.f90:9615-9646 */
static double dexc_no_ref(const orc_chem *s, int p, int q, int r, int t) {
  int n = s->norb;
  if (p == q || r == t || p == r || q == t || p == t || q == r) return 0.0;
  if (p <= n && q <= n) return orc_hamiltonian_chem(s, BIT(p - 1) | BIT(q - 1), 0, BIT(r - 1) | BIT(t - 1), 0, 2);
  if (p > n && q > n) return orc_hamiltonian_chem(s, 0, BIT(p - n - 1) | BIT(q - n - 1), 0, BIT(r - n - 1) | BIT(t - n - 1), 2);
  return orc_hamiltonian_chem(s, BIT(p - 1), BIT(q - n - 1), BIT(r - 1), BIT(t - n - 1), 2);
}
typedef struct { int r, s; double a; int seq; } rsh;
static int cmp_rsh(const void *x, const void *y) {   /* descending absH, stable */
  const rsh *a = x, *b = y;
  if (a->a > b->a) return -1; if (a->a < b->a) return 1; return a->seq - b->seq;
}
/* chemistry.f90:900-993 */
void orc_chem_setup_hb(orc_chem *s) {
  int n = s->norb;
  s->n_pq = (int)combine_2_indices(n, 2 * n);
  s->pq_ind = calloc(s->n_pq + 1, sizeof(int64_t)); s->pq_count = calloc(s->n_pq + 1, sizeof(int));
  int64_t cap = 1 << 16, cnt = 0; rsh *all = malloc(cap * sizeof(rsh));
  s->max_double = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int p = 1; p <= n; p++) {
      int sym_p = s->orbsym[p];
      int q0 = pass ? n + p : p + 1, q1 = pass ? 2 * n : n;
      for (int q = q0; q <= q1; q++) {
        int sym_q = s->prod[sym_p][s->orbsym[pass ? q - n : q]];
        int64_t e = combine_2_indices(p, q), first = cnt;
        s->pq_ind[e] = cnt + 1; s->pq_count[e] = 0;
        for (int r = 1; r <= n; r++) {
          int sym_r = s->prod[sym_q][s->orbsym[r]];
          for (int k = 1; k <= s->num_orb_by_sym[sym_r]; k++) {
            int t = s->which_orb_by_sym[sym_r][k];
            if (!pass && t < r) continue;
            if (pass) t += n;
            double h = fabs(dexc_no_ref(s, p, q, r, t));
            if (h != 0.0) {
              if (cnt == cap) { cap *= 2; all = realloc(all, cap * sizeof(rsh)); }
              all[cnt].r = r; all[cnt].s = t; all[cnt].a = h; all[cnt].seq = (int)(cnt - first); cnt++;
              s->pq_count[e]++;
            }
          }
        }
        if (s->pq_count[e] > 1) qsort(all + first, s->pq_count[e], sizeof(rsh), cmp_rsh);
        if (s->pq_count[e] > 0 && all[first].a > s->max_double) s->max_double = all[first].a;
      }
    }
  s->n_hb = cnt;
  s->hb_r = malloc(cnt * sizeof(int)); s->hb_s = malloc(cnt * sizeof(int)); s->hb_absH = malloc(cnt * sizeof(double));
  for (int64_t i = 0; i < cnt; i++) { s->hb_r[i] = all[i].r; s->hb_s[i] = all[i].s; s->hb_absH[i] = all[i].a; }
  free(all);
}

/* chemistry.f90:6926-6947 and 7087-7108: the optional core / virtual masks.  Returns 1 when the new determinant is to be skipped. */
static int active_space_skip(const orc_chem *s, det_t nu, det_t nd) {
  if (!s->as_mode) return 0;
  const int outside = ((s->as_core_up & nu) != s->as_core_up) || ((s->as_core_dn & nd) != s->as_core_dn) || (s->as_virt_up & nu) != 0 || (s->as_virt_dn & nd) != 0;
  return s->as_mode == 1 ? outside : !outside;
}
void orc_set_active_space(orc_chem *s, det_t core_up, det_t core_dn, det_t virt_up, det_t virt_dn, int mode) {
  s->as_mode = mode; s->as_core_up = core_up; s->as_core_dn = core_dn; s->as_virt_up = virt_up; s->as_virt_dn = virt_dn;
}
/* chemistry.f90:6819-7159 with matrix_elements present; active-space masks as set by orc_set_active_space; no eps_big */
int orc_find_important_connected_dets_chem(const orc_chem *s, det_t up, det_t dn, double eps,
                                           det_t *cu, det_t *cd, double *el, int cap) {
  const double sqrt2 = sqrt(2.0), sqrt2inv = 1.0 / sqrt2;
  int n = s->norb, nc = 0, occ_u[ORC_MAXORB], occ_d[ORC_MAXORB], nu = 0, nd = 0;
#define EMIT(U, D, M) do { if (nc < cap) { cu[nc] = (U); cd[nc] = (D); if (el) el[nc] = (M); } nc++; } while (0)
  EMIT(up, dn, 0.0);
  for (det_t t = up; t; t &= t - 1) occ_u[nu++] = trailz(t) + 1;
  for (det_t t = dn; t; t &= t - 1) occ_d[nd++] = trailz(t) + 1;
  for (int p = 1; p <= s->nelec; p++) {
    int isup = p <= s->nup, pe = isup ? occ_u[p - 1] : occ_d[p - s->nup - 1];
    for (int r = 1; r <= n; r++) {
      if (btest(isup ? up : dn, r - 1)) continue;
      if (s->orbsym[pe] != s->orbsym[r]) continue;
      det_t nu_ = up, nd_ = dn;
      if (isup) nu_ = (up & ~BIT(pe - 1)) | BIT(r - 1); else nd_ = (dn & ~BIT(pe - 1)) | BIT(r - 1);
      if (active_space_skip(s, nu_, nd_)) continue;
      if (s->time_sym) {
        if (nu_ == nd_ && s->z < 0) continue;
        if (up == nd_ && dn == nu_) continue;
      }
      double m = orc_hamiltonian_chem(s, up, dn, nu_, nd_, 1);
      if (fabs(m) < eps) continue;
      if (s->time_sym) {
        if (up == dn && nu_ != nd_) m = sqrt2inv * m;
        if (nu_ == nd_ && up != dn) m = sqrt2 * m;
        if (nu_ > nd_) { det_t t = nu_; nu_ = nd_; nd_ = t; m = s->z * m; }
      }
      EMIT(nu_, nd_, m);
    }
  }
  if (eps > s->max_double) return nc;
  int pe1[ORC_MAXORB * ORC_MAXORB], pe2[ORC_MAXORB * ORC_MAXORB], np = 0;
  for (int p = 0; p < nu; p++) for (int q = p + 1; q < nu; q++) { pe1[np] = occ_u[p]; pe2[np++] = occ_u[q]; }
  for (int p = 0; p < nd; p++) for (int q = p + 1; q < nd; q++) { pe1[np] = occ_d[p] + n; pe2[np++] = occ_d[q] + n; }
  for (int p = 0; p < nu; p++) for (int q = 0; q < nd; q++) { pe1[np] = occ_u[p]; pe2[np++] = occ_d[q] + n; }
  for (int ip = 0; ip < np; ip++) {
    int p = pe1[ip], q = pe2[ip], p2 = p, q2 = q;
    int both_dn = (p > n && q > n), swapped = (p <= n && q > n && p > q - n);
    if (both_dn) { p2 = p - n; q2 = q - n; }
    if (swapped) { p2 = q - n; q2 = p + n; }
    int64_t e = combine_2_indices(p2, q2);
    for (int h = 0; h < s->pq_count[e]; h++) {
      int64_t k = s->pq_ind[e] - 1 + h;
      if (s->hb_absH[k] <= eps) break;
      int r = s->hb_r[k], t = s->hb_s[k];
      if (both_dn) { r += n; t += n; }
      if (swapped) { int rt = t - n; t = r + n; r = rt; }
      if (r <= n ? btest(up, r - 1) : btest(dn, r - n - 1)) continue;
      if (t <= n ? btest(up, t - 1) : btest(dn, t - n - 1)) continue;
      det_t nu_ = up, nd_ = dn;
      if (p <= n) nu_ &= ~BIT(p - 1); else nd_ &= ~BIT(p - n - 1);
      if (q <= n) nu_ &= ~BIT(q - 1); else nd_ &= ~BIT(q - n - 1);
      if (r <= n) nu_ |= BIT(r - 1); else nd_ |= BIT(r - n - 1);
      if (t <= n) nu_ |= BIT(t - 1); else nd_ |= BIT(t - n - 1);
      if (active_space_skip(s, nu_, nd_)) continue;
      if (s->time_sym) {
        if (nu_ == nd_ && s->z < 0) continue;
        if (up == nd_ && dn == nu_) continue;
      }
      double m = orc_hamiltonian_chem(s, up, dn, nu_, nd_, 2);
      if (s->time_sym) {
        if (up == dn && nu_ != nd_) m = sqrt2inv * m;
        if (nu_ == nd_ && up != dn) m = sqrt2 * m;
        if (nu_ > nd_) { det_t tt = nu_; nu_ = nd_; nd_ = tt; m = s->z * m; }
      }
      EMIT(nu_, nd_, m);
    }
  }
#undef EMIT
  return nc;
}

/* =================================================================== SpMV */
/* more_tools.f90:3622-3670, general branch; 1-based indices as in the reference */
void orc_spmv_sym_upper(int64_t n, const int64_t *row_counts, const int64_t *indices,
                        const double *values, const double *x, double *y) {
  for (int64_t i = 0; i < n; i++) y[i] = 0.0;
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < row_counts[i]; j++, k++) {
      int64_t m = indices[k] - 1;
      y[i] = y[i] + values[k] * x[m];
      if (i != m) y[m] = y[m] + values[k] * x[i];
    }
}

/* ================================================== sparse H among a list */
static int64_t bsearch_det(const det_t *up, const det_t *dn, int64_t n, det_t u, det_t d) {
  int64_t lo = 0, hi = n - 1;
  while (lo <= hi) {
    int64_t mid = (lo + hi) / 2;
    if (up[mid] == u && dn[mid] == d) return mid;
    if (up[mid] < u || (up[mid] == u && dn[mid] < d)) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}
/* Lower triangle, diagonal first in each row, then columns ascending.  The list must be
 * sorted by (up,dn).  Connections are enumerated and looked up (pure function of the
 * list; the reference's own builder, chemistry.f90:7639-8010, is a "next" row). */
int64_t orc_build_sparse_ham(const orc_chem *s, int64_t n, const det_t *up, const det_t *dn,
                             int64_t **row_counts, int64_t **indices, double **values) {
  int64_t cap = n * 64 + 1024, nnz = 0;
  int64_t *rc = calloc(n, sizeof(int64_t)), *idx = malloc(cap * sizeof(int64_t));
  double *val = malloc(cap * sizeof(double));
  int ccap = 400000; det_t *cu = malloc(ccap * sizeof(det_t)), *cd = malloc(ccap * sizeof(det_t));
  int64_t *cols = malloc(ccap * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) {
    int nc = connected_all(s, up[i], dn[i], cu, cd, NULL, ccap, 0), m = 0;
    for (int c = 1; c < nc; c++) {
      det_t a = cu[c], b = cd[c];
      if (s->time_sym && a > b) { det_t t = a; a = b; b = t; }
      int64_t j = bsearch_det(up, dn, n, a, b);
      if (j >= 0 && j < i) cols[m++] = j;
    }
    /* sort + unique columns */
    for (int a = 1; a < m; a++) { int64_t v = cols[a]; int b = a - 1; while (b >= 0 && cols[b] > v) { cols[b + 1] = cols[b]; b--; } cols[b + 1] = v; }
    if (nnz + m + 1 > cap) { cap = 2 * cap + m; idx = realloc(idx, cap * sizeof(int64_t)); val = realloc(val, cap * sizeof(double)); }
    idx[nnz] = i + 1; val[nnz] = orc_hamiltonian(s, up[i], dn[i], up[i], dn[i]); nnz++; rc[i] = 1;
    int64_t last = -1;
    for (int a = 0; a < m; a++) {
      if (cols[a] == last) continue; last = cols[a];
      double h = orc_hamiltonian(s, up[i], dn[i], up[last], dn[last]);
      if (h == 0.0) continue;
      idx[nnz] = last + 1; val[nnz] = h; nnz++; rc[i]++;
    }
  }
  free(cu); free(cd); free(cols);
  *row_counts = rc; *indices = idx; *values = val;
  return nnz;
}

/* ===================================================== uniform2 proposal */
/* k-th (1-based) orbital of symmetry sym that is empty in det and != skip (1-based, 0=none).
 * Equivalent to the reference's "bump excite_to past occupied orbitals" loops
 * (chemistry.f90:4521-4530, 4594-4628) because both orbital lists are ascending. */
static int kth_open_sym(const orc_chem *s, det_t det, int sym, int k, int skip) {
  for (int a = 1; a <= s->num_orb_by_sym[sym]; a++) {
    int o = s->which_orb_by_sym[sym][a];
    if (btest(det, o - 1) || o == skip) continue;
    if (--k == 0) return o;
  }
  return 0;
}
static int kth_open(const orc_chem *s, det_t det, int k) {   /* chemistry.f90:4566-4574 */
  for (int o = 1; o <= s->norb; o++) { if (btest(det, o - 1)) continue; if (--k == 0) return o; }
  return 0;
}
static int nocc_sym(const orc_chem *s, det_t det, int sym) {
  int c = 0;
  for (det_t t = det; t; t &= t - 1) if (s->orbsym[trailz(t) + 1] == sym) c++;
  return c;
}
/* chemistry.f90:2003-2200 (uniform-proposal branch): are det_i and det_j connected, at which
 * excitation level, and with which probability would off_diagonal_move_chem have proposed
 * det_j from det_i (without the level factor n_single/n_total or n_double/n_total). */
static int is_connected_chem(const orc_chem *s, det_t iu, det_t id, det_t ju, det_t jd, int *level, double *prob) {
  int upc = 0, dnc = 0, du1 = 0, du2 = 0, du3 = 0, du4 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0;
  *level = -1; *prob = 0.0;
  if (iu != ju) {
    det_t a = iu & ~ju, b = ju & ~iu;
    upc = popcnt(a);
    if (upc > 2 || upc != popcnt(b)) return 0;
    du1 = trailz(a) + 1; du3 = trailz(b) + 1;
    if (upc == 2) { du2 = trailz(a & (a - 1)) + 1; du4 = trailz(b & (b - 1)) + 1; }
  }
  if (id != jd) {
    det_t a = id & ~jd, b = jd & ~id;
    dnc = popcnt(a);
    if (dnc > 2 || dnc != popcnt(b)) return 0;
    dd1 = trailz(a) + 1; dd3 = trailz(b) + 1;
    if (dnc == 2) { dd2 = trailz(a & (a - 1)) + 1; dd4 = trailz(b & (b - 1)) + 1; }
  }
  *level = upc + dnc;
  if (*level > 2) { *level = -1; return 0; }
  const int ne = s->nelec - 2 * s->n_core_orb;
  const det_t orbs = maskr(s->norb);
  if (*level == 1) {
    det_t det = id; int d1 = dd1, d2 = dd3;
    if (upc == 1) { det = iu; d1 = du1; d2 = du3; }
    int sym1 = s->orbsym[d1];
    if (sym1 != s->orbsym[d2]) return 0;
    int i_open = 0;
    for (int i = 1; i <= s->norb; i++) if (!btest(det, i - 1) && s->orbsym[i] == sym1) i_open++;
    *prob = 1.0 / ((ne) * (i_open));
    return 1;
  }
  if (*level == 2) {
    int d1, d2, d3, d4; det_t det1, det2; double tp, tp2;
    if (upc == 2) { d1 = du1; d2 = du2; d3 = du3; d4 = du4; det1 = iu | BIT(d3 - 1); det2 = iu | BIT(d4 - 1); tp = 2.0 / (s->norb - s->nup); tp2 = 2.0 / (s->norb - s->nup); }
    else if (dnc == 2) { d1 = dd1; d2 = dd2; d3 = dd3; d4 = dd4; det1 = id | BIT(d3 - 1); det2 = id | BIT(d4 - 1); tp = 2.0 / (s->norb - s->ndn); tp2 = 2.0 / (s->norb - s->ndn); }
    else { d1 = du1; d2 = dd1; d3 = du3; d4 = dd3; det2 = iu; det1 = id; tp = 1.0 / (s->norb - s->nup); tp2 = 1.0 / (s->norb - s->ndn); }
    int sym1 = s->prod[s->orbsym[d1]][s->orbsym[d2]];
    if (sym1 != s->prod[s->orbsym[d3]][s->orbsym[d4]]) return 0;
    int i_open = 0, i_open2 = 0;
    for (int i = 1; i <= s->norb; i++) if (!btest(det1, i - 1) && s->prod[s->orbsym[d3]][s->orbsym[i]] == sym1) i_open++;
    for (int i = 1; i <= s->norb; i++) if (!btest(det2, i - 1) && s->prod[s->orbsym[i]][s->orbsym[d4]] == sym1) i_open2++;
    (void)orbs;
    if (i_open == 0 && i_open2 != 0) *prob = (1.0 / ((ne) * (ne - 1))) * ((tp2 / (i_open2)));
    if (i_open2 == 0 && i_open != 0) *prob = (1.0 / ((ne) * (ne - 1))) * ((tp / (i_open)));
    if (i_open2 != 0 && i_open != 0) *prob = (1.0 / ((ne) * (ne - 1))) * ((tp / (i_open)) + (tp2 / (i_open2)));
    return 1;
  }
  return 1;      /* level 0: same determinant */
}

/* chemistry.f90:4237-5084, importance_sampling=0 (both time_sym branches).  Returns det_j and
 * weight_j = -tau*H_ij/proposal_prob (0 and det_j partly built if no valid move, exactly
 * like the reference's early returns).  n_draws = rannyu calls consumed. */
void orc_off_diagonal_move_chem(const orc_chem *s, orc_rng *g, double tau, det_t iu, det_t id,
                                det_t *pju, det_t *pjd, double *weight_j, int *n_draws) {
  int nup = s->nup, ndn = s->ndn, norb = s->norb, nc = s->n_core_orb, nelec = s->nelec, draws = 0;
  int occ[2 * ORC_MAXORB], n_occ_up = 0, n_occ = 0;
  for (det_t t = iu; t; t &= t - 1) occ[n_occ++] = trailz(t) + 1;
  n_occ_up = n_occ;
  for (det_t t = id; t; t &= t - 1) occ[n_occ++] = trailz(t) + 1;
  det_t ju = iu, jd = id;
  *weight_j = 0; *pju = ju; *pjd = jd;
  int n_single = (nup - nc) * (norb - nup) + (ndn - nc) * (norb - ndn);
  int n_double_up = (nup - nc) * (nup - nc - 1) * (norb - nup) * (norb - nup - 1) / 4;
  int n_double_dn = (ndn - nc) * (ndn - nc - 1) * (norb - ndn) * (norb - ndn - 1) / 4;
  int n_double_both = (nup - nc) * (norb - nup) * (ndn - nc) * (norb - ndn);
  int n_double = n_double_up + n_double_dn + n_double_both, n_total = n_single + n_double;
  int level, e1 = 0, e2 = 0, tot_spin = 0;
  double prob = 1.0;
#define RI(n) (draws++, orc_random_int(g, (n)))
  if (RI(n_total) > n_single) {
    level = 2; prob = n_double / (double)n_total;
    e1 = RI(nelec - 2 * nc); e2 = RI(nelec - 2 * nc - 1);
    if (e2 == e1) e2 = nelec - 2 * nc;
    if (e1 > nup - nc) { tot_spin -= 1; e1 += 2 * nc; } else { tot_spin += 1; e1 += nc; }
    if (e2 > nup - nc) { tot_spin -= 1; e2 += 2 * nc; } else { tot_spin += 1; e2 += nc; }
  } else {
    level = 1; prob = prob * n_single / (n_total * 1.0);
    e1 = RI(nelec - 2 * nc);
    if (e1 > nup - nc) { tot_spin = -1; e1 += 2 * nc; } else { tot_spin = 1; e1 += nc; }
  }
  { int t = e1 + e2; e1 = t - (e1 > e2 ? e1 : e2); e2 = t - e1; }   /* 4460-4462 */
  int sym1 = 1, o;
  if (level == 1) {
    o = occ[e2 - 1];
    if (e2 <= n_occ_up) ju &= ~BIT(o - 1); else jd &= ~BIT(o - 1);
    sym1 = s->orbsym[o];
  } else {
    o = occ[e1 - 1];
    if (e1 <= n_occ_up) ju &= ~BIT(o - 1); else jd &= ~BIT(o - 1);
    sym1 = s->orbsym[o];
    o = occ[e2 - 1];
    if (e2 <= n_occ_up) ju &= ~BIT(o - 1); else jd &= ~BIT(o - 1);
    sym1 = s->prod[sym1][s->orbsym[o]];
  }
  *pju = ju; *pjd = jd;
  int i_open, to1, to2, sp1, sym2; double temp1;
  if (level == 1) {
    prob = prob / (nelec - 2 * nc);
    det_t occdet = (tot_spin == 1) ? iu : id;
    i_open = s->num_orb_by_sym[sym1] - nocc_sym(s, occdet, sym1);
    if (i_open == 0) goto done;
    to1 = RI(i_open); prob = prob / i_open;
    o = kth_open_sym(s, occdet, sym1, to1, 0);
    if (tot_spin == 1) ju |= BIT(o - 1); else jd |= BIT(o - 1);
  } else {
    prob = prob * 2.0 / (1.0 * (nelec - 2 * nc) * (nelec - 2 * nc - 1));
    if (tot_spin == 2 || tot_spin == -2) {
      det_t occdet = (tot_spin == 2) ? iu : id; int nsp = (tot_spin == 2) ? nup : ndn;
      to1 = RI(norb - nsp); prob = prob / (norb - nsp);
      sp1 = kth_open(s, occdet, to1);
      if (tot_spin == 2) ju |= BIT(sp1 - 1); else jd |= BIT(sp1 - 1);
      *pju = ju; *pjd = jd;
      sym2 = s->prod[s->orbsym[sp1]][sym1];
      int same = (sym2 == s->orbsym[sp1]);
      i_open = s->num_orb_by_sym[sym2] - nocc_sym(s, occdet, sym2) - (same ? 1 : 0);
      if (i_open == 0) goto done;
      to2 = RI(i_open); temp1 = 1.0 / i_open;
      o = kth_open_sym(s, occdet, sym2, to2, same ? sp1 : 0);
      if (tot_spin == 2) ju |= BIT(o - 1); else jd |= BIT(o - 1);
      int sy = s->prod[sym2][sym1];
      i_open = s->num_orb_by_sym[sy] - nocc_sym(s, occdet, sy) - (same ? 1 : 0);
      if (i_open == 0) prob = prob * temp1; else prob = prob * (temp1 + (1.0 / i_open));
    } else {
      prob = prob * 1.0 / (2 * norb - ndn - nup);
      to1 = RI(2 * norb - nup - ndn);
      det_t d1, d2; int first_up = (to1 <= norb - nup);
      if (first_up) { d1 = iu; d2 = id; } else { d1 = id; d2 = iu; to1 -= (norb - nup); }
      sp1 = kth_open(s, d1, to1);
      if (first_up) ju |= BIT(sp1 - 1); else jd |= BIT(sp1 - 1);
      *pju = ju; *pjd = jd;
      sym2 = s->prod[sym1][s->orbsym[sp1]];
      i_open = s->num_orb_by_sym[sym2] - nocc_sym(s, d2, sym2);
      if (i_open == 0) goto done;
      to2 = RI(i_open); temp1 = 1.0 / i_open;
      o = kth_open_sym(s, d2, sym2, to2, 0);
      if (first_up) jd |= BIT(o - 1); else ju |= BIT(o - 1);
      int sy = s->prod[sym2][sym1];
      i_open = s->num_orb_by_sym[sy] - nocc_sym(s, d1, sy);
      if (i_open != 0) prob = prob * (temp1 + (1.0 / i_open)); else prob = prob * temp1;
    }
  }
  *pju = ju; *pjd = jd;
  if (s->time_sym) {                               /* chemistry.f90:4988-5044 */
    const double sqrt2 = sqrt(2.0);
    const double norm_i = (iu == id) ? sqrt2 : 1.0;
    double me;
    if ((ju == iu && jd == id) || (jd == iu && ju == id)) goto done;     /* already in the diagonal */
    if (ju == jd) {
      if (s->z != 1) goto done;
      me = orc_hamiltonian_chem(s, iu, id, ju, jd, level);
      me = (sqrt2 / norm_i) * me;
    } else {
      double m1 = orc_hamiltonian_chem(s, iu, id, ju, jd, level), psym; int lsym;
      if (is_connected_chem(s, iu, id, jd, ju, &lsym, &psym)) {
        double m2 = orc_hamiltonian_chem(s, iu, id, jd, ju, lsym);
        if (lsym == 1) prob = prob + (psym * (n_single / (double)n_total));
        if (lsym == 2) prob = prob + (psym * (n_double / (double)n_total));
        me = (1.0 / norm_i) * (m1 + s->z * m2);
      } else me = (1.0 / norm_i) * (m1);
    }
    if (ju > jd) { det_t t = ju; ju = jd; jd = t; me = me * s->z; }
    *pju = ju; *pjd = jd;
    *weight_j = -tau * me / prob;
  } else {
    double me = orc_hamiltonian_chem(s, iu, id, ju, jd, level);
    *weight_j = -tau * me / prob;
  }
done:
#undef RI
  if (n_draws) *n_draws = draws;
}

/* ================================================================== walk */
orc_walk *orc_walk_new(int64_t mwalk) {
  orc_walk *w = calloc(1, sizeof(orc_walk));
  w->mwalk = mwalk;
  w->up = calloc(mwalk, sizeof(det_t)); w->dn = calloc(mwalk, sizeof(det_t));
  w->wt = calloc(mwalk, sizeof(double));
  w->imp_distance = malloc(mwalk); w->initiator = calloc(mwalk, 1);
  w->matrix_elements = malloc(mwalk * sizeof(double));
  w->e_num_walker = malloc(mwalk * sizeof(double)); w->e_den_walker = malloc(mwalk * sizeof(double));
  for (int64_t i = 0; i < mwalk; i++) {            /* do_walk.f90:1133-1155 */
    w->imp_distance[i] = 1; w->matrix_elements[i] = 1e51; w->e_num_walker[i] = 1e51; w->e_den_walker[i] = 1e51;
  }
  return w;
}
void orc_walk_free(orc_walk *w) {
  if (!w) return;
  free(w->up); free(w->dn); free(w->wt); free(w->imp_distance); free(w->initiator);
  free(w->matrix_elements); free(w->e_num_walker); free(w->e_den_walker);
  free(w->prj_counts); free(w->prj_indices); free(w->prj_values);
  free(w->ct_up); free(w->ct_dn); free(w->ct_num); free(w->ct_den); free(w->sign_perm); free(w);
}

/* do_walk.f90:5169-5197 + 5411-5614 : stable sort on (up,dn), all 8 arrays permuted */
static void msort_idx(const det_t *up, const det_t *dn, int64_t *a, int64_t *tmp, int64_t n) {
  if (n < 2) return;
  int64_t na = (n + 1) / 2, nb = n - na;
  msort_idx(up, dn, a, tmp, na); msort_idx(up, dn, a + na, tmp, nb);
  memcpy(tmp, a, na * sizeof(int64_t));
  int64_t i = 0, j = na, k = 0;
  while (i < na && j < n) {
    int64_t x = tmp[i], y = a[j];
    if (up[x] < up[y] || (up[x] == up[y] && dn[x] <= dn[y])) a[k++] = tmp[i++]; else a[k++] = a[j++];
  }
  while (i < na) a[k++] = tmp[i++];
}
/* ---- ownership: mpi_routines.f90:354-379 (djb_hash), 257-289 (hash), 419-445 (get_det_owner) ----
 * hash = PRIME; for word in (det_up, det_dn + OFFSET): test = IEOR(word, X'5555555555555555') * PRIME;
 * for j = 0, 8, ..., 120: hash = (ishft(hash,5) + hash) + ishft(test, -j).   All INTEGER(16), wrapping; ishft with a
 * negative count is a LOGICAL right shift; the BOZ constant fills the low 64 bits only. */
typedef unsigned __int128 u128_t;
static u128_t orc_u128(uint64_t lo, uint64_t hi) { return ((u128_t)hi << 64) | (u128_t)lo; }
static u128_t orc_djb_hash_u(u128_t det_up, u128_t det_dn) {
  const u128_t PRIME = orc_u128(0x000000000000013Bull, 0x0000000001000000ull);            /* 309485009821345068724781371 = 2^88 + 315 (the FNV-128 prime) */
  const u128_t OFFSET = orc_u128(0x62B821756295C58Dull, 0x6C62272E07BB0142ull);           /* 144066263297769815596495629667062367629 */
  u128_t hash = PRIME;
  for (int i = 1; i <= 2; i++) {
    u128_t tmp = (i == 1) ? det_up : det_dn;
    if (i == 2) tmp = tmp + OFFSET;
    for (int j = 0; j < 128; j += 8) {
      u128_t test_nk = tmp ^ (u128_t)0x5555555555555555ull;
      test_nk = test_nk * PRIME;
      hash = ((hash << 5) + hash) + (test_nk >> j);
    }
  }
  return hash;
}
void orc_djb_hash(uint64_t up_lo, uint64_t up_hi, uint64_t dn_lo, uint64_t dn_hi, uint64_t hash_out[2]) {
  const u128_t h = orc_djb_hash_u(orc_u128(up_lo, up_hi), orc_u128(dn_lo, dn_hi));
  hash_out[0] = (uint64_t)h; hash_out[1] = (uint64_t)(h >> 64);
}
/* hashx = int(abs(mod(acc, int(range,ik)))): MOD takes the sign of the dividend */
int orc_get_det_owner(uint64_t up_lo, uint64_t up_hi, uint64_t dn_lo, uint64_t dn_hi, int ncores) {
  if (ncores == 1) return 0;
  const __int128 acc = (__int128)orc_djb_hash_u(orc_u128(up_lo, up_hi), orc_u128(dn_lo, dn_hi));
  __int128 m = acc % (__int128)ncores;
  if (m < 0) m = -m;
  return (int)m;
}

/* ---- all-host-cores variant of the step (the "OpenMP over walkers" CPU baseline of SURVEY section 8d(ii)) ----
 * orc_set_threads(n > 1) makes the COUNTER-discipline step run its spawn loop, its sort and its permutations on n
 * OpenMP threads.  Same algorithms, same operation order per walker, bit-identical results (the COUNTER streams are keyed
 * by walker / child / determinant, so no draw depends on which thread takes it); the linear scans of
 * merge_original_with_spawned2, reduce_my_walker and the estimator stay serial. */
static int g_orc_threads = 1;
void orc_set_threads(int n) { g_orc_threads = n > 1 ? n : 1; }
int  orc_get_threads(void) { return g_orc_threads; }
/* the same recursion with its two halves as OpenMP tasks (each half merges through its own part of tmp[0..n)) */
static void msort_idx_mt(const det_t *up, const det_t *dn, int64_t *a, int64_t *tmp, int64_t n, int depth) {
  if (n < 2) return;
  if (depth <= 0 || n < 8192) { msort_idx(up, dn, a, tmp, n); return; }
  int64_t na = (n + 1) / 2, nb = n - na;
#pragma omp task default(shared)
  msort_idx_mt(up, dn, a, tmp, na, depth - 1);
#pragma omp task default(shared)
  msort_idx_mt(up, dn, a + na, tmp + na, nb, depth - 1);
#pragma omp taskwait
  memcpy(tmp, a, na * sizeof(int64_t));
  int64_t i = 0, j = na, k = 0;
  while (i < na && j < n) {
    int64_t x = tmp[i], y = a[j];
    if (up[x] < up[y] || (up[x] == up[y] && dn[x] <= dn[y])) a[k++] = tmp[i++]; else a[k++] = a[j++];
  }
  while (i < na) a[k++] = tmp[i++];
}
void orc_merge_sort_walkers(orc_walk *w, int64_t n) {
  int64_t *ord = malloc(n * sizeof(int64_t)), *tmp = malloc((n + 2) * sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) ord[i] = i;
  if (g_orc_threads > 1) {
    int depth = 0; while ((1 << depth) < 2 * g_orc_threads) depth++;
#pragma omp parallel num_threads(g_orc_threads)
#pragma omp single
    msort_idx_mt(w->up, w->dn, ord, tmp, n, depth);
  } else msort_idx(w->up, w->dn, ord, tmp, n);
#define PERM(T, A) do { T *b = malloc(n * sizeof(T)); _Pragma("omp parallel for num_threads(g_orc_threads) if(g_orc_threads > 1)") \
                        for (int64_t i = 0; i < n; i++) b[i] = (A)[ord[i]]; \
                        memcpy((A), b, n * sizeof(T)); free(b); } while (0)
  PERM(det_t, w->up); PERM(det_t, w->dn); PERM(double, w->wt); PERM(int8_t, w->imp_distance);
  PERM(int8_t, w->initiator); PERM(double, w->matrix_elements); PERM(double, w->e_num_walker); PERM(double, w->e_den_walker);
#undef PERM
  free(ord); free(tmp);
}

static double ipow(int b, int e) { double r = 1; for (int i = 0; i < e; i++) r *= b; return r; }  /* integer ** integer */
/* do_walk.f90:6838-6872 / the inlined copies at 5952-5972 and 5989-6038 */
static int check_initiator(orc_walk *w, int64_t i, const orc_step_params *p, int *i_perm) {
  double thr = p->r_initiator * ipow(w->imp_distance[i] - p->initiator_min_distance > 0 ? w->imp_distance[i] - p->initiator_min_distance : 0, p->initiator_power);
  double aw = fabs(w->wt[i]); int d = w->imp_distance[i];
  if (w->initiator[i] == 3 && p->r_initiator >= 0) {
    int sg = w->sign_perm[(*i_perm)++];
    if (w->wt[i] * sg < 1.0) w->wt[i] = sg;
  } else if (w->initiator[i] == 2 && ((aw <= thr && d > 0) || ((aw <= p->r_initiator && !p->c_t_initiator) && d == -2))) {
    w->initiator[i] = 1;
  } else if (w->initiator[i] < 2 && ((aw > thr && d >= 0) || ((aw > p->r_initiator || p->c_t_initiator) && d == -2))) {
    w->initiator[i] = (int8_t)(w->initiator[i] + 1);
  }
  return (((w->wt[i] == 0 && (w->initiator[i] != 3 || p->r_initiator < 0)) || w->initiator[i] == 0) && w->imp_distance[i] >= 1);
}
#define DMIN(a, b) ((a) < (b) ? (a) : (b))
/* do_walk.f90:5866-6083 (chem: e_num/e_den merged too).  0-based restatement of the
 * nshift scan; returns the new nwalk. */
int64_t orc_merge_original_with_spawned2(orc_walk *w, int64_t nwalk, const orc_step_params *p) {
  if (nwalk <= 0) return nwalk;
  int64_t nshift = 0; int i_perm = 0;
  for (int64_t iw = 1; iw < nwalk; iw++) {
    int64_t t = iw - nshift - 1;                   /* previous kept slot */
    if (w->up[iw] == w->up[t] && w->dn[iw] == w->dn[t]) {
      nshift++;                                   /* now t == iw - nshift */
      if (w->wt[iw] * w->wt[t] > 0) {
        if (w->initiator[iw] > w->initiator[t]) w->initiator[t] = w->initiator[iw];
        w->e_num_walker[t] = DMIN(w->e_num_walker[t], w->e_num_walker[iw]);
        w->e_den_walker[t] = DMIN(w->e_den_walker[t], w->e_den_walker[iw]);
      }
      w->matrix_elements[t] = DMIN(w->matrix_elements[t], w->matrix_elements[iw]);
      if (w->imp_distance[t] == -2) { if (w->imp_distance[iw] == 0) w->imp_distance[t] = 0; }
      else if (w->imp_distance[iw] == -2) { if (w->imp_distance[t] != 0) w->imp_distance[t] = -2; }
      else if (w->imp_distance[t] != 0 && w->imp_distance[t] != -2) {
        int a = abs(w->imp_distance[iw]);
        if (a < w->imp_distance[t]) w->imp_distance[t] = (int8_t)a;
      }
      if (!(w->wt[iw] * w->wt[t] > 0)) {
        if (fabs(w->wt[t]) < fabs(w->wt[iw])) { if (w->initiator[t] != 3 || p->r_initiator == -1.0) w->initiator[t] = w->initiator[iw]; }
        else if (fabs(w->wt[t]) == fabs(w->wt[iw])) { if (w->initiator[t] != 3 || p->r_initiator == -1.0) w->initiator[t] = 0; }
      }
      if (!(w->imp_distance[t] == 0 && w->imp_distance[iw] == -1)) w->wt[t] = w->wt[t] + w->wt[iw];
    } else {
      if (check_initiator(w, t, p, &i_perm)) nshift++;
      int64_t u = iw - nshift;
      w->up[u] = w->up[iw]; w->dn[u] = w->dn[iw]; w->wt[u] = w->wt[iw]; w->initiator[u] = w->initiator[iw];
      w->e_num_walker[u] = w->e_num_walker[iw]; w->e_den_walker[u] = w->e_den_walker[iw];
      w->imp_distance[u] = w->imp_distance[iw]; w->matrix_elements[u] = w->matrix_elements[iw];
      if (w->imp_distance[iw] == -1) w->imp_distance[u] = 1;
    }
  }
  int64_t last = nwalk - nshift - 1;
  int discard = check_initiator(w, last, p, &i_perm);
  /* 6032-6036 runs between the flag update and the discard test of the last det */
  for (int64_t i = 0; i <= last; i++) if (w->imp_distance[i] == -1) w->imp_distance[i] = 1;
  discard = (((w->wt[last] == 0 && (w->initiator[last] != 3 || p->r_initiator < 0)) || w->initiator[last] == 0) && w->imp_distance[last] >= 1);
  if (discard) nshift++;
  int64_t nn = nwalk - nshift;
  for (int64_t i = nn; i < w->mwalk && i < nwalk; i++) { w->e_num_walker[i] = 1e51; w->e_den_walker[i] = 1e51; w->matrix_elements[i] = 1e51; }
  return nn;
}

/* Rank of a determinant among all determinants with the same electron numbers in (up, then dn) order: the colexicographic
 * rank of a bit string orders strings of equal popcount like their integer values.  Not in the reference: the key of the
 * COUNTER discipline's rounding draw (a GPU implementation has this number at hand as its sort key). */
uint64_t orc_det_rank(int norb, int ndn, det_t up, det_t dn) {
  static uint64_t bn[65][65]; static int init = 0;
  if (!init) { for (int a = 0; a < 65; a++) { bn[a][0] = 1; for (int b = 1; b <= a; b++) bn[a][b] = (b == a) ? 1 : bn[a - 1][b - 1] + bn[a - 1][b]; } init = 1; }
  uint64_t ru = 0, rd = 0; int i = 1;
  for (det_t x = up; x; x &= x - 1, i++) ru += bn[trailz(x)][i];
  i = 1;
  for (det_t x = dn; x; x &= x - 1, i++) rd += bn[trailz(x)][i];
  return ru * bn[norb][ndn] + rd;
}
/* do_walk.f90:7196-7254 (hf_to_psit = false) */
int64_t orc_reduce_my_walker(orc_walk *w, int64_t n, const orc_step_params *p) {
  for (int64_t i = 0; i < n; i++)
    if (w->imp_distance[i] >= 1 && fabs(w->wt[i]) < p->min_wt) {
      /* COUNTER discipline: the draw is keyed by the determinant itself, so that no rank among the
       * merged walkers is needed to find it (REPLAY takes the next number of the one stream) */
      orc_rng_seek(&w->rng, 2, w->key_norb ? orc_det_rank(w->key_norb, w->key_ndn, w->up[i], w->dn[i])
                                           : (uint64_t)w->up[i] * 0x9E3779B97F4A7C15ull + (uint64_t)w->dn[i]);
      if (orc_rannyu(&w->rng) < (fabs(w->wt[i]) / p->min_wt)) w->wt[i] = copysign(p->min_wt, w->wt[i]);
      else w->wt[i] = 0.0;
    }
  int64_t nshift = 0;
  for (int64_t i = 0; i < n; i++) {
    if (w->wt[i] == 0.0 && w->imp_distance[i] >= 1) nshift++;
    else if (nshift) {
      int64_t u = i - nshift;
      w->wt[u] = w->wt[i]; w->up[u] = w->up[i]; w->dn[u] = w->dn[i];
      w->e_num_walker[u] = w->e_num_walker[i]; w->e_den_walker[u] = w->e_den_walker[i];
      w->initiator[u] = w->initiator[i]; w->imp_distance[u] = w->imp_distance[i]; w->matrix_elements[u] = w->matrix_elements[i];
    }
  }
  return n - nshift;
}

/* join_walker2, do_walk.f90:6990-7103: walkers of one sign with |w| < min_wt (permanent initiators
 * excepted) are joined along the list -- the pair's weight goes to one of the two with probability
 * proportional to its own weight, one rannyu per join -- until the running weight exceeds min_wt;
 * all positive ones first, then the negative ones; zero-weight walkers are then removed. */
int64_t orc_join_walker2(orc_walk *w, int64_t n, const orc_step_params *p) {
  for (int pass = 0; pass < 2; pass++) {
    int ipair = 0; int64_t i2 = 0;
    for (int64_t i = 0; i < n; i++) {
      const double wi = w->wt[i];
      if (!((pass == 0 ? wi > 0.0 : wi < 0.0) && fabs(wi) < p->min_wt && w->initiator[i] < 3)) continue;
      if (!ipair) { ipair = 1; i2 = i; continue; }
      const double wttot = fabs(wi) + fabs(w->wt[i2]);
      orc_rng_seek(&w->rng, 2, (uint64_t)i);
      if (orc_rannyu(&w->rng) > (fabs(wi) / wttot)) {
        w->wt[i2] = copysign(wttot, w->wt[i2]); w->wt[i] = 0.0;
        w->e_num_walker[i] = 1e51; w->e_den_walker[i] = 1e51;
      } else {
        w->wt[i] = copysign(wttot, wi); w->wt[i2] = 0.0;
        w->e_num_walker[i2] = 1e51; w->e_den_walker[i2] = 1e51;
        i2 = i;
      }
      if (wttot > p->min_wt) ipair = 0;
    }
  }
  int64_t nshift = 0;
  for (int64_t i = 0; i < n; i++) {
    if (w->wt[i] == 0.0) nshift++;
    else if (nshift) {
      int64_t u = i - nshift;
      w->wt[u] = w->wt[i]; w->up[u] = w->up[i]; w->dn[u] = w->dn[i];
      w->e_num_walker[u] = w->e_num_walker[i]; w->e_den_walker[u] = w->e_den_walker[i];
      w->initiator[u] = w->initiator[i]; w->imp_distance[u] = w->imp_distance[i]; w->matrix_elements[u] = w->matrix_elements[i];
    }
  }
  return n - nshift;
}

/* system dispatch of the walk: 'chem' or 'heg' (do_walk.f90:3599-3633, 3745-3769) */
typedef struct { const orc_chem *chem; const orc_heg *heg; const orc_hub *hub; const orc_hb *hb;      /* hb: chem with proposal_method fast_heatbath */
                 int psit; det_t first_up, first_dn; } orc_sys;      /* psit: hf_to_psit = .true., the first state is dets_up/dn_psi_t(1) (sqmc_oracle_psit.c) */
static double sys_diag(const orc_sys *y, det_t u, det_t d) {
  if (y->hub) return orc_hamiltonian_hubbard(y->hub, u, d, u, d);
  return y->chem ? orc_hamiltonian(y->chem, u, d, u, d) : orc_hamiltonian_heg(y->heg, u, d, u, d);
}
static void sys_move(const orc_sys *y, orc_rng *g, double tau, det_t u, det_t d, det_t *ju, det_t *jd, double *wj, int *nd) {
  if (y->hub) orc_off_diagonal_move_hubbard(y->hub, g, tau, u, d, ju, jd, wj, nd);
  else if (y->chem) orc_off_diagonal_move_chem(y->chem, g, tau, u, d, ju, jd, wj, nd);
  else orc_off_diagonal_move_heg(y->heg, g, tau, u, d, ju, jd, wj, nd);
}

/* do_walk.f90:3538-3800, uniform2 proposal, hf_to_psit=.false.  Returns status
 * (0 ok, 1 nwalk>MWALK, 3 negative diagonal factor after equilibration). */
static int move_uniform2(const orc_sys *s, orc_walk *w, const orc_step_params *p, int64_t iw, int64_t *attempts) {
  int spawn, use_wt;
  if (s->psit && w->up[iw] == s->first_up && w->dn[iw] == s->first_dn) return 0;     /* 3574: for the first state all moves are deterministic */
  if (fabs(w->wt[iw]) < p->always_spawn_cutoff_wt) {
    /* COUNTER discipline: the gate's draw is keyed by the determinant's rank in (up, dn) order, like the rounding draw */
    orc_rng_seek(&w->rng, 0, w->key_norb ? orc_det_rank(w->key_norb, w->key_ndn, w->up[iw], w->dn[iw]) : (uint64_t)iw);
    spawn = (orc_rannyu(&w->rng) < fabs(w->wt[iw] / p->always_spawn_cutoff_wt)); use_wt = 0; w->n_spawn_draws++;
  } else { spawn = 1; use_wt = 1; }
  if (spawn) {
    long nchild; double wchild;
    if (use_wt) { nchild = lround(fabs(w->wt[iw])); if (nchild < 1) nchild = 1; wchild = w->wt[iw] / nchild; }
    else { nchild = 1; wchild = copysign(p->always_spawn_cutoff_wt, w->wt[iw]); }
    for (long c = 1; c <= nchild; c++) {
      det_t ju, jd; double wj; int nd;
      orc_rng_seek(&w->rng, 1, (uint64_t)(*attempts));
      det_t ju2[2], jd2[2]; double wj2[2] = {0.0, 0.0}; int lev2[2], nnew = 1;
      if (s->hb) {          /* fast_heatbath: up to two determinants per child, each through add_walker (do_walk.f90:3604-3611, 7584-7697) */
        orc_off_diagonal_move_chem_heatbath(s->chem, s->hb, &w->rng, p->tau, w->up[iw], w->dn[iw], ju2, jd2, wj2, lev2, &nd);
        nnew = 2;
      } else { sys_move(s, &w->rng, p->tau, w->up[iw], w->dn[iw], &ju2[0], &jd2[0], &wj2[0], &nd); }
      w->n_spawn_draws += nd; (*attempts)++;
      for (int kk = 0; kk < nnew; kk++) {
      ju = ju2[kk]; jd = jd2[kk];
      wj = wchild * wj2[kk];
      if (s->psit && ju == s->first_up && jd == s->first_dn) wj = 0;      /* 3676 / 7642: no stochastic spawning onto the first state */
      if (wj != 0) {
        int64_t k = w->nwalk++;
        if (w->nwalk > w->mwalk) return 1;
        w->up[k] = ju; w->dn[k] = jd; w->wt[k] = wj;
        if (w->imp_distance[iw] == -2) w->imp_distance[k] = p->c_t_initiator ? 1 : 2;
        else w->imp_distance[k] = (int8_t)((w->imp_distance[iw] < 126 ? w->imp_distance[iw] : 126) + 1);
        if (p->semistochastic && w->imp_distance[iw] == 0) w->imp_distance[k] = -1;
        w->initiator[k] = (w->initiator[iw] >= 2) ? 1 : 0;
        if (p->c_t_initiator && w->imp_distance[iw] == -2) w->initiator[k] = 1;
        if (p->semistochastic && w->imp_distance[iw] == 0) w->initiator[k] = 1;
        w->e_num_walker[k] = 1e51; w->e_den_walker[k] = 1e51; w->matrix_elements[k] = 1e51;
      }
      }
    }
  }
  if (!p->semistochastic || w->imp_distance[iw] >= 1) {
    double hii;
    if (w->matrix_elements[iw] > 1e50) {
      hii = sys_diag(s, w->up[iw], w->dn[iw]);
      w->matrix_elements[iw] = hii;
    } else hii = w->matrix_elements[iw];
    double f = 1.0 + p->tau * (p->e_trial - hii);
    if (f < 0) { if (p->reached_w_abs_gen > 1) return 3; f = 0; }
    w->wt[iw] = w->wt[iw] * f;
  }
  return 0;
}

/* move_uniform2 for all walkers on g_orc_threads threads (COUNTER discipline only).  Pass 1: gate and child count of every
 * walker (stream keyed by the walker index); serial prefix sum = the running `attempts` counter of the serial loop; pass 2:
 * every child from its own stream (keyed by its attempt number) into scratch, death/clone of the parent; pass 3 (serial):
 * the children that made a walker are appended in attempt order, exactly where the serial loop puts them. */
static int move_uniform2_mt(const orc_sys *s, orc_walk *w, const orc_step_params *p, int64_t n0, int64_t *attempts) {
  int64_t *nchild = malloc((n0 + 1) * sizeof(int64_t)), *off = malloc((n0 + 1) * sizeof(int64_t));
  double *wchild = malloc((n0 + 1) * sizeof(double));
  int64_t gate_draws = 0;
#pragma omp parallel for num_threads(g_orc_threads) schedule(static) reduction(+:gate_draws)
  for (int64_t iw = 0; iw < n0; iw++) {
    int spawn, use_wt;
    if (fabs(w->wt[iw]) < p->always_spawn_cutoff_wt) {
      orc_rng g = w->rng; orc_rng_seek(&g, 0, w->key_norb ? orc_det_rank(w->key_norb, w->key_ndn, w->up[iw], w->dn[iw]) : (uint64_t)iw);
      spawn = (orc_rannyu(&g) < fabs(w->wt[iw] / p->always_spawn_cutoff_wt)); use_wt = 0; gate_draws++;
    } else { spawn = 1; use_wt = 1; }
    long nc = 0; double wc = 0.0;
    if (spawn) {
      if (use_wt) { nc = lround(fabs(w->wt[iw])); if (nc < 1) nc = 1; wc = w->wt[iw] / nc; }
      else { nc = 1; wc = copysign(p->always_spawn_cutoff_wt, w->wt[iw]); }
    }
    nchild[iw] = nc; wchild[iw] = wc;
  }
  int64_t tot = 0;
  for (int64_t iw = 0; iw < n0; iw++) { off[iw] = tot; tot += nchild[iw]; }
  const int spc = s->hb ? 2 : 1;                    /* slots per child */
  det_t *cu = malloc((spc * tot + 2) * sizeof(det_t)), *cd = malloc((spc * tot + 2) * sizeof(det_t)); double *cw = malloc((spc * tot + 2) * sizeof(double));
  int64_t draws = 0; int bad = 0;
#pragma omp parallel for num_threads(g_orc_threads) schedule(dynamic, 256) reduction(+:draws) reduction(|:bad)
  for (int64_t iw = 0; iw < n0; iw++) {
    orc_rng g = w->rng;
    for (int64_t c = 0; c < nchild[iw]; c++) {
      det_t ju, jd; double wj; int nd;
      orc_rng_seek(&g, 1, (uint64_t)(off[iw] + c));
      const int64_t slot = spc * (off[iw] + c);
      if (s->hb) {
        det_t ju2[2], jd2[2]; double wj2[2]; int lev2[2];
        orc_off_diagonal_move_chem_heatbath(s->chem, s->hb, &g, p->tau, w->up[iw], w->dn[iw], ju2, jd2, wj2, lev2, &nd);
        for (int kk = 0; kk < 2; kk++) { cu[slot + kk] = ju2[kk]; cd[slot + kk] = jd2[kk]; cw[slot + kk] = wchild[iw] * wj2[kk]; }
      } else {
        sys_move(s, &g, p->tau, w->up[iw], w->dn[iw], &ju, &jd, &wj, &nd);
        cu[slot] = ju; cd[slot] = jd; cw[slot] = wchild[iw] * wj;
      }
      draws += nd;
    }
    if (!p->semistochastic || w->imp_distance[iw] >= 1) {
      double hii;
      if (w->matrix_elements[iw] > 1e50) { hii = sys_diag(s, w->up[iw], w->dn[iw]); w->matrix_elements[iw] = hii; }
      else hii = w->matrix_elements[iw];
      double f = 1.0 + p->tau * (p->e_trial - hii);
      if (f < 0) { if (p->reached_w_abs_gen > 1) bad |= 1; f = 0; }
      w->wt[iw] = w->wt[iw] * f;
    }
  }
  int st = 0;
  for (int64_t iw = 0; iw < n0 && !st; iw++) {
    for (int64_t c = 0; c < spc * nchild[iw]; c++) {
      const double wj = cw[spc * off[iw] + c];
      if (wj != 0) {
        int64_t k = w->nwalk++;
        if (w->nwalk > w->mwalk) { st = 1; break; }
        w->up[k] = cu[spc * off[iw] + c]; w->dn[k] = cd[spc * off[iw] + c]; w->wt[k] = wj;
        if (w->imp_distance[iw] == -2) w->imp_distance[k] = p->c_t_initiator ? 1 : 2;
        else w->imp_distance[k] = (int8_t)((w->imp_distance[iw] < 126 ? w->imp_distance[iw] : 126) + 1);
        if (p->semistochastic && w->imp_distance[iw] == 0) w->imp_distance[k] = -1;
        w->initiator[k] = (w->initiator[iw] >= 2) ? 1 : 0;
        if (p->c_t_initiator && w->imp_distance[iw] == -2) w->initiator[k] = 1;
        if (p->semistochastic && w->imp_distance[iw] == 0) w->initiator[k] = 1;
        w->e_num_walker[k] = 1e51; w->e_den_walker[k] = 1e51; w->matrix_elements[k] = 1e51;
      }
    }
  }
  if (!st && bad) st = 3;
  w->n_spawn_draws += gate_draws + draws; *attempts = tot;
  free(nchild); free(off); free(wchild); free(cu); free(cd); free(cw);
  return st;
}

/* more_tools.f90:4041-4098 : walkers scanned from the last to the first; lookups only for
 * walkers whose e_num is still the 1e51 sentinel.  The shrinking upper bound of the
 * reference is an optimisation of the same search (both lists sorted). */
static void search_list_and_update(orc_walk *w, int64_t n, double acc[7]) {
  for (int64_t i = n - 1; i >= 0; i--) {
    if (w->e_num_walker[i] > 1e50) {
      int64_t j = bsearch_det(w->ct_up, w->ct_dn, w->n_ct, w->up[i], w->dn[i]);
      if (j < 0) { w->e_num_walker[i] = 0; w->e_den_walker[i] = 0; }
      else { w->e_num_walker[i] = w->ct_num[j]; w->e_den_walker[i] = w->ct_den[j]; }
    }
    double en = w->e_num_walker[i] * w->wt[i], ed = w->e_den_walker[i] * w->wt[i];
    if (en != 0.0) {
      if (fabs(ed) < 1e-22) ed = fabs(ed);
      acc[1] += ed; acc[3] += ed * ed; acc[5] += fabs(ed);
      acc[0] += en; acc[2] += en * en; acc[4] += en * copysign(1.0, ed); acc[6] += en * ed;
    }
  }
}

/* One MC step, do_walk.f90:2186-2790 for semistochastic chem, ncores=1, hf_to_psit=.false.,
 * run_type 'none'.  Population control (2880-2901) stays with the caller. */
static int walk_step_sys(const orc_sys *s, orc_walk *w, const orc_step_params *p, double out[16]);
int orc_walk_step(const orc_chem *c, orc_walk *w, const orc_step_params *p, double out[16]) {
  orc_sys y = {c, NULL, NULL, NULL, 0, 0, 0};
  return walk_step_sys(&y, w, p, out);
}
int orc_walk_step_heatbath(const orc_chem *c, const orc_hb *hb, orc_walk *w, const orc_step_params *p, double out[16]) {
  orc_sys y = {c, NULL, NULL, hb, 0, 0, 0};
  return walk_step_sys(&y, w, p, out);
}
int orc_walk_step_hubbard(const orc_hub *h, orc_walk *w, const orc_step_params *p, double out[16]) {
  orc_sys y = {NULL, NULL, h, NULL, 0, 0, 0};
  return walk_step_sys(&y, w, p, out);
}
int orc_walk_step_heg(const orc_heg *h, orc_walk *w, const orc_step_params *p, double out[16]) {
  orc_sys y = {NULL, h, NULL, NULL, 0, 0, 0};
  return walk_step_sys(&y, w, p, out);
}
static double orc_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double orc_prof[8];            /* seconds spent in the stages of the step since the caller last zeroed it: move, project, sort, merge, reduce, estimate */
static int walk_step_sys(const orc_sys *s, orc_walk *w, const orc_step_params *p, double out[16]) {
  int64_t n0 = w->nwalk, nimp = 0, attempts = 0;
  double t0 = orc_now(), t1;
#define LAP(K) do { t1 = orc_now(); orc_prof[K] += t1 - t0; t0 = t1; } while (0)
  int64_t *loc = NULL; double *impw = NULL, *dw = NULL;
  w->n_spawn_draws = 0;
  if (p->semistochastic) {                         /* 2188-2212 */
    loc = malloc((w->n_imp + 1) * sizeof(int64_t)); impw = malloc((w->n_imp + 1) * sizeof(double)); dw = malloc((w->n_imp + 1) * sizeof(double));
    for (int64_t i = 0; i < n0 && nimp < w->n_imp; i++) if (w->imp_distance[i] == 0) { loc[nimp] = i; impw[nimp] = w->wt[i]; nimp++; }
    if (nimp != w->n_imp) { free(loc); free(impw); free(dw); return 5; }
  }
  if (g_orc_threads > 1 && w->rng.mode == 1) {     /* all-cores variant: same gate, children and death/clone per walker, threads over walkers */
    int st = move_uniform2_mt(s, w, p, n0, &attempts);
    if (st) { free(loc); free(impw); free(dw); return st; }
  } else
  for (int64_t i = 0; i < n0; i++) {               /* 2220-2231 */
    int st = move_uniform2(s, w, p, i, &attempts);
    if (st) { free(loc); free(impw); free(dw); return st; }
  }
  LAP(0);
  if (p->semistochastic) {                         /* 2255-2325 */
    double *x = malloc((nimp + 1) * sizeof(double));
    for (int64_t i = 0; i < nimp; i++) x[i] = w->wt[loc[i]];
    orc_spmv_sym_upper(nimp, w->prj_counts, w->prj_indices, w->prj_values, x, dw);
    for (int64_t i = 0; i < nimp; i++) dw[i] = dw[i] + p->e_trial * p->tau * impw[i];
    for (int64_t i = 0; i < nimp; i++) w->wt[loc[i]] = w->wt[loc[i]] + dw[i];
    free(x);
  }
  free(loc); free(impw); free(dw);
  int64_t n = w->nwalk;
  LAP(1);
  orc_merge_sort_walkers(w, n);                    /* 2335 */
  LAP(2);
  double wabs_before = 0; for (int64_t i = 0; i < n; i++) wabs_before += fabs(w->wt[i]);
  int64_t nbefore = n;
  n = orc_merge_original_with_spawned2(w, n, p);   /* 2373 */
  LAP(3);
  if (p->semistochastic) n = orc_reduce_my_walker(w, n, p);   /* 2473 */
  else n = orc_join_walker2(w, n, p);                          /* 2475 */
  w->nwalk = n;
  LAP(4);
  for (int64_t i = 0; i < n; i++) w->wt[i] = w->wt[i] * p->reweight_factor_inv;   /* 2487 */
  if (n == 0) return 4;
  double w_gen = 0, w2 = 0, w_abs = 0, w_abs_imp = 0, w_perm = 0; int ip = 0;      /* 2573-2598 */
  for (int64_t i = 0; i < n; i++) {
    if (w->initiator[i] == 3) w_perm += w->wt[i] * w->sign_perm[ip++];
    w_gen += w->wt[i]; w2 += w->wt[i] * w->wt[i]; w_abs += fabs(w->wt[i]);
    if (w->imp_distance[i] == 0 || (w->imp_distance[i] == -2 && p->c_t_initiator)) w_abs_imp += fabs(w->wt[i]);
  }
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  search_list_and_update(w, n, acc);               /* 2755-2759 */
  out[0] = w_gen; out[1] = w_abs; out[2] = acc[1]; out[3] = acc[0]; out[4] = w_perm; out[5] = (double)n;
  out[6] = w_abs_imp; out[7] = (double)nbefore; out[8] = w2; out[9] = acc[2]; out[10] = acc[3];
  out[11] = acc[4]; out[12] = acc[5]; out[13] = acc[6]; out[14] = wabs_before; out[15] = (double)attempts;
  w->rng.step++;
  LAP(5);
#undef LAP
  return 0;
}


/* ==================================================================== HEG */
static double sum_sq3(const double *v, int n) { double s = 0; for (int j = 0; j < n; j++) s = s + v[j] * v[j]; return s; }
/* generate_k_vectors, heg.f90:643-749; ordering = shell_sort_real_rank2, generic_sort.f90:554-591 */
orc_heg *orc_heg_new(int n_dim, double r_s, int nelec, int nup, double cutoff_radius) {
  const double EPS = 1.0e-15, pi = 4.0 * atan(1.0);
  if (n_dim != 2 && n_dim != 3) return NULL;
  orc_heg *h = calloc(1, sizeof(orc_heg));
  h->n_dim = n_dim; h->r_s = r_s; h->nelec = nelec; h->nup = nup; h->ndn = nelec - nup;
  double density = (n_dim == 2) ? 1.0 / (pi * (r_s * r_s)) : 3.0 / (4.0 * pi * (r_s * r_s * r_s));
  h->length_cell = pow(nelec / density, 1.0 / n_dim);
  int n_max = (int)(cutoff_radius + EPS); h->n_max = n_max;
  int side = 2 * n_max + 1, ntot = 1; for (int j = 0; j < n_dim; j++) ntot *= side;
  double *kv = malloc((size_t)ntot * 3 * sizeof(double)), *values = malloc(side * sizeof(double));
  for (int i = -n_max; i <= n_max; i++) values[i + n_max] = 2 * pi / h->length_cell * i;
  int idx = 0;
  if (n_dim == 3) { for (int i = 0; i < side; i++) for (int j = 0; j < side; j++) for (int k = 0; k < side; k++) { kv[3 * idx] = values[i]; kv[3 * idx + 1] = values[j]; kv[3 * idx + 2] = values[k]; idx++; } }
  else { for (int i = 0; i < side; i++) for (int j = 0; j < side; j++) { kv[3 * idx] = values[i]; kv[3 * idx + 1] = values[j]; kv[3 * idx + 2] = 0.0; idx++; } }
  /* shell sort on |k|^2, increments n/2, *5/11 (2 -> 1) */
  for (int inc = ntot / 2; inc > 0; inc = (inc == 2) ? 1 : inc * 5 / 11) {
    for (int i = inc; i < ntot; i++) {
      int j = i; double t[3] = {kv[3 * i], kv[3 * i + 1], kv[3 * i + 2]};
      while (j >= inc) {
        if (sum_sq3(kv + 3 * (j - inc), n_dim) <= sum_sq3(t, n_dim)) break;
        memcpy(kv + 3 * j, kv + 3 * (j - inc), 3 * sizeof(double));
        j -= inc;
      }
      kv[3 * j] = t[0]; kv[3 * j + 1] = t[1]; kv[3 * j + 2] = t[2];
    }
  }
  int norb = 0;
  for (int i = 0; i < ntot; i++) { if (sqrt(sum_sq3(kv + 3 * i, n_dim)) > 2 * pi / h->length_cell * cutoff_radius + EPS) break; norb++; }
  if (norb > ORC_MAXORB) { free(kv); free(values); free(h); return NULL; }
  h->norb = norb;
  for (int i = 1; i <= norb; i++) for (int j = 0; j < 3; j++) {
    h->k[i][j] = kv[3 * (i - 1) + j];
    h->krel[i][j] = (int)lround(h->k[i][j] * h->length_cell / (2 * pi));
  }
  free(kv); free(values);
  return h;
}
void orc_heg_free(orc_heg *h) { free(h); }

static double inv_k2(const orc_heg *h, int p, int q) {   /* FOUR_PI / sum((k_p - k_q)**2) */
  const double FOUR_PI = 4.0 * (4.0 * atan(1.0));
  double s = 0; for (int j = 0; j < h->n_dim; j++) { double d = h->k[p][j] - h->k[q][j]; s = s + d * d; }
  return FOUR_PI / s;
}
/* get_gamma_exp, heg.f90:811-842: for each listed orbital present in det, the number of occupied orbitals below it */
static int gamma_exp(det_t det, det_t eor) {
  int g = 0;
  for (det_t e = eor & det; e; e &= e - 1) { int o = trailz(e); g += popcnt(det & maskr(o)); }
  return g;
}
/* heg.f90:845-1011 */
double orc_hamiltonian_heg(const orc_heg *h, det_t iu, det_t id, det_t ju, det_t jd) {
  const double EPS = 1.0e-15, L = h->length_cell; const int nd = h->n_dim;
  double me = 0.0;
  if (iu == ju && id == jd) {
    for (det_t d = iu; d; d &= d - 1) { int p = trailz(d) + 1; me = me + sum_sq3(h->k[p], nd) * 0.5; }
    for (det_t d = id; d; d &= d - 1) { int p = trailz(d) + 1; me = me + sum_sq3(h->k[p], nd) * 0.5; }
    double pot = 0.0;
    for (det_t a = iu; a; a &= a - 1) for (det_t b = a & (a - 1); b; b &= b - 1) pot = pot + inv_k2(h, trailz(a) + 1, trailz(b) + 1);
    for (det_t a = id; a; a &= a - 1) for (det_t b = a & (a - 1); b; b &= b - 1) pot = pot + inv_k2(h, trailz(a) + 1, trailz(b) + 1);
    return me - pot / (L * L * L);
  }
  det_t eu = iu ^ ju, ed = id ^ jd;
  int neu = popcnt(eu), ned = popcnt(ed);
  if (neu + ned != 4) return 0.0;
  double mc[3] = {0, 0, 0}; int op = 0, oq = 0, os = 0;
  for (int sp = 0; sp < 2; sp++) {
    det_t e = sp ? ed : eu, di = sp ? id : iu;
    for (; e; e &= e - 1) {
      int o = trailz(e) + 1;
      if (btest(di, o - 1)) { for (int j = 0; j < nd; j++) mc[j] = mc[j] - h->k[o][j]; if (!op) op = o; }
      else { for (int j = 0; j < nd; j++) mc[j] = mc[j] + h->k[o][j]; if (!oq) oq = o; else if (!os) os = o; }
    }
  }
  if (sum_sq3(mc, nd) * (L * L) > EPS) return 0.0;
  double pot = inv_k2(h, op, oq);
  if (neu != 2) pot = pot - inv_k2(h, op, os);
  int g = gamma_exp(iu, eu) + gamma_exp(ju, eu) + gamma_exp(id, ed) + gamma_exp(jd, ed);
  if (g & 1) pot = -pot;
  return pot / (L * L * L);
}

/* heg.f90:1344-1598 */
void orc_off_diagonal_move_heg(const orc_heg *h, orc_rng *g, double tau, det_t iu, det_t id,
                               det_t *pju, det_t *pjd, double *weight_j, int *n_draws) {
  const double EPS = 1.0e-15;
  const int nelec = h->nelec, nup = h->nup, ndn = h->ndn, norb = h->norb, nd = h->n_dim;
  int draws = 0;
#define RI(n) (draws++, orc_random_int(g, (n)))
  det_t ju = iu, jd = id;
  int e1 = RI(nelec), e2;
  do { e2 = RI(nelec); } while (e1 == e2);
  int spin = ((e1 > nup) ? -1 : 1) + ((e2 > nup) ? -1 : 1);
  double from[3] = {0, 0, 0}, to1[3] = {0, 0, 0};
  int ie = 0;
  for (int i = 1; i <= norb; i++) if (btest(iu, i - 1)) { ie++; if (ie == e1 || ie == e2) { for (int j = 0; j < nd; j++) from[j] = from[j] + h->k[i][j]; ju &= ~BIT(i - 1); } }
  for (int i = 1; i <= norb; i++) if (btest(id, i - 1)) { ie++; if (ie == e1 || ie == e2) { for (int j = 0; j < nd; j++) from[j] = from[j] + h->k[i][j]; jd &= ~BIT(i - 1); } }
  int found = 0; double prob = 1.0;
  /* first hole in channel A (k-th empty orbital of det_i), partner hole in channel B by momentum conservation */
  int first_up, second_up, to1n; float denom;
  if (spin == 2) { to1n = RI(norb - nup); first_up = 1; second_up = 1; denom = (float)(nelec * (nelec - 1) * (norb - nup)); }
  else if (spin == -2) { to1n = RI(norb - ndn); first_up = 0; second_up = 0; denom = (float)(nelec * (nelec - 1) * (norb - ndn)); }
  else {
    to1n = RI(2 * norb - nup - ndn); denom = (float)(nelec * (nelec - 1) * (2 * norb - nelec));
    if (to1n <= norb - nup) { first_up = 1; second_up = 0; } else { to1n -= (norb - nup); first_up = 0; second_up = 1; }
  }
  { det_t di = first_up ? iu : id; int io = 0;
    for (int i = 1; i <= norb; i++) if (!btest(di, i - 1)) { io++; if (io == to1n) { for (int j = 0; j < nd; j++) to1[j] = to1[j] + h->k[i][j]; if (first_up) ju |= BIT(i - 1); else jd |= BIT(i - 1); } } }
  { det_t busy = second_up ? (iu | ju) : (id | jd);
    for (int i = 1; i <= norb; i++) if (!btest(busy, i - 1)) {
      int ok = 0;
      for (int j = 0; j < nd; j++) if (fabs(from[j] - (to1[j] + h->k[i][j])) < EPS) ok++;
      if (ok == nd) { if (second_up) ju |= BIT(i - 1); else jd |= BIT(i - 1); prob = (double)(4.0f / denom); found = 1; break; }
    } }
  *weight_j = 0; *pju = ju; *pjd = jd;
  if (found) {
    double me = orc_hamiltonian_heg(h, iu, id, ju, jd);
    double acc = tau * fabs(me) / prob;
    *weight_j = acc * copysign(1.0, -me);
  }
#undef RI
  if (n_draws) *n_draws = draws;
}

int orc_connected_heg(const orc_heg *h, det_t up, det_t dn, det_t *cu, det_t *cd, double *el, int cap) {
  int n = 0, norb = h->norb;
#define PUSH(U, D) do { if (n < cap) { cu[n] = (U); cd[n] = (D); if (el) el[n] = orc_hamiltonian_heg(h, up, dn, (U), (D)); } n++; } while (0)
  PUSH(up, dn);
  /* all pairs of electrons (p,q) and holes (r,s) with k_p + k_q = k_r + k_s (integer k) */
  int occ[2 * ORC_MAXORB], nocc = 0;
  for (det_t d = up; d; d &= d - 1) occ[nocc++] = trailz(d) + 1;
  int nu = nocc;
  for (det_t d = dn; d; d &= d - 1) occ[nocc++] = trailz(d) + 1 + norb;
  for (int a = 0; a < nocc; a++) for (int b = a + 1; b < nocc; b++) {
    int p = occ[a], q = occ[b], pu = p <= norb, qu = q <= norb;
    int ps = pu ? p : p - norb, qs = qu ? q : q - norb;
    for (int r = 1; r <= norb; r++) for (int s_ = 1; s_ <= norb; s_++) {
      /* r takes p's spin, s takes q's spin; same-spin pairs need r < s to avoid double counting */
      if (pu == qu && s_ <= r) continue;
      det_t ru = pu ? up : dn, su = qu ? up : dn;
      if (btest(ru, r - 1) || btest(su, s_ - 1)) continue;
      int ok = 1;
      for (int j = 0; j < 3; j++) if (h->krel[ps][j] + h->krel[qs][j] != h->krel[r][j] + h->krel[s_][j]) ok = 0;
      if (!ok) continue;
      det_t nu_ = up, nd_ = dn;
      if (pu) nu_ &= ~BIT(ps - 1); else nd_ &= ~BIT(ps - 1);
      if (qu) nu_ &= ~BIT(qs - 1); else nd_ &= ~BIT(qs - 1);
      if (pu) nu_ |= BIT(r - 1); else nd_ |= BIT(r - 1);
      if (qu) nu_ |= BIT(s_ - 1); else nd_ |= BIT(s_ - 1);
      PUSH(nu_, nd_);
    }
  }
  (void)nu;
#undef PUSH
  return n;
}

int64_t orc_build_sparse_ham_heg(const orc_heg *h, int64_t n, const det_t *up, const det_t *dn,
                                 int64_t **row_counts, int64_t **indices, double **values) {
  int64_t cap = n * 64 + 1024, nnz = 0;
  int64_t *rc = calloc(n, sizeof(int64_t)), *idx = malloc(cap * sizeof(int64_t)); double *val = malloc(cap * sizeof(double));
  for (int64_t i = 0; i < n; i++) {
    if (nnz + i + 2 > cap) { cap = 2 * cap + i; idx = realloc(idx, cap * sizeof(int64_t)); val = realloc(val, cap * sizeof(double)); }
    idx[nnz] = i + 1; val[nnz] = orc_hamiltonian_heg(h, up[i], dn[i], up[i], dn[i]); nnz++; rc[i] = 1;
    for (int64_t j = 0; j < i; j++) {
      if (popcnt(up[i] ^ up[j]) + popcnt(dn[i] ^ dn[j]) != 4) continue;
      double v = orc_hamiltonian_heg(h, up[i], dn[i], up[j], dn[j]);
      if (v == 0.0) continue;
      idx[nnz] = j + 1; val[nnz] = v; nnz++; rc[i]++;
    }
  }
  *row_counts = rc; *indices = idx; *values = val;
  return nnz;
}


/* ==================================================================== real-space Hubbard */
orc_hub *orc_hub_new(int l_x, int l_y, int pbc, int nup, int ndn, double t, double U) {
  if (l_x < 1 || l_y < 1 || l_x * l_y > ORC_MAXORB || nup < 0 || ndn < 0 || nup > l_x * l_y || ndn > l_x * l_y || nup + ndn < 1) return NULL;
  orc_hub *h = calloc(1, sizeof(orc_hub));
  h->l_x = l_x; h->l_y = l_y; h->pbc = pbc; h->nsites = l_x * l_y; h->nup = nup; h->ndn = ndn; h->t = t; h->U = U;
  return h;
}
void orc_hub_free(orc_hub *h) { free(h); }

/* more_tools.f90:223-355 */
int orc_get_nbr(int l_x, int l_y, int pbc, int site, int nbr_type) {
  int y_1 = ((site - 1) / l_x) + 1, x_1 = site - ((y_1 - 1) * l_x);
  int allowed = 1, y_2 = y_1, x_2 = x_1;
  if (nbr_type == 1) {                                   /* LEFT */
    x_2 = x_1 - 1; y_2 = y_1;
    if (!pbc) allowed = (x_2 > 0);
    else { if (x_2 == 0) x_2 = l_x; if (x_2 == x_1) allowed = 0; }
  }
  if (nbr_type == 2) {                                   /* RIGHT */
    x_2 = x_1 + 1; y_2 = y_1;
    if (!pbc) allowed = (x_2 <= l_x);
    else { if (x_2 == l_x + 1) x_2 = 1; if (x_2 == x_1) allowed = 0; }
  }
  if (nbr_type == 3) {                                   /* UP */
    x_2 = x_1; y_2 = y_1 + 1;
    if (!pbc) allowed = (y_2 <= l_y);
    else { if (y_2 == l_y + 1) y_2 = 1; if (y_2 == y_1) allowed = 0; }
  }
  if (nbr_type == 4) {                                   /* DOWN */
    x_2 = x_1; y_2 = y_1 - 1;
    if (!pbc) allowed = (y_2 > 0);
    else { if (y_2 == 0) y_2 = l_y; if (y_2 == y_1) allowed = 0; }
  }
  return allowed ? ((y_2 - 1) * l_x) + x_2 : -1;
}

/* more_tools.f90:140-176: parity of the occupied sites strictly between the two positions */
int orc_fermionic_phase(det_t config, int site_1, int site_2) {
  int lo = site_1 < site_2 ? site_1 : site_2, hi = site_1 < site_2 ? site_2 : site_1, num_betn = 0;
  for (int i = lo + 1; i <= hi - 1; i++) if (btest(config, i - 1)) num_betn++;
  return (num_betn % 2 == 0) ? 1 : -1;
}

/* hubbard.f90:1536-1644 */
double orc_hamiltonian_hubbard(const orc_hub *h, det_t iu, det_t id, det_t ju, det_t jd) {
  int ns = h->nsites;
  if (iu == ju && id == jd) {
    det_t both = iu & id; int docc = 0;
    for (int i = 1; i <= ns; i++) if (btest(both, i - 1)) docc++;
    return h->U * docc;
  }
  det_t c1 = 0, c2 = 0, cfg = 0; int flag = 0;
  if (iu == ju) { c1 = id & ~jd; c2 = ~id & jd; cfg = id; flag = 1; }
  if (id == jd) { c1 = iu & ~ju; c2 = ~iu & ju; cfg = iu; flag = 1; }
  if (!flag) return 0.0;
  int pos_1 = 0, pos_2 = 0, num_ones = 0;
  for (int i = 1; i <= ns; i++) if (btest(c1, i - 1)) { num_ones++; if (num_ones == 1) pos_1 = i; else return 0.0; }
  if (num_ones == 0) return 0.0;
  num_ones = 0;
  for (int i = 1; i <= ns; i++) if (btest(c2, i - 1)) { num_ones++; if (num_ones == 1) pos_2 = i; else return 0.0; }
  if (num_ones == 0) return 0.0;
  return -h->t * orc_fermionic_phase(cfg, pos_1 < pos_2 ? pos_1 : pos_2, pos_1 < pos_2 ? pos_2 : pos_1);
}

double orc_hamiltonian_hubbard_checked(const orc_hub *h, det_t iu, det_t id, det_t ju, det_t jd) {
  if (iu == ju && id == jd) return orc_hamiltonian_hubbard(h, iu, id, ju, jd);
  det_t a, b;
  if (iu == ju) { a = id & ~jd; b = jd & ~id; } else if (id == jd) { a = iu & ~ju; b = ju & ~iu; } else return 0.0;
  if (popcnt(a) != 1 || popcnt(b) != 1) return 0.0;
  int p1 = trailz(a) + 1, p2 = trailz(b) + 1, bond = 0;
  for (int k = 1; k <= 4; k++) if (orc_get_nbr(h->l_x, h->l_y, h->pbc, p1, k) == p2) bond = 1;
  return bond ? orc_hamiltonian_hubbard(h, iu, id, ju, jd) : 0.0;
}

/* hubbard.f90:2992-3120 with choose_random_electron 1024-1058 (vmc off) */
void orc_off_diagonal_move_hubbard(const orc_hub *h, orc_rng *g, double tau, det_t iu, det_t id,
                                   det_t *ju, det_t *jd, double *weight_j, int *n_draws) {
  int draws = 0;
  *weight_j = 0; *ju = iu; *jd = id;
  int chosen_site, chosen_spin;
  for (;;) {
    chosen_site = orc_random_int(g, 2 * h->l_x * h->l_y); draws++;
    if (chosen_site % 2 == 0) { chosen_spin = 0; chosen_site = chosen_site / 2; if (btest(id, chosen_site - 1)) break; }
    else { chosen_spin = 1; chosen_site = (chosen_site + 1) / 2; if (btest(iu, chosen_site - 1)) break; }
  }
  det_t det = chosen_spin ? iu : id;
  int ctr = 0, allowed_nbrs[4];
  for (int k = 1; k <= 4; k++) {
    int nbr = orc_get_nbr(h->l_x, h->l_y, h->pbc, chosen_site, k);
    if (nbr > 0 && !btest(det, nbr - 1)) allowed_nbrs[ctr++] = nbr;
  }
  if (ctr == 0) { if (n_draws) *n_draws = draws; return; }
  int target_k = orc_random_int(g, ctr); draws++;
  int target_site = allowed_nbrs[target_k - 1];
  det = (det | BIT(target_site - 1)) & ~BIT(chosen_site - 1);
  if (chosen_spin == 0) *jd = det; else *ju = det;
  double me = orc_hamiltonian_hubbard(h, iu, id, *ju, *jd);
  double proposal_prob_inv = (double)((h->nup + h->ndn) * ctr);
  double acceptance_prob = fabs(me) * tau * proposal_prob_inv;
  *weight_j = acceptance_prob * copysign(1.0, -me);
  if (n_draws) *n_draws = draws;
}

/* hubbard.f90:5306-5457 */
int orc_connected_hubbard(const orc_hub *h, det_t up, det_t dn, det_t *cu, det_t *cd, double *el, int cap) {
  int n = 0, num_e = 0, docc = 0, nel = h->nup + h->ndn;
  for (int i_ctr = 1; i_ctr <= 2 * h->nsites; i_ctr++) {
    if (num_e == nel) break;
    int i = (i_ctr + 1) / 2, is_up = (i_ctr % 2 != 0);
    det_t config = is_up ? up : dn, other = is_up ? dn : up;
    if (!btest(config, i - 1)) continue;
    num_e++;
    if (is_up && btest(other, i - 1)) docc++;
    for (int j = 1; j <= 4; j++) {
      int nbr = orc_get_nbr(h->l_x, h->l_y, h->pbc, i, j);
      if (nbr > 0 && !btest(config, nbr - 1)) {
        det_t nc = (config & ~BIT(i - 1)) | BIT(nbr - 1);
        if (n < cap) { cu[n] = is_up ? nc : up; cd[n] = is_up ? dn : nc; if (el) el[n] = -h->t * (double)orc_fermionic_phase(config, i, nbr); }
        n++;
      }
    }
  }
  /* the diagonal entry U*doubly_occupied: doubly occupied sites are only counted while electrons
   * are being enumerated (the loop exits once all are found), which covers every up electron */
  if (n < cap) { cu[n] = up; cd[n] = dn; if (el) el[n] = h->U * docc; }
  n++;
  return n;
}

int64_t orc_build_sparse_ham_hubbard(const orc_hub *h, int64_t n, const det_t *up, const det_t *dn,
                                     int64_t **row_counts, int64_t **indices, double **values) {
  int64_t cap = n * 16 + 1024, nnz = 0;
  int64_t *rc = calloc(n, sizeof(int64_t)), *idx = malloc(cap * sizeof(int64_t)); double *val = malloc(cap * sizeof(double));
  for (int64_t i = 0; i < n; i++) {
    if (nnz + i + 2 > cap) { cap = 2 * cap + i; idx = realloc(idx, cap * sizeof(int64_t)); val = realloc(val, cap * sizeof(double)); }
    idx[nnz] = i + 1; val[nnz] = orc_hamiltonian_hubbard(h, up[i], dn[i], up[i], dn[i]); nnz++; rc[i] = 1;
    for (int64_t j = 0; j < i; j++) {
      if (popcnt(up[i] ^ up[j]) + popcnt(dn[i] ^ dn[j]) != 2) continue;
      double v = orc_hamiltonian_hubbard_checked(h, up[i], dn[i], up[j], dn[j]);
      if (v == 0.0) continue;
      idx[nnz] = j + 1; val[nnz] = v; nnz++; rc[i]++;
    }
  }
  *row_counts = rc; *indices = idx; *values = val;
  return nnz;
}

/* ==================================================================== hf_to_psit step variant */
#include "sqmc_oracle_psit.c"

/* ==================================================================== the driver around the step */
#include "sqmc_oracle_ctl.c"

/* ==================================================================== the set-up in front of the walk */
#include "sqmc_oracle_setup.c"
