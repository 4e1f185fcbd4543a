/*
 * sqmc_oracle_ctl.c -- CPU ORACLE (test infrastructure, NOT product code): what the reference's walk driver does around the step --
 * the initial population (do_walk.f90:1245-1373) and the population-control scalars (do_walk.f90:2175-2184, 2880-2923) -- in C next
 * to the step, so that the checker's copy of this logic is not the product's (sqmc_amd/host.py, sqmc_gpu_run) under another name.
 * Textually included at the end of sqmc_oracle.c.
 */

/* ---- initial population, do_walk.f90:1245-1373 as the text runs: every deterministic-space determinant with weight 0 (initiator 2,
 * imp_distance 0), then every Psi_T determinant (label order, 1258) with weight w_abs_gen_begin c / sum|c| -- permanent initiator
 * (3, imp_distance 0) where |c| is within 1e-3 of max|c|, else initiator 2 at the default imp_distance 1 (1133-1136) --, all weights
 * divided by min(w_begin max|c| / sum|c|, 1), merge_sort2_up_dn, merge_original_with_spawned2(0, 0, ...).
 * With hf_to_psit: ALL of C(T) instead of Psi_T (1267-1300: weight w_begin on its first determinant only, imp_distance -2), and after
 * the merge every initiator flag but 3 is set to 2 (1367-1373).  psi_* must be in label order; returns nwalk (w->sign_perm / n_perm set). */
int64_t orc_initial_population(orc_walk *w, int64_t n_imp, const det_t *imp_up, const det_t *imp_dn, int64_t n_psi, const det_t *psi_up, const det_t *psi_dn,
                               const double *psi_c, double w_abs_gen_begin, const orc_step_params *p, int hf_to_psit, int64_t n_ct, const det_t *ct_up, const det_t *ct_dn) {
  int64_t iw = 0;
  for (int64_t i = 0; i < n_imp; i++, iw++) {            /* 1246-1253 */
    w->up[iw] = imp_up[i]; w->dn[iw] = imp_dn[i]; w->wt[iw] = 0.0; w->initiator[iw] = 2; w->imp_distance[iw] = 0;
    w->matrix_elements[iw] = 1e51; w->e_num_walker[iw] = 1e51; w->e_den_walker[iw] = 1e51;
  }
  double cmax = 0.0, csum = 0.0;
  for (int64_t k = 0; k < n_psi; k++) { if (fabs(psi_c[k]) > cmax) cmax = fabs(psi_c[k]); csum += fabs(psi_c[k]); }
  free(w->sign_perm); w->sign_perm = malloc(n_psi + 1); w->n_perm = 0;
  if (hf_to_psit) {
    for (int64_t k = 0; k < n_ct; k++, iw++) {           /* 1269-1299 */
      w->up[iw] = ct_up[k]; w->dn[iw] = ct_dn[k]; w->imp_distance[iw] = -2;
      w->matrix_elements[iw] = 1e51; w->e_num_walker[iw] = 1e51; w->e_den_walker[iw] = 1e51;
      if (k == 0) {
        w->wt[iw] = w_abs_gen_begin;
        if (fabs(fabs(psi_c[0]) - cmax) < 1e-3) { w->sign_perm[w->n_perm++] = (int8_t)(psi_c[0] > 0 ? 1 : -1); w->initiator[iw] = 3; }
        else w->initiator[iw] = 2;
      } else { w->wt[iw] = 0.0; w->initiator[iw] = 2; }
    }
  } else {
    for (int64_t k = 0; k < n_psi; k++, iw++) {          /* 1302-1326 */
      w->up[iw] = psi_up[k]; w->dn[iw] = psi_dn[k]; w->wt[iw] = w_abs_gen_begin * psi_c[k] / csum; w->imp_distance[iw] = 1;
      w->matrix_elements[iw] = 1e51; w->e_num_walker[iw] = 1e51; w->e_den_walker[iw] = 1e51;
      if (fabs(fabs(psi_c[k]) - cmax) < 1e-3) { w->sign_perm[w->n_perm++] = (int8_t)(psi_c[k] > 0 ? 1 : -1); w->initiator[iw] = 3; w->imp_distance[iw] = 0; }
      else w->initiator[iw] = 2;
    }
  }
  const double scale = (w_abs_gen_begin * cmax / csum < 1.0) ? w_abs_gen_begin * cmax / csum : 1.0;      /* 1333-1335 */
  for (int64_t i = 0; i < iw; i++) w->wt[i] = w->wt[i] / scale;
  orc_merge_sort_walkers(w, iw);                          /* 1349 */
  int64_t n = orc_merge_original_with_spawned2(w, iw, p); /* 1366 */
  if (hf_to_psit) for (int64_t i = 0; i < n; i++) if (w->initiator[i] != 3) w->initiator[i] = 2;      /* 1367-1373 */
  w->nwalk = n;
  return n;
}

/* ---- population control.  Per step, from the step's sums: e_est from the cumulated numerator and |denominator| (the reference adds a
 * block's running sums to those of the finished blocks, 2885-2887: the same number up to the order of the additions as long as the
 * denominator keeps its sign inside a block; this restatement and the library both cumulate per step -- DESIGN.md section 2), then
 * e_trial and reweight_factor_inv (2894-2901), the end of the tau ramp (2913-2923). */
typedef struct {
  double tau_sav, tau, tau_prev, e_trial, e_est, w_target, w_abs_gen, r_init_sav, r_init, irp, pop_exp, rfi, rfi_max, e_num_cum, e_den_cum;
  int64_t istep, n_equil;
  int reached, pad;
} orc_popctl;
void orc_popctl_init(orc_popctl *c, double tau, double e_trial, double w_target, double r_initiator, double initiator_rescale_power, double pop_exp,
                     double rfi_max_multiplier, int64_t n_equil_steps) {
  memset(c, 0, sizeof(*c));
  c->tau_sav = c->tau = c->tau_prev = tau; c->e_trial = c->e_est = e_trial; c->w_target = w_target;
  c->r_init_sav = c->r_init = r_initiator; c->irp = initiator_rescale_power; c->pop_exp = pop_exp;
  c->rfi = 1.0; c->rfi_max = 1.0 + rfi_max_multiplier * tau;           /* 1416 */
  c->n_equil = n_equil_steps;
}
/* 2175-2184: tau and r_initiator follow the population until the target is first reached; returns tau_ratio for minus_tau_H_values */
double orc_popctl_pre_step(orc_popctl *c, double w_abs_gen) {
  if (c->reached != 0) return 1.0;
  const double f = 1.0 + log(c->w_target / w_abs_gen);
  c->tau = c->tau_sav * f;
  const double ratio = c->tau / c->tau_prev;
  c->r_init = c->r_init_sav * pow(f, c->irp);
  return ratio;
}
/* 2880-2923; returns the tau_ratio of 2916-2921 (1 unless the target was reached in this step) */
double orc_popctl_post_step(orc_popctl *c, const double out[16]) {
  c->istep++;
  const double w_abs_gen = out[1], e_den_gen = out[2], e_num_gen = out[3];
  if (e_den_gen != 0.0) c->e_num_cum += e_num_gen * (e_den_gen > 0 ? 1.0 : -1.0);
  c->e_den_cum += fabs(e_den_gen);
  if (c->e_den_cum != 0.0) c->e_est = c->e_num_cum / c->e_den_cum;
  const double pw = (c->tau * c->pop_exp < 1.0) ? c->tau * c->pop_exp : 1.0;
  double r;
  if (c->istep <= c->n_equil) {
    const double d = c->e_est - c->e_trial, ad = fabs(d) < 1.0 ? fabs(d) : 1.0;
    c->e_trial = c->e_trial + (d > 0 ? ad : (d < 0 ? -ad : 0.0));       /* sign(min(|d|,1), d); d = 0 adds +0 */
    r = pow(c->w_target / w_abs_gen, pw);
  } else r = (1.0 / (1.0 + c->tau * (c->e_trial - c->e_est))) * pow(c->w_target / w_abs_gen, pw);
  if (r < 0.5) r = 0.5;
  if (r > 2.0) r = 2.0;
  if (r > c->rfi_max) r = c->rfi_max;
  c->rfi = r;
  double ratio = 1.0;
  if (c->reached == 0 && w_abs_gen >= c->w_target) {
    c->reached = 2; ratio = c->tau_sav / c->tau; c->tau = c->tau_sav; c->r_init = c->r_init_sav;
  }
  c->tau_prev = c->tau; c->w_abs_gen = w_abs_gen;
  return ratio;
}
