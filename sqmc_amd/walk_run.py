"""Walk decks (run_type `none`): the reference's input grammar for a projector Monte Carlo walk
(read_input, do_walk.f90:222-404; read_chem, chemistry.f90:119-245; read_heg, heg.f90:102-170;
read_hubbard's lattice scalars) driven on the GPU path, with the host-side control plane the
reference keeps on the host: equilibration in units of nblk_eq blocks until population, permanent
initiator weight and block energy have each gone up-down-up (do_walk.f90:3109-3159), the
generation and block estimators with their first-order bias correction and error bars
(2792-2843, 2986-3040), and the output lines of SURVEY appendix B (per-step equilibration line,
per-block line, `Energy=`).  Every step runs inside libsqmc_gpu (sqmc_gpu_run, one call per
block); this module only parses, schedules blocks and does the statistics.

Not every deck is a GPU deck: the proposal must be `uniform2` (the one the reference itself uses
for C2, SURVEY finding 4), importance_sampling 0, hf_to_psit f, and the trial wavefunction /
deterministic space come from ONE connect-diagonalise-truncate pass (diff_from_psi_t f with
size_deterministic; trial_wf_iterations <= 1).  Anything else stops with a message naming the line."""
import math
import re
import sys
import time

import numpy as np

from .run import _logical, _numbers


def parse_walk_deck(text):
    lines = [l for l in text.splitlines() if l.strip() and not l.lstrip().startswith("!")]
    it = iter(range(len(lines)))
    nxt = lambda: lines[next(it)]
    d = {}
    first = nxt()
    digits = "".join(ch for ch in first[:33] if ch.isdigit())           # '(4i4,x,4i4)'
    d["irand_seed"] = [[int(digits[4 * i:4 * i + 4]) for i in range(4)], [int(digits[16 + 4 * i:20 + 4 * i]) for i in range(4)]]
    d["run_type"] = nxt().split()[0].strip("'\"").lower()
    if d["run_type"] not in ("none", "no_fixed_node"):
        raise SystemExit("sqmc_amd.walk_run: run_type %r is not a plain projector walk (none / no_fixed_node)" % d["run_type"])
    a = _numbers(nxt(), 4); d.update(nstep=int(a[0]), nblk=int(a[1]), nblk_eq=int(a[2]), ipr=int(a[3]))
    a = _numbers(nxt(), 3); d.update(w_abs_gen_begin=a[0], w_abs_gen_target=a[1], mwalk=int(a[2]))
    a = _numbers(nxt(), 2); d.update(tau_multiplier=a[0], tau=a[1])
    a = _numbers(nxt(), 2); d.update(rfi_max_multiplier=a[0], rfi_max=a[1])
    a = _numbers(nxt(), 3); d.update(population_control_exponent=a[0], e_trial_initial=a[1], min_wt=a[2])
    toks = nxt().replace(",", " ").split()
    d["proposal_method"] = toks[0].strip("'\"").lower()
    a = [float(t) for t in toks[1:5]]
    d.update(importance_sampling=int(a[0]), r_initiator=a[1], initiator_power=int(a[2]), initiator_rescale_power=a[3])
    toks = nxt().replace(",", " ").split()
    d["semistochastic"], d["use_exponential_projector"] = _logical(toks[0]), _logical(toks[1])
    d.update(hf_to_psit=False, c_t_initiator=False, always_spawn_cutoff_wt=0.5, size_deterministic=0)
    if d["semistochastic"]:
        d["diff_from_psi_t"] = _logical(nxt().split()[0])
        if d["diff_from_psi_t"]:
            raise SystemExit("sqmc_amd.walk_run: diff_from_psi_t = t (a deterministic space grown separately from Psi_T) is not on the GPU path; use f + size_deterministic")
        d["size_deterministic"] = int(_numbers(nxt(), 1)[0])
        toks = nxt().replace(",", " ").split()
        d.update(hf_to_psit=_logical(toks[0]), c_t_initiator=_logical(toks[1]), always_spawn_cutoff_wt=float(toks[2]))
    toks = nxt().replace(",", " ").split()
    d["hamiltonian_type"] = toks[0].strip("'\"").lower()
    if d["hamiltonian_type"] == "chem":
        a = _numbers(nxt(), 2); d.update(nelec=int(a[0]), nup=int(a[1]))
        d["point_group"] = nxt().split()[0].strip("'\"").lower()
        d["time_sym"] = _logical(nxt().split()[0])
        d["z"] = int(_numbers(nxt(), 1)[0]) if d["time_sym"] else 1
        d["norb"] = int(_numbers(nxt(), 1)[0])
        d["n_core_orb"] = int(_numbers(nxt(), 1)[0])
        d["trial_wf_iters"] = int(_numbers(nxt(), 1)[0])
        n = max(d["trial_wf_iters"], 1)
        d["norb_trial_wf"] = [int(v) for v in _numbers(nxt(), 1)][:n]
        d["n_initiators_trial_wf"] = [int(v) for v in _numbers(nxt(), 1)][:n]
        d["n_truncate_trial_wf"] = [int(v) for v in _numbers(nxt(), 1)][:n]
        d["orbital_symmetries"] = [int(x) for x in _numbers(nxt(), d["norb"])]
    elif d["hamiltonian_type"] == "heg":
        d["n_dim"] = int(_numbers(nxt(), 1)[0]); d["r_s"] = _numbers(nxt(), 1)[0]
        a = _numbers(nxt(), 2); d.update(nelec=int(a[0]), nup=int(a[1]), cutoff_radius=_numbers(nxt(), 1)[0])
    elif d["hamiltonian_type"] == "hubbard2":
        # the lattice scalars of read_hubbard in the order l_x l_y / pbc / t U / nup ndn; its trial-wavefunction block is not read
        a = _numbers(nxt(), 2); d.update(l_x=int(a[0]), l_y=int(a[1]))
        d["pbc"] = _logical(nxt().split()[0])
        a = _numbers(nxt(), 2); d.update(t=a[0], U=a[1])
        a = _numbers(nxt(), 2); d.update(nup=int(a[0]), ndn=int(a[1]))
    else:
        raise SystemExit("sqmc_amd.walk_run: hamiltonian_type %r has no GPU operator (chem, heg, hubbard2)" % d["hamiltonian_type"])
    if d["proposal_method"] not in ("uniform2", "uniform") and not (d["proposal_method"] == "fast_heatbath" and d["hamiltonian_type"] == "chem"):
        raise SystemExit("sqmc_amd.walk_run: proposal_method %r is not on the GPU path (uniform2: off_diagonal_move_chem / _heg / _hubbard; fast_heatbath: chem)" % d["proposal_method"])
    if d["proposal_method"] == "fast_heatbath" and d["hf_to_psit"]:
        raise SystemExit("sqmc_amd.walk_run: proposal_method fast_heatbath with hf_to_psit = t is not built")
    if d["importance_sampling"] != 0:
        raise SystemExit("sqmc_amd.walk_run: importance_sampling must be 0")
    if d["hf_to_psit"] and d["hamiltonian_type"] == "hubbard2":
        raise SystemExit("sqmc_amd.walk_run: hf_to_psit = t needs the first determinant of Psi_T to be the first determinant of C(T) (chem, heg)")
    if d["use_exponential_projector"]:
        raise SystemExit("sqmc_amd.walk_run: use_exponential_projector = t is not on the GPU path (linear projector)")
    if d["hamiltonian_type"] == "chem" and d["trial_wf_iters"] > 1:
        raise SystemExit("sqmc_amd.walk_run: trial_wf_iterations > 1 is not on the GPU path (one connect-diagonalise-truncate pass)")
    return d


class WalkStats:
    """The estimators of do_walk.f90: per generation (2792-2843) and per block (2986-3040), running
    means and variances by Welford's update, ratio of averages with its first-order bias removed."""

    def __init__(self):
        self.zero()

    def zero(self):                     # the zeroing at the start of every equilibration set and of the main run (2138-2151)
        self.passes = 0
        self.n_ave = self.n_vn1 = self.d_ave = self.d_vn1 = self.nd_vn1 = 0.0          # generation: num, den, covariance
        self.e_genabs_ave = self.e_genabs_err = 0.0
        self.e_gen_ave = self.e_gen_vn1 = self.e_gen_gen1 = self.e_genpp_ave = 0.0
        self.e_genp_del = self.e_genp_e_genpp_vn1 = 0.0
        self.t_corr_nonint = 0.0
        self.iblk = 0
        self.bn_ave = self.bn_vn1 = self.bd_ave = self.bd_vn1 = self.bnd_vn1 = 0.0     # block
        self.e_blkabs_ave = self.e_blkabs_err = 0.0
        self.e_num_genabs_cum = self.e_den_genabs_cum = 0.0
        self.e_num_blkabs_cum = self.e_den_blkabs_cum = 0.0
        self.nwalk_cum = self.nwalk_before_cum = self.w_abs_cum = self.w_abs_before_cum = self.w_cum = self.w_perm_cum = 0.0

    @staticmethod
    def _ratio(n_ave, d_ave, n_var, d_var, nd_var, count):
        r = n_ave / d_ave
        if count < 2:
            return r, 0.0
        ave = r / (1 + (d_var / d_ave ** 2 - nd_var / (n_ave * d_ave)) / count)
        arg = (n_var / n_ave ** 2 + d_var / d_ave ** 2 - 2 * nd_var / (n_ave * d_ave)) / count
        return ave, abs(r) * math.sqrt(max(arg, 0.0))

    def generation(self, e_num_gen, e_den_gen):
        self.passes += 1
        p = self.passes
        n, dd = e_num_gen * math.copysign(1.0, e_den_gen), abs(e_den_gen)
        self.e_num_genabs_cum += n; self.e_den_genabs_cum += dd
        n_del = n - self.n_ave; self.n_ave += n_del / p; self.n_vn1 += (n - self.n_ave) * n_del
        d_del = dd - self.d_ave; self.d_ave += d_del / p; self.d_vn1 += (dd - self.d_ave) * d_del
        self.nd_vn1 += (n - self.n_ave) * d_del
        if p >= 2:
            self.e_genabs_ave, self.e_genabs_err = self._ratio(self.n_ave, self.d_ave, self.n_vn1 / (p - 1), self.d_vn1 / (p - 1), self.nd_vn1 / (p - 1), p)
        else:
            self.e_genabs_ave, self.e_genabs_err = self.n_ave / self.d_ave, 0.0
        # non-integrated autocorrelation time from the lag-1 covariance of e_gen (2845-2872)
        e_gen = e_num_gen / e_den_gen
        e_gen_del = e_gen - self.e_gen_ave
        self.e_gen_ave += e_gen_del / p
        self.e_gen_vn1 += (e_gen - self.e_gen_ave) * e_gen_del
        if p >= 2:
            e_gen_var = self.e_gen_vn1 / (p - 1)
            e_genpp_del = e_gen - self.e_genpp_ave
            self.e_genpp_ave = (p * self.e_gen_ave - self.e_gen_gen1) / (p - 1)
            if p >= 3:
                self.e_genp_e_genpp_vn1 += self.e_genp_del * e_genpp_del
                cov = self.e_genp_e_genpp_vn1 / (p - 2)
                if cov > 0 and e_gen_var > 0 and cov < e_gen_var:
                    self.t_corr_nonint = 1 + 2 * (-1 / math.log(cov / e_gen_var))
        else:
            self.e_gen_gen1 = e_gen
        self.e_genp_del = e_gen_del

    def block(self, e_num_blk, e_den_blk):
        self.iblk += 1
        k = self.iblk
        n, dd = e_num_blk * math.copysign(1.0, e_den_blk), abs(e_den_blk)
        self.e_num_blkabs_cum += n; self.e_den_blkabs_cum += dd
        n_del = n - self.bn_ave; self.bn_ave += n_del / k; self.bn_vn1 += (n - self.bn_ave) * n_del
        d_del = dd - self.bd_ave; self.bd_ave += d_del / k; self.bd_vn1 += (dd - self.bd_ave) * d_del
        self.bnd_vn1 += (n - self.bn_ave) * d_del
        if k >= 2:
            self.e_blkabs_ave, self.e_blkabs_err = self._ratio(self.bn_ave, self.bd_ave, self.bn_vn1 / (k - 1), self.bd_vn1 / (k - 1), self.bnd_vn1 / (k - 1), k)
        else:
            self.e_blkabs_ave, self.e_blkabs_err = self.bn_ave / self.bd_ave, 0.0

    @property
    def e_genabs_av(self):
        return self.e_num_genabs_cum / self.e_den_genabs_cum if self.e_den_genabs_cum else 0.0

    @property
    def t_corr(self):
        return (self.e_blkabs_err / self.e_genabs_err) ** 2 if (self.iblk >= 2 and self.e_genabs_err > 0) else 0.0

    @property
    def e_blkabs_corrected_err(self):
        tc = self.t_corr
        return self.e_blkabs_err * math.sqrt(max(1.0, self.t_corr_nonint / tc)) if tc > 0 else self.e_blkabs_err


def _nint(x):
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def run_walk(deck, fcidump="FCIDUMP", out=sys.stdout, walkalize=None, max_equil_sets=50, psit_con_in=None, psit_con_out=None,
             dtm_elems_in=None, dtm_elems_out=None):
    import torch            # noqa: F401  one libamdhip64 per process
    import sqmc_amd
    from . import host as H
    p = lambda *a: (print(*a, file=out), out.flush())
    sqmc_amd.set_device(0)
    d = deck
    semi = 1 if d["semistochastic"] else 0
    if d["hamiltonian_type"] == "chem":
        hst = H.ChemHost(fcidump, d["nelec"], d["nup"], d["point_group"], time_sym=d["time_sym"], z=d["z"], n_core_orb=d["n_core_orb"])
        if hst.norb != d["norb"]:
            raise SystemExit("norb of the deck (%d) and of %s (%d) differ" % (d["norb"], fcidump, hst.norb))
        n_psi = d["n_truncate_trial_wf"][0] if d["trial_wf_iters"] >= 1 else 1
        skw = dict(n_truncate_trial_wf=n_psi, size_deterministic=max(d["size_deterministic"], 1), tau_multiplier=d["tau_multiplier"])
    elif d["hamiltonian_type"] == "heg":
        hst = H.HegHost(d["n_dim"], d["r_s"], d["nelec"], d["nup"], d["cutoff_radius"])
        skw = dict(n_truncate_trial_wf=1, size_deterministic=max(d["size_deterministic"], 1), tau_multiplier=d["tau_multiplier"])
    else:
        hst = H.HubbardHost(d["l_x"], d["l_y"], d["pbc"], d["nup"], d["ndn"], d["t"], d["U"])
        skw = dict(n_truncate_trial_wf=20, size_deterministic=max(d["size_deterministic"], 1), tau_multiplier=d["tau_multiplier"])
    p("nstep, nblk, nblk_eq, ipr=%8d%8d%5d%5d" % (d["nstep"], d["nblk"], d["nblk_eq"], d["ipr"]))
    p("w_abs_gen_begin, w_abs_gen_target, MWALK=%8d%9d%10d" % (int(d["w_abs_gen_begin"]), int(d["w_abs_gen_target"]), d["mwalk"]))
    p("tau_multiplier, tau=%13.8f%13.8f" % (d["tau_multiplier"], d["tau"]))
    p("population_control_exponent, e_trial_initial, min_wt=%11.5f%11.5f%6.2f" % (d["population_control_exponent"], d["e_trial_initial"], d["min_wt"]))
    p("proposal_method, importance_sampling, r_initiator, initiator_power, initiator_min_distance, initiator_rescale_power= %s%3d%6.2f%3d%3d%6.3f"
      % (d["proposal_method"], 0, d["r_initiator"], d["initiator_power"], 0, d["initiator_rescale_power"]))
    p("\nsemistochastic run" if semi else "\nnot semistochastic run")
    target = d["w_abs_gen_target"]
    mwalk = int(max(d["mwalk"], 4 * (target / d["min_wt"] + d["size_deterministic"])))      # do_walk.f90:665, 856
    psit = bool(d["hf_to_psit"]) and semi
    p(" Replacing HF state with trial wave function" if psit else " NOT replacing HF state with trial wave function")      # do_walk.f90:383-387
    g = hst.gpu(rng_mode=H.RNG_COUNTER, seed=tuple(d["irand_seed"][1]), mwalk=mwalk)
    t0 = time.perf_counter()
    # the trial wave function is rediagonalised among its own determinants, as generate_space_iterate leaves it (semistoch.f90:575, 706-712)
    s = hst.setup_walk(g, rediagonalize=True, **skw) if d["hamiltonian_type"] != "hubbard2" else hst.setup_walk(g, **skw)
    if d["tau"] != 0:                                         # an explicit tau overrides tau_multiplier (do_walk.f90:1396-1412)
        s.prj_values = s.prj_values * (d["tau"] / s.tau); s.tau = d["tau"]
    n_core = d.get("n_core_orb", 0)
    # the reference's restart files (use_psit_con_in/out, use_elems_in/out; hamiltonian_mod.f90:1340-1400): C(T) and the
    # deterministic space can come from, or go to, the text files a reference run writes and reads
    if psit_con_in:
        p(" Reading in the local energies from the file " + psit_con_in)
        s.ct_up, s.ct_dn, s.ct_num, s.ct_den = H.read_psit_connections(psit_con_in, hst.nup, hst.ndn, n_core)
        in_t = np.abs(s.ct_den) > 1e-12
        s.psi_up, s.psi_dn, s.psi_c = s.ct_up[in_t], s.ct_dn[in_t], s.ct_den[in_t]
        s.e_trial0 = float(np.dot(s.ct_num, s.ct_den) / np.dot(s.ct_den, s.ct_den))
    if dtm_elems_in and semi:
        p(" Reading in the matrix elements from the file " + dtm_elems_in)
        s.imp_up, s.imp_dn, s.prj_counts, s.prj_indices, hv, s.e_var = H.read_dtm_elems(dtm_elems_in, hst.nup, hst.ndn, n_core)
        s.prj_values = -s.tau * hv
    if psit_con_out:
        p(" Dumping the local energies into file " + psit_con_out)
        H.write_psit_connections(psit_con_out, s.psi_up, s.ct_up, s.ct_dn, s.ct_num, s.ct_den, hst.nup, hst.ndn, hst.norb, n_core)
    if dtm_elems_out and semi:
        p(" Dumping the deterministic matrix elements into file " + dtm_elems_out)
        H.write_dtm_elems(dtm_elems_out, s.imp_up, s.imp_dn, s.prj_counts, s.prj_indices, s.prj_values / (-s.tau), s.e_var, n_core)
    if d["proposal_method"] == "fast_heatbath":
        # setup_efficient_heatbath + check_heatbath_unbiased (chemistry.f90:1002-1225, 9330-9375), done by the library
        p(" Checking whether there is any potential bias for using heatbath on this system...")
        if not g.setup_efficient_heatbath():
            p(" Check failed! Heatbath is (potentially) biased for this system. Aborting run...")
            g.close()
            raise SystemExit("Heatbath may be biased for this system!")
        p(" Check passed! Heatbath is unbiased for this system!")
        mwalk_hb = int(max(d["mwalk"], 3.5 * (target / d["min_wt"] + d["size_deterministic"])))      # do_walk.f90:668-669
        p("1Setting MWALK=3.5*(w_abs_gen_target/min_wt+n_imp)=%10d" % mwalk_hb)
    p("ndet_psi_t, ndet_psi_t_connected, n_imp=%8d%10d%8d" % (len(s.psi_up), len(s.ct_up), len(s.imp_up)))
    p("tau=%12.8f  variational energy of the set-up space=%16.8f" % (s.tau, s.e_var))
    if psit:
        # hf_to_psit: the matrix loses its first row and column, all of C(T) becomes resident, MWALK follows do_walk.f90:653-655
        ix, cdet, diag, pcnt, pidx, pval, in_imp = H.psit_tables(g, s)
        need = int(max(d["mwalk"], 3 * (target / d["min_wt"] + len(s.ct_up))))
        if need > mwalk:
            g.close()
            mwalk = need
            g = hst.gpu(rng_mode=H.RNG_COUNTER, seed=tuple(d["irand_seed"][1]), mwalk=mwalk)
        p("Setting MWALK=3*(w_abs_gen_target/min_wt+ndet_psi_t_connected)/ncores=%10d" % mwalk)
        g.set_projector(pcnt, pidx, pval)
        g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
        g.set_hf_to_psit(ix, cdet, diag)
        wk = H.initial_walkers_psit(s, cdet, in_imp, d["w_abs_gen_begin"])
    else:
        if semi:
            g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
        g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
        wk = H.initial_walkers(s, d["w_abs_gen_begin"], r_initiator=d["r_initiator"], initiator_power=d["initiator_power"])
    if not semi:
        wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)
        keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
        wk = {k: v[keep] for k, v in wk.items()}
    g.upload_walkers(wk)
    e_trial = d["e_trial_initial"] if d["e_trial_initial"] != 0 else s.e_trial0
    pc = H.PopControl(s.tau, e_trial, target, r_initiator=d["r_initiator"], initiator_rescale_power=d["initiator_rescale_power"],
                      pop_exp=d["population_control_exponent"], rfi_max_multiplier=d["rfi_max_multiplier"], n_equil_steps=10 ** 18)
    if d["rfi_max"] != 0:
        pc.rfi_max = d["rfi_max"]
    w_abs = float(np.abs(wk["wt"]).sum())
    nstep, nblk, nblk_eq = d["nstep"], d["nblk"], d["nblk_eq"]
    st = WalkStats()
    wfile = open(walkalize, "w") if walkalize else None
    ntimes, iblkk = 1, 1
    eq_w = eq_perm = eq_e = eq_all = 0
    w_abs_blk_prev = w_perm_blk_prev = 0.0
    e_blk_prev = 1e300
    n_walker_steps, t_steps = 0.0, 0.0
    announced_reached = False
    g.set_chained_runs(True)          # one sqmc_gpu_run per block: the last step of a block enqueues the head of the next block's first step
    while iblkk <= ntimes * nblk_eq + nblk:
        in_equil = iblkk <= ntimes * nblk_eq
        if (nblk_eq == 1 or iblkk % nblk_eq == 1) and iblkk <= ntimes * nblk_eq + 1:
            st.zero(); pc.e_num_cum = pc.e_den_cum = 0.0          # e_est restarts from the running block
        pc.n_equil = 10 ** 18 if in_equil else 0                 # e_trial follows e_est only during equilibration (2894-2901)
        pcc = pc.to_c(w_abs, min_wt=d["min_wt"], cutoff=d["always_spawn_cutoff_wt"], initiator_power=d["initiator_power"], semistochastic=semi)
        pcc.c_t_initiator = 1 if d["c_t_initiator"] else 0
        if not in_equil:
            pcc.istep, pcc.n_equil = 1, 0
        ts = time.perf_counter()
        stats, _ = g.run(pcc, nstep, True)
        t_steps += time.perf_counter() - ts
        pc.from_c(pcc); w_abs = pcc.w_abs_gen
        e_num_blk = e_den_blk = w_abs_blk = w_perm_blk = w_imp_blk = 0.0
        for k in range(nstep):
            o = stats[k]
            w_gen, w_abs_gen, e_den_gen, e_num_gen, w_perm, nwalk, w_imp, nbefore = o[0], o[1], o[2], o[3], o[4], int(o[5]), o[6], o[7]
            st.generation(e_num_gen, e_den_gen)
            e_num_blk += e_num_gen; e_den_blk += e_den_gen; w_abs_blk += w_abs_gen; w_perm_blk += w_perm; w_imp_blk += w_imp
            st.nwalk_cum += nwalk; st.nwalk_before_cum += nbefore; st.w_abs_cum += w_abs_gen; st.w_abs_before_cum += o[14]; st.w_cum += w_gen
            st.w_perm_cum += w_perm
            n_walker_steps += nwalk
            if wfile:
                wfile.write("%10d%12.6f%13.6E%19.12f%9d\n" % (nstep * (iblkk - 1) + k + 1, 1.0, w_abs_gen, e_num_gen / e_den_gen, nwalk))
            if not announced_reached and w_abs_gen >= target:
                announced_reached = True
                p("\nw_abs_gen_target=%9d reached at iblkk, istep=%6d%6d, tau, r_initiator reset to actual tau=%10.6f%10.6f\n" % (int(target), iblkk, k + 1, s.tau, d["r_initiator"]))
        st.block(e_num_blk, e_den_blk)
        e_blk = e_num_blk / e_den_blk
        p("iblk, w_perm_initiator, nwalk, w_abs, w_abs_imp=%6d%9.1f%9d%9d%8d e_blk=%10.4f e=%14.8f(%8d)%14.8f(%8d)%14.8f(%8d) e_trial, rew_fac_inv, t_c=%11.5f%11.5f%8.1f"
          % (st.iblk, w_perm_blk / nstep, nwalk, _nint(w_abs_blk / nstep), _nint(w_imp_blk / nstep), e_blk, st.e_genabs_av, _nint(1e8 * st.e_genabs_err),
             st.e_blkabs_ave, _nint(1e8 * st.e_blkabs_err), st.e_blkabs_ave, _nint(1e8 * st.e_blkabs_corrected_err), pc.e_trial, pc.rfi, st.t_corr_nonint))
        # ---- equilibration judged on blocks: population, permanent-initiator weight and energy each up-down-up (3109-3159)
        reached = pc.reached == 2
        tail = ", w_perm_initiator, w_abs_imp, e_blk_ave, e_blk=%9.1f%9.1f%14.6f%14.6f" % (w_perm_blk / nstep, w_imp_blk / nstep, st.e_blkabs_ave, e_blk)
        if reached and eq_w == 0 and w_abs_blk <= w_abs_blk_prev: eq_w = 1
        if eq_w == 1 and w_abs_blk >= w_abs_blk_prev:
            eq_w = 2; p("Equilibration of population achieved at iblkk=%6d" % iblkk + tail)
        if reached and eq_perm == 0 and w_perm_blk <= w_perm_blk_prev: eq_perm = 1
        if eq_perm == 1 and w_perm_blk >= w_perm_blk_prev:
            eq_perm = 2; p("Equilibration of perm_initiator population achieved at iblkk=%6d" % iblkk + tail)
        if reached and eq_e == 0 and e_blk >= e_blk_prev: eq_e = 1
        if eq_e == 1 and e_blk <= e_blk_prev:
            eq_e = 2; p("Equilibration of energy achieved at iblkk=%6d" % iblkk + tail)
        if eq_w == 2 and eq_perm == 2 and eq_e == 2 and eq_all <= 1:
            eq_all = 2; p("Equilibration of everything achieved at iblkk=%6d" % iblkk + tail)
        if eq_all <= 1 and iblkk == ntimes * nblk_eq:
            if ntimes >= max_equil_sets:
                p("sqmc_amd: not equilibrated after %d sets of nblk_eq blocks; starting the main run" % ntimes)
                eq_all = 2
            else:
                ntimes += 1
        w_abs_blk_prev, w_perm_blk_prev, e_blk_prev = w_abs_blk, w_perm_blk, e_blk
        iblkk += 1
    g.set_chained_runs(False)
    passes = float(nstep) * nblk
    p("e_genabs_ave, e_genabs_err, e_blkabs_ave, e_blkabs_err=%12.4E%12.4E%12.4E%12.4E" % (st.e_genabs_ave, st.e_genabs_err, st.e_blkabs_ave, st.e_blkabs_err))
    ratio_n = st.nwalk_cum / st.nwalk_before_cum if st.nwalk_before_cum else 0.0
    ratio_w = st.w_abs_cum / st.w_abs_before_cum if st.w_abs_before_cum else 0.0
    p("w_perm_initiator_av, nwalk_av, nwalk_before_merge_av, ratio, w_av, w_abs_av, w_abs_before_merge_av, ratio=%8.1f%10.1f%10.1f%6.2f%10.1f%10.1f%10.1f%6.2f"
      % (st.w_perm_cum / passes, st.nwalk_cum / passes, st.nwalk_before_cum / passes, ratio_n, st.w_cum / passes, st.w_abs_cum / passes, st.w_abs_before_cum / passes, ratio_w))
    p("Energy=%14.8f(%8d)%14.8f(%8d)%14.8f(%8d) energy_exact=%10.5f t_corr_nonint, t_corr, nstep=%8.2f%8.2f%6d"
      % (st.e_genabs_av, _nint(1e8 * st.e_genabs_err), st.e_blkabs_ave, _nint(1e8 * st.e_blkabs_err), st.e_blkabs_ave, _nint(1e8 * st.e_blkabs_corrected_err),
         0.0, st.t_corr_nonint, st.t_corr, nstep))
    p("\nsqmc_amd: set-up %.2f s; %d steps in %.3f s inside libsqmc_gpu: %.3e walker-steps/s" % (t0 and (time.perf_counter() - t0 - t_steps), int(nstep * (iblkk - 1)), t_steps, n_walker_steps / t_steps))
    if wfile:
        wfile.write("%6d%6d%9d%12.6f%10.6f nstep, nblk, w_abs_gen_target, e_trial, tau\n" % (nstep, nblk, int(target), pc.e_trial, s.tau))
        wfile.close()
    g.close()
    return dict(energy=st.e_blkabs_ave, energy_err=st.e_blkabs_err, energy_gen=st.e_genabs_av, energy_gen_err=st.e_genabs_err, n_blocks_total=iblkk - 1,
                n_equil_sets=ntimes, e_trial=pc.e_trial, nwalk_av=st.nwalk_cum / passes, tau=s.tau, n_imp=len(s.imp_up), n_ct=len(s.ct_up),
                walker_steps_per_s=n_walker_steps / t_steps)
