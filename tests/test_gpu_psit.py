"""GPU parity of the step variant hf_to_psit = .true.: the HIP path through the C ABI against the CPU
oracle's restatement of the reference's three-segment algorithm (oracle/sqmc_oracle_psit.c) on identical inputs.

PARITY UNPINNED BY THE REFERENCE: it ships no walk fixture and its merge for this variant cannot run as written
(tests/golden/README_hf_to_psit.md).  What is compared: walker lists, weights and flags bit for bit (the GPU keeps ONE list sorted
by a key that puts C(T) first; the oracle keeps the reference's segments), the 16 sums to the tolerance of test_gpu_parity."""
import numpy as np
import pytest

from conftest import gpu_ctx_from_oracle, gpu_ctx_heg
from test_gpu_parity import _sums_close, SEED

pytestmark = pytest.mark.gpu


def _pair(oracle, sysm, s, q, w_begin, rng_mode, sum_order=1, mwalk=600000, heg=False):
    import sqmc_amd
    wk = oracle.initial_walkers_psit(s, q, w_begin)
    ow = oracle.OracleWalk(sysm, s, wk, mwalk, SEED, rng_mode=rng_mode, psit=q, sum_order=sum_order)
    mk = gpu_ctx_heg if heg else gpu_ctx_from_oracle
    g = mk(sysm, rng_mode=sqmc_amd.RNG_REPLAY if rng_mode == 0 else sqmc_amd.RNG_COUNTER, seed=SEED, mwalk=mwalk)
    g.set_projector(q.prj_counts, q.prj_indices, q.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    g.set_hf_to_psit(q.loc_psit + 1, q.cdet, q.diag_elems, sum_order)
    g.upload_walkers(wk)
    return ow, g, float(np.abs(wk["wt"]).sum())


def _same_walkers(wg, wc, n_ct):
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        if not np.array_equal(wg[k], wc[k]):
            bad = np.nonzero(wg[k][:min(len(wg[k]), len(wc[k]))] != wc[k][:min(len(wg[k]), len(wc[k]))])[0]
            return "%s differs (%d / %d walkers; first at %s of n_ct %d)" % (k, len(wg[k]), len(wc[k]), bad[:3], n_ct)
    held = wc["matrix_elements"] < 1e50             # the library fills H_ii when it needs it, as the reference does
    if not np.array_equal(wg["matrix_elements"][held], wc["matrix_elements"][held]):
        return "matrix_elements differ"
    return ""


def _lockstep(oracle, ow, g, pc, w_abs, nsteps, n_ct, check_every=10, min_wt=0.5):
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        prm = pc.params(min_wt=min_wt)
        st, oc = ow.step(prm)
        assert st == 0, (st, it)
        og = g.step(prm)
        assert og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (it, og, oc)
        assert _sums_close(og, oc), (it, og, oc)
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
        if (it + 1) % check_every == 0 or it + 1 == nsteps:
            msg = _same_walkers(g.download_walkers(), ow.walkers(), n_ct)
            assert not msg, (it, msg)
    return w_abs


@pytest.fixture(scope="module")
def c2_psit(oracle, c2_walk):
    s = oracle.setup_walk(c2_walk, 100, 1000, 0.1, rediagonalize=True)
    return s, oracle.psit_setup(c2_walk, s)


def test_psit_counter_trajectory_bit_exact(oracle, c2_walk, c2_psit):
    """150 COUNTER steps on C2 r = 1.24253 from the first state alone through the tau ramp to 10^4 walkers' worth of weight and on:
    spawns onto C(T), onto survivors outside it and onto new determinants, discards, rounding, T^-1."""
    s, q = c2_psit
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 100.0, 1)
    pc = oracle.PopControl(s.tau, s.e_trial0, 10000)
    _lockstep(oracle, ow, g, pc, w_abs, 150, len(s.ct_up), check_every=15)
    wc = ow.walkers()
    n_out = len(wc["up"]) - len(s.ct_up)
    assert n_out == ow.n_outside_ct() and n_out > 5000
    assert pc.reached == 2
    g.close(); ow.close()


def test_psit_replay_trajectory_bit_exact(oracle, c2_walk, c2_psit):
    """the reference's single rannyu stream, draw for draw: the zero-weight determinants of C(T) each take a gate draw (do_walk.f90:3577),
    the first state takes none (3574); the stream's state after every block of steps is the oracle's"""
    s, q = c2_psit
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 50.0, 0)
    pc = oracle.PopControl(s.tau, s.e_trial0, 3000)
    for _ in range(4):
        w_abs = _lockstep(oracle, ow, g, pc, w_abs, 10, len(s.ct_up), check_every=10)
        assert g.rng_state() == ow.rng_state()
    g.close(); ow.close()


def test_psit_left_to_right_sums_bit_exact(oracle, c2_walk, c2_psit):
    """sum_order 0: the three long sums in the reference's own order (one lane on the GPU), against the oracle in that order"""
    s, q = c2_psit
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 100.0, 1, sum_order=0)
    pc = oracle.PopControl(s.tau, s.e_trial0, 5000)
    _lockstep(oracle, ow, g, pc, w_abs, 40, len(s.ct_up), check_every=20)
    g.close(); ow.close()


def test_psit_c_t_initiator_and_time_sym(oracle, c2_hci):
    """the shipped decks' conventions (time-reversal representatives, z = +1) and c_t_initiator = true: children of C(T) are born
    initiators at distance 1 (3703-3706, 3723), and only the first n_permanent_initiator slots of C(T) see check_initiator (2456-2460)"""
    s = oracle.setup_walk(c2_hci, 30, 200, 0.1, rediagonalize=True)
    q = oracle.psit_setup(c2_hci, s)
    ow, g, w_abs = _pair(oracle, c2_hci, s, q, 100.0, 1)
    pc = oracle.PopControl(s.tau, s.e_trial0, 8000)
    n_ct = len(s.ct_up)
    for it in range(60):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        prm = pc.params()
        prm["c_t_initiator"] = 1
        st, oc = ow.step(prm)
        assert st == 0
        og = g.step(prm)
        assert og[5] == oc[5] and og[15] == oc[15] and _sums_close(og, oc), (it, og, oc)
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
    msg = _same_walkers(g.download_walkers(), ow.walkers(), n_ct)
    assert not msg, msg
    g.close(); ow.close()


def test_psit_heg_trajectory_bit_exact(oracle, heg14):
    """the electron gas (hamiltonian_type heg): the same variant with off_diagonal_move_heg / hamiltonian_heg as the operator pair"""
    s = oracle.setup_walk_heg(heg14, 250, 0.1, 20, rediagonalize=True)
    q = oracle.psit_setup(heg14, s)
    ow, g, w_abs = _pair(oracle, heg14, s, q, 50.0, 1, heg=True)
    pc = oracle.PopControl(s.tau, s.e_trial0, 5000)
    _lockstep(oracle, ow, g, pc, w_abs, 60, len(s.ct_up), check_every=20)
    g.close(); ow.close()


@pytest.mark.parametrize("fuse", [True, False])
def test_psit_pipelined_steps_bit_exact(oracle, c2_walk, c2_psit, monkeypatch, fuse):
    """past the target population a chained host's step enqueues the next step's head behind its own tail (sqmc_gpu_set_chained_runs);
    with hf_to_psit the gate of the next step is written by k_anneal<., 1> (everything outside C(T)) and k_psit_finish (the C(T)
    segment, whose weights are final only there), or by k_gate in the head (SQMC_PSIT_NO_GATE_FUSION).  Same trajectory as the oracle's
    unpipelined steps, bit for bit, through 120 steps of which 100 are pipelined."""
    if not fuse:
        monkeypatch.setenv("SQMC_PSIT_NO_GATE_FUSION", "1")
    s, q = c2_psit
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 100.0, 1)
    g.set_chained_runs(True)
    pc = oracle.PopControl(s.tau, s.e_trial0, 3000)
    _lockstep(oracle, ow, g, pc, w_abs, 120, len(s.ct_up), check_every=20)
    assert pc.reached == 2
    g.close(); ow.close()


def test_psit_heg_pipelined_steps_bit_exact(oracle, heg14):
    """the same for the electron gas: chained steps past the target population against the oracle's unpipelined ones"""
    s = oracle.setup_walk_heg(heg14, 200, 0.1, n_truncate_trial_wf=20, rediagonalize=True)
    q = oracle.psit_setup(heg14, s)
    ow, g, w_abs = _pair(oracle, heg14, s, q, 100.0, 1, heg=True)
    g.set_chained_runs(True)
    pc = oracle.PopControl(s.tau, s.e_trial0, 250)
    _lockstep(oracle, ow, g, pc, w_abs, 160, len(s.ct_up), check_every=20)
    assert pc.reached == 2
    g.close(); ow.close()


def test_psit_run_loop_equals_single_steps_at_the_bench_size(c2_walk):
    """BASELINE configs[1]'s population (w_abs_gen_target 10^5; 76,900 resident C(T) determinants + the survivors outside) with
    hf_to_psit: sqmc_gpu_run -- population control in the library, pipelined steps past the target -- leaves the same walkers, bit
    for bit, as single unpipelined sqmc_gpu_step calls driven by the host's PopControl; and the layout's invariants hold there."""
    from sqmc_amd import host as H
    h = H.ChemHost(c2_walk_fcidump(), 8, 4, "d2h")
    res = []
    for mode in ("run", "step"):
        gw = H.GpuWalk(h, 1e5, w_begin=1e4, hf_to_psit=True, seed=(1346, 5634, 6635, 4361))
        if mode == "run":
            gw.run(260, keep_stats=False)
        else:
            for _ in range(260):
                gw.step()
        res.append(gw.g.download_walkers())
        reached, n_ct, setup = gw.pc.reached, len(gw.setup.ct_up), gw.setup
        gw.close()
        assert reached == 2
    a, b = res
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        assert np.array_equal(a[k], b[k]), k
    assert len(a["up"]) > n_ct + 40000
    assert np.array_equal(a["up"][:n_ct], setup.ct_up) and np.array_equal(a["dn"][:n_ct], setup.ct_dn)
    u, d = a["up"][n_ct:], a["dn"][n_ct:]
    assert np.all((u[1:] > u[:-1]) | ((u[1:] == u[:-1]) & (d[1:] > d[:-1]))) and a["imp_distance"][n_ct:].min() >= 1
    assert np.all(np.abs(a["wt"][n_ct:]) > 0)


def test_psit_run_loop_and_energy(c2_walk):
    """sqmc_gpu_run with hf_to_psit (population control inside the library) from the product's own set-up: the projected energy of the
    transformed walk agrees with the untransformed walk's and with this geometry's HCI+PT2 total (-75.72854 Ha, pinned to the
    reference's printed value in test_oracle) within the scatter of short runs; invariants of the list layout hold."""
    import sqmc_amd
    from sqmc_amd import host as H
    h = H.ChemHost(c2_walk_fcidump(), 8, 4, "d2h")
    es = {}
    for psit in (True, False):
        vals = []
        for seed in ((1346, 5634, 6635, 4361), (2726, 5165, 6543, 6524), (911, 2202, 3303, 4405)):
            gw = H.GpuWalk(h, 20000, w_begin=100.0, hf_to_psit=psit, seed=seed)
            gw.pc.n_equil = 1500
            gw.run(1500, keep_stats=False)
            _, tot = gw.run(2500, keep_stats=False)
            vals.append(tot[3] / tot[2])
            if psit:
                w = gw.g.download_walkers()
                n_ct = len(gw.setup.ct_up)
                assert np.array_equal(w["up"][:n_ct], gw.setup.ct_up) and np.array_equal(w["dn"][:n_ct], gw.setup.ct_dn)
                assert set(np.unique(w["imp_distance"][:n_ct]).tolist()) <= {0, -2} and w["imp_distance"][n_ct:].min() >= 1
                u, d = w["up"][n_ct:], w["dn"][n_ct:]
                assert np.all((u[1:] > u[:-1]) | ((u[1:] == u[:-1]) & (d[1:] > d[:-1])))
                assert np.count_nonzero(w["imp_distance"] == 0) == len(gw.setup.imp_up)
            gw.close()
        es[psit] = np.array(vals)
    mean = {k: v.mean() for k, v in es.items()}
    err = {k: max(v.std(ddof=1) / np.sqrt(len(v)), 3e-4) for k, v in es.items()}
    assert abs(mean[True] - mean[False]) < 4 * np.hypot(err[True], err[False]), (es, mean, err)
    assert abs(mean[True] - (-75.72854)) < 4 * err[True] + 2e-3, (es, mean, err)      # + the initiator bias at this population


def c2_walk_fcidump():
    from conftest import FCIDUMP
    return FCIDUMP


def test_psit_counter_trajectory_past_2_20_slots(oracle, c2_walk, c2_psit):
    """long lists: only the spawns are sorted and merged into the (key'-ordered) residents, three slots per thread in k_anneal<3, 1>.  Ten
    steps from 10^6 walkers' worth of weight on the first state: every step but the first has more than 2^20 sorted slots."""
    s, q = c2_psit
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 1.0e6, 1, mwalk=8000000)
    pc = oracle.PopControl(s.tau, s.e_trial0, 1000000)
    nb = []
    for it in range(10):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        prm = pc.params()
        st, oc = ow.step(prm)
        assert st == 0
        og = g.step(prm)
        assert og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15] and _sums_close(og, oc), (it, og, oc)
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]; nb.append(int(og[7]))
    assert max(nb) > (1 << 20), nb
    msg = _same_walkers(g.download_walkers(), ow.walkers(), len(s.ct_up))
    assert not msg, msg
    g.close(); ow.close()


def test_psit_unpacked_keys_and_annihilate_door(oracle, c2_walk, c2_psit, monkeypatch):
    """the two-array key layout (wide keys) carries the C(T)-first key as well, and sqmc_gpu_annihilate -- the caller's own spawns through
    the same tail -- folds a hand-made list (spawns onto C(T) determinants with zero and non-zero weight, onto the first state's
    neighbours, onto new determinants in pairs that cancel) exactly as the oracle's three-segment merge does"""
    import sqmc_amd
    s, q = c2_psit
    monkeypatch.setenv("SQMC_FORCE_UNPACKED", "1")
    ow, g, w_abs = _pair(oracle, c2_walk, s, q, 100.0, 1)
    pc = oracle.PopControl(s.tau, s.e_trial0, 4000)
    _lockstep(oracle, ow, g, pc, w_abs, 30, len(s.ct_up), check_every=15)
    g.close(); ow.close()


def test_psit_refuses_what_the_reference_assumes(oracle, c2_walk, c2_psit):
    import sqmc_amd
    s, q = c2_psit
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=300000)
    g.set_projector(q.prj_counts, q.prj_indices, q.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    ix = (q.loc_psit + 1).copy()
    bad = ix.copy(); bad[0] = 2                                  # Psi_T does not begin with C(T)'s first determinant
    with pytest.raises(sqmc_amd.SqmcGpuError):
        g.set_hf_to_psit(bad, q.cdet, q.diag_elems)
    with pytest.raises(sqmc_amd.SqmcGpuError):
        g.set_hf_to_psit(ix[::-1].copy(), q.cdet, q.diag_elems)  # not in label order
    g.set_hf_to_psit(ix, q.cdet, q.diag_elems)
    wk = oracle.initial_walkers_psit(s, q, 50.0)
    short = {k: v[:-1] for k, v in wk.items()}                   # not all of C(T)
    with pytest.raises(sqmc_amd.SqmcGpuError):
        g.upload_walkers(short)
    wrong = {k: v.copy() for k, v in wk.items()}
    wrong["imp_distance"][5] = 1                                 # a C(T) determinant flagged as a stochastic one
    with pytest.raises(sqmc_amd.SqmcGpuError):
        g.upload_walkers(wrong)
    g.upload_walkers(wk)
    prm = oracle.PopControl(s.tau, s.e_trial0, 4000).params()
    prm["semistochastic"] = 0
    with pytest.raises(sqmc_amd.SqmcGpuError):
        g.step(prm)                                              # the variant is semistochastic by definition
    g.close()
