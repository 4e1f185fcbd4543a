"""CPU checks of the oracle's restatement of the step variant hf_to_psit = .true. (oracle/sqmc_oracle_psit.c).
The reference holds no fixture for it -- tests/golden/README_hf_to_psit.md says what stands in: a frozen trajectory, invariants of the
list layout, the agreement of the two sum orders, what the literal text of merge_my_original_with_spawned3 does, and the physics."""
import ctypes as C
import json
import os
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = [1346, 5634, 6635, 4361]


@pytest.fixture(scope="module")
def small(oracle, c2_walk):
    """20-determinant Psi_T (rediagonalised), 100-determinant deterministic space: C(T) has 2 10^4 determinants, a step costs 2 ms"""
    s = oracle.setup_walk(c2_walk, 20, 100, 0.1, rediagonalize=True)
    return s, oracle.psit_setup(c2_walk, s)


def _run(oracle, sysm, s, q, nsteps, mode=1, quirks=0, sum_order=1, w_begin=50.0, target=2000, seed=SEED, psit=True, n_equil=10**9, each=None):
    wk = oracle.initial_walkers_psit(s, q, w_begin) if psit else oracle.initial_walkers(s, w_begin)
    ow = oracle.OracleWalk(sysm, s, wk, 400000, list(seed), rng_mode=mode, psit=q if psit else None, quirks=quirks, sum_order=sum_order)
    pc = oracle.PopControl(s.tau, s.e_trial0, target, n_equil_steps=n_equil)
    w_abs = float(np.abs(wk["wt"]).sum())
    outs = []
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0: ow.scale_projector(r)
        st, out = ow.step(pc.params())
        if st != 0:
            break
        r = pc.post_step(out)
        if r != 1.0: ow.scale_projector(r)
        w_abs = out[1]
        outs.append(out.copy())
        if each is not None and each(it, ow):
            break
    w = ow.walkers()
    n_out = ow.n_outside_ct() if psit else 0
    ow.close()
    return np.array(outs), w, n_out


def test_ordered_sum_is_the_64_ary_tree(oracle):
    L = oracle.lib()
    L.orc_ordered_sum.restype = C.c_double
    L.orc_ordered_sum.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    rng = np.random.default_rng(5)

    def tree(x):
        x = list(x)
        while len(x) > 1:
            nxt = []
            for b in range(0, len(x), 64):
                s = x[b]
                for v in x[b + 1:b + 64]:
                    s = s + v
                nxt.append(s)
            x = nxt
        return x[0]
    for n in (1, 2, 63, 64, 65, 127, 4096, 4097, 76900, 262145):
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
        a = x.copy(); t = L.orc_ordered_sum(a.ctypes.data_as(C.c_void_p), n, 1)
        assert t == tree(x.tolist())
        a = x.copy(); lr = L.orc_ordered_sum(a.ctypes.data_as(C.c_void_p), n, 0)
        acc = x[0]
        for v in x[1:].tolist():
            acc = acc + v
        assert lr == acc
        assert abs(t - lr) <= 1e-12 * np.abs(x).sum()


def test_psit_fixture_reproduces(oracle, c2_walk):
    """eight steps of either RNG discipline from the committed start reproduce the committed sums, determinants and RNG state
    (written by tests/golden/make_golden.py from this oracle: guards the restatement against accidental change)"""
    gold = json.load(open(os.path.join(GOLD, "psit_c2_8steps.json")))
    s = oracle.setup_walk(c2_walk, 100, 1000, 0.1, coeffs="pt1")
    q = oracle.psit_setup(c2_walk, s)
    for mode in (0, 1):
        gm = gold["modes"][str(mode)]
        wk = oracle.initial_walkers_psit(s, q, gold["w_abs_gen_begin"])
        ow = oracle.OracleWalk(c2_walk, s, wk, 400000, gold["seed"], rng_mode=mode, psit=q)
        pc = oracle.PopControl(s.tau, s.e_trial0, gold["w_target"])
        w_abs = float(np.abs(wk["wt"]).sum())
        for k in range(8):
            r = pc.pre_step(w_abs)
            if r != 1.0: ow.scale_projector(r)
            st, out = ow.step(pc.params())
            assert st == 0
            assert [float(x).hex() for x in out] == gm["steps"][k], (mode, k)
            r = pc.post_step(out)
            if r != 1.0: ow.scale_projector(r)
            w_abs = out[1]
        w = ow.walkers()
        assert ow.rng_state() == gm["rng_after"] and ow.n_outside_ct() == gm["n_outside_ct"]
        assert int(np.bitwise_xor.reduce(w["up"] * np.uint64(0x9E3779B97F4A7C15) + w["dn"])) == gm["det_checksum"]
        assert float(np.sum(w["wt"] * np.arange(1, len(w["wt"]) + 1) % 7.0)).hex() == gm["wt_checksum"]
        ow.close()


def test_psit_layout_invariants_every_step(oracle, c2_walk, small):
    """C(T) keeps its determinants and flags of kind, the segment outside it is strictly ordered and disjoint from C(T), nothing
    outside carries a flag of the deterministic space or of C(T), the first state is never spawned onto stochastically"""
    s, q = small
    n_ct = len(s.ct_up)
    ct = set(zip(s.ct_up.tolist(), s.ct_dn.tolist()))
    seen = {"discard": 0}

    def each(it, ow):
        w = ow.walkers()
        assert np.array_equal(w["up"][:n_ct], s.ct_up) and np.array_equal(w["dn"][:n_ct], s.ct_dn)
        assert np.array_equal(w["imp_distance"][:n_ct] == 0, q.in_imp) and set(np.unique(w["imp_distance"][:n_ct]).tolist()) <= {0, -2}
        u, d = w["up"][n_ct:], w["dn"][n_ct:]
        assert len(u) == ow.n_outside_ct()
        assert np.all((u[1:] > u[:-1]) | ((u[1:] == u[:-1]) & (d[1:] > d[:-1])))
        assert not (set(zip(u.tolist(), d.tolist())) & ct)
        assert len(u) == 0 or (w["imp_distance"][n_ct:].min() >= 1 and w["initiator"][n_ct:].max() <= 2 and w["initiator"][n_ct:].min() >= 1)
        assert np.all(np.abs(w["wt"][n_ct:]) > 0)
        return False
    for mode in (0, 1):
        outs, w, n_out = _run(oracle, c2_walk, s, q, 80, mode=mode, each=each)
        assert len(outs) == 80 and n_out > 300
        assert np.all(outs[:, 7] >= outs[:, 5])                        # the merge only removes
        assert w["initiator"][0] == 3 and w["imp_distance"][0] == 0 and w["wt"][0] >= 1.0


def test_psit_tree_and_left_to_right_sums_agree(oracle, c2_walk, small):
    s, q = small
    for mode in (0, 1):
        a, wa, na = _run(oracle, c2_walk, s, q, 60, mode=mode, sum_order=0)
        b, wb, nb = _run(oracle, c2_walk, s, q, 60, mode=mode, sum_order=1)
        assert na == nb and np.array_equal(wa["up"], wb["up"]) and np.array_equal(wa["dn"], wb["dn"])
        assert np.array_equal(wa["initiator"], wb["initiator"]) and np.array_equal(wa["imp_distance"], wb["imp_distance"])
        assert np.max(np.abs(wa["wt"] - wb["wt"])) < 1e-10 * np.max(np.abs(wa["wt"]))
        assert np.allclose(a, b, rtol=1e-10, atol=1e-10)


def test_psit_literal_splice_loses_the_order(oracle, c2_walk, small):
    """quirk bit 0 = do_walk.f90:6709-6813 as written: within a few dozen steps the segment outside C(T) is no longer a strictly
    ordered list of distinct occupied determinants (README_hf_to_psit.md, Q1)"""
    s, q = small
    n_ct = len(s.ct_up)
    broken = {}

    def each(it, ow):
        w = ow.walkers()
        u, d = w["up"][n_ct:], w["dn"][n_ct:]
        ok = np.all((u[1:] > u[:-1]) | ((u[1:] == u[:-1]) & (d[1:] > d[:-1])))
        if not ok:
            broken["step"] = it
        return not ok
    _run(oracle, c2_walk, s, q, 200, mode=1, quirks=1, each=each)
    assert "step" in broken and broken["step"] < 150, broken


def test_psit_literal_flags_leak_the_deterministic_space(oracle, c2_walk, small):
    """quirk bit 1 = no imp_distance -1 -> 1: determinants outside C(T) end up flagged -1 and their children's determinants 0, the flag of
    the deterministic space (README_hf_to_psit.md, Q2)"""
    s, q = small
    n_ct = len(s.ct_up)
    outs, w, n_out = _run(oracle, c2_walk, s, q, 60, mode=1, quirks=2)
    flags = set(np.unique(w["imp_distance"][n_ct:]).tolist())
    assert -1 in flags, flags
    assert np.count_nonzero(w["imp_distance"] == 0) > len(s.imp_up) or 0 in flags


def test_psit_setup_refuses_what_the_reference_assumes(oracle, c2_walk, small):
    s, q = small
    import copy
    s2 = copy.copy(s)
    s2.imp_up = np.append(s.imp_up, np.uint64(0b1111 << 20)); s2.imp_dn = np.append(s.imp_dn, np.uint64(0b1111 << 20))      # a determinant C(T) does not hold
    with pytest.raises(ValueError):
        oracle.psit_setup(c2_walk, s2)
    s3 = copy.copy(s)
    s3.psi_up, s3.psi_dn, s3.psi_c = s.psi_up[1:], s.psi_dn[1:], s.psi_c[1:]            # Psi_T without the first determinant of C(T)
    o = np.lexsort((s.psi_dn, s.psi_up))
    if o[0] != 0:
        s3.psi_up, s3.psi_dn, s3.psi_c = np.delete(s.psi_up, o[0]), np.delete(s.psi_dn, o[0]), np.delete(s.psi_c, o[0])
    with pytest.raises(ValueError):
        oracle.psit_setup(c2_walk, s3)


def test_psit_energy_agrees_with_the_untransformed_walk(oracle, c2_walk, small):
    """the transformed and the untransformed projector have the same dominant eigenvector: projected energies of short walks agree
    within their scatter, and sit at the near-FCI total of this geometry (-75.72854 Ha) up to the initiator bias of 3000 walkers"""
    s, q = small
    e = {True: [], False: []}
    for psit in (True, False):
        for seed in ([1346, 5634, 6635, 4361], [2726, 5165, 6543, 6524], [77, 1234, 2345, 3457]):
            outs, _, _ = _run(oracle, c2_walk, s, q, 1600, mode=1, target=3000, seed=seed, psit=psit, n_equil=500)
            e[psit].append(outs[500:, 3].sum() / outs[500:, 2].sum())
    m = {k: float(np.mean(v)) for k, v in e.items()}
    sd = {k: max(float(np.std(v, ddof=1)) / np.sqrt(len(v)), 1e-3) for k, v in e.items()}
    assert abs(m[True] - m[False]) < 4 * np.hypot(sd[True], sd[False]), (e, m, sd)
    assert abs(m[True] + 75.72854) < 0.012, (e, m)
