"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical
inputs.  Integer/bit results must be identical; fp64 results are compared bit for bit where
the GPU keeps the reference's operation order (matrix elements, spawn weights, deterministic
projection) and to 1e-12 relative where it uses tree reductions (estimator sums)."""
import ctypes as C
import os
import numpy as np
import pytest

from conftest import gpu_ctx_from_oracle

pytestmark = pytest.mark.gpu
SEED = [1346, 5634, 6635, 4361]


def _sums_close(og, oc):
    """The step's 16 sums, GPU (fixed-shape tree reductions) against the oracle, which adds in the reference's serial
    order: a serial sum of N terms carries up to N/2 ulp of its running value (sum w^2 at 10^6 walkers is one huge term
    -- the HF weight squared -- plus 4 10^5 small ones, each rounded against it: 1.4e-11 relative observed), so the
    tolerance is 1e-11 + N eps relative, plus 2e-12 of the sum of ABSOLUTE values for the signed sums that nearly cancel."""
    og, oc = np.asarray(og), np.asarray(oc)
    n = max(float(oc[7]), 1.0)
    scale = np.full(16, abs(oc[1]))                                  # sums of w
    scale[[2, 3, 11, 12]] = max(abs(oc[12]), abs(oc[3]))            # sums of e_den w / e_num w
    scale[[8]] = abs(oc[8]); scale[[9, 10, 13]] = np.abs(oc[[9, 10, 13]])
    return bool(np.all(np.abs(og - oc) <= (1e-11 + 2.3e-16 * n) * np.abs(oc) + 2e-12 * scale))


def _random_dets(rng, norb, nel, n):
    out = np.zeros(n, np.uint64)
    for i in range(n):
        occ = rng.choice(norb, nel, replace=False)
        out[i] = np.uint64(sum(1 << int(o) for o in occ))
    return out


def _excite(rng, det, norb, k):
    occ = [o for o in range(norb) if (int(det) >> o) & 1]
    emp = [o for o in range(norb) if not (int(det) >> o) & 1]
    a = rng.choice(occ, k, replace=False); b = rng.choice(emp, k, replace=False)
    d = int(det)
    for x in a: d &= ~(1 << int(x))
    for x in b: d |= (1 << int(x))
    return np.uint64(d)


@pytest.mark.parametrize("which", ["walk", "hci"])
def test_hamiltonian_batch_bit_exact(oracle, c2_walk, c2_hci, which):
    sysm = c2_walk if which == "walk" else c2_hci
    g = gpu_ctx_from_oracle(sysm)
    rng = np.random.default_rng(7)
    n = 4000
    iu, id_ = _random_dets(rng, 26, 4, n), _random_dets(rng, 26, 4, n)
    ju, jd = iu.copy(), id_.copy()
    for i in range(n):
        m = i % 6
        if m == 1: ju[i] = _excite(rng, iu[i], 26, 1)
        elif m == 2: jd[i] = _excite(rng, id_[i], 26, 1)
        elif m == 3: ju[i] = _excite(rng, iu[i], 26, 2)
        elif m == 4: ju[i] = _excite(rng, iu[i], 26, 1); jd[i] = _excite(rng, id_[i], 26, 1)
        elif m == 5: jd[i] = _excite(rng, id_[i], 26, 2)
    if which == "hci":   # time-reversal representatives, incl. up == dn
        for i in range(0, n, 7): id_[i] = iu[i]
        sw = iu > id_; iu[sw], id_[sw] = id_[sw], iu[sw]
        sw = ju > jd; ju[sw], jd[sw] = jd[sw], ju[sw]
    h_gpu = g.hamiltonian_batch(iu, id_, ju, jd)
    h_cpu = np.array([sysm.ham(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(iu, id_, ju, jd)])
    g.close()
    assert np.count_nonzero(h_cpu) > n // 5
    assert np.array_equal(h_gpu, h_cpu)          # bit for bit


def test_proposal_kernel_bit_exact(oracle, c2_walk):
    """10^4 off_diagonal_move_chem calls from fixed rannyu states: det_j, weight_j and the RNG
    state after the call are identical to the oracle's."""
    L = oracle.lib()
    g = gpu_ctx_from_oracle(c2_walk)
    rng = np.random.default_rng(11)
    n = 10000
    up, dn = _random_dets(rng, 26, 4, n), _random_dets(rng, 26, 4, n)
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32)
    seeds[:, 3] |= 1
    tau = 0.0053
    ju, jd, wj, sa = g.propose_batch(tau, up, dn, seeds)
    g.close()
    r = oracle.Rng()
    a, b, w, nd = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    nz = 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        L.orc_off_diagonal_move_chem(c2_walk.h, C.byref(r), tau, int(up[i]), int(dn[i]), C.byref(a), C.byref(b), C.byref(w), C.byref(nd))
        assert w.value == wj[i], (i, w.value, wj[i])
        if w.value != 0.0:
            nz += 1
            assert (a.value, b.value) == (int(ju[i]), int(jd[i]))
        assert [r.l[k] for k in range(4)] == list(sa[i])
    assert nz > n // 2


def test_spmv_matches_oracle(oracle, c2_setup):
    import sqmc_amd
    s = c2_setup
    x = np.random.default_rng(3).standard_normal(len(s.prj_counts))
    y_cpu = oracle.spmv_sym_upper(s.prj_counts, s.prj_indices, s.prj_values, x)
    plan = sqmc_amd.SpmvPlan(s.prj_counts, s.prj_indices, s.prj_values)
    y_gpu = plan.apply(x)
    plan.close()
    assert np.allclose(y_gpu, y_cpu, rtol=1e-12, atol=1e-14)
    # linearity (size-independent property)
    plan = sqmc_amd.SpmvPlan(s.prj_counts, s.prj_indices, s.prj_values)
    x2 = np.random.default_rng(4).standard_normal(len(x))
    assert np.allclose(plan.apply(2 * x - 3 * x2), 2 * plan.apply(x) - 3 * plan.apply(x2), rtol=1e-11, atol=1e-13)
    plan.close()


def test_device_davidson_matches_oracle_and_lapack(oracle, c2_setup):
    """sqmc_gpu_davidson (basis resident on the device) against the oracle's restatement of davidson_sparse and against LAPACK on the
    dense matrix: the deterministic-space Hamiltonian of C2 (1000 determinants; the projector is -tau H), one and two states, with
    and without start vectors; and a matrix with a degenerate lowest pair where the start vector decides which state is tracked."""
    import sqmc_amd
    s = c2_setup
    counts, idx = s.prj_counts, s.prj_indices
    val = s.prj_values / (-s.tau)
    n = len(counts)
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    diag = val[starts]
    A = np.zeros((n, n)); r = np.repeat(np.arange(n), counts)
    A[r, idx - 1] = val; A[idx - 1, r] = val
    wl, Xl = np.linalg.eigh(A)
    plan = sqmc_amd.SpmvPlan(counts, idx, val)
    try:
        for k in (1, 2):
            ev, X, nmv = plan.davidson(diag, k=k)
            wo, Xo = oracle.davidson_sparse(counts, idx, val, k)
            assert np.allclose(ev, wo[:k], rtol=0, atol=1e-9) and 0 < nmv < 400
            # the unit start vectors live in the totally symmetric sector: the states found are the lowest ones LAPACK finds there
            for q in range(k):
                j = int(np.argmin(np.abs(wl - ev[q])))
                assert abs(wl[j] - ev[q]) < 1e-8
                assert min(np.abs(X[:, q] - Xl[:, j]).max(), np.abs(X[:, q] + Xl[:, j]).max()) < 1e-4
                assert min(np.abs(X[:, q] - Xo[:, q]).max(), np.abs(X[:, q] + Xo[:, q]).max()) < 1e-4
                assert abs(np.dot(X[:, q], X[:, q]) - 1.0) < 1e-12
                assert np.abs(plan.apply(X[:, q]) - ev[q] * X[:, q]).max() < 1e-4
        # a start vector: same state, fewer products
        ev1, X1, n1 = plan.davidson(diag, k=1)
        guess = X1[:, 0] + 1e-3 * np.random.default_rng(1).standard_normal(n)
        ev2, X2, n2 = plan.davidson(diag, k=1, v0=guess.reshape(-1, 1))
        assert abs(ev2[0] - ev1[0]) < 1e-9 and n2 < n1
    finally:
        plan.close()
    # run-to-run identical bits (fixed reduction trees)
    plan = sqmc_amd.SpmvPlan(counts, idx, val)
    a = plan.davidson(diag, k=2); b = plan.davidson(diag, k=2)
    plan.close()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    # two decoupled blocks (the second one the first shifted up by 0.5): a start inside a block stays there -- from the second block
    # the iteration finds THAT block's lowest state, not the matrix's
    rng = np.random.default_rng(7)
    m = 40
    B = rng.standard_normal((m, m)); B = B + B.T + np.diag(np.arange(m) * 3.0)
    Z = np.zeros((m, m)); D = np.block([[B, Z], [Z, B + 0.5 * np.eye(m)]])
    c2, i2, v2 = [], [], []
    for i in range(2 * m):
        cols = [i] + [j for j in range(i) if D[i, j] != 0.0]
        c2.append(len(cols)); i2 += [j + 1 for j in cols]; v2 += [D[i, j] for j in cols]
    plan = sqmc_amd.SpmvPlan(np.array(c2), np.array(i2), np.array(v2))
    try:
        for half in (0, 1):
            v0 = np.zeros((2 * m, 1)); v0[half * m, 0] = 1.0
            ev, X, _ = plan.davidson(np.diag(D).copy(), k=1, v0=v0)
            assert abs(ev[0] - (np.linalg.eigvalsh(B)[0] + 0.5 * half)) < 1e-9
            assert np.abs(X[(1 - half) * m:(2 - half) * m, 0]).max() == 0.0      # never leaves its block
    finally:
        plan.close()


@pytest.mark.parametrize("which,eps", [("hci", 2e-4), ("hci", 5e-3), ("walk", 1e-3)])
def test_hci_connections_match_oracle(oracle, c2_walk, c2_hci, which, eps):
    sysm = c2_walk if which == "walk" else c2_hci
    g = gpu_ctx_from_oracle(sysm)
    r, s_, a, pi, pc = sysm.hb_tables()
    g.set_hb_tables(r, s_, a, pi, pc, sysm.s.max_double)
    # reference list: HF + some of its strongest connections, with mixed-size coefficients
    cu, cd, el = sysm.important_connected(sysm.hf_up, sysm.hf_dn, 1e-2)
    keys = sorted(set(zip(cu.tolist(), cd.tolist())))[:40]
    ref_up = np.array([k[0] for k in keys], np.uint64); ref_dn = np.array([k[1] for k in keys], np.uint64)
    coef = np.linspace(1.0, 0.02, len(keys)) * np.where(np.arange(len(keys)) % 3 == 0, -1, 1)
    gu, gd, gnum, gden = g.hci_connections(ref_up, ref_dn, coef, eps)
    g.close()
    acc, den = {}, {}
    for u, d, c in zip(ref_up.tolist(), ref_dn.tolist(), coef.tolist()):
        xu, xd, xe = sysm.important_connected(u, d, eps / abs(c))
        for k, (p, q, h) in enumerate(zip(xu.tolist(), xd.tolist(), xe.tolist())):
            acc[(p, q)] = acc.get((p, q), 0.0) + h * c
            den[(p, q)] = den.get((p, q), 0.0) + (c if k == 0 else 0.0)
    ks = sorted(acc)
    assert len(ks) == len(gu)
    assert [k[0] for k in ks] == gu.tolist() and [k[1] for k in ks] == gd.tolist()    # sorted, unique, identical set
    assert np.allclose(gnum, [acc[k] for k in ks], rtol=1e-12, atol=1e-15)
    assert np.allclose(gden, [den[k] for k in ks], rtol=0, atol=0)


@pytest.mark.parametrize("which", ["hci", "walk"])
def test_hci_connections_active_space_masks(oracle, c2_walk, c2_hci, which):
    """The core / virtual masks of find_important_connected_dets_chem (chemistry.f90:6840-6846, 6926-6947, 7087-7108; built by
    hci.f90:150-182 from n_var_e_up/dn and n_var_orbs): generator restricted to the active space (mode 1) and to its
    complement (mode 2), GPU against oracle; the two modes partition what the unmasked generator finds.  Active space here:
    the lowest up/dn orbital frozen (core), the last 8 orbitals virtual."""
    sysm = c2_walk if which == "walk" else c2_hci
    g = gpu_ctx_from_oracle(sysm)
    r, s_, a, pi, pc = sysm.hb_tables()
    g.set_hb_tables(r, s_, a, pi, pc, sysm.s.max_double)
    norb = 26
    core = 1                                   # orbital 1
    virt = ((1 << norb) - 1) ^ ((1 << (norb - 8)) - 1)
    cu, cd, el = sysm.important_connected(sysm.hf_up, sysm.hf_dn, 1e-2)
    keys = sorted(set(zip(cu.tolist(), cd.tolist())))[:30]
    ref_up = np.array([k[0] for k in keys], np.uint64); ref_dn = np.array([k[1] for k in keys], np.uint64)
    coef = np.linspace(1.0, 0.05, len(keys)) * np.where(np.arange(len(keys)) % 2 == 0, -1, 1)
    eps = 3e-4
    got = {}
    for mode in (0, 1, 2):
        g.hci_set_active_space(core, core, virt, virt, mode)
        sysm.set_active_space(core, core, virt, virt, mode)
        gu, gd, gnum, gden = g.hci_connections(ref_up, ref_dn, coef, eps)
        acc = {}
        for u, d, c in zip(ref_up.tolist(), ref_dn.tolist(), coef.tolist()):
            xu, xd, xe = sysm.important_connected(u, d, eps / abs(c))
            for p, q, h in zip(xu.tolist(), xd.tolist(), xe.tolist()):
                acc[(p, q)] = acc.get((p, q), 0.0) + h * c
        ks = sorted(acc)
        assert [k[0] for k in ks] == gu.tolist() and [k[1] for k in ks] == gd.tolist(), mode
        assert np.allclose(gnum, [acc[k] for k in ks], rtol=1e-12, atol=1e-15)
        got[mode] = dict(zip(zip(gu.tolist(), gd.tolist()), gnum.tolist()))
    sysm.set_active_space(0, 0, 0, 0, 0)
    g.close()
    refs = set(keys)
    inside = set(got[1]) - refs; outside = set(got[2]) - refs; everything = set(got[0]) - refs
    assert inside and outside and not (inside & outside) and (inside | outside) == everything
    for k in everything:                       # a connection's sum splits by SOURCE: each (reference, connection) pair goes to exactly one mode
        assert abs(got[0][k] - (got[1].get(k, 0.0) + got[2].get(k, 0.0))) < 1e-12


def _run_pair(oracle, sysm, setup, rng_mode, nsteps, w_begin, w_target, mwalk=400000, n_equil=10**9, e_trial=-75.72, chained=False):
    g = gpu_ctx_from_oracle(sysm, rng_mode=rng_mode, seed=SEED, mwalk=mwalk)
    g.set_projector(setup.prj_counts, setup.prj_indices, setup.prj_values)
    g.set_ct_table(setup.ct_up, setup.ct_dn, setup.ct_num, setup.ct_den)
    wk = oracle.initial_walkers(setup, w_begin)
    g.upload_walkers(wk)
    if chained: g.set_chained_runs(True)
    ow = oracle.OracleWalk(sysm, setup, wk, mwalk, SEED, rng_mode=rng_mode)
    pc = oracle.PopControl(setup.tau, e_trial, w_target, n_equil_steps=n_equil)
    w_abs = np.abs(wk["wt"]).sum()
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        prm = pc.params()
        st, out_c = ow.step(prm)
        assert st == 0
        out_g = g.step(prm)
        # integer bookkeeping identical; sums to tree-reduction tolerance
        for k in (5, 7, 15):
            assert out_g[k] == out_c[k], (it, k, out_g[k], out_c[k])
        assert _sums_close(out_g, out_c), (it, [(k, out_g[k], out_c[k]) for k in range(16) if out_g[k] != out_c[k]])
        r = pc.post_step(out_c)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = out_c[1]
    wg, wc = g.download_walkers(), ow.walkers()
    rng_g, rng_c = g.rng_state(), ow.rng_state()
    _run_pair.tail_stats = g.tail_stats()
    g.close(); ow.close()
    return wg, wc, rng_g, rng_c, out_g, out_c


def _cached_hii_agree(wg, wc):
    cached = wc["matrix_elements"] < 1e50
    early = wg["matrix_elements"][~cached]
    return np.array_equal(wg["matrix_elements"][cached], wc["matrix_elements"][cached]) and bool(np.all((early > 1e50) | (np.abs(early) < 1e3)))


def test_walk_replay_trajectory_bit_exact(oracle, c2_walk, c2_setup):
    """60 steps with the reference's single rannyu stream: after the last step the walker list
    (dets, weights, initiator, imp_distance, cached H_ii / e_loc) and the RNG state are
    identical to the oracle's."""
    wg, wc, rng_g, rng_c, og, oc = _run_pair(oracle, c2_walk, c2_setup, 0, 60, 10, 2000)
    assert rng_g == rng_c
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert np.array_equal(wg["matrix_elements"], wc["matrix_elements"])
    assert len(wg["up"]) > 1200


def test_walk_counter_trajectory_bit_exact(oracle, c2_walk, c2_setup):
    """Production RNG discipline at a larger population (grown for 150 steps)."""
    wg, wc, _, _, og, oc = _run_pair(oracle, c2_walk, c2_setup, 1, 150, 50, 20000)
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert len(wg["up"]) > 3000


def test_walk_counter_trajectory_bit_exact_at_bench_size(oracle, c2_walk, c2_setup):
    """BASELINE.json configs[1] at its own size: 10^5 walkers' worth of weight from the first step on, so that within a few
    steps the lists are those of the bench (> 10^5 occupied determinants, > 2.5 10^5 sorted slots: hundreds of sort, scan and
    annihilation tiles, long runs of equal determinants on the heavy ones).  Walkers, weights and flags bit for bit
    against the oracle after 40 steps (integer bookkeeping and sums after every one)."""
    wg, wc, _, _, og, oc = _run_pair(oracle, c2_walk, c2_setup, 1, 40, 100000, 100000, mwalk=1000000)
    assert int(og[5]) > 100000 and int(og[7]) > 200000
    if os.environ.get("SQMC_BUCKET_FORCE_RETRY"):
        assert _run_pair.tail_stats[1] >= 10                                         # the rollback really ran
    elif not any(os.environ.get(k) for k in ("SQMC_FORCE_UNPACKED", "SQMC_MERGE_SORT_MIN", "SQMC_ANNEAL_ITEMS")) and os.environ.get("SQMC_BUCKET") != "0":
        # this size is what the short-list tail is for; while all the weight still sits on 10^3 determinants the key ranges are too
        # uneven for it and the radix tail runs instead
        assert _run_pair.tail_stats[0] >= 5 and _run_pair.tail_stats[1] <= 8
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    # cached H_ii: the same value wherever the oracle holds one.  The short-list tail computes the H_ii of a determinant in the
    # step that creates it, the reference (and the radix tail) in the step after (1e51 until then): a determinant the oracle
    # still has the sentinel for may carry its value already -- every weight of the next step depends on it being right.
    assert _cached_hii_agree(wg, wc)


@pytest.mark.parametrize("system", ["c2", "c2_wide", "heg", "heatbath"])
def test_two_kernel_annihilation_on_short_lists(oracle, c2_walk, c2_setup, heg14, c2_10e, monkeypatch, system):
    """The annihilation of long pipelined lists (k_anneal<3, 0, 1>: fold, round, compaction inside the tile; k_anneal_split_scan;
    k_anneal_place: estimator sums, row table, next gate and child offsets) forced onto short lists (SQMC_ANNEAL_SPLIT_MIN, radix tail),
    where the oracle follows every step: C2 (packed 28-bit keys: only while the process has not yet fixed its choice of tail -- run
    alone; c2_wide forces the two-array key layout, which always takes the radix tail), the electron gas (wide keys: the slot index in its own array, the gate
    fused as well) and the fast_heatbath proposal (two slots per child, 33-bit keys).  Chained steps; walkers bit for bit."""
    import sqmc_amd
    from conftest import gpu_ctx_heg
    monkeypatch.setenv("SQMC_ANNEAL_SPLIT_MIN", "1")
    monkeypatch.setenv("SQMC_BUCKET", "0")
    hb = None
    if system == "c2_wide": monkeypatch.setenv("SQMC_FORCE_UNPACKED", "1")
    if system in ("c2", "c2_wide"):
        sysm, su, g = c2_walk, c2_setup, gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=400000)
    elif system == "heg":
        sysm = heg14; su = oracle.setup_walk_heg(heg14, 300, 0.1)
        g = gpu_ctx_heg(heg14, rng_mode=sqmc_amd.RNG_COUNTER, seed=SEED, mwalk=400000)
    else:
        sysm = c2_10e; hb = oracle.HeatBath(sysm); su = oracle.setup_walk(sysm, 100, 1000, 0.1)
        g = gpu_ctx_from_oracle(sysm, rng_mode=1, seed=SEED, mwalk=400000)
        g.set_heatbath_tables(hb.fortran_arrays())
    g.set_projector(su.prj_counts, su.prj_indices, su.prj_values)
    g.set_ct_table(su.ct_up, su.ct_dn, su.ct_num, su.ct_den)
    wk = oracle.initial_walkers(su, 100)
    g.upload_walkers(wk)
    g.set_chained_runs(True)
    ow = oracle.OracleWalk(sysm, su, wk, 400000, SEED, rng_mode=1, heatbath=hb)
    pc = oracle.PopControl(su.tau, su.e_trial0, {"c2": 1500, "c2_wide": 1500, "heg": 250, "heatbath": 600}[system])
    w_abs = float(np.abs(wk["wt"]).sum())
    try:
        for it in range({"c2": 160, "c2_wide": 160, "heg": 200, "heatbath": 220}[system]):
            r = pc.pre_step(w_abs)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            st, oc = ow.step(pc.params())
            og = g.step(pc.params())
            assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (system, it, og, oc)
            assert _sums_close(og, oc), (system, it)
            r = pc.post_step(oc)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            w_abs = oc[1]
        wg, wc = g.download_walkers(), ow.walkers()
    finally:
        g.close(); ow.close()
        if hb is not None: hb.close()
    assert pc.reached == 2
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), (system, k)
    assert _cached_hii_agree(wg, wc)


@pytest.mark.parametrize("chained", [False, True])
def test_walk_counter_trajectory_bit_exact_past_2_20_slots(oracle, c2_walk, c2_setup, chained):
    """The variants long lists switch to -- 8-bit radix passes over the spawns only and a merge with the walkers, which are in
    order already; 3 slots per thread in the annihilation kernel; the large scan tiles -- start at 2^20 sorted slots.  Twelve
    steps from 10^6 walkers' worth of weight put every step but the first there: bit for bit against the oracle.  chained: the steps past
    the target population are pipelined, and their annihilation is the two-kernel form (k_anneal<3, 0, 1>, k_anneal_split_scan,
    k_anneal_place: no look-back, the next gate and child offsets written as the walkers are placed)."""
    wg, wc, _, _, og, oc = _run_pair(oracle, c2_walk, c2_setup, 1, 12, 1000000, 1000000, mwalk=8000000, chained=chained)
    assert int(og[7]) > (1 << 20)
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    if chained: assert _cached_hii_agree(wg, wc)
    else: assert np.array_equal(wg["matrix_elements"], wc["matrix_elements"])


@pytest.mark.parametrize("w_target,nsteps", [(200000, 120), (800000, 140)])
def test_walk_invariants_large(oracle, c2_walk, c2_setup, w_target, nsteps):
    """Size-independent properties at populations the oracle is not run at (the larger one is past
    the switch to the large-input sort and scan variants: > 2^20 sorted elements per step): walkers
    stay sorted and unique, det-space walkers are all present, weights of stochastic walkers are
    never below min_wt, sums reported by the step equal sums recomputed from the download."""
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=8 * w_target + 100000)
    s = c2_setup
    g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    wk = oracle.initial_walkers(s, 1000 if w_target <= 200000 else w_target)
    g.upload_walkers(wk)
    pc = oracle.PopControl(s.tau, -75.72, w_target, n_equil_steps=10**9)
    w_abs = np.abs(wk["wt"]).sum()
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0: g.scale_projector(r)
        prm = pc.params()
        out = g.step(prm)
        r = pc.post_step(out)
        if r != 1.0: g.scale_projector(r)
        w_abs = out[1]
    w = g.download_walkers()
    g.close()
    rfi_used = prm["reweight_factor_inv"]                     # the factor of the step that produced these weights
    key = (w["up"].astype(object) << 26) | w["dn"].astype(object)
    assert all(key[i] < key[i + 1] for i in range(len(key) - 1))
    assert (w["imp_distance"] == 0).sum() == len(s.prj_counts)
    sto = w["imp_distance"] >= 1
    assert np.all(np.abs(w["wt"][sto]) >= 0.5 * rfi_used * (1 - 1e-12))
    assert len(w["up"]) == int(out[5]) and int(out[7]) > (1 << 20 if w_target > 500000 else 0)
    assert np.isclose(np.abs(w["wt"]).sum(), out[1], rtol=1e-12)
    assert np.isclose(w["wt"].sum(), out[0], rtol=1e-10, atol=1e-8)
    assert np.isclose((w["e_num"] * w["wt"]).sum(), out[3], rtol=1e-10)
    assert -75.9 < out[3] / out[2] < -75.5


def test_gpu_reproduces_golden_walk_fixture(oracle, c2_walk):
    """The committed five-step fixture (tests/golden/walk_c2_5steps.json), REPLAY stream."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "walk_c2_5steps.json")))
    setup = oracle.setup_walk(c2_walk, 100, 1000, 0.1, coeffs="pt1")
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=0, seed=gold["seed"], mwalk=200000)
    g.set_projector(setup.prj_counts, setup.prj_indices, setup.prj_values)
    g.set_ct_table(setup.ct_up, setup.ct_dn, setup.ct_num, setup.ct_den)
    wk = oracle.initial_walkers(setup, gold["w_abs_gen_begin"])
    g.upload_walkers(wk)
    pc = oracle.PopControl(setup.tau, gold["e_trial"], gold["w_target"])
    w_abs = float(np.abs(wk["wt"]).sum())
    for k in range(5):
        r = pc.pre_step(w_abs)
        if r != 1.0: g.scale_projector(r)
        out = g.step(pc.params())
        ref = np.array([float.fromhex(x) for x in gold["steps"][k]])
        assert out[5] == ref[5] and out[7] == ref[7] and out[15] == ref[15]
        assert np.allclose(out, ref, rtol=1e-11, atol=1e-11)
        r = pc.post_step(ref)
        if r != 1.0: g.scale_projector(r)
        w_abs = ref[1]
    w = g.download_walkers()
    assert g.rng_state() == gold["rng_after"]
    assert int(np.bitwise_xor.reduce(w["up"] * np.uint64(0x9E3779B97F4A7C15) + w["dn"])) == gold["det_checksum"]
    g.close()


def test_host_tables_and_sparse_ham_match_oracle(oracle, c2_walk, c2_setup):
    """The product's own input handling (sqmc_amd/host.py) builds the same tables as the
    oracle; the GPU all-pairs Hamiltonian builder returns the oracle's matrix bit for bit."""
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    assert (h.hf_up, h.hf_dn) == (c2_walk.hf_up, c2_walk.hf_dn)
    assert np.array_equal(h.integrals, c2_walk.integrals()) and np.array_equal(h.combine_2, c2_walk.combine_2())
    assert np.array_equal(h.orbsym, c2_walk.orbsym())
    g = h.gpu()
    hb = h.hb_tables(g)
    r, s_, a, pi, pc = c2_walk.hb_tables()
    assert np.array_equal(hb[0], r) and np.array_equal(hb[1], s_) and np.array_equal(hb[2], a)
    assert np.array_equal(hb[3], pi) and np.array_equal(hb[4], pc) and hb[5] == c2_walk.s.max_double
    rc, ix, vl = g.build_sparse_ham(c2_setup.imp_up, c2_setup.imp_dn)
    oc, oi, ov = c2_walk.build_sparse_ham(c2_setup.imp_up, c2_setup.imp_dn)
    assert np.array_equal(rc, oc) and np.array_equal(ix, oi) and np.array_equal(vl, ov)
    g.set_hb_tables(*hb)
    ws = H.setup_walk(h, g)
    assert len(ws.psi_up) == len(c2_setup.psi_up) and len(ws.imp_up) == len(c2_setup.imp_up)
    assert abs(ws.e_var - c2_setup.e_var) < 1e-9 and abs(ws.tau - c2_setup.tau) < 1e-15
    # C(T) from the GPU generator holds every determinant with a non-zero numerator of the oracle's
    oct_ = {(int(a_), int(b_)): (n_, d_) for a_, b_, n_, d_ in zip(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)}
    for a_, b_, n_, d_ in zip(ws.ct_up[::37], ws.ct_dn[::37], ws.ct_num[::37], ws.ct_den[::37]):
        on, od = oct_[(int(a_), int(b_))]
        assert abs(on - n_) < 1e-9 and abs(od - d_) < 1e-9
    g.close()


def test_hci_time_sym_hamiltonian_builder(oracle, c2_hci):
    """time-reversal-symmetrised matrix (the HCI deck's conventions) from the GPU builder."""
    g = gpu_ctx_from_oracle(c2_hci)
    cu, cd, _ = c2_hci.important_connected(c2_hci.hf_up, c2_hci.hf_dn, 2e-3)
    keys = sorted(set(zip(cu.tolist(), cd.tolist())))
    up = np.array([k[0] for k in keys], np.uint64); dn = np.array([k[1] for k in keys], np.uint64)
    rc, ix, vl = g.build_sparse_ham(up, dn)
    oc, oi, ov = c2_hci.build_sparse_ham(up, dn)
    g.close()
    assert np.array_equal(rc, oc) and np.array_equal(ix, oi) and np.array_equal(vl, ov)
    w, _ = oracle.lowest_eigs(rc, ix, vl, k=1)
    assert abs(w[0] - (-75.654492433)) < 5e-9       # HCI iteration 1 energy of the reference-pinned oracle run


def _ham_builder_worker(outdir, time_sym, allpairs):
    import os, sys
    if allpairs:
        os.environ["SQMC_HAM_ALLPAIRS"] = "1"           # read once per process
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=bool(time_sym), z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 1e-3, eps_sched=(2e-3, 2e-3), n_states=1, max_iters=3)
    order = H.sort_dets(up, dn)
    rc, ix, vl = g.build_sparse_ham(up[order], dn[order])
    g.close()
    np.savez(os.path.join(outdir, "ham_%d_%d.npz" % (time_sym, allpairs)), up=up[order], dn=dn[order], rc=rc, ix=ix, vl=vl)


@pytest.mark.parametrize("time_sym", [0, 1])
def test_string_group_hamiltonian_builder_equals_all_pairs(tmp_path, time_sym):
    """generate_sparse_ham_chem_upper_triangular two ways on a ~10^4-determinant HCI space: candidates from the alpha/beta
    string groups (hbuild_kernels.h, the default) against every pair tested (k_build_ham, SQMC_HAM_ALLPAIRS=1): the same rows,
    the same column order, the same bits -- with and without time-reversal symmetrisation (whose second source, the
    spin-flipped partner, finds some pairs twice)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for allpairs in (0, 1):
        pr = ctx.Process(target=_ham_builder_worker, args=(str(tmp_path), time_sym, allpairs))
        pr.start(); pr.join(600)
        assert pr.exitcode == 0
    a = np.load(str(tmp_path / ("ham_%d_0.npz" % time_sym))); b = np.load(str(tmp_path / ("ham_%d_1.npz" % time_sym)))
    assert len(a["up"]) > 3000 and np.array_equal(a["up"], b["up"]) and np.array_equal(a["dn"], b["dn"])
    assert np.array_equal(a["rc"], b["rc"]) and np.array_equal(a["ix"], b["ix"]) and np.array_equal(a["vl"], b["vl"])


def test_fortran_host_example():
    """The Fortran iso_c_binding host (sqmc_amd/fortran/example_spmv.f90) runs against the library."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(__file__)), "sqmc_amd", "fortran", "example_spmv")
    if not os.path.exists(exe):
        pytest.skip("Fortran example not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "fortran host: OK" in out.stdout, out.stdout + out.stderr


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_run_loop_equals_stepwise_population_control(oracle, c2_walk, c2_setup):
    """sqmc_gpu_run (population control inside the library) walks the same trajectory as
    sqmc_gpu_step driven by the host-side PopControl."""
    from sqmc_amd import host as H
    res = []
    for mode in ("step", "run"):
        g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=400000)
        g.set_projector(c2_setup.prj_counts, c2_setup.prj_indices, c2_setup.prj_values)
        g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
        wk = oracle.initial_walkers(c2_setup, 100)
        g.upload_walkers(wk)
        pc = H.PopControl(c2_setup.tau, -75.72, 3000, n_equil_steps=40)
        w_abs = float(np.abs(wk["wt"]).sum())
        if mode == "step":
            for _ in range(80):
                r = pc.pre_step(w_abs)
                if r != 1.0: g.scale_projector(r)
                out = g.step(pc.params())
                r = pc.post_step(out)
                if r != 1.0: g.scale_projector(r)
                w_abs = out[1]
        else:
            cpc = pc.to_c(w_abs)
            stats, totals = g.run(cpc, 80)
            out = stats[-1]
            assert np.allclose(totals, stats.sum(axis=0))
        res.append((g.download_walkers(), out.copy()))
        g.close()
    (wa, oa), (wb, ob) = res
    # the run loop pipelines its steps once the target population is reached (the annihilation kernel does the next
    # step's gate, the scan carries the final sums): same arithmetic, so the same bits as the unpipelined single steps
    assert np.array_equal(wa["up"], wb["up"]) and np.array_equal(wa["dn"], wb["dn"])
    assert np.array_equal(wa["wt"], wb["wt"])
    assert np.array_equal(oa, ob)


def test_hci_variational_matches_reference_run():
    """BASELINE.json configs[4]: HCI on C2 cc-pVDZ, eps1=1e-4 (schedule 2x2e-4), one state, the
    shipped deck's conventions (time_sym, z=+1, hf_symmetry=1), everything heavy on the GPU.
    The reference's own run recorded 694 -> 47038 -> 118626 -> 126386 -> 126708 determinants and
    E_var = -75.727563003; with the reference's Davidson iteration restated
    on the host the counts agree exactly."""
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 1e-4, eps_sched=(2e-4, 2e-4), n_states=1)
    g.close()
    assert hist == [1, 694, 47038, 118626, 126386, 126708]         # the reference's sequence, determinant for determinant
    assert abs(e[0] - (-75.727563003)) < 2e-9
    assert abs(np.dot(w[:, 0], w[:, 0]) - 1.0) < 1e-9


def test_hci_two_states_match_reference_run():
    """The deck exactly as shipped (C2_v2z_curve/r1.24253/i_1sigma_g: eps_var 1e-3 after two
    iterations at 2e-3, TWO states, time_sym): the reference's run went 650 -> 3767 -> 11787 ->
    12705 -> 12776 determinants with E_var(1) = -75.719473642, E_var(2) = -75.631097209
    (BASELINE.md section 2).  Needs the reference's own Davidson iteration: the second root is
    tracked from a unit vector on the second determinant of the list."""
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 1e-3, eps_sched=(2e-3, 2e-3), n_states=2)
    g.close()
    assert hist == [1, 650, 3767, 11787, 12705, 12776]
    assert abs(e[0] - (-75.719473642)) < 2e-9 and abs(e[1] - (-75.631097209)) < 2e-9
    assert abs(np.dot(w[:, 0], w[:, 1])) < 1e-8


def test_time_sym_proposals_bit_exact(oracle, c2_hci):
    """off_diagonal_move_chem with time_sym=.true., z=+1: second pathway through the time-reversed
    determinant in matrix element and generation probability, representative swap."""
    L = oracle.lib()
    g = gpu_ctx_from_oracle(c2_hci)
    rng = np.random.default_rng(21)
    n = 6000
    up, dn = _random_dets(rng, 26, 4, n), _random_dets(rng, 26, 4, n)
    up[::9] = dn[::9]                                   # closed-shell representatives too
    sw = up > dn; up[sw], dn[sw] = dn[sw], up[sw]
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32); seeds[:, 3] |= 1
    tau = 0.0053
    ju, jd, wj, sa = g.propose_batch(tau, up, dn, seeds)
    g.close()
    r = oracle.Rng(); a, b, w, nd = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    nz = 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        L.orc_off_diagonal_move_chem(c2_hci.h, C.byref(r), tau, int(up[i]), int(dn[i]), C.byref(a), C.byref(b), C.byref(w), C.byref(nd))
        assert w.value == wj[i], (i, w.value, wj[i])
        if w.value != 0.0:
            nz += 1
            assert (a.value, b.value) == (int(ju[i]), int(jd[i])) and a.value <= b.value
    assert nz > n // 3


@pytest.mark.parametrize("rng_mode,nsteps", [(0, 40), (1, 100)])
def test_time_sym_walk_trajectory_bit_exact(oracle, c2_hci, c2_setup_ts, rng_mode, nsteps):
    wg, wc, rng_g, rng_c, og, oc = _run_pair(oracle, c2_hci, c2_setup_ts, rng_mode, nsteps, 20, 5000)
    if rng_mode == 0:
        assert rng_g == rng_c
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert np.all(wg["up"] <= wg["dn"]) and len(wg["up"]) > 1100
    assert -75.85 < og[3] / og[2] < -75.5


def test_heg_matrix_elements_and_proposals_bit_exact(oracle, heg14):
    from conftest import gpu_ctx_heg
    L = oracle.lib()
    g = gpu_ctx_heg(heg14)
    rng = np.random.default_rng(31)
    cu, cd, _ = heg14.connected(heg14.hf_up, heg14.hf_dn, with_elems=False)
    # pairs: every connection of HF, plus connections of connections, plus unrelated pairs
    iu = np.concatenate((np.full(len(cu), heg14.hf_up, np.uint64), cu[1:200], _random_dets(rng, 19, 7, 300)))
    id_ = np.concatenate((np.full(len(cu), heg14.hf_dn, np.uint64), cd[1:200], _random_dets(rng, 19, 7, 300)))
    ju = np.concatenate((cu, cu[2:201], _random_dets(rng, 19, 7, 300)))
    jd = np.concatenate((cd, cd[2:201], _random_dets(rng, 19, 7, 300)))
    h_gpu = g.hamiltonian_batch(iu, id_, ju, jd)
    h_cpu = np.array([heg14.ham(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(iu, id_, ju, jd)])
    assert np.array_equal(h_gpu, h_cpu) and np.count_nonzero(h_cpu) > len(cu) // 2
    n = 8000
    up, dn = _random_dets(rng, 19, 7, n), _random_dets(rng, 19, 7, n)
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32); seeds[:, 3] |= 1
    tau = 0.0012
    pju, pjd, wj, sa = g.propose_batch(tau, up, dn, seeds)
    g.close()
    r = oracle.Rng(); a, b, w, nd = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    nz = 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        L.orc_off_diagonal_move_heg(heg14.h, C.byref(r), tau, int(up[i]), int(dn[i]), C.byref(a), C.byref(b), C.byref(w), C.byref(nd))
        assert w.value == wj[i], (i, w.value, wj[i])
        assert [r.l[k] for k in range(4)] == list(sa[i])
        if w.value != 0.0:
            nz += 1
            assert (a.value, b.value) == (int(pju[i]), int(pjd[i]))
    assert nz > n // 10


def test_chained_runs_walk_the_same_trajectory():
    """sqmc_gpu_set_chained_runs: a block-structured host (one sqmc_gpu_run per block) keeps the step pipeline primed across
    its calls -- the last step of a call enqueues the head of the first step of the next.  The walk must be the one an
    unchained host walks, bit for bit; a pending head must be forgotten when something else comes next (a download, a
    projector rescale, other step parameters), and switching chaining off must leave nothing behind."""
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    def fresh():
        return H.GpuWalk(hst, 2e4, seed=SEED, mwalk=400000)
    a = fresh(); a.run(340, keep_stats=False); wa = a.g.download_walkers(); a.close()
    b = fresh(); b.run(300, keep_stats=False)
    assert b.pc.reached == 2
    b.g.set_chained_runs(True)
    sb = [b.run(n, keep_stats=True)[0] for n in (7, 1, 13, 19)]
    mid = b.g.download_walkers()                       # a pending head is forgotten here ...
    b.g.set_chained_runs(False)
    wb = b.g.download_walkers()
    assert all(np.array_equal(mid[k], wb[k]) for k in ("up", "dn", "wt"))
    for k in ("up", "dn", "wt", "initiator", "imp_distance"):
        assert np.array_equal(wa[k], wb[k]), k
    # the last step of `a` computed the H_ii of the determinants it created inside its tail kernel; `b`'s left them to spare blocks
    # of the k_spawn of the head that was pending at the download: the same cached values, bit for bit
    assert np.array_equal(wa["matrix_elements"], wb["matrix_elements"])
    assert int((wb["matrix_elements"] < 1e50).sum()) > len(wb["up"]) // 2
    # ... and when the caller comes back with another tau: the step must run as if nothing had been enqueued
    b.g.set_chained_runs(True)
    b.run(5, keep_stats=False)
    c = fresh(); c.run(345, keep_stats=False)
    for w_ in (b, c):
        w_.pc.tau_sav *= 0.5; w_.pc.tau = w_.pc.tau_prev = w_.pc.tau_sav
        w_.g.scale_projector(0.5)
    b.run(6, keep_stats=False); c.run(6, keep_stats=False)
    b.g.set_chained_runs(False)
    wb2, wc2 = b.g.download_walkers(), c.g.download_walkers()
    b.close(); c.close()
    for k in ("up", "dn", "wt"):
        assert np.array_equal(wb2[k], wc2[k]), k
    # a host that calls sqmc_gpu_step itself (population control on its side between the steps): chained, every step past the target
    # enqueues its successor's head; the walk is the unchained one
    d, e = fresh(), fresh()
    d.run(300, keep_stats=False); e.run(300, keep_stats=False)
    d.g.set_chained_runs(True)
    od = np.array([d.step().copy() for _ in range(25)]); oe = np.array([e.step().copy() for _ in range(25)])
    d.g.set_chained_runs(False)
    wd, we = d.g.download_walkers(), e.g.download_walkers()
    d.close(); e.close()
    assert np.array_equal(od, oe)
    for k in ("up", "dn", "wt"):
        assert np.array_equal(wd[k], we[k]), k


@pytest.mark.parametrize("rng_mode,nsteps", [(0, 60), (1, 120)])
def test_heg_walk_trajectory_bit_exact(oracle, heg14, heg_setup, rng_mode, nsteps):
    """BASELINE.json configs[3] system (14-electron 3D HEG) at a size the oracle runs: the same
    step pipeline with off_diagonal_move_heg / hamiltonian_heg as the operator plugin."""
    from conftest import gpu_ctx_heg
    s = heg_setup
    g = gpu_ctx_heg(heg14, rng_mode=rng_mode, seed=SEED, mwalk=400000)
    g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    wk = oracle.initial_walkers(s, 20)
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(heg14, s, wk, 400000, SEED, rng_mode=rng_mode)
    pc = oracle.PopControl(s.tau, s.e_trial0, 3000)
    w_abs = float(np.abs(wk["wt"]).sum())
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        st, oc = ow.step(pc.params())
        og = g.step(pc.params())
        assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (it, og, oc)
        assert _sums_close(og, oc), (it, [(k, og[k], oc[k]) for k in range(16) if og[k] != oc[k]])
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
    wg, wc = g.download_walkers(), ow.walkers()
    if rng_mode == 0:
        assert g.rng_state() == ow.rng_state()
    g.close(); ow.close()
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert 58.0 < og[3] / og[2] < 58.6


def test_heg_host_setup_matches_oracle(oracle, heg14, heg_setup):
    """sqmc_amd.host.HegHost builds the same plane-wave table as the oracle and an equivalent walk set-up"""
    from sqmc_amd import host as H
    hh = H.HegHost(3, 0.5, 14, 7, 1.49)
    assert hh.norb == heg14.norb and hh.length_cell == heg14.length_cell
    assert np.array_equal(hh.k_vectors, heg14.k_vectors())
    cu, cd, _ = heg14.connected(heg14.hf_up, heg14.hf_dn, with_elems=False)
    pu, pd = hh.connected(hh.hf_up, hh.hf_dn)
    assert sorted(zip(cu.tolist(), cd.tolist())) == list(zip(pu.tolist(), pd.tolist()))
    w = H.GpuWalk(hh, 20000, w_begin=200, size_deterministic=250, n_truncate_trial_wf=1)
    s = w.setup
    assert abs(s.e_var - heg_setup.e_var) < 1e-9 and abs(s.tau - heg_setup.tau) < 1e-15 and len(s.ct_up) == len(heg_setup.ct_up)
    stats, tot = w.run(300)
    e = (stats[100:, 3] * np.sign(stats[100:, 2])).sum() / np.abs(stats[100:, 2]).sum()
    w.close()
    assert 58.26 < e < 58.29          # the reference's HCI total energy for this system: 58.27597 (o_det_ref:436)


def test_walk_deck_end_to_end(tmp_path):
    """A run_type `none` deck in the reference's grammar (tests/golden/C2_r1.24253_i_walk: the walk
    smoke test of BASELINE.md -- C2 cc-pVDZ, uniform2, semistochastic, size_deterministic 1000, Psi_T
    100 dets, target 1e4, 100-step blocks) through `python -m sqmc_amd.run`'s walk runner: set-up
    figures of the reference run (1002 deterministic dets, tau 0.005314, ~16,400 occupied and ~32,500
    pre-merge determinants), the reference's block-wise equilibration and output lines, and an energy
    that agrees with the near-FCI HCI total energy of this geometry (-75.7285) within the error bar."""
    import io, os
    from conftest import FCIDUMP
    from sqmc_amd.walk_run import parse_walk_deck, run_walk
    deck = parse_walk_deck(open(os.path.join(os.path.dirname(__file__), "golden", "C2_r1.24253_i_walk")).read())
    buf = io.StringIO()
    wal = str(tmp_path / "walkalize")
    r = run_walk(deck, FCIDUMP, out=buf, walkalize=wal)
    txt = buf.getvalue()
    assert r["n_imp"] == 1002 and abs(r["tau"] - 0.005314) < 1e-6
    assert 15800 < r["nwalk_av"] < 17000
    assert "Equilibration of everything achieved" in txt and "Energy=" in txt and "w_abs_gen_target=    10000 reached" in txt
    assert txt.count("iblk, w_perm_initiator, nwalk, w_abs, w_abs_imp=") == r["n_blocks_total"] == r["n_equil_sets"] * 2 + 4
    assert abs(r["energy"] - (-75.7285)) < max(5 * r["energy_err"], 3e-3), (r["energy"], r["energy_err"])
    rec = open(wal).read().splitlines()
    assert len(rec) == 100 * r["n_blocks_total"] + 1 and rec[-1].endswith("nstep, nblk, w_abs_gen_target, e_trial, tau")
    # restart files of the reference: dump C(T) and the deterministic space, start a second run from them
    pc, de = str(tmp_path / "psit_connections.out"), str(tmp_path / "dtm_projector.out")
    r1 = run_walk(deck, FCIDUMP, out=io.StringIO(), psit_con_out=pc, dtm_elems_out=de)
    buf2 = io.StringIO()
    r2 = run_walk(deck, FCIDUMP, out=buf2, psit_con_in=pc, dtm_elems_in=de)
    assert "Reading in the local energies" in buf2.getvalue() and "Reading in the matrix elements" in buf2.getvalue()
    assert r2["n_imp"] == 1002 and r2["tau"] == r1["tau"] and 0 < r1["n_ct"] - r2["n_ct"] < 0.2 * r1["n_ct"]     # entries with |e_num| <= 1e-10 are not written
    assert abs(r2["energy"] - r1["energy"]) < 5 * (r1["energy_err"] + r2["energy_err"]) + 1e-3


def test_survey_walk_deck_statistical_pin():
    """The one external walk observable on file: BASELINE.md section 2's reference smoke run of
    tests/golden/C2_r1.24253_i_walk_survey printed Energy= -75.71813(81) after four 100-step blocks.  Four independent seeds
    of the same deck and schedule on the GPU path: same set-up figures (1002 deterministic determinants, tau, ~16,400
    occupied).  Four blocks are less than one autocorrelation time, so single runs scatter by 2.6 mHa (eight seeds on file,
    profiles/r02_survey_walk_pin.txt: mean -75.72859, scatter 0.0026); their mean must agree with this geometry's near-FCI
    HCI+PT2 total, -75.72854, and must not be the smoke value, which lies 4 sigma of a single run above it (a run that
    measured before its population had equilibrated).  The test fails if the walk drifts to either side."""
    import io, os
    from conftest import FCIDUMP
    from sqmc_amd.walk_run import parse_walk_deck, run_walk
    text = open(os.path.join(os.path.dirname(__file__), "golden", "C2_r1.24253_i_walk_survey")).read()
    es = []
    for k in range(4):
        deck = parse_walk_deck(text)
        deck["irand_seed"][1][3] = (deck["irand_seed"][1][3] + 2 * k) % 10000
        r = run_walk(deck, FCIDUMP, out=io.StringIO())
        assert r["n_imp"] == 1002 and abs(r["tau"] - 0.005314) < 1e-6 and 15800 < r["nwalk_av"] < 17000
        assert abs(r["energy"] - (-75.72854)) < 4 * 0.0026, (k, r["energy"], r["energy_err"])
        es.append(r["energy"])
    mean = float(np.mean(es))
    assert abs(mean - (-75.72854)) < 3 * 0.0026 / 2, (mean, es)              # three standard errors of a four-run mean
    assert mean - (-75.71813) < -0.005, (mean, es)


@pytest.mark.parametrize("deckname,lo,hi", [("heg14_i_walk", 58.270, 58.280), ("hubbard4x4_i_walk", -12.5, -10.0)])
def test_walk_decks_of_the_other_systems(deckname, lo, hi):
    """Walk decks for the electron gas (the system of the reference's e2e fixtures, whose HCI total
    energy is 58.27597) and for the real-space Hubbard lattice of BASELINE.json configs[0] (exact
    ground state -13.62 t; at 1e4 walkers the initiator approximation sits well above it)."""
    import io, os
    from sqmc_amd.walk_run import parse_walk_deck, run_walk
    deck = parse_walk_deck(open(os.path.join(os.path.dirname(__file__), "golden", deckname)).read())
    buf = io.StringIO()
    r = run_walk(deck, out=buf)
    assert "Equilibration of everything achieved" in buf.getvalue() and "Energy=" in buf.getvalue()
    assert lo < r["energy"] < hi, r
    assert r["nwalk_av"] > 5000 and r["energy_err"] < 0.05


def test_hubbard_matrix_elements_and_proposals_bit_exact(oracle, hub44):
    """hamiltonian_hubbard / off_diagonal_move_hubbard (real-space Hubbard,
    BASELINE.json configs[0] lattice) on the GPU against the oracle: values, determinants and the
    rannyu state after the (rejection-sampled, variable-length) draw sequence."""
    from conftest import gpu_ctx_hub
    L = oracle.lib()
    g = gpu_ctx_hub(hub44)
    rng = np.random.default_rng(77)
    n = 6000
    up, dn = _random_dets(rng, 16, 8, n), _random_dets(rng, 16, 8, n)
    # pairs: each determinant with itself, with one of its hops, with an unrelated determinant, with a non-bond single hop
    ju, jd = up.copy(), dn.copy()
    for i in range(n):
        if i % 4 == 1:
            cu, cd, _ = hub44.connected(int(up[i]), int(dn[i]), with_elems=False)
            k = int(rng.integers(0, len(cu))); ju[i], jd[i] = cu[k], cd[k]
        elif i % 4 == 2:
            ju[i], jd[i] = _random_dets(rng, 16, 8, 1)[0], dn[i]
        elif i % 4 == 3:
            ju[i], jd[i] = _random_dets(rng, 16, 8, 1)[0], _random_dets(rng, 16, 8, 1)[0]
    h_gpu = g.hamiltonian_batch(up, dn, ju, jd)
    h_cpu = np.array([hub44.ham(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(up, dn, ju, jd)])
    assert np.array_equal(h_gpu, h_cpu) and np.count_nonzero(h_cpu) > n // 3
    for i in range(1, n, 4):      # on lattice bonds the bond test changes nothing: the reference's own routine agrees
        assert hub44.ham_unchecked(int(up[i]), int(dn[i]), int(ju[i]), int(jd[i])) == h_cpu[i]
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32); seeds[:, 3] |= 1
    tau = 0.0052
    pju, pjd, wj, sa = g.propose_batch(tau, up, dn, seeds)
    g.close()
    r = oracle.Rng(); a, b, w, nd = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    nz, long_draws = 0, 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        L.orc_off_diagonal_move_hubbard(hub44.h, C.byref(r), tau, int(up[i]), int(dn[i]), C.byref(a), C.byref(b), C.byref(w), C.byref(nd))
        assert w.value == wj[i], (i, w.value, wj[i])
        assert [r.l[k] for k in range(4)] == list(sa[i])
        long_draws += nd.value > 3
        if w.value != 0.0:
            nz += 1
            assert (a.value, b.value) == (int(pju[i]), int(pjd[i]))
    assert nz > n // 2 and long_draws > n // 10       # the rejection loop of choose_random_electron was exercised


@pytest.mark.parametrize("rng_mode,semi,nsteps,w_begin,w_target", [(0, 1, 60, 50, 4000), (1, 1, 120, 50, 4000), (0, 0, 50, 50, 4000),
                                                                  (1, 1, 30, 100000, 100000)])      # the last: ten times configs[0]'s population from step one
def test_hubbard_walk_trajectory_bit_exact(oracle, hub44, hub_setup, rng_mode, semi, nsteps, w_begin, w_target):
    """BASELINE.json configs[0] (4x4 Hubbard, U/t = 4, half filling): the step pipeline with
    off_diagonal_move_hubbard / hamiltonian_hubbard as the operator pair, semistochastic and plain
    (join_walker2) variants, against the oracle walker for walker."""
    from conftest import gpu_ctx_hub
    s = hub_setup
    mwalk = 400000 if w_target < 50000 else 1500000
    g = gpu_ctx_hub(hub44, rng_mode=rng_mode, seed=SEED, mwalk=mwalk)
    if semi:
        g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    wk = oracle.initial_walkers(s, w_begin)
    if not semi:                                   # a purely stochastic population
        wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)
        keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
        wk = {k: v[keep] for k, v in wk.items()}
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(hub44, s, wk, mwalk, SEED, rng_mode=rng_mode)
    pc = oracle.PopControl(s.tau, s.e_trial0, w_target)
    w_abs = float(np.abs(wk["wt"]).sum())
    for it in range(nsteps):
        r = pc.pre_step(w_abs)
        if r != 1.0 and semi:
            ow.scale_projector(r); g.scale_projector(r)
        st, oc = ow.step(pc.params(semistochastic=semi))
        og = g.step(pc.params(semistochastic=semi))
        assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (it, og, oc)
        assert _sums_close(og, oc), (it, [(k, og[k], oc[k]) for k in range(16) if og[k] != oc[k]])
        r = pc.post_step(oc)
        if r != 1.0 and semi:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
    wg, wc = g.download_walkers(), ow.walkers()
    if rng_mode == 0:
        assert g.rng_state() == ow.rng_state()
    g.close(); ow.close()
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert len(wg["up"]) > (300 if w_target < 50000 else 50000)


def test_hubbard_host_setup_matches_oracle_and_walk_runs(oracle, hub44, hub_setup):
    """sqmc_amd.host.HubbardHost: same lattice, same connected lists, equivalent walk set-up; a
    short walk at 1e4 target (configs[0]'s population) lands below the variational energy of the
    small set-up space and above the exact ground state (-13.6219 t for 4x4, U/t = 4)."""
    from sqmc_amd import host as H
    hh = H.HubbardHost(4, 4, True, 8, 8, 1.0, 4.0)
    assert (hh.hf_up, hh.hf_dn) == (hub44.hf_up, hub44.hf_dn)
    for site in range(1, 17):
        assert hh.nbr[site - 1] == [max(oracle.lib().orc_get_nbr(4, 4, 1, site, k), 0) for k in (1, 2, 3, 4)]
    cu, cd, _ = hub44.connected(hub44.hf_up, hub44.hf_dn, with_elems=False)
    pu, pd = hh.connected(hh.hf_up, hh.hf_dn)
    assert sorted(zip(cu.tolist(), cd.tolist())) == list(zip(pu.tolist(), pd.tolist()))
    w = H.GpuWalk(hh, 10000, w_begin=200, size_deterministic=500, n_truncate_trial_wf=20, tau_multiplier=0.5)
    s = w.setup
    assert abs(s.e_var - hub_setup.e_var) < 1e-9 and s.tau == hub_setup.tau and len(s.ct_up) == len(hub_setup.ct_up)
    assert np.array_equal(s.imp_up, hub_setup.imp_up) and np.allclose(s.ct_num, hub_setup.ct_num, atol=1e-9)
    stats, tot = w.run(1500)
    e = (stats[700:, 3] * np.sign(stats[700:, 2])).sum() / np.abs(stats[700:, 2]).sum()
    w.close()
    print("4x4 Hubbard projected energy", e, "population", stats[-1, 1], "dets", stats[-1, 5])
    assert -14.5 < e < s.e_var


def test_hci_pt2_matches_oracle_and_reference_run(oracle, c2_hci):
    """Epstein-Nesbet PT2 on the GPU path: against the oracle on a ~4k-determinant space, and
    against the reference's own run at eps1=1e-4 / eps2=1e-6 (dE_PT = -0.000979165,
    E_total = -75.728542168)."""
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 2e-3, eps_sched=(2e-3,), n_states=1, max_iters=2)
    assert 1500 < len(up) < 6000
    d_gpu, n_gpu = H.hci_pt2(h, g, up, dn, w[:, 0], float(e[0]), 2e-5)
    d_cpu, n_cpu = oracle.hci_pt2(c2_hci, up, dn, w[:, 0], float(e[0]), 2e-5)
    assert n_gpu - len(up) == n_cpu
    assert abs(d_gpu - d_cpu) < 1e-12 and -0.05 < d_gpu < -0.005
    # the connected space in 7 slices of the key range: same determinants, same sums
    d_sl, n_sl = H.hci_pt2(h, g, up, dn, w[:, 0], float(e[0]), 2e-5, n_slices=7)
    assert n_sl == n_gpu and abs(d_sl - d_gpu) < 1e-14
    # the one-call entry against the same sum assembled from the batch doors on the host
    d_doors, n_doors = H.hci_pt2_by_doors(h, g, up, dn, w[:, 0], float(e[0]), 2e-5, n_slices=3)
    assert n_doors == n_gpu and abs(d_doors - d_gpu) < 1e-14
    up, dn, w, e, hist = H.hci_variational(h, g, 1e-4, eps_sched=(2e-4, 2e-4), n_states=1)
    g.close()
    # the reference converts to the determinant basis before PT (hci.f90:648-659): 6.56 M connections there
    d, n = H.hci_pt2_determinant_basis(h, up, dn, w[:, 0], float(e[0]), 1e-6)
    assert 6.55e6 < n < 6.57e6
    assert abs(d - (-0.000979165)) < 2e-9
    assert abs(e[0] + d - (-75.728542168)) < 2e-9


def test_fortran_host_walk(tmp_path):
    """A Fortran host (sqmc_amd/fortran/example_walk.f90, iso_c_binding module only) starts the
    C2 walk from the same tables and must land on exactly the state the Python-driven run reaches:
    the boundary is the C ABI, not the Python around it."""
    import os, subprocess
    from sqmc_amd import host as H
    root = os.path.dirname(os.path.dirname(__file__))
    exe = os.path.join(root, "sqmc_amd", "fortran", "example_walk")
    if not os.path.exists(exe):
        pytest.skip("Fortran example not built")
    hst = H.ChemHost(os.path.join(root, "tests", "golden", "C2_r1.24253_FCIDUMP"), 8, 4, "d2h")
    walk = H.GpuWalk(hst, 20000, w_begin=2000, seed=SEED, mwalk=400000)
    wk = H.initial_walkers(walk.setup, 2000)
    deck = str(tmp_path / "c2.deck")
    H.dump_walk_deck(deck, hst, walk.setup, wk, 20000, walk.pc.e_trial, seed=SEED, mwalk=400000)
    out = subprocess.run([exe, deck, "60", "20"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    last = [l for l in out.stdout.splitlines() if l.startswith("fortran walk:")][0].split()
    f_steps, f_nwalk = int(last[2]), int(last[3])
    f_wabs, f_etrial, f_num, f_den = (float(x) for x in last[4:8])
    walk.run(60)
    assert f_steps == 60 and f_nwalk == walk.g.num_walkers()
    assert f_wabs == walk.w_abs and f_etrial == walk.pc.e_trial
    assert f_num == walk.pc.e_num_cum and f_den == walk.pc.e_den_cum
    walk.close()


@pytest.mark.parametrize("rng_mode,rfi,heavy", [(0, 1.0, False), (1, 1.0, False), (1, 0.93, False), (0, 1.0, True), (1, 0.93, True)])
def test_annihilate_door_matches_oracle_merge(oracle, c2_walk, c2_setup, rng_mode, rfi, heavy):
    """sqmc_gpu_annihilate against merge_sort2_up_dn + merge_original_with_spawned2 +
    reduce_my_walker of the oracle on a collision-heavy hand-made spawn list: many spawns per
    determinant in both signs, exact cancellations, spawns onto deterministic-space and onto the
    permanent-initiator determinant, spawns from the deterministic space (imp_distance -1),
    non-initiator spawns onto empty determinants, zero-weight proposals.  `heavy`: a few determinants
    collect thousands of spawns each (the runs of equal keys span several tiles of the annihilation
    kernel: wavefront-cooperative folds, in mixed signs and -- one determinant -- in one sign)."""
    rs = np.random.RandomState(1234 + rng_mode)
    main = oracle.initial_walkers(c2_setup, 300)
    n0 = len(main["up"])
    pool = rs.choice(len(c2_setup.ct_up), 6 if heavy else 80, replace=False)
    ns = 14000 if heavy else 4000
    from_main = rs.rand(ns) < (0.03 if heavy else 0.4)
    im, ip = rs.randint(0, n0, ns), pool[rs.randint(0, len(pool), ns)]
    up = np.where(from_main, main["up"][im], c2_setup.ct_up[ip]).astype(np.uint64)
    dn = np.where(from_main, main["dn"][im], c2_setup.ct_dn[ip]).astype(np.uint64)
    wt = rs.choice([-1.0, 1.0], ns) * rs.choice([0.05, 0.2, 0.25, 0.4, 0.5, 0.75, 1.0, 1.5], ns)     # few distinct values: exact cancellations happen
    wt[rs.rand(ns) < 0.05] = 0.0
    impd = rs.choice([-1, 1, 2, 3, 5], ns).astype(np.int8)
    init = np.where(impd == -1, 1, rs.randint(0, 2, ns)).astype(np.int8)
    if heavy:      # the children of one parent: thousands of equal same-sign weights onto one determinant (and a second one in the deterministic space)
        one = (~from_main) & (ip == pool[0])
        wt[one] = 0.3; impd[one] = 2; init[one] = 0
        imp_dets = np.nonzero(main["imp_distance"] == 0)[0]
        sel = rs.rand(ns) < 0.12
        up[sel], dn[sel] = main["up"][imp_dets[5]], main["dn"][imp_dets[5]]
        wt[sel] = -0.2; impd[sel] = np.where(rs.rand(int(sel.sum())) < 0.5, -1, 2); init[sel] = 1
    prm = dict(tau=c2_setup.tau, e_trial=-75.7, reweight_factor_inv=rfi, r_initiator=1.0, min_wt=0.5, always_spawn_cutoff_wt=0.5,
               initiator_power=0, initiator_min_distance=0, c_t_initiator=0, semistochastic=1, reached_w_abs_gen=2)
    # ---- oracle: the three routines in the order of do_walk.f90:2335-2473, then the reweighting of :2487
    ow = oracle.OracleWalk(c2_walk, c2_setup, main, 50000, list(SEED), rng_mode=rng_mode)
    w, nz = ow.w, np.nonzero(wt)[0]
    for k, j in enumerate(nz):
        i = n0 + k
        w.up[i], w.dn[i], w.wt[i], w.imp_distance[i], w.initiator[i] = int(up[j]), int(dn[j]), float(wt[j]), int(impd[j]), int(init[j])
        w.matrix_elements[i] = w.e_num_walker[i] = w.e_den_walker[i] = 1e51
    n = n0 + len(nz)
    p = oracle.StepParams(**prm)
    L = oracle.lib()
    L.orc_reduce_my_walker.restype = C.c_int64
    L.orc_reduce_my_walker.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_merge_sort_walkers(ow.h, n)
    n = L.orc_merge_original_with_spawned2(ow.h, n, C.byref(p))
    n = L.orc_reduce_my_walker(ow.h, n, C.byref(p))
    ow.w.nwalk = n
    ref = ow.walkers(); ow.close()
    ref["wt"] = ref["wt"] * rfi
    # ---- GPU door
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=rng_mode, seed=SEED, mwalk=50000)
    g.set_projector(c2_setup.prj_counts, c2_setup.prj_indices, c2_setup.prj_values)
    g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
    g.upload_walkers(main)
    out = g.annihilate(prm, dict(up=up, dn=dn, wt=wt, imp_distance=impd, initiator=init))
    got = g.download_walkers(); g.close()
    assert len(got["up"]) == n == int(out[5])
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        assert np.array_equal(got[k], ref[k]), k
    assert int(out[7]) == n0 + len(nz)                      # nwalk_before_merge: zero-weight proposals are no walkers
    assert n < n0 + len(nz) - 1000                          # the list really collided
    if heavy:
        assert n < n0 + 400


@pytest.mark.parametrize("trial", range(12))
def test_annihilate_door_random_parameters(oracle, c2_walk, c2_setup, trial):
    """The merge / initiator / rounding rules under randomly drawn step parameters (r_initiator incl.
    the -1 switch, initiator_power, initiator_min_distance, c_t_initiator, min_wt) and spawn lists
    whose runs range from one to several thousand records in mixed signs -- the wavefront-cooperative
    folds take their general path there.  Walkers, weights and flags equal the oracle's in both RNG disciplines."""
    rs = np.random.RandomState(900 + trial)
    rng_mode = trial % 2
    main = oracle.initial_walkers(c2_setup, 300)
    n0 = len(main["up"])
    npool = int(rs.choice([2, 5, 40, 300]))
    pool = rs.choice(len(c2_setup.ct_up), npool, replace=False)
    ns = int(rs.choice([3000, 9000]))
    from_main = rs.rand(ns) < rs.choice([0.02, 0.3])
    im, ip = rs.randint(0, n0, ns), pool[rs.randint(0, len(pool), ns)]
    up = np.where(from_main, main["up"][im], c2_setup.ct_up[ip]).astype(np.uint64)
    dn = np.where(from_main, main["dn"][im], c2_setup.ct_dn[ip]).astype(np.uint64)
    wt = rs.choice([-1.0, 1.0], ns, p=[0.3, 0.7]) * rs.choice([0.05, 0.2, 0.25, 0.4, 0.5, 0.75, 1.0, 1.5, 2.5], ns)
    wt[rs.rand(ns) < 0.04] = 0.0
    impd = rs.choice([-1, 1, 2, 3, 5, 127], ns, p=[0.12, 0.38, 0.25, 0.15, 0.08, 0.02]).astype(np.int8)      # what move_uniform2 can hand out (do_walk.f90:3703-3717)
    init = np.where(impd == -1, 1, rs.randint(0, 3, ns)).astype(np.int8)
    prm = dict(tau=c2_setup.tau, e_trial=-75.7, reweight_factor_inv=float(rs.choice([1.0, 0.97, 1.02])), r_initiator=float(rs.choice([0.5, 1.0, 2.0, -1.0])),
               min_wt=float(rs.choice([0.3, 0.5, 1.0])), always_spawn_cutoff_wt=0.5, initiator_power=int(rs.choice([0, 1, 2])),
               initiator_min_distance=int(rs.choice([0, 1, 2])), c_t_initiator=int(rs.choice([0, 1])), semistochastic=1, reached_w_abs_gen=2)
    ow = oracle.OracleWalk(c2_walk, c2_setup, main, 60000, list(SEED), rng_mode=rng_mode)
    w, nz = ow.w, np.nonzero(wt)[0]
    for k, j in enumerate(nz):
        i = n0 + k
        w.up[i], w.dn[i], w.wt[i], w.imp_distance[i], w.initiator[i] = int(up[j]), int(dn[j]), float(wt[j]), int(impd[j]), int(init[j])
        w.matrix_elements[i] = w.e_num_walker[i] = w.e_den_walker[i] = 1e51
    n = n0 + len(nz)
    p = oracle.StepParams(**prm)
    L = oracle.lib()
    L.orc_reduce_my_walker.restype = C.c_int64
    L.orc_reduce_my_walker.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_merge_original_with_spawned2.restype = C.c_int64
    L.orc_merge_sort_walkers(ow.h, n)
    n = L.orc_merge_original_with_spawned2(ow.h, n, C.byref(p))
    n = L.orc_reduce_my_walker(ow.h, n, C.byref(p))
    ow.w.nwalk = n
    ref = ow.walkers(); ow.close()
    ref["wt"] = ref["wt"] * prm["reweight_factor_inv"]
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=rng_mode, seed=SEED, mwalk=60000)
    g.set_projector(c2_setup.prj_counts, c2_setup.prj_indices, c2_setup.prj_values)
    g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
    g.upload_walkers(main)
    try:
        out = g.annihilate(prm, dict(up=up, dn=dn, wt=wt, imp_distance=impd, initiator=init))
        got = g.download_walkers()
    finally:
        g.close()
    assert len(got["up"]) == n == int(out[5]), (len(got["up"]), n, prm)
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        assert np.array_equal(got[k], ref[k]), (k, prm)


def test_annihilate_door_refuses_made_up_deterministic_flags(oracle, c2_walk, c2_setup):
    """Spawn records that claim imp_distance 0 on determinants outside the deterministic space (no move
    of the reference produces them) must come back as status 5, 'locations of my imp broken'
    (do_walk.f90:2204), not as an out-of-range write."""
    from sqmc_amd import SqmcGpuError
    main = oracle.initial_walkers(c2_setup, 300)
    rs = np.random.RandomState(5)
    pool = rs.choice(len(c2_setup.ct_up), 3000, replace=False)
    up, dn = c2_setup.ct_up[pool].astype(np.uint64), c2_setup.ct_dn[pool].astype(np.uint64)
    prm = dict(tau=c2_setup.tau, e_trial=-75.7, reweight_factor_inv=1.0, r_initiator=1.0, min_wt=0.5, always_spawn_cutoff_wt=0.5,
               initiator_power=0, initiator_min_distance=0, c_t_initiator=0, semistochastic=1, reached_w_abs_gen=2)
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=60000)
    g.set_projector(c2_setup.prj_counts, c2_setup.prj_indices, c2_setup.prj_values)
    g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
    g.upload_walkers(main)
    try:
        with pytest.raises(SqmcGpuError) as ei:
            g.annihilate(prm, dict(up=up, dn=dn, wt=np.full(3000, 1.5), imp_distance=np.zeros(3000, np.int8), initiator=np.ones(3000, np.int8)))
        assert ei.value.code == 5
    finally:
        g.close()


def test_spawn_only_sort_and_merge_is_bit_exact():
    """Large lists sort only the spawns and merge them into the walkers, which are in order already
    (`SQMC_MERGE_SORT_MIN`, default 2^20 slots).  With the threshold at 0 the trajectory, annihilation
    and sharded-invariant tests must pass unchanged: same walkers, weights and flags as the oracle."""
    import subprocess, sys
    env = dict(os.environ, SQMC_MERGE_SORT_MIN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "trajectory_bit_exact or annihilate_door or time_sym"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
def test_long_list_sort_tiles_on_short_lists_are_bit_exact():
    """Lists of 2^20 keys and more sort in tiles of 4096 keys (`rs_hist_kernel<., 16>`, `rs_scatter_kernel<., ., 16>`);
    SQMC_SORT_BIG_TILE=2 gives every list those tiles (ragged last tile, few tiles): the trajectory and annihilation-door tests
    must pass unchanged against the oracle."""
    import subprocess, sys
    env = dict(os.environ, SQMC_SORT_BIG_TILE="2", SQMC_BUCKET="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "walk_trajectory_bit_exact or annihilate_door"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
def test_radix_rows_scanned_by_blocks_is_bit_exact():
    """The per-tile digit histograms of a radix pass are scanned one wave per digit row when a row is short (<= 1,024 tiles, every
    other test) and one block per row beyond; SQMC_RS_SCAN_BLOCKS=1 forces the long-row kernel on the short rows: the trajectory and
    annihilation-door tests must pass unchanged against the oracle."""
    import subprocess, sys
    env = dict(os.environ, SQMC_RS_SCAN_BLOCKS="1", SQMC_BUCKET="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "walk_trajectory_bit_exact or annihilate_door"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("items", [3, 4])
def test_annihilation_tile_shapes_are_bit_exact(items):
    """k_anneal handles 2, 3 or 4 sorted slots per thread depending on the length of the list (2 below 2^20 slots, where
    every other test runs).  SQMC_ANNEAL_ITEMS forces the shapes of the long lists on the short ones: the trajectory,
    annihilation-door and time-reversal tests must pass unchanged against the oracle."""
    import subprocess, sys
    env = dict(os.environ, SQMC_ANNEAL_ITEMS=str(items))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "trajectory_bit_exact or annihilate_door or time_sym"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_error_statuses_match_reference_stops(oracle, c2_walk, c2_setup):
    """The reference's own `stop`s come back as status codes with its texts, on
    the same inputs for which the oracle reports them."""
    from sqmc_amd import SqmcGpuError
    main = oracle.initial_walkers(c2_setup, 300)
    n0 = len(main["up"])
    prm = dict(tau=c2_setup.tau, e_trial=-75.7, reweight_factor_inv=1.0, r_initiator=1.0, min_wt=0.5, always_spawn_cutoff_wt=0.5,
               initiator_power=0, initiator_min_distance=0, c_t_initiator=0, semistochastic=1, reached_w_abs_gen=2)

    def ctx(mwalk, ct=True, walkers=main):
        g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=mwalk)
        g.set_projector(c2_setup.prj_counts, c2_setup.prj_indices, c2_setup.prj_values)
        if ct:
            g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
        g.upload_walkers(walkers)
        return g

    # 'nwalk>MWALK' (do_walk.f90:3690): room for the walkers but not for their spawns
    g = ctx(n0 + 50)
    with pytest.raises(SqmcGpuError) as e:
        g.step(prm)
    assert e.value.code == 1 and "nwalk>MWALK" in str(e.value)
    g.close()
    ow = oracle.OracleWalk(c2_walk, c2_setup, main, n0 + 50, list(SEED), rng_mode=1)
    assert ow.step(prm)[0] == 1
    ow.close()
    # 'diagonal_factor<0 after target population has been reached' (do_walk.f90:3788)
    # (needs walkers outside the deterministic space: a few ordinary steps first, the same on both sides)
    big = dict(prm, tau=50.0)
    g = ctx(200000)
    ow = oracle.OracleWalk(c2_walk, c2_setup, main, 200000, list(SEED), rng_mode=1)
    for _ in range(5):
        g.step(prm)
        assert ow.step(prm)[0] == 0
    with pytest.raises(SqmcGpuError) as e:
        g.step(big)
    assert e.value.code == 3 and "diagonal_factor<0" in str(e.value)
    g.close()
    assert ow.step(big)[0] == 3
    ow.close()
    # 'locations of my imp broken' (do_walk.f90:2204): a deterministic-space walker is missing
    keep = np.ones(n0, bool); keep[np.nonzero(main["imp_distance"] == 0)[0][3]] = False
    broken = {k: v[keep] for k, v in main.items()}
    with pytest.raises(SqmcGpuError) as e:
        ctx(200000, walkers=broken)
    assert e.value.code == 5
    # argument errors: unsorted walkers, no C(T) table
    swapped = {k: v.copy() for k, v in main.items()}
    for k in swapped:
        swapped[k][[5, 6]] = swapped[k][[6, 5]]
    with pytest.raises(SqmcGpuError) as e:
        ctx(200000, walkers=swapped)
    assert e.value.code == -1 and "sorted" in str(e.value)
    g = ctx(200000, ct=False)
    with pytest.raises(SqmcGpuError) as e:
        g.step(prm)
    assert e.value.code == -1
    # and the context is still usable after a refused call
    g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
    out = g.step(prm)
    assert out[5] > 0
    g.close()


@pytest.mark.parametrize("r", ["1.0", "1.1", "1.2", "1.3", "1.4", "1.6", "1.8", "2.0"])
def test_binding_curve_geometries_bit_exact(oracle, r):
    """BASELINE.json configs[2]: every other point of the C2 binding curve (different orbital orders and
    symmetry labels in the FCIDUMP, stretched bonds with a multi-reference Psi_T).  Tables, Psi_T /
    deterministic space from the host path, and a counter-mode trajectory, all against the oracle."""
    import os
    from sqmc_amd import host as H
    path = os.path.join(os.path.dirname(__file__), "golden", "curve", "C2_r%s_FCIDUMP" % r)
    sysm = oracle.ChemSystem(path, 8, 4, "d2h", time_sym=False, hf_mode=0)
    setup = oracle.setup_walk(sysm, 100, 1000, 0.1)
    # host-side tables of the product path agree with the oracle's for this geometry
    hst = H.ChemHost(path, 8, 4, "d2h")
    assert np.array_equal(hst.combine_2.reshape(-1), sysm.combine_2().reshape(-1))
    assert np.array_equal(np.asarray(hst.orbsym), np.asarray(sysm.orbsym())) and np.array_equal(hst.integrals, sysm.integrals())
    g = hst.gpu(mwalk=0)
    s2 = hst.setup_walk(g, 100, 1000, 0.1)
    g.close()
    assert len(s2.psi_up) == len(setup.psi_up) and len(s2.imp_up) == len(setup.imp_up)
    assert abs(s2.e_var - setup.e_var) < 1e-9 and abs(s2.tau - setup.tau) < 1e-15 and abs(s2.e_trial0 - setup.e_trial0) < 1e-8
    wg, wc, _, _, og, oc = _run_pair(oracle, sysm, setup, 1, 60, 50, 5000, e_trial=setup.e_trial0)
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert len(wg["up"]) > 1500


@pytest.mark.parametrize("r", ["1.3", "1.6"])
def test_hci_deck_conventions_across_curve(oracle, r):
    """The shipped HCI decks say `&hf_det hf_symmetry=1 /`: the starting determinant comes from the
    reference's descent, and between r = 1.3 and 1.6 A (3 sigma_g / 1 pi_u crossing) it is NOT the
    first four orbitals of the FCIDUMP.  Host tables and the HCI iteration (reference Davidson from
    that determinant) against the oracle."""
    import os
    from sqmc_amd import host as H
    path = os.path.join(os.path.dirname(__file__), "golden", "curve", "C2_r%s_FCIDUMP" % r)
    sysm = oracle.ChemSystem(path, 8, 4, "d2h", time_sym=True, z=1, hf_mode=1, hf_symmetry=1)
    h = H.ChemHost(path, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    assert (h.hf_up, h.hf_dn) == (sysm.hf_up, sysm.hf_dn)
    assert np.array_equal(h.orbsym, sysm.orbsym()) and np.array_equal(h.combine_2, sysm.combine_2())
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 2e-3, n_states=1)
    g.close()
    ou, od, ow_, oe, ohist = oracle.hci_variational(sysm, 2e-3, n_states=1)
    assert hist == ohist and abs(e[0] - oe[0]) < 1e-9
    assert np.array_equal(up, ou) and np.array_equal(dn, od)


@pytest.mark.parametrize("rng_mode,nsteps", [(0, 50), (1, 120)])
def test_non_semistochastic_walk_trajectory_bit_exact(oracle, c2_walk, c2_setup, rng_mode, nsteps):
    """semistochastic = f: no deterministic space or projection, death/cloning on every walker,
    and join_walker2 (do_walk.f90:6990-7103) instead of the stochastic rounding.  Same trajectory
    as the oracle in both RNG disciplines."""
    wk = oracle.initial_walkers(c2_setup, 100)
    wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)   # a purely stochastic population
    keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
    wk = {k: v[keep] for k, v in wk.items()}
    g = gpu_ctx_from_oracle(c2_walk, rng_mode=rng_mode, seed=SEED, mwalk=400000)
    g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(c2_walk, c2_setup, wk, 400000, SEED, rng_mode=rng_mode)
    pc = oracle.PopControl(c2_setup.tau, -75.72, 4000)
    w_abs = np.abs(wk["wt"]).sum()
    njoin = 0
    for it in range(nsteps):
        pc.pre_step(w_abs)
        prm = pc.params(semistochastic=0)
        st, out_c = ow.step(prm)
        assert st == 0
        out_g = g.step(prm)
        for k in (5, 7, 15):
            assert out_g[k] == out_c[k], (it, k, out_g[k], out_c[k])
        assert _sums_close(out_g, out_c), (it, [(k, out_g[k], out_c[k]) for k in range(16) if out_g[k] != out_c[k]])
        pc.post_step(out_c)
        w_abs = out_c[1]
    wg, wc = g.download_walkers(), ow.walkers()
    if rng_mode == 0:
        assert g.rng_state() == ow.rng_state()          # the single rannyu stream ends in the same state
    g.close(); ow.close()
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"])
    assert len(wg["up"]) > 800
    small = (np.abs(wg["wt"]) < 0.5 * pc.rfi * (1 - 1e-12)) & (wg["initiator"] < 3)
    assert small.sum() <= 2                      # at most the unfinished last chain of each sign


def test_plain_annihilate_door_with_chains_longer_than_a_tile(oracle, c2_walk, c2_setup):
    """join_walker2 in the COUNTER discipline is parallel on the GPU (k_join_gather + k_join_par): every candidate follows the
    chain that would begin at it, the chains that really begin are found by pointer doubling, tile by tile (2048 candidates),
    an unfinished chain being carried into the next tile.  A hand-made list whose positive candidates are thousands of tiny
    weights -- chains of ~2500 members that span tiles, carriers that change across a tile boundary -- beside ordinary negative
    ones, through sqmc_gpu_annihilate with semistochastic = 0, against merge_sort2_up_dn + merge_original_with_spawned2 +
    join_walker2 of the oracle."""
    rs = np.random.RandomState(77)
    main = oracle.initial_walkers(c2_setup, 300)
    main["imp_distance"] = np.where(main["imp_distance"] == 0, 1, main["imp_distance"]).astype(np.int8)
    keep = ~((main["wt"] == 0) & (main["initiator"] < 3))
    main = {k: v[keep] for k, v in main.items()}
    n0 = len(main["up"])
    ns = 9000
    ip = rs.choice(len(c2_setup.ct_up), ns, replace=False)
    up, dn = c2_setup.ct_up[ip].astype(np.uint64), c2_setup.ct_dn[ip].astype(np.uint64)
    pos_ = rs.rand(ns) < 0.6
    wt = np.where(pos_, 2.0e-4 * (1.0 + rs.randint(0, 8, ns)), -rs.choice([0.05, 0.11, 0.2, 0.26, 0.4, 0.45], ns))
    impd = rs.choice([1, 2, 3, 5], ns).astype(np.int8)
    init = rs.randint(0, 2, ns).astype(np.int8)
    prm = dict(tau=c2_setup.tau, e_trial=-75.7, reweight_factor_inv=1.0, r_initiator=-1.0, min_wt=0.5, always_spawn_cutoff_wt=0.5,
               initiator_power=0, initiator_min_distance=0, c_t_initiator=0, semistochastic=0, reached_w_abs_gen=2)
    ow = oracle.OracleWalk(c2_walk, c2_setup, main, 50000, list(SEED), rng_mode=1)
    w = ow.w
    for k in range(ns):
        i = n0 + k
        w.up[i], w.dn[i], w.wt[i], w.imp_distance[i], w.initiator[i] = int(up[k]), int(dn[k]), float(wt[k]), int(impd[k]), int(init[k])
        w.matrix_elements[i] = w.e_num_walker[i] = w.e_den_walker[i] = 1e51
    n = n0 + ns
    p = oracle.StepParams(**prm)
    L = oracle.lib()
    L.orc_join_walker2.restype = C.c_int64
    L.orc_join_walker2.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_merge_sort_walkers(ow.h, n)
    n = L.orc_merge_original_with_spawned2(ow.h, n, C.byref(p))
    n_merged = n
    n = L.orc_join_walker2(ow.h, n, C.byref(p))
    ow.w.nwalk = n
    ref = ow.walkers(); ow.close()
    import os
    for serial in (False, True):            # SQMC_SERIAL_JOIN: the one-lane k_join walks the same chains with the same draws
        if serial:
            os.environ["SQMC_SERIAL_JOIN"] = "1"
        try:
            g = gpu_ctx_from_oracle(c2_walk, rng_mode=1, seed=SEED, mwalk=50000)
            g.set_ct_table(c2_setup.ct_up, c2_setup.ct_dn, c2_setup.ct_num, c2_setup.ct_den)
            g.upload_walkers(main)
            out = g.annihilate(prm, dict(up=up, dn=dn, wt=wt, imp_distance=impd, initiator=init))
            got = g.download_walkers(); g.close()
        finally:
            os.environ.pop("SQMC_SERIAL_JOIN", None)
        assert len(got["up"]) == n == int(out[5])
        for k in ("up", "dn", "wt", "imp_distance", "initiator"):
            assert np.array_equal(got[k], ref[k]), (serial, k)
    assert n_merged > 8000 and n < n_merged - 5000                  # thousands of joins happened
    assert (ref["wt"] > 0.5).sum() >= 1 and int(pos_.sum()) > 2 * 2048     # positive chains closed above min_wt; their members fill more than two tiles


def test_shipped_hci_deck_end_to_end(tmp_path):
    """The reference's own input deck C2_v2z_curve/r1.24253/i_1sigma_g, unchanged, through
    `python -m sqmc_amd.run`: deck grammar, hf_symmetry descent, two-state HCI, PT in the
    determinant basis.  Every number the reference's run printed: 12776
    determinants, E_var(1) = -75.719473642, PT(1) = -0.008762133, E_tot(1) = -75.728235774,
    E_var(2) = -75.631097209, E_tot(2) = -75.638806440, 3.36 M / 3.24 M PT connections."""
    import io, os
    from conftest import FCIDUMP
    from sqmc_amd import run as R
    deck = R.parse_hci_deck(open(os.path.join(os.path.dirname(__file__), "golden", "C2_r1.24253_i_1sigma_g")).read())
    assert deck["n_states"] == 2 and deck["eps_var_sched"] == [2e-3, 2e-3] and deck["hf_symmetry"] == 1 and deck["time_sym"]
    buf = io.StringIO()
    res = R.run_hci(deck, FCIDUMP, out=buf)
    txt = buf.getvalue()
    assert res["ndets"] == 12776
    (e1, d1, n1), (e2, d2, n2) = res["states"]
    assert abs(e1 - (-75.719473642)) < 2e-9 and abs(d1 - (-0.008762133)) < 2e-8 and abs(e1 + d1 - (-75.728235774)) < 2e-8
    assert abs(e2 - (-75.631097209)) < 2e-9 and abs(e2 + d2 - (-75.638806440)) < 2e-8
    assert abs(n1 - 3.36e6) < 1e4 and abs(n2 - 3.24e6) < 1e4
    # the lines C2_v2z_curve/runall greps for
    assert "Total energy(1)=" in txt and "Total energy(2)=" in txt and "Variational energy(1)=" in txt
    assert "Iteration   5 eps1=1.0E-3 ndets=    12776" in txt


def test_triplet_deck_matches_oracle(oracle):
    """The other shipped deck of the same directory, i_3pi_u: time-reversal ANTIsymmetric states
    (z = -1: determinants with up == dn drop out) of irrep 2.  No reference run is recorded for it,
    so the GPU path is held against the oracle: same starting determinant, iteration history, E_var."""
    import io, os
    from conftest import FCIDUMP
    from sqmc_amd import run as R
    deck = R.parse_hci_deck(open(os.path.join(os.path.dirname(__file__), "golden", "C2_r1.24253_i_3pi_u")).read())
    assert deck["z"] == -1 and deck["hf_symmetry"] == 2 and deck["n_states"] == 1
    sysm = oracle.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=True, z=-1, hf_mode=1, hf_symmetry=2)
    ou, od, ow_, oe, ohist = oracle.hci_variational(sysm, deck["eps_var"], eps_sched=tuple(deck["eps_var_sched"]), n_states=1)
    deck["eps_pt"] = 1e-5                                     # a lighter PT stage: it is checked elsewhere
    res = R.run_hci(deck, FCIDUMP, out=io.StringIO())
    assert res["ndets"] == len(ou) == ohist[-1]
    assert abs(res["states"][0][0] - oe[0]) < 1e-9


def test_device_resident_hamiltonian_plan_matches_host_path(oracle, c2_hci):
    """sqmc_gpu_build_spmv_plan (Hamiltonian built, symmetrised and kept on the GPU) against the
    two-step path (sqmc_gpu_build_sparse_ham -> host -> sqmc_gpu_spmv_prepare) and the oracle's
    reference loop, on a time-symmetrised HCI space of a few thousand determinants."""
    from conftest import FCIDUMP
    import sqmc_amd
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    cu, cd, _, _ = g.hci_connections([h.hf_up], [h.hf_dn], [1.0], 1e-4)
    cu, cd, _, _ = g.hci_connections(cu, cd, np.full(len(cu), 0.05), 5e-3)
    assert 1500 < len(cu) < 60000
    counts, idx, val = g.build_sparse_ham(cu, cd)
    plan_a = sqmc_amd.SpmvPlan(counts, idx, val)
    plan_b, diag, nnz = sqmc_amd.SpmvPlan.from_dets(g, cu, cd)
    starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
    assert nnz == len(val) and np.array_equal(diag, val[starts])
    rs = np.random.RandomState(5)
    for _ in range(3):
        x = rs.randn(len(cu))
        ya, yb = plan_a.apply(x), plan_b.apply(x)
        yo = oracle.spmv_sym_upper(counts, idx, val, x)
        assert np.allclose(ya, yb, rtol=1e-13, atol=1e-13) and np.allclose(yb, yo, rtol=1e-12, atol=1e-12)
        assert np.array_equal(yb, plan_b.apply(x))                 # same bits on a repeated call
    plan_a.close(); plan_b.close(); g.close()


def test_fortran_host_hci(tmp_path):
    """A Fortran host (sqmc_amd/fortran/example_hci.f90) runs the variational stage of the shipped
    two-state deck through the iso_c_binding module: connection generation, Hamiltonian + matvec plan
    and the Davidson matvec on the GPU, list bookkeeping and the Krylov problem in Fortran.  It must
    reach the reference's numbers (12776 determinants, both variational energies)."""
    import os, subprocess
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    root = os.path.dirname(os.path.dirname(__file__))
    exe = os.path.join(root, "sqmc_amd", "fortran", "example_hci")
    if not os.path.exists(exe):
        pytest.skip("Fortran example not built")
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    hb = h.hb_tables(g)
    g.close()
    deck = str(tmp_path / "c2_hci.deck")
    H.dump_hci_deck(deck, h, hb, 1e-3, eps_sched=(2e-3, 2e-3), n_states=2)
    out = subprocess.run([exe, deck, "2e-5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    last = [l for l in out.stdout.splitlines() if l.startswith("fortran hci:")][0].split()
    assert int(last[2]) == 12776
    # the PT stage through the same module (sqmc_gpu_hci_pt2) against the Python host on its own wavefunction
    pt = [l for l in out.stdout.splitlines() if l.startswith("fortran pt2:")][0].split()
    g = h.gpu(); g.set_hb_tables(*hb)
    up, dn, w, e, hist = H.hci_variational(h, g, 1e-3, eps_sched=(2e-3, 2e-3), n_states=2)
    d_py, n_py = H.hci_pt2(h, g, up, dn, w[:, 0], float(e[0]), 2e-5)
    g.close()
    assert int(pt[2]) == n_py and abs(float(pt[3]) - d_py) < 1e-8 and d_py < -1e-3
    assert abs(float(last[3]) - (-75.719473642)) < 2e-9 and abs(float(last[4]) - (-75.631097209)) < 2e-9
    its = [l for l in out.stdout.splitlines() if l.startswith("Iteration")]
    assert [int(l.split("ndets=")[1].split()[0]) for l in its] == [1, 650, 3767, 11787, 12705, 12776]


def test_heg_hci_reproduces_reference_e2e_golden_output():
    """The reference's own end-to-end fixture for the electron gas, src/e2e_tests/heg/o_det_ref (deck
    i_det: 3D, 14 electrons, r_s = 0.5, cutoff 1.49, eps_var 1e-3, eps_pt 2e-7), on the GPU path:
    1 -> 277 -> 9475 determinants (:226, :307, :379), E_var 58.282597 / 58.276906085 (:307, :434),
    501881 connected determinants in the PT stage, lowering -0.000939196, total 58.275966889
    (:431-437).  Connection generation (sqmc_gpu_hci_connections, HEG branch), Hamiltonian + matvec
    plan, Davidson matvec and PT sums all come from the GPU."""
    from sqmc_amd import host as H
    hst = H.HegHost(3, 0.5, 14, 7, 1.49)
    assert hst.norb == 19
    g = hst.gpu()
    up, dn, w, e, hist = H.hci_variational(hst, g, 1e-3, n_states=1)
    assert hist == [1, 277, 9475]
    assert abs(e[0] - 58.276906085) < 2e-9
    d, n = H.hci_pt2(hst, g, up, dn, w[:, 0], float(e[0]), 2e-7)
    # the deterministic piece of the semistochastic PT run of the same directory (deck i_st: eps_pt_big =
    # 8.192e-4; o_st_ref:873 prints it as the first number in parentheses)
    d_big, _ = H.hci_pt2(hst, g, up, dn, w[:, 0], float(e[0]), 8.1920e-4)
    d_sl, n_sl = H.hci_pt2(hst, g, up, dn, w[:, 0], float(e[0]), 2e-7, n_slices=4)       # sliced connected space: identical
    assert n_sl == n and abs(d_sl - d) < 1e-15
    g.close()
    assert abs(d_big - (-0.000199339)) < 2e-9
    assert n == 501881
    assert abs(d - (-0.000939196)) < 2e-9 and abs(e[0] + d - 58.275966889) < 2e-9
    assert abs(e[0] + d + hst.madelung_energy() - 48.051813420) < 2e-9          # 'Total energy (includ. Madelung)' :438


def test_heg_semistochastic_pt_reproduces_reference_samples():
    """The reference's second e2e fixture for the electron gas, src/e2e_tests/heg/o_st_ref (deck i_st =
    i_det + `&selected_ci n_mc=200 eps_pt_big=8.1920e-4 /`): the semistochastic PT, sample by sample.
    Same rannyu stream (seed 1 of the input line), alias tables, merging of repeats and estimator: the
    first samples print -0.000628947, -0.000488905, -0.000786277, -0.000940695, -0.000707866 (:442-454),
    the run stops after 143 samples (target error 1e-5) at -0.000729402 +- 0.000009966 on top of the
    deterministic -0.000199339, total 58.275977344 (:868-874)."""
    from sqmc_amd import host as H
    hst = H.HegHost(3, 0.5, 14, 7, 1.49)
    g = hst.gpu()
    up, dn, w, e, hist = H.hci_variational(hst, g, 1e-3, n_states=1)
    res = H.hci_pt2_stochastic(hst, g, up, dn, w[:, 0], float(e[0]), 2e-7, 8.1920e-4, 200, 1e-5, seed=(2726, 5165, 6543, 6524), max_samples=400)
    g.close()
    ref5 = [-0.000628947, -0.000488905, -0.000786277, -0.000940695, -0.000707866]
    assert all(abs(a - b) < 1.5e-9 for a, b in zip(res["samples"][:5], ref5))
    assert len(res["samples"]) == 143 and abs(res["samples"][-1] - (-0.000829319)) < 1.5e-9
    assert abs(res["pt_big"] - (-0.000199339)) < 2e-9
    assert abs(res["pt_diff"] - (-0.000729402)) < 2e-9 and abs(res["pt_diff_std_dev"] - 0.000009966) < 2e-9
    assert abs(e[0] + res["pt_big"] + res["pt_diff"] - 58.275977344) < 3e-9


@pytest.mark.parametrize("deckname", ["heg_e2e_i_det", "heg_e2e_i_st"])
def test_reference_e2e_heg_decks_run_unchanged(deckname):
    """The two input decks of the reference's own end-to-end test directory (src/e2e_tests/heg/i_det,
    i_st; copied as data), unchanged, through `python -m sqmc_amd.run`, against the golden outputs that
    sit next to them (o_det_ref, o_st_ref): the numbers e2e_check.py extracts (variational energy, PT
    lowering, its error bar) and the totals, to the printed digits instead of its 0.01 / 0.05 tolerances."""
    import io, os
    from sqmc_amd import run as R
    deck = R.parse_hci_deck(open(os.path.join(os.path.dirname(__file__), "golden", deckname)).read())
    buf = io.StringIO()
    res = R.run_hci(deck, out=buf)
    txt = buf.getvalue()
    assert res["hist"] == [1, 277, 9475] and abs(res["e_var"] - 58.276906085) < 2e-9
    assert "Variational energy=" in txt and "Second-order PT energy lowering=" in txt      # what e2e_check.py greps for
    if deckname.endswith("i_det"):
        assert res["n_connected"] == 501881
        assert abs(res["pt"] - (-0.000939196)) < 2e-9 and abs(res["e_total"] - 58.275966889) < 2e-9
        assert abs(res["e_total"] + res["madelung"] - 48.051813420) < 2e-9
    else:
        assert res["n_samples"] == 143
        assert abs(res["pt"] - (-0.000928741)) < 2e-9 and abs(res["pt_err"] - 0.000009966) < 2e-9
        assert abs(res["e_total"] - 58.275977344) < 3e-9 and abs(res["e_total"] + res["madelung"] - 48.051823875) < 3e-9


def test_semistochastic_pt_chem_agrees_with_deterministic():
    """No reference fixture exists for the chemistry variant of the semistochastic PT, so it is held
    against the deterministic PT of the same space (C2, eps_var 2e-3, determinant basis after
    time_symmetrized_to_dets): the estimate of PT(eps_pt) must agree within its own error bar (4 sigma)
    and the deterministic piece must equal hci_pt2 at eps_pt_big."""
    import copy
    from conftest import FCIDUMP
    from sqmc_amd import host as H
    h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
    g = h.gpu()
    g.set_hb_tables(*h.hb_tables(g))
    up, dn, w, e, hist = H.hci_variational(h, g, 2e-3, n_states=1)
    g.close()
    plain = copy.copy(h); plain.time_sym = False
    du, dd, dc = H.time_symmetrized_to_dets(up, dn, w[:, 0], h.z)
    gp = plain.gpu()
    gp.set_hb_tables(*plain.hb_tables(gp))
    det, _ = H.hci_pt2(plain, gp, du, dd, dc, float(e[0]), 1e-6)
    big, _ = H.hci_pt2(plain, gp, du, dd, dc, float(e[0]), 2e-4)
    r = H.hci_pt2_stochastic(plain, gp, du, dd, dc, float(e[0]), 1e-6, 2e-4, 300, 1e-4, max_samples=2000)
    gp.close()
    assert abs(r["pt_big"] - big) < 1e-14
    assert r["pt_diff_std_dev"] <= 1e-4 * 1.0001 and len(r["samples"]) >= 10
    assert abs(r["pt_big"] + r["pt_diff"] - det) < 4 * r["pt_diff_std_dev"]


@pytest.mark.gpu
def test_gate_fused_into_annihilation_is_bit_exact():
    """Pipelined steps of sqmc_gpu_run: the annihilation kernel computes the next step's gate (keys, child counts, child
    weights) and the child-offset scan carries the final sums.  600 steps that reach the target population, with the fusion
    and with SQMC_NO_GATE_FUSION=1 (gate kernel of its own): every per-step sum and the final walkers must be the same bits."""
    import subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuse_ab.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "first differing step: []" in r.stdout and "walkers equal: True" in r.stdout


# ------------------------------------------------------------------ BASELINE.json configs[3]: HEG r_s = 1.0, 57 plane waves (56-bit keys)
def _heg_pair(oracle, hsys, s, rng_mode, nsteps, w_begin, w_target, mwalk):
    """oracle and GPU side by side on an electron-gas walk; integer bookkeeping equal after every step"""
    from conftest import gpu_ctx_heg
    g = gpu_ctx_heg(hsys, rng_mode=rng_mode, seed=SEED, mwalk=mwalk)
    g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    wk = oracle.initial_walkers(s, w_begin)
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(hsys, s, wk, mwalk, SEED, rng_mode=rng_mode)
    pc = oracle.PopControl(s.tau, s.e_trial0, w_target)
    w_abs = float(np.abs(wk["wt"]).sum())
    try:
        for it in range(nsteps):
            r = pc.pre_step(w_abs)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            st, oc = ow.step(pc.params())
            og = g.step(pc.params())
            assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (it, og, oc)
            assert _sums_close(og, oc), (it, [(k, og[k], oc[k]) for k in range(16) if og[k] != oc[k]])
            r = pc.post_step(oc)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            w_abs = oc[1]
        wg, wc = g.download_walkers(), ow.walkers()
        rng = (g.rng_state(), ow.rng_state())
    finally:
        g.close(); ow.close()
    return wg, wc, rng, og


def test_heg57_keys_are_unpacked(heg57):
    """the point of this system: C(57,7)^2 needs 56 key bits"""
    from math import comb
    assert heg57.norb == 57 and (comb(57, 7) ** 2).bit_length() == 56


def test_heg57_matrix_elements_and_proposals_bit_exact(oracle, heg57):
    """hamiltonian_heg (heg.f90:845-1011) and off_diagonal_move_heg (heg.f90:1344-1598) on the 57-plane-wave basis: every
    connection of HF, connections of connections, unrelated pairs; 10^4 proposals (det_j, weight, RNG state)."""
    from conftest import gpu_ctx_heg
    L = oracle.lib()
    g = gpu_ctx_heg(heg57)
    rng = np.random.default_rng(57)
    cu, cd, _ = heg57.connected(heg57.hf_up, heg57.hf_dn, with_elems=False)
    nr = 600
    iu = np.concatenate((np.full(len(cu), heg57.hf_up, np.uint64), cu[1:400], _random_dets(rng, 57, 7, nr)))
    id_ = np.concatenate((np.full(len(cu), heg57.hf_dn, np.uint64), cd[1:400], _random_dets(rng, 57, 7, nr)))
    ju = np.concatenate((cu, cu[2:401], _random_dets(rng, 57, 7, nr)))
    jd = np.concatenate((cd, cd[2:401], _random_dets(rng, 57, 7, nr)))
    # second-generation pairs: a connection of HF against the connections of another one
    c2u, c2d, _ = heg57.connected(int(cu[5]), int(cd[5]), with_elems=False)
    iu = np.concatenate((iu, np.full(len(c2u), cu[5], np.uint64))); id_ = np.concatenate((id_, np.full(len(c2u), cd[5], np.uint64)))
    ju = np.concatenate((ju, c2u)); jd = np.concatenate((jd, c2d))
    h_gpu = g.hamiltonian_batch(iu, id_, ju, jd)
    h_cpu = np.array([heg57.ham(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(iu, id_, ju, jd)])
    assert np.array_equal(h_gpu, h_cpu) and np.count_nonzero(h_cpu) > len(cu)
    n = 10000
    up = np.concatenate((_random_dets(rng, 57, 7, n // 2), np.resize(cu, n - n // 2)))
    dn = np.concatenate((_random_dets(rng, 57, 7, n // 2), np.resize(cd, n - n // 2)))
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32); seeds[:, 3] |= 1
    tau = 0.0013
    pju, pjd, wj, sa = g.propose_batch(tau, up, dn, seeds)
    g.close()
    r = oracle.Rng(); a, b, w, nd = C.c_uint64(), C.c_uint64(), C.c_double(), C.c_int()
    nz = 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        L.orc_off_diagonal_move_heg(heg57.h, C.byref(r), tau, int(up[i]), int(dn[i]), C.byref(a), C.byref(b), C.byref(w), C.byref(nd))
        assert w.value == wj[i], (i, w.value, wj[i])
        assert [r.l[k] for k in range(4)] == list(sa[i])
        if w.value != 0.0:
            nz += 1
            assert (a.value, b.value) == (int(pju[i]), int(pjd[i]))
    assert nz > n // 20


@pytest.mark.parametrize("rng_mode,nsteps", [(0, 110), (1, 160)])
def test_heg57_walk_trajectory_bit_exact(oracle, heg57, heg57_setup, rng_mode, nsteps):
    """configs[3]'s system through the whole step with two-array sort records (separate key / index arrays in the
    radix passes, the unpacked branches of the annihilation kernel, no gate fusion): REPLAY and COUNTER trajectories
    equal the oracle's walker for walker, bit for bit."""
    wg, wc, rng, og = _heg_pair(oracle, heg57, heg57_setup, rng_mode, nsteps, 20, 6000, 400000)
    if rng_mode == 0:
        assert rng[0] == rng[1]
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"]) and _cached_hii_agree(wg, wc)
    assert len(wg["up"]) > 1500 and int(wg["up"].max()) >= (1 << 32)           # determinants whose up string alone is wider than a packed key
    assert 13.0 < og[3] / og[2] < 13.7                                          # HF 13.60, correlated ground state below it


def test_heg57_walk_past_2_20_slots_bit_exact(oracle, heg57, heg57_setup):
    """Past 2^20 sorted slots the step sorts the spawns only (7 passes of 8 bits over 56-bit keys, payload array beside
    them) and merges them into the ordered walkers with the two-array merge-path kernel; 3 slots per thread in the
    annihilation kernel.  Eight steps from 10^6 walkers' worth of weight (every step but the first proposes more than 2^20
    children, about half of which find no momentum partner and sort behind the walkers), bit for bit against the oracle."""
    wg, wc, _, og = _heg_pair(oracle, heg57, heg57_setup, 1, 8, 1000000, 1000000, 8000000)
    assert int(og[15]) > (1 << 20)
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"]) and _cached_hii_agree(wg, wc)


@pytest.mark.parametrize("rng_mode,heavy", [(0, False), (1, True)])
def test_heg57_annihilate_door_matches_oracle_merge(oracle, heg57, heg57_setup, rng_mode, heavy):
    """sqmc_gpu_annihilate on the 57-plane-wave gas (unpacked sort records): a collision-heavy hand-made spawn list
    against merge_sort2_up_dn + merge_original_with_spawned2 + reduce_my_walker of the oracle.  `heavy`: a few
    determinants collect thousands of spawns (runs that span tiles: wavefront-cooperative folds through get_perm)."""
    s = heg57_setup
    rs = np.random.RandomState(570 + rng_mode)
    main = oracle.initial_walkers(s, 300)
    n0 = len(main["up"])
    pool = rs.choice(len(s.ct_up), 6 if heavy else 80, replace=False)
    ns = 14000 if heavy else 4000
    from_main = rs.rand(ns) < (0.03 if heavy else 0.4)
    im, ip = rs.randint(0, n0, ns), pool[rs.randint(0, len(pool), ns)]
    up = np.where(from_main, main["up"][im], s.ct_up[ip]).astype(np.uint64)
    dn = np.where(from_main, main["dn"][im], s.ct_dn[ip]).astype(np.uint64)
    wt = rs.choice([-1.0, 1.0], ns) * rs.choice([0.05, 0.2, 0.25, 0.4, 0.5, 0.75, 1.0, 1.5], ns)
    wt[rs.rand(ns) < 0.05] = 0.0
    impd = rs.choice([-1, 1, 2, 3, 5], ns).astype(np.int8)
    init = np.where(impd == -1, 1, rs.randint(0, 2, ns)).astype(np.int8)
    if heavy:
        one = (~from_main) & (ip == pool[0])
        wt[one] = 0.3; impd[one] = 2; init[one] = 0
        imp_dets = np.nonzero(main["imp_distance"] == 0)[0]
        sel = rs.rand(ns) < 0.12
        up[sel], dn[sel] = main["up"][imp_dets[5]], main["dn"][imp_dets[5]]
        wt[sel] = -0.2; impd[sel] = np.where(rs.rand(int(sel.sum())) < 0.5, -1, 2); init[sel] = 1
    prm = dict(tau=s.tau, e_trial=s.e_trial0, reweight_factor_inv=0.97, r_initiator=1.0, min_wt=0.5, always_spawn_cutoff_wt=0.5,
               initiator_power=0, initiator_min_distance=0, c_t_initiator=0, semistochastic=1, reached_w_abs_gen=2)
    ow = oracle.OracleWalk(heg57, s, main, 50000, list(SEED), rng_mode=rng_mode)
    w, nz = ow.w, np.nonzero(wt)[0]
    for k, j in enumerate(nz):
        i = n0 + k
        w.up[i], w.dn[i], w.wt[i], w.imp_distance[i], w.initiator[i] = int(up[j]), int(dn[j]), float(wt[j]), int(impd[j]), int(init[j])
        w.matrix_elements[i] = w.e_num_walker[i] = w.e_den_walker[i] = 1e51
    n = n0 + len(nz)
    p = oracle.StepParams(**prm)
    L = oracle.lib()
    L.orc_reduce_my_walker.restype = C.c_int64
    L.orc_reduce_my_walker.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.orc_merge_original_with_spawned2.restype = C.c_int64
    L.orc_merge_sort_walkers(ow.h, n)
    n = L.orc_merge_original_with_spawned2(ow.h, n, C.byref(p))
    n = L.orc_reduce_my_walker(ow.h, n, C.byref(p))
    ow.w.nwalk = n
    ref = ow.walkers(); ow.close()
    ref["wt"] = ref["wt"] * prm["reweight_factor_inv"]
    from conftest import gpu_ctx_heg
    g = gpu_ctx_heg(heg57, rng_mode=rng_mode, seed=SEED, mwalk=50000)
    g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    g.upload_walkers(main)
    try:
        out = g.annihilate(prm, dict(up=up, dn=dn, wt=wt, imp_distance=impd, initiator=init))
        got = g.download_walkers()
    finally:
        g.close()
    assert len(got["up"]) == n == int(out[5])
    for k in ("up", "dn", "wt", "imp_distance", "initiator"):
        assert np.array_equal(got[k], ref[k]), k
    assert int(out[7]) == n0 + len(nz) and n < n0 + len(nz) - 1000


def test_forced_unpacked_keys_on_c2_are_bit_exact():
    """SQMC_FORCE_UNPACKED=1 gives C2 (28-bit keys, normally packed with the walker index into one word) the two-array
    layout of wide keys.  The trajectory, annihilation-door and time-reversal tests must pass unchanged against the
    oracle: the two key layouts are compared with each other on the best-pinned system, including the spawn-only sort +
    two-array merge path (SQMC_MERGE_SORT_MIN=0)."""
    import subprocess, sys
    env = dict(os.environ, SQMC_FORCE_UNPACKED="1", SQMC_MERGE_SORT_MIN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "(trajectory_bit_exact and not heg57 and not past_2_20) or annihilate_door_matches or time_sym_walk"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("which", ["c2", "heg57"])
def test_det_owner_djb_matches_oracle(oracle, c2_walk, heg57, which):
    """A11: get_det_owner -> hash -> djb_hash (mpi_routines.f90:419-445, 257-289, 354-379) in its 128-bit wrapping
    arithmetic, GPU (sqmc_gpu_set_owner_hash 1 + sqmc_gpu_det_owner) against the oracle's restatement, for every rank
    count the sharded step accepts at its ends and in between."""
    from conftest import gpu_ctx_heg
    L = oracle.lib()
    L.orc_get_det_owner.argtypes = [C.c_uint64] * 4 + [C.c_int]
    rng = np.random.default_rng(11)
    if which == "c2":
        g, norb, ne = gpu_ctx_from_oracle(c2_walk), 26, 4
    else:
        g, norb, ne = gpu_ctx_heg(heg57), 57, 7
    n = 4000
    up, dn = _random_dets(rng, norb, ne, n), _random_dets(rng, norb, ne, n)
    try:
        g.set_owner_hash(1)
        for nranks in (1, 2, 3, 7, 8, 64, 255):
            got = g.det_owner(up, dn, nranks)
            ref = np.array([L.orc_get_det_owner(int(a), 0, int(b), 0, nranks) for a, b in zip(up, dn)])
            assert np.array_equal(got, ref), nranks
            if nranks == 8:
                assert np.bincount(got, minlength=8).min() > n // 16      # the reference's hash spreads these determinants
        g.set_owner_hash(0)
        assert not np.array_equal(g.det_owner(up, dn, 8), ref8 := np.array([L.orc_get_det_owner(int(a), 0, int(b), 0, 8) for a, b in zip(up, dn)]))
    finally:
        g.close()


@pytest.mark.parametrize("env,expect", [(dict(SQMC_BUCKET="0"), "off"), (dict(SQMC_BUCKET_FORCE_RETRY="3", SQMC_BUCKET_HOLDOFF="0"), "retry")])
def test_bucket_tail_variants_are_bit_exact(env, expect):
    """Short lists (< 2^20 sorted slots, packed keys, COUNTER discipline) take the bucket tail: block-local partition of the
    spawns into key ranges + one annihilation kernel per range, no global sort.  SQMC_BUCKET=0 sends the same tests through
    the radix tail; SQMC_BUCKET_FORCE_RETRY=3 makes every third bucket step behave as if a range did not fit its block's LDS:
    it raises the retry flag and is re-run through the radix tail (the rollback of the host's bookkeeping, and the
    pipelined head that must do nothing).  Trajectories, the annihilation door and the pipelined run must equal the oracle."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "(trajectory_bit_exact and not heg57 and not past_2_20) or annihilate_door_matches or run_loop_equals or gate_fused"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


# ------------------------------------------------------------------ efficient heat-bath proposal
@pytest.fixture(scope="module")
def c2_10e(oracle):
    """the shipped C2 cc-pVDZ integrals with 10 electrons: the synthetic system on which the reference's own check accepts
    fast_heatbath (tests/golden/README_heatbath.md)"""
    from conftest import FCIDUMP
    return oracle.ChemSystem(FCIDUMP, 10, 5, "d2h", time_sym=False, hf_mode=0)


@pytest.mark.parametrize("time_sym", [False, True])
def test_heatbath_proposals_bit_exact(oracle, c2_10e, time_sym):
    """off_diagonal_move_chem_efficient_heatbath (chemistry.f90:5086-5347): 10^4 proposals, both slots of each (the single AND
    double return included), determinants, weights and the RNG state after -- GPU kernel against the oracle, with the tables
    handed over in the reference's own array layout; without and with time-reversal symmetry (the second pathway through
    proposal_prob_efficient_heatbath, 5350-5549)."""
    from conftest import FCIDUMP
    sysm = c2_10e if not time_sym else oracle.ChemSystem(FCIDUMP, 10, 5, "d2h", time_sym=True, z=1, hf_mode=0)
    hb = oracle.HeatBath(sysm)
    assert hb.unbiased
    g = gpu_ctx_from_oracle(sysm)
    g.set_heatbath_tables(hb.fortran_arrays())
    rng = np.random.default_rng(10)
    cu, cd, _ = sysm.connected(sysm.hf_up, sysm.hf_dn, with_elems=False)
    n = 10000
    up = np.concatenate((np.full(1500, sysm.hf_up, np.uint64), np.resize(cu, 3500), _random_dets(rng, 26, 5, n - 5000)))
    dn = np.concatenate((np.full(1500, sysm.hf_dn, np.uint64), np.resize(cd, 3500), _random_dets(rng, 26, 5, n - 5000)))
    if time_sym:                                     # representatives only (up <= dn), as the walk holds them
        sw = up > dn
        up[sw], dn[sw] = dn[sw].copy(), up[sw].copy()
    seeds = rng.integers(0, 4096, size=(n, 4)).astype(np.int32); seeds[:, 3] |= 1
    tau = 0.0045
    pju, pjd, wj, sa = g.propose_heatbath_batch(tau, up, dn, seeds)
    g.close()
    L = oracle.lib()
    r = oracle.Rng()
    two = moves = 0
    for i in range(n):
        L.orc_setrn(C.byref(r), (C.c_int * 4)(*seeds[i]))
        ju, jd, w, lev, nd = (C.c_uint64 * 2)(), (C.c_uint64 * 2)(), (C.c_double * 2)(), (C.c_int * 2)(), C.c_int()
        nn = L.orc_off_diagonal_move_chem_heatbath(sysm.h, hb.h, C.byref(r), tau, int(up[i]), int(dn[i]), ju, jd, w, lev, C.byref(nd))
        assert [r.l[k] for k in range(4)] == list(sa[i]), i
        for k in range(2):
            assert w[k] == wj[i, k], (i, k, w[k], wj[i, k])
            if w[k] != 0.0:
                moves += 1
                assert (ju[k], jd[k]) == (int(pju[i, k]), int(pjd[i, k])), (i, k)
        two += (nn == 2 and w[0] != 0.0 and w[1] != 0.0)
    hb.close()
    assert moves > n // 3 and two > 20


@pytest.mark.parametrize("chained", [False, True])
def test_heatbath_walk_trajectory_bit_exact(oracle, c2_10e, chained):
    """A semistochastic walk with proposal_method fast_heatbath (two walker slots per child, do_walk.f90:3604-3611 ->
    add_walker 7584-7697) in the COUNTER discipline: 120 steps, walkers, weights and flags equal the oracle's.  chained: past the
    target population every step enqueues its successor's head (gate fused into the annihilation kernel, k_spawn<1, .> behind it)."""
    sysm = c2_10e
    hb = oracle.HeatBath(sysm)
    su = oracle.setup_walk(sysm, 100, 1000, 0.1)
    g = gpu_ctx_from_oracle(sysm, rng_mode=1, seed=SEED, mwalk=400000)
    g.set_heatbath_tables(hb.fortran_arrays())
    g.set_projector(su.prj_counts, su.prj_indices, su.prj_values)
    g.set_ct_table(su.ct_up, su.ct_dn, su.ct_num, su.ct_den)
    wk = oracle.initial_walkers(su, 50)
    g.upload_walkers(wk)
    if chained: g.set_chained_runs(True)
    ow = oracle.OracleWalk(sysm, su, wk, 400000, SEED, rng_mode=1, heatbath=hb)
    pc = oracle.PopControl(su.tau, su.e_trial0, 600 if chained else 8000)
    w_abs = float(np.abs(wk["wt"]).sum())
    try:
        for it in range(200 if chained else 120):
            r = pc.pre_step(w_abs)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            st, oc = ow.step(pc.params())
            og = g.step(pc.params())
            assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15], (it, og, oc)
            assert _sums_close(og, oc), (it, [(k, og[k], oc[k]) for k in range(16) if og[k] != oc[k]])
            r = pc.post_step(oc)
            if r != 1.0:
                ow.scale_projector(r); g.scale_projector(r)
            w_abs = oc[1]
        wg, wc = g.download_walkers(), ow.walkers()
    finally:
        g.close(); ow.close(); hb.close()
    for k in ("up", "dn", "imp_distance", "initiator"):
        assert np.array_equal(wg[k], wc[k]), k
    assert np.array_equal(wg["wt"], wc["wt"]) and len(wg["up"]) > (300 if chained else 3000)
    assert not chained or pc.reached == 2


def test_library_builds_the_heatbath_tables_itself(oracle, c2_walk, c2_10e):
    """sqmc_gpu_setup_efficient_heatbath (setup_efficient_heatbath chemistry.f90:1002-1225, setup_alias more_tools.f90:5603-5722,
    check_heatbath_unbiased 9330-9375 on the product side): every table equals the oracle's, entry for entry -- the partial sums in
    double, the four-index tables and their alias tables in single precision --, the 8-electron system the reference refuses is
    refused, and a walk driven by the library's own tables is the walk driven by the oracle's."""
    sysm = c2_10e
    hb = oracle.HeatBath(sysm)
    ref = hb.fortran_arrays()
    g = gpu_ctx_from_oracle(sysm, rng_mode=1, seed=SEED, mwalk=300000)
    assert g.setup_efficient_heatbath() is True and hb.unbiased
    mine = g.heatbath_tables()
    assert mine["n_orb_uniq_sym"] == hb.s.n_orb_uniq_sym and mine["size_same"] == ref["size_same"] and mine["size_opp"] == ref["size_opp"]
    for k in ("one", "two", "three_same", "three_opp", "q3_same", "q3_opp", "htot_same", "htot_opp", "four_same", "four_opp", "q4_same", "q4_opp"):
        a, b = np.asarray(mine[k]), np.asarray(ref[k]).reshape(-1)
        assert a.shape == b.shape and np.array_equal(a, b), (k, np.max(np.abs(a.astype(float) - b.astype(float))))
    for k in ("j3_same", "j3_opp", "j4_same", "j4_opp"):
        assert np.array_equal(np.asarray(mine[k]), np.asarray(ref[k]).reshape(-1)), k
    assert np.count_nonzero(mine["four_same"]) > 10000 and np.count_nonzero(mine["q4_opp"]) > 10000
    # a walk on the library's tables against the oracle's walk on its own
    su = oracle.setup_walk(sysm, 100, 1000, 0.1)
    g.set_projector(su.prj_counts, su.prj_indices, su.prj_values)
    g.set_ct_table(su.ct_up, su.ct_dn, su.ct_num, su.ct_den)
    wk = oracle.initial_walkers(su, 50)
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(sysm, su, wk, 300000, SEED, rng_mode=1, heatbath=hb)
    pc = oracle.PopControl(su.tau, su.e_trial0, 4000)
    w_abs = float(np.abs(wk["wt"]).sum())
    for it in range(40):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        st, oc = ow.step(pc.params())
        og = g.step(pc.params())
        assert st == 0 and og[5] == oc[5] and og[15] == oc[15] and _sums_close(og, oc), (it, og, oc)
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
    wg, wc = g.download_walkers(), ow.walkers()
    assert np.array_equal(wg["up"], wc["up"]) and np.array_equal(wg["dn"], wc["dn"]) and np.array_equal(wg["wt"], wc["wt"])
    g.close(); ow.close(); hb.close()
    # C2 with 8 electrons: 4 orbitals of unique symmetry are not fewer than max(nup, ndn) = 4 -- "Heatbath may be biased for this system!"
    g8 = gpu_ctx_from_oracle(c2_walk)
    assert g8.setup_efficient_heatbath() is False
    g8.close()


def test_heatbath_walk_replay_discipline(oracle, c2_10e):
    """fast_heatbath in the REPLAY discipline: one lane walks the reference's single rannyu stream through every proposal (alias draws,
    the second excitation of a single-and-double return) and the parallel k_spawn<1, .> resumes from the recorded states: walkers,
    weights and the stream's state equal the oracle's after every block of steps."""
    sysm = c2_10e
    hb = oracle.HeatBath(sysm)
    su = oracle.setup_walk(sysm, 100, 1000, 0.1)
    g = gpu_ctx_from_oracle(sysm, rng_mode=0, seed=SEED, mwalk=300000)
    g.set_heatbath_tables(hb.fortran_arrays())
    g.set_projector(su.prj_counts, su.prj_indices, su.prj_values)
    g.set_ct_table(su.ct_up, su.ct_dn, su.ct_num, su.ct_den)
    wk = oracle.initial_walkers(su, 50)
    g.upload_walkers(wk)
    ow = oracle.OracleWalk(sysm, su, wk, 300000, SEED, rng_mode=0, heatbath=hb)
    pc = oracle.PopControl(su.tau, su.e_trial0, 2500)
    w_abs = float(np.abs(wk["wt"]).sum())
    for it in range(40):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        st, oc = ow.step(pc.params())
        og = g.step(pc.params())
        assert st == 0 and og[5] == oc[5] and og[7] == oc[7] and og[15] == oc[15] and _sums_close(og, oc), (it, og, oc)
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g.scale_projector(r)
        w_abs = oc[1]
        if it % 10 == 9:
            assert g.rng_state() == ow.rng_state(), it
    wg, wc = g.download_walkers(), ow.walkers()
    g.close(); ow.close(); hb.close()
    for k in ("up", "dn", "imp_distance", "initiator", "wt"):
        assert np.array_equal(wg[k], wc[k]), k
    assert len(wg["up"]) > 1200
