"""Sharded (multi-rank) walk on ONE GPU: several processes share the card and exchange through
gloo (host-staged), which exercises everything of the multi-GPU path except the RCCL transport
itself -- owner hashing, per-destination bucketing, the three exchanges, local annihilation."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FCIDUMP = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
SEED = (1346, 5634, 6635, 4361)
NSTEPS, W_BEGIN, W_TARGET = 40, 2000, 20000


def _worker(rank, world, port, outdir, system="c2", w_begin=None, w_target=None, nsteps=None, walk_kw=None):
    W_BEGIN, W_TARGET, NSTEPS = w_begin or globals()["W_BEGIN"], w_target or globals()["W_TARGET"], nsteps or globals()["NSTEPS"]
    import torch                                   # before the HIP library (one libamdhip64 per process)
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    if system == "heg":       # 14 electrons, 19 plane waves: the reference's e2e HEG system (BASELINE.json configs[3] is its big brother)
        hst = H.HegHost(3, 0.5, 14, 7, 1.49)
        w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=400000, n_truncate_trial_wf=1, size_deterministic=250)
    elif system == "heg57":   # BASELINE.json configs[3] itself: r_s = 1.0, 57 plane waves, 56-bit sort keys (two-array sort records)
        hst = H.HegHost(3, 1.0, 14, 7, 2.3)
        w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=400000, n_truncate_trial_wf=1, size_deterministic=300)
    elif system == "hub":     # BASELINE.json configs[0]: 4x4 Hubbard, U/t = 4, half filling
        hst = H.HubbardHost(4, 4, True, 8, 8, 1.0, 4.0)
        w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=400000, n_truncate_trial_wf=20, size_deterministic=500,
                          tau_multiplier=0.5)
    elif system == "c2_plain":   # semistochastic = f: no deterministic space, join_walker2 local to a rank
        hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
        w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=400000, semistochastic=False)
    else:
        hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
        w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=400000, owner_hash=1 if system == "c2_djb" else 0, **(walk_kw or {}))
    outs = []
    for _ in range(NSTEPS):
        outs.append(w.step().copy())
    wk = w.g.download_walkers()
    owner = w.g.det_owner(wk["up"], wk["dn"], world)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), outs=np.array(outs), owner=owner, n_imp_global=w.n_imp_global,
             local_last=w.last_local, **wk)
    w.close()
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tmp_path, port, system="c2", **kw):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), system), kwargs=kw) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(600)
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


def _single_rank_worker(outdir, bucket, w_target=None, nsteps=None):
    W_TARGET, NSTEPS = w_target or globals()["W_TARGET"], nsteps or globals()["NSTEPS"]
    os.environ["SQMC_BUCKET"] = str(bucket)        # read once per process, hence a process of its own
    sys.path.insert(0, ROOT)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    ref = H.GpuWalk(hst, W_TARGET, w_begin=W_BEGIN, seed=SEED, mwalk=400000)
    outs = np.array([ref.step().copy() for _ in range(NSTEPS)])
    wk = ref.g.download_walkers()
    np.savez(os.path.join(outdir, "single%d.npz" % bucket), outs=outs, tail=np.array(ref.g.tail_stats()), **wk)
    ref.close()


def _single_rank_radix_tail(tmp_path, w_target=None, nsteps=None):
    """the plain single-rank walk on the radix tail (the tail the sharded step runs), in a process of its own"""
    import torch.multiprocessing as mp
    pr = mp.get_context("spawn").Process(target=_single_rank_worker, args=(str(tmp_path), 0, w_target, nsteps))
    pr.start(); pr.join(600)
    assert pr.exitcode == 0
    return np.load(os.path.join(str(tmp_path), "single0.npz"))


def test_one_rank_sharded_equals_single_rank_step(tmp_path, monkeypatch):
    """With one rank the sharded pipeline (bucket -> exchange -> unpack) must reproduce the
    single-rank step: bit for bit against the single-rank step on the same tail (the radix tail, the
    one the sharded step uses); against the short-list tail the estimator sums come out of another
    reduction tree, E_T follows them, and the weights agree to rounding."""
    monkeypatch.setenv("SQMC_SHARD_BUCKET", "0")      # bit for bit against the single-rank RADIX tail: the sharded step has to run that tail too (the short-list tail sums in another tree)
    import torch.multiprocessing as mp
    res = _run(1, tmp_path, 29541)[0]
    ctx = mp.get_context("spawn")
    for bucket in (0, 1):
        pr = ctx.Process(target=_single_rank_worker, args=(str(tmp_path), bucket))
        pr.start(); pr.join(600)
        assert pr.exitcode == 0
    wk = np.load(os.path.join(str(tmp_path), "single0.npz"))
    assert tuple(wk["tail"]) == (0, 0)
    assert np.array_equal(res["up"], wk["up"]) and np.array_equal(res["dn"], wk["dn"])
    assert np.array_equal(res["wt"], wk["wt"]) and np.array_equal(res["initiator"], wk["initiator"])
    assert np.allclose(res["outs"], wk["outs"], rtol=1e-12, atol=1e-12)
    wb = np.load(os.path.join(str(tmp_path), "single1.npz"))
    assert wb["tail"][0] > NSTEPS // 2
    assert np.array_equal(res["up"], wb["up"]) and np.array_equal(res["dn"], wb["dn"]) and np.array_equal(res["initiator"], wb["initiator"])
    assert np.allclose(res["wt"], wb["wt"], rtol=1e-11, atol=0) and np.allclose(res["outs"], wb["outs"], rtol=1e-11, atol=1e-11)


def _single_rank_plain_worker(outdir, w_begin, w_target, nsteps):
    sys.path.insert(0, ROOT)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    g = hst.gpu(rng_mode=H.RNG_COUNTER, seed=H.rank_seed(SEED, 0), mwalk=400000)
    s = hst.setup_walk(g, 100, 1000, 0.1)
    g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
    wk = H.initial_walkers(s, w_begin)
    wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)
    keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
    wk = {k: v[keep] for k, v in wk.items()}
    g.upload_walkers(wk)
    pc = H.PopControl(s.tau, s.e_trial0, w_target, n_equil_steps=10**9)
    w_abs, outs = float(np.abs(wk["wt"]).sum()), []
    for _ in range(nsteps):
        pc.pre_step(w_abs)
        out = g.step(pc.params(min_wt=0.5, semistochastic=0))
        pc.post_step(out)
        w_abs = out[1]; outs.append(out.copy())
    np.savez(os.path.join(outdir, "plain_single.npz"), outs=np.array(outs), **g.download_walkers())
    g.close()


def test_sharded_plain_walk(tmp_path):
    """semistochastic = f across ranks (do_walk.f90:2475: join_walker2 is local to a rank; no deterministic space, no projection
    exchange): one rank must walk the single-rank plain trajectory bit for bit; two ranks must keep the sharding invariants
    (disjoint ownership, globally unique determinants, identical all-reduced sums) and every walker outside any deterministic
    space."""
    import torch.multiprocessing as mp
    one = os.path.join(str(tmp_path), "one"); os.makedirs(one)
    res1 = _run(1, one, 29561, system="c2_plain")[0]
    pr = mp.get_context("spawn").Process(target=_single_rank_plain_worker, args=(one, W_BEGIN, W_TARGET, NSTEPS))
    pr.start(); pr.join(600)
    assert pr.exitcode == 0
    ref = np.load(os.path.join(one, "plain_single.npz"))
    for k in ("up", "dn", "wt", "initiator", "imp_distance"):
        assert np.array_equal(res1[k], ref[k]), k
    assert np.allclose(res1["outs"], ref["outs"], rtol=1e-12, atol=1e-12) and len(ref["up"]) > 2000
    two = os.path.join(str(tmp_path), "two"); os.makedirs(two)
    res = _run(2, two, 29562, system="c2_plain")
    assert np.array_equal(res[1]["outs"][:, :7], res[0]["outs"][:, :7])
    keys = []
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank) and np.all(r["imp_distance"] != 0)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
    assert len(keys) == len(set(keys)) and int(res[0]["outs"][-1][5]) == len(keys)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert -75.85 < e < -75.55


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_walk_invariants(tmp_path, world):
    res = _run(world, tmp_path, 29541 + world)
    # every rank saw the same all-reduced sums
    for r in res[1:]:
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
    keys = []
    n_imp = 0
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)                                   # ownership respected
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))                                          # sorted, unique locally
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
        assert len(r["up"]) == int(r["local_last"][5])
    assert len(keys) == len(set(keys))                                      # a determinant lives on one rank only
    assert n_imp == int(res[0]["n_imp_global"])                             # the deterministic space is complete
    out = res[0]["outs"][-1]
    assert int(out[5]) == len(keys)                                         # global nwalk = sum of the shards
    wsum = sum(float(np.abs(r["wt"]).sum()) for r in res)
    assert np.isclose(wsum, out[1], rtol=1e-12)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert -75.80 < e < -75.55
    assert out[1] > 1.5 * W_BEGIN                                           # the population grew towards the target


def test_sharded_walk_with_the_references_owner_hash(tmp_path):
    """Ownership by the reference's own get_det_owner / djb_hash (mpi_routines.f90:354-445; sqmc_gpu_set_owner_hash 1): the
    walk keeps every sharding invariant, and the owner the GPU assigned to every surviving determinant equals the CPU
    restatement's -- an ownership check against something other than the function that did the sharding."""
    import ctypes as C
    from oracle import oracle as O
    O.build()
    L = O.lib()
    L.orc_get_det_owner.argtypes = [C.c_uint64] * 4 + [C.c_int]
    res = _run(3, tmp_path, 29547, system="c2_djb")
    keys, n_imp = [], 0
    for rank, r in enumerate(res):
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
        assert np.all(r["owner"] == rank)
        assert all(L.orc_get_det_owner(int(a), 0, int(b), 0, 3) == rank for a, b in zip(r["up"], r["dn"]))
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
        assert len(k) > 1000                       # the hash balances: no rank is starved
    assert len(keys) == len(set(keys)) and n_imp == int(res[0]["n_imp_global"])
    assert int(res[0]["outs"][-1][5]) == len(keys)


def _nccl_worker(port, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    w = H.ShardedWalk(H.ChemHost(FCIDUMP, 8, 4, "d2h"), W_TARGET, 0, 1, w_begin=W_BEGIN, seed=SEED, mwalk=400000)
    outs = np.array([w.step().copy() for _ in range(10)])
    np.save(os.path.join(outdir, "nccl.npy"), outs)
    w.close()
    dist.destroy_process_group()


def test_sharded_step_over_rccl_single_rank(tmp_path):
    """The same orchestration with the nccl (= RCCL) backend and device tensors; one rank is all a
    one-GPU box can host, it still runs every collective call on the RCCL code path."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_worker, args=(29561, str(tmp_path)))
    p.start(); p.join(600)
    assert p.exitcode == 0
    outs = np.load(os.path.join(str(tmp_path), "nccl.npy"))
    assert outs.shape == (10, 16) and np.all(outs[:, 5] > 1000)


def _inlib_worker(port, outdir):
    import torch                                   # noqa: F401  (its librccl/libamdhip64 first)
    sys.path.insert(0, ROOT)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    w = H.ShardedWalk(hst, W_TARGET, 0, 1, w_begin=W_BEGIN, seed=SEED, mwalk=400000)
    w.attach_rccl()
    a = np.array([w.step().copy() for _ in range(NSTEPS // 2)])        # sqmc_gpu_shard_step
    b, _ = w.run(NSTEPS - NSTEPS // 2)                                 # sqmc_gpu_shard_run
    wk = w.g.download_walkers()
    split, nsplit = w.g.shard_time_split(reset=True)          # sqmc_gpu_shard_time_split: every step of both loops went through it
    assert nsplit == NSTEPS and all(v >= 0.0 for v in split.values()) and split["head_us"] + split["exchange_us"] + split["tail_us"] > 1.0
    assert split["of_which_waiting_for_the_gpu_us"] <= split["head_us"] + split["exchange_us"] + split["tail_us"] + 1e-9
    assert w.g.shard_time_split()[1] == 0
    np.savez(os.path.join(outdir, "inlib.npz"), outs=np.concatenate([a, b]), **wk)
    w.close()


def _inlib_long_worker(port, outdir, w_target, nsteps):
    import torch                                   # noqa: F401
    sys.path.insert(0, ROOT)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    w = H.ShardedWalk(hst, w_target, 0, 1, w_begin=W_BEGIN, seed=SEED, mwalk=400000)
    w.attach_rccl()
    outs, _ = w.run(nsteps)
    wk = w.g.download_walkers()
    np.savez(os.path.join(outdir, "inlib_long.npz"), outs=outs, reached=np.array([w.pc.reached]), **wk)
    w.close()


def test_in_library_pipelined_run_matches_plain_walk(tmp_path, monkeypatch):
    """sqmc_gpu_shard_run past the point where the target population is reached: from there on the steps are
    pipelined (the annihilation kernel computes the next gate, the next step's scan posts the all-reduced sums).
    One rank over real RCCL must still walk the plain single-rank trajectory, bit for bit."""
    monkeypatch.setenv("SQMC_SHARD_BUCKET", "0")      # bit for bit against the single-rank RADIX tail: the sharded step has to run that tail too (the short-list tail sums in another tree)
    import torch.multiprocessing as mp
    w_target, nsteps = 4000, 300
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_inlib_long_worker, args=(29575, str(tmp_path), w_target, nsteps))
    p.start(); p.join(600)
    assert p.exitcode == 0
    res = np.load(os.path.join(str(tmp_path), "inlib_long.npz"))
    assert int(res["reached"][0]) == 2
    wk = _single_rank_radix_tail(tmp_path, w_target, nsteps); outs = wk["outs"]
    assert np.array_equal(res["up"], wk["up"]) and np.array_equal(res["dn"], wk["dn"])
    assert np.array_equal(res["wt"], wk["wt"])
    assert np.allclose(res["outs"], outs, rtol=1e-12, atol=1e-12)


def test_in_library_rccl_exchange_single_rank(tmp_path, monkeypatch):
    """sqmc_gpu_comm_init / shard_step / shard_run: the exchanges issued by the library itself on
    an RCCL communicator (one rank is what a one-GPU box can host: every RCCL call still runs).
    Must reproduce the plain single-rank walk exactly."""
    monkeypatch.setenv("SQMC_SHARD_BUCKET", "0")      # bit for bit against the single-rank RADIX tail: the sharded step has to run that tail too (the short-list tail sums in another tree)
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_inlib_worker, args=(29571, str(tmp_path)))
    p.start(); p.join(600)
    assert p.exitcode == 0
    res = np.load(os.path.join(str(tmp_path), "inlib.npz"))
    wk = _single_rank_radix_tail(tmp_path); outs = wk["outs"]
    assert np.array_equal(res["up"], wk["up"]) and np.array_equal(res["dn"], wk["dn"])
    assert np.array_equal(res["wt"], wk["wt"])
    assert np.allclose(res["outs"], outs, rtol=1e-12, atol=1e-12)


def test_sharded_heg_walk_invariants(tmp_path):
    """BASELINE.json configs[3] in miniature: the homogeneous electron gas walk sharded over two
    ranks (momentum-conserving proposals, hamiltonian_heg), same invariants as for C2."""
    res = _run(2, tmp_path, 29581, system="heg")
    for r in res[1:]:
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
    keys, n_imp = [], 0
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
    assert len(keys) == len(set(keys)) and n_imp == int(res[0]["n_imp_global"])
    out = res[0]["outs"][-1]
    assert int(out[5]) == len(keys)
    assert np.isclose(sum(float(np.abs(r["wt"]).sum()) for r in res), out[1], rtol=1e-12)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert 58.0 < e < 58.7                  # HF energy 58.592675 (heg.f90 e2e reference), correlation lowers it
    assert out[1] > 1.5 * W_BEGIN


def test_sharded_heg57_walk_invariants(tmp_path):
    """BASELINE.json configs[3]'s own system (r_s = 1.0, 57 plane waves: 56-bit keys, the two-array sort records) sharded
    over two ranks: ownership, uniqueness, complete deterministic space, identical all-reduced sums."""
    res = _run(2, tmp_path, 29585, system="heg57")
    for r in res[1:]:
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
    keys, n_imp = [], 0
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
    assert len(keys) == len(set(keys)) and n_imp == int(res[0]["n_imp_global"])
    assert max(k[0] for k in keys) >= (1 << 32)
    out = res[0]["outs"][-1]
    assert int(out[5]) == len(keys)
    assert np.isclose(sum(float(np.abs(r["wt"]).sum()) for r in res), out[1], rtol=1e-12)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert 13.0 < e < 13.7                  # HF 13.60 at r_s = 1, the correlated state below it
    assert out[1] > 1.5 * W_BEGIN


def test_sharded_hubbard_walk_invariants(tmp_path):
    """BASELINE.json configs[0] lattice (4x4 Hubbard, U/t = 4, half filling) sharded over two ranks:
    off_diagonal_move_hubbard / hamiltonian_hubbard under the same ownership and exchange rules."""
    res = _run(2, tmp_path, 29591, system="hub")
    for r in res[1:]:
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
    keys, n_imp = [], 0
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
    assert len(keys) == len(set(keys)) and n_imp == int(res[0]["n_imp_global"])
    out = res[0]["outs"][-1]
    assert int(out[5]) == len(keys)
    assert np.isclose(sum(float(np.abs(r["wt"]).sum()) for r in res), out[1], rtol=1e-12)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert -14.5 < e < -8.0                 # between the exact ground state (-13.62) and the Neel determinant's neighbourhood
    assert out[1] > 1.5 * W_BEGIN


# ---------------------------------------------------------------- in-library exchange with several ranks on one GPU
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")


def _fake_rccl_lib():
    """tests/fake_rccl: the RCCL entry points the library binds, over POSIX shared memory (test infrastructure)"""
    import subprocess
    so, src = os.path.join(FAKE_DIR, "libfake_rccl.so"), os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-Wno-unused-result", src, "-o", so, "-lrt"])
    return so


def _inlib_multi_worker(rank, world, port, outdir, fake, w_target=None, nsteps=None, nofuse=False, overlap=False, w_begin=None, mwalk_of_rank=None, env=None, plain=False, walk_kw=None):
    os.environ.update(env or {})
    import torch                                   # noqa: F401
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    W_TARGET, NSTEPS, W_BEGIN = w_target or globals()["W_TARGET"], nsteps or globals()["NSTEPS"], w_begin or globals()["W_BEGIN"]
    if overlap: os.environ["SQMC_SHARD_OVERLAP"] = "1"     # second communicator + side stream (and with it the pipelined sharded run)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["SQMC_RCCL_LIB"] = fake             # before the library binds its communication entry points
    if nofuse: os.environ["SQMC_NO_GATE_FUSION"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)     # only carries the unique id (MPI_Bcast in the reference's build)
    import sqmc_amd
    from sqmc_amd import host as H
    sqmc_amd.set_device(0)
    hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
    w = H.ShardedWalk(hst, W_TARGET, rank, world, w_begin=W_BEGIN, seed=SEED, mwalk=(mwalk_of_rank or {}).get(rank, 400000), semistochastic=not plain, **(walk_kw or {}))
    w.attach_rccl()
    status, outs = 0, []
    try:
        for _ in range(NSTEPS // 2):                                   # sqmc_gpu_shard_step
            outs.append(w.step().copy())
        b, _ = w.run(NSTEPS - NSTEPS // 2)                             # sqmc_gpu_shard_run
        outs = np.concatenate([np.array(outs), b])
    except sqmc_amd.SqmcGpuError as exc:
        if not mwalk_of_rank:
            raise
        status, outs = exc.code, np.array(outs).reshape(-1, 16)        # a walk that is meant to stop: which status, after how many steps
    wk = w.g.download_walkers() if status == 0 else dict(up=np.zeros(0, np.uint64), dn=np.zeros(0, np.uint64), wt=np.zeros(0), imp_distance=np.zeros(0, np.int8))
    owner = w.g.det_owner(wk["up"], wk["dn"], world) if status == 0 else np.zeros(0, np.int32)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), outs=outs, owner=owner, n_imp_global=w.n_imp_global,
             reached=np.array([w.pc.reached]), status=np.array([status]), tail=np.array(w.g.tail_stats()), **wk)
    w.close()
    dist.barrier()
    dist.destroy_process_group()


def test_in_library_pipelined_exchange_two_ranks(tmp_path):
    """Two ranks (transport double) run well past the target population, so that sqmc_gpu_shard_run pipelines its steps:
    the annihilation kernel computes the next gate and the next step's scan posts the all-reduced sums.  The ownership
    invariants must hold, and the walk must be the same, bit for bit on every rank, as with SQMC_NO_GATE_FUSION=1."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    runs = []
    for k, nofuse in enumerate((False, True)):
        out = os.path.join(str(tmp_path), "nofuse%d" % k); os.makedirs(out)
        ps = [ctx.Process(target=_inlib_multi_worker, args=(r, 2, 29620 + k, out, fake, 4000, 300, nofuse, True)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join(300)
        alive = [p for p in ps if p.is_alive()]
        for p in alive: p.terminate()
        assert not alive, "in-library exchange did not finish (deadlock?)"
        assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
        runs.append([np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(2)])
    res = runs[0]
    assert all(int(r["reached"][0]) == 2 for r in res)
    assert np.array_equal(res[1]["outs"][:, :7], res[0]["outs"][:, :7])
    keys = []
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
    assert len(keys) == len(set(keys)) and int(res[0]["outs"][-1][5]) == len(keys)
    e = res[0]["outs"][100:, 3].sum() / res[0]["outs"][100:, 2].sum()
    assert -75.80 < e < -75.60
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a["outs"], b["outs"])
        assert np.array_equal(a["up"], b["up"]) and np.array_equal(a["dn"], b["dn"]) and np.array_equal(a["wt"], b["wt"])


def test_sharded_plain_walk_in_library(tmp_path):
    """The plain (semistochastic = f) sharded walk with the exchanges issued by the library (two ranks over the transport
    double; steps, then the pipelined run): the same walk as with the caller-driven exchanges."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    lib = os.path.join(str(tmp_path), "lib"); os.makedirs(lib)
    ps = [ctx.Process(target=_inlib_multi_worker, args=(r, 2, 29660, lib, fake), kwargs=dict(plain=True)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    alive = [p for p in ps if p.is_alive()]
    for p in alive: p.terminate()
    assert not alive, "in-library exchange did not finish (deadlock?)"
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    a = [np.load(os.path.join(lib, "rank%d.npz" % r)) for r in range(2)]
    host = os.path.join(str(tmp_path), "host"); os.makedirs(host)
    b = _run(2, host, 29661, system="c2_plain")
    for x, y in zip(a, b):
        assert np.array_equal(x["up"], y["up"]) and np.array_equal(x["dn"], y["dn"]) and np.all(x["imp_distance"] != 0)
        assert np.allclose(x["wt"], y["wt"], rtol=1e-9, atol=0) and np.allclose(x["outs"][:, :7], y["outs"][:, :7], rtol=1e-11, atol=1e-11)
    assert np.array_equal(a[1]["outs"][:, :7], a[0]["outs"][:, :7])


def test_sharded_short_list_tail_when_one_rank_gives_up(tmp_path):
    """The short-list (bucket) tail inside the sharded in-library step.  A bucket that outgrows its block makes a rank re-run its
    tail through the radix path -- with a communicator that has to be a collective decision: the flag rides in the status word of
    the sums' all-reduce, every rank learns of it from the same call, the rank concerned re-runs, and all of them reduce once more
    (the ranks whose tail was fine contribute the same local sums again).  Here rank 1 alone is made to give up on every third
    bucket step: the walk must be the one that never gives up (same determinants on the same ranks; weights to rounding: a re-run
    step sums its estimators in the radix tail's tree), nobody may hang, and only rank 1 may have re-run anything."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    runs = []
    for k, env in enumerate(({}, {"SQMC_BUCKET_FORCE_RETRY": "3", "SQMC_BUCKET_FORCE_RETRY_RANK": "1", "SQMC_BUCKET_HOLDOFF": "1"})):
        out = os.path.join(str(tmp_path), "run%d" % k); os.makedirs(out)
        ps = [ctx.Process(target=_inlib_multi_worker, args=(r, 2, 29640 + k, out, fake, 8000, 240), kwargs=dict(env=env)) for r in range(2)]
        for p in ps: p.start()
        for p in ps: p.join(300)
        alive = [p for p in ps if p.is_alive()]
        for p in alive: p.terminate()
        assert not alive, "sharded step with a re-run tail did not finish (deadlock?)"
        assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
        runs.append([np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(2)])
    plain, forced = runs
    assert all(int(r["tail"][0]) > 100 for r in plain) and all(int(r["tail"][1]) == 0 for r in plain)      # the bucket tail ran, nothing was re-run
    assert int(forced[1]["tail"][1]) >= 10 and int(forced[0]["tail"][1]) == 0
    for a, b in zip(plain, forced):
        assert np.array_equal(a["up"], b["up"]) and np.array_equal(a["dn"], b["dn"])
        assert np.allclose(a["wt"], b["wt"], rtol=1e-10, atol=0) and np.allclose(a["outs"], b["outs"], rtol=1e-10, atol=1e-10)
    for r in forced[1:]:
        assert np.array_equal(r["outs"][:, :7], forced[0]["outs"][:, :7])


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True)])
def test_in_library_exchange_with_several_ranks(tmp_path, world, overlap):
    """sqmc_gpu_shard_step / sqmc_gpu_shard_run with 2 and 3 ranks: real RCCL refuses several ranks on one
    GPU, so the library is pointed (SQMC_RCCL_LIB) at a transport double with the same entry points that
    moves the bytes through shared memory and blocks on every call.  What is under test is the library's
    side: the count matrix, send/receive offsets and sizes (the double checks that what one rank sends is
    what the other expects), one group per step on every rank, the all-reduce of the deterministic weights
    on the second communicator beside the spawn exchange, and the merged result: disjoint ownership, unique
    determinants, identical sums on every rank, a complete deterministic space."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_inlib_multi_worker, args=(r, world, 29600 + world + 4 * int(overlap), str(tmp_path), fake), kwargs=dict(overlap=overlap)) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    alive = [p for p in ps if p.is_alive()]
    for p in alive: p.terminate()
    assert not alive, "in-library exchange did not finish (deadlock?)"
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7])
    keys, n_imp = [], 0
    for rank, r in enumerate(res):
        assert np.all(r["owner"] == rank)
        k = [(int(a), int(b)) for a, b in zip(r["up"], r["dn"])]
        assert k == sorted(set(k))
        keys += k
        n_imp += int((r["imp_distance"] == 0).sum())
    assert len(keys) == len(set(keys)) and n_imp == int(res[0]["n_imp_global"])
    out = res[0]["outs"][-1]
    assert int(out[5]) == len(keys)
    assert np.isclose(sum(float(np.abs(r["wt"]).sum()) for r in res), out[1], rtol=1e-12)
    e = res[0]["outs"][10:, 3].sum() / res[0]["outs"][10:, 2].sum()
    assert -75.80 < e < -75.55 and out[1] > 1.5 * W_BEGIN


def test_in_library_exchange_equals_host_driven_at_tiny_population(tmp_path):
    """The two drivers of the sharded step -- exchanges issued by the library (here over the transport double) and by the
    host through torch.distributed -- walk the same trajectory bit for bit on three ranks.  The population is a handful
    of walkers, so that steps in which a rank spawns exactly one child (or none) occur: the per-destination counts of
    such a step must come from this step's bucketing pass, not from an earlier sort."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    world, kw = 3, dict(w_begin=1.5, w_target=3, nsteps=80)
    a_dir, b_dir = os.path.join(str(tmp_path), "inlib"), os.path.join(str(tmp_path), "host")
    os.makedirs(a_dir); os.makedirs(b_dir)
    ps = [ctx.Process(target=_inlib_multi_worker, args=(r, world, 29640, a_dir, fake), kwargs=dict(w_begin=1.5, w_target=3, nsteps=80)) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    alive = [p for p in ps if p.is_alive()]
    for p in alive: p.terminate()
    assert not alive and all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    a = [np.load(os.path.join(a_dir, "rank%d.npz" % r)) for r in range(world)]
    b = _run(world, b_dir, 29641, **kw)
    single = 0
    for ra, rb in zip(a, b):
        # the two all-reduces add the ranks' sums in different orders (rank order here, gloo's ring there): the reduced sums,
        # and through the population control every weight, agree to round-off; determinants, flags and counts exactly
        assert np.allclose(ra["outs"][:, :7], rb["outs"][:, :7], rtol=1e-11, atol=1e-13)
        assert np.array_equal(ra["outs"][:, [5, 7, 15]], rb["outs"][:, [5, 7, 15]])
        assert np.array_equal(ra["up"], rb["up"]) and np.array_equal(ra["dn"], rb["dn"]) and np.array_equal(ra["imp_distance"], rb["imp_distance"])
        assert np.allclose(ra["wt"], rb["wt"], rtol=1e-9, atol=0)
        single += int((ra["outs"][:, 15] == 1).sum())
    assert single > 0                        # the case under test really occurred


def test_in_library_pipelined_run_with_an_empty_shard(tmp_path):
    """Four ranks, a deterministic space of two determinants and a handful of walkers: ranks that own no deterministic
    determinant hold NO walker in some steps of the pipelined sqmc_gpu_shard_run.  Such a rank enqueues no head for the next
    step, but the all-reduce of the deterministic weights that its peers issue in front of theirs is a collective: it must
    take part, in the same position between the two all-reduces of the sums, or the ranks fall out of step (a hang, or sums
    added to weights).  The walk must equal the host-driven one, in which every exchange is issued by every rank by hand."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    world = 4
    kw = dict(w_begin=1.5, w_target=3, nsteps=160, walk_kw=dict(n_truncate_trial_wf=1, size_deterministic=2))
    a_dir, b_dir = os.path.join(str(tmp_path), "inlib"), os.path.join(str(tmp_path), "host")
    os.makedirs(a_dir); os.makedirs(b_dir)
    ps = [ctx.Process(target=_inlib_multi_worker, args=(r, world, 29650, a_dir, fake), kwargs=kw) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    alive = [p for p in ps if p.is_alive()]
    for p in alive: p.terminate()
    assert not alive and all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    a = [np.load(os.path.join(a_dir, "rank%d.npz" % r)) for r in range(world)]
    b = _run(world, b_dir, 29651, **kw)
    assert 0 < int(a[0]["n_imp_global"]) < world
    empty_in_run = 0
    for ra, rb in zip(a, b):
        assert int(ra["reached"][0]) == 2
        assert np.allclose(ra["outs"][:, :7], rb["outs"][:, :7], rtol=1e-11, atol=1e-13)
        assert np.array_equal(ra["outs"][:, [5, 7, 15]], rb["outs"][:, [5, 7, 15]])
        assert np.array_equal(ra["up"], rb["up"]) and np.array_equal(ra["dn"], rb["dn"]) and np.array_equal(ra["imp_distance"], rb["imp_distance"])
        assert np.allclose(ra["wt"], rb["wt"], rtol=1e-9, atol=0)
        empty_in_run += int((ra["outs"][80:, 7] == 0).sum())
    assert empty_in_run > 0                  # the case under test really occurred, in the half walked by sqmc_gpu_shard_run


def test_in_library_stop_is_collective(tmp_path):
    """A rank that runs out of walker slots ('nwalk>MWALK', do_walk.f90:3684-3690) must stop EVERY rank with that status
    in the same step -- the reference's mpi_stop -- instead of leaving its peers blocked in the next collective: rank 1
    gets a small MWALK, both ranks must return status 1 after the same number of steps, within the time limit."""
    import torch.multiprocessing as mp
    fake = _fake_rccl_lib()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_inlib_multi_worker, args=(r, 2, 29650, str(tmp_path), fake),
                      kwargs=dict(w_begin=2000, w_target=40000, nsteps=400, mwalk_of_rank={1: 9000})) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    alive = [p for p in ps if p.is_alive()]
    for p in alive: p.terminate()
    assert not alive, "a rank is still blocked in a collective after its peer stopped"
    assert all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    assert [int(r["status"][0]) for r in res] == [1, 1]
    assert len(res[0]["outs"]) == len(res[1]["outs"]) and len(res[0]["outs"]) > 3


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher around it starts two ranks itself (the parent never touches the GPU),
    forwards rank 0's single JSON line and reports n_gpus = 2; on a one-GPU box the ranks share the card and the
    exchanges go through gloo (SQMC_BENCH_BACKEND), which is the sharded code path minus the RCCL transport."""
    import json, subprocess
    env = dict(os.environ, SQMC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--equil", "30", "--target", "2e4",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 10 and line["scaling"] == "weak"
    assert "sharded x2" in line["config"]["parallelism"] and line["config"]["devices"] == 1
    assert line["value"] > 0 and -75.9 < line["config"]["projected_energy_Ha"] < -75.4
