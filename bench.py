#!/usr/bin/env python3
"""bench.py -- walker-steps/s of the semistochastic walk on C2 cc-pVDZ (BASELINE.json
configs[1]: r=1.24253 A, w_abs_gen_target=10^5, uniform2 proposal, size_deterministic=1000).

A "step" is one full MC step (spawn, death, deterministic projection, sort, annihilation,
rounding, estimators) over the whole walker population resident in HBM; metric =
sum over timed steps of occupied determinants after the merge (the nwalk column of the
reference's walkalize file, do_walk.f90:2930) / wall time.

N>1: weak scaling.  The global target grows with the number of GPUs, determinants are sharded
over ranks by hash ownership (as the reference shards them over MPI ranks) and every step runs
three exchanges over RCCL issued by the library itself (sqmc_gpu_shard_run): all-reduce of the
deterministic-space weights, all-to-all of the spawned walkers, all-reduce of the seven sums.
If the library's communicator cannot be created the same step runs with the exchanges driven
from Python through torch.distributed; if that fails too, independent replicas.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FCIDUMP = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# HBM bytes per launch from the PMC passes in profiles/ (FETCH_SIZE doubled as the guide prescribes for
# gfx950, + WRITE_SIZE, KiB -> bytes), default configuration only
TRAFFIC_K_ANNEAL = 3.63e7     # profiles/r01_bench_1e5_rocprof_summary.txt: (2*12508.1 + 10399.5) KiB (24 B per surviving walker of it: the next step's gate)
TRAFFIC_K_SPAWN = 1.97e7      # same file: (2*6500.5 + 6252.1) KiB


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000, help="timed steps (0.14 ms each at the default size: a short timed region is at the mercy of one host scheduling hiccup)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--target", type=float, default=1e5, help="w_abs_gen_target")
    ap.add_argument("--equil", type=int, default=400, help="untimed equilibration steps before warmup")
    ap.add_argument("--mwalk", type=int, default=0, help="walker capacity of the single-GPU walk (0: the reference's MWALK = 4 (target/min_wt + n_imp))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--system", default="c2", choices=["c2", "heg", "hubbard"], help="c2 = BASELINE.json configs[1] (the metric's config, default); "
                    "heg = the 14-electron 3D electron gas of configs[3]; hubbard = real-space Hubbard U/t=4 at half filling "
                    "(configs[0] lattice by default) -- both auxiliary, no CPU baseline leg")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak (default, the driver's contract): --target is the population per GPU; "
                    "strong: --target is the global population, split over the ranks by determinant ownership")
    ap.add_argument("--hubbard-lattice", default="4x4", help="l_x x l_y (periodic), e.g. 4x4 (configs[0]) or 6x4")
    ap.add_argument("--heg-rs", type=float, default=1.0)
    ap.add_argument("--heg-cutoff", type=float, default=2.3, help="plane-wave cutoff radius (2.3 -> 57 orbitals; the GPU path holds at most 64)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process only launches the N ranks (it never touches the GPU, so
        # nothing that has initialised HIP is ever forked or re-executed) and forwards rank 0's JSON line
        sys.exit(launch_ranks(args.gpus))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size counts\n" % (args.gpus, world))
    # stdout carries the one JSON line and nothing else: libraries that greet on fd 1 (RCCL prints its version banner there
    # when a communicator is made) are sent to stderr for the length of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if "SQMC_BENCH_DEVICE" in os.environ:       # rehearsal of the N>1 path on a one-GPU box
        local = int(os.environ["SQMC_BENCH_DEVICE"])
    import numpy as np
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    local = local % ndev                          # more ranks than devices (a rehearsal on a one-GPU box): ranks share cards
    import sqmc_amd
    from sqmc_amd import host as H
    dist = None
    multi = world > 1 or bool(os.environ.get("SQMC_BENCH_FORCE_SHARDED"))     # rehearsal of the N>1 code path with one rank
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local)
        backend = os.environ.get("SQMC_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on a one-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    comm_dev = "cuda" if (multi and backend == "nccl") else "cpu"
    sqmc_amd.set_device(local)

    if args.system == "heg":
        hst = H.HegHost(3, args.heg_rs, 14, 7, args.heg_cutoff)
        workload = "3D HEG r_s=%g, 14 electrons in %d plane waves, semistochastic walk" % (args.heg_rs, hst.norb)
    elif args.system == "hubbard":
        lx, ly = (int(v) for v in args.hubbard_lattice.lower().split("x"))
        hst = H.HubbardHost(lx, ly, True, lx * ly // 2, lx * ly // 2, 1.0, 4.0)
        workload = "%dx%d Hubbard U/t=4 half filling (periodic), real space (hubbard2), semistochastic walk" % (lx, ly)
    else:
        hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
        workload = "C2 cc-pVDZ r=1.24253 (8e,26o, D2h) semistochastic walk, uniform2 proposal"

    g_target = args.target * world if args.scaling == "weak" else args.target      # global population of the sharded walk

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    parallelism = "single GPU"
    walk, rccl_ranks = None, None
    if multi:
        # weak scaling: the global target grows with the number of GPUs, determinants are sharded by
        # hash ownership and spawns cross ranks through one RCCL all-to-all per step
        try:
            skw = dict(w_begin=min(g_target, 1e4), n_truncate_trial_wf=1, size_deterministic=500) if args.system == "heg" else {}
            if args.system == "hubbard":
                skw = dict(w_begin=min(g_target, 1e4), n_truncate_trial_wf=20, size_deterministic=500, tau_multiplier=0.5)
            walk = H.ShardedWalk(hst, g_target, rank, world, device_index=local, seed=(1346, 5634, 6635, 4361), **skw)
            ok = torch.ones(1, device=comm_dev)
        except Exception as exc:                      # keep the scaling run alive: independent replicas
            sys.stderr.write("rank %d: sharded set-up failed (%r); falling back to replicas\n" % (rank, exc))
            ok = torch.zeros(1, device=comm_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() < 1:
            if walk is not None:
                walk.close()
            walk = None
        if walk is not None:
            parallelism = "sharded x%d (hash-owned determinants), exchanges driven through torch.distributed" % world
            if (backend == "nccl" or os.environ.get("SQMC_RCCL_LIB")) and not os.environ.get("SQMC_BENCH_NO_INLIB"):     # SQMC_RCCL_LIB: rehearsal with the transport double of tests/fake_rccl
                try:
                    walk.attach_rccl()
                    walk.step()
                    ok = torch.ones(1, device=comm_dev)
                except Exception as exc:
                    sys.stderr.write("rank %d: in-library RCCL exchange unavailable (%r)\n" % (rank, exc))
                    ok = torch.zeros(1, device=comm_dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() >= 1:
                    rccl_ranks = walk.g.comm_size()
                    parallelism = "sharded x%d (hash-owned determinants), in-library RCCL (communicator of %d ranks): all-reduce + all-to-all of spawns + all-reduce per step" % (world, rccl_ranks)
                else:                                     # same walk, exchanges from Python
                    walk.close()
                    walk = H.ShardedWalk(hst, g_target, rank, world, device_index=local, seed=(1346, 5634, 6635, 4361), **skw)
    if walk is None:
        kw = dict(w_begin=min(args.target, 1e4), n_truncate_trial_wf=1, size_deterministic=500) if args.system == "heg" else {}
        if args.system == "hubbard":
            kw = dict(w_begin=min(args.target, 1e4), n_truncate_trial_wf=20, size_deterministic=500, tau_multiplier=0.5)
        if args.mwalk: kw["mwalk"] = args.mwalk
        walk = H.GpuWalk(hst, args.target, seed=H.rank_seed((1346, 5634, 6635, 4361), rank), **kw)
        if multi:
            parallelism = "replicas x%d (sharded path unavailable)" % world
    sharded = isinstance(walk, H.ShardedWalk)

    if sharded and not walk.in_library:
        for _ in range(args.equil + args.warmup):
            walk.step()
        walk.g.set_timing(1)
        fence()
        t0 = time.perf_counter()
        rows = [walk.step() for _ in range(args.steps)]
        fence()
        dt = time.perf_counter() - t0
        stats = np.array(rows)
        # out[5] (nwalk) is already the all-reduced global count; spawns (out[15]) are rank-local
        nwalk_sum, spawn_sum = float(stats[:, 5].sum()) / world, float(stats[:, 15].sum())
        e_num, e_den = float((stats[:, 3] * np.sign(stats[:, 2])).sum()), float(np.abs(stats[:, 2]).sum())
        timers_timed = walk.g.timing()
        spawn_ms = dict(timers_timed).get("spawn", float("nan"))
        walk.g.set_timing(2)
        for _ in range(10):
            walk.step()
        stage_ms = dict(walk.g.timing())
    else:
        # equilibration + warmup (untimed), then EXACTLY --steps timed steps inside sqmc_gpu_run
        walk.run(args.equil, keep_stats=False)
        walk.run(args.warmup, keep_stats=False)
        walk.g.set_timing(0 if os.environ.get('SQMC_BENCH_NO_EVENTS') else 1)           # HIP events around the k_spawn / k_anneal launches only, accumulated over the timed steps
        fence()
        t0 = time.perf_counter()
        stats, totals = walk.run(args.steps, keep_stats=True)
        fence()
        dt = time.perf_counter() - t0
        nwalk_sum, spawn_sum = float(totals[5]) / (world if sharded else 1), float(totals[15])      # sharded: nwalk is the global count
        e_num, e_den = float((stats[:, 3] * np.sign(stats[:, 2])).sum()), float(np.abs(stats[:, 2]).sum())
        timers_timed = walk.g.timing()                     # mean ms per k_spawn / k_anneal launch over the K timed steps
        spawn_ms = dict(timers_timed).get("spawn", float('nan'))
        walk.g.set_timing(2)                                # informational stage breakdown from an untimed tail
        walk.run(20, keep_stats=False)
        stage_ms = dict(walk.g.timing())
    tot = torch.tensor([nwalk_sum, spawn_sum, dt], dtype=torch.float64, device=comm_dev)
    if multi:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot)
        dt = float(mx[2])
    nwalk_all, spawn_all = float(tot[0]), float(tot[1])

    if rank == 0:
        value = nwalk_all / dt
        # Dominant kernel = the longest one on the step's critical path: k_anneal (annihilation + rounding + compaction
        # + estimator sums, one launch per step).  Its timer is the pair of HIP events hipExtLaunchKernelGGL attaches
        # to that launch on the library's stream: the kernel's own start/stop timestamps, what rocprofv3
        # --kernel-trace reports.  Algorithmic bytes per launch: 68 B per occupied determinant
        # (walker read + write) + 58 B per child proposal (26 B read-back + 32 B annihilation slot; the other 26 B
        # of a spawn's 84 B are its write in k_spawn, reported beside it).
        n_avg, s_avg = nwalk_sum / args.steps, spawn_sum / args.steps
        timers = {k: v for k, v in dict(timers_timed).items() if v == v}
        if "anneal" not in timers and "anneal" in stage_ms:      # fewer timed steps than the event stride: take the untimed tail's launches
            timers["anneal"], spawn_ms = stage_ms["anneal"], stage_ms.get("spawn", spawn_ms)
        if "anneal" in timers:
            dom, dom_ms, dom_bytes = "k_anneal", timers["anneal"], 68.0 * n_avg + 58.0 * s_avg
        else:                                  # semistochastic = f keeps the unfused tail: k_spawn is then the longest single kernel
            dom, dom_ms, dom_bytes = "k_spawn", spawn_ms, 26.0 * s_avg + 34.0 * n_avg
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        step_bytes = 68.0 * n_avg + 84.0 * s_avg
        default_cfg = (args.system == "c2" and args.target == 1e5 and world == 1)
        line = {
            "metric": "walker-steps/sec", "value": value, "unit": "walker-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if sharded else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload + ", w_abs_gen_target=%g, %s, min_wt 0.5, r_initiator 1" % (
                                   g_target if sharded else args.target,
                                   {"c2": "size_deterministic=1000, Psi_T 100 dets, tau_multiplier 0.1", "heg": "size_deterministic=500, Psi_T 1 det, tau_multiplier 0.1",
                                    "hubbard": "size_deterministic=500, Psi_T 20 dets, tau_multiplier 0.5"}[args.system]),
                       "occupied_dets_per_step": n_avg, "spawns_per_step": s_avg, "spawns_per_s": spawn_all / dt,
                       "projected_energy_Ha": e_num / e_den, "rng": "counter", "parallelism": parallelism,
                       "rccl_ranks": rccl_ranks, "devices": min(world, ndev)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": (TRAFFIC_K_ANNEAL if dom == "k_anneal" else TRAFFIC_K_SPAWN) if default_cfg else None, "ms_per_launch": dom_ms,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "other_kernels": {"k_spawn": {"ms_per_launch": spawn_ms, "algorithmic_bytes_per_launch": 26.0 * s_avg + 34.0 * n_avg,
                                                       "achieved": (26.0 * s_avg + 34.0 * n_avg) / (spawn_ms * 1e-3) / 1e9, "traffic": TRAFFIC_K_SPAWN if default_cfg else None}},
                         "whole_step": {"algorithmic_bytes": step_bytes, "achieved": step_bytes / (dt / args.steps) / 1e9,
                                        "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                         "stage_ms_per_step": stage_ms},
        }
        if not args.no_cpu_baseline and world == 1 and args.system == "c2":
            line["cpu_baseline"] = cpu_baseline(walk, hst, n_avg)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    walk.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(n):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun sets them), rendezvous on
    127.0.0.1.  Rank 0's stdout (the JSON line) is forwarded; the exit code is non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %r\n" % bad)
        return 1
    return 0


def cpu_baseline(walk, hst, n_avg, budget_s=15.0):
    """The oracle (single-thread C restatement of the reference step) on this box's host
    cores, started from the SAME equilibrated population, for a bounded number of steps."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    sysm = O.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=False, hf_mode=0)
    w = walk.g.download_walkers()
    # permanent-initiator signs travel with the walkers on the GPU; the HF det is the only one here
    w["perm_sign"] = np.where(w["initiator"] == 3, 1, 0).astype(np.int8)
    s = walk.setup
    ow = O.OracleWalk(sysm, s, w, walk.g.mwalk, [1346, 5634, 6635, 4361], rng_mode=1)
    ow.scale_projector(walk.pc.tau / s.tau)
    prm = walk.pc.params(min_wt=walk.min_wt)
    t0, n, nw = time.perf_counter(), 0, 0.0
    while time.perf_counter() - t0 < budget_s and n < 200:
        st, out = ow.step(prm)
        if st != 0:
            break
        nw += out[5]; n += 1
    dt = time.perf_counter() - t0
    ow.close()
    return {"value": nw / dt, "unit": "walker-steps/s", "cores": 1, "kind": "port",
            "sample": "%d oracle steps (%.1f s) continuing the GPU run's equilibrated population of %.0f determinants" % (n, dt, n_avg)}


if __name__ == "__main__":
    main()
