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
# HBM bytes per launch of the two kernels: read from the PMC summary the profiling script wrote (tools/profile_bench.sh ->
# profiles/*_traffic.json, with the commit it was taken at); never typed in here
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r03_bench_1e5_traffic.json")


def load_traffic():
    try:
        with open(TRAFFIC_JSON) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000, help="timed steps (0.14 ms each at the default size: a short timed region is at the mercy of one host scheduling hiccup)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--target", type=float, default=1e5, help="w_abs_gen_target")
    ap.add_argument("--equil", type=int, default=2000, help="untimed equilibration steps before warmup (2000 steps = 0.15 s of sustained load: the population is at its target after ~300, and the one stall of 60-80 ms this GPU shows 40-75 ms after a process starts loading it -- see DESIGN.md section 9 -- lies behind it)")
    ap.add_argument("--mwalk", type=int, default=0, help="walker capacity of the single-GPU walk (0: the reference's MWALK = 4 (target/min_wt + n_imp))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--system", default="c2", choices=["c2", "heg", "hubbard"], help="c2 = BASELINE.json configs[1] (the metric's config, default); "
                    "heg = the 14-electron 3D electron gas of configs[3]; hubbard = real-space Hubbard U/t=4 at half filling "
                    "(configs[0] lattice by default) -- both auxiliary, no CPU baseline leg")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak (default, the driver's contract): --target is the population per GPU; "
                    "strong: --target is the global population, split over the ranks by determinant ownership")
    ap.add_argument("--hubbard-lattice", default="4x4", help="l_x x l_y (periodic), e.g. 4x4 (configs[0]) or 6x4")
    ap.add_argument("--heg-rs", type=float, default=1.0)
    ap.add_argument("--hf-to-psit", action="store_true", help="auxiliary: the step variant hf_to_psit = .true. (first basis state = Psi_T, all of C(T) resident; "
                    "SURVEY section 8 row f4) on one GPU, c2 or heg; no CPU baseline leg")
    ap.add_argument("--proposal", default="uniform", choices=["uniform", "heatbath"], help="auxiliary: heatbath = proposal_method fast_heatbath (off_diagonal_move_chem_efficient_heatbath, "
                    "two walker slots per child) on the shipped C2 integrals with 10 electrons -- the synthetic system the reference's own check accepts; one GPU, no CPU baseline leg")
    ap.add_argument("--heg-cutoff", type=float, default=2.3, help="plane-wave cutoff radius (2.3 -> 57 orbitals; the GPU path holds at most 64)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process only launches the N ranks (it never touches the GPU, so
        # nothing that has initialised HIP is ever forked or re-executed) and forwards rank 0's JSON line
        sys.exit(launch_ranks(args.gpus))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("SQMC_BENCH_TEST_ONE_RANK_DIES"):     # tests/test_bench_launcher.py: rank 1 dies, rank 0 sits (as in a collective)
        if rank == 1:
            sys.exit(3)
        time.sleep(600)
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
    elif args.proposal == "heatbath":
        hst = H.ChemHost(FCIDUMP, 10, 5, "d2h")
        workload = "C2 cc-pVDZ integrals with 10 electrons (10e,26o, D2h; synthetic) semistochastic walk, fast_heatbath proposal"
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
        if args.proposal == "heatbath":
            kw["proposal"] = "heatbath"
            kw.setdefault("mwalk", int(4 * (args.target / 0.5 + 1000) * 1.5))       # two walker slots per child
        if args.hf_to_psit:
            kw["hf_to_psit"] = True
            kw.setdefault("w_begin", min(args.target, 1e4))
            if args.system == "heg": kw["n_truncate_trial_wf"] = 20
            workload += ", hf_to_psit"
        walk = H.GpuWalk(hst, args.target, seed=H.rank_seed((1346, 5634, 6635, 4361), rank), **kw)
        if multi:
            parallelism = "replicas x%d (sharded path unavailable)" % world
    sharded = isinstance(walk, H.ShardedWalk)
    slowest = None
    shard_split = None

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
        walk.g.set_timing(0 if os.environ.get('SQMC_BENCH_NO_EVENTS') else 1)           # HIP events around the annihilation kernel's launch only (every 8th step), accumulated from here on
        # A host that walks in blocks calls sqmc_gpu_run once per block; with chained runs the last step of a call enqueues the head
        # of the first step of the next one, as every step does for its successor.  Warm-up and timed region are two such calls: the K
        # timed steps then are K tails and K heads (the last one enqueues -- and the closing fence waits for -- the head of a step
        # after the region, as the warm-up's last step did for the region's first).  SQMC_BENCH_NO_CHAIN=1: the first timed step starts cold.
        chain = not sharded and not os.environ.get('SQMC_BENCH_NO_CHAIN')
        if chain: walk.g.set_chained_runs(True)
        walk.run(args.warmup, keep_stats=False)
        fence()
        if sharded: walk.g.shard_time_split(reset=True)
        t0 = time.perf_counter()
        stats, totals = walk.run(args.steps, keep_stats=True)
        fence()
        dt = time.perf_counter() - t0
        # N > 1: where rank 0's host spent the timed steps (head / exchange / tail, and how much of that waiting for the GPU's mail)
        shard_split = {k: round(v, 2) for k, v in walk.g.shard_time_split()[0].items()} if sharded else None
        nwalk_sum, spawn_sum = float(totals[5]) / (world if sharded else 1), float(totals[15])      # sharded: nwalk is the global count
        e_num, e_den = float((stats[:, 3] * np.sign(stats[:, 2])).sum()), float(np.abs(stats[:, 2]).sum())
        slowest = walk.g.slowest_steps()                   # wall clock of the slowest timed steps: host jitter, reruns
        if chain: walk.g.set_chained_runs(False)          # forgets the head the last timed step enqueued
        timers_timed = walk.g.timing()                     # mean ms per k_spawn / k_anneal launch (warm-up and timed steps)
        spawn_ms = dict(timers_timed).get("spawn", float('nan'))
        walk.g.set_timing(2)                                # informational stage breakdown from an untimed tail
        walk.run(20, keep_stats=False)
        stage_ms = dict(walk.g.timing())
    if spawn_ms != spawn_ms:                                # a semistochastic walk times only its annihilation kernel inside the timed region
        spawn_ms = stage_ms.get("spawn", spawn_ms)
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
        default_cfg = (args.system == "c2" and args.target == 1e5 and world == 1 and not args.hf_to_psit and args.proposal == "uniform")
        tr = load_traffic() if default_cfg else None

        def traffic_of(kern):
            """PMC traffic per launch: FETCH_SIZE weighted by the calibration of tools/calib_fetch.hip for this kernel's mix of
            streamed and gathered reads, + WRITE_SIZE; beside it the raw (x1) and guide-default (x2) readings of FETCH_SIZE"""
            if not tr or kern not in tr.get("kernels", {}):
                return None, None
            k = tr["kernels"][kern]
            return k["hbm_bytes_calibrated"], {"fetch_x1": k["hbm_bytes_x1"], "fetch_x2": k["hbm_bytes_x2"], "calibrated": k["hbm_bytes_calibrated"],
                                               "profiled_kernel": k.get("kernel", "")[:60],
                                               "source": os.path.relpath(TRAFFIC_JSON, ROOT), "commit": tr.get("commit")}

        tail = walk.g.tail_stats()
        # which annihilation kernel the timed launches were: the short-list tail's (one block per key range) or the radix tail's
        dom_name = "k_anneal_bucket" if (dom == "k_anneal" and tail[0] - tail[1] > (args.steps + args.warmup) // 2) else dom
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
                       "rccl_ranks": rccl_ranks, "devices": min(world, ndev),
                       "short_list_tail": dict(zip(("bucket_steps", "rerun_through_radix_tail"), tail)),
                       "slowest_steps_us": slowest, "chained_runs": bool(not sharded and not os.environ.get('SQMC_BENCH_NO_CHAIN')),
                       "sharded_step_host_split_us_rank0": shard_split},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic_of(dom)[0], "traffic_detail": traffic_of(dom)[1], "ms_per_launch": dom_ms,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "other_kernels": {"k_spawn": {"ms_per_launch": spawn_ms, "algorithmic_bytes_per_launch": 26.0 * s_avg + 34.0 * n_avg,
                                                       "achieved": (26.0 * s_avg + 34.0 * n_avg) / (spawn_ms * 1e-3) / 1e9, "traffic": traffic_of("k_spawn")[0]}},
                         "whole_step": {"algorithmic_bytes": step_bytes, "achieved": step_bytes / (dt / args.steps) / 1e9,
                                        "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
                         "stage_ms_per_step": stage_ms},
        }
        if not args.no_cpu_baseline and world == 1 and args.system == "c2" and not args.hf_to_psit and args.proposal == "uniform":
            cpu = cpu_leg(walk, hst, n_avg)
            line["cpu_baseline"] = cpu["one_core"]
            line["cpu_baseline_all_cores"] = cpu["all_cores"]
            # the other half of the metric: projected-energy error of the GPU path against the CPU path on identical input
            line["energy_error_Ha"] = cpu["energy_error_Ha"]
            line["energy_check"] = cpu["energy_check"]
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
    # A rank that dies after the rendezvous (out of memory, a HIP error) leaves its peers inside a collective: every child is polled,
    # the first non-zero exit takes the others down (terminate, then kill), and the whole run has a deadline (SQMC_BENCH_TIMEOUT_S).
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("SQMC_BENCH_TIMEOUT_S", "3000"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad_now = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad_now or time.time() > deadline:
            failed = bad_now or [(-1, "timeout")]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10.0
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10.0)
    out = b"".join(c for c in chunks if c)
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = failed or [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %r\n" % bad)
        return 1
    return 0


def host_cores():
    """CPU cores this process may actually use: the cgroup quota where there is one (a GPU box hands a 16-core share of a
    256-thread host to a one-GPU job), else the affinity mask; never more than the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            tok = open(path).read().split()
            quota = float(tok[0]) if tok[0] != "max" else -1.0
            period = float(tok[1]) if len(tok) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return int(os.environ.get("SQMC_BENCH_CORES", min(n, 16)))      # 16: the CPU share of a one-GPU box when no quota is visible


def cpu_leg(walk, hst, n_avg, budget_s=10.0):
    """The CPU side of the line, on this box's host cores (test infrastructure used as the checker and the baseline, never
    as the product): the single-thread C restatement of the reference step continues the GPU run's equilibrated
    population while a SECOND GPU context walks the same population in lock-step (same seed, same step counter, same
    parameters), so that the projected energies of the two paths over the same steps can be subtracted; then the
    all-host-cores variant (OpenMP over walkers) from the same population; then a short REPLAY-discipline run (the
    reference's single rannyu stream) for the per-step deviation at small N."""
    import numpy as np
    import sqmc_amd
    from sqmc_amd import host as H
    from oracle import oracle as O
    O.build()
    sysm = O.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=False, hf_mode=0)
    seed = [1346, 5634, 6635, 4361]
    w = walk.g.download_walkers()
    # permanent-initiator signs travel with the walkers on the GPU; the HF det is the only one here
    w["perm_sign"] = np.where(w["initiator"] == 3, 1, 0).astype(np.int8)
    s = walk.setup
    prm = walk.pc.params(min_wt=walk.min_wt)
    ratio = walk.pc.tau / s.tau

    def gpu_replica(rng_mode, walkers, mwalk):
        g = hst.gpu(rng_mode=rng_mode, seed=tuple(seed), mwalk=mwalk)
        g.set_projector(s.prj_counts, s.prj_indices, s.prj_values)
        g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
        g.upload_walkers(walkers)
        return g

    # ---- one core, in lock-step with a GPU replica
    O.lib().orc_set_threads(1)
    ow = O.OracleWalk(sysm, s, w, walk.g.mwalk, seed, rng_mode=1)
    g2 = gpu_replica(sqmc_amd.RNG_COUNTER, w, walk.g.mwalk)
    if ratio != 1.0:
        ow.scale_projector(ratio); g2.scale_projector(ratio)
    t_cpu, n, nw = 0.0, 0, 0.0
    num_c = den_c = num_g = den_g = 0.0
    dev_step, same_count = 0.0, True
    while t_cpu < budget_s and n < 200:
        t0 = time.perf_counter()
        st, out = ow.step(prm)
        t_cpu += time.perf_counter() - t0
        if st != 0:
            break
        og = g2.step(prm)
        nw += out[5]; n += 1
        num_c += out[3]; den_c += out[2]; num_g += og[3]; den_g += og[2]
        dev_step = max(dev_step, abs(og[3] / og[2] - out[3] / out[2]))
        same_count = same_count and og[5] == out[5] and og[15] == out[15]
    wg, wc = g2.download_walkers(), ow.walkers()
    identical = bool(same_count and np.array_equal(wg["up"], wc["up"]) and np.array_equal(wg["dn"], wc["dn"]) and np.array_equal(wg["wt"], wc["wt"]))
    g2.close(); ow.close()
    one = {"value": nw / t_cpu, "unit": "walker-steps/s", "cores": 1, "kind": "port",
           "sample": "%d steps of the single-thread C restatement (%.1f s) continuing the GPU run's equilibrated population of %.0f determinants" % (n, t_cpu, n_avg)}
    e_err = abs(num_g / den_g - num_c / den_c)
    check = {"steps": n, "e_proj_gpu_Ha": num_g / den_g, "e_proj_cpu_Ha": num_c / den_c, "max_per_step_dev_Ha": dev_step,
             "walkers_bit_identical_after": identical, "rng": "counter",
             "what": "|E_proj(GPU) - E_proj(CPU restatement)| over the same %d steps from the same %.0f-determinant population, same seed (target 1e-6 Ha)" % (n, n_avg)}
    # ---- all host cores
    cores = host_cores()
    O.lib().orc_set_threads(cores)
    ow = O.OracleWalk(sysm, s, w, walk.g.mwalk, seed, rng_mode=1)
    if ratio != 1.0:
        ow.scale_projector(ratio)
    ow.step(prm)                                             # thread pool start-up is not the measurement
    t_mt, n_mt, nw_mt = 0.0, 0, 0.0
    while t_mt < 0.6 * budget_s and n_mt < 400:
        t0 = time.perf_counter()
        st, out = ow.step(prm)
        t_mt += time.perf_counter() - t0
        if st != 0:
            break
        nw_mt += out[5]; n_mt += 1
    ow.close()
    O.lib().orc_set_threads(1)
    allc = {"value": nw_mt / t_mt if t_mt > 0 else None, "unit": "walker-steps/s", "cores": cores, "kind": "port",
            "sample": "%d steps (%.1f s) of the same restatement with its spawn loop, sort and permutations on %d OpenMP threads (bit-identical results); "
                      "the linear merge / rounding / estimator scans stay serial" % (n_mt, t_mt, cores)}
    # ---- REPLAY discipline at small N: the reference's single rannyu stream, draw for draw
    wk = H.initial_walkers(s, 200)
    ow = O.OracleWalk(sysm, s, wk, 400000, seed, rng_mode=0)
    g3 = gpu_replica(sqmc_amd.RNG_REPLAY, wk, 400000)
    pc = H.PopControl(s.tau, -75.72, 4000)
    w_abs, dev_r, ident_r = float(np.abs(wk["wt"]).sum()), 0.0, True
    for _ in range(80):
        r = pc.pre_step(w_abs)
        if r != 1.0:
            ow.scale_projector(r); g3.scale_projector(r)
        p2 = pc.params()
        st, oc = ow.step(p2)
        og = g3.step(p2)
        dev_r = max(dev_r, abs(og[3] / og[2] - oc[3] / oc[2]))
        ident_r = ident_r and og[5] == oc[5]
        r = pc.post_step(oc)
        if r != 1.0:
            ow.scale_projector(r); g3.scale_projector(r)
        w_abs = oc[1]
    ident_r = bool(ident_r and g3.rng_state() == ow.rng_state() and np.array_equal(g3.download_walkers()["wt"], ow.walkers()["wt"]))
    g3.close(); ow.close()
    check["replay_small_n"] = {"steps": 80, "w_abs_gen_target": 4000, "max_per_step_dev_Ha": dev_r, "bit_identical_incl_rng_state": ident_r}
    return {"one_core": one, "all_cores": allc, "energy_error_Ha": e_err, "energy_check": check}


if __name__ == "__main__":
    main()
