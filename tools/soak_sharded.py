#!/usr/bin/env python3
"""Soak of the sharded in-library step over the transport double (tests/fake_rccl): 3000 steps with 2 and 3 ranks, the short-list
tail made to give up on one rank only and on all ranks -- the collective re-run must neither hang nor change the walk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_gpu_sharded as T
import torch.multiprocessing as mp
if __name__ == "__main__":
    fake = T._fake_rccl_lib()
    ctx = mp.get_context("spawn")
    for k, (world, env) in enumerate(((3, {"SQMC_BUCKET_FORCE_RETRY": "5", "SQMC_BUCKET_FORCE_RETRY_RANK": "2", "SQMC_BUCKET_HOLDOFF": "0"}), (2, {}), (3, {"SQMC_BUCKET_FORCE_RETRY": "2", "SQMC_BUCKET_HOLDOFF": "0"}))):
        out = os.path.join(ROOT, "gpurun_out", "soak%d" % k); os.makedirs(out, exist_ok=True)
        ps = [ctx.Process(target=T._inlib_multi_worker, args=(r, world, 29700 + k, out, fake, 12000, 3000), kwargs=dict(env=env)) for r in range(world)]
        for p in ps: p.start()
        for p in ps: p.join(500)
        alive = [p for p in ps if p.is_alive()]
        for p in alive: p.terminate()
        res = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)] if not alive and all(p.exitcode == 0 for p in ps) else None
        print("soak", k, "world", world, env, "alive", len(alive), "exit", [p.exitcode for p in ps],
              "tails", [tuple(int(x) for x in r["tail"]) for r in res] if res else None,
              "sums equal", all(np.array_equal(r["outs"][:, :7], res[0]["outs"][:, :7]) for r in res) if res else None,
              "E", float(res[0]["outs"][500:, 3].sum() / res[0]["outs"][500:, 2].sum()) if res else None, flush=True)
