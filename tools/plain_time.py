#!/usr/bin/env python3
"""Time of a plain (semistochastic = f) walk step on one GPU: C2 at a given target, kernel stage timers."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sqmc_amd
from sqmc_amd import host as H
FCIDUMP = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
sqmc_amd.set_device(0)
hst = H.ChemHost(FCIDUMP, 8, 4, "d2h")
g = hst.gpu(rng_mode=H.RNG_COUNTER, seed=H.rank_seed((1346, 5634, 6635, 4361), 0), mwalk=int(8 * target + 200000))
s = hst.setup_walk(g, 100, 1000, 0.1)
g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
wk = H.initial_walkers(s, target / 10)
wk["imp_distance"] = np.where(wk["imp_distance"] == 0, 1, wk["imp_distance"]).astype(np.int8)
keep = ~((wk["wt"] == 0) & (wk["initiator"] < 3))
wk = {k: v[keep] for k, v in wk.items()}
g.upload_walkers(wk)
pc = H.PopControl(s.tau, s.e_trial0, target, n_equil_steps=10**9)
w_abs = float(np.abs(wk["wt"]).sum())
def steps(n):
    global w_abs
    for _ in range(n):
        pc.pre_step(w_abs)
        out = g.step(pc.params(min_wt=0.5, semistochastic=0))
        pc.post_step(out)
        w_abs = out[1]
    return out
out = steps(400)
t0 = time.perf_counter(); out = steps(200); dt = (time.perf_counter() - t0) / 200
print("plain walk target %g: %.3f ms/step, nwalk %d, w_abs %.0f, E %.5f" % (target, dt * 1e3, int(out[5]), out[1], out[3] / out[2]))
g.set_timing(2); steps(3)
print("stage timers of the last step (ms):", [(k, round(v, 4)) for k, v in g.timing()])
g.close()
