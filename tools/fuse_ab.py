# A/B: run loop with and without the gate fused into the annihilation kernel -- per-step sums and final walkers
import os, sys, subprocess, numpy as np
ROOT = os.getcwd(); os.makedirs("gpurun_out", exist_ok=True)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    from oracle import oracle as O
    O.build()
    import conftest as CF
    from sqmc_amd import host as H
    sysm = O.ChemSystem(CF.FCIDUMP, 8, 4, "d2h", time_sym=False, hf_mode=0)
    setup = O.setup_walk(sysm, 100, 1000, 0.1)
    g = CF.gpu_ctx_from_oracle(sysm, rng_mode=1, seed=[1346, 5634, 6635, 4361], mwalk=400000)
    g.set_projector(setup.prj_counts, setup.prj_indices, setup.prj_values)
    g.set_ct_table(setup.ct_up, setup.ct_dn, setup.ct_num, setup.ct_den)
    wk = O.initial_walkers(setup, 100)
    g.upload_walkers(wk)
    pc = H.PopControl(setup.tau, -75.72, 4000, n_equil_steps=40)
    cpc = pc.to_c(float(np.abs(wk["wt"]).sum()))
    stats, totals = g.run(cpc, 600)
    w = g.download_walkers()
    np.savez(sys.argv[1], stats=stats, up=w["up"], dn=w["dn"], wt=w["wt"], reached=np.array([cpc.reached_w_abs_gen]))
    raise SystemExit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "gpurun_out/ab_fused.npz"], env=env)
env["SQMC_NO_GATE_FUSION"] = "1"
subprocess.check_call([sys.executable, __file__, "gpurun_out/ab_unfused.npz"], env=env)
a = np.load("gpurun_out/ab_fused.npz"); b = np.load("gpurun_out/ab_unfused.npz")
print("reached", a["reached"], b["reached"], "nwalk", len(a["up"]), len(b["up"]))
sa, sb = a["stats"], b["stats"]
bad = np.where((sa != sb).any(axis=1))[0]
print("first differing step:", bad[:5], "of", len(sa))
if len(bad):
    k = bad[0]
    print("fused  ", sa[k]); print("unfused", sb[k])
    if k > 0: print("step before equal:", np.array_equal(sa[k-1], sb[k-1]), sa[k-1][:8])
same = len(a["up"]) == len(b["up"]) and np.array_equal(a["up"], b["up"]) and np.array_equal(a["dn"], b["dn"]) and np.array_equal(a["wt"], b["wt"])
print("walkers equal:", same)
if len(bad) or not same or int(a["reached"][0]) != 2: raise SystemExit(1)
