"""Lock-step run of the hf_to_psit step on the GPU and in the oracle that, at the first step whose sums differ, compares the two lists
in front of the merge (residents, spawn records): where a difference comes from.  Debugging aid.
usage: python tools/psit_debug.py [nsteps] [rng_mode] [w_begin] [target]"""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
import sqmc_amd
from conftest import gpu_ctx_from_oracle, FCIDUMP

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w_begin = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
target = float(sys.argv[4]) if len(sys.argv) > 4 else 20000.0
SEED = [1346, 5634, 6635, 4361]
O.build()
sysm = O.ChemSystem(FCIDUMP, 8, 4, "d2h", time_sym=False, hf_mode=0)
s = O.setup_walk(sysm, 100, 1000, 0.1, rediagonalize=True)
q = O.psit_setup(sysm, s)
wk = O.initial_walkers_psit(s, q, w_begin)
ow = O.OracleWalk(sysm, s, wk, 600000, SEED, rng_mode=mode, psit=q)
ow.debug_premerge()
g = gpu_ctx_from_oracle(sysm, rng_mode=mode, seed=SEED, mwalk=600000)
g.set_projector(q.prj_counts, q.prj_indices, q.prj_values)
g.set_ct_table(s.ct_up, s.ct_dn, s.ct_num, s.ct_den)
g.set_hf_to_psit(q.loc_psit + 1, q.cdet, q.diag_elems, 1)
g.upload_walkers(wk)
pc = O.PopControl(s.tau, s.e_trial0, target)
w_abs = float(np.abs(wk["wt"]).sum())
n_ct = len(s.ct_up)


def gpu_premerge():
    f = g.L.sqmc_gpu_debug_premerge
    f.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8
    n0, ns = C.c_int64(), C.c_int64()
    z = np.zeros(1)
    f(g.h, 0, C.byref(n0), C.byref(ns), None, None, None, None, None, None)
    n0, ns = n0.value, ns.value
    rw, su, sd, sw = np.zeros(n0), np.zeros(ns, np.uint64), np.zeros(ns, np.uint64), np.zeros(ns)
    sdd, si = np.zeros(ns, np.int8), np.zeros(ns, np.int8)
    a, b = C.c_int64(), C.c_int64()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    f(g.h, n0 + ns, C.byref(a), C.byref(b), p(rw), p(su), p(sd), p(sw), p(sdd), p(si))
    return n0, rw, su, sd, sw, sdd, si


for it in range(nsteps):
    r = pc.pre_step(w_abs)
    if r != 1.0:
        ow.scale_projector(r); g.scale_projector(r)
    prm = pc.params()
    st, oc = ow.step(prm)
    og = g.step(prm)
    bad = [k for k in range(16) if abs(og[k] - oc[k]) > 1e-9 * max(1.0, abs(oc[k]))]
    if st or bad:
        print("step", it, "status", st, "differing sums", bad, [(og[k], oc[k]) for k in bad])
        po, n0o = ow.premerge()
        n0, rw, su, sd, sw, sdd, si = gpu_premerge()
        print("residents", n0, n0o, "spawn slots", len(su), "oracle spawns", len(po["up"]) - n0o, "gpu nonzero", np.count_nonzero(sw))
        dr = np.nonzero(rw != po["wt"][:n0o])[0] if n0 == n0o else None
        print("resident weights differ at", None if dr is None else dr[:10], "of n_ct", n_ct)
        if dr is not None and len(dr):
            for i in dr[:10]:
                print("   slot", i, "gpu", rw[i], "oracle", po["wt"][i], "impd", po["imp_distance"][i], "init", po["initiator"][i])
        keep = sw != 0
        gs = sorted(zip(su[keep].tolist(), sd[keep].tolist(), sw[keep].tolist(), sdd[keep].tolist(), si[keep].tolist()))
        os_ = sorted(zip(po["up"][n0o:].tolist(), po["dn"][n0o:].tolist(), po["wt"][n0o:].tolist(), po["imp_distance"][n0o:].tolist(), po["initiator"][n0o:].tolist()))
        print("spawn multisets equal:", gs == os_, len(gs), len(os_))
        if gs != os_:
            sg, so = set(gs), set(os_)
            print("  only gpu:", sorted(sg - so)[:8]); print("  only oracle:", sorted(so - sg)[:8])
        print("sum |w| residents gpu/oracle", np.abs(rw).sum(), np.abs(po["wt"][:n0o]).sum(), " spawns", np.abs(sw).sum(), np.abs(po["wt"][n0o:]).sum())
        break
    r = pc.post_step(oc)
    if r != 1.0:
        ow.scale_projector(r); g.scale_projector(r)
    w_abs = oc[1]
else:
    print("no difference in", nsteps, "steps; nwalk", oc[5])
