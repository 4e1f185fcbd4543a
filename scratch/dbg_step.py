import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import oracle as O
from conftest import gpu_ctx_from_oracle, FCIDUMP
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s = O.ChemSystem(FCIDUMP, 8, 4, 'd2h', time_sym=False, hf_mode=0)
ws = O.setup_walk(s, 100, 1000, 0.1)
SEED = [1346, 5634, 6635, 4361]
g = gpu_ctx_from_oracle(s, rng_mode=mode, seed=SEED, mwalk=400000)
g.set_projector(ws.prj_counts, ws.prj_indices, ws.prj_values); g.set_ct_table(ws.ct_up, ws.ct_dn, ws.ct_num, ws.ct_den)
wk = O.initial_walkers(ws, 10); g.upload_walkers(wk)
ow = O.OracleWalk(s, ws, wk, 400000, SEED, rng_mode=mode)
pc = O.PopControl(ws.tau, -75.72, 2000)
w_abs = np.abs(wk['wt']).sum()
for it in range(3):
    r = pc.pre_step(w_abs)
    if r != 1.0: ow.scale_projector(r); g.scale_projector(r)
    prm = pc.params()
    st, oc = ow.step(prm); og = g.step(prm)
    print('step', it, 'cpu', oc[[5,7,15,1,0]], 'gpu', og[[5,7,15,1,0]])
    wc, wg = ow.walkers(), g.download_walkers()
    kc = {(int(a),int(b)):i for i,(a,b) in enumerate(zip(wc['up'],wc['dn']))}
    kg = {(int(a),int(b)):i for i,(a,b) in enumerate(zip(wg['up'],wg['dn']))}
    only_g = [k for k in kg if k not in kc]; only_c = [k for k in kc if k not in kg]
    print(' only gpu', len(only_g), 'only cpu', len(only_c))
    for k in only_g[:6]:
        i = kg[k]; print('  G', k, wg['wt'][i], wg['imp_distance'][i], wg['initiator'][i])
    for k in only_c[:6]:
        i = kc[k]; print('  C', k, wc['wt'][i], wc['imp_distance'][i], wc['initiator'][i])
    nd = 0
    for k in kc:
        if k in kg:
            i, j = kc[k], kg[k]
            if wc['wt'][i] != wg['wt'][j] or wc['initiator'][i] != wg['initiator'][j] or wc['imp_distance'][i] != wg['imp_distance'][j]:
                nd += 1
                if nd < 6: print('  diff', k, wc['wt'][i], wg['wt'][j], wc['initiator'][i], wg['initiator'][j], wc['imp_distance'][i], wg['imp_distance'][j])
    print(' ndiff', nd, 'rng', ow.rng_state(), g.rng_state())
    r = pc.post_step(oc)
    if r != 1.0: ow.scale_projector(r); g.scale_projector(r)
    w_abs = oc[1]
    if only_g or only_c or nd: break
