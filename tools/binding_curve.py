#!/usr/bin/env python3
"""BASELINE.json configs[2]: the C2 cc-pVDZ binding curve r = 1.0 ... 2.0 A on one GPU.

For every geometry (FCIDUMPs of the reference's C2_v2z_curve, kept as data fixtures under
tests/golden/): the semistochastic walk at w_abs_gen_target (default 10^6) -> projected energy,
block error and walker-steps/s; HCI (time_sym, eps_var as in the shipped i_1sigma_g decks) + the
deterministic Epstein-Nesbet PT2 -> E_var, E_total.  --cpu adds the same HCI variational energy
from the CPU oracle ("vs CPU energies").  One JSON line per geometry, a table at the end.
This is an auxiliary measurement, not the driver's bench line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch            # noqa: F401  before the HIP library: both must share one libamdhip64
import sqmc_amd
from sqmc_amd import host as H

GEOMS = ["1.0", "1.1", "1.2", "1.24253", "1.3", "1.4", "1.6", "1.8", "2.0"]


def fcidump(r):
    if r == "1.24253":
        return os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
    return os.path.join(ROOT, "tests", "golden", "curve", "C2_r%s_FCIDUMP" % r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", type=float, default=1e6)
    ap.add_argument("--equil", type=int, default=600)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--eps-var", type=float, default=1e-3)
    ap.add_argument("--eps-pt", type=float, default=1e-6)
    ap.add_argument("--geoms", default=",".join(GEOMS))
    ap.add_argument("--cpu", action="store_true", help="HCI variational energy from the CPU oracle as well")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    sqmc_amd.set_device(0)
    rows = []
    for r in args.geoms.split(","):
        rec = {"r_A": float(r)}
        # ---- HCI + PT2 (shipped deck conventions: time_sym=t, z=1, hf_symmetry=1)
        h = H.ChemHost(fcidump(r), 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
        g = h.gpu()
        g.set_hb_tables(*h.hb_tables(g))
        t0 = time.perf_counter()
        up, dn, w, e, hist = H.hci_variational(h, g, args.eps_var, eps_sched=(2 * args.eps_var, 2 * args.eps_var))
        t1 = time.perf_counter()
        de, nconn = H.hci_pt2_determinant_basis(h, up, dn, w[:, 0], float(e[0]), args.eps_pt)
        t2 = time.perf_counter()
        g.close()
        rec.update(hci_ndets=int(len(up)), e_var=float(e[0]), e_total_hci=float(e[0]) + de, hci_s=t1 - t0, pt2_s=t2 - t1)
        if args.cpu:
            from oracle import oracle as O
            O.build()
            sysm = O.ChemSystem(fcidump(r), 8, 4, "d2h", time_sym=True, z=1, hf_mode=1, hf_symmetry=1)
            t0 = time.perf_counter()
            res = O.hci_variational(sysm, args.eps_var, eps_sched=(2 * args.eps_var, 2 * args.eps_var))
            rec.update(e_var_cpu=float(res[3][0]), hci_cpu_s=time.perf_counter() - t0, hci_ndets_cpu=int(len(res[0])))
        # ---- walk (walk deck conventions: time_sym=f)
        hw = H.ChemHost(fcidump(r), 8, 4, "d2h")
        walk = H.GpuWalk(hw, args.target)
        walk.run(args.equil, keep_stats=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        stats, totals = walk.run(args.steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        num, den = stats[:, 3] * np.sign(stats[:, 2]), np.abs(stats[:, 2])
        nb = 20
        blk = [num[i::nb].sum() / den[i::nb].sum() for i in range(nb)]       # interleaved blocks: crude error bar
        rec.update(e_proj=float(num.sum() / den.sum()), e_proj_err=float(np.std(blk) / np.sqrt(nb - 1)), walk_steps=args.steps,
                   occupied_dets=float(totals[5] / args.steps), ms_per_step=dt / args.steps * 1e3, walker_steps_per_s=float(totals[5] / dt))
        walk.close()
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    print("\n  r/A      E_var(HCI)      E_total(HCI+PT2)    E_proj(walk)        +-        dets/step   ms/step" + ("   E_var(CPU)" if args.cpu else ""))
    for x in rows:
        print("%7.5f %16.9f %18.9f %16.9f %10.2e %12.0f %9.3f" % (x["r_A"], x["e_var"], x["e_total_hci"], x["e_proj"], x["e_proj_err"],
                                                                  x["occupied_dets"], x["ms_per_step"]) + ("  %14.9f" % x["e_var_cpu"] if args.cpu else ""))
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
