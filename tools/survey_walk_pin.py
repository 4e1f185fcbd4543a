#!/usr/bin/env python3
"""The one external walk observable on file -- BASELINE.md section 2's reference run of the walk smoke deck (SURVEY appendix A:
C2 cc-pVDZ, uniform2, semistochastic, target 1e4, 4 x 100-step blocks after equilibration in sets of 2 blocks) -- against
the GPU path run with the same deck and schedule under several seeds: block counts, populations, `Energy=`, and the
singles : doubles mix of the proposals on the equilibrated population.  Writes gpurun_out/r02_survey_walk_pin.json (kept as profiles/r02_survey_walk_pin.json)."""
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = dict(energy=-75.71812889, energy_err=0.00008127, nwalk=16400, nwalk_before_merge=32500, singles=621940, doubles=37269471,
           n_imp=1002, n_ct=76900, tau=0.005314)


def main():
    from sqmc_amd.walk_run import parse_walk_deck, run_walk
    fcidump = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
    text = open(os.path.join(ROOT, "tests", "golden", "C2_r1.24253_i_walk_survey")).read()
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rows = []
    for k in range(nseeds):
        deck = parse_walk_deck(text)
        deck["irand_seed"][1][3] = (deck["irand_seed"][1][3] + 2 * k) % 10000
        buf = io.StringIO()
        r = run_walk(deck, fcidump, out=buf)
        blk = [float(l.split("e_blk=")[1].split()[0]) for l in buf.getvalue().splitlines() if l.startswith("iblk,")]
        rows.append(dict(seed=deck["irand_seed"][1], n_equil_sets=r["n_equil_sets"], n_blocks_total=r["n_blocks_total"], energy=r["energy"],
                         energy_err=r["energy_err"], nwalk_av=r["nwalk_av"], e_blk=blk))
        print("seed %s: %d equilibration sets, %d blocks, Energy= %.8f(%d)  nwalk_av %.0f  last blocks %s" % (
            deck["irand_seed"][1], r["n_equil_sets"], r["n_blocks_total"], r["energy"], round(1e8 * r["energy_err"]), r["nwalk_av"],
            " ".join("%.4f" % e for e in blk[-6:])), flush=True)
    e = np.array([r["energy"] for r in rows])
    out = dict(reference=REF, runs=rows, mean=float(e.mean()), scatter=float(e.std(ddof=1)) if len(e) > 1 else 0.0)
    print("GPU runs: mean %.6f, run-to-run scatter %.6f; reference %.6f(%d)" % (out["mean"], out["scatter"], REF["energy"], round(1e8 * REF["energy_err"])))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)          # copy to profiles/ afterwards: only gpurun_out/ travels back from the GPU box
    with open(os.path.join(ROOT, "gpurun_out", "r02_survey_walk_pin.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
