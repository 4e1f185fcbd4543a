#!/usr/bin/env python3
"""Regenerates the small fixtures in this directory.

  rannyu.json            first 1000 rannyu() reals for both seeds of the reference's input
                         line 1, produced by the REAL reference object code (oracle/_ref,
                         built from /root/reference/src/rannyu.f90 unmodified) -- needs
                         /root/reference, i.e. only runs in the build container.
  walk_c2_5steps.json    five semistochastic steps on C2 cc-pVDZ from the CPU oracle (the
                         reference holds no walk fixture; this one freezes the restatement).
  C2_r1.24253_FCIDUMP    copied data file of the reference (C2_v2z_curve/r1.24253/FCIDUMP).
  curve/C2_r*_FCIDUMP    the integral files of the other eight geometries of the same directory
                         (C2_v2z_curve/r*/FCIDUMP): input data of BASELINE.json configs[2].
  C2_r1.24253_i_1sigma_g, C2_r1.24253_i_3pi_u
                         the two input decks shipped next to that FCIDUMP (16 lines of run
                         parameters each: input data, read by python -m sqmc_amd.run in the tests).

  heg_e2e_i_det, heg_e2e_i_st
                         the two input decks of the reference's end-to-end test directory
                         (src/e2e_tests/heg/i_det, i_st: run parameters, 21 and 25 lines); the numbers
                         of the golden outputs next to them (o_det_ref, o_st_ref) are transcribed into
                         the tests with their line numbers.

  psit_c2_8steps.json    eight steps of the hf_to_psit step variant on the same molecule from the CPU
                         oracle, both RNG disciplines (README_hf_to_psit.md: what that restatement is and is not pinned by).

Data files are copied byte for byte (cp); nothing here is reference source code.
"""
import ctypes as C
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def rannyu_fixture():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsqmc_ref.so"), mode=os.RTLD_LAZY)
    recs = []
    for seed in ([2726, 5165, 6543, 6524], [1346, 5634, 6635, 4361]):
        R.ref_setrn((C.c_int * 4)(*seed))
        out = np.zeros(1000)
        R.ref_rannyu_fill(C.c_int(1000), out.ctypes.data_as(C.c_void_p))
        recs.append({"seed": seed, "hex": [float(x).hex() for x in out]})
    json.dump(recs, open(os.path.join(HERE, "rannyu.json"), "w"))


def walk_fixture():
    fcidump = os.path.join(HERE, "C2_r1.24253_FCIDUMP")
    s = O.ChemSystem(fcidump, 8, 4, "d2h", time_sym=False, hf_mode=0)
    ws = O.setup_walk(s, 100, 1000, 0.1, coeffs="pt1")
    gold = {"w_abs_gen_begin": 50, "seed": [1346, 5634, 6635, 4361], "e_trial": -75.72, "w_target": 2000, "steps": []}
    wk = O.initial_walkers(ws, gold["w_abs_gen_begin"])
    ow = O.OracleWalk(s, ws, wk, 200000, gold["seed"], rng_mode=0)
    pc = O.PopControl(ws.tau, gold["e_trial"], gold["w_target"])
    w_abs = float(np.abs(wk["wt"]).sum())
    for _ in range(5):
        r = pc.pre_step(w_abs)
        if r != 1.0: ow.scale_projector(r)
        st, out = ow.step(pc.params())
        assert st == 0
        gold["steps"].append([float(x).hex() for x in out])
        r = pc.post_step(out)
        if r != 1.0: ow.scale_projector(r)
        w_abs = out[1]
    w = ow.walkers()
    gold["rng_after"] = ow.rng_state()
    gold["det_checksum"] = int(np.bitwise_xor.reduce(w["up"] * np.uint64(0x9E3779B97F4A7C15) + w["dn"]))
    json.dump(gold, open(os.path.join(HERE, "walk_c2_5steps.json"), "w"))


def psit_fixture():
    fcidump = os.path.join(HERE, "C2_r1.24253_FCIDUMP")
    s = O.ChemSystem(fcidump, 8, 4, "d2h", time_sym=False, hf_mode=0)
    ws = O.setup_walk(s, 100, 1000, 0.1, coeffs="pt1")           # no eigensolver: bit-reproducible inputs (the variant runs with any Psi_T)
    q = O.psit_setup(s, ws)
    gold = {"w_abs_gen_begin": 50, "seed": [1346, 5634, 6635, 4361], "w_target": 3000, "modes": {}}
    for mode in (0, 1):
        wk = O.initial_walkers_psit(ws, q, gold["w_abs_gen_begin"])
        ow = O.OracleWalk(s, ws, wk, 400000, gold["seed"], rng_mode=mode, psit=q)
        pc = O.PopControl(ws.tau, ws.e_trial0, gold["w_target"])
        w_abs = float(np.abs(wk["wt"]).sum())
        steps = []
        for _ in range(8):
            r = pc.pre_step(w_abs)
            if r != 1.0: ow.scale_projector(r)
            st, out = ow.step(pc.params())
            assert st == 0
            steps.append([float(x).hex() for x in out])
            r = pc.post_step(out)
            if r != 1.0: ow.scale_projector(r)
            w_abs = out[1]
        w = ow.walkers()
        gold["modes"][str(mode)] = {"steps": steps, "rng_after": ow.rng_state(), "n_outside_ct": ow.n_outside_ct(),
                                    "det_checksum": int(np.bitwise_xor.reduce(w["up"] * np.uint64(0x9E3779B97F4A7C15) + w["dn"])),
                                    "wt_checksum": float(np.sum(w["wt"] * np.arange(1, len(w["wt"]) + 1) % 7.0)).hex()}
        ow.close()
    json.dump(gold, open(os.path.join(HERE, "psit_c2_8steps.json"), "w"))


if __name__ == "__main__":
    psit_fixture()
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsqmc_ref.so")):
        rannyu_fixture()
    walk_fixture()
