#!/usr/bin/env python3
"""Auxiliary measurement (not the driver's bench line): HCI variational stage of
BASELINE.json configs[4] on one GPU + bandwidth of the Davidson matvec on the final matrix."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch            # before the HIP library: both must share one libamdhip64 (torch's copy is loaded first)
import sqmc_amd
from sqmc_amd import host as H

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--eps-var", type=float, default=1e-4)
ap.add_argument("--eps-pt", type=float, default=1e-6)
ap.add_argument("--pt-slices", type=int, default=1, help="slices of the connected space in the PT stage")
args = ap.parse_args()
FCIDUMP = os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP")
t0 = time.perf_counter()
h = H.ChemHost(FCIDUMP, 8, 4, "d2h", time_sym=True, z=1, hf_symmetry=1)
g = h.gpu()
g.set_hb_tables(*h.hb_tables(g))
t1 = time.perf_counter()
up, dn, w, e, hist = H.hci_variational(h, g, args.eps_var, eps_sched=(2 * args.eps_var, 2 * args.eps_var))
t2 = time.perf_counter()
de_pt, n_conn = H.hci_pt2_determinant_basis(h, up, dn, w[:, 0], float(e[0]), args.eps_pt, n_slices=args.pt_slices)
t2b = time.perf_counter()
order = H.sort_dets(up, dn)
ta = time.perf_counter(); counts, idx, val = g.build_sparse_ham(up[order], dn[order]); tb = time.perf_counter()
plan = sqmc_amd.SpmvPlan(counts, idx, val)
n, nnz = len(counts), len(val)
nnz_full = 2 * nnz - n
import ctypes as C
L = sqmc_amd.load_library()
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
for _ in range(5):
    L.sqmc_gpu_spmv_apply(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1)
torch.cuda.synchronize(); t3 = time.perf_counter()
reps = 50
for _ in range(reps):
    L.sqmc_gpu_spmv_apply(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1)
torch.cuda.synchronize(); t4 = time.perf_counter()
ms = (t4 - t3) / reps * 1e3
os.environ["SQMC_SPMV_PROBE_NO_GATHER"] = "1"          # what the product would cost if the gather of x were free (wrong sums: timing only)
yp = torch.empty_like(x)
for _ in range(3):
    L.sqmc_gpu_spmv_apply(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(yp.data_ptr()), 1)
tp = time.perf_counter()
for _ in range(reps):
    L.sqmc_gpu_spmv_apply(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(yp.data_ptr()), 1)
ms_probe = (time.perf_counter() - tp) / reps * 1e3
os.environ.pop("SQMC_SPMV_PROBE_NO_GATHER")
# the other layout (stored triangle + fp64 atomics for the transposed half), measured on the same matrix
os.environ["SQMC_SPMV_UPPER_ATOMIC"] = "1"
plan2 = sqmc_amd.SpmvPlan(counts, idx, val)
y2 = torch.empty_like(x)
for _ in range(3):
    L.sqmc_gpu_spmv_apply(plan2.h, C.c_void_p(x.data_ptr()), C.c_void_p(y2.data_ptr()), 1)
torch.cuda.synchronize(); t5 = time.perf_counter()
for _ in range(20):
    L.sqmc_gpu_spmv_apply(plan2.h, C.c_void_p(x.data_ptr()), C.c_void_p(y2.data_ptr()), 1)
torch.cuda.synchronize(); ms_atomic = (time.perf_counter() - t5) / 20 * 1e3
os.environ.pop("SQMC_SPMV_UPPER_ATOMIC")
err_atomic = float((y2 - y).abs().max() / y.abs().max())
plan2.close()
alg = 20.0 * nnz + 20.0 * n            # 20 B per stored nonzero + 20 B per row
print(json.dumps({"hci_variational_s": t2 - t1, "setup_s": t1 - t0, "ndets_history": hist, "e_var": float(e[0]), "eps_var": args.eps_var, "pt2_eps": args.eps_pt, "pt2_slices": args.pt_slices, "pt2_delta_e": de_pt, "pt2_connected_dets": n_conn, "pt2_s": t2b - t2, "e_total": float(e[0]) + de_pt,
                  "build_sparse_ham_s": tb - ta, "n": n, "nnz_upper": nnz, "spmv_ms": ms, "spmv_ms_if_the_gather_were_free": ms_probe, "spmv_upper_atomic_ms": ms_atomic, "spmv_upper_atomic_rel_dev": err_atomic,
                  "spmv_algorithmic_GBs": alg / (ms * 1e-3) / 1e9, "spmv_frac_of_8TBs": alg / (ms * 1e-3) / 1e9 / 8000.0,
                  "spmv_moved_GBs_full_csr": (12.0 * nnz_full + 8.0 * nnz_full + 20.0 * n) / (ms * 1e-3) / 1e9}))
