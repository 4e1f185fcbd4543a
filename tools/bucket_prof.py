#!/usr/bin/env python3
"""Where the time of k_anneal_bucket goes: rebuilds the library with -DBUCKET_PROF on the GPU box, runs the bench
configuration for a few hundred steps and prints, per phase, the mean / max over the buckets of the last step (wall clock,
100 MHz ticks -> us).  The instrumented build is a file of its own (sqmc_amd/_lib.py: the flags are part of its name); the product's library is not touched."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SQMC_EXTRA_CFLAGS"] = "-DBUCKET_PROF"
import torch  # noqa: F401
import sqmc_amd
sqmc_amd.build_library()
from sqmc_amd import host as H

target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
if os.environ.get("SQMC_PROF_SYSTEM") == "hubbard":
    hst = H.HubbardHost(4, 4, True, 8, 8, 1.0, 4.0)
    w = H.GpuWalk(hst, target, seed=(1346, 5634, 6635, 4361), w_begin=min(target, 1e4), n_truncate_trial_wf=20, size_deterministic=500, tau_multiplier=0.5)
else:
    hst = H.ChemHost(os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP"), 8, 4, "d2h")
    w = H.GpuWalk(hst, target, seed=(1346, 5634, 6635, 4361))
w.run(500, keep_stats=False)
L = sqmc_amd.load_library()
buf = (C.c_uint64 * (16 * 1024))()
assert L.sqmc_gpu_debug_bprof(buf) == 0
old = np.array(buf, dtype=np.int64).reshape(1024, 16)
w.run(2, keep_stats=False)
assert L.sqmc_gpu_debug_bprof(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(1024, 16)
a = a[a[:, 0] != old[:, 0]]                    # the buckets of the last step only
print("tail stats", w.g.tail_stats())
nb = len(a)
t0 = a[:, 0].min()
names = ["ticket+rows issue", "rows scan", "gather words", "sort", "records", "merge order", "fold+round", "chunk scan", "lookback", "compaction", "sums"]
print("buckets", nb, "kernel span %.1f us" % ((a[:, 10].max() - t0) / 100.0))
print("start skew: mean %.1f max %.1f us" % ((a[:, 0] - t0).mean() / 100.0, (a[:, 0] - t0).max() / 100.0))
seq = [0, 1, 2, 3, 4, 5, 6, 9, 8, 11, 10]        # stamp 7 sits right behind 9
names = ["rows (loads + scan)", "gather words", "LDS sort", "records", "merge order", "fold + round + gate count", "chunk scan + look-back", "compaction", "H_ii of new determinants", "block sums"]
for k in range(1, len(seq)):
    d = (a[:, seq[k]] - a[:, seq[k - 1]]) / 100.0
    print("%-28s mean %6.2f  max %6.2f us" % (names[k - 1], d.mean(), d.max()))
srt = (a[:, 3] - a[:, 2]) / 100.0
order = np.argsort(-srt)[:6]
print("slowest sorts: " + ", ".join("%.1f us (S %d R %d, started %.1f)" % (srt[i], a[i, 12], a[i, 13], (a[i, 0] - t0) / 100.0) for i in order))
life = (a[:, 10] - a[:, 0]) / 100.0
order = np.argsort(-(a[:, 6] - a[:, 0]))[:6]
print("latest to publish: " + ", ".join("%.1f us after its start (S %d R %d)" % ((a[i, 6] - a[i, 0]) / 100.0, a[i, 12], a[i, 13]) for i in order))
print("H_ii phase: wait for the block's last compaction thread mean %.2f max %.2f us; the passes mean %.2f max %.2f us; determinants per bucket mean %.0f max %d" % (
    ((a[:, 14] - a[:, 8]) / 100.0).mean(), ((a[:, 14] - a[:, 8]) / 100.0).max(), ((a[:, 11] - a[:, 14]) / 100.0).mean(), ((a[:, 11] - a[:, 14]) / 100.0).max(), a[:, 15].mean(), a[:, 15].max()))
print("S: mean %.0f max %d; R mean %.0f min %d max %d; T max %d" % (a[:, 12].mean(), a[:, 12].max(), a[:, 13].mean(), a[:, 13].min(), a[:, 13].max(), (a[:, 12] + a[:, 13]).max()))
ids = np.nonzero(np.array(buf, dtype=np.int64).reshape(1024, 16)[:, 0] != old[:, 0])[0]
top = np.argsort(-(a[:, 12] + a[:, 13]))[:8]
print("fullest buckets (id: R + S): " + ", ".join("%d: %d + %d" % (ids[i], a[i, 13], a[i, 12]) for i in top))
np.save(os.path.join(ROOT, "gpurun_out", "bprof_rs.npy"), np.stack([ids, a[:, 13], a[:, 12], (a[:, 6] - a[:, 0]), (a[:, 3] - a[:, 2]), (a[:, 6] - a[:, 5])], axis=1))
print("first buckets (id: R + S): " + ", ".join("%d: %d + %d" % (ids[i], a[i, 13], a[i, 12]) for i in range(6)))
d = (a[:, 10] - a[:, 0]) / 100.0
print("bucket life: mean %.1f max %.1f us; end skew mean %.1f" % (d.mean(), d.max(), (a[:, 10].max() - a[:, 10]).mean() / 100.0))
w.close()
