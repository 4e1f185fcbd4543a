#!/usr/bin/env python3
"""Where the time of k_spawn goes at the bench size: rebuilds the library with -DSPAWN_PROF on the GPU box, runs the bench
configuration and prints, per phase, the mean / max over the spawning blocks of the last step (wall clock, 100 MHz ticks).
The instrumented build is a file of its own (sqmc_amd/_lib.py: the flags are part of its name); the product's library is not touched."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SQMC_EXTRA_CFLAGS"] = "-DSPAWN_PROF"
import torch  # noqa: F401
import sqmc_amd
sqmc_amd.build_library()
from sqmc_amd import host as H

target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e5
hst = H.ChemHost(os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP"), 8, 4, "d2h")
w = H.GpuWalk(hst, target, seed=(1346, 5634, 6635, 4361), w_begin=min(target, 1e4))
w.run(500 if target <= 2e5 else 1200, keep_stats=False)
L = sqmc_amd.load_library()
buf = (C.c_uint64 * (8 * 8192))()
assert L.sqmc_gpu_debug_prof(buf) == 0
old = np.array(buf, dtype=np.int64).reshape(8192, 8)
w.run(2, keep_stats=False)
assert L.sqmc_gpu_debug_prof(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(8192, 8)
live = (a[:, 5] != old[:, 5]) & (a[:, 5] > a[:, 0])           # blocks that spawned in the last step (stamp 5 = end)
a = a[live]
t0 = a[:, 0].min()
print("tail stats", w.g.tail_stats(), "spawning blocks", len(a), "kernel span %.1f us" % ((a[:, 5].max() - t0) / 100.0))
print("start skew: mean %.1f max %.1f us" % ((a[:, 0] - t0).mean() / 100.0, (a[:, 0] - t0).max() / 100.0))
names = ["table staging + first probe + splitters", "parent window (256-way probes + LDS window)", "parent search + loads", "proposal", "weight + emit", "partition (bucket search, ranks, offsets, words)"]
for k in range(1, 6):
    d = (a[:, k] - a[:, k - 1]) / 100.0
    print("%-52s mean %6.2f  max %6.2f us" % (names[k - 1] if k < 5 else names[5], d.mean(), d.max()))
d = (a[:, 5] - a[:, 0]) / 100.0
print("block life: mean %.1f max %.1f us" % (d.mean(), d.max()))
w.close()
