#!/usr/bin/env python3
"""Where the time of k_anneal goes for long lists: a -DANNEAL_PROF build (a file of its own, sqmc_amd/_lib.py) stamps every tile at
its phase boundaries (100 MHz wall clock); the bench population at the given target runs a few hundred steps and the tiles of the last
step are summarised.  usage: python tools/anneal_prof.py [target]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SQMC_EXTRA_CFLAGS"] = "-DANNEAL_PROF"
import torch  # noqa: F401
import sqmc_amd
sqmc_amd.build_library()
from sqmc_amd import host as H
target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
hst = H.ChemHost(os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP"), 8, 4, "d2h")
w = H.GpuWalk(hst, target, seed=(1346, 5634, 6635, 4361), w_begin=min(target, 1e4))
w.run(1200, keep_stats=False)
L = sqmc_amd.load_library()
buf = (C.c_uint64 * (8 * 16384))()
assert L.sqmc_gpu_debug_aprof(buf) == 0
a = np.array(buf, dtype=np.int64).reshape(16384, 8)
a = a[a[:, 0] > 0]
a = a[a[:, 5] > a[:, 0]]
t0 = a[:, 0].min()
print("tiles", len(a), "kernel span %.1f us" % ((a[:, 5].max() - t0) / 100.0), "tile start: mean %.1f max %.1f us" % ((a[:, 0] - t0).mean() / 100.0, (a[:, 0] - t0).max() / 100.0))
names = ["load + stage + fold", "pre-merge sums", "REPLAY rank (COUNTER: nothing)", "rounding + scans + look-back", "compaction + estimator + gate"]
for k in range(5):
    d = (a[:, k + 1] - a[:, k]) / 100.0
    print("%-34s mean %7.2f  p90 %7.2f  max %7.2f us" % (names[k], d.mean(), np.percentile(d, 90), d.max()))
life = (a[:, 5] - a[:, 0]) / 100.0
print("tile life: mean %.1f p90 %.1f max %.1f us" % (life.mean(), np.percentile(life, 90), life.max()))
w.close()
