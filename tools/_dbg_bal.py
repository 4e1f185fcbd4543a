import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SQMC_EXTRA_CFLAGS"] = "-DBUCKET_PROF"
import torch
import sqmc_amd
sqmc_amd.build_library(force=True)
from sqmc_amd import host as H
hst = H.ChemHost(os.path.join(ROOT, "tests", "golden", "C2_r1.24253_FCIDUMP"), 8, 4, "d2h")
w = H.GpuWalk(hst, 1e5, seed=(1346, 5634, 6635, 4361))
w.run(400, keep_stats=False)
L = sqmc_amd.load_library()
buf = (C.c_uint64 * (16 * 1024))()
prev = None
for it in range(10):
    w.run(2 if it else 3, keep_stats=False)
    L.sqmc_gpu_debug_bprof(buf)
    a = np.array(buf, dtype=np.int64).reshape(1024, 16)[:256]
    R, S = a[:, 13].astype(float), a[:, 12].astype(float)
    cost = R + 1.7 * S
    top = np.argsort(-cost)[:5]
    kb3 = (C.c_uint32 * (3 * 1025))(); pos = (C.c_uint32 * 1025)(); sc = (C.c_uint32 * 1025)(); stt = (C.c_int * 8)()
    L.sqmc_gpu_debug_buckets(w.g.h, kb3, pos, sc, stt)
    K = np.array(kb3, dtype=np.int64).reshape(3, 1025); P = np.array(pos, dtype=np.int64); SC = np.array(sc, dtype=np.int64)
    print("   state next/scbuf/use/scB/kbB", list(stt)[:7], "n0", SC[256])
    for bb in (193, 194, 195):
        print("   b %d: K0 %d K1 %d K2 %d | pos %d..%d (R %d) scount %d | prof R %d S %d" % (bb, K[0, bb], K[1, bb], K[2, bb], P[bb], P[bb + 1], P[bb + 1] - P[bb], SC[bb], R[bb], S[bb]))
    print("it %d: S max %d, cost mean %.0f max %.0f std %.0f; top: %s" % (it, S.max(), cost.mean(), cost.max(), cost.std(), ", ".join("%d:%d+%d" % (i, R[i], S[i]) for i in top)), flush=True)
w.close()
os.environ.pop("SQMC_EXTRA_CFLAGS")
sqmc_amd.build_library(force=True)
