timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained or run_loop or fortran_host_walk or trajectory_bit_exact_at_bench" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -8 gpurun_out/t_sel.log
python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch, sqmc_amd
from sqmc_amd import host as H
hst = H.ChemHost("tests/golden/C2_r1.24253_FCIDUMP", 8, 4, "d2h")
for chain in (False, True):
    w = H.GpuWalk(hst, 1e5, seed=(1346, 5634, 6635, 4361))
    w.run(2000, keep_stats=False)
    w.g.set_chained_runs(chain)
    for _ in range(50): w.step()
    t0 = time.perf_counter()
    for _ in range(500): w.step()
    dt = time.perf_counter() - t0
    w.g.set_chained_runs(False)
    print("python-driven step(), chained =", chain, ": %.1f us per step" % (dt / 500 * 1e6)); w.close()
PY
