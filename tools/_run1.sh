timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trajectory_bit_exact_at_bench or bucket or chained" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
grep -q "rc=0" gpurun_out/t_sel.log && \
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv.log 2>&1 && \
timeout -k 10 120 python bench.py --steps 2000 --warmup 50 --no-cpu-baseline > gpurun_out/b_1e5_c.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_1e5_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"].get("slowest_steps_us"), {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()})
PY
timeout -k 10 400 python tools/bucket_prof.py > gpurun_out/bprof.log 2>&1; grep "H_ii\|kernel span\|compaction\|look-back" gpurun_out/bprof.log
