timeout -k 10 1000 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -12 gpurun_out/t_sel.log
grep -q "rc=0" gpurun_out/t_sel.log && \
SQMC_BENCH_FORCE_SHARDED=1 timeout -k 10 200 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/b_sh1.log 2>&1 && \
true
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_sh1.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["n_gpus"], d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"]["short_list_tail"], d["config"].get("slowest_steps_us"), {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()})
PY
