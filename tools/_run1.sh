timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chained or run_loop or trajectory_bit_exact_at_bench" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -6 gpurun_out/t_sel.log
grep -q "rc=0" gpurun_out/t_sel.log && \
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv.log 2>&1 && \
SQMC_BENCH_NO_CHAIN=1 timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv_nochain.log 2>&1 && \
timeout -k 10 120 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > gpurun_out/b_1e5_c.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/b_full.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"]["short_list_tail"], d["config"].get("slowest_steps_us"), d.get("energy_error_Ha"))
PY
