export SQMC_COMMIT=$1
bash tools/profile_bench.sh r02_bench_1e5 && bash tools/profile_bench.sh r02_bench_1e6 --target 1e6
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv.log 2>&1
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --target 1e6 --no-cpu-baseline > gpurun_out/b_1e6.log 2>&1
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --target 1e7 --no-cpu-baseline > gpurun_out/b_1e7.log 2>&1
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --system heg --target 1e6 --no-cpu-baseline > gpurun_out/b_heg.log 2>&1
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --system hubbard --target 1e5 --no-cpu-baseline > gpurun_out/b_hub.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"]["short_list_tail"], {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()})
PY
