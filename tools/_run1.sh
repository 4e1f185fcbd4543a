timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hubbard or trajectory or bucket or annihilate or collision or chained" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
grep -q "rc=0" gpurun_out/t_sel.log && \
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 1000 --warmup 20 --system hubbard --target 1e5 --no-cpu-baseline > gpurun_out/b_hub.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_1e5_drv.log"))+sorted(glob.glob("gpurun_out/b_hub.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"]["short_list_tail"], d["config"].get("slowest_steps_us"), {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()}, round(d["config"]["projected_energy_Ha"],5))
PY
SQMC_PROF_SYSTEM=hubbard timeout -k 10 400 python tools/bucket_prof.py > gpurun_out/bprof_hub.log 2>&1; grep "fold\|look-back\|kernel span" gpurun_out/bprof_hub.log
