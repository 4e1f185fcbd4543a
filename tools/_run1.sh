timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trajectory or heg57 or past_2_20 or hubbard or run_loop or annihilate" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
grep -q "rc=0" gpurun_out/t_sel.log && \
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --equil 400 --target 1e6 --no-cpu-baseline > gpurun_out/b_1e6.log 2>&1 && \
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --equil 400 --target 1e7 --no-cpu-baseline > gpurun_out/b_1e7.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --equil 400 --system heg --target 1e6 --no-cpu-baseline > gpurun_out/b_heg.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 30 --warmup 10 --equil 400 --system hubbard --target 1e7 --no-cpu-baseline > gpurun_out/b_hub7.log 2>&1 && \
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_1e5_drv.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_1e*.log"))+sorted(glob.glob("gpurun_out/b_h*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()})
PY
