for i in 1 2 3 4 5 6; do timeout -k 10 200 python bench.py --steps 1000 --warmup 5 --equil 2000 --no-cpu-baseline > gpurun_out/b_eq_$i.log 2>&1; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_eq_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"].get("slowest_steps_us"))
PY
