timeout -k 10 300 python bench.py --steps 40000 --warmup 50 --no-cpu-baseline > gpurun_out/b_e_bucket.log 2>&1
SQMC_BUCKET=0 timeout -k 10 300 python bench.py --steps 40000 --warmup 50 --no-cpu-baseline > gpurun_out/b_e_radix.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_e_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", d["config"]["short_list_tail"], "E_proj", d["config"]["projected_energy_Ha"])
PY
