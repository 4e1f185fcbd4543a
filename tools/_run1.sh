timeout -k 10 500 python bench.py --steps 20 --warmup 5 --equil 400 --system heg --target 1e7 --no-cpu-baseline > gpurun_out/b_heg7.log 2>&1
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --equil 600 --system heg --target 1e5 --no-cpu-baseline > gpurun_out/b_heg5.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b_heg[57].log")):
    ok=False
    for l in open(f):
        if l.startswith("{"):
            ok=True; d=json.loads(l); print(f, d["steps"], round(d["ms_per_step"]*1000,1), "us", round(d["value"]/1e9,3), "e9 w-steps/s", d["config"]["occupied_dets_per_step"], {k:round(v*1000,1) for k,v in d["roofline"]["stage_ms_per_step"].items()})
    if not ok: print(f, open(f).read()[-500:])
PY
