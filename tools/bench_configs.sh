#!/bin/bash
# the other BASELINE configurations with the current build, one JSON line each (auxiliary measurement)
out=gpurun_out/r03_bench_configs.jsonl
: > $out
for args in "--system hubbard --target 1e4" "--system hubbard --target 1e5" "--system hubbard --target 1e6" "--system hubbard --target 1e7" "--system heg --target 1e6" "--target 1e6" "--target 1e7"; do
  echo "== $args" >> gpurun_out/r03_bench_configs.log
  timeout -k 10 280 python bench.py $args --steps 400 --warmup 20 --no-cpu-baseline >> $out 2>> gpurun_out/r03_bench_configs.log || echo "{\"failed\": \"$args\"}" >> $out
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_bench_configs.jsonl"):
    d = json.loads(l)
    if "failed" in d: print(d); continue
    print("%-90s %8.4f ms  %.3e  E=%s" % (d["config"]["workload"][:90], d["ms_per_step"], d["value"], d["config"]["projected_energy_Ha"]))
PY
