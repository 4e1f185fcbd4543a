for v in "X=1" "SQMC_HII_IN_SPAWN=1"; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,1), 'us/step', {k:round(v*1000,1) for k,v in d['roofline']['stage_ms_per_step'].items()}, d['config']['short_list_tail'])"
done
echo "== driver flags"; timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,1), 'us/step', d['roofline']['frac'], d['roofline']['whole_step'])"
timeout -k 10 800 python -m pytest tests -m gpu -q --maxfail=5 -k "not sharded" 2>&1 | tail -4
