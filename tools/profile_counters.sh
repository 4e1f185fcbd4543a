#!/bin/bash
# Stall / LDS / L2 counters of the bench's kernels, in separate rocprofv3 --pmc passes (never combined with a trace), filtered against
# what `rocprofv3 -L` lists on the box so that one unknown name does not cost a pass.
# usage (on the box, from the repo root): bash tools/profile_counters.sh <name> [bench args...]   ->  gpurun_out/<name>_counters.txt
name=$1; shift
out=gpurun_out/prof_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 rocprofv3 -L > $out/counters_available.txt 2>&1
pick() { python3 - "$out/counters_available.txt" "$@" <<'PY'
import re, sys
have = set(re.findall(r"[A-Za-z][A-Za-z0-9_]+", open(sys.argv[1]).read()))
print(" ".join(c for c in sys.argv[2:] if c in have))
PY
}
i=0
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_FLAT SQ_INSTS_GDS" \
             "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  cs=$(pick $group)
  [ -z "$cs" ] && continue
  timeout -k 10 300 rocprofv3 --pmc $cs --output-format csv -d $out/pmc$i -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline "$@" > $out.pmc$i.log 2>&1
  echo "pass $i ($cs) done rc=$?" >> gpurun_out/progress_$name.txt
done
python3 tools/summarize_counters.py $out gpurun_out/${name}_counters.txt > /dev/null
