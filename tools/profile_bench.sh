#!/bin/bash
# Collects the rocprofv3 evidence for the default bench line on the GPU box: kernel trace + stats,
# then the PMC counters in SEPARATE passes (never combined with a trace), and condenses them.
# usage (on the box, from the repo root): bash tools/profile_bench.sh <name> [bench args...]
set -e
name=$1; shift
out=gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $out.trace.log 2>&1
echo "trace done" > gpurun_out/progress_$name.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline "$@" > $out.fetch.log 2>&1
echo "fetch done" >> gpurun_out/progress_$name.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline "$@" > $out.write.log 2>&1
echo "write done" >> gpurun_out/progress_$name.txt
timeout -k 10 300 rocprofv3 --pmc OccupancyPercent MeanOccupancyPerCU --output-format csv -d $out/occ -- python3 bench.py --steps 30 --warmup 5 --equil 300 --no-cpu-baseline "$@" > $out.occ.log 2>&1
echo "occ done" >> gpurun_out/progress_$name.txt
# FETCH_SIZE calibration on known byte counts (8-byte streams, 32-byte gathers), same session
mkdir -p $out && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o $out/calib_fetch
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calib -- $out/calib_fetch > $out.calib.log 2>&1
echo "calib done" >> gpurun_out/progress_$name.txt
python3 tools/summarize_prof.py $out gpurun_out/${name}_rocprof_summary.txt > /dev/null
cp $out/trace/*/*kernel_stats.csv gpurun_out/${name}_kernel_stats.csv
# usage note: pass the commit as SQMC_COMMIT=<hash> in the environment; copy gpurun_out/${name}_* into profiles/ afterwards
