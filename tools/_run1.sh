export SQMC_COMMIT=$1
bash tools/profile_bench.sh r02_bench_1e5 && bash tools/profile_bench.sh r02_bench_1e6 --target 1e6
ls gpurun_out/r02_bench_1e*
