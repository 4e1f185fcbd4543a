timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log
tail -4 gpurun_out/t_all.log
