timeout -k 10 1000 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -25 gpurun_out/t_sel.log
