timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fortran" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -4 gpurun_out/t_sel.log
