timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "walk_deck or survey_walk" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -6 gpurun_out/t_sel.log
