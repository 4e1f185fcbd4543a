timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "survey_walk_deck" --durations=3 > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sel.log
tail -12 gpurun_out/t_sel.log
