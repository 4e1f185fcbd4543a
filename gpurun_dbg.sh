for v in "SQMC_BUCKET=0" "SQMC_NO_ST3=1" "SQMC_NO_EARLY_HII=1" "SQMC_BUCKET_NO_OFFSETS=1" "X=1"; do
  echo "== $v"; env $v timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_walk_counter_trajectory_bit_exact or run_loop_equals" 2>&1 | tail -3
done
