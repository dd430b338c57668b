#!/bin/bash
# DDN_FUZZ_BASE sweep over every seed-aware GPU test: bases $1..$2
fail=0
for b in $(seq $1 $2); do
  DDN_FUZZ_BASE=$b timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_fuzz2_gpu.py tests/test_fuzz_rx4_gpu.py tests/test_isch_gpu.py tests/test_rs28_gpu.py tests/test_rs_gpu.py tests/test_fec_gpu.py tests/test_block_gpu.py -x -q -m gpu 2>&1 | tail -1 | grep -q " passed" || { echo FAIL base $b; fail=1; }
done
echo "sweep $1..$2 fail=$fail"
