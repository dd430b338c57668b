#!/bin/bash
python -m pytest tests/test_rx_gpu.py tests/test_rx_carrier.py tests/test_nonfinite_gpu.py tests/test_e2e_p25.py tests/test_e2e_voice.py tests/test_real_capture.py tests/test_slicer_gpu.py -x -q -m gpu 2>&1 | tail -3
fail=0
for b in 0 1 2 3 4 5 6 7; do DDN_FUZZ_BASE=$b python -m pytest tests/test_fuzz_gpu.py tests/test_fuzz2_gpu.py -x -q 2>&1 | tail -1 | grep -q " passed" || { echo FAIL fuzz base $b; fail=1; }; done
echo fuzz fail=$fail
DDN_RX_DBG=0 python tools/bench_rx.py 4096 48000 8 16 32 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], j['matched_filter'], round(j['ms'], 3), j['symbols'], j['syncs'])"
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_lean.json; python -c "
import json; j=json.load(open('gpurun_out/bench_lean.json')); print(j['value'], j['ms_per_step'], j['roofline'])"
