#!/bin/bash
python -m pytest tests/test_rx_gpu.py tests/test_rx_carrier.py tests/test_nonfinite_gpu.py tests/test_e2e_p25.py tests/test_e2e_voice.py tests/test_real_capture.py -x -q -m gpu 2>&1 | tail -3
python tools/scratch/rx_cyc.py 16 2>&1 | grep cpw
for d in 0 2048; do
    echo "== DDN_RX_DBG=$d"
    DDN_RX_DBG=$d python tools/bench_rx.py 4096 48000 8 16 32 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], j['matched_filter'], round(j['ms'], 3), j['symbols'], j['syncs'])"
done
