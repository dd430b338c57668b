#!/bin/bash
# A/B of the receive-loop variants: DDN_RX_DBG bit 1024 = no lean trip, 2048 = serial crossing search
python -m pytest tests/test_rx_gpu.py tests/test_rx_carrier.py tests/test_nonfinite_gpu.py tests/test_e2e_p25.py tests/test_e2e_voice.py tests/test_real_capture.py -x -q -m gpu 2>&1 | tail -3
for d in 0 1024 2048 3072; do
  echo "== DDN_RX_DBG=$d"
  DDN_RX_DBG=$d python tools/bench_rx.py 4096 48000 8 16 32 2>&1 | grep '"matched_filter": 1' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], round(j['ms'], 3), j['symbols'], j['syncs'])"
done
