#!/bin/bash
python -m pytest tests/test_rx_gpu.py tests/test_rx_carrier.py tests/test_nonfinite_gpu.py tests/test_e2e_p25.py tests/test_e2e_voice.py tests/test_real_capture.py -x -q -m gpu 2>&1 | tail -3
for d in 0 4096; do
  for sp in "" 1; do
    echo "== DBG=$d spread=$sp"
    DDN_RX_DBG=$d DDN_BENCH_SPREAD=$sp python tools/bench_rx.py 4096 48000 8 16 2>/dev/null | grep "^{" | grep '"matched_filter": 1' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['cpw'], j['matched_filter'], round(j['ms'], 3), j['symbols'], j['syncs'])"
  done
done
