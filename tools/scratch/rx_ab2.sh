#!/bin/bash
for d in 1024 2048 0; do
  echo "== DDN_RX_DBG=$d"
  DDN_RX_DBG=$d python -m pytest tests/test_rx_gpu.py -q -m gpu 2>&1 | tail -8
done
