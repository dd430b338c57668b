#!/bin/bash
python -m pytest tests/test_rx4_gpu.py tests/test_voice_gather_gpu.py -x -q -m gpu 2>&1 | tail -2
for b in 0 1 2; do DDN_FUZZ_BASE=$b python -m pytest tests/test_fuzz_rx4_gpu.py -x -q 2>&1 | tail -1; done
for d in 0 1024; do echo "== DDN_RX4_DBG=$d"; DDN_RX4_DBG=$d python tools/bench_rx4.py 4096 48000 2>/dev/null | tail -4; DDN_RX4_DBG=$d python tools/bench_rx4.py 1365 48000 2>/dev/null | tail -4; done
