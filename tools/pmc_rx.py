#!/usr/bin/env python3
"""Torch-free driver for PMC collection on k_p25_rx (4096 channels x 48000 discriminator samples, host-buffer path)."""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DDN_NO_TORCH", "1")
import ddn  # noqa: E402
import orc  # noqa: E402

B, n = 4096, 48000
FR = int(sys.argv[2]) if len(sys.argv) > 2 else 864
base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=FR)
x = np.tile(base, (B // 64, 1))
rx = ddn.P25Rx(B, lock_symbols=FR - 24, use_matched_filter=1, channels_per_wave=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
for _ in range(2):
    rec, fl, cnt = rx.run(x)
print("ok", int(cnt.sum()))
