"""Per-section cycle profile of k_p25_rxw (tools/scratch/prof/libdsdneo_hip_prof.so: the kernel with readcyclecounter marks).
usage: python tools/scratch/prof/run_prof.py [frame_dibits]"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ddn
ddn.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdsdneo_hip_prof.so")
import orc

FR = int(sys.argv[1]) if len(sys.argv) > 1 else 864
B, n = 4096, 48000
base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=FR)
x = np.tile(base, (B // 64, 1))
rx = ddn.P25Rx(B, lock_symbols=FR - 24, use_matched_filter=1, channels_per_wave=16)
rec, fl, cnt = rx.run(x)
print("ok", int(cnt.sum()))
