#!/usr/bin/env python3
"""Torch-free driver for PMC passes over the secondary kernels (one launch of each at the stage-bench shapes, through the
host-buffer entry points): receive loop + matched filter, Gardner, stand-alone slicer, resampler, CQPSK chain."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DDN_NO_TORCH", "1")
import ddn  # noqa: E402
import orc  # noqa: E402

l = ddn.lib()
B, n = 4096, 48000
rng = np.random.default_rng(1)
base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=864)
x = np.tile(base, (B // 64, 1))
ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1).run(x)

h = C.c_void_p()
assert l.ddn_resampler_create(B, 5, 4, C.byref(h)) == 0
k = l.ddn_resampler_out_len(h, n)
out = np.zeros((B, k), np.float32)
assert l.ddn_resampler_run_host(h, x.ctypes.data, n, out.ctypes.data, k) == 0
l.ddn_resampler_destroy(h)

sym = np.tile(np.stack([orc.synth_c4fm_symbols(100 + c, 4800) for c in range(8)]), (B // 8, 1)).astype(np.float32)
hs = C.c_void_p()
assert l.ddn_slicer_batch_create(B, 0, C.byref(hs)) == 0
rec = np.zeros((B, 4800, 10), np.uint8)
assert l.ddn_p25_slicer_run_host(hs, sym.ctypes.data, 4800, rec.ctypes.data) == 0
l.ddn_slicer_batch_destroy(hs)

iq = np.tile(orc.synth_qpsk_f32(77, 8, 4893, 10, noise=0.05), (B // 8, 1, 1))
ht = C.c_void_p()
assert l.ddn_ted_batch_create(B, 10, 4800, 0.0, C.byref(ht)) == 0
nn = iq.shape[1]
so = np.zeros((B, nn // 2 + 8, 2), np.float32)
cnt = np.zeros(B, np.int32)
assert l.ddn_gardner_run_host(ht, iq.ctypes.data, nn, so.ctypes.data, nn // 2 + 8, cnt.ctypes.data) == 0
l.ddn_ted_batch_destroy(ht)

q = np.tile(orc.synth_dqpsk_f32(12, 8, 4808, 5), (B // 8, 1, 1))
ddn.CqpskBatch(B, rate=24000, block_len=4096).run(q)
print("ok")
