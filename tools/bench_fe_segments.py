#!/usr/bin/env python3
"""Front-end launch time: one batch of B channels against the same channels as three segments (ddn_batch_set_segments: the shared
front end of the mixed chain), HIP events around the launches.  usage: bench_fe_segments.py [B] [n]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import ddn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
l = ddn.lib()
dev = torch.device("cuda")
iq = torch.randint(0, 256, (B, n, 2), dtype=torch.uint8, device=dev)
out = torch.empty((B, n), dtype=torch.float32, device=dev)


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


one = ddn.Batch(B, block_len=8192)
print("one batch of %d: %.3f ms" % (B, timed(lambda: one.run_device(iq.data_ptr(), n, out.data_ptr(), None))))
third = B // 3
counts = [B - 2 * third, third, third]
for cnts, profs in ((counts, [ddn.LPF_P25_C4FM, ddn.LPF_12K5, ddn.LPF_6K25]), (counts, [ddn.LPF_P25_C4FM] * 3), ([B], [ddn.LPF_P25_C4FM])):
    seg = ddn.Batch(B, block_len=8192)
    cnt, prof = np.array(cnts, np.int32), np.array(profs, np.int32)
    assert l.ddn_batch_set_segments(seg.h, len(cnts), cnt.ctypes.data, prof.ctypes.data) == 0
    first = np.concatenate([[0], np.cumsum(cnts)[:-1]])
    ins = (C.c_void_p * len(cnts))(*[iq.data_ptr() + int(f) * n * 2 for f in first])
    outs = (C.c_void_p * len(cnts))(*[out.data_ptr() + int(f) * n * 4 for f in first])
    print("segments %s profiles %s: %.3f ms" % (cnts, profs, timed(lambda: l.ddn_front_end_run_segments(seg.h, ins, n, outs, None))))
