#!/usr/bin/env python3
"""Minimal driver for PMC collection: runs k_front_end_fused on the bench shape (4096 x 48000 cu8) a few times through
the host-buffer entry point, with no PyTorch in the process (rocprofv3 --pmc serialises every kernel, so the thousands
of tiny torch kernels bench.py uses to synthesise its input must not be in the traced process).

    DDN_NO_TORCH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p -o p -- python tools/pmc_front_end.py
"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
os.environ.setdefault("DDN_NO_TORCH", "1")
import ddn  # noqa: E402

B, n = 4096, 48000
rng = np.random.default_rng(0)
ph = np.cumsum(rng.choice([-0.084, -0.028, 0.028, 0.084], size=(64, n)), axis=1)
one = np.stack([127.5 + 108.0 * np.cos(ph), 127.5 + 108.0 * np.sin(ph)], axis=2).astype(np.uint8)
iq = np.tile(one, (B // 64, 1, 1))
b = ddn.Batch(B, block_len=8192)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = b.run_host(iq, n)
print("ok", float(np.abs(out).max()))
