#!/usr/bin/env python3
"""Front-end phase table: three calls of one 4096 x 48000 batch on clean C4FM or on noise, for the per-wave cycle print of the
experiments library (DDN_LIB_PATH=dsd-neo_amd/libdsdneo_hip_exp.so DDN_DBG=64 DDN_DBG_PRINT=1).  usage: fe_phase_times.py clean|noise"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ddn, orc
B, n = 4096, 48000
kind = sys.argv[1]
if kind == "clean":
    base = orc.synth_c4fm_cu8(0, 16, n)
    iq = torch.from_numpy(np.tile(base, (B // 16, 1, 1))).cuda()
else:
    iq = torch.randint(0, 256, (B, n, 2), dtype=torch.uint8, device="cuda")
out = torch.empty((B, n), dtype=torch.float32, device="cuda")
b = ddn.Batch(B, block_len=8192)
for _ in range(3):
    b.run_device(iq.data_ptr(), n, out.data_ptr(), None)
torch.cuda.synchronize()
