#!/usr/bin/env python3
"""Front end alone, 4096 x 48000 cu8 (clean C4FM and noise): ms per launch, HIP events around ten launches.
usage: bench_fe.py   (DDN_LIB_PATH selects a variant build, tools/build_variant.sh)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ddn, orc
B, n = 4096, 48000
out = torch.empty((B, n), dtype=torch.float32, device="cuda")
res = []
for kind in ("clean", "noise"):
    if kind == "clean":
        base = orc.synth_c4fm_cu8(0, 16, n)
        iq = torch.from_numpy(np.tile(base, (B // 16, 1, 1))).cuda()
    else:
        iq = torch.randint(0, 256, (B, n, 2), dtype=torch.uint8, device="cuda")
    b = ddn.Batch(B, block_len=8192)
    for _ in range(3):
        b.run_device(iq.data_ptr(), n, out.data_ptr(), None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.run_device(iq.data_ptr(), n, out.data_ptr(), None)
    e1.record()
    torch.cuda.synchronize()
    res.append("%s %.3f ms" % (kind, e0.elapsed_time(e1) / 10))
print("front end 4096 x 48000:", ", ".join(res), "(%s)" % os.path.basename(ddn.LIB_PATH))
