#!/usr/bin/env python3
"""configs[3] shape on one GPU through ddn_mixed_chain (P25 Phase 1 + DMR + NXDN48 groups), for rocprofv3 kernel traces:
usage: bench_mixed.py [channels] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import ddn
from conftest import golden

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 48000
third = B // 3
Bp, Bd, Bn = B - 2 * third, third, third
if os.environ.get("MIX"):  # "p25,dmr,nxdn" channel counts (timing experiments: which splits leave every loop workgroup resident)
    Bp, Bd, Bn = (int(v) for v in os.environ["MIX"].split(","))
voice, ctrl = bench.make_base_traffic(n)
iq = np.stack([(voice if k == "voice" else ctrl)[i] for k, i in (bench.channel_source(c) for c in range(Bp))])
dev = torch.device("cuda")


def tile(name, lo, hi, Bc):
    x = torch.from_numpy(np.ascontiguousarray(golden(name)["iq"][lo:hi], np.uint8)).to(dev)
    off = (torch.arange(Bc, device=dev) * 37) % (x.shape[0] - n)
    return x[off[:, None] + torch.arange(n, device=dev)[None, :]].contiguous()


d_p, d_d, d_n = torch.from_numpy(iq).to(dev), tile("iq_dmr_t3_ras_cc.npz", 0, 96000, Bd), tile("iq_nxdn48.npz", 60000, 288000, Bn)
m = ddn.MixedChainC(Bp, Bd, Bn, n, overlap=int(os.environ.get("MIX_OVERLAP", "0")))  # (ddn_mixed_chain_config.overlap)
for _ in range(2):
    m.run(d_p.data_ptr(), d_d.data_ptr(), d_n.data_ptr())
m.wait()
t0 = time.perf_counter()
for _ in range(steps):
    m.run(d_p.data_ptr(), d_d.data_ptr(), d_n.data_ptr())
m.wait()
print("mixed %d channels: %.3f ms per step" % (B, (time.perf_counter() - t0) / steps * 1e3))
