"""Where the PCIe-inclusive step of bench.py spends its time: ddn_p25_chain_run_host with no result copies, with the small results
only, with everything - steady state over 6 steps each, the bench's own traffic."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ddn  # noqa: E402

B, n = bench.B_PER_GPU, bench.N_SAMPLES
voice, ctrl = bench.make_base_traffic(n)
dev = torch.device("cuda:0")
kinds = [bench.channel_source(c) for c in range(B)]
d_iq = torch.empty((B, n, 2), dtype=torch.uint8, device=dev)
dv, dc = torch.from_numpy(voice).to(dev), torch.from_numpy(ctrl).to(dev)
is_v = torch.tensor([k == "voice" for k, _ in kinds], device=dev)
d_iq[is_v] = dv[torch.tensor([b for k, b in kinds if k == "voice"], device=dev)]
d_iq[~is_v] = dc[torch.tensor([b for k, b in kinds if k == "ctrl"], device=dev)]
torch.cuda.synchronize()
chain = ddn.P25ChainC(B, n, block_len=bench.BLOCK)
l = ddn.lib()
S, V, st, E = B * chain.F, B * chain.Fv * 9, chain.stride, chain.E
sizes = {"records10": B * st * 10, "flags": B * st, "counts": B * 4, "events": B * E * 16, "n_events": B * 4, "event_data": B * E * 16,
         "nid4": S * 16, "tsbk": 3 * S * 12, "pcm": V * 640}
print({k: round(v / 1e6, 1) for k, v in sizes.items()}, "MB; iq", B * n * 2 / 1e6)


def pin(nb):
    p = C.c_void_p()
    assert l.ddn_host_alloc_pinned(nb, C.byref(p)) == 0
    return p


h_iq = [pin(B * n * 2) for _ in range(2)]
for p in h_iq:
    assert l.ddn_device_download(p, d_iq.data_ptr(), B * n * 2) == 0
bufs = [{k: pin(v) for k, v in sizes.items()} for _ in range(3)]


def run(fields, steps=12):
    outs = []
    for b in bufs:
        o = ddn.P25ChainHostOut()
        for k in fields:
            setattr(o, k, b[k].value)
        outs.append(o)
    for k in range(3):
        chain.run_host(h_iq[k & 1], outs[k % 3] if fields else None)
    chain.wait()
    t0 = time.perf_counter()
    for k in range(steps):
        chain.run_host(h_iq[(k + 1) & 1], outs[k % 3] if fields else None)
    chain.wait()
    return (time.perf_counter() - t0) / steps * 1e3


for name, f in (("H2D only", ()), ("+ small results", ("counts", "n_events", "nid4", "tsbk", "events", "event_data")),
                ("+ pcm", ("counts", "n_events", "nid4", "tsbk", "events", "event_data", "pcm")),
                ("+ records + flags (everything)", tuple(sizes))):
    print("%-32s %.2f ms per step   (D2H route 0x%x)" % (name, run(f), l.ddn_p25_chain_d2h_route(chain.h)), flush=True)
for _ in range(3):
    chain.run_pipelined(d_iq.data_ptr())
chain.wait()
t0 = time.perf_counter()
for _ in range(6):
    chain.run_pipelined(d_iq.data_ptr())
chain.wait()
print("device-resident pipelined        %.2f ms per step" % ((time.perf_counter() - t0) / 6 * 1e3))
