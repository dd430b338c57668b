#!/usr/bin/env python3
"""Can a device -> pinned-host copy run beside the receive loop?  The loop (matched filter + k_p25_rxw, 4096 x 48000 of the bench
traffic) on one stream; on another, a 130 MB hipMemcpyAsync D2H that is released when the loop's stream reaches the call.  Prints the
loop's time alone, the copy's time alone, and both when they run together (when the copy ended relative to the loop's start)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import ddn

B, n = 4096, 48000
voice, ctrl = bench.make_base_traffic(n)
idx = [bench.channel_source(c) for c in range(B)]
iq = np.stack([(voice if k == "voice" else ctrl)[i] for k, i in idx])
d_iq = torch.from_numpy(iq).cuda()
fe = ddn.Batch(B, block_len=8192)
disc = torch.zeros((B, n), dtype=torch.float32, device="cuda")
l = ddn.lib()
rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=8, handlers=True)
ms = l.ddn_p25_rx_max_symbols(rx.h, n)
rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
ev = torch.zeros((B, 256, 4), dtype=torch.int32, device="cuda")
nev = torch.zeros((B,), dtype=torch.int32, device="cuda")
assert l.ddn_p25_rx_set_events(rx.h, ev.data_ptr(), nev.data_ptr(), 256) == 0
nb = 130 << 20
d_src = torch.zeros(nb, dtype=torch.uint8, device="cuda")
h_dst = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
fe.run_device(d_iq.data_ptr(), n, disc.data_ptr(), None)
torch.cuda.synchronize()


def loop(stream):
    assert l.ddn_p25_rx_run(rx.h, disc.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, stream.cuda_stream) == 0


for mode in ("loop alone", "copy alone", "both: copy released at the loop's start", "both: copy released 1 ms of GPU time before the loop"):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record(sA)
        if mode == "both: copy released 1 ms of GPU time before the loop":
            with torch.cuda.stream(sB):
                sB.wait_event(e0)
                c0.record(sB)
                h_dst.copy_(d_src, non_blocking=True)
                c1.record(sB)
            torch.cuda._sleep(2400000)  # default stream ~1 ms
            sA.wait_stream(torch.cuda.current_stream())
        if mode != "copy alone":
            loop(sA)
        e1.record(sA)
        if mode in ("copy alone", "both: copy released at the loop's start"):
            with torch.cuda.stream(sB):
                sB.wait_event(e0)
                c0.record(sB)
                h_dst.copy_(d_src, non_blocking=True)
                c1.record(sB)
        torch.cuda.synchronize()
        if rep == 2:
            msg = "%-56s loop stream %.2f ms" % (mode, e0.elapsed_time(e1))
            if mode != "loop alone":
                msg += " | copy ran %.2f .. %.2f ms after the start (%.1f GB/s)" % (e0.elapsed_time(c0), e0.elapsed_time(c1),
                                                                                  nb / (c0.elapsed_time(c1) * 1e-3) / 1e9)
            print(msg, flush=True)
