#!/usr/bin/env python3
"""Receive loops of the configs[3] mix side by side: each loop alone, then every pair and all three on streams of their own
(handlers in the loops, 1365 channels each, one second of discriminator output).  Shows what the loops cost each other.
usage: python tools/loops_side_by_side.py [B]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ddn  # noqa: E402
import rx4  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1365
n = 48000
l = ddn.lib()
z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
p = lambda t: t.data_ptr()


def fsk4(name, cap, lpf, proto):
    disc = torch.from_numpy(rx4.capture_disc(cap, lpf)[:n + 4096]).cuda()
    idx = (torch.arange(n, device="cuda")[None, :] + (torch.arange(B, device="cuda")[:, None] * 37) % 4096)
    x = disc[idx].contiguous()
    rx = ddn.Fsk4Rx(B, proto, rf_mod=0, handlers=True)
    if os.environ.get("FSK4_CPW"):  # channels per wavefront (the mixed chain picks 4 at 4096 channels; a loop alone picks 1 up to 1536)
        assert l.ddn_fsk4_rx_set_channels_per_wave(rx.h, int(os.environ["FSK4_CPW"])) == 0
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
    cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
    spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
    keep = (x, rx, rec, fl, pay, cnt, ns, spos, spat, pre, prel)

    def run(stream):
        assert l.ddn_fsk4_rx_run(rx.h, p(x), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, stream) == 0
    return name, run, keep


def p25():
    voice, ctrl = bench.make_base_traffic(n)
    iq = np.stack([(voice if k == "voice" else ctrl)[i] for k, i in (bench.channel_source(c) for c in range(B))])
    d_iq = torch.from_numpy(iq).cuda()
    fe = ddn.Batch(B, block_len=8192)
    disc = z((B, n), torch.float32)
    fe.run_device(d_iq.data_ptr(), n, disc.data_ptr(), None)
    torch.cuda.synchronize()
    rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=int(os.environ.get("P25_CPW", "8")), handlers=True)
    ms = l.ddn_p25_rx_max_symbols(rx.h, n)
    rec, fl, cnt = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B,), torch.int32)
    ev, nev = z((B, 256, 4), torch.int32), z((B,), torch.int32)
    assert l.ddn_p25_rx_set_events(rx.h, ev.data_ptr(), nev.data_ptr(), 256) == 0
    keep = (disc, rx, rec, fl, cnt, ev, nev)

    def run(stream):
        assert l.ddn_p25_rx_run(rx.h, p(disc), n, p(rec), p(fl), p(cnt), ms, stream) == 0
    return "p25", run, keep


loops = [p25(), fsk4("dmr", "iq_dmr_t3_ras_cc.npz", 2, ddn.FSK4_DMR), fsk4("nxdn48", "iq_nxdn48.npz", 1, ddn.FSK4_NXDN48)]
if os.environ.get("TWINS"):  # two objects of the same kernel side by side: what two DIFFERENT kernels cost each other beyond that
    loops += [fsk4("dmr'", "iq_dmr_t3_ras_cc.npz", 2, ddn.FSK4_DMR), p25()]
streams = [torch.cuda.Stream() for _ in loops]
combos = [c for k in (1, 2, 3) for c in itertools.combinations(range(3), k)]
if os.environ.get("TWINS"):
    combos = [(1,), (1, 3), (0,), (0, 4), (1, 2)]
for k in (1,):
    for combo in combos:
        ts = []
        for it in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in combo:
                streams[i].wait_event(e0)
                loops[i][1](streams[i].cuda_stream)
            for i in combo:
                torch.cuda.current_stream().wait_stream(streams[i])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-18s %.3f ms (matched filter + loop, B = %d each)" % (" + ".join(loops[i][0] for i in combo), float(np.median(ts[1:])), B), flush=True)
