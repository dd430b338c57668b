#!/usr/bin/env python3
"""Receive-loop kernel time on the bench traffic (4096 channels x 48000 samples, half voice half control): configured in-frame
lengths (the round-2 shape) against the reference's handlers inside the loop, per lanes-per-wave choice.
usage: python tools/bench_rx_handlers.py [B] [n]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
voice, ctrl = bench.make_base_traffic(n)
idx = [bench.channel_source(c) for c in range(B)]
if os.environ.get("TRAFFIC") in ("voice", "ctrl"):  # timing experiments: one kind of channel only
    idx = [(os.environ["TRAFFIC"], i) for _, i in idx]
iq = np.stack([(voice if k == "voice" else ctrl)[i] for k, i in idx])
d_iq = torch.from_numpy(iq).cuda()
fe = ddn.Batch(B, block_len=8192)
disc = torch.zeros((B, n), dtype=torch.float32, device="cuda")
l = ddn.lib()
locks = np.array([840 if k == "voice" else int(os.environ.get("LOCK_CC", "156")) for k, _ in idx], np.int32)
for mode, cpw in [(m, int(c)) for m, c in (x.split(":") for x in os.environ.get("MODES", "locks:8,handlers:8,handlers:16,locks:16").split(","))]:
    # (the product library reads no environment: the schedule selectors and the filter-in-the-loop switch go through the setters)
    rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=cpw, handlers=(mode == "handlers"),
                   debug_flags=int(os.environ.get("DDN_RX_DBG", "0"), 0) & 0x7FFFFFFF, filter_in_loop=os.environ.get("FIL", "0") == "1")
    if mode == "locks":
        assert l.ddn_p25_rx_set_lock_symbols(rx.h, locks.ctypes.data) == 0
    ms = l.ddn_p25_rx_max_symbols(rx.h, n)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ev = torch.zeros((B, 256, 4), dtype=torch.int32, device="cuda")
    nev = torch.zeros((B,), dtype=torch.int32, device="cuda")
    if mode == "handlers":
        assert l.ddn_p25_rx_set_events(rx.h, ev.data_ptr(), nev.data_ptr(), 256) == 0
    assert l.ddn_p25_rx_set_timing(rx.h, 1) == 0
    for step in range(6):
        fe.run_device(d_iq.data_ptr(), n, disc.data_ptr(), None)
        assert l.ddn_p25_rx_run(rx.h, disc.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
        torch.cuda.synchronize()
        if step == 2:
            assert l.ddn_p25_rx_set_timing(rx.h, 1) == 0
    t = (C.c_float * 2)()
    k = C.c_int()
    assert l.ddn_p25_rx_get_timing_avg(rx.h, t, C.byref(k)) == 0
    flc = fl.cpu().numpy()
    extra = ""
    if mode == "handlers":
        e = ev.cpu().numpy()
        ne = nev.cpu().numpy()
        allev = np.concatenate([e[c, :min(ne[c], 256)] for c in range(B)])
        if int(os.environ.get("DDN_RX_DBG", "0"), 0) & 131072:
            for kd in (1, 2):
                a = allev[(allev[:, 1] == kd) & ((allev[:, 2] & 15) == 0)]
                extra += " | kind %d stamps(cycles): %.0f %.0f %.0f" % (kd, (a[:, 2] >> 4).mean() * 16, (a[:, 3] & 0xFFFF).mean() * 16,
                                                                      ((a[:, 3] >> 16) & 0xFFFF).mean() * 16)
        elif int(os.environ.get("DDN_RX_DBG", "0"), 0) & 1073741824:  # DDN_RX_CYCLES build: e[2] = cycles the request lay unserved
            for kd in (1, 2):
                a = allev[allev[:, 1] == kd]
                extra += " | kind %d: pick-up delay mean %.0f pct[10,50,80,90,95,99] %s, service mean %.0f" % (
                    kd, a[:, 2].mean(), np.percentile(a[:, 2], [10, 50, 80, 90, 95, 99]).astype(int).tolist(), a[:, 3].mean())
        elif int(os.environ.get("DDN_RX_DBG", "0"), 0) & 65536:
            for kd in (1, 2):
                a = allev[allev[:, 1] == kd]
                for path in (0, 1):
                    v = a[a[:, 2] == path][:, 3]
                    if len(v):
                        extra += " | kind %d path %d: n %d cycles mean %.0f max %d" % (kd, path, len(v), v.mean(), v.max())
                        extra += " pct[10,50,80,90,95,99] %s" % np.percentile(v, [10, 50, 80, 90, 95, 99]).astype(int).tolist()
        else:
            extra = " events/ch %.1f nid_ok %.3f tsbk_crc %.3f" % (ne.mean(), (allev[allev[:, 1] == 1][:, 2] > 0).mean(),
                                                                   (allev[allev[:, 1] == 2][:, 3] & 1).mean())
    if mode == "handlers" and int(os.environ.get("DDN_RX_DBG", "0"), 0) & 65536:
        tot = np.zeros(2)
        for c in range(0, 64):
            o = (C.c_longlong * 2)()
            assert l.ddn_p25_rx_debug_counters(rx.h, c, o) == 0
            tot += [o[0], o[1]]
        extra += " | lane wait: %.0f cycles per request (%d requests)" % (tot[1] / max(tot[0], 1), tot[0])
    if int(os.environ.get("DDN_RX_DBG", "0"), 0) & 8192:      # library built with EXTRA=-DDDN_RX_CYCLES=1
        r = rec.cpu().numpy().reshape(B, -1)
        tails = np.stack([r[c, -192:] for c in range(0, B, cpw)]).copy().view(np.int64).reshape(-1, 3, 8)
        tiles = (n + 127) // 128
        for w, nm in enumerate(["recurrence", "loader", "winprep"]):
            extra += "\n   %s busy/tile %.0f wait/tile %.0f cycles" % (nm, tails[:, w, 0].mean() / tiles, tails[:, w, 1].mean() / tiles)
        extra += "\n   staging wave per tile (both halves): stage %.0f drain %.0f window summaries %.0f cycles" % tuple(
            tails[:, 2, 2 + k].mean() / tiles for k in range(3))
        tt = tails[:, 0]
        for k in range(3):
            extra += "\n   trip kind %d: %.2f per tile at %.0f cycles (share of busy %.2f)" % (
                k, tt[:, 5 + k].mean() / tiles, tt[:, 2 + k].sum() / max(1, tt[:, 5 + k].sum()), tt[:, 2 + k].sum() / max(1, tt[:, 0].sum()))
    if int(os.environ.get("DDN_RX_DBG", "0"), 0) & 8192:
        sec = np.stack([r[c, -256:-192] for c in range(0, B, cpw)]).copy().view(np.int64)
        nstd = max(1, tt[:, 5].sum())
        extra += "\n   std trip sections (cycles per trip): top %.0f search %.0f mean %.0f inframe %.0f hunt %.0f emit %.0f" % tuple(
            sec[:, k].sum() / nstd for k in range(6))
        extra += "\n   bulk hunting passes: %.2f per tile at %.0f cycles" % (sec[:, 7].sum() / (len(sec) * tiles), sec[:, 6].sum() / max(1, sec[:, 7].sum()))
        blk = np.stack([r[c, -384:-320] for c in range(0, B, cpw)]).copy().view(np.int64).sum(axis=0)
        extra += "\n   bulk pass per owner lane (%.2f owners per pass, %.1f symbols each): set-up %.0f masks + slip chain %.0f means + sync test %.0f stores %.0f window reduction %.0f owner's words %.0f cycles" % (
            (blk[7] / max(1, sec[:, 7].sum()), blk[6] / max(1, blk[7])) + tuple(blk[k] / max(1, blk[7]) for k in range(6)))
        run = np.stack([r[c, -320:-256] for c in range(0, B, cpw)]).copy().view(np.int64).sum(axis=0)
        extra += "\n   lean runs: %.2f per tile, %.2f trips and %.2f phases each (%.2f of them polling a handler mailbox), %.0f cycles inside the run per trip" % (
            run[0] / (len(sec) * tiles), run[1] / max(1, run[0]), run[7] / max(1, run[0]), run[5] / max(1, run[0]), run[6] / max(1, run[1]))
        extra += "\n   a lean run's pass: %.0f cycles from the pass's top to the bulk test's end, %.0f from there to the run's first trip" % (
            run[2] / max(1, run[0]), run[3] / max(1, run[0]))
    if mode == "handlers" and int(os.environ.get("DDN_RX_DBG", "0"), 0) & 8192:
        hm = np.stack([r[c, -448:-384] for c in range(0, B, cpw)]).copy().view(np.int64).sum(axis=0)
        extra += "\n   handler wave per tile: filter passes %.2f at %.0f cycles, decisions %.2f at %.0f cycles, idle polls %.1f" % (
            hm[1] / (len(sec) * tiles), hm[0] / max(1, hm[1]), hm[3] / (len(sec) * tiles), hm[2] / max(1, hm[3]), hm[4] / (len(sec) * tiles))
    print("%-8s cpw %2d: loop %.3f ms (mf %.3f) in-frame share %.3f syncs/ch %.1f%s" % (
        mode, cpw, t[1], t[0], (flc & 1).mean() * ms / (n / 10), (flc & 2).sum() / B, extra), flush=True)
