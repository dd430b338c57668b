#!/usr/bin/env python3
"""Times the DMR / NXDN48 receive loop (ddn_fsk4_rx_run) at the bench shape: B channels x n discriminator samples built
from the committed captures (tests/golden), per-channel rotations.  usage: bench_rx4.py [B] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import ddn  # noqa: E402
import rx4  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
    l = ddn.lib()
    for name, cap, lpf, proto, rf in (("dmr c4fm", "iq_dmr_t3_ras_cc.npz", 2, ddn.FSK4_DMR, 0), ("dmr gfsk", "iq_dmr_t3_ras_cc.npz", 2, ddn.FSK4_DMR, 2),
                                      ("nxdn48", "iq_nxdn48.npz", 1, ddn.FSK4_NXDN48, 0)):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        disc = torch.from_numpy(rx4.capture_disc(cap, lpf)[:n + 4096]).cuda()
        idx = (torch.arange(n, device="cuda")[None, :] + (torch.arange(B, device="cuda")[:, None] * 37) % 4096)
        x = disc[idx].contiguous()
        rx = ddn.Fsk4Rx(B, proto, rf_mod=rf, handlers=bool(int(os.environ.get("HANDLERS", "0"))))
        if os.environ.get("FSK4_CPW"):  # channels per wavefront (the mixed chain runs 4 at 4096 channels)
            assert l.ddn_fsk4_rx_set_channels_per_wave(rx.h, int(os.environ["FSK4_CPW"])) == 0
        ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
        rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
        cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
        spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
        p = lambda t: t.data_ptr()
        assert l.ddn_fsk4_rx_set_timing(rx.h, 1) == 0
        ts = []
        for it in range(6):
            assert l.ddn_fsk4_rx_run(rx.h, p(x), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
            t = np.zeros(2, np.float32)
            assert l.ddn_fsk4_rx_get_timing(rx.h, t.ctypes.data) == 0
            ts.append(t.copy())
        t = np.median(np.stack(ts[2:]), axis=0)
        print("%-9s B=%d n=%d: matched filter %.3f ms, loop %.3f ms, syncs/ch %.1f, symbols/ch %.0f, in-frame share %.3f" % (
            name, B, n, t[0], t[1], float(ns.float().mean()), float(cnt.float().mean()), float((fl & 1).float().sum() / cnt.float().sum())))


if __name__ == "__main__":
    main()
