#!/usr/bin/env python3
"""Stress run for the receive loop's unfiltered-row switch: 32 channels x 220000 samples of carriers that come and go at random
(silence or noise in the gaps, several calls of odd lengths), the default loop against ddn_p25_rx_set_debug_flags bit 16384 (every tile's
unfiltered samples staged): records, flags, counts, decisions must be identical.  usage: python tools/stress_unfiltered_row.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ddn, p25gen
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_rx_handlers_gpu as T
bad = 0
for seed in range(6):
    rng = np.random.default_rng(1000 + seed)
    B, n = 32, 220000
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        pos = int(rng.integers(0, 3000))
        while pos < n - 8000:
            ln = int(rng.integers(4000, 30000))
            s = T._traffic(int(rng.integers(0, 10000)), min(ln, n - pos), [90.0, 4000.0, 9000.0][int(rng.integers(0, 3))])
            x[c, pos:pos + len(s)] = s
            pos += len(s)
            gap = int(rng.integers(3000, 45000))
            if rng.integers(0, 3) == 0:
                g = min(gap, n - pos)
                if g > 0:
                    x[c, pos:pos + g] = rng.normal(0, float(rng.choice([300.0, 2000.0, 8000.0])), g).astype(np.float32)
            pos += gap
    outs = []
    for dbg in (0, 16384):
        # several calls of odd lengths: the hint crosses call boundaries too
        rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=[4, 8][seed % 2], handlers=True, max_events=8192, debug_flags=dbg)
        acc = []
        cuts = [0, 50001, 50002, 131072, 131200, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            rec, fl, cnt = rx.run(x[:, a:b])
            acc.append((rec.copy(), fl.copy(), cnt.copy(), rx.events.copy(), rx.n_events.copy(), rx.event_data.copy()))
        outs.append(acc)
    same = all(np.array_equal(p, q) for ca, cb in zip(outs[0], outs[1]) for p, q in zip(ca, cb))
    print("seed", seed, "identical" if same else "DIFFERENT", "symbols", int(sum(int(a[2].sum()) for a in outs[0])))
    bad += 0 if same else 1
print("FAILED" if bad else "ALL IDENTICAL")
