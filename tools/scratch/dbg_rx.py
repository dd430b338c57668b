import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
import numpy as np, ddn, orc
B, n, frame = 4, 30000, 432
x, _, _ = orc.synth_p25_disc(31, B, n, frame_dibits=frame, noise=500.0)
rx = ddn.P25Rx(B, lock_symbols=frame - 24, use_matched_filter=1, channels_per_wave=16)
rec, fl, cnt = rx.run(x)
for c in range(B):
    o = orc.OracleP25Rx(lock_symbols=frame - 24, use_filter=1)
    sym, r4, f = o.run(x[c])
    k = int(cnt[c])
    g4, gs = orc.unpack_records10(rec[c, :k])
    m = min(k, len(sym))
    bad = np.flatnonzero(gs[:m].view(np.uint32) != sym[:m].view(np.uint32))
    print(c, k, len(sym), "first bad", bad[:5], "acc", np.flatnonzero(f & 2)[:6], np.flatnonzero(fl[c,:k] & 2)[:6])
    if len(bad):
        b = bad[0]
        print(" gpu", gs[b-2:b+3], fl[c, b-2:b+3], " orc", sym[b-2:b+3], f[b-2:b+3])
