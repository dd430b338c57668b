import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
import ddn, orc
B, n, frame = 70, 30000, 432
x, _, _ = orc.synth_p25_disc(31, B, n, frame_dibits=frame, noise=500.0)
def run(dbg, cpw):
    os.environ["DDN_RX_DBG"] = str(dbg)
    rx = ddn.P25Rx(B, lock_symbols=frame - 24, use_matched_filter=0, channels_per_wave=cpw)
    return rx.run(x)
for cpw in (8, 16, 32):
    a = run(3072, cpw); b = run(1024, cpw)
    nbad = 0
    for c in range(B):
        ka, kb = int(a[2][c]), int(b[2][c])
        ra, sa = orc.unpack_records10(a[0][c, :ka]); rb, sb = orc.unpack_records10(b[0][c, :kb])
        m = min(ka, kb)
        d = np.flatnonzero(sa[:m].view(np.uint32) != sb[:m].view(np.uint32))
        if ka != kb or d.size:
            nbad += 1
            if nbad <= 4:
                i = int(d[0]) if d.size else m
                print("cpw", cpw, "ch", c, "counts", ka, kb, "first diff sym", i, "flags a", a[1][c, max(0,i-3):i+3], "b", b[1][c, max(0,i-3):i+3])
                print("   a", sa[max(0,i-2):i+3], "\n   b", sb[max(0,i-2):i+3])
    print("cpw", cpw, "bad channels", nbad, "of", B)
