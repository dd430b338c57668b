import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
import numpy as np
import ddn, orc
BASE, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed + 7919 * BASE)
passes = int(rng.choice([0, 0, 0, 1, 2]))
profile = int(rng.choice([2, 4, 4, 5, 1]))
blk = int(rng.choice([135, 200, 1000, 2048, 4096, 8192, 8191, 12345])) if passes == 0 else int(rng.choice([1024, 2048, 8192]))
blk = max(blk, 135 << passes)
squelch = float(rng.choice([0.0, 0.0, 0.0005, 0.02]))
B = int(rng.integers(1, 40))
fmt_cf32 = bool(rng.integers(0, 2))
n_calls = int(rng.integers(1, 4))
lens = [int(rng.integers(1, 4)) * blk for _ in range(n_calls - 1)] + [int(rng.integers(1, 3 * blk))]
if passes:
    lens[-1] = max(1 << passes, (lens[-1] >> passes) << passes)
n = sum(lens)
iq = orc.synth_c4fm_cu8(int(rng.integers(0, 1000)), B, n, sps=10 << passes)
if squelch > 0:
    iq[:, n // 3: n // 2] = 127
print("passes", passes, "profile", profile, "blk", blk, "sq", squelch, "B", B, "cf32", fmt_cf32, "lens", lens, "n", n, "quiet", n // 3, n // 2)
x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32) if fmt_cf32 else iq
b = ddn.Batch(B, lpf_profile=profile, block_len=blk, squelch_level=squelch, input_format=ddn.IN_CF32 if fmt_cf32 else ddn.IN_CU8)
if passes:
    b.set_decimation(passes)
got, pos = [], 0
for ln in lens:
    got.append(b.run_host(x[:, pos:pos + ln], ln)); pos += ln
got = np.concatenate(got, axis=1)
nb = 0
badlist = []
for c in range(B):
    fe = orc.OracleFrontEnd(profile=profile, squelch=squelch, downsample_passes=passes)
    want, pos = [], 0
    for ln in lens:
        want.append(fe.run_cu8(iq[c, pos:pos + ln], blk)); pos += ln
    want = np.concatenate(want)
    bad = np.flatnonzero(got[c].view(np.uint32) != want.view(np.uint32))
    if len(bad):
        nb += 1
        badlist.append(c)
        if nb <= 1:
            i = bad[0]
            print("ch", c, "nbad", len(bad), "first", i, "last", bad[-1], "got", got[c, i:i + 4], "want", want[i:i + 4])
            ob = blk >> passes
            for k in range(0, len(want), ob):
                g, w = got[c, k:k + ob], want[k:k + ob]
                print("  blk", k // ob, "gotzero", bool((g == 0).all()), "wantzero", bool((w == 0).all()), "eq", bool(np.array_equal(g.view(np.uint32), w.view(np.uint32))))
print("bad channels", nb, "of", B, badlist)
