import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
import numpy as np, ddn, orc
seed = 6
rng = np.random.default_rng(3000 + seed)
sps = int(rng.choice([4, 5, 5, 10, 8])); sym_rate = 6000 if sps == 4 else 4800; rate = sps * sym_rate
blk = int(rng.choice([333, 1000, 2048, 4096, 8192])); lpf = int(rng.integers(0, 2)); B = int(rng.integers(1, 20))
iq = orc.synth_dqpsk_f32(int(rng.integers(0, 999)), B, int(rng.integers(600, 2500)), sps, cfo=float(rng.choice([0.0, 0.002, 0.01])))
n = iq.shape[1]
n_calls = int(rng.integers(1, 4))
lens = [int(rng.integers(1, 3)) * blk for _ in range(n_calls - 1)]
if sum(lens) >= n - 8: lens = []
last = n - sum(lens)
if last % blk in (1, 2, 3): last -= 4
lens.append(last)
print("sps", sps, "rate", rate, "blk", blk, "lpf", lpf, "B", B, "n", n, "lens", lens)
b = ddn.CqpskBatch(B, rate=rate, sym_rate=sym_rate, lpf_enable=lpf, block_len=blk)
pos = 0; got = [[] for _ in range(B)]
for ln in lens:
    sym, cnt = b.run(iq[:, pos:pos + ln])
    for c in range(B): got[c].append(sym[c, :cnt[c]])
    pos += ln
for c in range(min(B, 4)):
    fe = orc.OracleCqpskFe(rate=rate, sym_rate=sym_rate, lpf_enable=lpf)
    want = []; pos = 0; bounds = []
    for ln in lens:
        w = fe.run(iq[c, pos:pos + ln], blk); want.append(w); bounds.append(len(w)); pos += ln
    want = np.concatenate(want); g = np.concatenate(got[c])
    bad = np.flatnonzero(g.view(np.uint32) != want.view(np.uint32))
    print(c, len(g), len(want), "bounds", np.cumsum(bounds), "first bad", bad[:4])
