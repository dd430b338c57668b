import sys, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/dsd-neo_amd/bindings')
import orc
from test_oracle_p25p2_xcch import oracle_xcch, oracle_duid, crc12_ok, crc16_ok
iq = np.load('/root/repo/tests/golden/iq_p25p2_cc.npz')['iq'].astype(np.float64) - 127.5
x = iq[:,0] + 1j*iq[:,1]
# channel filter: moving average of 4 samples
k = np.ones(5)/5
x = np.convolve(x, k, mode='same')
sps = 8
d = np.angle(x[sps:] * np.conj(x[:-sps]))
best = None
for ph in range(sps):
    s = d[ph::sps]
    # distance to nearest of +-pi/4, +-3pi/4
    t = np.abs(np.abs(s) - np.pi/4); u = np.abs(np.abs(s) - 3*np.pi/4)
    e = np.minimum(t,u).mean()
    if best is None or e < best[0]: best = (e, ph)
print("timing", best)
s = d[best[1]::sps]
dib = np.where(s >= 0, np.where(s < np.pi/2, 0, 1), np.where(s > -np.pi/2, 2, 3)).astype(np.uint8)
rel = np.minimum(np.abs(np.abs(np.abs(s) - np.pi/2)) * 400, 255).astype(np.int16)   # crude reliability: distance from the pi/2 boundary
sync = np.array([1,1,1,3,1,1,3,1,1,1,1,3,3,3,1,3,3,3,3,3], np.uint8)
hits = []
for inv in (0, 1):
    pat = sync ^ (2 if inv else 0)
    for i in range(len(dib) - 20):
        if (dib[i:i+20] != pat).sum() <= 1:
            hits.append((i, inv))
print(len(dib), "dibits; sync hits", hits[:20])

first = hits[0][0] + 20
slots = []
t0 = first % 180
off = [0, 1, 74, 75, 244, 245, 318, 319]
res = []
for t in range(t0, len(dib) - 180, 180):
    dd = dib[t:t+180]
    bits = np.zeros(360, np.uint8); bits[0::2] = dd >> 1; bits[1::2] = dd & 1
    llr = np.repeat(np.maximum(rel[t:t+180], 1), 2).astype(np.int16)
    w = 0
    for k in range(8): w = (w << 1) | int(bits[off[k]])
    r8 = np.minimum(np.abs(llr[off]), 255).astype(np.uint8)
    du = oracle_duid(w, r8)
    ec1, pl1, u1 = oracle_xcch(1, bits, llr)
    ec0, pl0, u0 = oracle_xcch(0, bits, llr)
    res.append((t, hex(w), du, ec1, crc12_ok(pl1, 168) if ec1 >= 0 else -1, crc16_ok(pl1) if ec1 >= 0 else -1, ec0, crc12_ok(pl0, 144) if ec0 >= 0 else -1))
for r in res[:40]: print(r)
print("SACCH RS ok:", sum(1 for r in res if r[3] >= 0), "crc16 ok:", sum(1 for r in res if r[5] == 1), "crc12 ok:", sum(1 for r in res if r[4] == 1), "of", len(res))
o = orc.oracle()
o.orc_isch_lookup.argtypes = [C.c_uint64]; o.orc_isch_lookup.restype = C.c_int
out = []
for t in range(t0, len(dib) - 180, 180):
    dd = dib[t:t+180]
    bits = np.zeros(360, np.uint8); bits[0::2] = dd >> 1; bits[1::2] = dd & 1
    w = 0
    for k in range(40): w = (w << 1) | int(bits[320 + k])
    v = o.orc_isch_lookup(C.c_uint64(w))
    out.append((t, hex(w), v, (v >> 5) & 3, (v >> 3) & 3, (v >> 2) & 1, v & 3) if v >= 0 else (t, hex(w), v))
for r in out[:30]: print(r)
