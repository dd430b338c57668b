"""A YSF frame encoder for the tests (the reference only decodes): FICH fields -> CRC16 -> four Golay(24,12) words -> K = 5 rate 1/2
(1 + D^3 + D^4, 1 + D + D^2 + D^4, four flush bits) -> 20 x 5 dibit interleave, behind the FUSION_SYNC word, in front of 360 payload
dibits.  Every stage is the inverse of a pinned decoder stage and is found from it: the Golay parity of each data bit and the CRC bits
are solved for with the CPU restatement's decoders (oracle/ddn_oracle_ysf.c), the convolutional code is checked by a round trip."""
import ctypes as C
import functools

import numpy as np

import orc

SYNC = np.array([int(c) for c in "31111311313113131131"], np.uint8)      # FUSION_SYNC (include/dsd-neo/core/sync_patterns.h:30)


def _o():
    o = orc.oracle()
    o.orc_golay_dmr_decode.argtypes = [C.c_int, C.c_void_p]
    o.orc_golay_dmr_decode.restype = C.c_int
    o.orc_ysf_crc16.argtypes = [C.c_void_p, C.c_int]
    o.orc_ysf_crc16.restype = C.c_uint16
    return o


@functools.lru_cache(maxsize=None)
def _golay_parity():
    """parity word (12 bits as an int) of each of the 12 unit data words: the one 24-bit word the decoder leaves untouched"""
    o = _o()
    par = []
    for i in range(12):
        found = None
        for p in range(4096):
            w = np.zeros(24, np.uint8)
            w[i] = 1
            w[12:] = [(p >> (11 - k)) & 1 for k in range(12)]
            a = w.copy()
            if o.orc_golay_dmr_decode(24, a.ctypes.data) and np.array_equal(a, w):
                found = p
                break
        assert found is not None
        par.append(found)
    return tuple(par)


def golay24(data12):
    p = 0
    for i in range(12):
        if data12[i]:
            p ^= _golay_parity()[i]
    return np.concatenate([np.asarray(data12, np.uint8), np.array([(p >> (11 - k)) & 1 for k in range(12)], np.uint8)])


def crc_bits(data32):
    """the 16 bits behind data32 that make ysf_crc16 over all 48 come out 0 (the function is affine in them)"""
    o = _o()

    def f(c16):
        b = np.concatenate([np.asarray(data32, np.uint8), np.asarray(c16, np.uint8)])
        return int(o.orc_ysf_crc16(b.ctypes.data, 48))

    f0 = f(np.zeros(16, np.uint8))
    cols = [f(np.eye(16, dtype=np.uint8)[j]) ^ f0 for j in range(16)]
    m = np.array([[(cols[j] >> (15 - i)) & 1 for j in range(16)] + [(f0 >> (15 - i)) & 1] for i in range(16)], np.uint8)
    r, piv = 0, []
    for col in range(16):                   # Gauss-Jordan over GF(2): sum_j c_j cols[j] = f0
        k = next((i for i in range(r, 16) if m[i, col]), None)
        if k is None:
            continue
        m[[r, k]] = m[[k, r]]
        for i in range(16):
            if i != r and m[i, col]:
                m[i] ^= m[r]
        piv.append(col)
        r += 1
    c = np.zeros(16, np.uint8)
    for i, col in enumerate(piv):
        c[col] = m[i, 16]
    assert f(c) == 0
    return c


def conv_k5(bits):
    sr, out = 0, []
    for b in list(bits) + [0, 0, 0, 0]:
        sr = ((sr << 1) | int(b)) & 0x1F
        out.append(((bin(sr & 0x19).count("1") & 1) << 1) | (bin(sr & 0x17).count("1") & 1))
    return np.array(out, np.uint8)


def fich_dibits(fi, dt, cm=1, bn=0, bt=0, fn=0, ft=0, mr=0, vp=1, st=0, sc=0, cs=0):
    v = lambda x, n: [(x >> (n - 1 - k)) & 1 for k in range(n)]
    b = np.array(v(fi, 2) + v(cs, 2) + v(cm, 2) + v(bn, 2) + v(bt, 2) + v(fn, 3) + v(ft, 3) + [0, 0] + v(mr, 3) + [vp] + v(dt, 2) + [st] + v(sc, 7),
                 np.uint8)
    f48 = np.concatenate([b, crc_bits(b)])
    tb = np.concatenate([golay24(f48[12 * i:12 * i + 12]) for i in range(4)])
    buf = conv_k5(tb)                       # 100 dibits; the decoder reads buf[j + 5 i] = input[i + 20 j]
    inp = np.zeros(100, np.uint8)
    for i in range(20):
        for j in range(5):
            inp[i + 20 * j] = buf[5 * i + j]
    return inp


def frame(rng, fi, dt, break_fich=False, **kw):
    """-> 480 dibits: sync + FICH + random payload (break_fich: enough FICH dibits flipped for Golay / CRC to fail)"""
    f = fich_dibits(fi, dt, **kw)
    if break_fich:
        k = rng.choice(100, 40, replace=False)
        f[k] ^= rng.integers(1, 4, 40).astype(np.uint8)
    return np.concatenate([SYNC, f, rng.integers(0, 4, 360).astype(np.uint8)])
