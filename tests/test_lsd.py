"""P25p1 low speed data, (16,8) cyclic code (src/protocol/p25/p25_lsd.c): oracle vs the compiled reference (all 256
parity bytes, every 0 / 1 / 2-bit error pattern of sample codewords, soft decode on noisy LLRs), GPU vs oracle, and the
LDU gather of the two LSD codewords."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import ddn
import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def _o():
    o = orc.oracle()
    o.orc_p25_lsd_fec_16x8.argtypes = [C.c_void_p]
    o.orc_p25_lsd_fec_16x8_soft.argtypes = [C.c_void_p, C.c_void_p]
    return o


def codeword(d):
    p = _o().orc_p25_lsd_parity(int(d))
    return np.array([(d >> (7 - k)) & 1 for k in range(8)] + [(p >> (7 - k)) & 1 for k in range(8)], np.uint8)


def cases(seed, n):
    rng = np.random.default_rng(FZ + seed)
    bits = np.zeros((n, 16), np.uint8)
    llr = np.zeros((n, 16), np.int16)
    for i in range(n):
        cw = codeword(rng.integers(0, 256))
        ne = int(rng.integers(0, 4))
        pos = rng.choice(16, ne, replace=False)
        mag = rng.integers(70, 400, 16)
        mag[pos] = rng.integers(0, 90, ne)                    # flipped bits tend to be the weak ones
        weak = rng.choice(16, int(rng.integers(0, 5)), replace=False)
        mag[weak] = np.minimum(mag[weak], rng.integers(0, 64, len(weak)))
        cw[pos] ^= 1
        bits[i] = cw
        llr[i] = (mag * np.where(cw == 1, 1, -1)).astype(np.int16)
    return bits, llr


@needs_ref
def test_lsd_oracle_vs_reference():
    r, o = orc.ref(), _o()
    tab = (C.c_uint8 * 256).in_dll(r, "lsd_parity")
    assert [o.orc_p25_lsd_parity(d) for d in range(256)] == list(tab)
    r.p25_lsd_fec_16x8.argtypes = [C.c_void_p]
    r.p25_lsd_fec_16x8_soft.argtypes = [C.c_void_p, C.c_void_p]
    for d in (0, 1, 0x5A, 0xFF, 0x80, 0x39):                  # every pattern of up to two flipped bits
        base = codeword(d)
        pats = [()] + [(i,) for i in range(16)] + [(i, j) for i in range(16) for j in range(i + 1, 16)]
        for pat in pats:
            a = base.copy()
            a[list(pat)] ^= 1
            b = a.copy()
            assert r.p25_lsd_fec_16x8(a.ctypes.data) == o.orc_p25_lsd_fec_16x8(b.ctypes.data) and np.array_equal(a, b)
    bits, llr = cases(3, 4000)
    for i in range(len(bits)):
        a, b = bits[i].copy(), bits[i].copy()
        l = np.ascontiguousarray(llr[i])
        ra, rb = r.p25_lsd_fec_16x8_soft(a.ctypes.data, l.ctypes.data), o.orc_p25_lsd_fec_16x8_soft(b.ctypes.data, l.ctypes.data)
        assert ra == rb and np.array_equal(a, b), i


@pytest.mark.gpu
def test_lsd_gpu_vs_oracle(built):
    l, o = ddn.lib(), _o()
    bits, llr = cases(4, 6000)
    for soft in (0, 1):
        got = bits.copy()
        ok = np.zeros(len(bits), np.uint8)
        assert l.ddn_fec_p25_lsd_host(got.ctypes.data, llr.ctypes.data if soft else None, len(bits), ok.ctypes.data) == 0
        for i in range(len(bits)):
            b = bits[i].copy()
            li = np.ascontiguousarray(llr[i])
            w = o.orc_p25_lsd_fec_16x8_soft(b.ctypes.data, li.ctypes.data) if soft else o.orc_p25_lsd_fec_16x8(b.ctypes.data)
            assert ok[i] == w and np.array_equal(got[i], b), (soft, i)
        assert 0 < ok.sum() < len(ok)
    one = codeword(0xC3)
    one[3] ^= 1
    assert l.p25_lsd_fec_16x8(one.ctypes.data) == 1 and np.array_equal(one, codeword(0xC3))
    two = codeword(0x11)
    two[[1, 9]] ^= 1
    lr = np.full(16, 300, np.int16)
    lr[[1, 9]] = 5
    assert l.p25_lsd_fec_16x8(two.copy().ctypes.data) == 0
    assert l.p25_lsd_fec_16x8_soft(two.ctypes.data, lr.ctypes.data) == 1 and np.array_equal(two, codeword(0x11))


def test_lsd_layout(built):
    import p25gen
    t = np.zeros(16, np.int32)
    assert ddn.lib().ddn_p25p1_layout_ldu_lsd(t.ctypes.data) in (863, 864)
    # the LSD sits between voice frames 8 and 9: after the last parity-word slot, skipping status positions
    d, p = p25gen.ldu1_positions()
    assert t[0] > p.max() + 72 and np.all(np.diff(t) >= 1) and not np.any(t % 36 == 35) and t[-1] < 864 - 72


# ---- CRC-CCITT16 of trunking blocks (src/protocol/p25/p25_crc.c:18-76) ---------------------------------------------------------
def _crc_cases(seed, n, nbytes=12):
    rng = np.random.default_rng(FZ + seed)
    b = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    o = orc.oracle()
    o.orc_p25_crc16_ok.argtypes = [C.c_void_p, C.c_int]
    for i in range(0, n, 2):                                  # make every other block valid: find its CRC by search-free algebra
        crc = 0
        for k in range(nbytes - 2):
            for j in range(7, -1, -1):
                bit = (int(b[i, k]) >> j) & 1
                crc = ((crc << 1) ^ 0x1021) & 0xFFFF if ((crc >> 15) & 1) ^ bit else (crc << 1) & 0xFFFF
        crc ^= 0xFFFF
        b[i, nbytes - 2], b[i, nbytes - 1] = crc >> 8, crc & 0xFF
    return b, o


@needs_ref
def test_crc16_oracle_vs_reference():
    b, o = _crc_cases(1, 600)
    r = orc.ref()
    r.crc16_lb_bridge.argtypes = [C.c_void_p, C.c_int]
    good = 0
    for i in range(len(b)):
        bits = np.unpackbits(b[i]).astype(np.int32)
        want = r.crc16_lb_bridge(bits.ctypes.data, 80)
        assert o.orc_p25_crc16_ok(np.ascontiguousarray(b[i]).ctypes.data, 10) == want
        good += want == 0
    assert good >= 300


@pytest.mark.gpu
def test_crc16_gpu_vs_oracle(built):
    l = ddn.lib()
    for nbytes in (12, 3, 30):
        b, o = _crc_cases(2 + nbytes, 3000, nbytes)
        ok = np.zeros(len(b), np.uint8)
        assert l.ddn_fec_p25_crc16_host(b.ctypes.data, nbytes, len(b), ok.ctypes.data) == 0
        want = np.array([o.orc_p25_crc16_ok(np.ascontiguousarray(b[i]).ctypes.data, nbytes - 2) == 0 for i in range(len(b))])
        assert np.array_equal(ok.astype(bool), want) and 1400 < ok.sum() < 1600
    b, _ = _crc_cases(9, 4, 12)
    assert l.crc16_lb_bridge(np.unpackbits(b[0]).astype(np.int32).ctypes.data, 80) == 0
    assert l.crc16_lb_bridge(np.unpackbits(b[1]).astype(np.int32).ctypes.data, 80) == 65535
    assert l.crc16_lb_bridge(np.zeros(300, np.int32).ctypes.data, 224) == 65535   # beyond the reference's 190-bit buffer
