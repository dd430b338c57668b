"""GPU parity: batched Golay(24,12,8)/(18,6,8) and RS(24,12,13)/(24,16,9)/(36,20,17) decoders vs the CPU oracle
(pinned to the reference's check_and_fix_* in tests/test_oracle_rs.py) — data bits, status and fixed counts identical,
through the batch entry points and the reference-named single-codeword drop-ins."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import ddn
import fecgen
from test_oracle_rs import oracle_golay, oracle_rs

pytestmark = pytest.mark.gpu
CODE_ID = {"24_12_13": 0, "24_16_9": 1, "36_20_17": 2}


@pytest.mark.parametrize("length", [6, 12])
def test_golay_batch(built, length):
    rng = np.random.default_rng(FZ + 40 + length)
    d, p = fecgen.gen_golay24(rng, 20000, length)
    d[:2000] = rng.integers(0, 2, (2000, length))
    p[:2000] = rng.integers(0, 2, (2000, 12))
    d[2000, 1] = 2                       # invalid bit value -> rc 1, untouched
    p[2001, 11] = 7
    want_d, want_rc, want_fx = oracle_golay(d, p)
    got = d.copy()
    st = np.zeros(len(d), np.uint8)
    fx = np.zeros(len(d), np.int32)
    assert ddn.lib().ddn_fec_golay24_host(length, got.ctypes.data, p.ctypes.data, len(d), st.ctypes.data,
                                          fx.ctypes.data) == 0
    assert np.array_equal(got, want_d) and np.array_equal(st, want_rc) and np.array_equal(fx, want_fx)


@pytest.mark.parametrize("code", list(fecgen.P25_RS_CODES))
def test_rs_batch(built, code):
    rng = np.random.default_rng(FZ + 50 + CODE_ID[code])
    d, p = fecgen.gen_p25_rs(rng, code, 5000, max_extra=4)
    d[:300] = rng.integers(0, 2, d[:300].shape)
    p[:300] = rng.integers(0, 2, p[:300].shape)
    d[300] = 0
    p[300] = 0                                               # all-zero word: clean
    d[301, 0, 0] = 5                                         # non-binary byte counts as a set bit, rewritten as 1
    want_d, want_rc = oracle_rs(code, d, p)
    got = d.copy()
    st = np.zeros(len(d), np.uint8)
    assert ddn.lib().ddn_fec_p25_rs_host(CODE_ID[code], got.ctypes.data, p.ctypes.data, len(d), st.ctypes.data) == 0
    assert np.array_equal(st, want_rc)
    assert np.array_equal(got, want_d)
    assert 0 < st.sum() < len(st)


def test_dropin_names(built):
    rng = np.random.default_rng(FZ + 60)
    l = ddn.lib()
    for length, fn in ((6, l.check_and_fix_golay_24_6), (12, l.check_and_fix_golay_24_12)):
        d, p = fecgen.gen_golay24(rng, 12, length)
        want_d, want_rc, want_fx = oracle_golay(d, p)
        for i in range(len(d)):
            x = d[i].copy()
            f = C.c_int(-1)
            assert fn(x.ctypes.data, p[i].ctypes.data, C.byref(f)) == want_rc[i]
            assert f.value == want_fx[i] and np.array_equal(x, want_d[i])
    for code, fn in (("24_12_13", l.check_and_fix_reedsolomon_24_12_13), ("24_16_9", l.check_and_fix_reedsolomon_24_16_9),
                     ("36_20_17", l.check_and_fix_redsolomon_36_20_17)):
        d, p = fecgen.gen_p25_rs(rng, code, 8)
        want_d, want_rc = oracle_rs(code, d, p)
        for i in range(len(d)):
            x = d[i].copy()
            assert fn(x.ctypes.data, p[i].ctypes.data) == want_rc[i]
            assert np.array_equal(x, want_d[i])


def test_hamming_soft_batch(built):
    from test_oracle_rs import gen_soft_reliab, oracle_hamming_soft
    rng = np.random.default_rng(FZ + 71)
    n = 20000
    d = rng.integers(0, 2, (n, 6)).astype(np.uint8)
    p = np.stack([d[:, 0] ^ d[:, 1] ^ d[:, 2] ^ d[:, 5], d[:, 0] ^ d[:, 1] ^ d[:, 3] ^ d[:, 5],
                  d[:, 0] ^ d[:, 2] ^ d[:, 3] ^ d[:, 4], d[:, 1] ^ d[:, 2] ^ d[:, 3] ^ d[:, 4]], axis=1)
    bits = np.ascontiguousarray(np.concatenate([d, p], axis=1).astype(np.uint8))
    flips = rng.random(bits.shape) < rng.choice([0.0, 0.08, 0.2], (n, 1))
    bits ^= flips.astype(np.uint8)
    rel = gen_soft_reliab(rng, bits.shape, flips)
    bits[10, 3] = 2
    want, wrc = oracle_hamming_soft(bits, rel)
    out = np.zeros_like(bits)
    st = np.zeros(n, np.uint8)
    assert ddn.lib().ddn_fec_hamming_10_6_3_soft_host(bits.ctypes.data, rel.ctypes.data, n, out.ctypes.data, st.ctypes.data) == 0
    assert np.array_equal(st, wrc) and np.array_equal(out, want)
    one = np.zeros(10, np.uint8)
    assert ddn.lib().hamming_10_6_3_soft(bits[7].ctypes.data, rel[7].ctypes.data, one.ctypes.data) == wrc[7]
    assert np.array_equal(one, want[7])


@pytest.mark.parametrize("length", [6, 12])
def test_golay_soft_batch(built, length):
    from test_oracle_rs import gen_soft_reliab, oracle_golay_soft
    rng = np.random.default_rng(FZ + 73 + length)
    n = 6000
    d = np.zeros((n, length), np.uint8)
    p = np.zeros((n, 12), np.uint8)
    flips = np.zeros((n, length + 12), bool)
    for i in range(n):
        d12 = np.zeros(12, np.uint8)
        d12[12 - length:] = rng.integers(0, 2, length)
        w = np.concatenate([d12[12 - length:], fecgen.golay24_encode(d12)])
        pos = rng.choice(length + 12, int(rng.integers(0, 7)), replace=False)
        w[pos] ^= 1
        flips[i, pos] = True
        d[i], p[i] = w[:length], w[length:]
    rel = gen_soft_reliab(rng, flips.shape, flips)
    d[20, 0] = 3
    want, wrc, wfx = oracle_golay_soft(d, p, rel)
    got = d.copy()
    st = np.zeros(n, np.uint8)
    fx = np.zeros(n, np.int32)
    assert ddn.lib().ddn_fec_golay24_soft_host(length, got.ctypes.data, p.ctypes.data, rel.ctypes.data, n, st.ctypes.data,
                                               fx.ctypes.data) == 0
    assert np.array_equal(st, wrc) and np.array_equal(fx, wfx) and np.array_equal(got, want)
    fn = ddn.lib().check_and_fix_golay_24_6_soft if length == 6 else ddn.lib().check_and_fix_golay_24_12_soft
    x = d[5].copy()
    f = C.c_int(-1)
    assert fn(x.ctypes.data, p[5].ctypes.data, rel[5].ctypes.data, C.byref(f)) == wrc[5]
    assert f.value == wfx[5] and np.array_equal(x, want[5])


@pytest.mark.parametrize("code", list(fecgen.P25_RS_CODES))
def test_rs_soft_reliability_batch(built, code):
    """Hard decode, then ranked-erasure retries: data and status equal the oracle pinned to p25p1_rs_*_soft_reliability."""
    from test_oracle_rs import gen_rs_soft, oracle_rs_soft_rel
    rng = np.random.default_rng(FZ + 400 + CODE_ID[code])
    d, p, drel, prel = gen_rs_soft(rng, code, 4000)
    want, wrc = oracle_rs_soft_rel(code, d, p, drel, prel)
    got = d.copy()
    st = np.zeros(len(d), np.uint8)
    assert ddn.lib().ddn_fec_p25_rs_soft_host(CODE_ID[code], got.ctypes.data, p.ctypes.data, drel.ctypes.data,
                                              prel.ctypes.data, len(d), st.ctypes.data) == 0
    assert np.array_equal(st, wrc)
    assert np.array_equal(got, want)
    assert 0 < st.sum() < len(st)
    fn = {"24_12_13": ddn.lib().p25p1_rs_24_12_13_soft_reliability, "24_16_9": ddn.lib().p25p1_rs_24_16_9_soft_reliability,
          "36_20_17": ddn.lib().p25p1_rs_36_20_17_soft_reliability}[code]
    for i in (0, 1, 2, 3):
        x = d[i].copy()
        assert fn(x.ctypes.data, p[i].ctypes.data, drel[i].ctypes.data, prel[i].ctypes.data) == wrc[i]
        assert np.array_equal(x, want[i])
