"""CPU: Golay(24,12,8) / (18,6,8) and RS(24,12,13) / (24,16,9) / (36,20,17) restatement (oracle/ddn_oracle_rs.c) pinned
bit for bit against the reference's compiled check_and_fix_* entry points (oracle/_ref)."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import fecgen
import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
VP = C.c_void_p


def oracle_golay(data, par):
    o = orc.oracle()
    o.orc_golay_24_decode.argtypes = [VP, C.c_int, VP, VP]
    n, ln = data.shape
    out = data.copy()
    rc = np.zeros(n, np.int32)
    fx = np.zeros(n, np.int32)
    f = C.c_int(0)
    for i in range(n):
        rc[i] = o.orc_golay_24_decode(out[i].ctypes.data, ln, par[i].ctypes.data, C.byref(f))
        fx[i] = f.value
    return out, rc, fx


def oracle_rs(code, data, par):
    n_par, n_data, t = fecgen.P25_RS_CODES[code]
    o = orc.oracle()
    o.orc_p25_rs_decode.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int]
    out = data.copy()
    rc = np.zeros(data.shape[0], np.int32)
    for i in range(data.shape[0]):
        rc[i] = o.orc_p25_rs_decode(out[i].ctypes.data, par[i].ctypes.data, n_par, n_data, t)
    return out, rc


def ref_golay(data, par):
    r = orc.ref()
    fn = r.check_and_fix_golay_24_6 if data.shape[1] == 6 else r.check_and_fix_golay_24_12
    fn.argtypes = [VP, VP, VP]
    out = data.copy()
    rc = np.zeros(data.shape[0], np.int32)
    fx = np.zeros(data.shape[0], np.int32)
    f = C.c_int(0)
    for i in range(data.shape[0]):
        rc[i] = fn(out[i].ctypes.data, par[i].ctypes.data, C.byref(f))
        fx[i] = f.value
    return out, rc, fx


def ref_rs(code, data, par):
    r = orc.ref()
    fn = {"24_12_13": r.check_and_fix_reedsolomon_24_12_13, "24_16_9": r.check_and_fix_reedsolomon_24_16_9,
          "36_20_17": r.check_and_fix_redsolomon_36_20_17}[code]
    fn.argtypes = [VP, VP]
    out = data.copy()
    rc = np.zeros(data.shape[0], np.int32)
    for i in range(data.shape[0]):
        rc[i] = fn(out[i].ctypes.data, par[i].ctypes.data)
    return out, rc


def test_encoders_give_codewords(built):
    rng = np.random.default_rng(FZ + 2)
    for code in fecgen.P25_RS_CODES:
        d, p = fecgen.gen_p25_rs(rng, code, 20, max_extra=-fecgen.P25_RS_CODES[code][2])   # no errors
        out, rc = oracle_rs(code, d, p)
        assert not rc.any() and np.array_equal(out, d)
    for ln in (6, 12):
        d, p = fecgen.gen_golay24(np.random.default_rng(FZ + 3), 1, ln)
    d12 = rng.integers(0, 2, (50, 12)).astype(np.uint8)
    p = np.stack([fecgen.golay24_encode(x) for x in d12])
    out, rc, fx = oracle_golay(d12, p)
    assert not rc.any() and not fx.any() and np.array_equal(out, d12)


@needs_ref
@pytest.mark.parametrize("length", [6, 12])
def test_golay_matches_reference(built, length):
    rng = np.random.default_rng(FZ + 10 + length)
    d, p = fecgen.gen_golay24(rng, 6000, length)
    # plus pure noise words and invalid (non-binary) inputs
    d[:500] = rng.integers(0, 2, (500, length))
    p[:500] = rng.integers(0, 2, (500, 12))
    d[500, 2] = 2
    p[501, 7] = 3
    a = oracle_golay(d, p)
    b = ref_golay(d, p)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[1].sum() > 0 and (a[2] == 3).sum() > 0 and (a[2] == 2).sum() > 0


@needs_ref
def test_golay_exhaustive_error_patterns(built):
    """Every error pattern of weight <= 4 on one codeword (length 12): data, return code and fixed count."""
    import itertools
    rng = np.random.default_rng(FZ + 5)
    d0 = rng.integers(0, 2, 12).astype(np.uint8)
    w0 = np.concatenate([d0, fecgen.golay24_encode(d0)])
    pats = [()] + [c for k in (1, 2, 3) for c in itertools.combinations(range(24), k)]
    pats += [c for c in itertools.combinations(range(24), 4)][::7]
    d = np.zeros((len(pats), 12), np.uint8)
    p = np.zeros((len(pats), 12), np.uint8)
    for i, c in enumerate(pats):
        w = w0.copy()
        w[list(c)] ^= 1
        d[i], p[i] = w[:12], w[12:]
    a = oracle_golay(d, p)
    b = ref_golay(d, p)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@needs_ref
@pytest.mark.parametrize("code", list(fecgen.P25_RS_CODES))
def test_rs_matches_reference(built, code):
    rng = np.random.default_rng(FZ + hash(code) & 0xFFFF)
    d, p = fecgen.gen_p25_rs(rng, code, 3000, max_extra=4)
    d[:200] = rng.integers(0, 2, d[:200].shape)          # pure noise
    p[:200] = rng.integers(0, 2, p[:200].shape)
    a = oracle_rs(code, d, p)
    b = ref_rs(code, d, p)
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], b[0])
    assert 0 < a[1].sum() < len(a[1])


def oracle_hamming_soft(bits, rel):
    o = orc.oracle()
    o.orc_hamming_10_6_3_soft.argtypes = [VP, VP, VP]
    out = np.zeros_like(bits)
    rc = np.zeros(len(bits), np.int32)
    for i in range(len(bits)):
        rc[i] = o.orc_hamming_10_6_3_soft(bits[i].ctypes.data, rel[i].ctypes.data, out[i].ctypes.data)
    return out, rc


def oracle_golay_soft(data, par, rel):
    o = orc.oracle()
    o.orc_golay_24_soft.argtypes = [VP, C.c_int, VP, VP, VP]
    out = data.copy()
    rc = np.zeros(len(data), np.int32)
    fx = np.zeros(len(data), np.int32)
    f = C.c_int(0)
    for i in range(len(data)):
        rc[i] = o.orc_golay_24_soft(out[i].ctypes.data, data.shape[1], par[i].ctypes.data, rel[i].ctypes.data, C.byref(f))
        fx[i] = f.value
    return out, rc, fx


def gen_soft_reliab(rng, bits_shape, flipped_mask):
    """Plausible reliabilities: flipped bits tend to be weak, with some confidently-wrong ones and out-of-range values."""
    rel = rng.integers(60, 256, bits_shape).astype(np.int32)
    weak = rng.integers(0, 90, bits_shape)
    rel = np.where(flipped_mask & (rng.random(bits_shape) < 0.8), weak, rel)
    rel[rng.random(bits_shape) < 0.02] = 400          # clamps to 255
    rel[rng.random(bits_shape) < 0.02] = -7           # clamps to 0
    return np.ascontiguousarray(rel, np.int32)


@needs_ref
def test_hamming_soft_matches_reference(built):
    r = orc.ref()
    r.hamming_10_6_3_soft.argtypes = [VP, VP, VP]
    rng = np.random.default_rng(FZ + 31)
    n = 6000
    d = rng.integers(0, 2, (n, 6)).astype(np.uint8)
    p = np.stack([d[:, 0] ^ d[:, 1] ^ d[:, 2] ^ d[:, 5], d[:, 0] ^ d[:, 1] ^ d[:, 3] ^ d[:, 5],
                  d[:, 0] ^ d[:, 2] ^ d[:, 3] ^ d[:, 4], d[:, 1] ^ d[:, 2] ^ d[:, 3] ^ d[:, 4]], axis=1)
    bits = np.concatenate([d, p], axis=1).astype(np.uint8)
    flips = rng.random(bits.shape) < rng.choice([0.0, 0.08, 0.2], (n, 1))
    bits ^= flips.astype(np.uint8)
    rel = gen_soft_reliab(rng, bits.shape, flips)
    bits[10, 3] = 2                                      # invalid bit value
    got, rc = oracle_hamming_soft(bits, rel)
    for i in range(n):
        out = np.zeros(10, np.uint8)
        want = r.hamming_10_6_3_soft(bits[i].ctypes.data, rel[i].ctypes.data, out.ctypes.data)
        assert want == rc[i], i
        assert np.array_equal(out, got[i]), i
    assert set(np.unique(rc)) == {0, 1, 2}


@needs_ref
@pytest.mark.parametrize("length", [6, 12])
def test_golay_soft_matches_reference(built, length):
    r = orc.ref()
    fn = r.check_and_fix_golay_24_6_soft if length == 6 else r.check_and_fix_golay_24_12_soft
    fn.argtypes = [VP, VP, VP, VP]
    rng = np.random.default_rng(FZ + 33 + length)
    n = 2500
    d = np.zeros((n, length), np.uint8)
    p = np.zeros((n, 12), np.uint8)
    flips = np.zeros((n, length + 12), bool)
    for i in range(n):
        d12 = np.zeros(12, np.uint8)
        d12[12 - length:] = rng.integers(0, 2, length)
        w = np.concatenate([d12[12 - length:], fecgen.golay24_encode(d12)])
        ne = int(rng.integers(0, 7))
        pos = rng.choice(length + 12, ne, replace=False)
        w[pos] ^= 1
        flips[i, pos] = True
        d[i], p[i] = w[:length], w[length:]
    rel = gen_soft_reliab(rng, flips.shape, flips)
    d[20, 0] = 3                                          # invalid
    got, rc, fx = oracle_golay_soft(d, p, rel)
    f = C.c_int(0)
    for i in range(n):
        x = d[i].copy()
        want = fn(x.ctypes.data, p[i].ctypes.data, rel[i].ctypes.data, C.byref(f))
        assert want == rc[i] and f.value == fx[i], (i, want, rc[i], f.value, fx[i])
        assert np.array_equal(x, got[i]), i
    assert rc.sum() > 0 and (rc == 0).sum() > n // 2


def oracle_rs_soft_rel(code, data, par, drel, prel):
    n_par, n_data, t = fecgen.P25_RS_CODES[code]
    o = orc.oracle()
    o.orc_p25_rs_soft_reliability.argtypes = [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int]
    out = data.copy()
    rc = np.zeros(data.shape[0], np.int32)
    for i in range(data.shape[0]):
        rc[i] = o.orc_p25_rs_soft_reliability(out[i].ctypes.data, par[i].ctypes.data, drel[i].ctypes.data,
                                              prel[i].ctypes.data, n_par, n_data, t)
    return out, rc


def gen_rs_soft(rng, code, n):
    """Codewords with up to 2t bad symbols; bad symbols mostly carry low reliability (so erasures help), some do not."""
    n_par, n_data, t = fecgen.P25_RS_CODES[code]
    data = np.zeros((n, n_data, 6), np.uint8)
    par = np.zeros((n, n_par, 6), np.uint8)
    drel = rng.integers(64, 256, (n, n_data)).astype(np.uint8)
    prel = rng.integers(64, 256, (n, n_par)).astype(np.uint8)
    for i in range(n):
        d = rng.integers(0, 64, n_data)
        w = np.array(list(fecgen.rs63_encode(d, t)) + list(d))
        ne = int(rng.integers(0, 2 * t + 3))
        pos = rng.choice(n_par + n_data, min(ne, n_par + n_data), replace=False)
        w[pos] ^= rng.integers(1, 64, len(pos))
        for q in pos:
            low = rng.random() < 0.8
            v = int(rng.integers(0, 64)) if low else int(rng.integers(64, 256))
            if q < n_par:
                prel[i, q] = v
            else:
                drel[i, q - n_par] = v
        par[i] = fecgen.syms_to_bits6(w[:n_par])
        data[i] = fecgen.syms_to_bits6(w[n_par:])
    return data, par, drel, prel


@needs_ref
@pytest.mark.parametrize("code", list(fecgen.P25_RS_CODES))
def test_rs_soft_reliability_matches_reference(built, code):
    r = orc.ref()
    fn = {"24_12_13": r.p25p1_rs_24_12_13_soft_reliability, "24_16_9": r.p25p1_rs_24_16_9_soft_reliability,
          "36_20_17": r.p25p1_rs_36_20_17_soft_reliability}[code]
    fn.argtypes = [VP, VP, VP, VP]
    rng = np.random.default_rng(FZ + 200 + len(code))
    d, p, drel, prel = gen_rs_soft(rng, code, 1500)
    got, rc = oracle_rs_soft_rel(code, d, p, drel, prel)
    for i in range(len(d)):
        x = d[i].copy()
        want = fn(x.ctypes.data, p[i].ctypes.data, drel[i].ctypes.data, prel[i].ctypes.data)
        assert want == rc[i], (i, want, rc[i])
        assert np.array_equal(x, got[i]), i
    assert 0 < rc.sum() < len(rc)


@needs_ref
def test_rs_explicit_erasures_match_reference(built):
    """check_and_fix_*_soft with caller-given erasure lists, incl. duplicates / out-of-range / too many."""
    r, o = orc.ref(), orc.oracle()
    o.orc_p25_rs_decode_soft.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP, C.c_int]
    rng = np.random.default_rng(FZ + 300)
    fns = {"24_12_13": r.check_and_fix_reedsolomon_24_12_13_soft, "24_16_9": r.check_and_fix_reedsolomon_24_16_9_soft,
           "36_20_17": r.check_and_fix_redsolomon_36_20_17_soft}
    for code, fn in fns.items():
        fn.argtypes = [VP, VP, VP, C.c_int]
        n_par, n_data, t = fecgen.P25_RS_CODES[code]
        d, p = fecgen.gen_p25_rs(rng, code, 800, max_extra=t)
        for i in range(len(d)):
            ne = int(rng.integers(0, 2 * t + 2))
            er = rng.integers(0, n_par + n_data + (3 if i % 17 == 0 else 0), ne).astype(np.int32)   # may repeat / overflow
            a, b = d[i].copy(), d[i].copy()
            want = fn(a.ctypes.data, p[i].ctypes.data, er.ctypes.data if ne else None, ne)
            got = o.orc_p25_rs_decode_soft(b.ctypes.data, p[i].ctypes.data, n_par, n_data, t, er.ctypes.data if ne else None, ne)
            assert want == got, (code, i, want, got, ne)
            assert np.array_equal(a, b), (code, i)
