"""CPU: the oracle's trellis / Viterbi restatement reproduces the reference's golden vectors (produced by the
reference's own compiled decoders) and the reference's own known-answer vectors, bit for bit."""
import json
import os

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import fecgen
import orc
from conftest import golden, HERE


def test_interleave_table_shape(built):
    il = fecgen.tables()["il"]
    assert sorted(il.tolist()) == list(range(98))
    assert il[:6].tolist() == [0, 1, 8, 9, 16, 17] and il[26:30].tolist() == [2, 3, 10, 11]


def test_p25_half_rate_golden(built):
    g = golden("fec_p25_half_rate.npz")
    out, met = fecgen.oracle_p25_half_rate(g["llr"])
    assert np.array_equal(out, g["out"]) and np.array_equal(met, g["metric"])


def test_p25_half_rate_recovers_clean_codewords(built):
    rng = np.random.default_rng(FZ + 3)
    llr, st = fecgen.gen_p25_half_rate(rng, 64, sigma=0.0, random_frac=0.0)
    out, met = fecgen.oracle_p25_half_rate(llr)
    d = st[:, :48]
    want = ((d[:, 0::4] << 6) | (d[:, 1::4] << 4) | (d[:, 2::4] << 2) | d[:, 3::4]).astype(np.uint8)
    assert np.array_equal(out, want) and (met == 0).all()


def test_r34_golden_and_reference_kats(built):
    g = golden("fec_r34.npz")
    assert np.array_equal(fecgen.oracle_r34(g["dibits"]), g["out_hard"])
    assert np.array_equal(fecgen.oracle_r34(g["dibits"], g["reliab"]), g["out_soft"])
    # the reference's own known-answer vectors (tests/protocol/dmr/dmr_r34_reference_vectors.h)
    kat = json.load(open(os.path.join(HERE, "golden", "kat_r34_reference_vectors.json")))
    d = np.array([k["dibits"] for k in kat], np.uint8)
    p = np.array([k["payload"] for k in kat], np.uint8)
    assert np.array_equal(fecgen.oracle_r34(d), p)


def test_nxdn_conv_golden(built):
    g = golden("fec_nxdn_conv.npz")
    for name in ("facch", "sacch", "udch", "long"):
        steps, nbits, soft = [int(x) for x in g[name + "_cfg"]]
        out, _ = fecgen.oracle_nxdn(g[name + "_sym"], g[name + "_rel"] if soft else None, steps, nbits)
        assert np.array_equal(out, g[name + "_out"]), name


def test_nxdn_conv_decodes_clean_codeword_and_carries_metrics(built):
    rng = np.random.default_rng(FZ + 4)
    steps = 96
    data = rng.integers(0, 2, (8, steps)).astype(np.int64)
    data[:, -4:] = 0
    sym = (2 * fecgen.conv_k5_encode(data)).astype(np.uint8)
    out, m = fecgen.oracle_nxdn(sym, None, steps, steps - 4)
    bits = np.unpackbits(out, axis=1)[:, :steps - 4]
    assert np.array_equal(bits, data[:, :steps - 4].astype(np.uint8))
    # metrics persist between decodes in the reference (file-static): they come back non-zero and can be fed in
    out2, m2 = fecgen.oracle_nxdn(sym, None, steps, steps - 4, metrics=m)
    assert np.array_equal(out2, out) and m.any() and m2.shape == m.shape


def test_viterbi_k5_golden(built):
    g = golden("fec_viterbi_k5.npz")
    for name in ("lsf", "ysf", "stream"):
        punct = g[name + "_punct"]
        out, cost, _ = fecgen.oracle_m17(g[name + "_soft"], punct if punct.size else None)
        assert np.array_equal(out, g[name + "_out"]) and np.array_equal(cost, g[name + "_cost"]), name


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_fresh_inputs_against_compiled_reference(built):
    import ctypes as C
    VP = C.c_void_p
    r = orc.ref()
    r.p25_12_soft_llr.argtypes = [VP, VP, VP]
    r.dmr_r34_viterbi_decode_soft.argtypes = [VP, VP, VP]
    rng = np.random.default_rng(FZ + 77)
    llr, _ = fecgen.gen_p25_half_rate(rng, 300, sigma=400.0)
    out, met = fecgen.oracle_p25_half_rate(llr)
    for i in range(300):
        o = np.zeros(12, np.uint8)
        m = r.p25_12_soft_llr(None, llr[i].ctypes.data, o.ctypes.data)
        assert m == met[i] and np.array_equal(o, out[i])
    d, rel, _ = fecgen.gen_r34(rng, 300, p_err=0.08)
    want = fecgen.oracle_r34(d, rel)
    for i in range(300):
        o = np.zeros(18, np.uint8)
        r.dmr_r34_viterbi_decode_soft(d[i].ctypes.data, rel[i].ctypes.data, o.ctypes.data)
        assert np.array_equal(o, want[i])


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_p25_half_rate_list_matches_reference(built):
    """p25_12_soft_llr_list: 8 survivors per state, de-duplicated candidates sorted by metric."""
    import ctypes as C
    o, r = orc.oracle(), orc.ref()

    class Cand(C.Structure):
        _fields_ = [("bytes", C.c_uint8 * 12), ("metric", C.c_uint32)]

    r.p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    o.orc_p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(FZ + 77)
    llr, _ = fecgen.gen_p25_half_rate(rng, 300, sigma=500.0, random_frac=0.3)
    llr[5] = 0                      # all ties
    llr[6] = 32767
    llr[7, ::2] = -32768
    for mx in (8, 3, 1):
        for i in range(llr.shape[0]):
            cand = (Cand * 8)()
            nr = r.p25_12_soft_llr_list(None, llr[i].ctypes.data, C.addressof(cand), mx)
            ob = np.zeros((8, 12), np.uint8)
            om = np.zeros(8, np.uint32)
            no = o.orc_p25_12_soft_llr_list(llr[i].ctypes.data, ob.ctypes.data, om.ctypes.data, mx)
            assert no == nr, (i, mx)
            for k in range(nr):
                assert bytes(cand[k].bytes) == ob[k].tobytes() and cand[k].metric == om[k], (i, mx, k)


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_r34_list_matches_reference(built):
    """dmr_r34_viterbi_decode_list: 32 survivors per state, hard and reliability-weighted."""
    import ctypes as C
    o, r = orc.oracle(), orc.ref()

    class Cand(C.Structure):
        _fields_ = [("metric", C.c_int), ("bytes18", C.c_uint8 * 18)]

    r.dmr_r34_viterbi_decode_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    o.orc_r34_decode_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(FZ + 78)
    d, rel, _ = fecgen.gen_r34(rng, 120, p_err=0.06, random_frac=0.3)
    rel[3] = 0                      # every weighted cost 0: all paths tie
    rel[4] = 255
    for weighted in (0, 1):
        for mx in (32, 5, 1):
            for i in range(d.shape[0]):
                cand = (Cand * 32)()
                cnt = C.c_int(0)
                rp = rel[i].ctypes.data if weighted else None
                assert r.dmr_r34_viterbi_decode_list(d[i].ctypes.data, rp, C.addressof(cand), mx, C.byref(cnt)) == 0
                om = np.zeros(32, np.int32)
                ob = np.zeros((32, 18), np.uint8)
                no = o.orc_r34_decode_list(d[i].ctypes.data, rp, mx, om.ctypes.data, ob.ctypes.data)
                assert no == cnt.value, (i, weighted, mx)
                for k in range(no):
                    assert cand[k].metric == om[k] and bytes(cand[k].bytes18) == ob[k].tobytes(), (i, weighted, mx, k)
