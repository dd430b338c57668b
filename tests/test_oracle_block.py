"""CPU: block-code restatement (BCH(63,16,11), P25p1 NID incl. Chase search, Hamming(10,6,3)) vs golden vectors
produced by the reference's compiled decoders."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps

import fecgen
import orc
from conftest import golden

VP = C.c_void_p


def oracle_nid(bits, rel, obs, par, prel, thr=64):
    o = orc.oracle()
    o.orc_p25p1_nid_decode.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]
    n = bits.shape[0]
    out = np.zeros((n, 4), np.int32)
    for i in range(n):
        o.orc_p25p1_nid_decode(bits[i].ctypes.data, rel[i].ctypes.data if rel is not None else None, int(obs[i]),
                               int(par[i]), int(prel[i]), thr, out[i].ctypes.data)
    return out


def test_bch_generator_and_clean_codewords(built):
    g = fecgen.bch_63_16_generator()
    assert g.bit_length() == 48  # degree 47 = 63 - 16
    o = orc.oracle()
    o.orc_bch_63_16_decode.argtypes = [VP, VP, VP]
    rng = np.random.default_rng(FZ + 1)
    for _ in range(50):
        data = rng.integers(0, 2, 16).astype(np.uint8)
        cw = fecgen.bch_63_16_encode(data)
        assert np.array_equal(cw[:16], data)
        for ne in (0, 1, 11):
            x = cw.copy()
            x[rng.choice(63, ne, replace=False)] ^= 1
            d = np.zeros(16, np.uint8)
            e = C.c_int(0)
            assert o.orc_bch_63_16_decode(x.ctypes.data, d.ctypes.data, C.byref(e)) == 1
            assert e.value == ne and np.array_equal(d, data)


def test_nid_golden(built):
    g = golden("fec_p25p1_nid.npz")
    thr = int(g["threshold"])
    assert np.array_equal(oracle_nid(g["bits"], g["rel"], g["obs"], g["parity"], g["parity_rel"], thr), g["out_soft"])
    assert np.array_equal(oracle_nid(g["bits"], None, g["obs"], g["parity"], g["parity_rel"], thr), g["out_hard"])
    o = orc.oracle()
    o.orc_bch_63_16_decode.argtypes = [VP, VP, VP]
    bits = np.ascontiguousarray(g["bits"])      # keep alive: g[...] materialises a fresh array on every access
    for i in range(bits.shape[0]):
        d = np.zeros(16, np.uint8)
        e = C.c_int(0)
        row = bits[i]
        ok = o.orc_bch_63_16_decode(row.ctypes.data, d.ctypes.data, C.byref(e))
        assert ok == g["bch"][i, 0]
        if ok:
            assert e.value == g["bch"][i, 1] and np.array_equal(d, g["bch"][i, 2:])


def test_hamming_golden(built):
    t = golden("fec_hamming_10_6_3.npz")["table"]
    o = orc.oracle()
    o.orc_hamming_10_6_3.argtypes = [C.c_int, VP]
    for w in range(1024):
        f = C.c_int(0)
        e = o.orc_hamming_10_6_3(w, C.byref(f))
        assert e == t[w, 6]
        want = f.value if e == 1 else (w >> 4)
        assert [(want >> (5 - k)) & 1 for k in range(6)] == t[w, :6].tolist()
