"""Helpers for the DMR / NXDN block-code tests: oracle and compiled-reference wrappers, test inputs."""
import ctypes as C
import itertools

import numpy as np

import orc

CODES = {0: (7, 4, "Hamming_7_4"), 1: (12, 8, "Hamming_12_8"), 2: (13, 9, "Hamming_13_9"), 3: (15, 11, "Hamming_15_11"),
         4: (16, 11, "Hamming_16_11_4"), 5: (20, 8, "Golay_20_8"), 6: (24, 12, "Golay_24_12"), 7: (16, 7, "QR_16_7_6")}

# tests/fec/test_fec_bptc_rs.c:19-26 (reference-held BPTC(196,96) code word; payload bit i = ((17 i + i / 5) & 1), R = 1 0 1)
BPTC_KAT = [0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0,
            1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0,
            1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0,
            1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0,
            1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 1, 0, 1, 1,
            0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 0]


def words_for(code, rng, n_random=4000):
    """u8 [m][n]: every word for n <= 16; all patterns of weight <= 4 on two code words + random words otherwise"""
    n = CODES[code][0]
    if n <= 16:
        v = np.arange(1 << n, dtype=np.uint32)
        return ((v[:, None] >> np.arange(n)[None, :]) & 1).astype(np.uint8)
    rows = []
    base = [np.zeros(n, np.uint8)]
    for w in range(0, 5):
        for pos in itertools.combinations(range(n), w):
            e = np.zeros(n, np.uint8)
            e[list(pos)] = 1
            rows.append(e)
    rows = np.array(rows)
    rnd = rng.integers(0, 2, size=(n_random, n), dtype=np.uint8)
    return np.concatenate([rows, rnd])


def oracle_decode(code, words, nb=1):
    o = orc.oracle()
    n, k, _ = CODES[code]
    w = np.ascontiguousarray(words, np.uint8).copy()
    items = w.reshape(-1, nb * n)
    dec = np.zeros((items.shape[0], nb * k), np.uint8)
    ok = np.zeros(items.shape[0], np.uint8)
    for i in range(items.shape[0]):
        p = C.c_void_p(items[i].ctypes.data)
        if code == 0:
            ok[i] = o.orc_hamming_7_4_decode(p)
        elif code <= 4:
            ok[i] = o.orc_hamming_multi_decode(code - 1, p, C.c_void_p(dec[i].ctypes.data), nb)
        elif code in (5, 6):
            ok[i] = o.orc_golay_dmr_decode(n, p)
        else:
            ok[i] = o.orc_qr_16_7_6_decode(p)
    return items, dec, ok


def ref_decode(code, words, nb=1):
    r = orc.ref()
    r.InitAllFecFunction()
    n, k, name = CODES[code]
    fn = getattr(r, name + "_decode")
    fn.restype = C.c_bool
    w = np.ascontiguousarray(words, np.uint8).copy()
    items = w.reshape(-1, nb * n)
    dec = np.zeros((items.shape[0], nb * k), np.uint8)
    ok = np.zeros(items.shape[0], np.uint8)
    for i in range(items.shape[0]):
        p = C.c_void_p(items[i].ctypes.data)
        if 1 <= code <= 4:
            ok[i] = fn(p, C.c_void_p(dec[i].ctypes.data), nb)
        else:
            ok[i] = fn(p)
    return items, dec, ok


def bptc_inputs(rng, n):
    """the reference-held code word with 0..6 random flips, interleaved copies too"""
    base = np.array(BPTC_KAT, np.uint8)
    out = np.tile(base, (n, 1))
    for i in range(n):
        k = int(rng.integers(0, 7)) if i % 3 else int(rng.integers(0, 30))
        out[i, rng.choice(196, k, replace=False)] ^= 1
    out[0] = base
    return out


def oracle_bptc(x, deinterleave):
    o = orc.oracle()
    o.orc_bptc_196x96.restype = C.c_uint32
    n = x.shape[0]
    out, r3, errs = np.zeros((n, 96), np.uint8), np.zeros((n, 3), np.uint8), np.zeros(n, np.uint32)
    ub = np.zeros(n, bool)
    for i in range(n):
        xi = np.ascontiguousarray(x[i])
        errs[i] = o.orc_bptc_196x96(C.c_void_p(xi.ctypes.data), deinterleave, C.c_void_p(out[i].ctypes.data), C.c_void_p(r3[i].ctypes.data))
        ub[i] = bool(o.orc_bptc_last_col0_failed())
    oracle_bptc.undefined = ub          # items on which the reference reads an uninitialised array (first column uncorrectable)
    return out, r3, errs


