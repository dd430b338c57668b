"""Case builders and oracle wrappers for BPTC(128,77) and the reverse-channel BPTC 16 x 2 (src/fec/bptc.c:167-336)."""
import ctypes as C

import numpy as np

import orc

H16114 = [0x0009AF, 0x00135E, 0x0026BC, 0x0044D7, 0x008765]   # parity-check rows, bit j = position j (ddn_tables_fec3.h)
RC_PERM = [0, 24, 1, 25, 2, 26, 3, 27, 4, 28, 5, 29, 6, 30, 7, 31, 8, 16, 9, 17, 10, 18, 11, 19, 12, 20, 13, 21, 14, 22, 15, 23]


def hamming_row(data11):
    """the 16-bit Hamming(16,11,4) code word (bit per byte) with these eleven data bits"""
    w = sum(int(b) << j for j, b in enumerate(data11))
    for p in range(32):
        c = w | p << 11
        if all(bin(c & h).count("1") % 2 == 0 for h in H16114):
            return np.array([(c >> j) & 1 for j in range(16)], np.uint8)
    raise AssertionError


def matrix128(rng, row_errs, col_errs=0):
    """a valid 8 x 16 matrix, then row_errs[r] flipped bits in row r (0..6) and col_errs flipped bits in the parity row"""
    m = np.zeros((8, 16), np.uint8)
    for r in range(7):
        m[r] = hamming_row(rng.integers(0, 2, 11))
    m[7] = m[:7].sum(0) & 1
    for r, e in enumerate(row_errs):
        m[r, rng.choice(16, e, replace=False)] ^= 1
    m[7, rng.choice(16, col_errs, replace=False)] ^= 1
    return m


def word32(rng, n_err, odd):
    row = hamming_row(rng.integers(0, 2, 11))
    m = np.concatenate([row, row ^ 1 if odd else row]).astype(np.uint8)
    m[rng.choice(32, n_err, replace=False)] ^= 1
    x = np.zeros(32, np.uint8)
    x[:] = m[RC_PERM]                 # matrix[perm[i]] = input[i]
    return x


def oracle_128x77(m):
    o = orc.oracle()
    o.orc_bptc_128x77.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_bptc_128x77.restype = C.c_uint32
    out = np.zeros(77, np.uint8)
    f = C.c_int(0)
    m = np.ascontiguousarray(m, np.uint8)
    rc = o.orc_bptc_128x77(m.ctypes.data, out.ctypes.data, C.addressof(f))
    return rc, out, f.value


def oracle_16x2(x, odd):
    o = orc.oracle()
    o.orc_bptc_16x2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    o.orc_bptc_16x2.restype = C.c_uint32
    out = np.zeros(32, np.uint8)
    f = C.c_int(0)
    rc = o.orc_bptc_16x2(np.ascontiguousarray(x, np.uint8).ctypes.data, out.ctypes.data, odd, C.addressof(f))
    return rc, out, f.value


def cases128(rng, n):
    out = []
    for i in range(n):
        kind = i % 5
        if kind == 0:
            out.append(matrix128(rng, [0] * 7))
        elif kind == 1:
            out.append(matrix128(rng, rng.integers(0, 2, 7)))
        elif kind == 2:
            out.append(matrix128(rng, rng.integers(0, 4, 7), int(rng.integers(0, 3))))
        elif kind == 3:
            out.append(matrix128(rng, [0, 2, 0, 1, 2, 0, 0], 1))       # failing rows after good ones: the stale-line rule
        else:
            out.append(rng.integers(0, 2, (8, 16)).astype(np.uint8) | (rng.integers(0, 2, (8, 16)).astype(np.uint8) << 1))
    return out
