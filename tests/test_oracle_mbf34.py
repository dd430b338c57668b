"""CPU: the P25 Phase 1 confirmed-data (rate 3/4, LLR) decoder restatement (oracle/ddn_oracle_fec.c: orc_p25_mbf34_list / _best)
against the reference's own p25p1_mbf34.c compiled in place (oracle/_ref): clean blocks, blocks with a few weak or flipped dibits,
noise - candidates (bytes, metric, order, count) and the plain best path."""
import ctypes as C
import os

import numpy as np
import pytest

import orc

FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def tables():
    o = orc.oracle()
    for f in ("orc_tbl_r34_point_to_nibble", "orc_tbl_r34_fsm"):
        getattr(o, f).restype = C.POINTER(C.c_uint8)
    p2n = np.array(o.orc_tbl_r34_point_to_nibble()[:16])
    fsm = np.array(o.orc_tbl_r34_fsm()[:64])
    il = np.zeros(98, np.uint8)
    o.orc_trellis_interleave_98.argtypes = [C.c_void_p]
    o.orc_trellis_interleave_98(il.ctypes.data)
    return p2n, fsm, il


def encode34(bytes18):
    """18 bytes -> 98 dibits on the air (48 tribits + a flushing zero state through the rate 3/4 FSM, interleaved)"""
    p2n, fsm, il = tables()
    bits = np.unpackbits(np.asarray(bytes18, np.uint8))
    tri = [int(bits[3 * k] << 2 | bits[3 * k + 1] << 1 | bits[3 * k + 2]) for k in range(48)] + [0]
    st, dib = 0, []
    for t in tri:
        nib = int(p2n[fsm[st * 8 + t] & 15])
        dib += [nib >> 2, nib & 3]
        st = t
    out = np.zeros(98, np.uint8)
    out[:] = np.array(dib, np.uint8)[il]            # received dibit i carries de-interleaved dibit il[i]
    return out


def llr_of(dibits, rng, strong=200, weak_at=(), flip_at=(), noise=0):
    llr = np.zeros(196, np.int16)
    for i, d in enumerate(dibits):
        for b in range(2):
            bit = (int(d) >> (1 - b)) & 1
            mag = strong
            if i in weak_at:
                mag = int(rng.integers(1, 12))
            v = mag if bit else -mag
            if i in flip_at:
                v = -v
            llr[2 * i + b] = v + (int(rng.integers(-noise, noise + 1)) if noise else 0)
    return llr


def oracle_list(llr, mx=8):
    o = orc.oracle()
    o.orc_p25_mbf34_list.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    by, me = np.zeros((8, 18), np.uint8), np.zeros(8, np.uint32)
    n = o.orc_p25_mbf34_list(np.ascontiguousarray(llr, np.int16).ctypes.data, mx, by.ctypes.data, me.ctypes.data)
    return n, by[:n].copy(), me[:n].copy()


def oracle_best(llr):
    o = orc.oracle()
    o.orc_p25_mbf34_best.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros(18, np.uint8)
    m = o.orc_p25_mbf34_best(np.ascontiguousarray(llr, np.int16).ctypes.data, out.ctypes.data)
    return m, out


def cases(rng, n):
    out = []
    for k in range(n):
        data = rng.integers(0, 256, 18).astype(np.uint8)
        dib = encode34(data)
        kind = k % 5
        if kind == 0:
            llr = llr_of(dib, rng)
        elif kind == 1:
            llr = llr_of(dib, rng, weak_at=set(rng.choice(98, int(rng.integers(1, 12)), replace=False).tolist()))
        elif kind == 2:
            llr = llr_of(dib, rng, strong=60, flip_at=set(rng.choice(98, int(rng.integers(1, 6)), replace=False).tolist()), noise=20)
        elif kind == 3:
            llr = llr_of(dib, rng, strong=int(rng.integers(1, 4)), noise=3)        # many equal metrics: the tie order matters
        else:
            llr = rng.integers(-300, 301, 196).astype(np.int16)
        out.append((data, llr))
    return out


def test_clean_blocks_decode_to_what_was_sent():
    rng = np.random.default_rng(3 + FZ)
    for data, llr in cases(rng, 10)[::5]:
        n, by, me = oracle_list(llr)
        assert n >= 1 and np.array_equal(by[0], data) and me[0] == 0
        m, out = oracle_best(llr)
        assert m == 0 and np.array_equal(out, data)


@needs_ref
def test_restatement_equals_the_compiled_reference():
    r = C.CDLL(orc.REF_SO)
    r.p25_mbf34_decode_soft_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    r.p25_mbf34_decode_soft.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(17 + FZ)
    dummy = np.zeros(98, np.uint8)
    seen = set()
    for data, llr in cases(rng, 300):
        for mx in (8, 3):
            cand = np.zeros((8, 24), np.uint8)                 # {u8 bytes[18], pad, u32 metric}
            n_ref = r.p25_mbf34_decode_soft_list(dummy.ctypes.data, np.ascontiguousarray(llr).ctypes.data, cand.ctypes.data, mx)
            n, by, me = oracle_list(llr, mx)
            assert n == n_ref, (n, n_ref)
            assert np.array_equal(by, cand[:n, :18]) and np.array_equal(me, cand[:n, 20:24].copy().view(np.uint32).reshape(-1))
            seen.add(n)
        out = np.zeros(18, np.uint8)
        m_ref = r.p25_mbf34_decode_soft(dummy.ctypes.data, np.ascontiguousarray(llr).ctypes.data, out.ctypes.data)
        m, ob = oracle_best(llr)
        assert m == m_ref and np.array_equal(ob, out)
    assert 8 in seen and 3 in seen


def test_confirmed_block_generator_round_trip():
    """tests/p25gen.py's confirmed-data block (DBSN | CRC9 | 16 bytes, rate 3/4) decodes to itself with a matching CRC9 - the CRC9
    against the compiled reference's ComputeCrc9Bit when it is present"""
    import p25gen
    rng = np.random.default_rng(77 + FZ)
    for k in range(20):
        blk = p25gen.confirmed_block(k, rng.integers(0, 256, 16))
        n, by, me = oracle_list(llr_of(p25gen.encode_three_quarter_rate(blk), rng))
        assert n >= 1 and np.array_equal(by[0], blk) and me[0] == 0
        bits = [(int(blk[0]) >> (7 - i)) & 1 for i in range(7)] + list(np.unpackbits(blk[2:]))
        assert p25gen.crc9(bits) == (((int(blk[0]) & 1) << 8) | int(blk[1]))
    assert np.array_equal(p25gen.encode_three_quarter_rate(blk), encode34(blk))


@needs_ref
def test_crc9_equals_the_reference():
    """the generator's / kernel's CRC9 (polynomial 0x059, inverted) against ComputeCrc9Bit compiled from src/protocol/dmr/dmr_utils.c"""
    import p25gen
    r = orc.ref()
    r.ComputeCrc9Bit.restype = C.c_uint16
    r.ComputeCrc9Bit.argtypes = [C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(5 + FZ)
    for n in (135, 135, 96, 176, 7, 1):
        bits = rng.integers(0, 2, n).astype(np.uint8)
        assert p25gen.crc9(bits) == r.ComputeCrc9Bit(bits.ctypes.data, n)
