"""CPU: the P25 Phase 2 FACCH / SACCH burst decode restatement (oracle/ddn_oracle_rs.c: orc_p25p2_xcch - burst gather, RS(63,35)
with the fixed erasures, the ranked soft-erasure retries) against the reference's own pieces compiled in place (oracle/_ref:
ez_rs28_facch / _sacch of src/fec/ez.cpp and p25p2_facch_soft_erasures / _sacch_soft_erasures of
src/protocol/p25/phase2/p25p2_soft.c behind the gather / retry loops of p25p2_frame.c:408-495,652-671)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import rs28

FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
N_PL = {0: 156, 1: 180}


def oracle_xcch(kind, bits, llr, threshold=64):
    o = orc.oracle()
    o.orc_p25p2_xcch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    out, used = np.zeros(180, np.uint8), C.c_int(0)
    b, l = np.ascontiguousarray(bits, np.uint8), np.ascontiguousarray(llr, np.int16)
    ec = o.orc_p25p2_xcch(kind, b.ctypes.data, l.ctypes.data, threshold, out.ctypes.data, C.byref(used))
    return ec, out[:N_PL[kind]].copy(), used.value


def ref_xcch(kind, bits, llr, ts):
    r = C.CDLL(orc.REF_SO)
    r.refh_p25p2_xcch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    out, used = np.zeros(180, np.uint8), C.c_int(0)
    b, l = np.ascontiguousarray(bits, np.uint8), np.ascontiguousarray(llr, np.int16)
    ec = r.refh_p25p2_xcch(kind, b.ctypes.data, l.ctypes.data, ts, out.ctypes.data, C.byref(used))
    return ec, out[:N_PL[kind]].copy(), used.value


def cases(rng, n):
    out = []
    for i in range(n):
        kind = i & 1
        n_err = int(rng.integers(0, 14))
        weak = int(rng.integers(0, n_err + 1))
        out.append((kind,) + rs28.make_xcch_burst(rng, kind, n_err, weak, int(rng.integers(0, 6))) + (n_err, weak))
    return out


def test_clean_and_repairable_bursts_decode_to_what_was_sent():
    rng = np.random.default_rng(41 + FZ)
    seen_dynamic = 0
    for kind in (0, 1):
        for n_err in (0, 3, 5 if kind == 0 else 8):      # (28 parity symbols less 18 / 11 fixed erasures: 5 / 8 errors)
            bits, llr, sent = rs28.make_xcch_burst(rng, kind, n_err, 0)
            ec, pl, used = oracle_xcch(kind, bits, llr)
            assert ec >= 0 and used == 0 and np.array_equal(pl, sent), (kind, n_err, ec)
        for n_err in ((7, 8) if kind == 0 else (10, 11)):                      # beyond the fixed erasures' reach: the weak symbols have to be found
            bits, llr, sent = rs28.make_xcch_burst(rng, kind, n_err, n_err)
            ec, pl, used = oracle_xcch(kind, bits, llr)
            assert ec >= 0 and used == 1 and np.array_equal(pl, sent), (kind, n_err, ec)
            seen_dynamic += used
        bits, llr, sent = rs28.make_xcch_burst(rng, kind, 15, 0)
        ec, pl, used = oracle_xcch(kind, bits, llr)
        assert ec < 0 and np.array_equal(pl, bits[rs28.XCCH_PAYLOAD_POS[kind]])      # hopeless: as received
    assert seen_dynamic == 4


@needs_ref
def test_restatement_equals_the_compiled_reference_pieces():
    rng = np.random.default_rng(43 + FZ)
    kinds = set()
    for k, (kind, bits, llr, sent, n_err, weak) in enumerate(cases(rng, 300)):
        want = ref_xcch(kind, bits, llr, k % 4)
        got = oracle_xcch(kind, bits, llr)
        assert got[0] == want[0] and got[2] == want[2] and np.array_equal(got[1], want[1]), (k, kind, n_err, weak, got[0], want[0])
        kinds.add((kind, want[0] >= 0, want[2]))
    assert len(kinds) >= 6, kinds


def oracle_duid(received, rel, threshold=64):
    o = orc.oracle()
    o.orc_p25p2_duid_lookup_soft.argtypes = [C.c_int, C.c_void_p, C.c_int]
    r = None if rel is None else np.ascontiguousarray(rel, np.uint8)
    return o.orc_p25p2_duid_lookup_soft(int(received), None if r is None else r.ctypes.data, threshold)


def test_duid_soft_lookup_known_answers_of_the_reference():
    """tests/protocol/p25/test_p25_p2_reliability.c:970-974 and :1372-1440 - every answer the reference's suite holds for
    p25p2_duid_lookup_soft (threshold 64, the default)"""
    s = [200] * 8
    assert oracle_duid(0x17, [0] * 8) == 1 and oracle_duid(0x17, None) == 1
    assert oracle_duid(0x07, s[:7] + [5]) == 1                       # a valid hard decision is kept
    assert oracle_duid(0x03, s[:6] + [5, 5]) == 0                    # the weakest invalid bits are flipped
    assert oracle_duid(0x03, s) == -1                                # confident invalid bits are not
    assert oracle_duid(0x80, [5] + s[1:]) == 0 and oracle_duid(0x80, s) == -1 and oracle_duid(0x80, [5, 5] + s[2:]) == -1
    r = list(s)
    for k in (3, 5, 6, 7):
        r[k] = 5
    assert oracle_duid(0x03, r) == -1                                # tied candidates: the hard answer stands


@pytest.mark.skipif(not os.path.exists("/root/reference/src/protocol/p25/phase2/p25p2_frame.c"), reason="reference tree not present")
def test_duid_hard_table_equals_the_source_table():
    """the rule the restatement (and the kernel) derive the 256-entry table from, against the table in the reference's source"""
    import re
    src = open("/root/reference/src/protocol/p25/phase2/p25p2_frame.c").read()
    body = src[src.index("duid_lookup[256] = {"):]
    body = body[body.index("{") + 1:body.index("};")]
    vals = [int(v) for v in re.findall(r"-?\d+", re.sub(r"//[^\n]*", "", body))]
    assert len(vals) == 256
    o = orc.oracle()
    assert [o.orc_p25p2_duid_hard(r) for r in range(256)] == vals


def scramble_bits(wacn, sysid, nac, count):
    """p25p2_generate_scramble_bits (p25p2_scramble.c:12-26) in Python"""
    s = (wacn * 16777216 + sysid * 4096 + nac) & ((1 << 64) - 1)
    out = np.zeros(count, np.uint8)
    for i in range(count):
        out[i] = (s >> 43) & 1
        b = ((s >> 33) ^ (s >> 19) ^ (s >> 14) ^ (s >> 8) ^ (s >> 3) ^ (s >> 43)) & 1
        s = ((s << 1) | b) & ((1 << 64) - 1)
    return out


@needs_ref
def test_scramble_sequence_equals_the_reference():
    r = C.CDLL(orc.REF_SO)
    r.p25p2_generate_scramble_bits.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(59 + FZ)
    for _ in range(6):
        w, s, n = int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 12)), int(rng.integers(1, 1 << 12))
        out = np.zeros(4320, np.uint8)
        r.p25p2_generate_scramble_bits(w, s, n, out.ctypes.data, 4320)
        assert np.array_equal(out, scramble_bits(w, s, n, 4320))


def oracle_ess(pl, pll, pa, pal, threshold=64):
    o = orc.oracle()
    o.orc_p25p2_ess.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
    out, ec = np.zeros(96, np.uint8), C.c_int(0)
    a, b, c, d = (np.ascontiguousarray(x) for x in (pl, pll, pa, pal))
    acc = o.orc_p25p2_ess(a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, threshold, out.ctypes.data, C.byref(ec))
    return acc, ec.value, out


@needs_ref
def test_ess_restatement_equals_the_compiled_reference_pieces():
    r = C.CDLL(orc.REF_SO)
    r.refh_p25p2_ess.argtypes = [C.c_void_p] * 6
    rng = np.random.default_rng(67 + FZ)
    classes = set()
    for k in range(300):
        n_err = int(rng.integers(0, 24))
        pl, pll, pa, pal, sent = rs28.make_ess_case(rng, n_err, int(rng.integers(0, n_err + 1)), int(rng.integers(0, 6)))
        out, ec = np.zeros(96, np.uint8), C.c_int(0)
        acc = r.refh_p25p2_ess(pl.ctypes.data, pll.ctypes.data, pa.ctypes.data, pal.ctypes.data, out.ctypes.data, C.byref(ec))
        got = oracle_ess(pl, pll, pa, pal)
        assert got[0] == acc and got[1] == ec.value and np.array_equal(got[2], out), (k, n_err, got[:2], acc, ec.value)
        classes.add((acc, ec.value >= 15 if acc else False))
        if acc and n_err <= 14:
            assert np.array_equal(out, sent)
    assert len(classes) >= 3, classes


def test_p25p2_voice_schedule_is_the_measured_ambe_dibit_map():
    """p25p2_frame.c:250-262 (c0..c3 / csubset: where bit x of a 4V / 2V frame goes) against the AMBE 3600x2450 dibit schedule measured
    from the compiled reference (include/dsd-neo/core/ambe_interleave.h through tools/gen_tables_ambe.py): the same, read bit by bit"""
    import rx4
    c = [[23, 5, 22, 4, 21, 3, 20, 2, 19, 1, 18, 0, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6],
         [10, 9, 8, 7, 6, 5, 22, 4, 21, 3, 20, 2, 19, 1, 18, 0, 17, 16, 15, 14, 13, 12, 11],
         [3, 2, 1, 0, 10, 9, 8, 7, 6, 5, 4], [13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0]]
    cs = [0, 0, 1, 2] * 4 + [0, 0, 1, 3] * 2 + [0, 1, 1, 3] * 5 + [0, 1, 2, 3] * 7
    its = [iter(x) for x in c]
    sched = [(w, next(its[w])) for w in cs]
    m = np.asarray(rx4.ambe2450_map())
    assert len(sched) == 72 and all(sched[2 * i] == (m[i][0], m[i][1]) and sched[2 * i + 1] == (m[i][2], m[i][3]) for i in range(36))


def crc12_ok(bits, length):
    """crc12_xb_bridge(payload, length) == 0 (src/protocol/p25/p25_crc.c:78-147)"""
    reg = 0
    for k in range(length + 12):
        reg = (reg << 1) | (int(bits[k]) & 1 if k < length else 0)
        if reg & 0x1000:
            reg ^= 0x1897
    got = 0
    for k in range(12):
        got = (got << 1) | (int(bits[length + k]) & 1)
    return int(((reg ^ 0xFFF) & 0xFFF) == got)


def crc16_ok(bits, length=164):
    c = 0
    for k in range(length):
        c = (((c << 1) ^ 0x1021) if (((c >> 15) & 1) ^ (int(bits[k]) & 1)) else (c << 1)) & 0xFFFF
    c ^= 0xFFFF
    got = 0
    for k in range(16):
        got = (got << 1) | (int(bits[length + k]) & 1)
    return int(c == got)


def with_crc12(rng, n_pl):
    """random MAC PDU bits with a good CRC12 in the last 12"""
    b = rng.integers(0, 2, n_pl).astype(np.uint8)
    reg = 0
    for k in range(n_pl):
        reg = (reg << 1) | (int(b[k]) if k < n_pl - 12 else 0)
        if reg & 0x1000:
            reg ^= 0x1897
    v = (reg ^ 0xFFF) & 0xFFF
    b[n_pl - 12:] = [(v >> (11 - k)) & 1 for k in range(12)]
    assert crc12_ok(b, n_pl - 12)
    return b


@needs_ref
def test_mac_crcs_equal_the_reference():
    r = C.CDLL(orc.REF_SO)
    r.crc12_xb_bridge.argtypes = [C.c_void_p, C.c_int]
    r.crc16_lb_bridge.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(79 + FZ)
    for k in range(60):
        n_pl = 156 if k & 1 else 180
        b = with_crc12(rng, n_pl) if k % 3 else rng.integers(0, 2, n_pl).astype(np.uint8)
        if k % 7 == 3:
            b[int(rng.integers(0, n_pl))] ^= 1
        ib = b.astype(np.int32)
        assert crc12_ok(b, n_pl - 12) == int(r.crc12_xb_bridge(ib.ctypes.data, n_pl - 12) == 0), k
        if n_pl == 180:
            assert crc16_ok(b) == int(r.crc16_lb_bridge(ib.ctypes.data, 164) == 0), k
