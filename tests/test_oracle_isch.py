"""CPU: the P25 Phase 2 I-ISCH lookup restatement (oracle/ddn_oracle_rs.c, table measured from the compiled reference) against the
reference's compiled isch_lookup / isch_lookup_soft (src/fec/ez.cpp:325-384): exact words, every error weight up to 9, the
7 / 7 ties with the S-ISCH word, soft reliabilities."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def table():
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "ddn_tables_isch.h")).read()
    t = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{10})ULL", txt.split("DDN_ISCH_TABLE_INIT")[1].split("}")[0])]
    assert len(t) == 128
    return t


def words(rng, n):
    """received words: codewords (and the S-ISCH word) with 0..9 flipped bits, tie words, plain noise"""
    t = table()
    S = 0x575D57F7FF
    out = []
    for _ in range(n):
        kind = rng.integers(0, 10)
        base = S if kind == 0 else t[int(rng.integers(0, 128))]
        if kind == 9:
            out.append(int(rng.integers(0, 1 << 40, dtype=np.uint64)))
            continue
        w = base
        for p in rng.choice(40, size=int(rng.integers(0, 10)), replace=False):
            w ^= 1 << int(p)
        out.append(w)
    for c in t:                                   # every 7 / 7 tie with the S-ISCH word
        d = c ^ S
        if bin(d).count("1") == 14:
            bits = [p for p in range(40) if d >> p & 1]
            for rot in range(14):
                w = c
                for p in (bits[rot:] + bits[:rot])[:7]:
                    w ^= 1 << p
                out.append(w)
    return out


def words_high(rng, n):
    """words with stray bits above the 40-bit field (bit 47 always, others at random): the reference does not mask its argument -
    dsd_popcount64 counts those bits toward the distance, isch_weighted_mismatch_cost walks the 40 field bits only
    (src/fec/ez.cpp:335-352) - so neither does the restatement nor the kernel, and no reliability outside the row is read"""
    t = table()
    out = []
    for _ in range(n):
        w = t[int(rng.integers(0, 128))] if rng.integers(0, 8) else 0x575D57F7FF
        for p in rng.choice(40, size=int(rng.integers(0, 7)), replace=False):
            w ^= 1 << int(p)
        w |= 1 << 47
        for p in rng.choice(np.arange(40, 64), size=int(rng.integers(0, 3)), replace=False):
            w |= 1 << int(p)
        out.append(w)
    return out


def oracle_hard(w):
    o = orc.oracle()
    o.orc_isch_lookup.argtypes = [C.c_uint64]
    return o.orc_isch_lookup(w)


def oracle_soft(w, rel):
    o = orc.oracle()
    o.orc_isch_lookup_soft.argtypes = [C.c_uint64, C.c_void_p]
    return o.orc_isch_lookup_soft(w, rel.ctypes.data if rel is not None else None)


@needs_ref
def test_isch_lookup_equals_compiled_reference():
    r = C.CDLL(orc.REF_SO)
    r.isch_lookup.argtypes = [C.c_uint64]
    r.isch_lookup_soft.argtypes = [C.c_uint64, C.c_void_p]
    rng = np.random.default_rng(9 + FZ)
    seen = set()
    for w in words(rng, 6000) + words_high(rng, 1500):
        a, b = r.isch_lookup(w), oracle_hard(w)
        assert a == b, (hex(w), a, b)
        seen.add(a >= 0)
        rel = rng.integers(0, 256, 40).astype(np.uint8) if rng.integers(0, 4) else None
        if rel is not None and rng.integers(0, 3) == 0:
            rel[:] = rng.integers(0, 3, 40)       # many equal costs: the tie-break order matters
        a, b = r.isch_lookup_soft(w, rel.ctypes.data if rel is not None else None), oracle_soft(w, rel)
        assert a == b, (hex(w), a, b)
    assert seen == {True, False}


def test_isch_table_is_the_affine_code_it_should_be():
    t = table()
    assert len(set(t)) == 128 and min(bin(a ^ b).count("1") for i, a in enumerate(t) for b in t[:i]) == 16
    for i in range(128):
        for j in (1, 2, 4, 8, 16, 32, 64):
            assert t[i] ^ t[i ^ j] == t[0] ^ t[j]
        assert oracle_hard(t[i]) == i
