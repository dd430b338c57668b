#!/usr/bin/env python3
"""Generates ddn_tables_isch.h (dsd-neo_amd/csrc/ and oracle/): the 128 I-ISCH codewords of P25 Phase 2's (40,9,16) code and the
S-ISCH word, as the reference's isch_lookup() knows them (src/fec/ez.cpp:283-340).

MEASURED from the compiled reference (oracle/_ref), which only answers "which codeword is within 7 bits of this word":
  1. random 40-bit words are looked up until some are accepted (a quarter of a percent are);
  2. an accepted word is walked to the edge of its codeword's radius-7 ball (flip bits while the answer stays), and at the edge
     every bit whose flip loses the answer agrees with the codeword, every other one differs: that is the codeword;
  3. the I-ISCH words are an affine code in their 7-bit index (checked): a handful of recovered codewords give the 7 basis
     differences and word 0, all 128 follow and are verified one by one as exact matches of isch_lookup();
  4. the S-ISCH word (answer -2 on an exact match, unlike the -2 of "nothing within 7 bits") is found the same way by sampling.
Build container only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

MASK = (1 << 40) - 1


def main():
    r = C.CDLL(orc.REF_SO)
    look = r.isch_lookup
    look.argtypes = [C.c_uint64]
    look.restype = C.c_int
    rng = np.random.default_rng(12345)

    def edge_codeword(x, idx):
        """x is within 7 bits of codeword idx -> that codeword.  Flipping ever more distinct bit positions must leave the ball;
        the word before the first flip that loses the answer lies at distance exactly 7 (minimum distance 16: nothing else
        answers for a word 8 bits away).  At that edge every bit whose flip keeps the answer differs from the codeword."""
        z = x
        for p in range(40):
            y = z ^ (1 << p)
            if look(y) != idx:
                break
            z = y
        else:
            raise AssertionError("never left the ball")
        c = z
        for q in range(40):
            if look(z ^ (1 << q)) == idx:
                c ^= 1 << q
        return c

    found = {}
    tries = 0
    while len(found) < 24:
        x = int(rng.integers(0, 1 << 40, dtype=np.uint64)) & MASK
        tries += 1
        i = look(x)
        if i >= 0 and i not in found:
            c = edge_codeword(x, i)
            assert look(c) == i
            found[i] = c
    # affine structure: c(i) ^ c(j) depends on i ^ j only
    idx = sorted(found)
    base_i = idx[0]
    diffs = {i ^ base_i: found[i] ^ found[base_i] for i in idx[1:]}
    # Gaussian elimination over GF(2) on the 7-bit index differences
    basis = {}
    for d, v in diffs.items():
        dd, vv = d, v
        for b in sorted(basis, reverse=True):
            if dd >> b & 1 and dd.bit_length() - 1 == b:
                pass
        # reduce by existing pivots (pivot = highest set bit)
        changed = True
        while dd and changed:
            changed = False
            hb = dd.bit_length() - 1
            if hb in basis:
                dd ^= basis[hb][0]
                vv ^= basis[hb][1]
                changed = True
        if dd:
            basis[dd.bit_length() - 1] = (dd, vv)
    assert len(basis) == 7, "not enough independent index differences: %d" % len(basis)

    def lin(d):
        v = 0
        while d:
            hb = d.bit_length() - 1
            bd, bv = basis[hb]
            d ^= bd
            v ^= bv
        return v

    c0 = found[base_i] ^ lin(base_i)
    table = [c0 ^ lin(i) for i in range(128)]
    for i, c in enumerate(table):
        assert look(c) == i, (i, hex(c))
    for i in idx:
        assert table[i] == found[i]
    print("I-ISCH: 128 codewords recovered from %d lookups and verified" % tries)
    # S-ISCH: the TDMA standard's superframe sync word.  Its presence in the reference's table shows only in ties: it lies 14
    # bits from some codewords, and a word 7 bits from both is answered by whichever entry the reference's unordered_map visits
    # first (strict "<" on the distance).  Measure that outcome per codeword; any -2 among them also proves the word is there.
    S = 0x575D57F7FF
    s_wins = []
    for i, c in enumerate(table):
        d = c ^ S
        if bin(d).count("1") != 14:
            assert bin(d).count("1") > 14
            s_wins.append(0)
            continue
        bits = [p for p in range(40) if d >> p & 1]
        outs = set()
        for rot in range(3):                      # three different halves of the 14 differing bits: same answer each time
            w = c
            for p in (bits[rot:] + bits[:rot])[:7]:
                w ^= 1 << p
            outs.add(look(w))
        assert len(outs) == 1 and outs <= {i, -2}, (i, outs)
        s_wins.append(1 if outs == {-2} else 0)
    assert any(s_wins), "the S-ISCH word never wins a tie: it cannot be told from an absent entry"
    print("S-ISCH ties: %d codewords at distance 14, S-ISCH visited first for %d of them" % (sum(1 for c in table if bin(c ^ S).count("1") == 14), sum(s_wins)))
    for out in (os.path.join(ROOT, "dsd-neo_amd", "csrc", "ddn_tables_isch.h"), os.path.join(ROOT, "oracle", "ddn_tables_isch.h")):
        with open(out, "w") as f:
            f.write("// GENERATED by tools/gen_tables_isch.py from the compiled reference (isch_lookup, src/fec/ez.cpp:283-340): the 128 I-ISCH\n")
            f.write("// codewords of P25 Phase 2's (40,9,16) code by index, the S-ISCH word (answer -2), and for every codeword 14 bits from\n")
            f.write("// the S-ISCH word whether the reference's table walk meets the S-ISCH entry first (it then wins a 7 / 7 tie).\n")
            f.write("#pragma once\n#include <stdint.h>\n#define DDN_ISCH_S_WORD 0x%010XULL\n" % S)
            f.write("#define DDN_ISCH_TABLE_INIT {\\\n")
            for i in range(0, 128, 4):
                f.write("    " + ", ".join("0x%010XULL" % table[i + k] for k in range(4)) + ",\\\n")
            f.write("}\n#define DDN_ISCH_S_FIRST_INIT {\\\n")
            for i in range(0, 128, 32):
                f.write("    " + ", ".join(str(v) for v in s_wins[i:i + 32]) + ",\\\n")
            f.write("}\n")
    print("wrote ddn_tables_isch.h")
    return table


if __name__ == "__main__":
    main()
