"""CPU: the P25 Phase 2 RS(63,35) errors-and-erasures restatement (oracle/ddn_oracle_rs.c, orc_ez_rs28) pinned against the
reference's compiled ez_rs28_ess / _facch / _sacch (src/fec/ez.cpp over the vendored ezpwd decoder, oracle/_ref): return
value and payload bits, for clean words, correctable errors + erasures, and words beyond the code's capacity."""
import os

import numpy as np
import pytest

import orc
import rs28

FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def test_encoder_makes_codewords_the_oracle_accepts():
    rng = np.random.default_rng(1 + FZ)
    for kind in range(3):
        for _ in range(20):
            pl, pa, er, sent = rs28.make_case(rng, kind, 0, 0, 0)
            got, rc = rs28.oracle_rs28(kind, pl, pa, er)
            # FACCH / SACCH: the punctured parity symbols are erasures whose true values are (almost surely) non-zero
            assert rc == (0 if kind == 0 else len(er)) and np.array_equal(got, sent)


@needs_ref
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rs28_oracle_equals_compiled_reference(kind):
    rng = np.random.default_rng(100 + kind + FZ)
    punct = 28 - rs28.N_PAR[kind]
    seen = {"ok": 0, "fail": 0, "fixed": 0}
    for it in range(700):
        n_extra = int(rng.integers(0, 29 - punct))
        budget = 28 - punct - n_extra
        mode = it % 4
        if mode == 0:      # inside the capacity: 2 e + erasures <= 28
            n_err = int(rng.integers(0, budget // 2 + 1))
        elif mode == 1:    # at and just over the edge
            n_err = budget // 2 + int(rng.integers(0, 3))
        elif mode == 2:    # far beyond
            n_err = int(rng.integers(budget // 2 + 1, 30))
        else:              # errors that the erasures cover
            n_err = int(rng.integers(0, 12))
        hits = int(rng.integers(0, n_err + 1)) if mode == 3 else int(rng.integers(0, 3))
        pl, pa, er, sent = rs28.make_case(rng, kind, n_err, n_extra, hits)
        want, rc_w = rs28.ref_rs28(kind, pl, pa, er)
        got, rc_g = rs28.oracle_rs28(kind, pl, pa, er)
        assert rc_g == rc_w, (it, n_err, n_extra, rc_g, rc_w)
        assert np.array_equal(got, want), (it, n_err, n_extra, rc_w)
        seen["ok" if rc_w >= 0 else "fail"] += 1
        seen["fixed"] += int(rc_w > 0 and np.array_equal(want, sent))
    assert seen["ok"] > 200 and seen["fail"] > 100 and seen["fixed"] > 150, seen
