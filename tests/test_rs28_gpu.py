"""GPU: P25 Phase 2 RS(63,35) sections (ddn_fec_rs28_* / ez_rs28_*) against the oracle restatement that
tests/test_oracle_rs28.py pins to the compiled reference: status and payload bits, clean / correctable / beyond capacity."""
import os

import numpy as np
import pytest

import ddn
import rs28

pytestmark = pytest.mark.gpu
FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))


def _batch(rng, kind, n):
    punct = 28 - rs28.N_PAR[kind]
    pl, pa, er, ne, sent = [], [], np.zeros((n, 28), np.int8), np.zeros(n, np.uint8), []
    for it in range(n):
        n_extra = int(rng.integers(0, 29 - punct))
        budget = 28 - punct - n_extra
        mode = it % 4
        n_err = (int(rng.integers(0, budget // 2 + 1)) if mode == 0 else budget // 2 + int(rng.integers(0, 3)) if mode == 1
                 else int(rng.integers(budget // 2 + 1, 30)) if mode == 2 else int(rng.integers(0, 12)))
        hits = int(rng.integers(0, n_err + 1)) if mode == 3 else int(rng.integers(0, 3))
        a, b, e, s = rs28.make_case(rng, kind, n_err, n_extra, hits)
        pl.append(a), pa.append(b), sent.append(s)
        er[it, :e.size] = e
        ne[it] = e.size
    return np.stack(pl).astype(np.uint8), np.stack(pa).astype(np.uint8), er, ne, np.stack(sent)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rs28_batch_equals_oracle(built, kind):
    rng = np.random.default_rng(300 + kind + FZ)
    n = 1500
    pl, pa, er, ne, sent = _batch(rng, kind, n)
    got = pl.copy()
    st = np.zeros(n, np.int32)
    assert ddn.lib().ddn_fec_rs28_host(kind, got.ctypes.data, pa.ctypes.data, er.ctypes.data, ne.ctypes.data, n, st.ctypes.data) == 0
    ok = fail = 0
    for i in range(n):
        want, rc = rs28.oracle_rs28(kind, pl[i].astype(np.int32), pa[i].astype(np.int32), er[i, :ne[i]].astype(np.int32))
        assert st[i] == rc, (i, st[i], rc)
        assert np.array_equal(got[i], want), i
        ok += rc >= 0
        fail += rc < 0
    assert ok > 400 and fail > 200


def test_rs28_drop_in_names(built):
    rng = np.random.default_rng(77 + FZ)
    l = ddn.lib()
    for kind, fn in enumerate((l.ez_rs28_ess, l.ez_rs28_facch, l.ez_rs28_sacch)):
        for n_err in (0, 3, 20):
            pl, pa, er, sent = rs28.make_case(rng, kind, n_err, 2, 1)
            want, rc = rs28.oracle_rs28(kind, pl, pa, er)
            a, b = pl.astype(np.int32).copy(), pa.astype(np.int32).copy()
            assert fn(a.ctypes.data, b.ctypes.data, er.ctypes.data, int(er.size)) == rc
            assert np.array_equal(a, want)
        # no erasures at all: NULL pointer
        pl, pa, er, sent = rs28.make_case(rng, 0, 4, 0, 0)
        want, rc = rs28.oracle_rs28(0, pl, pa, np.zeros(0, np.int32))
        a = pl.astype(np.int32).copy()
        assert l.ez_rs28_ess(a.ctypes.data, pa.astype(np.int32).ctypes.data, None, 0) == rc and np.array_equal(a, want) and rc == 4
