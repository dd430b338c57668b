"""GPU: batched P25 Phase 2 I-ISCH lookup (k_isch_lookup) and the reference-named single-word calls against the oracle."""
import ctypes as C

import numpy as np
import pytest

import ddn
from test_oracle_isch import FZ, oracle_hard, oracle_soft, table, words, words_high

pytestmark = pytest.mark.gpu


def test_isch_lookup_batch_equals_oracle(built):
    rng = np.random.default_rng(31 + FZ)
    ws = words(rng, 20000) + words_high(rng, 4000)      # incl. stray bits above the field (bit 47): counted, never indexed
    w = np.array(ws, dtype=np.uint64)
    out = np.full(len(ws), 99, np.int32)
    assert ddn.lib().ddn_fec_isch_lookup_host(w.ctypes.data, None, len(ws), out.ctypes.data) == 0
    exp = np.array([oracle_hard(int(x)) for x in ws], np.int32)
    assert np.array_equal(out, exp)
    assert (exp >= 0).any() and (exp == -2).any()
    rel = rng.integers(0, 256, (len(ws), 40)).astype(np.uint8)
    rel[::3] = rng.integers(0, 3, (len(rel[::3]), 40))
    assert ddn.lib().ddn_fec_isch_lookup_host(w.ctypes.data, rel.ctypes.data, len(ws), out.ctypes.data) == 0
    exp = np.array([oracle_soft(int(x), rel[i]) for i, x in enumerate(ws)], np.int32)
    assert np.array_equal(out, exp)


def test_isch_reference_named_calls(built):
    t = table()
    rng = np.random.default_rng(5 + FZ)
    assert ddn.lib().isch_lookup(t[77]) == 77 and ddn.lib().isch_lookup(0x575D57F7FF) == -2
    for wv in words(rng, 40)[:60]:
        rel = rng.integers(0, 256, 40).astype(np.uint8)
        assert ddn.lib().isch_lookup(wv) == oracle_hard(wv)
        assert ddn.lib().isch_lookup_soft(wv, rel.ctypes.data) == oracle_soft(wv, rel)
        assert ddn.lib().isch_lookup_soft(wv, None) == oracle_hard(wv)
    assert ddn.lib().ddn_fec_isch_lookup_host(None, None, 1, None) != 0
