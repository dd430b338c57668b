"""GPU: the P25 confirmed-data rate 3/4 LLR list decoder (k_p25_mbf34_list behind ddn_fec_p25_mbf34_list_batch and the reference-named
p25_mbf34_decode_soft_list) against the restatement pinned to the compiled reference (tests/test_oracle_mbf34.py)."""
import ctypes as C

import numpy as np
import pytest

import ddn
from test_oracle_mbf34 import FZ, cases, oracle_list

pytestmark = pytest.mark.gpu


def test_mbf34_list_batch_equals_oracle(built):
    rng = np.random.default_rng(23 + FZ)
    cs = cases(rng, 600)
    llr = np.stack([c[1] for c in cs]).astype(np.int16)
    n = len(cs)
    for mx in (8, 3):
        cand = np.zeros((n, 8, 24), np.uint8)
        cnt = np.full(n, -1, np.int32)
        assert ddn.lib().ddn_fec_p25_mbf34_list_host(llr.ctypes.data, n, mx, cand.ctypes.data, cnt.ctypes.data) == 0
        for i in range(n):
            k, by, me = oracle_list(llr[i], mx)
            assert cnt[i] == k, (i, mx, cnt[i], k)
            assert np.array_equal(cand[i, :k, :18], by), (i, mx)
            assert np.array_equal(cand[i, :k, 20:24].copy().view(np.uint32).reshape(-1), me), (i, mx)
    # clean blocks come back as sent
    for i in range(0, n, 5):
        assert np.array_equal(cand[i, 0, :18], cs[i][0])


def test_reference_named_call(built):
    rng = np.random.default_rng(5 + FZ)
    data, llr = cases(rng, 3)[1]
    cand = np.zeros((8, 24), np.uint8)
    dummy = np.zeros(98, np.uint8)
    k = ddn.lib().p25_mbf34_decode_soft_list(dummy.ctypes.data, np.ascontiguousarray(llr).ctypes.data, cand.ctypes.data, 8)
    n, by, me = oracle_list(llr)
    assert k == n and np.array_equal(cand[:n, :18], by)
    assert ddn.lib().p25_mbf34_decode_soft_list(None, None, None, 8) == 0
