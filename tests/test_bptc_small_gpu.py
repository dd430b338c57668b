"""GPU: BPTC(128,77) / reverse-channel BPTC 16 x 2 kernels and the reference-named calls against the restatement."""
import numpy as np
import pytest

import bptc_small as bs
import ddn

pytestmark = pytest.mark.gpu


def test_bptc_128x77_batch_equals_oracle(built):
    rng = np.random.default_rng(40)
    cs = bs.cases128(rng, 3000)
    x = np.stack(cs).reshape(len(cs), 128)
    out = np.zeros((len(cs), 77), np.uint8)
    errs = np.zeros(len(cs), np.uint32)
    assert ddn.lib().ddn_fec_bptc_128x77_host(x.ctypes.data, len(cs), out.ctypes.data, errs.ctypes.data) == 0
    for i, m in enumerate(cs):
        rc, want, _ = bs.oracle_128x77(m)
        assert errs[i] == rc and np.array_equal(out[i], want), i
    one = np.zeros(77, np.uint8)
    m = np.ascontiguousarray(cs[1])
    assert ddn.lib().BPTC_128x77_Extract_Data(m.ctypes.data, one.ctypes.data) == errs[1] and np.array_equal(one, out[1])
    assert ddn.lib().ddn_fec_bptc_128x77_host(None, 1, None, None) != 0


@pytest.mark.parametrize("odd", [0, 1])
def test_bptc_16x2_batch_equals_oracle(built, odd):
    rng = np.random.default_rng(50 + odd)
    xs = [bs.word32(rng, i % 4, odd) if i % 5 else rng.integers(0, 4, 32).astype(np.uint8) for i in range(3000)]
    x = np.stack(xs)
    out = np.zeros((len(xs), 32), np.uint8)
    errs = np.zeros(len(xs), np.uint32)
    assert ddn.lib().ddn_fec_bptc_16x2_host(x.ctypes.data, len(xs), odd, out.ctypes.data, errs.ctypes.data) == 0
    for i, v in enumerate(xs):
        rc, want, _ = bs.oracle_16x2(v, odd)
        assert errs[i] == rc and np.array_equal(out[i], want), i
    one = np.zeros(32, np.uint8)
    assert ddn.lib().BPTC_16x2_Extract_Data(xs[3].copy().ctypes.data, one.ctypes.data, odd) == errs[3] and np.array_equal(one, out[3])
