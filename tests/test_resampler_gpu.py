"""GPU parity: batched rational resampler (k_resample through the C-ABI) vs the CPU oracle — bit-exact floats, identical
output counts across ragged call splits — and the dsd_resampler_process_block drop-in driven on a state object designed
by the reference itself (compiled reference, oracle/_ref)."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class GpuResampler:
    def __init__(self, B, L, M):
        self.h = C.c_void_p()
        assert ddn.lib().ddn_resampler_create(B, L, M, C.byref(self.h)) == 0, ddn.lib().ddn_last_error()
        self.B = B

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        n = x.shape[1]
        k = ddn.lib().ddn_resampler_out_len(self.h, n)
        out = np.zeros((self.B, k + 3), np.float32)
        rc = ddn.lib().ddn_resampler_run_host(self.h, x.ctypes.data, n, out.ctypes.data, k + 3)
        assert rc == 0, ddn.lib().ddn_last_error()
        return out[:, :k]

    def __del__(self):
        ddn.lib().ddn_resampler_destroy(self.h)


@pytest.mark.parametrize("L,M", [(1, 1), (2, 1), (1, 2), (5, 4), (4, 5), (3, 7), (10, 3), (160, 147), (147, 160), (25, 24)])
def test_resampler_vs_oracle(built, L, M):
    B, n = 37, 9000
    rng = np.random.default_rng(L * 100 + M)
    x = (rng.normal(0, 9000, (B, n)) + 12000 * np.sin(np.arange(n) * 0.07)).astype(np.float32)
    g = GpuResampler(B, L, M)
    t = np.zeros(16 * L, np.float32)
    assert ddn.lib().ddn_resampler_get_taps(g.h, t.ctypes.data, t.size) == 16 * L
    assert np.array_equal(bits(t), bits(orc.OracleResampler(L, M).taps()))
    cuts = [0, 1, 3, 17, 18, 31, 2000, 2001, 7777, n]   # blocks shorter than the 16-tap window included
    got = [g.run(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for c in range(B):
        o = orc.OracleResampler(L, M)
        for (a, b), gg in zip(zip(cuts[:-1], cuts[1:]), got):
            want = o.run(x[c, a:b])
            assert gg.shape[1] == len(want), (L, M, c, a, b)
            assert np.array_equal(bits(gg[c]), bits(want)), (L, M, c, a, b)


def test_resampler_errors(built):
    h = C.c_void_p()
    assert ddn.lib().ddn_resampler_create(4, 0, 1, C.byref(h)) != 0
    assert ddn.lib().ddn_resampler_create(4, 513, 1, C.byref(h)) != 0
    assert ddn.lib().ddn_resampler_create(4, 1, (1 << 22) + 1, C.byref(h)) != 0
    for L, M in ((1, 30), (1, 200), (3, 1000), (512, 511), (512, 40000)):  # steep decimations shrink the workgroup's output tile
        g35 = GpuResampler(2, L, M)
        x = np.sin(np.arange(50000) * 0.01).astype(np.float32) * np.ones((2, 1), np.float32)
        got = g35.run(x)
        want = orc.OracleResampler(L, M).run(x[0])
        assert got.shape[1] == len(want) and np.array_equal(bits(got[1]), bits(want)), (L, M)
    g = GpuResampler(2, 3, 2)
    x = np.ones((2, 100), np.float32)
    out = np.zeros((2, 10), np.float32)
    assert ddn.lib().ddn_resampler_run_host(g.h, x.ctypes.data, 100, out.ctypes.data, 10) != 0   # 150 outputs needed
    assert ddn.lib().ddn_resampler_out_len(g.h, 100) == 150                                      # state untouched


@needs_ref
@pytest.mark.parametrize("L,M", [(5, 4), (147, 160), (2, 1)])
def test_dropin_on_reference_state(built, L, M):
    """The state object comes from the reference's own dsd_resampler_design; one copy is advanced by the reference's
    dsd_resampler_process_block, the other by this library's function of the same name."""
    rng = np.random.default_rng(5)
    x = rng.normal(0, 8000, 4000).astype(np.float32)
    a, b = orc.RefResampler(L, M), orc.RefResampler(L, M)
    for lo, hi in [(0, 5), (5, 6), (6, 40), (40, 2500), (2500, 4000)]:
        want = a.run(x[lo:hi])
        out = np.zeros(len(want) + 2, np.float32)
        xi = np.ascontiguousarray(x[lo:hi])
        k = ddn.lib().dsd_resampler_process_block(C.byref(b.st), xi.ctypes.data, xi.size, out.ctypes.data, out.size)
        assert k == len(want) and np.array_equal(bits(out[:k]), bits(want)), (lo, hi)
        assert (a.st.phase, a.st.hist_head) == (b.st.phase, b.st.hist_head)
        ha = np.ctypeslib.as_array(a.st.hist, (32,))
        hb = np.ctypeslib.as_array(b.st.hist, (32,))
        assert np.array_equal(bits(ha), bits(hb))
    out = np.zeros(1, np.float32)
    assert ddn.lib().dsd_resampler_process_block(C.byref(b.st), x.ctypes.data, 100, out.ctypes.data, 1) == -1
