"""GPU parity: batched P25p1 slicer (10-byte capture records) and P25 matched filter vs the reference's goldens and
the CPU oracle — every field bit-exact, carried state included."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
from conftest import golden

pytestmark = pytest.mark.gpu


class GpuSlicer:
    def __init__(self, B, negative=0):
        self.h = C.c_void_p()
        assert ddn.lib().ddn_slicer_batch_create(B, negative, C.byref(self.h)) == 0, ddn.lib().ddn_last_error()
        self.B = B

    def run(self, sym):
        sym = np.ascontiguousarray(sym, np.float32)
        n = sym.shape[1]
        rec = np.zeros((self.B, n, 10), np.uint8)
        assert ddn.lib().ddn_p25_slicer_run_host(self.h, sym.ctypes.data, n, rec.ctypes.data) == 0
        return rec

    def filt(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        assert ddn.lib().ddn_p25_matched_filter_run_host(self.h, x.ctypes.data, x.shape[1], y.ctypes.data) == 0
        return y

    def thresholds(self, ch):
        t = np.zeros(5, np.float32)
        assert ddn.lib().ddn_slicer_batch_get_thresholds(self.h, ch, t.ctypes.data) == 0
        return t

    def __del__(self):
        ddn.lib().ddn_slicer_batch_destroy(self.h)


def test_slicer_golden(built):
    g = golden("sym_p25_slicer.npz")
    for name, neg in (("pos", 0), ("neg", 1)):
        s = GpuSlicer(1, neg)
        sym = g[name + "_sym"]
        a = s.run(sym[None, :1777])
        b = s.run(sym[None, 1777:])     # carried window / rings / sums across calls
        rec, sy = orc.unpack_records10(np.concatenate([a, b], axis=1)[0])
        assert np.array_equal(rec, g[name + "_rec"])
        assert np.array_equal(sy.view(np.uint32), sym.view(np.uint32))
        assert np.array_equal(s.thresholds(0).view(np.uint32), g[name + "_thr_last"].view(np.uint32))


def test_slicer_batch_vs_oracle(built):
    B, n = 70, 3000
    sym = np.stack([orc.synth_c4fm_symbols(100 + c, n, scale=0.5 + 0.02 * c, noise=300 + 60 * c) for c in range(B)])
    sym[5] = 0.0                       # dead channel: thresholds collapse, spans hit their epsilon floor
    sym[6, 1000:] = 1e9                # absurd level: magnitude conversion wraps like the host's lrintf narrowing
    rec, _ = orc.unpack_records10(GpuSlicer(B).run(sym))
    want, thr = orc.oracle_slicer(sym)
    assert np.array_equal(rec, want)


def test_slicer_call_splits_mix_sequential_and_parallel_kernels(built):
    """Calls shorter than 256 symbols take the one-lane-per-channel kernel, longer ones the parallel decomposition
    (ddn_slicer_par.hip); both carry the same state (128-symbol window, 1024-deep rings, binary64 sums), so any split of a
    stream - including one that wraps the 1024-deep ring several times - gives the oracle's records."""
    B, n = 9, 5200
    sym = np.stack([orc.synth_c4fm_symbols(300 + c, n, scale=0.7 + 0.05 * c, noise=500) for c in range(B)])
    cuts = [0, 10, 300, 301, 557, 1700, 1955, 4100, 4130, n]
    s = GpuSlicer(B, 1)
    got = np.concatenate([s.run(sym[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
    rec, _ = orc.unpack_records10(got)
    want, thr = orc.oracle_slicer(sym, negative=1)
    assert np.array_equal(rec, want)
    for c in (0, B - 1):
        assert np.array_equal(s.thresholds(c).view(np.uint32), np.asarray(thr[c], np.float32).view(np.uint32))


def test_matched_filter(built):
    g = golden("sym_p25_matched_filter.npz")
    s = GpuSlicer(1)
    y = np.concatenate([s.filt(g["x"][None, :1000]), s.filt(g["x"][None, 1000:1050]), s.filt(g["x"][None, 1050:])],
                       axis=1)[0]
    assert np.array_equal(y.view(np.uint32), g["y"].view(np.uint32))
    B, n = 9, 5000
    x = np.stack([orc.synth_c4fm_symbols(50 + c, n) for c in range(B)])
    assert np.array_equal(GpuSlicer(B).filt(x).view(np.uint32), orc.oracle_p25_filter(x).view(np.uint32))
