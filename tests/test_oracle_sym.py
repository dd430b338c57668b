"""CPU: slicer / soft-decision and matched-filter restatement vs goldens produced by the reference's own compiled
dsd_dibit.c and dsd_filters.c."""
import numpy as np

import orc
from conftest import golden


def test_slicer_golden(built):
    g = golden("sym_p25_slicer.npz")
    for name, neg in (("pos", 0), ("neg", 1)):
        rec, thr = orc.oracle_slicer(g[name + "_sym"][None], neg)
        assert np.array_equal(rec[0], g[name + "_rec"])
        assert np.array_equal(thr[0].view(np.uint32), g[name + "_thr_last"].view(np.uint32))
    # polarity flips the dibit mapping: +3 level is dibit 1 on positive sync, 3 on negative sync
    rp, _ = orc.oracle_slicer(np.full((1, 300), 24000.0, np.float32), 0)
    rn, _ = orc.oracle_slicer(np.full((1, 300), 24000.0, np.float32), 1)
    assert rp[0, -1, 0] == 1 and rn[0, -1, 0] == 3


def test_matched_filter_golden_and_properties(built):
    g = golden("sym_p25_matched_filter.npz")
    y = orc.oracle_p25_filter(g["x"][None])[0]
    assert np.array_equal(y.view(np.uint32), g["y"].view(np.uint32))
    # unity DC gain (src/dsp/dsd_filters.c:138-152): a constant comes out unchanged after the 90-sample fill
    c = orc.oracle_p25_filter(np.full((1, 400), 1000.0, np.float32))[0]
    assert abs(c[-1] - 1000.0) < 1e-2
