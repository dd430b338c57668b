"""Carrier loss in the receive loop (src/dsp/dsd_frame_sync.c:2753-2760,3037-3053 -> noCarrier(), src/engine/engine.c:1838-1847):
a hunt of 1800 symbols without a sync (10200 after an inverted-polarity sync) clears the crossing latch, gates the matched
filter off again and makes the next symbol re-initialise timing and slicer (src/dsp/dsd_symbol.c:1306-1341)."""
import numpy as np
import pytest

import ddn
import orc


def traffic(seed, negative, gap_symbols, n_after=3):
    """two frames, `gap_symbols` of noise, n_after frames; discriminator scale"""
    rng = np.random.default_rng(seed)
    a, _, _ = orc.synth_p25_disc(seed, 1, 2 * 8640 + 400, frame_dibits=864, negative=negative)
    b, _, _ = orc.synth_p25_disc(seed + 1, 1, n_after * 8640 + 400, frame_dibits=864, negative=negative)
    gap = (rng.standard_normal(gap_symbols * 10) * 900.0).astype(np.float32)
    return np.concatenate([a[0], gap, b[0]]).astype(np.float32)


def test_oracle_resets_after_1800_symbols_without_sync(built):
    x = traffic(3, False, 3500)
    full = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
    sym, rec4, fl = full.run(x)
    acc = np.flatnonzero(fl & 2)
    k = int(np.argmax(np.diff(acc)))                          # the noise gap
    assert np.diff(acc)[k] > 3400
    # the frame ends 840 symbols after its sync, the hunt then runs 1800 symbols dry: stop a little later, inside the gap
    cut_sym = acc[k] + 1 + 840 + 1800 + 60
    rx = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
    s1, _, f1 = rx.run(x[:cut_sym * 10])
    assert not (f1[acc[k] + 1:] & 2).any()
    c, um, lm, mx, mn, mxr, mnr = rx.thresholds()
    # symbol_reset_rtl_fsk_discriminator_slicer's values; the hunt copies max / min into the reference levels (dsd_frame_sync.c:2316-2336)
    assert (c, um, lm, mx, mn, mxr, mnr) == (0.0, 20000.0, -20000.0, 30000.0, -30000.0, 30000.0, -30000.0)
    # before the timeout the thresholds are still those of the last frame
    rx2 = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
    rx2.run(x[:(acc[k] + 1 + 840 + 1700) * 10])
    assert 0 < rx2.thresholds()[3] < 20000
    # a split call carries hunt position and pending reset: same records as one call
    rx3 = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
    a1 = rx3.run(x[:(acc[k] + 1 + 840 + 1799) * 10 + 3])
    a2 = rx3.run(x[(acc[k] + 1 + 840 + 1799) * 10 + 3:])
    assert np.array_equal(np.concatenate([a1[1], a2[1]]), rec4) and np.array_equal(np.concatenate([a1[2], a2[2]]), fl)


@pytest.mark.gpu
@pytest.mark.parametrize("cpw", [8, 16, 32, 64])
@pytest.mark.parametrize("negative,gap", [(False, 3600), (False, 2560), (True, 11400)])
def test_carrier_loss_gpu_equals_oracle(built, cpw, negative, gap):
    B = 5
    tr = [traffic(10 + c, negative, gap + 37 * c) for c in range(B)]
    n = min(len(t) for t in tr)
    xs = np.stack([t[:n] for t in tr])
    want = [orc.OracleP25Rx(lock_symbols=840, use_filter=1).run(xs[c]) for c in range(B)]
    for split in (None, n // 2 + 123):
        rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1, channels_per_wave=cpw)
        if split is None:
            rec, fl, cnt = rx.run(xs)
            parts = [(rec, fl, cnt)]
        else:
            parts = [rx.run(np.ascontiguousarray(xs[:, :split])), rx.run(np.ascontiguousarray(xs[:, split:]))]
        for c in range(B):
            r4 = np.concatenate([orc.unpack_records10(p[0][c, :p[2][c]])[0] for p in parts])
            sy = np.concatenate([orc.unpack_records10(p[0][c, :p[2][c]])[1] for p in parts])
            f = np.concatenate([p[1][c, :p[2][c]] for p in parts])
            ws, wr, wf = want[c]
            assert len(sy) == len(ws) and np.array_equal(sy.view(np.uint32), ws.view(np.uint32)), (cpw, c, split)
            assert np.array_equal(r4, wr) and np.array_equal(f, wf), (cpw, c, split)
