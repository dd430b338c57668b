"""GPU parity: the batched P25p1 receive loop (ddn_p25_rx_*) against the CPU oracle (oracle/ddn_oracle_rx.c) on
synthetic P25p1 frame streams — symbols, records, flags, counts and carried thresholds bit-exact, for every
channels-per-wavefront layout, with and without the matched filter, across call boundaries."""
import numpy as np
import pytest

import ddn
import orc

pytestmark = pytest.mark.gpu


def _oracle_all(x, lock, use_filter, splits=None):
    out = []
    for c in range(x.shape[0]):
        rx = orc.OracleP25Rx(lock_symbols=lock, use_filter=use_filter)
        if splits is None:
            sym, rec, fl = rx.run(x[c])
        else:
            parts = [rx.run(x[c][a:b]) for a, b in zip(splits[:-1], splits[1:])]
            sym, rec, fl = (np.concatenate([p[k] for p in parts]) for k in range(3))
        out.append((sym, rec, fl, rx.thresholds()))
    return out


def _compare(got, want, ch):
    rec, fl, cnt = got
    sym_o, rec_o, fl_o, _ = want
    k = int(cnt[ch])
    assert k == len(sym_o), (ch, k, len(sym_o))
    r4, sy = orc.unpack_records10(rec[ch, :k])
    assert np.array_equal(sy.view(np.uint32), sym_o.view(np.uint32)), ch
    assert np.array_equal(fl[ch, :k], fl_o), ch
    assert np.array_equal(r4, rec_o), ch


@pytest.mark.parametrize("cpw", [8, 16, 32, 64])
@pytest.mark.parametrize("use_filter", [0, 1])
def test_rx_vs_oracle(built, cpw, use_filter):
    B, n, frame = 70, 30000, 432
    x, _, _ = orc.synth_p25_disc(31, B, n, frame_dibits=frame, noise=500.0)
    x[3] = 0.0                                    # dead channel
    x[4] = np.random.default_rng(1).normal(0, 6000, n).astype(np.float32)   # noise only: hunts (and slips) forever
    x[5] *= -1.0                                  # inverted polarity
    want = _oracle_all(x, frame - 24, use_filter)
    rx = ddn.P25Rx(B, lock_symbols=frame - 24, use_matched_filter=use_filter, channels_per_wave=cpw)
    got = rx.run(x)
    for c in range(B):
        _compare(got, want[c], c)
        assert np.array_equal(rx.thresholds(c).view(np.uint32), want[c][3].view(np.uint32)), c
    assert (got[1][0] & 2).sum() >= 4             # syncs were actually found


def test_rx_call_split_invariance(built):
    """Carried state: timing mid-symbol, hunting window, filter cold start straddling a call boundary."""
    B, frame = 20, 432
    splits = [0, 7, 701, 702, 5000, 5090, 12345, 20000]
    x, _, _ = orc.synth_p25_disc(32, B, splits[-1], frame_dibits=frame)
    want = _oracle_all(x, frame - 24, 1)
    rx = ddn.P25Rx(B, lock_symbols=frame - 24, use_matched_filter=1)
    recs, fls = [[] for _ in range(B)], [[] for _ in range(B)]
    for a, b in zip(splits[:-1], splits[1:]):
        rec, fl, cnt = rx.run(x[:, a:b])
        for c in range(B):
            recs[c].append(rec[c, :cnt[c]])
            fls[c].append(fl[c, :cnt[c]])
    for c in range(B):
        rec = np.concatenate(recs[c])
        fl = np.concatenate(fls[c])
        r4, sy = orc.unpack_records10(rec)
        assert np.array_equal(sy.view(np.uint32), want[c][0].view(np.uint32)), c
        assert np.array_equal(r4, want[c][1]) and np.array_equal(fl, want[c][2]), c
        assert np.array_equal(rx.thresholds(c).view(np.uint32), want[c][3].view(np.uint32)), c


def test_rx_fractional_sps_and_payload(built):
    """50 kHz / 4.8 kHz: 10 or 11 samples per symbol from the fractional accumulator (no matched filter); and at
    48 kHz the decoded payload equals what was sent."""
    B = 8
    x, _, _ = orc.synth_p25_disc(33, B, 20000, frame_dibits=432)
    rx = ddn.P25Rx(B, out_rate=50000, lock_symbols=408, use_matched_filter=0)
    got = rx.run(x)
    for c in range(B):
        o = orc.OracleP25Rx(out_rate=50000, lock_symbols=408, use_filter=0)
        sym, rec, fl = o.run(x[c])
        _compare(got, (sym, rec, fl, None), c)
    frame = 864
    x, dib, _ = orc.synth_p25_disc(34, B, 60000, frame_dibits=frame)
    rx = ddn.P25Rx(B, lock_symbols=frame - 24)
    rec, fl, cnt = rx.run(x)
    for c in range(B):
        acc = np.flatnonzero(fl[c, :cnt[c]] & 2)
        assert len(acc) >= 4
        a = acc[3]
        payload = rec[c, a + 1:a + 1 + frame - 24, 0] & 3
        hits = [f for f in range(dib.shape[1] // frame) if np.array_equal(dib[c, f * frame + 24:(f + 1) * frame], payload)]
        assert len(hits) == 1, c


def test_rx_full_batch_properties(built):
    """BASELINE configs[2] shape: 4096 channels x 48000 discriminator samples.  Size-independent properties over the
    whole batch (every replica of a channel yields byte-identical records whatever its position in the batch, symbol
    counts agree with the oracle's) + a sample of channels bit for bit against the oracle."""
    import torch
    B, n, base_ch = 4096, 48000, 64
    base, _, _ = orc.synth_p25_disc(9, base_ch, n, frame_dibits=864)
    x = torch.from_numpy(np.tile(base, (B // base_ch, 1))).cuda()
    rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1)
    ms = ddn.lib().ddn_p25_rx_max_symbols(rx.h, n)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert ddn.lib().ddn_p25_rx_run(rx.h, x.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    torch.cuda.synchronize()
    r = rec.view(B // base_ch, base_ch, ms, 10)
    f = fl.view(B // base_ch, base_ch, ms)
    c = cnt.view(B // base_ch, base_ch)
    assert bool((c == c[0:1]).all()) and bool((f == f[0:1]).all()) and bool((r == r[0:1]).all())
    cn = c[0].cpu().numpy()
    for ch in (0, 17, 63):
        o = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
        sym, rec4, flo = o.run(base[ch])
        k = int(cn[ch])
        r4, sy = orc.unpack_records10(rec[B - base_ch + ch, :k].cpu().numpy())   # the last replica
        assert k == len(sym) and np.array_equal(sy.view(np.uint32), sym.view(np.uint32))
        assert np.array_equal(r4, rec4) and np.array_equal(fl[B - base_ch + ch, :k].cpu().numpy(), flo)


def test_rx_per_channel_lock_symbols(built):
    """One batch mixing traffic classes: every channel gets its own in-frame length (ddn_p25_rx_set_lock_symbols) and
    must equal an oracle instance configured with that length."""
    import ctypes as C
    B, n = 20, 24000
    frames = [180, 360, 432, 864]
    xs = [orc.synth_p25_disc(40 + c, 1, n, frame_dibits=frames[c % 4])[0][0] for c in range(B)]
    x = np.stack(xs)
    locks = np.array([frames[c % 4] - 24 if c % 5 else 0 for c in range(B)], np.int32)     # some channels never lock
    rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1)
    assert ddn.lib().ddn_p25_rx_set_lock_symbols(rx.h, locks.ctypes.data) == 0
    bad = locks.copy()
    bad[3] = -1
    assert ddn.lib().ddn_p25_rx_set_lock_symbols(rx.h, bad.ctypes.data) != 0
    rec, fl, cnt = rx.run(x)
    for c in range(B):
        o = orc.OracleP25Rx(lock_symbols=int(locks[c]), use_filter=1)
        sym, rec4, flo = o.run(x[c])
        k = int(cnt[c])
        r4, sy = orc.unpack_records10(rec[c, :k])
        assert k == len(sym) and np.array_equal(sy.view(np.uint32), sym.view(np.uint32)), c
        assert np.array_equal(r4, rec4) and np.array_equal(fl[c, :k], flo), c
    # NULL restores the configured value everywhere
    assert ddn.lib().ddn_p25_rx_set_lock_symbols(rx.h, None) == 0
