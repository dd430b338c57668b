"""GPU parity of the receive loop with the reference's handlers inside it (ddn_p25_rx_set_handlers): records, flags, counts
and the list of handler decisions equal the oracle's (oracle/ddn_oracle_handlers.c) - on the reference's own captures and on
synthetic traffic of every frame type, clean and noisy (the NID's Chase search and the list decoder proper), in one call and
across call splits."""
import numpy as np
import pytest

import ddn
import orc
import p25gen
from conftest import golden

pytestmark = pytest.mark.gpu


def _oracle(x, use_filter=1):
    rx = orc.OracleP25Rx(lock_symbols=-1, use_filter=use_filter)
    sym, rec, fl = rx.run(x)
    ev = [[e[0], e[1], e[2], (e[3] & 0xFFFF) | ((e[4] & 0xFFFF) << 16)] for e in rx.events.rows()]
    return sym, rec, fl, np.array(ev, np.int64).reshape(-1, 4), rx.events.data()


def _check(rec, fl, cnt, events, n_events, c, want, event_data=None):
    sym_o, rec_o, fl_o, ev_o, evd_o = want
    k = int(cnt[c])
    assert k == len(sym_o), (c, k, len(sym_o))
    r4, sy = orc.unpack_records10(rec[c, :k])
    bad = np.flatnonzero(fl[c, :k] != fl_o)
    assert bad.size == 0, (c, bad[:5], fl[c, bad[:5]], fl_o[bad[:5]])
    assert np.array_equal(sy.view(np.uint32), sym_o.view(np.uint32)), c
    assert np.array_equal(r4, rec_o), c
    ne = int(n_events[c])
    got = events[c, :ne].astype(np.int64) & np.array([-1, -1, -1, 0xFFFFFFFF])
    assert ne == len(ev_o) and np.array_equal(got, ev_o & np.array([-1, -1, -1, 0xFFFFFFFF])), (c, got[:6], ev_o[:6])
    if event_data is not None:       # what each decision decoded: NID result words, TSDU / PDU header block bytes
        bad = np.flatnonzero((event_data[c, :ne] != evd_o).any(axis=1))
        assert bad.size == 0, (c, bad[:4], event_data[c, bad[:2]], evd_o[bad[:2]], ev_o[bad[:2]])


@pytest.mark.parametrize("name", ["iq_p25p1_c4fm_cc.npz", "iq_p25p1_c4fm_vc.npz"])
def test_reference_captures_with_handlers(built, name):
    g = golden(name)
    iq = np.ascontiguousarray(g["iq"])
    disc = ddn.Batch(1, block_len=8192).run_host(iq[None], iq.shape[0])
    want = _oracle(disc[0])
    rx = ddn.P25Rx(1, use_matched_filter=1, handlers=True)
    rec, fl, cnt = rx.run(disc)
    _check(rec, fl, cnt, rx.events, rx.n_events, 0, want, rx.event_data)
    ev = rx.events[0, :rx.n_events[0]]
    if "cc" in name:      # every TSDU: NID ok, three blocks, CRC16 good, last-block flag on the third
        tsbk = ev[ev[:, 1] == 2]
        assert len(tsbk) >= 72 and np.all(tsbk[:, 3] & 1) and np.all(((tsbk[:, 3] >> 24) & 1) == (tsbk[:, 2] == 2))
    else:
        nid = ev[ev[:, 1] == 1][1:]
        assert set((nid[:, 3] >> 16) & 0xFF) == {5, 10}


def _traffic(seed, n, noise, weak_nids=False):
    """one channel's dibit stream: frames of every kind back to back"""
    rng = np.random.default_rng(seed)
    parts = []
    kinds = rng.permutation(9)
    while sum(len(p) for p in parts) * 10 < n:
        for k in kinds:
            if k == 0:
                parts.append(p25gen.make_frames(rng, 2, 0x293, crc=True, blocks=3)[0])
            elif k == 1:
                parts.append(p25gen.make_frames(rng, 2, 0x293, crc=True, blocks=1)[0])
            elif k == 2:
                parts.append(p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=2)[0])
            elif k == 3:
                parts.append(p25gen.make_pdu(rng, 0x293, int(rng.integers(0, 5))))
            elif k == 4:
                parts.append(p25gen.make_pdu(rng, 0x293, 12, sap=61))
            elif k == 5:
                parts.append(p25gen.make_pdu(rng, 0x293, 3, good_crc=False))
            elif k == 6:
                parts.append(p25gen.frame_with_duid(rng, 0x293, [0x0, 0x3, 0xF, 0x5, 0xA][int(rng.integers(0, 5))],
                                                    [339, 15, 159, 807, 807][0] + 30))
            elif k == 7:
                d = [0x3, 0xF, 0x5][int(rng.integers(0, 3))]
                parts.append(p25gen.frame_with_duid(rng, 0x293, d, {0x3: 15, 0xF: 159, 0x5: 807}[d] + 3))
            else:
                parts.append(p25gen.frame_with_duid(rng, 0x293, 0x9, 40))      # undefined DUID: the handler returns at once
            parts.append(np.full(int(rng.integers(0, 30)), 0, np.int8))
    dib = np.concatenate(parts)
    scale = np.ones(len(dib))
    if weak_nids:           # every third frame: a NID only the Chase search decodes
        starts = np.cumsum([0] + [len(p) for p in parts])[:-1:2]
        for st in starts[::3]:
            sc = p25gen.weaken_nid(dib, int(st), rng, strong=int(rng.integers(9, 12)), weak=int(rng.integers(2, 4)))
            scale = np.minimum(scale, sc)
    return p25gen.modulate_disc(dib, lead=200 + 13 * (seed % 17), noise=noise, seed=seed, scale=scale)[:n]


@pytest.mark.parametrize("cpw,fil", [(4, 0), (8, 0), (16, 0), (4, 1), (8, 1), (16, 1)])
def test_every_frame_type_clean_and_noisy(built, cpw, fil):
    """fil = 1: the matched filter computed inside the loop kernel (ddn_p25_rx_set_filter_in_loop; sixteen channels per workgroup keep
    the filter kernel whatever the switch says) - same records, flags and decisions"""
    B, n = 24, 40000
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        noise = [80.0, 3000.0, 12000.0, 16000.0][c % 4]
        s = _traffic(100 + c, n, noise, weak_nids=(c % 3 == 0))
        x[c, :len(s)] = s
    x[5] *= -1.0                                   # inverted polarity
    x[7] = np.random.default_rng(7).normal(0, 6000, n).astype(np.float32)
    want = [_oracle(x[c], 1) for c in range(B)]
    rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=cpw, handlers=True, max_events=2048, filter_in_loop=bool(fil))
    rec, fl, cnt = rx.run(x)
    for c in range(B):
        _check(rec, fl, cnt, rx.events, rx.n_events, c, want[c], rx.event_data)
    ev = np.concatenate([rx.events[c, :rx.n_events[c]] for c in range(B)])
    nid = ev[ev[:, 1] == 1]
    tsbk = ev[ev[:, 1] == 2]
    assert (nid[:, 2] > 0).sum() > 100 and (nid[:, 2] <= 0).sum() > 3          # decoded and failed NIDs
    assert (nid[:, 2] == 2).sum() >= 1                                         # parity override
    assert ((tsbk[:, 3] >> 16) & 0xFF).max() >= 1                              # a list candidate other than the best one
    assert (ev[:, 1] == 3).sum() > 10


@pytest.mark.parametrize("cpw", [4, 8])
def test_hunting_pass_of_all_rows_at_once_equals_one_owner_at_a_time(built, monkeypatch, cpw):
    """two / four channels per recurrence wave: the lanes that hunt take the bulk hunting pass together, each in its own row of the
    wavefront (ddn_rx.hip); the records, flags and decisions are those of the pass taken one owner after the other
    (ddn_p25_rx_set_debug_flags bit 4096).  Channels whose frames end together and channels that lose their carrier make sure several lanes hunt at once."""
    B, n = 32, 30000
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        s = _traffic(300 + c // 4, n, [90.0, 5000.0][c % 2])          # four neighbours carry the same frames: they hunt together
        x[c, :len(s)] = s
    x[9, 12000:] = 0.0
    x[10, :9000] = 0.0
    outs = []
    for dbg, fil in ((0, False), (4096, False), (0, True)):   # (third run: the matched filter inside the loop, carrier losses incl.)
        rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=cpw, handlers=True, max_events=2048, filter_in_loop=fil, debug_flags=dbg)
        rec, fl, cnt = rx.run(x)
        outs.append((rec.copy(), fl.copy(), cnt.copy(), rx.events.copy(), rx.n_events.copy(), rx.event_data.copy()))
    for k in (1, 2):
        for a, b in zip(outs[0], outs[k]):
            assert np.array_equal(a, b)
    for c in (0, 9, 10, 31):
        _check(outs[0][0], outs[0][1], outs[0][2], outs[0][3], outs[0][4], c, _oracle(x[c], 1), outs[0][5])


@pytest.mark.parametrize("cpw", [4, 8])
def test_unfiltered_row_left_in_hbm_for_channels_in_sync(built, cpw):
    """The staging wave leaves the unfiltered samples of a half in HBM while the matched filters of its channels are on and warm; the
    recurrence lanes ask for them again ahead of time - 64 hunting symbols before the count that gates a filter off (no carrier) can
    be reached, while a filter is off, and through its next cold start (ddn_rx.hip, stage_half_load; bit 30 of tile_done).  Carriers
    that come and go at offsets spread over the 128-sample tile: same records, flags and decisions as with every tile staged whole
    (ddn_p25_rx_set_debug_flags bit 16384), and as the oracle's."""
    B, n = 8, 130000
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        a0, a1 = 37 * c, 20000 + 211 * c              # carrier, silence (no carrier after 1800 symbols without a sync) ...
        b0, b1 = 61000 + 173 * c, 84000 + 59 * c      # ... carrier again: cold start of the filter; and lost again
        s = _traffic(400 + c, a1 - a0, [90.0, 4000.0][c % 2])
        x[c, a0:a0 + len(s)] = s
        s = _traffic(420 + c, b1 - b0, [90.0, 4000.0][c % 2])
        x[c, b0:b0 + len(s)] = s
    x[3, 30000:50000] = np.random.default_rng(3).normal(0, 900, 20000).astype(np.float32)   # noise instead of silence
    outs = []
    for dbg in (0, 16384):
        rx = ddn.P25Rx(B, use_matched_filter=1, channels_per_wave=cpw, handlers=True, max_events=4096, debug_flags=dbg)
        rec, fl, cnt = rx.run(x)
        outs.append((rec.copy(), fl.copy(), cnt.copy(), rx.events.copy(), rx.n_events.copy(), rx.event_data.copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    for c in (0, 3, 5, 6):
        _check(outs[0][0], outs[0][1], outs[0][2], outs[0][3], outs[0][4], c, _oracle(x[c], 1), outs[0][5])
    for c in range(B):                                   # both carriers were received: decoded NIDs before and after the gap
        ev = outs[0][3][c, :outs[0][4][c]]
        at = ev[(ev[:, 1] == 1) & (ev[:, 2] > 0), 0]
        assert (at < 2200).sum() >= 1 and (at > 6000).sum() >= 1, (c, at[:4], at[-4:])


@pytest.mark.parametrize("fil", [0, 1])
def test_call_splits(built, fil):
    """decisions, history ring and handler words carried across calls (a block straddling a call boundary); fil = 1: with the filter
    inside the loop the carried filter memory is staged ahead of each call's first tile"""
    B = 6
    splits = [0, 5, 3000, 3001, 9000, 9640, 17000, 26000]
    x = np.zeros((B, splits[-1]), np.float32)
    for c in range(B):
        s = _traffic(200 + c, splits[-1], [100.0, 14000.0][c % 2], weak_nids=(c >= 4))
        x[c, :len(s)] = s
    want = [_oracle(x[c], 1) for c in range(B)]
    rx = ddn.P25Rx(B, use_matched_filter=1, handlers=True, max_events=1024, filter_in_loop=bool(fil))
    recs, fls, evs, evds = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)]
    base = np.zeros(B, np.int64)
    for a, b in zip(splits[:-1], splits[1:]):
        rec, fl, cnt = rx.run(x[:, a:b])
        for c in range(B):
            recs[c].append(rec[c, :cnt[c]])
            fls[c].append(fl[c, :cnt[c]])
            e = rx.events[c, :rx.n_events[c]].astype(np.int64)
            e[:, 0] += base[c]
            evs[c].append(e)
            evds[c].append(rx.event_data[c, :rx.n_events[c]])
            base[c] += cnt[c]
    for c in range(B):
        rec = np.concatenate(recs[c])[None]
        fl = np.concatenate(fls[c])[None]
        ev = np.concatenate(evs[c])[None]
        _check(rec, fl, np.array([rec.shape[1]]), ev, np.array([ev.shape[1]]), 0, want[c], np.concatenate(evds[c])[None])
