"""GPU: ddn_cq_rx (the symbol-rate receive loop behind the CQPSK demodulator) against its CPU oracle (oracle/ddn_oracle_cqrx.c, in-frame
path pinned to the compiled dsd_dibit.c): records (dibit, reliability, LLRs, symbol), flags, the handlers' events and payloads - bit
for bit, across ragged call boundaries, on synthetic P25 Phase 1 traffic, rotated constellations, inverted polarity, Phase 2 syncs,
and on the reference's own CQPSK captures from cu8 I/Q with nothing but device kernels in between."""
import numpy as np
import pytest

import ddn
import orc
import p25gen
from conftest import golden

pytestmark = pytest.mark.gpu

LEVEL = np.array([1.0, 3.0, -1.0, -3.0], np.float32)      # raw dibit -> symbol level
INV = {0: {c: r for r, c in enumerate([0, 1, 2, 3])}, 2: {c: r for r, c in enumerate([3, 2, 1, 0])},
       3: {c: r for r, c in enumerate([1, 3, 0, 2])}, 4: {c: r for r, c in enumerate([2, 0, 3, 1])}}


def symbols_of(dibits, rng, noise=0.25, map_idx=0, invert=False, offset=0.0):
    d = np.asarray(dibits, np.int64)
    if invert:
        d = d ^ 2
    raw = np.vectorize(INV[map_idx].get)(d)
    return (LEVEL[raw] + noise * rng.standard_normal(len(d)) + offset).astype(np.float32)


def p1_traffic(rng, kind):
    nac = 0x293
    if kind == "ctrl":
        parts = [p25gen.make_frames(rng, 1, nac, crc=True, blocks=1 + k % 3)[0] for k in range(14)]
    elif kind == "voice":
        import mbe
        frames = np.stack([mbe.imbe_encode(b) for b in mbe.random_imbe_bits(rng, (18,))])
        parts = [p25gen.make_hdu(rng, nac)[0], p25gen.make_ldus(rng, 2, nac, frames)[0], p25gen.make_tdulc(rng, nac)[0], p25gen.make_tdu(nac)]
    else:
        parts = [p25gen.make_frames(rng, 1, nac, crc=True, blocks=2)[0], p25gen.make_pdu_coded(rng, nac, blks=3)[0],
                 p25gen.make_tdu(nac), p25gen.make_frames(rng, 1, nac, crc=k_bad(rng), blocks=1)[0]]
    gap = lambda: rng.integers(0, 4, int(rng.integers(5, 60)))
    out = [rng.integers(0, 4, 40)]
    for p in parts:
        out += [np.asarray(p) & 3, gap()]
    return np.concatenate(out)


def k_bad(rng):
    return bool(rng.integers(0, 2))


def run_oracle(sym, protocol, snr=-100.0):
    rx = orc.OracleCqRx(protocol, -1 if protocol == orc.CQ_P25P1 else 700, snr)
    rec, fl = rx.run(sym)
    rows, data = rx.events.rows(), rx.events.data()
    return rec, fl, [(r, d) for r, d in zip(rows, data)]


def check_channel(got_rec, got_fl, got_ev, sym, want_rec, want_fl, want_ev, tag):
    n = len(sym)
    assert np.array_equal(got_fl[:n] & 0x7F, want_fl), (tag, np.flatnonzero((got_fl[:n] & 0x7F) != want_fl)[:5])
    assert np.array_equal(got_rec[:n, 0].astype(np.int32), want_rec[:, 0]), (tag, "dibit")
    assert np.array_equal(got_rec[:n, 1].astype(np.int32), want_rec[:, 1]), (tag, "reliability")
    llr = got_rec[:n, 2:6].copy().view(np.int16).reshape(n, 2).astype(np.int32)
    assert np.array_equal(llr, want_rec[:, 2:4]), (tag, "llr")
    assert np.array_equal(got_rec[:n, 6:10].copy().view(np.uint32).reshape(-1), sym.view(np.uint32)), (tag, "symbol")
    # events: oracle (pos, kind, a, b, c) + data4 against the device's packed words (ddn_p25_rx_set_events)
    assert len(got_ev) == len(want_ev), (tag, len(got_ev), len(want_ev))
    for (pos, kind, a, b, d4), ((wp, wk, wa, wb, wc), wd) in zip(got_ev, want_ev):
        assert (pos, kind) == (wp, wk), (tag, pos, kind, wp, wk)
        assert np.array_equal(d4, wd), (tag, pos, kind, d4, wd)
        if kind == 1:
            assert a == wa and (b & 0xFFFF) == (wb & 0xFFFF) and ((b >> 16) & 0xFF) == (wc & 0xFF), (tag, "nid", a, b, wa, wb, wc)


def run_gpu_in_calls(streams, protocol, cuts, snr=0.0):
    """streams: list of f32 arrays (one per channel); cuts: call boundaries as fractions -> per channel (rec, flags, events with
    stream-wide positions)"""
    B = len(streams)
    rx = ddn.CqRx(B, protocol, 0, snr)
    recs, fls, evs = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)]
    done = [0] * B
    for f in list(cuts) + [1.0]:
        upto = [int(round(len(s) * f)) for s in streams]
        take = [u - d for u, d in zip(upto, done)]
        n = max(max(take), 1)
        blk = np.zeros((B, n), np.float32)
        for c in range(B):
            blk[c, :take[c]] = streams[c][done[c]:upto[c]]
        rec, fl, cnt, ev = rx.run(blk, take)
        for c in range(B):
            assert cnt[c] == take[c]
            recs[c].append(rec[c, :take[c]])
            fls[c].append(fl[c, :take[c]])
            evs[c] += [(p + done[c], k, a, b, d4) for (p, k, a, b, d4) in ev[c]]
        done = upto
    rx.close()
    return [np.concatenate(r) for r in recs], [np.concatenate(f) for f in fls], evs


@pytest.mark.parametrize("seed", [1, 2])
def test_p25p1_synthetic_traffic_bit_exact_across_calls(built, seed):
    rng = np.random.default_rng(100 + seed)
    streams, tags = [], []
    for kind in ("ctrl", "voice", "data"):
        for map_idx, inv in ((0, False), (0, True), (3, False), (2, True), (4, False)):
            d = p1_traffic(rng, kind)
            streams.append(symbols_of(d, rng, noise=0.22 + 0.1 * rng.random(), map_idx=map_idx, invert=inv, offset=0.15 * rng.standard_normal()))
            tags.append((kind, map_idx, inv))
    streams.append(symbols_of(rng.integers(0, 4, 4500), rng))            # no sync at all: the 1800-symbol timeout, twice
    tags.append(("noise", 0, False))
    rec, fl, ev = run_gpu_in_calls(streams, ddn.CQ_P25P1, cuts=(0.13, 0.5, 0.505, 0.81))
    n_sync = 0
    for c, s in enumerate(streams):
        wr, wf, we = run_oracle(s, orc.CQ_P25P1)
        check_channel(rec[c], fl[c], ev[c], s, wr, wf, we, tags[c])
        n_sync += int(((wf & 2) != 0).sum())
        if tags[c][0] != "noise" and tags[c][1] in (0, 2):
            assert ((wf & 2) != 0).sum() >= 3, tags[c]                      # (N1200 / P1200 need the level fit: most syncs still land)
    assert n_sync > 60


def test_snr_weight_and_phase2_lock(built):
    rng = np.random.default_rng(7)
    p2 = np.array([int(c) for c in "11131131111333133333"])
    streams = []
    for inv in (False, True, False):
        parts = [rng.integers(0, 4, 30)]
        for _ in range(6):
            parts += [p2, rng.integers(0, 4, 700), rng.integers(0, 4, int(rng.integers(0, 25)))]
        streams.append(symbols_of(np.concatenate(parts), rng, noise=0.3, invert=inv, map_idx=2 if inv else 0))
    rec, fl, ev = run_gpu_in_calls(streams, ddn.CQ_P25P2, cuts=(0.3, 0.77), snr=14.0)
    for c, s in enumerate(streams):
        wr, wf, we = run_oracle(s, orc.CQ_P25P2, snr=14.0)
        check_channel(rec[c], fl[c], ev[c], s, wr, wf, we, ("p2", c))
        assert ((wf & 2) != 0).sum() >= 5 and ((wf & 1) != 0).sum() >= 5 * 700


@pytest.mark.parametrize("name", ["iq_p25p1_cqpsk_cc.npz", "iq_p25p1_cqpsk_vc.npz", "iq_p25p1_cqpsk_cc_simulcast.npz"])
def test_reference_captures_from_iq_on_the_device(built, name):
    """cu8 I/Q of the reference's P25 Phase 1 CQPSK captures -> ddn_cqpsk_run -> ddn_cq_rx, device kernels only: equal to the oracle
    chain, and the known answers of the reference's own full-chain tests come out of the event payloads"""
    g = golden(name)
    iq = np.ascontiguousarray(g["iq"])
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    want_sym = orc.OracleCqpskFe(rate=48000).run(x, 8192)
    fe = ddn.CqpskBatch(1, rate=48000, block_len=8192, input_format=ddn.IN_CU8)
    sym, cnt = fe.run(iq[None])
    assert cnt[0] == len(want_sym) and np.array_equal(sym[0, :cnt[0]].view(np.uint32), want_sym.view(np.uint32))
    s = np.ascontiguousarray(sym[0, :cnt[0]])
    rec, fl, ev = run_gpu_in_calls([s], ddn.CQ_P25P1, cuts=(0.4,))
    wr, wf, we = run_oracle(s, orc.CQ_P25P1)
    check_channel(rec[0], fl[0], ev[0], s, wr, wf, we, name)
    tsbk = [np.asarray(d4[:3], np.int32).view(np.uint8) for (_, k, _, _, d4) in ev[0] if k == 2 and (d4[3] & 1)]
    nids = [d4 for (_, k, _, _, d4) in ev[0] if k == 1]
    if "vc" in name:
        duids = [int(d[2]) for d in nids if d[0] > 0]
        assert duids.count(5) >= 3 and duids.count(10) >= 3 and duids.count(15) >= 7 and all(int(d[1]) == 0x106 for d in nids)
    elif "simulcast" in name:
        assert len(tsbk) >= 20 and any((b[0] & 0x3F) == 0x02 for b in tsbk)        # Group Voice Channel Grant Update
    else:
        net = [b for b in tsbk if (b[0] & 0x3F) == 0x3B]
        assert len(tsbk) >= 50 and len(net) >= 3
        for b in net:
            assert ((int(b[3]) << 12) | (int(b[4]) << 4) | (int(b[5]) >> 4), ((int(b[5]) & 0xF) << 8) | int(b[6])) == (0x92065, 0x0D5)
