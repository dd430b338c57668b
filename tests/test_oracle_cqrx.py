"""The symbol-rate receive loop behind the CQPSK demodulator (oracle/ddn_oracle_cqrx.c):
 * its in-frame path (use_symbol + the CQPSK slice + rotation map + soft metrics) against the reference's own compiled dsd_dibit.c,
   driven through the metrics hooks the reference provides for exactly this (oracle/_ref, refh_cq_slicer_*);
 * the whole loop (hunting with the rotated-constellation retries, the handlers in frame) on the reference's P25 Phase 1 CQPSK captures:
   the known answers its own full-chain tests expect (tests/CMakeLists.txt:8901-8918)."""
import ctypes as C

import numpy as np
import pytest

import orc
from conftest import golden

P25P1_POS, P25P1_NEG, P25P2_POS, P25P2_NEG = 0, 1, 35, 36


def _symbols(seed, n, noise=0.35, offset=0.0):
    rng = np.random.default_rng(seed)
    lv = np.array([1.0, 3.0, -1.0, -3.0])[rng.integers(0, 4, n)]
    return (lv + noise * rng.standard_normal(n) + offset).astype(np.float32)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("synctype,map_idx,snr", [(P25P1_POS, 0, -100.0), (P25P1_NEG, 0, -100.0), (P25P1_POS, 2, 12.5), (P25P1_POS, 3, 30.0),
                                                   (P25P1_NEG, 4, 3.0), (P25P2_POS, 0, -100.0), (P25P2_NEG, 2, 18.0), (P25P1_POS, 1, -100.0)])
def test_in_frame_path_equals_the_compiled_dsd_dibit(synctype, map_idx, snr):
    r = orc.ref()
    r.refh_cq_slicer_create.restype = C.c_void_p
    r.refh_cq_slicer_create.argtypes = [C.c_int, C.c_int, C.c_double]
    r.refh_cq_slicer_destroy.argtypes = [C.c_void_p]
    r.refh_cq_slicer_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    # symbols with a DC offset (the centre has to follow), outliers and exact threshold hits
    sym = _symbols(synctype * 7 + map_idx, 2600, offset=0.2)
    sym[100:110] = [2.0, -2.0, 0.0, 7.5, -9.0, 1.9999999, -2.0000002, 4.0, 0.5, -0.5]
    h = r.refh_cq_slicer_create(synctype, map_idx, snr)
    want = np.zeros((len(sym), 4), np.int32)
    thr = np.zeros((len(sym), 5), np.float32)
    r.refh_cq_slicer_run(h, sym.ctypes.data, len(sym), want.ctypes.data, thr.ctypes.data)
    r.refh_cq_slicer_destroy(h)
    got, gthr = orc.oracle_cq_inframe(sym, map_idx, 1 if synctype in (P25P1_NEG, P25P2_NEG) else 0, snr)
    assert np.array_equal(got, want), np.flatnonzero((got != want).any(axis=1))[:5]
    assert np.array_equal(gthr.view(np.uint32), thr.view(np.uint32))
    assert len(np.unique(want[:, 0])) == 4 and want[:, 1].min() < 100 < want[:, 1].max()


def _capture_symbols(name):
    g = golden(name)
    x = ((g["iq"].astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    return orc.OracleCqpskFe(rate=48000).run(x, 8192)


def _blocks(events, kind):
    rows, data = events.rows(), events.data()
    return [(r, d) for r, d in zip(rows, data) if r[1] == kind]


def test_control_channel_capture_wacn_and_system():
    """DECODE_IQ_P25P1_CQPSK_CC expects "WACN: 92065; SYS: 0D5": symbols -> this loop with the handlers in frame -> the TSBK blocks the
    handlers decoded (list-8 half-rate + CRC16): every block CRC-clean, the Network Status Broadcasts carry that WACN / system id"""
    sym = _capture_symbols("iq_p25p1_cqpsk_cc.npz")
    rx = orc.OracleCqRx(orc.CQ_P25P1)
    rec, fl = rx.run(sym)
    assert int((fl & 2).astype(bool).sum()) >= 50                      # syncs
    nids = _blocks(rx.events, orc.HEV_P25_NID)
    assert len(nids) >= 50 and all(r[2] > 0 and r[4] == 7 for r, _ in nids[1:])    # every NID decodes, DUID 7 (TSDU)
    tsbk = _blocks(rx.events, orc.HEV_P25_TSBK)
    assert len(tsbk) >= 50 and all(d[3] & 1 for _, d in tsbk[1:])       # CRC16 good
    net = []
    for _, d in tsbk:
        by = np.array([d[0], d[1], d[2]], np.int32).view(np.uint8)
        if (by[0] & 0x3F) == 0x3B:
            net.append(((int(by[3]) << 12) | (int(by[4]) << 4) | (int(by[5]) >> 4), ((int(by[5]) & 0xF) << 8) | int(by[6])))
    assert len(net) >= 3 and all(v == (0x92065, 0x0D5) for v in net)


def test_simulcast_capture_decodes_the_grant_update():
    """DECODE_IQ_P25P1_CQPSK_SIMULCAST_CC expects "Group Voice Channel Grant Update - Implicit" (TSBK opcode 0x02)"""
    sym = _capture_symbols("iq_p25p1_cqpsk_cc_simulcast.npz")
    rx = orc.OracleCqRx(orc.CQ_P25P1)
    rx.run(sym)
    tsbk = _blocks(rx.events, orc.HEV_P25_TSBK)
    good = [np.array([d[0], d[1], d[2]], np.int32).view(np.uint8) for _, d in tsbk if d[3] & 1]
    assert len(good) >= 20
    assert any((by[0] & 0x3F) == 0x02 for by in good)


def test_voice_capture_frames():
    """DECODE_IQ_P25P1_CQPSK_VOICE ("Group Voice Channel User"): the loop follows the call - LDU1 / LDU2 alternate, every NID decodes,
    the in-frame lengths the handlers give put the next sync exactly where the next frame starts"""
    sym = _capture_symbols("iq_p25p1_cqpsk_vc.npz")
    rx = orc.OracleCqRx(orc.CQ_P25P1)
    rec, fl = rx.run(sym)
    nids = [r for r, _ in _blocks(rx.events, orc.HEV_P25_NID)]
    duids = [r[4] for r in nids if r[2] > 0]
    assert len(duids) == len(nids) >= 14 and all(r[3] == 0x106 for r in nids)      # every NID decodes, one NAC
    assert duids.count(15) >= 7 and duids.count(0) == 1 and duids.count(5) >= 3 and duids.count(10) >= 3   # TDULCs, HDU, LDU1 / LDU2
    gaps = list(np.diff(np.flatnonzero(fl & 2)))
    assert gaps.count(216) >= 7 and gaps.count(396) == 1 and gaps.count(864) >= 5  # TDULC, HDU, LDU frame lengths


def test_rotated_constellation_is_found_and_corrected():
    """a stream whose demodulator came up 90 degrees off (every dibit through the inverse of the N1200 map) syncs on the rotated
    retry, and the in-frame dibits come out corrected: the same dibits as the unrotated stream gives"""
    sym = _capture_symbols("iq_p25p1_cqpsk_cc.npz")
    a = orc.OracleCqRx(orc.CQ_P25P1)
    rec_a, fl_a = a.run(sym)
    # rotate: level of raw dibit q -> level of the raw dibit the map sends back to q.  N1200 = {1, 3, 0, 2}
    level = {0: 1.0, 1: 3.0, 2: -1.0, 3: -3.0}
    raw = np.where(sym >= 2, 1, np.where(sym >= 0, 0, np.where(sym >= -2, 2, 3)))
    inv = {1: 0, 3: 1, 0: 2, 2: 3}                                       # corrected -> raw under N1200
    resid = sym - np.vectorize(level.get)(raw)
    rot = (np.vectorize(lambda q: level[inv[q]])(raw) + resid).astype(np.float32)
    b = orc.OracleCqRx(orc.CQ_P25P1)
    rec_b, fl_b = b.run(rot)
    sa, sb = np.flatnonzero(fl_a & 2), np.flatnonzero(fl_b & 2)
    assert len(sb) >= len(sa) - 2 and np.all((fl_b[sb] >> 4) == 3)        # map index 3 = N1200
    inframe = (fl_a & 1).astype(bool) & (fl_b & 1).astype(bool)
    assert inframe.sum() > 5000 and np.mean(rec_a[inframe, 0] == rec_b[inframe, 0]) > 0.999
