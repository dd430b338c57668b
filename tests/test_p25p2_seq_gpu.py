"""GPU: ddn_p25p2_groups_batch (P25 Phase 2 above the bursts: processP2() on the 700 dibits behind a sync, batched over channels x
groups) against the whole-stream CPU restatement (tests/p2seq.py) - on the reference's own Phase 2 capture (the known SACCH MAC
PDUs), and on synthetic multi-channel traffic across call boundaries (carried scramble offset, 4V counters, ESS-B fragments)."""
import os

import numpy as np
import pytest

import ddn
import p2capture
import p2seq
from test_oracle_p25p2_capture import SACCH_OCTETS
from test_oracle_p25p2_seq import capture_groups

pytestmark = pytest.mark.gpu
FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))


def compare(got, want_rows, c):
    """got = P25P2Groups.run() arrays; want_rows = run_groups() dicts of channel c (rows in order)"""
    info, pay, fr, rel, ess = got
    G = info.shape[1]
    n = 0
    for k, w in enumerate(want_rows):
        g, ts = divmod(k, 4)
        i = info[c, g, ts]
        tag = (c, g, ts, w["action"])
        assert (i[0], i[1], i[2], i[3], i[4]) == (w["duid"], w["isch"], w["offset"], w["slot"], w["action"]), (tag, i.tolist())
        a = w["action"]
        if a in (p2seq.A_4V, p2seq.A_2V):
            assert i[6] == w["fourv"], tag
            assert np.array_equal(fr[c, g, ts], w["fr"]) and np.array_equal(rel[c, g, ts], w["rel"]), tag
            if a == p2seq.A_2V:
                assert i[5] == w["ec"] and bool(i[7] & 8) == bool(w["ess_ok"]) and np.array_equal(ess[c, g, ts], w["ess"]), (tag, i.tolist(), w["ec"])
        elif a in (p2seq.A_SACCH_S, p2seq.A_SACCH_C, p2seq.A_FACCH_C, p2seq.A_FACCH_S, p2seq.A_LCCH_C, p2seq.A_LCCH_S):
            assert i[5] == w["ec"] and (i[7] & 1) == w["used"] and bool(i[7] & 2) == bool(w["crc12"]), (tag, i.tolist(), w["ec"], w["used"])
            if a not in (p2seq.A_FACCH_C, p2seq.A_FACCH_S):
                assert bool(i[7] & 4) == bool(w["crc16"]), tag
            assert np.array_equal(pay[c, g, ts], w["payload"]), tag
        else:
            assert i[5] == 0 and i[7] == 0 and not pay[c, g, ts].any() and not fr[c, g, ts].any(), tag
        n += 1
    assert n == G * 4
    return n


def test_the_reference_capture_through_the_sequencing_stage(built):
    gb, gl, sf = capture_groups()
    seed = p2capture.WACN * 16777216 + p2capture.SYSID * 4096 + p2capture.NAC
    # the same capture as three channels: whole, and split over two calls at different places (the state is carried)
    want = p2seq.run_groups(gb, gl, p2capture.WACN, p2capture.SYSID, p2capture.NAC, p2seq.new_state())
    st = ddn.P25P2Groups([seed, seed, 0x1000])          # third channel: no valid site (wacn 0)
    three = lambda a: np.stack([a, a, a])
    got = st.run(three(gb), three(gl))
    compare(got, want, 0)
    compare(got, want, 1)
    no_site = p2seq.run_groups(gb, gl, 0, 1, 0, p2seq.new_state())
    compare(got, no_site, 2)
    info, pay = got[0], got[1]
    octets = [bytes(np.packbits(pay[0, g, ts])[:12]).hex() for g in range(len(gb)) for ts in range(4) if info[0, g, ts, 4] == p2seq.A_SACCH_S]
    assert octets == SACCH_OCTETS[:len(octets)] and len(octets) >= 9
    assert all(info[0, g, ts, 7] & 2 for g in range(len(gb)) for ts in range(4) if info[0, g, ts, 4] == p2seq.A_SACCH_S)
    st2 = ddn.P25P2Groups([seed])
    a = st2.run(gb[None, :7], gl[None, :7])
    b = st2.run(gb[None, 7:], gl[None, 7:])
    compare(a, want[:28], 0)
    compare(b, want[28:], 0)


def test_synthetic_channels_across_call_boundaries(built):
    rng = np.random.default_rng(11 + FZ)
    Cn, G = 9, 24
    sites = [(int(rng.integers(1, 0xFFFFF)), int(rng.integers(1, 0xFFF)), int(rng.integers(1, 0xFFF))) for _ in range(Cn)]
    sites[4] = (0xFFFFF, 5, 9)                        # all-ones WACN: no valid site
    streams = []
    for c, (w, s, n) in enumerate(sites):
        plan = None
        if c == 2:                                     # unknown DUIDs in pairs: groups end early, 4V counters restart
            plan = lambda k: "err" if k % 23 in (8, 9) or k % 31 == 5 else None
        if c == 3:                                     # a 2V burst with no 4V before it, 4V runs longer than four
            plan = lambda k: ("2v" if k % 14 == 0 else "4v") if k % 2 == 0 else None
        streams.append(p2seq.make_stream(rng, G, w, s, n, start_sf=int(rng.integers(0, 12)), noise=0.004 if c % 2 else 0.0, plan=plan))
    bits = np.stack([b for b, _ in streams])
    llr = np.stack([l for _, l in streams])
    seeds = [w * 16777216 + s * 4096 + n for (w, s, n) in sites]
    obj = ddn.P25P2Groups(seeds)
    states = [p2seq.new_state() for _ in range(Cn)]
    seen = set()
    for lo, hi in ((0, 5), (5, 6), (6, 17), (17, 24)):
        got = obj.run(bits[:, lo:hi], llr[:, lo:hi])
        for c, (w, s, n) in enumerate(sites):
            want = p2seq.run_groups(bits[c, lo:hi], llr[c, lo:hi], w, s, n, states[c])
            compare(got, want, c)
            seen |= {(r["action"], r["ec"] >= 0) for r in want}
    for a in range(1, 11):
        assert any(k[0] == a for k in seen), (a, seen)
    assert (p2seq.A_2V, False) in seen and (p2seq.A_SACCH_S, False) in seen, seen


def test_dibit_stream_to_mac_pdus_on_the_capture(built):
    """the capture's dibits -> sync cut -> groups -> bursts, all on the device.  The first S-ISCH of the capture (dibits 65..84) has one
    wrong dibit, so the exact test locks on the second (245..264) and the first groups start at 265 + 720 k, superframe slots 3, 7, 11:
    the SACCH burst of slot 11 opens its group and decodes (the known PDUs 1, 3, 5, ...: "P25p2 SACCH", what the reference's own
    DECODE_IQ_P25P2_CC test asks of this capture, tests/CMakeLists.txt:8923); the one of slot 10 is the fourth timeslot of a group the
    two unknown DUIDs before it already ended.  The same stream inverted, and handed over in two pieces."""
    dib, rel = p2capture.dibits()
    n = len(dib)
    llr2 = np.repeat(np.maximum(rel, 1)[:, None], 2, axis=1).astype(np.int16)
    seed = p2capture.WACN * 16777216 + p2capture.SYSID * 4096 + p2capture.NAC
    pos, wb, wl, wcur = p2seq.sync_cut(dib, llr2)
    assert pos[:3] == [265, 985, 1705] and len(pos) == 15
    obj = ddn.P25P2Groups([seed, seed])
    ng, gp, co, res, gb, gl = obj.run_stream(np.stack([dib, dib ^ 2]), np.stack([llr2, llr2]), max_groups=20)
    assert ng.tolist() == [15, 15] and gp[0, :15].tolist() == pos and gp[1, :15].tolist() == pos and co.tolist() == [wcur, wcur]
    assert np.array_equal(gb[0, :15], wb) and np.array_equal(gl[0, :15], wl) and np.array_equal(gb[1, :15], wb)
    assert np.array_equal(np.abs(gl[1, :15].astype(np.int32)), np.abs(wl.astype(np.int32)))
    want = p2seq.run_groups(wb, wl, p2capture.WACN, p2capture.SYSID, p2capture.NAC, p2seq.new_state())
    trimmed = tuple(a[:, :15] for a in res)
    compare(trimmed, want, 0)
    info, pay = res[0], res[1]
    assert (info[0, 15:, :, 4] == 0).all() and (info[0, 15:, :, 0] == -3).all()
    for c in (0, 1):
        octets = [bytes(np.packbits(pay[c, g, ts])[:12]).hex() for g in range(15) for ts in range(4) if info[c, g, ts, 4] == p2seq.A_SACCH_S]
        # (a damaged S-ISCH at dibit 4585 moves the lock by one timeslot: from the seventh group on both SACCH bursts open their group)
        assert octets == [SACCH_OCTETS[k] for k in (1, 3, 4, 5, 6, 7, 8, 9)], (c, octets)
        assert [int(info[c, g, 0, 2]) for g in range(15) if info[c, g, 0, 4] == p2seq.A_SACCH_S] == [11, 11, 10, 10, 10]
    # two pieces: the second call starts where the first one's cursor points
    cut = 6000
    o2 = ddn.P25P2Groups([seed])
    ng1, gp1, co1, r1, _, _ = o2.run_stream(dib[None, :cut], llr2[None, :cut], max_groups=20)
    p1, b1, l1, c1 = p2seq.sync_cut(dib[:cut], llr2[:cut])
    assert ng1[0] == len(p1) and gp1[0, :len(p1)].tolist() == p1 and co1[0] == c1
    ng2, gp2, co2, r2, _, _ = o2.run_stream(dib[None, c1:], llr2[None, c1:], max_groups=20)
    assert [int(v) + c1 for v in gp2[0, :ng2[0]]] == pos[len(p1):]
    compare(tuple(a[:, :ng1[0]] for a in r1), want[:4 * len(p1)], 0)
    compare(tuple(a[:, :ng2[0]] for a in r2), want[4 * len(p1):], 0)


def test_sync_cut_equals_oracle_on_planted_streams(built):
    import torch
    rng = np.random.default_rng(19 + FZ)
    Cn, n = 40, 5000
    dib = rng.integers(0, 4, (Cn, n)).astype(np.uint8)
    llr2 = rng.integers(-32768, 32767, (Cn, n, 2)).astype(np.int16)
    for c in range(Cn):
        for _ in range(int(rng.integers(0, 9))):
            at = int(rng.integers(0, n - 20))
            dib[c, at:at + 20] = p2seq.SYNC20 ^ (2 if rng.integers(0, 3) == 0 else 0)
        if c % 5 == 0:
            dib[c, n - 20:] = p2seq.SYNC20                    # a sync ending on the last dibit
        if c % 7 == 0:
            dib[c, :20] = p2seq.SYNC20                        # and one on the first 20
    cursor = [int(rng.integers(0, 30)) if c % 3 == 0 else 0 for c in range(Cn)]
    for mg in (8, 2):
        obj = ddn.P25P2Groups([0x123456789] * Cn)
        ng, gp, co, res, gb, gl = obj.run_stream(dib, llr2, cursor=cursor, max_groups=mg)
        for c in range(Cn):
            pos, wb, wl, wcur = p2seq.sync_cut(dib[c], llr2[c], cursor[c], mg)
            assert ng[c] == len(pos) and gp[c, :len(pos)].tolist() == pos and co[c] == wcur, (c, mg, ng[c], pos, co[c], wcur)
            assert np.array_equal(gb[c, :len(pos)], wb) and np.array_equal(gl[c, :len(pos)], wl), (c, mg)


def test_host_pointer_calls_on_the_capture(built):
    """ddn_p25p2_sync_cut_host -> ddn_p25p2_groups_host with plain host buffers (what a C caller does): the capture's dibits to the
    known SACCH PDUs, the carried state in the caller's memory across two calls"""
    import ctypes as C
    l = ddn.lib()
    dib, rel = p2capture.dibits()
    llr2 = np.ascontiguousarray(np.repeat(np.maximum(rel, 1)[:, None], 2, axis=1).astype(np.int16))
    dib = np.ascontiguousarray(dib)
    seed = np.array([p2capture.WACN * 16777216 + p2capture.SYSID * 4096 + p2capture.NAC], np.uint64)
    state = np.zeros(ddn.P25P2_SEQ_STATE_BYTES, np.uint8)
    octets, cursor, base = [], np.zeros(1, np.int32), 0
    for lo, hi in ((0, 6000), (None, len(dib))):
        lo = base if lo is None else lo
        d, q = np.ascontiguousarray(dib[lo:hi]), np.ascontiguousarray(llr2[lo:hi])
        mg = 12
        ng, gp, co = np.zeros(1, np.int32), np.zeros(mg, np.int32), np.zeros(1, np.int32)
        gb, gl = np.zeros((mg, 1400), np.uint8), np.zeros((mg, 1400), np.int16)
        assert l.ddn_p25p2_sync_cut_host(d.ctypes.data, q.ctypes.data, 1, len(d), len(d), None, mg, ng.ctypes.data, gp.ctypes.data, co.ctypes.data,
                                         gb.ctypes.data, gl.ctypes.data) == 0
        n = int(ng[0])
        info, pay = np.zeros((mg * 4, 8), np.int32), np.zeros((mg * 4, 180), np.uint8)
        fr, rl, ess = np.zeros((mg * 4, 384), np.uint8), np.zeros((mg * 4, 384), np.uint8), np.zeros((mg * 4, 96), np.uint8)
        assert l.ddn_p25p2_groups_host(gb.ctypes.data, gl.ctypes.data, 1, mg, ng.ctypes.data, seed.ctypes.data, state.ctypes.data, 64,
                                       info.ctypes.data, pay.ctypes.data, fr.ctypes.data, rl.ctypes.data, ess.ctypes.data) == 0
        octets += [bytes(np.packbits(pay[r])[:12]).hex() for r in range(4 * n) if info[r, 4] == p2seq.A_SACCH_S]
        assert (info[4 * n:, 4] == 0).all()
        base = lo + int(co[0])
    assert octets == [SACCH_OCTETS[k] for k in (1, 3, 4, 5, 6, 7, 8, 9)]
    assert int(state.view(np.int32)[0]) in (2, 6, 10)          # the carried scramble offset came back to the caller
