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
