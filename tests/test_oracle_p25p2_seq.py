"""CPU: the Phase 2 sequencing restatement (tests/p2seq.py: ISCH -> scramble offset -> slot, de-scrambling, DUID dispatch, the
two-unknown-DUIDs rule, ESS assembly over 4V / 2V) on the reference's own Phase 2 capture, fed the way processP2() is - the 700
dibits behind every other S-ISCH (the sync the frame search stops on after the previous group's 700 dibits) - and on synthetic
traffic where what was sent is known."""
import numpy as np

import p2capture
import p2seq
from test_oracle_p25p2_capture import SACCH_OCTETS, expected_isch


def capture_groups():
    bits, llr, sf, hits = p2capture.timeslots()
    n = len(bits) // 4
    gb = bits[:4 * n].reshape(n, 1440)[:, :1400]
    gl = llr[:4 * n].reshape(n, 1440)[:, :1400]
    return gb, gl, sf


def test_capture_groups_give_the_known_sacch_pdus_through_the_sequencing():
    gb, gl, sf = capture_groups()
    st = p2seq.new_state()
    res = p2seq.run_groups(gb, gl, p2capture.WACN, p2capture.SYSID, p2capture.NAC, st)
    assert len(res) == 4 * len(gb) == 64
    got = []
    for i, r in enumerate(res):
        # every group of this capture carries the channel-1 I-ISCH that names its place: offset = superframe slot of its first timeslot
        assert r["offset"] == sf[i - i % 4], (i, r["offset"])
        if i % 4 != 3:
            assert r["isch"] == expected_isch(i), i
        if sf[i] >= 10:
            assert r["action"] == p2seq.A_SACCH_S and r["slot"] == sf[i] % 2 and r["ec"] == 11 and r["crc12"] == 1, (i, r["action"], r["ec"])
            got.append(bytes(np.packbits(r["payload"])[:12]).hex())
        elif i % 4 < 2 or sf[i - i % 4] >= 10:
            assert r["action"] == p2seq.A_ERR and r["duid"] == 10, (i, r["action"], r["duid"])   # " DUID ERR 10"
        else:
            assert r["action"] == p2seq.A_NONE                                                      # the group ended on its second error
    assert got == SACCH_OCTETS[:len(got)] and len(got) >= 9


def test_without_a_site_the_scrambled_bursts_are_skipped():
    gb, gl, sf = capture_groups()
    res = p2seq.run_groups(gb, gl, 0, p2capture.SYSID, p2capture.NAC, p2seq.new_state())
    assert all(r["action"] == p2seq.A_NOSITE for i, r in enumerate(res) if sf[i] >= 10)


def test_synthetic_traffic_decodes_to_what_was_sent():
    rng = np.random.default_rng(5)
    wacn, sysid, nac = 0xA441, 0x2D7, 0x4A1
    gb, gl = p2seq.make_stream(rng, 30, wacn, sysid, nac, start_sf=2)
    st = p2seq.new_state()
    res = p2seq.run_groups(gb[:13], gl[:13], wacn, sysid, nac, st) + p2seq.run_groups(gb[13:], gl[13:], wacn, sysid, nac, st)
    acts = [r["action"] for r in res]
    for a in (p2seq.A_4V, p2seq.A_2V, p2seq.A_SACCH_S, p2seq.A_SACCH_C, p2seq.A_FACCH_C, p2seq.A_FACCH_S, p2seq.A_LCCH_C, p2seq.A_LCCH_S,
              p2seq.A_ERR):
        assert acts.count(a) >= 2, (a, acts.count(a))
    # offsets follow the superframe: group g starts at slot (2 + 4 g) mod 12
    assert [r["offset"] % 12 for r in res[::4]] == [(2 + 4 * g) % 12 for g in range(30)]
    x = [r for r in res if r["action"] in (p2seq.A_SACCH_S, p2seq.A_SACCH_C, p2seq.A_FACCH_C, p2seq.A_FACCH_S, p2seq.A_LCCH_C, p2seq.A_LCCH_S)]
    assert sum(r["ec"] >= 0 for r in x) >= 0.6 * len(x)          # up to 11 bad symbols: beyond the FACCH's five
    v2 = [r for r in res if r["action"] == p2seq.A_2V]
    assert sum(r["ess_ok"] for r in v2) >= len(v2) - 2 and len(v2) >= 5
    assert any(r["fourv"] == 3 for r in res if r["action"] == p2seq.A_4V)


def test_two_unknown_duids_end_the_group_and_zero_the_4v_counters():
    rng = np.random.default_rng(6)
    wacn, sysid, nac = 0x12345, 0x111, 0x222
    plan = lambda k: "err" if k in (9, 11) else ("4v" if k % 2 == 0 else "facch_c")       # group 2 = slots 8..11: 4V err 4V err
    gb, gl = p2seq.make_stream(rng, 5, wacn, sysid, nac, start_sf=0, plan=plan)
    res = p2seq.run_groups(gb, gl, wacn, sysid, nac, p2seq.new_state())
    assert [r["action"] for r in res[8:12]] == [p2seq.A_4V, p2seq.A_ERR, p2seq.A_4V, p2seq.A_ERR]
    assert [r["fourv"] for r in res if r["action"] == p2seq.A_4V] == [0, 1, 2, 3, 0, 1, 0, 1, 2, 3]   # slots 0 2 4 6 8 10 | 12 14 16 18
    # the second error came last in its group; one more unknown DUID a slot earlier hides the rest of the group
    plan2 = lambda k: "err" if k in (8, 9) else ("4v" if k % 2 == 0 else "facch_c")
    gb, gl = p2seq.make_stream(rng, 4, wacn, sysid, nac, start_sf=0, plan=plan2)
    res = p2seq.run_groups(gb, gl, wacn, sysid, nac, p2seq.new_state())
    assert [r["action"] for r in res[8:12]] == [p2seq.A_ERR, p2seq.A_ERR, p2seq.A_NONE, p2seq.A_NONE]
    assert [r["duid"] for r in res[10:12]] == [-3, -3]


def test_sync_cut_on_the_capture_dibits():
    """frame_sync_try_p25p2()'s exact test over the capture's dibits: the first S-ISCH (one wrong dibit) is passed over, the lock moves
    by a timeslot where another one is damaged; the groups decode the known SACCH PDUs 1, 3, 4 .. 9"""
    dib, rel = p2capture.dibits()
    llr2 = np.repeat(np.maximum(rel, 1)[:, None], 2, axis=1).astype(np.int16)
    pos, gb, gl, cur = p2seq.sync_cut(dib, llr2)
    assert pos == [265, 985, 1705, 2425, 3145, 3865, 5125, 5845, 6565, 7285, 8005, 8725, 9445, 10165, 10885] and cur == 11585
    res = p2seq.run_groups(gb, gl, p2capture.WACN, p2capture.SYSID, p2capture.NAC, p2seq.new_state())
    got = [bytes(np.packbits(r["payload"])[:12]).hex() for r in res if r["action"] == p2seq.A_SACCH_S]
    assert got == [SACCH_OCTETS[k] for k in (1, 3, 4, 5, 6, 7, 8, 9)]
    # inverted polarity gives the same groups; a cursor past the first syncs skips them; a cut stream resumes at the cursor
    pos2, gb2, _, _ = p2seq.sync_cut(dib ^ 2, llr2)
    assert pos2 == pos and np.array_equal(gb2, gb)
    assert p2seq.sync_cut(dib, llr2, cursor=pos[0] + 700)[0] == pos[1:] and p2seq.sync_cut(dib, llr2, cursor=300)[0][0] == 805
    p1, _, _, c1 = p2seq.sync_cut(dib[:6000], llr2[:6000])
    p2, _, _, _ = p2seq.sync_cut(dib[c1:], llr2[c1:])
    assert p1 + [v + c1 for v in p2] == pos
