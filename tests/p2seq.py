"""Test scaffolding: P25 Phase 2 above the burst layer - what processP2() does with the 700 dibits behind a sync
(src/protocol/p25/phase2/p25p2_frame.c:1760-1798): the four ISCH words -> p2_scramble_offset (p25p2_process_isch(), :708-745), the
slot the group starts on (offset % 2), the de-scrambling (process_Frame_Scramble(), :372-392), and per timeslot the DUID and its
dispatch (p25p2_process_duid() / p25p2_duid_dispatch(), :1462-1478,1580-1640,1742-1760: 4V / 2V / SACCH / FACCH / LCCH, the
valid-site gate, two unknown DUIDs end the group and zero the 4V counters), ESS-B fragments filed under fourv_counter and decoded
with the 2V burst's ESS-A (:902-925,1377-1411,1435-1452).

run_groups() is the whole-stream CPU restatement of that sequencing for one channel, built on the oracle's burst-layer functions
(oracle/ddn_oracle_rs.c: orc_isch_lookup_soft, orc_p25p2_duid_lookup_soft, orc_p25p2_xcch, orc_p25p2_ess; each pinned to the
reference's compiled pieces in tests/test_oracle_p25p2_xcch.py).  p25p2_frame.c itself does not compile here (it needs the trunking
tree), so the sequencing is pinned by the reference's own Phase 2 capture: the ten SACCH MAC PDUs come out of run_groups() when it is
fed the groups behind each S-ISCH the way processP2() is (tests/test_oracle_p25p2_seq.py).

make_stream() builds synthetic traffic: superframes of twelve timeslots, the I-ISCH words of TIA-102.BBAC's layout, voice calls
(4V 4V 4V 4V 2V with a valid ESS spread over them), FACCH / SACCH / LCCH bursts clear and scrambled, unknown DUIDs."""
import ctypes as C

import numpy as np

import orc
import rs28
from test_oracle_isch import table as isch_table
from test_oracle_p25p2_xcch import oracle_duid, oracle_ess, oracle_xcch, scramble_bits, crc12_ok, crc16_ok

DUID_OFFSETS = [0, 1, 74, 75, 244, 245, 318, 319]
DUID_CANON = [0x00, 0x17, 0x2E, 0x39, 0x4B, 0x5C, 0x65, 0x72, 0x8D, 0x9A, 0xA3, 0xB4, 0xC6, 0xD1, 0xE8, 0xFF]
S_ISCH_WORD = 0x575D57F7FF
# action codes of a timeslot (include/ddn_hip.h: DDN_P2_*)
A_NONE, A_4V, A_2V, A_SACCH_S, A_SACCH_C, A_FACCH_C, A_FACCH_S, A_LCCH_C, A_LCCH_S, A_ERR, A_NOSITE = range(11)
ACTION_OF_DUID = {0: A_4V, 6: A_2V, 3: A_SACCH_S, 12: A_SACCH_C, 15: A_FACCH_C, 9: A_FACCH_S, 13: A_LCCH_C, 4: A_LCCH_S}
NEEDS_SITE = {A_4V, A_2V, A_SACCH_S, A_FACCH_S, A_LCCH_S}
VOICE_OFF = (2, 76, 172, 246)


def new_state():
    return {"offset": 0, "fourv": [0, 0], "ess_b": np.zeros((2, 96), np.uint8), "ess_b_llr": np.zeros((2, 96), np.int16)}


def site_valid(wacn, sysid, nac):
    """p25p2_duid_has_valid_site(), p25p2_frame.c:1455-1459"""
    return wacn != 0 and nac != 0 and sysid != 0 and wacn != 0xFFFFF and nac != 0xFFF and sysid != 0xFFF


def rows_of_group(bits1400, llr1400):
    """the group as four timeslot rows of 360; bits / metrics 1400..1439 were never captured (p2bit stays 0 there, the reliability
    helper answers 0 from 1400 on: p25p2_frame.c:155-163,354-370)"""
    b, l = np.zeros(1440, np.uint8), np.zeros(1440, np.int16)
    b[:1400], l[:1400] = bits1400, llr1400
    return b.reshape(4, 360), l.reshape(4, 360)


def isch_of(bits360, llr360):
    o = orc.oracle()
    o.orc_isch_lookup_soft.argtypes = [C.c_uint64, C.c_void_p]
    w = 0
    for k in range(40):
        w = (w << 1) | int(bits360[320 + k])
    if w == S_ISCH_WORD:
        return -2
    r40 = np.minimum(np.abs(llr360[320:360].astype(np.int32)), 255).astype(np.uint8)
    return o.orc_isch_lookup_soft(C.c_uint64(w), r40.ctypes.data)


def duid_of(bits360, llr360, threshold=64):
    w = 0
    for k in DUID_OFFSETS:
        w = (w << 1) | int(bits360[k])
    r8 = np.minimum(np.abs(llr360[DUID_OFFSETS].astype(np.int32)), 255).astype(np.uint8)
    return oracle_duid(w, r8, threshold)


def voice_frames(xb, xl, count):
    """p25p2_unpack_voice_frames() through the measured AMBE 2450 dibit map (tests/test_oracle_p25p2_xcch.py pins the schedule)"""
    import rx4
    m = np.asarray(rx4.ambe2450_map())
    fr, rl = np.zeros((4, 4, 24), np.uint8), np.zeros((4, 4, 24), np.uint8)
    for f, off in enumerate(VOICE_OFF[:count]):
        for x in range(72):
            row, col = (m[x // 2][0], m[x // 2][1]) if x % 2 == 0 else (m[x // 2][2], m[x // 2][3])
            fr[f, row, col] = xb[off + x]
            rl[f, row, col] = min(abs(int(xl[off + x])), 255)
    return fr, rl


def run_groups(groups_bits, groups_llr, wacn, sysid, nac, state, threshold=64):
    """-> list of per-timeslot dicts, four per group in order; state (new_state()) is carried across calls"""
    seq = scramble_bits(wacn, sysid, nac, 4320)
    two = np.concatenate([seq, seq, seq])
    valid = site_valid(wacn, sysid, nac)
    out = []
    for gb, gl in zip(groups_bits, groups_llr):
        rb, rl = rows_of_group(gb, gl)
        isch = [isch_of(rb[f], rl[f]) for f in range(4)]
        for f in range(4):
            v = isch[f]
            if v > -1 and ((v >> 5) & 3) == 1:
                loc = (v >> 3) & 3
                if loc == 0:
                    state["offset"] = 12 - f
                elif loc == 1:
                    state["offset"] = 4 - f
                elif loc == 2:
                    state["offset"] = 8 - f
        off = state["offset"]
        slot = off % 2
        errs, dead = 0, False
        for ts in range(4):
            r = {"isch": isch[ts], "offset": off, "duid": -3, "action": A_NONE, "slot": -1, "ec": 0, "used": 0, "crc12": 0, "crc16": 0,
                 "fourv": 0, "payload": np.zeros(180, np.uint8), "fr": np.zeros((4, 4, 24), np.uint8), "rel": np.zeros((4, 4, 24), np.uint8),
                 "ess": np.zeros(96, np.uint8), "ess_ok": 0}
            out.append(r)
            if dead:
                continue
            d = duid_of(rb[ts], rl[ts], threshold)
            r["duid"], r["slot"] = d, slot
            s0 = 20 + 360 * (off + ts)
            x = rb[ts] ^ two[s0:s0 + 360]
            xl = np.where(two[s0:s0 + 360] == 1, -rl[ts].astype(np.int32), rl[ts].astype(np.int32)).astype(np.int16)
            act = ACTION_OF_DUID.get(d, A_ERR)
            if act in NEEDS_SITE and not valid:
                act = A_NOSITE
            r["action"] = act
            if act == A_ERR:
                errs += 1
                if errs > 1:
                    state["fourv"] = [0, 0]
                    dead = True
                    continue
            elif act in (A_SACCH_S, A_SACCH_C, A_LCCH_C, A_LCCH_S, A_FACCH_C, A_FACCH_S):
                kind = 0 if act in (A_FACCH_C, A_FACCH_S) else 1
                scr = act in (A_SACCH_S, A_FACCH_S, A_LCCH_S)
                ec, pl, used = oracle_xcch(kind, x if scr else rb[ts], xl if scr else rl[ts], threshold)
                r["ec"], r["used"] = ec, used
                r["payload"][:len(pl)] = pl
                r["crc12"] = crc12_ok(pl, 144 if kind == 0 else 168)
                r["crc16"] = crc16_ok(pl) if kind == 1 else 0
            elif act == A_4V:
                r["fr"], r["rel"] = voice_frames(x, xl, 4)
                fv = state["fourv"][slot]
                r["fourv"] = fv
                if fv == 0:
                    state["ess_b"][slot][:] = 0
                    state["ess_b_llr"][slot][:] = 0
                state["ess_b"][slot][24 * fv:24 * fv + 24] = x[148:172]
                state["ess_b_llr"][slot][24 * fv:24 * fv + 24] = xl[148:172]
                state["fourv"][slot] = (fv + 1) & 3
            elif act == A_2V:
                r["fr"], r["rel"] = voice_frames(x, xl, 2)
                r["fourv"] = state["fourv"][slot]
                pa = np.concatenate([x[148:244], x[246:318]]).astype(np.uint8)
                pal = np.concatenate([xl[148:244], xl[246:318]]).astype(np.int16)
                acc, ec, pl = oracle_ess(state["ess_b"][slot].copy(), state["ess_b_llr"][slot].copy(), pa, pal, threshold)
                r["ess_ok"], r["ec"], r["ess"] = acc, ec, pl
                state["fourv"][slot] = 0
            slot ^= 1
    return out


# ---- traffic --------------------------------------------------------------------------------------------------------------------
def _put_duid(bits, d):
    w = DUID_CANON[d]
    bits[DUID_OFFSETS] = [(w >> (7 - k)) & 1 for k in range(8)]


def _word_bits(w):
    return np.array([(w >> (39 - k)) & 1 for k in range(40)], np.uint8)


def make_stream(rng, n_groups, wacn, sysid, nac, start_sf=0, noise=0.0, plan=None, voice=None):
    """-> (bits u8 [n_groups][1400], llr i16 [n_groups][1400]): a TDMA channel from superframe slot start_sf on.  Slot s of the
    superframe carries logical channel s % 2; channel 0 a voice call (4V 4V 4V 4V 2V, ESS valid), channel 1 signalling in turn
    (FACCH scrambled / clear, unknown DUIDs); slots 10 / 11 SACCH (scrambled / LCCH clear) - or plan(sf_index) -> kind.
    ISCH behind slot s: S-ISCH for s % 4 in (1, 2), I-ISCH channel 0 for s % 4 == 3, channel 1 for s % 4 == 0."""
    t = isch_table()
    seq = scramble_bits(wacn, sysid, nac, 4320)
    nts = 4 * n_groups
    bits = rng.integers(0, 2, (nts, 360)).astype(np.uint8)
    llr = (rng.integers(160, 240, (nts, 360)) * rng.choice([-1, 1], (nts, 360))).astype(np.int16)
    voice_k = 0
    ess = None
    for i in range(nts):
        sf = (start_sf + i) % 12
        kind = plan(start_sf + i) if plan else None
        if kind is None:
            if sf >= 10:
                kind = "sacch_s" if sf == 10 else "lcch_c"
            elif sf % 2 == 0:
                kind = "4v" if voice_k % 5 < 4 else "2v"
            else:
                kind = ("facch_s", "facch_c", "err", "sacch_c", "lcch_s")[(i // 2) % 5]
        body = rng.integers(0, 2, 360).astype(np.uint8)
        scr = True
        if kind in ("4v", "2v"):
            if voice_k % 5 == 0 or ess is None:
                ess = rs28.make_ess_case(rng, int(rng.integers(0, 8)), 0)
            pl, _, pa, _, _ = ess
            if kind == "4v":
                body[148:172] = pl[24 * (voice_k % 5 % 4):24 * (voice_k % 5 % 4) + 24]
                _put_duid(body, 0)
            else:
                body[148:244], body[246:318] = pa[:96], pa[96:]
                _put_duid(body, 6)
            if voice is not None:      # voice() -> the next AMBE 3600x2450 frame [4][24] of this logical channel, planted through the interleave
                import rx4
                m = np.asarray(rx4.ambe2450_map())
                for f in range(4 if kind == "4v" else 2):
                    fr = voice()
                    for x in range(72):
                        row, col = (m[x // 2][0], m[x // 2][1]) if x % 2 == 0 else (m[x // 2][2], m[x // 2][3])
                        body[VOICE_OFF[f] + x] = fr[row, col]
            voice_k += 1
        elif kind == "err":
            _put_duid(body, (1, 2, 5, 7, 8, 10, 11, 14)[int(rng.integers(0, 8))])
        else:
            xk = 0 if kind.startswith("facch") else 1
            b, l, _ = rs28.make_xcch_burst(rng, xk, int(rng.integers(0, 12)), int(rng.integers(0, 5)), int(rng.integers(0, 4)))
            body, llr[i] = b.astype(np.uint8), l
            _put_duid(body, {"sacch_s": 3, "sacch_c": 12, "facch_c": 15, "facch_s": 9, "lcch_c": 13, "lcch_s": 4}[kind])
            scr = kind.endswith("_s")
        if scr:
            keep = body[DUID_OFFSETS].copy()
            s0 = 20 + 360 * sf
            body = body ^ np.concatenate([seq, seq])[s0:s0 + 360]
            body[DUID_OFFSETS] = keep
            # (the metric's sign follows the transmitted bit; make_xcch_burst's weak positions stay weak)
        if sf % 4 in (1, 2):
            body[320:360] = _word_bits(S_ISCH_WORD)
        else:
            chan = 0 if sf % 4 == 3 else 1
            loc = ((sf + 1) // 4) % 3
            body[320:360] = _word_bits(t[(chan << 5) | (loc << 3) | int(rng.integers(0, 4))])
        bits[i] = body
    if noise > 0:
        flip = rng.random((nts, 360)) < noise
        bits ^= flip.astype(np.uint8)
        llr = np.where(flip, (llr.astype(np.int32) * rng.integers(0, 60, (nts, 360)) // 240), llr).astype(np.int16)
    return bits.reshape(n_groups, 1440)[:, :1400].copy(), llr.reshape(n_groups, 1440)[:, :1400].copy()


# ---- the dibit-level sync cut ----------------------------------------------------------------------------------------------------
SYNC20 = np.array([1, 1, 1, 3, 1, 1, 3, 1, 1, 1, 1, 3, 3, 3, 1, 3, 3, 3, 3, 3], np.uint8)      # P25P2_SYNC, sync_patterns.h:36


def sync_cut(dibits, llr2, cursor=0, max_groups=1 << 30):
    """frame_sync_try_p25p2()'s exact 20-dibit test (dsd_frame_sync.c:800-816) over a dibit stream, the window filling from `cursor`;
    behind a sync processP2() takes 700 dibits and the search starts again -> (positions, bits [g][1400], llr [g][1400], cursor_out)"""
    n = len(dibits)
    pos, gb, gl = [], [], []
    start, out = max(cursor, 0), None
    while out is None:
        found, inv = -1, False
        for e in range(start + 19, n):
            w = dibits[e - 19:e + 1]
            if np.array_equal(w, SYNC20) or np.array_equal(w, SYNC20 ^ 2):
                found, inv = e, not np.array_equal(w, SYNC20)
                break
        if found < 0:
            out = max(start, n - 19)
        elif found + 700 >= n or len(pos) >= max_groups:
            out = found - 19
        else:
            d = dibits[found + 1:found + 701] ^ (2 if inv else 0)
            l = llr2[found + 1:found + 701].astype(np.int32).copy()
            if inv:
                l[:, 0] = np.where(l[:, 0] == -32768, 32767, -l[:, 0])
            b = np.zeros(1400, np.uint8)
            b[0::2], b[1::2] = d >> 1, d & 1
            pos.append(found + 1)
            gb.append(b)
            gl.append(l.reshape(1400).astype(np.int16))
            start = found + 701
    return pos, np.array(gb, np.uint8).reshape(-1, 1400), np.array(gl, np.int16).reshape(-1, 1400), min(out, n)
