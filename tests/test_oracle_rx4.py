"""The profile-driven receive-loop restatement (oracle/ddn_oracle_rx4.c) on the CPU: with the P25p1 profile it must equal
the pinned P25p1 loop (oracle/ddn_oracle_rx.c) symbol for symbol; the DMR / NXDN48 profiles are exercised on the
reference's own regression captures."""
import numpy as np
import pytest

import orc
import rx4
from conftest import golden


def _p25_stream(seed, n_ldus=3):
    import p25gen
    rng = np.random.default_rng(seed)
    dib = p25gen.make_ldus(rng, n_ldus, 0x293, None)[0] if hasattr(p25gen, "make_ldus") else None
    return dib


@pytest.mark.parametrize("cap", ["iq_p25p1_c4fm_cc.npz", "iq_p25p1_c4fm_vc.npz"])
@pytest.mark.parametrize("lock,flt", [(840, 1), (345, 1), (0, 0), (1700, 1)])
def test_p25_profile_equals_the_pinned_p25_loop(built, cap, lock, flt):
    g = golden(cap)
    disc = orc.OracleFrontEnd().run_cu8(np.ascontiguousarray(g["iq"], np.uint8), 8192)
    a = orc.OracleP25Rx(lock_symbols=lock, use_filter=flt)
    b = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_P25P1, use_filter=flt, lock=[lock, 0, 0, 0]))
    # two calls with a ragged split: carried state must agree as well
    cut = 50001
    for part in (disc[:cut], disc[cut:]):
        sa, ra, fa = a.run(part)
        out = b.run(part)
        assert np.array_equal(sa.view(np.uint32), out["sym"].view(np.uint32))
        assert np.array_equal(ra, out["rec4"])
        assert np.array_equal(fa, out["fl"] & 7)
        assert np.array_equal(a.thresholds().view(np.uint32), b.thresholds().view(np.uint32))
    assert (out["fl"] & 2).sum() > 0 or lock == 0


def test_p25_profile_noise_and_carrier_loss(built):
    rng = np.random.default_rng(4)
    g = golden("iq_p25p1_c4fm_cc.npz")
    disc = orc.OracleFrontEnd().run_cu8(np.ascontiguousarray(g["iq"], np.uint8), 8192)
    x = np.concatenate([disc[:30000], (rng.standard_normal(40000) * 900).astype(np.float32), -disc[:40000]])
    a = orc.OracleP25Rx(lock_symbols=300, use_filter=1)
    b = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_P25P1, lock=[300, 0, 0, 0]))
    sa, ra, fa = a.run(x)
    out = b.run(x)
    assert np.array_equal(sa.view(np.uint32), out["sym"].view(np.uint32)) and np.array_equal(ra, out["rec4"])
    assert np.array_equal(fa, out["fl"] & 7)
    assert (fa & 4).any() and (fa & 2).sum() > 10


def _dmr_bursts(out, inverted):
    """slot types / info words of every data-class sync whose 54 live dibits are inside the stream"""
    import fec3
    res = []
    cls_data = 1 if inverted else 0          # with -xr the voice word marks a data burst
    for k, pos in enumerate(out["sync_pos"]):
        if out["sync_pat"][k] != cls_data or pos + 55 > len(out["pay"]):
            continue
        st, info, cach = rx4.dmr_burst_fields(out["pre"][k], out["rec4"][pos + 1:pos + 55, 0], inverted)
        items, _, ok = fec3.oracle_decode(5, st[None, :].copy())
        o96, r3, errs = fec3.oracle_bptc(info[None, :].copy(), 1)
        res.append(dict(ok=bool(ok[0]), cc=rx4.bits_int(items[0][:4]), dt=rx4.bits_int(items[0][4:8]), pdu=o96[0], errs=int(errs[0])))
    return res


@pytest.mark.parametrize("rf_mod", [0, 2])
def test_dmr_ras_control_channel_known_answers(built, rf_mod):
    """The reference's DMR Tier III RAS control-channel capture through front end (12.5 kHz profile) -> DMR receive loop ->
    slot type Golay(20,8) -> BPTC(196,96): every burst is a CSBK under colour code 0 - DECODE_IQ_DMR_T3_RAS_CC_COLOR_CODE
    asserts "Color Code=00" - and the C_ALOHA PDUs carry the system identity DECODE_IQ_DMR_T3_RAS_CC asserts:
    "C_ALOHA_SYS_PARMS: Large; Net ID: 1; Site ID: 1" (tests/CMakeLists.txt:8936-8947; field layout
    src/protocol/dmr/dmr_csbk.c:2698-2735,2876-2890; C_ALOHA = CSBKO 25, system identity code in bits 40..53)."""
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=rf_mod)).run(disc)
    acc = out["sync_pos"]
    assert len(acc) >= 60 and np.all(out["sync_pat"] == 0) and np.all(np.diff(acc)[1:] == 144)
    bursts = _dmr_bursts(out, 0)[1:]          # the first sync has no 90-dibit history behind it
    assert len(bursts) >= 60
    assert all(b["ok"] and b["cc"] == 0 and b["dt"] == 3 and b["errs"] == 0 for b in bursts)
    aloha = [b["pdu"] for b in bursts if rx4.bits_int(b["pdu"][2:8]) == 25]
    assert len(aloha) >= 10
    for pdu in aloha:
        # dmr_syscode_decode_model(), src/protocol/dmr/dmr_csbk.c:2698-2735: model 2 = "Large", net = bits 42..45, site = 46..53
        assert rx4.bits_int(pdu[40:42]) == 2
        assert rx4.bits_int(pdu[42:46]) == 1 and rx4.bits_int(pdu[46:54]) == 1


@pytest.mark.parametrize("cap", ["iq_dmr_t3_cc.npz", "iq_dmr_voice.npz"])
def test_dmr_inverted_captures_decode_clean_under_xr(built, cap):
    """These two captures are discriminator audio of inverted polarity as this chain (pinned to the reference's front end)
    sees it: only the BS voice word matches.  Under the reference's -xr rules (opts->inverted_dmr: voice word = data burst,
    digitize() un-inverts, cached dibits ^= 2) every burst decodes clean: Golay(20,8) slot types under one colour code,
    BPTC(196,96) without residual errors and the CSBK CRC-CCITT (mask 0xA5A5) on every CSBK: the channel's colour code is 1.
    The reference's suite asserts "Color Code=02" on both with plain -fs (tests/CMakeLists.txt:8925-8930): that reading - the
    voice handler decoding the sync word's own dibits as an EMB - is reproduced with the handlers in the loop, see
    tests/test_oracle_handlers.py::test_dmr_plain_fs_reading_prints_color_code_02 and DESIGN.md 5g."""
    disc = rx4.capture_disc(cap, 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=0, inverted=1)).run(disc)
    assert len(out["sync_pos"]) >= 60 and np.all(out["sync_pat"] == 1)
    bursts = _dmr_bursts(out, 1)[1:]
    good = [b for b in bursts if b["ok"]]
    assert len(good) >= len(bursts) - 4
    assert len({b["cc"] for b in good}) == 1
    csbk = [b for b in good if b["dt"] == 3]
    assert len(csbk) >= 30
    for b in csbk:
        crc = (~rx4.crc_ccitt_bits(b["pdu"][:80])) & 0xFFFF
        assert b["errs"] == 0 and (crc ^ rx4.bits_int(b["pdu"][80:96])) == 0xA5A5


def test_nxdn48_capture_frame_sync_cadence(built):
    """NXDN48 capture (2400 baud, 20 samples/symbol, 6.25 kHz channel filter): frame sync words 192 symbols apart, accepted
    from the second one on (a first match only becomes lastsynctype, dsd_frame_sync.c:1547-1556)."""
    disc = rx4.capture_disc("iq_nxdn48.npz", 1)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_NXDN48)).run(disc)
    acc = out["sync_pos"]
    assert len(acc) >= 20
    d = np.diff(acc)
    assert np.mean(d == 192) > 0.8


def test_nxdn48_known_answer_src_901(built):
    """The reference's NXDN48 capture through front end (6.25 kHz profile) -> NXDN48 receive loop -> de-scramble -> SACCH
    de-interleave / de-puncture -> K=5 decoder (soft, then the hard-decision retry the reference falls back to) -> CRC6 ->
    four-part superframes: the VCALL messages carry source unit 901 - the string DECODE_IQ_NXDN48 asserts ("Src=901",
    tests/CMakeLists.txt:8948; VCALL layout: message type in bits 2..7, source unit ID in bits 24..39)."""
    import fecgen
    disc = rx4.capture_disc("iq_nxdn48.npz", 1)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_NXDN48)).run(disc)
    rows, lich_ok, n_fallback = [], 0, 0
    for pos in out["sync_pos"]:
        if pos + 183 > len(out["sym"]):
            break
        lich, pok, ss, sr, fs, fr = rx4.nxdn_frame_fields(out["rec4"][pos + 1:pos + 183, 0], out["rec4"][pos + 1:pos + 183, 1])
        lich_ok += int(pok)
        dec, _ = fecgen.oracle_nxdn(ss[None].copy(), sr[None].copy(), 36, 32)
        t = np.unpackbits(dec[0])[:32]
        if not rx4.nxdn_crc_ok(t, 0):       # nxdn_hard_fallback_decode(): trellis_decode on the de-punctured hard bits
            t = rx4.oracle_trellis_decode((ss.reshape(-1) >> 1)[None], 32)[0]
            n_fallback += 1
        if rx4.nxdn_crc_ok(t, 0):
            rows.append(t)
    assert lich_ok >= 55 and len(rows) >= 50 and n_fallback > 0
    msgs = rx4.nxdn_superframes(rows)
    vcall = [m for ran, m in msgs if rx4.bits_int(m[2:8]) == 1]
    assert len(vcall) >= 4 and all(ran == 1 for ran, _ in msgs)
    assert all(rx4.bits_int(m[24:40]) == 901 for m in vcall)


@pytest.mark.skipif(not orc.have_ref(), reason="needs oracle/_ref")
def test_trellis_decode_restatement_equals_reference(built):
    """orc_trellis_decode == the compiled trellis_decode (src/core/util/dsd_misc.c:24-71) on clean, noisy and random inputs"""
    import ctypes as C
    r = orc.ref()
    r.trellis_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    r.trellis_decode.restype = None
    rng = np.random.default_rng(12)
    for ln in (32, 92, 5):
        src = rng.integers(0, 2, (300, 2 * ln + 6), dtype=np.uint8)
        # a third of the rows: valid code words with a few flips (register starts at 0, generators 0x19 / 0x17)
        for i in range(0, 300, 3):
            bits = rng.integers(0, 2, ln + 3, dtype=np.uint8)
            reg, enc = 0, []
            for b in bits:
                reg = ((reg << 1) | int(b)) & 0x1F
                enc += [bin(reg & 0x19).count("1") & 1, bin(reg & 0x17).count("1") & 1]
            src[i] = np.array(enc, np.uint8)
            src[i, rng.choice(2 * ln, int(rng.integers(0, 4)), replace=False)] ^= 1
        got = rx4.oracle_trellis_decode(src, ln)
        for i in range(300):
            want = np.zeros(ln, np.uint8)
            s_i = np.ascontiguousarray(src[i])
            r.trellis_decode(want.ctypes.data, s_i.ctypes.data, ln)
            assert np.array_equal(got[i], want), (ln, i)


def test_window_and_slip_rules_against_the_reference_held_answers(built):
    """The loop's two per-modulation rule functions against the expectations the reference's own test holds for
    select_window_* and symbol_adjust_timing_index (tests/dsp/test_dsp_symbol_replay.c:397-443)."""
    import ctypes as C
    o = orc.oracle()
    o.orc_fsk4_adjust_timing.argtypes = [C.c_int] * 7 + [C.c_void_p]
    o.orc_fsk4_window.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    o.orc_fsk4_window.restype = None

    def window(rf_mod, narrow):
        l, r = C.c_int(0), C.c_int(0)
        o.orc_fsk4_window(rf_mod, narrow, C.byref(l), C.byref(r))
        return l.value, r.value

    assert window(0, 1) == (1, 2)        # C4FM with a YSF / DMR type held: left edge 1
    assert window(0, 0) == (2, 2)
    assert window(1, 0) == (1, 2)        # QPSK
    assert window(2, 0) == (1, 1)        # GFSK

    def adjust(sps, centre, rf_mod, jitter, have_sync, span, start_i):
        ja = C.c_int(99)
        return o.orc_fsk4_adjust_timing(sps, centre, rf_mod, jitter, have_sync, span, start_i, C.byref(ja)), ja.value

    assert adjust(20, 9, 0, 8, 0, 20, 0) == (-1, -1)
    assert adjust(20, 9, 0, 12, 0, 20, 0) == (1, -1)
    assert adjust(10, 4, 1, 3, 0, 10, 0)[0] == 1 and adjust(10, 4, 1, 7, 0, 10, 0)[0] == -1
    assert adjust(10, 4, 2, 4, 0, 10, 0)[0] == -1 and adjust(10, 4, 2, 6, 0, 10, 0)[0] == 1
    assert adjust(10, 4, 0, 4, 0, 10, 0)[0] == -1 and adjust(10, 4, 0, 6, 0, 10, 0)[0] == 1
    assert adjust(10, 4, 0, 4, 1, 10, 0) == (0, 4)       # in sync: no slip, latch kept
    assert adjust(10, 4, 0, 4, 0, 1, 0) == (0, 4)        # one-sample span
    assert adjust(10, 4, 0, 4, 0, 10, 1) == (1, 4)       # not at the symbol's first sample


def test_nxdn96_known_answer_ran_00(built):
    """The reference's NXDN96 capture (4800 baud; DECODE_IQ_NXDN96 asserts "RAN 00", tests/CMakeLists.txt:8949) through front end
    (12.5 kHz profile) -> the receive loop on the NXDN96 profile (level ring 24, the DMR matched filter, NXDN's sync words and two-match
    rule) -> de-scramble -> SACCH de-interleave / de-puncture -> K=5 decode -> CRC6: frames 192 symbols apart, the SACCH parts count
    3, 2, 1, 0 through the superframe and every one carries Radio Access Number 0"""
    import fecgen
    disc = rx4.capture_disc("iq_nxdn96.npz", 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_NXDN96, rf_mod=2)).run(disc)
    acc = out["sync_pos"]
    assert len(acc) >= 40 and np.mean(np.diff(acc) == 192) > 0.8
    rows, lich_ok = [], 0
    for pos in acc:
        if pos + 183 > len(out["sym"]):
            break
        lich, pok, ss, sr, fs, fr = rx4.nxdn_frame_fields(out["rec4"][pos + 1:pos + 183, 0], out["rec4"][pos + 1:pos + 183, 1])
        lich_ok += int(pok)
        dec, _ = fecgen.oracle_nxdn(ss[None].copy(), sr[None].copy(), 36, 32)
        t = np.unpackbits(dec[0])[:32]
        if not rx4.nxdn_crc_ok(t, 0):
            t = rx4.oracle_trellis_decode((ss.reshape(-1) >> 1)[None], 32)[0]
        if rx4.nxdn_crc_ok(t, 0):
            rows.append(t)
    assert lich_ok >= 38 and len(rows) >= 34
    assert all(rx4.bits_int(t[2:8]) == 0 for t in rows)                       # RAN 00
    parts = [rx4.bits_int(t[0:2]) for t in rows]
    assert sum(1 for a, b in zip(parts, parts[1:]) if (a - b) % 4 == 1) >= len(parts) - 6
