"""The protocol handlers' hold on the receive loop (oracle/ddn_oracle_handlers.c): how many symbols a frame is read in
frame is decided by what the handler decodes - the NID's DUID, a TSDU block's last-block flag, a DMR burst's TACT / slot
type / EMB, an NXDN LICH - exactly as the reference's handlers pull dibits.  Anchors: the reference's full-chain known
answers on its own captures (tests/CMakeLists.txt:8888-8948), with no handler length configured by the caller."""
import numpy as np
import pytest

import orc
import rx4
from conftest import golden


def _p25(name):
    g = golden(name)
    disc = orc.OracleFrontEnd().run_cu8(g["iq"], 8192)
    rx = orc.OracleP25Rx(lock_symbols=-1, use_filter=1)
    sym, rec4, fl = rx.run(disc)
    return g, disc, sym, rec4, fl, rx.events.rows()


def test_p25_control_capture_tsdu_blocks_until_the_last_block_flag(built):
    """p25p1_c4fm_cc: every TSDU is three half-rate blocks; processTSBK reads on until the decoded block's last-block flag
    (p25p1_tsbk.c:1055-1071).  Every block of every TSDU passes its CRC16, the flag sits on block 2, the NAC is 0x140
    (DECODE_IQ_P25P1_C4FM_CC) - and the loop stays in frame for 33 + 3 x 101 symbols, then hunts for 24 + 0."""
    g, _, sym, rec4, fl, ev = _p25("iq_p25p1_c4fm_cc.npz")
    nid = [e for e in ev if e[1] == orc.HEV_P25_NID]
    assert nid[0][2] <= 0 and nid[0][4] == 0xFF            # the first sync: the matched filter switches on under the NID
    good = nid[1:]
    want_nac = int(bytes(g["expected_nac_hex"]).decode(), 16)
    assert len(good) >= 24 and all(e[2] == 1 and e[3] == want_nac and e[4] == 7 for e in good)
    tsbk = [e for e in ev if e[1] == orc.HEV_P25_TSBK]
    assert len(tsbk) >= 72 and all(e[3] & 1 for e in tsbk)              # CRC16 good on every block
    assert all((e[4] >> 8) == (1 if e[2] == 2 else 0) for e in tsbk)      # last-block flag on the third block only
    assert all((e[4] & 0xFF) == 0 for e in tsbk)                          # the best-metric candidate is the CRC-clean one
    acc = np.flatnonzero(fl & 2)
    assert np.all(np.diff(acc)[1:] == 360)
    for a in acc[1:-1]:
        assert np.all(fl[a + 1:a + 1 + 336] & 1) and not np.any(fl[a + 337:a + 360] & 1)


def test_p25_voice_capture_frame_lengths_by_duid(built):
    """p25p1_c4fm_vc: HDU-length gap, then LDU2 / LDU1 alternate 864 symbols apart, each read in frame for 33 + 807."""
    _, _, sym, rec4, fl, ev = _p25("iq_p25p1_c4fm_vc.npz")
    nid = [e for e in ev if e[1] == orc.HEV_P25_NID][1:]
    assert len(nid) >= 8 and all(e[2] == 1 for e in nid) and len({e[3] for e in nid}) == 1
    d = [e[4] for e in nid]
    assert set(d) == {5, 10} and all(a != b for a, b in zip(d, d[1:]))
    acc = np.flatnonzero(fl & 2)
    assert np.all(np.diff(acc)[1:] == 864)
    for a in acc[1:-1]:
        assert np.all(fl[a + 1:a + 841] & 1) and not np.any(fl[a + 841:a + 864] & 1)


@pytest.mark.parametrize("name", ["iq_p25p1_c4fm_cc.npz", "iq_p25p1_c4fm_vc.npz"])
def test_p25_profile_loop_with_handlers_equals_the_p25_loop(built, name):
    g, disc, sym, rec4, fl, ev = _p25(name)
    rx = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_P25P1, handler=1))
    out = rx.run(disc)
    assert np.array_equal(out["sym"].view(np.uint32), sym.view(np.uint32)) and np.array_equal(out["rec4"], rec4)
    assert np.array_equal(out["fl"] & 7, fl & 7) and rx.events.rows() == ev


def test_p25_frame_lengths_of_every_duid(built):
    """synthetic stream with one frame of every DUID the dispatcher knows (dispatch_p25p1.c:391-403): in-frame symbol
    counts after the sync are 33 + {HDU 339, TDU 15, TDULC 159, LDU 807, TSDU 101 per block} and 33 for an undefined DUID"""
    import p25gen
    rng = np.random.default_rng(5)
    want = []
    dib = []
    for duid, body in ((0x0, 339), (0x3, 15), (0xF, 159), (0x5, 807), (0xA, 807), (0x9, 0)):
        d = p25gen.frame_with_duid(rng, 0x293, duid, body + 40)
        dib.append(d)
        want.append(33 + body)
    d3, _ = p25gen.make_frames(rng, 2, 0x293, crc=True, blocks=3)
    d1, _ = p25gen.make_frames(rng, 2, 0x293, crc=True, blocks=1)
    want += [33 + 303] * 2 + [33 + 101] * 2
    x = p25gen.modulate_disc(np.concatenate(dib + [d3, d1]), lead=400, noise=60.0, seed=9)
    rx = orc.OracleP25Rx(lock_symbols=-1, use_filter=0)
    sym, rec4, fl = rx.run(x)
    acc = np.flatnonzero(fl & 2)
    assert len(acc) == len(want)
    got = []
    for a in acc:
        k = a + 1
        while k < len(fl) and (fl[k] & 1) and not (fl[k] & 2):
            k += 1
        got.append(k - a - 1)
    assert got == want


def test_dmr_plain_fs_reading_prints_color_code_02(built):
    """DECODE_IQ_DMR_VOICE / DECODE_IQ_DMR_T3_CC (tests/CMakeLists.txt:8925-8930) assert "Color Code=02" under plain -fs.
    Both captures are discriminator audio of inverted polarity: only the BS *voice* word ever matches, so the reference
    runs dmrBSBootstrap() / dmrBS() on them (dsd_frame_sync.c:1235-1253 with inverted_dmr = 0, dispatch_dmr.c:80-101).
    The TACT survives the inversion (its seven bits are the high bits of seven dibits; the all-ones word is a Hamming(7,4)
    code word), every burst carries the voice word, and check_dmr_bs_emb_and_confidence() (dmr_bs.c:337-385) runs
    QR(16,7,6) on the word's own first and last four dibits as if they were an EMB: once the frozen thresholds put two of
    those sync dibits in the inner region the "EMB" corrects to colour code 2, two such bursts on a slot lock the colour-code
    gate (dmr_confidence.c:93-105,120-148) and every later burst prints "Color Code=02" (dmr_bs.c:580-581)."""
    for cap in ("iq_dmr_t3_cc.npz", "iq_dmr_voice.npz"):
        disc = rx4.capture_disc(cap, 2)
        rx = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=2, handler=1))
        out = rx.run(disc)
        ev = rx.events.rows()
        assert np.all(out["sync_pat"] == 1)                          # BS voice word only
        printed = [e[2] for e in ev if e[1] == orc.HEV_DMR_CC_PRINT]
        assert printed.count(2) >= 4, (cap, printed)
        if cap == "iq_dmr_t3_cc.npz":
            # the colour code 2 is read off bursts whose "EMB" is the voice sync word itself: after the second sync every
            # burst of this control channel carries the word, and the bursts that lock the gate decode it to 2
            vb = [e for e in ev if e[1] == orc.HEV_DMR_VOICE_BURST and e[0] > out["sync_pos"][1]]
            assert all(e[4] & 1 for e in vb) and {e[3] for e in vb} == {25, 2}


def test_dmr_ras_capture_prints_color_code_00_through_the_data_handler(built):
    """dmr_t3_ras_cc under -fs: BS data word -> dmr_data_sync() from the dispatcher: TACT, slot type Golay(20,8) after five
    live dibits, colour-code gate locked by the second CSBK, "Color Code=00" on every dispatched burst
    (DECODE_IQ_DMR_T3_RAS_CC_COLOR_CODE), 54 live + 66 skipped symbols in frame per burst."""
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)
    rx = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=2, handler=1))
    out = rx.run(disc)
    ev = rx.events.rows()
    assert np.all(out["sync_pat"] == 0) and len(out["sync_pos"]) >= 60
    printed = [e[2] for e in ev if e[1] == orc.HEV_DMR_CC_PRINT]
    assert len(printed) >= 58 and set(printed) == {0}
    data = [e for e in ev if e[1] == orc.HEV_DMR_DATA]
    assert all(e[2] == 1 and e[3] == 0 and (e[4] & 0xFF) == 3 for e in data[1:])
    fl = out["fl"]
    for a in out["sync_pos"][2:-1]:
        assert np.all(fl[a + 1:a + 121] & 1) and not (fl[a + 121] & 1)


def test_nxdn48_lich_gate(built):
    """nxdn_frame(): 8 LICH dibits, then 174 more only when the LICH's parity and profile check out
    (nxdn_frame.c:592-604); the capture's known answer (Src=901) is unchanged by it"""
    disc = rx4.capture_disc("iq_nxdn48.npz", 1)
    rx = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_NXDN48, handler=1))
    out = rx.run(disc)
    ev = [e for e in rx.events.rows() if e[1] == orc.HEV_NXDN_LICH]
    assert len(ev) == len(out["sync_pos"]) and sum(e[2] for e in ev) >= 55
    fl = out["fl"]
    for (pos, _, ok, lich, par), a in zip(ev, out["sync_pos"]):
        assert pos == a + 8
        if a + 200 < len(fl):
            n_in = 0
            while fl[a + 1 + n_in] & 1 and not fl[a + 1 + n_in] & 2:
                n_in += 1
            assert n_in == (182 if ok else 8)


# ---- P25 data-unit header fallbacks (p25_mpdu_finalize_header) ------------------------------------------------------------------------
# The reference's own known answers (tests/protocol/p25/test_p25_p1_mdpu_helpers.c:160-165,463-521: three CRC16-clean headers; "best
# rep", "majority rebuilt", "combined header" cases).  Its test feeds the static function through stubs (fake CRC words, a stubbed list
# decoder); here the same three outcomes are reached through real inputs - the CRC16 is computed, the list decoder runs.
HDR_A = bytes([0x00, 0x00, 0x00, 0x10, 0x0A, 0x11, 0x11, 0x00, 0x01, 0x01, 0xAE, 0x8E])
HDR_B = bytes([0x80, 0x00, 0x00, 0x10, 0x0A, 0x22, 0x22, 0x00, 0x02, 0x02, 0x7A, 0x83])
HDR_C = bytes([0xBB, 0x00, 0x00, 0xAB, 0xCD, 0xE1, 0x23, 0x81, 0x23, 0x00, 0x51, 0x97])


def _finalize(reps, llr, n):
    import ctypes as C
    o = orc.oracle()
    o.orc_p25_mpdu_finalize_header.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rb = np.ascontiguousarray(np.array([list(r) for r in reps], np.uint8))
    ll = np.ascontiguousarray(llr, np.int16)
    out = np.zeros(12, np.uint8)
    how = o.orc_p25_mpdu_finalize_header(rb.ctypes.data, ll.ctypes.data, n, out.ctypes.data)
    return how, bytes(out.tolist())


def _llr_of(bytes12, amp=600):
    import p25gen
    d = np.asarray(p25gen.encode_half_rate(list(bytes12)), np.int64)
    l = np.zeros(196, np.int64)
    l[0::2] = (2 * ((d >> 1) & 1) - 1) * amp
    l[1::2] = (2 * (d & 1) - 1) * amp
    return l


def test_mpdu_header_reference_vectors_are_crc_clean():
    import p25gen
    for h in (HDR_A, HDR_B, HDR_C):
        assert p25gen.crc16_ccitt(np.frombuffer(h[:10], np.uint8)) == (h[10] << 8 | h[11])


def test_mpdu_finalize_header_best_repetition():
    """repetition 1 is the only one whose CRC16 holds (reference test: hdr_rep_crc {7, 0, 9}) -> it is the header"""
    junk = bytes(range(0xA0, 0xAC))
    how, out = _finalize([junk, HDR_A, junk], np.zeros((3, 196)), 3)
    assert (how, out) == (1, HDR_A)
    how, out = _finalize([HDR_B, HDR_A, junk], np.zeros((3, 196)), 3)       # the FIRST clean one
    assert (how, out) == (0, HDR_B)
    how, out = _finalize([junk, HDR_A, junk], np.zeros((3, 196)), 1)        # the header announced data: only block 0 counts (:382-383)
    assert how == (64 | 16) and out == junk


def test_mpdu_finalize_header_majority():
    """no repetition is clean and the summed LLRs decode to nothing clean: two of three, bit by bit (reference test: three copies of
    header C -> header C, CRC 0); here each copy has its own bit error"""
    reps = []
    for k, (byte, bit) in enumerate(((0, 0x80), (5, 0x01), (11, 0x10))):
        b = bytearray(HDR_C)
        b[byte] ^= bit
        reps.append(bytes(b))
    how, out = _finalize(reps, np.zeros((3, 196)), 3)
    assert (how, out) == (64, HDR_C)
    reps[1] = reps[0]                                                        # two copies share the error: it wins, CRC16 stays bad
    how, out = _finalize(reps, np.zeros((3, 196)), 3)
    assert how == (64 | 16) and out == reps[0]


def test_mpdu_finalize_header_combined_llrs():
    """no repetition decodes clean on its own, the position-wise sum of the three does (reference test: stubbed candidates {junk, header B}
    -> header B, soft-combined counter 1).  Each repetition is header B with a different third of the block wiped out with strong wrong
    values: the decoded bytes are whatever they are, the sum is right two to one everywhere"""
    good = _llr_of(HDR_B)
    llr = np.stack([good, good, good])
    for k in range(3):
        llr[k, 64 * k:64 * k + 64] *= -1
    import chain_stream
    reps = []
    for k in range(3):
        by, ok, _ = chain_stream.oracle_tsbk(llr[k])
        assert not ok
        reps.append(bytes(by.tolist()))
    how, out = _finalize(reps, llr, 3)
    assert (how, out) == (32, HDR_B)
    # saturation (saturating_llr_add :36-45): 3 x 32767 stays 32767 and still decodes
    how, out = _finalize(reps, np.clip(llr * 1000, -32768, 32767), 3)
    assert (how, out) == (32, HDR_B)
