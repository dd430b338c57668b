"""The reference's own P25p1 C4FM regression captures through the restated chain (front end -> receive loop -> NID).

The reference's full-chain test on p25p1_c4fm_cc asserts the decoded payload field "NAC/CC: 140"
(tests/CMakeLists.txt:8888-8894).  dsd_symbol.c / dsd_frame_sync.c cannot be compiled here, so this known answer is
what anchors the unpinned symbolizer / sync restatement on real signals: the chain must lock on every frame and the
BCH-protected NAC it reads must be 0x140 with no bit errors.  The voice capture must show alternating LDU1 / LDU2
frames 864 symbols apart under one NAC."""
import numpy as np
import pytest

import orc
from conftest import golden
from test_oracle_block import oracle_nid


def nids_from_records(rec4, fl, count):
    acc = np.flatnonzero(fl[:count] & 2)
    rows = []
    keep = [k for k in range(33) if k != 11]
    for a in acc:
        if a + 34 > count:
            break
        nd = rec4[a + 1:a + 34][keep]
        bits = np.stack([(nd[:, 0] >> 1) & 1, nd[:, 0] & 1], axis=1).reshape(64).astype(np.uint8)
        rel = np.minimum(np.abs(np.stack([nd[:, 2], nd[:, 3]], axis=1)), 255).reshape(64).astype(np.uint8)
        rows.append((int(a), bits, rel))
    return rows


def decode_nids(rows, fn):
    bits = np.ascontiguousarray(np.stack([r[1][:63] for r in rows]))
    rel = np.ascontiguousarray(np.stack([r[2][:63] for r in rows]))
    par = np.array([r[1][63] for r in rows], np.uint8)
    prel = np.array([r[2][63] for r in rows], np.uint8)
    return fn(bits, rel, np.zeros(len(rows), np.int32), par, prel)


def oracle_chain(iq, lock):
    disc = orc.OracleFrontEnd().run_cu8(iq, 8192)
    sym, rec4, fl = orc.OracleP25Rx(lock_symbols=lock, use_filter=1).run(disc)
    return disc, sym, rec4, fl


def test_control_channel_capture_decodes_reference_nac(built):
    g = golden("iq_p25p1_c4fm_cc.npz")
    want_nac = int(bytes(g["expected_nac_hex"]).decode(), 16)
    _, sym, rec4, fl = oracle_chain(g["iq"], 156)
    rows = nids_from_records(rec4, fl, len(sym))
    out = decode_nids(rows, oracle_nid)
    good = out[1:]                                   # the first hit is a false sync before the threshold warm start
    assert len(good) >= 24
    assert np.all(good[:, 0] == 1) and np.all(good[:, 1] == want_nac) and np.all(good[:, 2] == 7)
    assert np.all(good[:, 3] == 0)                   # no BCH corrections needed on a clean capture
    assert np.all(np.abs(np.diff([r[0] for r in rows[1:]]) - 360) <= 1)    # 3-block TSDUs, 360 symbols apart


def test_voice_capture_alternates_ldu1_ldu2(built):
    g = golden("iq_p25p1_c4fm_vc.npz")
    _, sym, rec4, fl = oracle_chain(g["iq"], 840)
    rows = nids_from_records(rec4, fl, len(sym))
    out = decode_nids(rows, oracle_nid)[1:]
    assert len(out) >= 8 and np.all(out[:, 0] == 1) and len(set(out[:, 1])) == 1 and np.all(out[:, 3] == 0)
    assert list(out[:, 2]) == [10, 5] * (len(out) // 2) + [10] * (len(out) % 2)
    assert np.all(np.diff([r[0] for r in rows[1:]]) == 864)


@pytest.mark.gpu
@pytest.mark.parametrize("name,lock", [("iq_p25p1_c4fm_cc.npz", 156), ("iq_p25p1_c4fm_vc.npz", 840)])
def test_real_capture_gpu_equals_oracle(built, name, lock):
    import ddn
    g = golden(name)
    iq = np.ascontiguousarray(g["iq"])
    disc_o, sym_o, rec_o, fl_o = oracle_chain(iq, lock)
    n = iq.shape[0]
    disc = ddn.Batch(1, block_len=8192).run_host(iq[None], n)
    assert np.array_equal(disc[0].view(np.uint32), disc_o.view(np.uint32))
    rec, fl, cnt = ddn.P25Rx(1, lock_symbols=lock, use_matched_filter=1).run(disc)
    k = int(cnt[0])
    r4, sy = orc.unpack_records10(rec[0, :k])
    assert k == len(sym_o) and np.array_equal(sy.view(np.uint32), sym_o.view(np.uint32))
    assert np.array_equal(r4, rec_o) and np.array_equal(fl[0, :k], fl_o)

    def gpu_nid(bits, rel, obs, par, prel):
        out = np.zeros((len(bits), 4), np.int32)
        assert ddn.lib().ddn_p25p1_nid_decode_host(bits.ctypes.data, rel.ctypes.data, obs.ctypes.data, par.ctypes.data,
                                                   prel.ctypes.data, 64, len(bits), out.ctypes.data) == 0
        return out

    rows = nids_from_records(r4, fl[0], k)
    assert np.array_equal(decode_nids(rows, gpu_nid), decode_nids(rows, oracle_nid))


# ---- CQPSK control-channel capture: the reference's full-chain test expects "WACN: 92065; SYS: 0D5" ----------------
FS = np.array([int(c) for c in "111113113311333313133333"])


def crc16_tsbk_ok(by12):
    """TSBK CRC: CCITT x^16+x^12+x^5+1 over the first 80 bits, inverted, in the last two bytes (TIA-102.AABB)."""
    bits = np.unpackbits(np.asarray(by12, np.uint8))
    reg = 0
    for b in bits[:80]:
        fb = ((reg >> 15) & 1) ^ int(b)
        reg = (reg << 1) & 0xFFFF
        if fb:
            reg ^= 0x1021
    return (reg ^ 0xFFFF) == ((int(by12[10]) << 8) | int(by12[11]))


def cqpsk_tsbk_inputs(sym):
    """symbols -> fixed 4-level slice (frame_sync_slice_cqpsk_dibit, src/dsp/dsd_frame_sync.c:2076-2088, centre 0) ->
    frame syncs -> per TSDU the 98 coded dibits as hard LLR pairs (+-100)."""
    import p25gen
    d = np.where(sym >= 2, 1, np.where(sym >= 0, 0, np.where(sym >= -2, 2, 3))).astype(np.int64)
    hits = [i for i in range(len(d) - 180) if np.array_equal(d[i:i + 24], FS)]
    bp = np.array(p25gen.block_positions())
    llr = np.zeros((len(hits), 196), np.int16)
    for k, h in enumerate(hits):
        blk = d[h + bp]
        llr[k, 0::2] = np.where((blk >> 1) & 1, 100, -100)
        llr[k, 1::2] = np.where(blk & 1, 100, -100)
    return hits, llr


def check_cqpsk_payload(hits, blocks):
    assert len(hits) >= 50 and np.all(np.diff(hits) % 180 == 0)       # TSDUs of one or two blocks
    assert all(crc16_tsbk_ok(b) for b in blocks)
    net = [b for b in blocks if (int(b[0]) & 0x3F) == 0x3B]          # Network Status Broadcast
    assert len(net) >= 3
    for b in net:
        wacn = (int(b[3]) << 12) | (int(b[4]) << 4) | (int(b[5]) >> 4)
        sysid = ((int(b[5]) & 0xF) << 8) | int(b[6])
        assert (wacn, sysid) == (0x92065, 0x0D5)


def test_cqpsk_control_channel_capture_decodes_reference_wacn_sysid(built):
    import fecgen
    g = golden("iq_p25p1_cqpsk_cc.npz")
    x = ((g["iq"].astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    sym = orc.OracleCqpskFe(rate=48000).run(x, 8192)
    hits, llr = cqpsk_tsbk_inputs(sym)
    blocks, _ = fecgen.oracle_p25_half_rate(llr)
    check_cqpsk_payload(hits, blocks)


@pytest.mark.gpu
def test_cqpsk_real_capture_gpu(built):
    import ddn
    g = golden("iq_p25p1_cqpsk_cc.npz")
    iq = np.ascontiguousarray(g["iq"])
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    want = orc.OracleCqpskFe(rate=48000).run(x, 8192)
    b = ddn.CqpskBatch(1, rate=48000, block_len=8192, input_format=ddn.IN_CU8)
    sym, cnt = b.run(iq[None])
    assert cnt[0] == len(want) and np.array_equal(sym[0, :cnt[0]].view(np.uint32), want.view(np.uint32))
    hits, llr = cqpsk_tsbk_inputs(sym[0, :cnt[0]])
    out = np.zeros((len(hits), 12), np.uint8)
    met = np.zeros(len(hits), np.int32)
    assert ddn.lib().ddn_fec_p25_12_soft_host(llr.ctypes.data, len(hits), out.ctypes.data, met.ctypes.data) == 0
    check_cqpsk_payload(hits, out)


# ---- voice capture: LDU1 link control through Hamming(10,6,3) + RS(24,12,13) -----------------------------------------
def ldu1_words(rec4, fl, count):
    """-> for every LDU1 of the capture: (bits10 [24,10] (12 data words then 12 parity words), reliab [24,10])."""
    import p25gen
    dpos, ppos = p25gen.ldu1_positions()
    pos = np.concatenate([dpos, ppos]) - 24            # relative to the first dibit after the sync
    rows = nids_from_records(rec4, fl, count)
    nid = decode_nids(rows, oracle_nid)
    out = []
    for (a, _, _), n in zip(rows, nid):
        if n[0] != 1 or n[2] != 5 or a + 1 + 840 > count:
            continue
        d = rec4[a + 1:a + 1 + 840]
        w = d[pos]                                      # [24, 5, 4]
        bits = np.stack([(w[:, :, 0] >> 1) & 1, w[:, :, 0] & 1], axis=2).reshape(24, 10).astype(np.uint8)
        rel = np.stack([np.abs(w[:, :, 2]), np.abs(w[:, :, 3])], axis=2).reshape(24, 10).astype(np.int32)
        out.append((bits, rel))
    return out


def link_control(words, hamming_hard, rs_decode):
    """The reference's order: Hamming per hex word (p25p1_ldu.c:189-221), then RS(24,12,13) over the 12 + 12 words
    (p25p1_ldu1.c:233-245).  Returns the 72 link-control bits, most significant hex word first."""
    bits, _ = words
    fixed, errs = hamming_hard(np.ascontiguousarray(bits))
    assert np.all(errs < 2)                             # clean capture: nothing the hard decoder cannot fix
    data = np.ascontiguousarray(fixed[:12, :6].reshape(1, 12, 6))
    par = np.ascontiguousarray(fixed[12:, :6].reshape(1, 12, 6))
    out, rc = rs_decode(data, par)
    assert rc[0] == 0
    return out[0][::-1].reshape(72)                     # hex_data[11] is the first word on the air


def _oracle_hamming(bits):
    import ctypes as C
    o = orc.oracle()
    o.orc_hamming_10_6_3.argtypes = [C.c_int, C.c_void_p]
    out = bits.copy()
    errs = np.zeros(len(bits), np.int32)
    for i in range(len(bits)):
        w = 0
        for b in bits[i]:
            w = (w << 1) | int(b)
        f = C.c_int(0)
        errs[i] = o.orc_hamming_10_6_3(w, C.byref(f))
        if errs[i] == 1:
            out[i, :6] = [(f.value >> (5 - k)) & 1 for k in range(6)]
    return out, errs


def _check_lc(lcs):
    assert len(lcs) >= 4
    for lc in lcs:
        lcf = int("".join(map(str, lc[:8])), 2)
        mfid = int("".join(map(str, lc[8:16])), 2)
        assert lcf == 0x00 and mfid == 0x00            # "Group Voice Channel User" (TIA-102.AABF LCO 0)
    tg = {int("".join(map(str, lc[32:48])), 2) for lc in lcs}
    src = {int("".join(map(str, lc[48:72])), 2) for lc in lcs}
    assert len(tg) == 1 and len(src) == 1 and tg != {0} and src != {0}    # one call: same talkgroup and source throughout


def test_voice_capture_link_control_is_group_voice_channel_user(built):
    from test_oracle_rs import oracle_rs
    g = golden("iq_p25p1_c4fm_vc.npz")
    _, sym, rec4, fl = oracle_chain(g["iq"], 840)
    lcs = [link_control(w, _oracle_hamming, lambda d, p: oracle_rs("24_12_13", d, p)) for w in ldu1_words(rec4, fl, len(sym))]
    _check_lc(lcs)


@pytest.mark.gpu
def test_voice_capture_link_control_gpu(built):
    import ddn
    g = golden("iq_p25p1_c4fm_vc.npz")
    iq = np.ascontiguousarray(g["iq"])
    disc = ddn.Batch(1, block_len=8192).run_host(iq[None], iq.shape[0])
    rec, fl, cnt = ddn.P25Rx(1, lock_symbols=840, use_matched_filter=1).run(disc)
    r4, _ = orc.unpack_records10(rec[0, :cnt[0]])

    def gpu_hamming(bits):
        b = bits.copy()
        e = np.zeros(len(b), np.uint8)
        assert ddn.lib().ddn_fec_hamming_10_6_3_host(b.ctypes.data, len(b), e.ctypes.data) == 0
        return b, e

    def gpu_rs(d, p):
        x = d.copy()
        st = np.zeros(len(d), np.uint8)
        assert ddn.lib().ddn_fec_p25_rs_host(0, x.ctypes.data, p.ctypes.data, len(d), st.ctypes.data) == 0
        return x, st

    _check_lc([link_control(w, gpu_hamming, gpu_rs) for w in ldu1_words(r4, fl[0], int(cnt[0]))])


def ldu2_ess(rec4, fl, count, hamming_hard, rs_decode):
    """Every LDU2 of the capture -> (algid, kid, mi bits) after Hamming(10,6,3) + RS(24,16,9)
    (src/protocol/p25/phase1/p25p1_ldu2.c:141-170,256-262)."""
    import p25gen
    dpos, ppos = p25gen.ldu2_positions()
    pos = np.concatenate([dpos, ppos]) - 24
    rows = nids_from_records(rec4, fl, count)
    nid = decode_nids(rows, oracle_nid)
    out = []
    for (a, _, _), n in zip(rows, nid):
        if n[0] != 1 or n[2] != 10 or a + 1 + 840 > count:
            continue
        w = rec4[a + 1:a + 1 + 840][pos]
        bits = np.ascontiguousarray(np.stack([(w[:, :, 0] >> 1) & 1, w[:, :, 0] & 1], axis=2).reshape(24, 10).astype(np.uint8))
        fixed, errs = hamming_hard(bits)
        assert np.all(errs < 2)
        data = np.ascontiguousarray(fixed[:16, :6].reshape(1, 16, 6))
        par = np.ascontiguousarray(fixed[16:, :6].reshape(1, 8, 6))
        d, rc = rs_decode(data, par)
        assert rc[0] == 0
        hx = d[0]                                            # hex_data[0..15]
        mi = np.concatenate([hx[r] for r in range(15, 3, -1)])
        algid = int("".join(map(str, list(hx[3]) + list(hx[2][:2]))), 2)
        kid = int("".join(map(str, list(hx[2][2:]) + list(hx[1]) + list(hx[0]))), 2)
        out.append((algid, kid, mi))
    return out


def _check_ess(ess):
    assert len(ess) >= 4
    for algid, kid, mi in ess:
        assert algid == 0x80 and kid == 0 and not mi.any()    # clear voice: ALGID 0x80, KID 0, MI 0


def test_voice_capture_ldu2_encryption_sync_is_clear(built):
    from test_oracle_rs import oracle_rs
    g = golden("iq_p25p1_c4fm_vc.npz")
    _, sym, rec4, fl = oracle_chain(g["iq"], 840)
    _check_ess(ldu2_ess(rec4, fl, len(sym), _oracle_hamming, lambda d, p: oracle_rs("24_16_9", d, p)))


@pytest.mark.gpu
def test_voice_capture_ldu2_gpu(built):
    import ddn
    g = golden("iq_p25p1_c4fm_vc.npz")
    iq = np.ascontiguousarray(g["iq"])
    disc = ddn.Batch(1, block_len=8192).run_host(iq[None], iq.shape[0])
    rec, fl, cnt = ddn.P25Rx(1, lock_symbols=840, use_matched_filter=1).run(disc)
    r4, _ = orc.unpack_records10(rec[0, :cnt[0]])

    def gpu_hamming(bits):
        b = bits.copy()
        e = np.zeros(len(b), np.uint8)
        assert ddn.lib().ddn_fec_hamming_10_6_3_host(b.ctypes.data, len(b), e.ctypes.data) == 0
        return b, e

    def gpu_rs(d, p):
        x = d.copy()
        st = np.zeros(len(d), np.uint8)
        assert ddn.lib().ddn_fec_p25_rs_host(1, x.ctypes.data, p.ctypes.data, len(d), st.ctypes.data) == 0
        return x, st

    _check_ess(ldu2_ess(r4, fl[0], int(cnt[0]), gpu_hamming, gpu_rs))


# ---- the remaining P25 Phase 1 known answers the reference holds (tests/CMakeLists.txt:8907-8921) ----------------------
# OP25-compatible orientation maps of the differential demodulator's dibits (include/dsd-neo/core/p25_cqpsk_dibit.h:29-35);
# frame_sync_find_rotated_p25_cqpsk_map (src/dsp/dsd_frame_sync.c:580-597) tries X2400, N1200, P1200 once the plain and
# reverse-polarity patterns failed
CQPSK_MAPS = {"identity": (0, 1, 2, 3), "rev_p": (2, 3, 0, 1), "x2400": (3, 2, 1, 0), "n1200": (1, 3, 0, 2), "p1200": (2, 0, 3, 1)}


def cqpsk_dibits(sym):
    """fixed 4-level slice + the orientation map under which the first frame sync appears (reference search order)"""
    raw = np.where(sym >= 2, 1, np.where(sym >= 0, 0, np.where(sym >= -2, 2, 3))).astype(np.int64)
    first = {}
    for name in ("identity", "rev_p", "x2400", "n1200", "p1200"):
        d = np.array(CQPSK_MAPS[name])[raw]
        hits = [i for i in range(len(d) - 24) if np.array_equal(d[i:i + 24], FS)]
        if hits:
            first[name] = hits[0]
    assert first, "no P25p1 frame sync under any orientation"
    name = min(first, key=lambda k: (first[k], list(CQPSK_MAPS).index(k)))
    return np.array(CQPSK_MAPS[name])[raw], name


def cqpsk_records(d):
    """dibits -> the (rec4, flags) shape the record-based helpers above take: hard decisions at full confidence"""
    rec4 = np.zeros((len(d), 4), np.int32)
    rec4[:, 0] = d
    rec4[:, 1] = 255
    rec4[:, 2] = np.where((d >> 1) & 1, 255, -255)
    rec4[:, 3] = np.where(d & 1, 255, -255)
    fl = np.zeros(len(d), np.uint8)
    for i in range(len(d) - 24):
        if np.array_equal(d[i:i + 24], FS):
            fl[i + 23] = 2
    return rec4, fl


def check_cqpsk_voice(sym, hamming, rs):
    d, name = cqpsk_dibits(sym)
    assert name == "n1200"                                   # this capture sits a quarter turn per symbol off: map N1200
    rec4, fl = cqpsk_records(d)
    rows = nids_from_records(rec4, fl, len(d))
    nid = decode_nids(rows, oracle_nid)
    assert np.all(nid[:, 0] == 1) and len(set(nid[:, 1])) == 1
    duids = list(nid[:, 2])
    assert duids.count(15) >= 5 and 0 in duids and duids[-6:] == [5, 10, 5, 10, 5, 10]   # TDULCs, HDU, then the voice call
    lcs = [link_control(w, hamming, rs) for w in ldu1_words(rec4, fl, len(d))]
    assert len(lcs) >= 3
    lcf = [int("".join(map(str, lc[:8])), 2) for lc in lcs]
    # this call alternates "Group Voice Channel Update" (LCF 0x42: implicit MFID, LCO 2) with "Group Voice Channel User"
    # (LCF 0x00, MFID 0x00) - the string DECODE_IQ_P25P1_CQPSK_VOICE waits for
    assert set(lcf) <= {0x00, 0x42} and 0x00 in lcf
    user = [lc for lc, f in zip(lcs, lcf) if f == 0x00]
    for lc in user:
        assert int("".join(map(str, lc[8:16])), 2) == 0
        assert int("".join(map(str, lc[32:48])), 2) != 0 and int("".join(map(str, lc[48:72])), 2) != 0   # talkgroup, source


def check_simulcast(hits, blocks):
    assert len(hits) >= 50 and all(crc16_tsbk_ok(b) for b in blocks)
    ops = [int(b[0]) & 0x3F for b in blocks]
    upd = [b for b in blocks if (int(b[0]) & 0x3F) == 0x02 and int(b[1]) == 0]
    assert len(upd) >= 1, ops         # TSBK 0x02 = MAC 0x42 "Group Voice Channel Grant Update - Implicit" (p25p2_vpdu.c:1645-1653)


def _cqpsk_sym(name):
    g = golden(name)
    x = ((g["iq"].astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    return g, orc.OracleCqpskFe(rate=48000).run(x, 8192)


def test_cqpsk_voice_capture_is_group_voice_channel_user(built):
    """DECODE_IQ_P25P1_CQPSK_VOICE (tests/CMakeLists.txt:8907-8912)"""
    from test_oracle_rs import oracle_rs
    _, sym = _cqpsk_sym("iq_p25p1_cqpsk_vc.npz")
    check_cqpsk_voice(sym, _oracle_hamming, lambda d, p: oracle_rs("24_12_13", d, p))


def test_cqpsk_simulcast_capture_carries_grant_update(built):
    """DECODE_IQ_P25P1_CQPSK_SIMULCAST_CC (tests/CMakeLists.txt:8913-8921): two-ray fading, every TSBK still passes its CRC"""
    import fecgen
    _, sym = _cqpsk_sym("iq_p25p1_cqpsk_cc_simulcast.npz")
    hits, llr = cqpsk_tsbk_inputs(sym)
    blocks, _ = fecgen.oracle_p25_half_rate(llr)
    check_simulcast(hits, blocks)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["iq_p25p1_cqpsk_vc.npz", "iq_p25p1_cqpsk_cc_simulcast.npz"])
def test_cqpsk_remaining_known_answers_gpu(built, name):
    import ddn
    g, want = _cqpsk_sym(name)
    iq = np.ascontiguousarray(g["iq"])
    b = ddn.CqpskBatch(1, rate=48000, block_len=8192, input_format=ddn.IN_CU8)
    sym, cnt = b.run(iq[None])
    assert cnt[0] == len(want) and np.array_equal(sym[0, :cnt[0]].view(np.uint32), want.view(np.uint32))
    if "simulcast" in name:
        hits, llr = cqpsk_tsbk_inputs(sym[0, :cnt[0]])
        out = np.zeros((len(hits), 12), np.uint8)
        met = np.zeros(len(hits), np.int32)
        assert ddn.lib().ddn_fec_p25_12_soft_host(llr.ctypes.data, len(hits), out.ctypes.data, met.ctypes.data) == 0
        check_simulcast(hits, out)
    else:
        def gpu_hamming(bits):
            x = bits.copy()
            e = np.zeros(len(x), np.uint8)
            assert ddn.lib().ddn_fec_hamming_10_6_3_host(x.ctypes.data, len(x), e.ctypes.data) == 0
            return x, e

        def gpu_rs(d, p):
            x = d.copy()
            st = np.zeros(len(d), np.uint8)
            assert ddn.lib().ddn_fec_p25_rs_host(0, x.ctypes.data, p.ctypes.data, len(d), st.ctypes.data) == 0
            return x, st
        check_cqpsk_voice(sym[0, :cnt[0]], gpu_hamming, gpu_rs)


def test_control_channel_capture_every_tsbk_passes_crc(built):
    """Every trellis block of every TSDU of the reference's C4FM control-channel capture decodes to a TSBK whose CRC-16
    holds (multi-block TSDUs are walked until the Last Block flag): a timing slip of one sample or a mis-sliced dibit the
    trellis had to absorb would show up here as a failed CRC or a non-zero path metric."""
    import ctypes as C
    import ddn
    import fecgen
    g = golden("iq_p25p1_c4fm_cc.npz")
    _, sym, rec4, fl = oracle_chain(g["iq"], 336)                 # 24 + 336 = the capture's three-block TSDUs
    rows = nids_from_records(rec4, fl, len(sym))
    nid = decode_nids(rows, oracle_nid)
    n_blocks = n_frames = 0
    for (a, _, _), nd in list(zip(rows, nid))[1:]:
        if nd[0] != 1 or nd[2] != 7:
            continue
        n_frames += 1
        for b in range(3):
            pos = np.zeros(98, np.int32)
            end = ddn.lib().ddn_p25p1_layout_trellis_block(b, pos.ctypes.data)
            if a - 23 + end > len(sym):
                break
            blk = rec4[a - 23 + pos]
            llr = np.stack([blk[:, 2], blk[:, 3]], axis=1).reshape(1, 196).astype(np.int16)
            out, met = fecgen.oracle_p25_half_rate(np.ascontiguousarray(llr))
            assert crc16_tsbk_ok(out[0]), (a, b)
            n_blocks += 1
            if out[0][0] & 0x80:                                  # Last Block
                break
    assert n_frames >= 24 and n_blocks >= 48
