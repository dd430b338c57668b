"""P25p1 framer: host offset tables (CPU) and the device-side sync index / field gathers / IMBE index (GPU).

CPU: ddn_p25p1_layout_* vs the independent python walk in tests/p25gen.py (which the real-capture tests anchor on the
reference's full-chain known answers).  GPU: everything downstream of the receive loop stays on the device - sync index,
NID gather -> BCH, trellis-block gather -> 1/2-rate Viterbi, LDU word gather -> Hamming, voice-frame index -> IMBE
de-interleave - and equals the host-side extraction + oracle chain slot by slot."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import p25gen
from conftest import golden


def test_layout_tables_match_python_walk(built):
    l = ddn.lib()
    nid = np.zeros(32, np.int32)
    assert l.ddn_p25p1_layout_nid(nid.ctypes.data) == 57
    assert list(nid) == [24 + k for k in range(33) if k != 11]
    blk = np.zeros(98, np.int32)
    assert l.ddn_p25p1_layout_trellis_block(0, blk.ctypes.data) > 0 and list(blk) == p25gen.block_positions()
    b1 = np.zeros(98, np.int32)
    l.ddn_p25p1_layout_trellis_block(1, b1.ctypes.data)
    assert b1[0] > blk[-1] and not np.any(b1 % 36 == 35) and len(set(b1)) == 98
    assert l.ddn_p25p1_layout_trellis_block(3, b1.ctypes.data) < 0
    for ldu, fn in ((1, p25gen.ldu1_positions), (2, p25gen.ldu2_positions)):
        w = np.zeros(120, np.int32)
        end = l.ddn_p25p1_layout_ldu_words(ldu, w.ctypes.data)
        d, p = fn()
        assert np.array_equal(w.reshape(24, 5), np.concatenate([d, p])) and end in (863, 864)
    first, st = np.zeros(9, np.int32), np.zeros(9, np.int32)
    assert l.ddn_p25p1_layout_ldu_imbe(first.ctypes.data, st.ctypes.data) in (863, 864)
    assert first[0] == 57 and st[0] == 21 and np.all(st == first % 36) and np.all(np.diff(first) >= 74)


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class Framer:
    def __init__(self, B, F):
        self.h = C.c_void_p()
        assert ddn.lib().ddn_p25p1_framer_create(B, F, C.byref(self.h)) == 0, ddn.lib().ddn_last_error()
        self.B, self.F = B, F

    def __del__(self):
        ddn.lib().ddn_p25p1_framer_destroy(self.h)


@pytest.mark.gpu
def test_device_chain_tsdu_traffic(built):
    """cu8 IQ -> front end -> rx loop -> framer -> NID BCH + 1/2-rate trellis without leaving the device."""
    import torch
    from test_e2e_p25 import B, N, NACS, _traffic, _oracle_chain, _check_decoded
    l = ddn.lib()
    iq, states = _traffic()
    want = _oracle_chain(iq)
    fe = ddn.Batch(B, block_len=8192)
    d_iq = _dev(iq)
    d_disc = torch.zeros((B, N), dtype=torch.float32, device="cuda")
    fe.run_device(d_iq.data_ptr(), N, d_disc.data_ptr())
    rx = ddn.P25Rx(B, lock_symbols=p25gen.FRAME - 24, use_matched_filter=1)
    ms = l.ddn_p25_rx_max_symbols(rx.h, N)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert l.ddn_p25_rx_run(rx.h, d_disc.data_ptr(), N, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    F = 32
    fr = Framer(B, F)
    assert l.ddn_p25p1_framer_index(fr.h, fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    S = B * F
    bits = torch.zeros((S, 63), dtype=torch.uint8, device="cuda")
    rel = torch.zeros((S, 63), dtype=torch.uint8, device="cuda")
    par = torch.zeros(S, dtype=torch.uint8, device="cuda")
    prel = torch.zeros(S, dtype=torch.uint8, device="cuda")
    v_nid = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_gather_nid(fr.h, rec.data_ptr(), cnt.data_ptr(), ms, bits.data_ptr(), rel.data_ptr(),
                                         par.data_ptr(), prel.data_ptr(), v_nid.data_ptr(), None) == 0
    obs = torch.zeros(S, dtype=torch.int32, device="cuda")
    nid = torch.zeros((S, 4), dtype=torch.int32, device="cuda")
    assert l.ddn_p25p1_nid_decode_batch(bits.data_ptr(), rel.data_ptr(), obs.data_ptr(), par.data_ptr(), prel.data_ptr(),
                                        64, S, nid.data_ptr(), None) == 0
    llr = torch.zeros((S, 196), dtype=torch.int16, device="cuda")
    v_blk = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_gather_trellis_block(fr.h, 0, rec.data_ptr(), cnt.data_ptr(), ms, llr.data_ptr(), None,
                                                   v_blk.data_ptr(), None) == 0
    out = torch.zeros((S, 12), dtype=torch.uint8, device="cuda")
    met = torch.zeros(S, dtype=torch.int32, device="cuda")
    assert l.ddn_fec_p25_12_soft_batch(llr.data_ptr(), S, out.data_ptr(), met.data_ptr(), None) == 0
    dib = torch.zeros((S, 98), dtype=torch.uint8, device="cuda")
    drl = torch.zeros((S, 98), dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_gather_r34_block(fr.h, 0, rec.data_ptr(), cnt.data_ptr(), ms, dib.data_ptr(), drl.data_ptr(),
                                               None, None) == 0
    torch.cuda.synchronize()
    ns = np.zeros(B, np.int32)
    pos = np.zeros((B, F), np.int32)
    assert l.ddn_p25p1_framer_get_syncs(fr.h, ns.ctypes.data, pos.ctypes.data) == 0
    dibh, drlh, rech, cnth = dib.cpu().numpy().reshape(B, F, 98), drl.cpu().numpy().reshape(B, F, 98), rec.cpu().numpy(), cnt.cpu().numpy()
    bp = np.array(p25gen.block_positions())
    for c in range(B):
        r4, _ = orc.unpack_records10(rech[c, :cnth[c]])
        for k in range(len(want[c]["nid"])):
            w = r4[int(pos[c, k]) - 23 + bp]
            assert np.array_equal(dibh[c, k], w[:, 0]) and np.array_equal(drlh[c, k], w[:, 1]), (c, k)
    nid, out, met = nid.cpu().numpy().reshape(B, F, 4), out.cpu().numpy().reshape(B, F, 12), met.cpu().numpy().reshape(B, F)
    v_blk = v_blk.cpu().numpy().reshape(B, F)
    bith, relh = bits.cpu().numpy().reshape(B, F, 63), rel.cpu().numpy().reshape(B, F, 63)
    parh, prelh = par.cpu().numpy().reshape(B, F), prel.cpu().numpy().reshape(B, F)
    for c in range(B):
        w = want[c]
        k = len(w["nid"])
        # the gathered NID fields themselves: hard bits, per-bit min(|llr|, 255) reliabilities, parity bit and its
        # reliability, as p25p1_read_nid_fields builds them (dispatch_p25p1.c:123-143) from the same records
        assert np.array_equal(bith[c, :k], w["bits"]) and np.array_equal(relh[c, :k], w["rel"]), c
        assert np.array_equal(parh[c, :k], w["par"]) and np.array_equal(prelh[c, :k], w["prel"]), c
        assert ns[c] == len(w["acc"]) and np.array_equal(pos[c, :ns[c]], w["acc"]), c   # every accepted sync, in order
        k = len(w["nid"])                                  # frames the host extraction found complete
        assert np.all(v_blk[c, :k] == 1) and np.all(v_blk[c, ns[c]:] == 0)
        assert np.array_equal(nid[c, :k], w["nid"]), c
        assert np.array_equal(out[c, :k], w["blocks"]) and np.array_equal(met[c, :k], w["met"]), c
        _check_decoded(c, w["acc"], nid[c, :k], out[c, :k], states[c])


@pytest.mark.gpu
def test_device_chain_voice_capture(built):
    """The reference's own voice capture: LDU word gathers -> Hamming, voice-frame index -> IMBE de-interleave, all
    from device-resident records; compared with host-side extraction (tests/p25gen.py positions) + the oracles."""
    import torch
    from test_real_capture import nids_from_records, decode_nids
    from test_oracle_block import oracle_nid
    l = ddn.lib()
    g = golden("iq_p25p1_c4fm_vc.npz")
    iq = np.ascontiguousarray(g["iq"])
    n = iq.shape[0]
    disc = ddn.Batch(1, block_len=8192).run_host(iq[None], n)
    rx = ddn.P25Rx(1, lock_symbols=840, use_matched_filter=1)
    ms = l.ddn_p25_rx_max_symbols(rx.h, n)
    d_disc = _dev(disc)
    rec = torch.zeros((1, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((1, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert l.ddn_p25_rx_run(rx.h, d_disc.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    F = 24
    fr = Framer(1, F)
    assert l.ddn_p25p1_framer_index(fr.h, fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    count = int(cnt.cpu()[0])
    r4, _ = orc.unpack_records10(rec.cpu().numpy()[0, :count])
    flh = fl.cpu().numpy()[0]
    rows = nids_from_records(r4, flh, count)
    nid = decode_nids(rows, oracle_nid)
    ns = np.zeros(1, np.int32)
    pos = np.zeros((1, F), np.int32)
    assert l.ddn_p25p1_framer_get_syncs(fr.h, ns.ctypes.data, pos.ctypes.data) == 0
    assert ns[0] >= len(rows) and np.array_equal(pos[0, :len(rows)], [r[0] for r in rows])
    for ldu, duid, posfn in ((1, 5, p25gen.ldu1_positions), (2, 10, p25gen.ldu2_positions)):
        wb = torch.zeros((F, 240), dtype=torch.uint8, device="cuda")
        wr = torch.zeros((F, 240), dtype=torch.uint8, device="cuda")
        vv = torch.zeros(F, dtype=torch.uint8, device="cuda")
        assert l.ddn_p25p1_framer_gather_ldu_words(fr.h, ldu, rec.data_ptr(), cnt.data_ptr(), ms, wb.data_ptr(),
                                                   wr.data_ptr(), vv.data_ptr(), None) == 0
        errs = torch.zeros(F * 24, dtype=torch.uint8, device="cuda")
        fixed = wb.clone()
        assert l.ddn_fec_hamming_10_6_3_batch(fixed.data_ptr(), F * 24, errs.data_ptr(), None) == 0
        torch.cuda.synchronize()
        wbh, wrh, vvh = wb.cpu().numpy().reshape(F, 24, 10), wr.cpu().numpy().reshape(F, 24, 10), vv.cpu().numpy()
        d, p = posfn()
        rel_pos = np.concatenate([d, p]) - 24
        seen = 0
        for k, ((a, _, _), nd) in enumerate(zip(rows, nid)):
            if nd[0] != 1 or nd[2] != duid or a + 1 + 840 > count:
                continue
            w = r4[a + 1:a + 1 + 840][rel_pos]              # [24, 5, 4]
            bits = np.stack([(w[:, :, 0] >> 1) & 1, w[:, :, 0] & 1], axis=2).reshape(24, 10)
            # soft_abs_i16 of each bit's own LLR (p25p1_ldu.c:135-152)
            rl = np.minimum(np.abs(np.stack([w[:, :, 2], w[:, :, 3]], axis=2)), 255).reshape(24, 10)
            assert vvh[k] == 1 and np.array_equal(wbh[k], bits) and np.array_equal(wrh[k], rl), (ldu, k)
            seen += 1
        assert seen >= 4
        assert int(errs.cpu().numpy().reshape(F, 24)[vvh == 1].max()) <= 2
        # ... and on through Reed-Solomon, still on the device: LDU1 link control / LDU2 encryption sync
        nd = 12 if ldu == 1 else 16
        dd = torch.zeros((F, nd, 6), dtype=torch.uint8, device="cuda")
        pp = torch.zeros((F, 24 - nd, 6), dtype=torch.uint8, device="cuda")
        rs_st = torch.zeros(F, dtype=torch.uint8, device="cuda")
        assert l.ddn_p25p1_framer_pack_ldu_rs(fr.h, ldu, fixed.data_ptr(), dd.data_ptr(), pp.data_ptr(), None) == 0
        assert l.ddn_fec_p25_rs_batch(0 if ldu == 1 else 1, dd.data_ptr(), pp.data_ptr(), F, rs_st.data_ptr(), None) == 0
        torch.cuda.synchronize()
        ddh, rsh = dd.cpu().numpy(), rs_st.cpu().numpy()
        for k, ((a, _, _), nd_) in enumerate(zip(rows, nid)):
            if nd_[0] != 1 or nd_[2] != duid or a + 1 + 840 > count:
                continue
            assert rsh[k] == 0, (ldu, k)
            if ldu == 1:   # "Group Voice Channel User": LCO 0, MFID 0 (the reference's DECODE_IQ_P25P1_C4FM_VOICE answer)
                lc = ddh[k][::-1].reshape(72)
                assert int("".join(map(str, lc[:8])), 2) == 0 and int("".join(map(str, lc[8:16])), 2) == 0, k
            else:          # clear voice: ALGID 0x80, KID 0
                hx = ddh[k]
                assert int("".join(map(str, list(hx[3]) + list(hx[2][:2]))), 2) == 0x80, k
                assert int("".join(map(str, list(hx[2][2:]) + list(hx[1]) + list(hx[0]))), 2) == 0, k
    # voice frames: index -> de-interleave, against the oracle on host-extracted dibits
    first = torch.zeros(F * 9, dtype=torch.int64, device="cuda")
    sc = torch.zeros(F * 9, dtype=torch.int32, device="cuda")
    assert l.ddn_p25p1_framer_imbe_index(fr.h, ms, first.data_ptr(), sc.data_ptr(), None) == 0
    ofr = torch.zeros((F * 9, 8, 23), dtype=torch.uint8, device="cuda")
    osf = torch.zeros((F * 9, 8, 23, 2), dtype=torch.uint8, device="cuda")
    ofl = torch.zeros(F * 9, dtype=torch.uint8, device="cuda")
    osc = torch.zeros(F * 9, dtype=torch.int32, device="cuda")
    assert l.ddn_p25p1_imbe_deinterleave_batch(rec.data_ptr(), ms, first.data_ptr(), sc.data_ptr(), F * 9,
                                               ofr.data_ptr(), osf.data_ptr(), ofl.data_ptr(), osc.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ofr, osf, ofl = ofr.cpu().numpy(), osf.cpu().numpy(), ofl.cpu().numpy()
    f9, s9 = np.zeros(9, np.int32), np.zeros(9, np.int32)
    l.ddn_p25p1_layout_ldu_imbe(f9.ctypes.data, s9.ctypes.data)
    checked = 0
    for k, ((a, _, _), nd) in enumerate(zip(rows, nid)):
        if nd[0] != 1 or nd[2] not in (5, 10) or a + 1 + 840 > count:
            continue
        for v in range(9):
            s0 = a - 23 + int(f9[v])
            want = orc.oracle_imbe_deinterleave(r4[s0:s0 + 80, 0], r4[s0:s0 + 80, 2], r4[s0:s0 + 80, 3], int(s9[v]))
            i = k * 9 + v
            assert ofl[i] == want[2] and np.array_equal(ofr[i], want[0]) and np.array_equal(osf[i], want[1]), (k, v)
            checked += 1
    assert checked >= 72 and np.all(ofl[ns[0] * 9:] == 0xFF)


def _hdu_python_walk():
    """Independent restatement of the HDU field order (p25p1_hdu.c:191-200,252-268): -> (hex [36][3], par [36][6])
    frame dibit indices, word order hex_data[0..19], hex_parity[0..15]."""
    idx = 57
    hexp, parp = [None] * 36, [None] * 36

    def take(n):
        nonlocal idx
        out = []
        while len(out) < n:
            if idx % 36 != 35:
                out.append(idx)
            idx += 1
        return out
    for seq in range(36):
        word = 19 - seq if seq < 20 else 20 + (15 - (seq - 20))
        hexp[word] = take(3)
        parp[word] = take(6)
    return np.array(hexp), np.array(parp), idx


def test_hdu_layout_matches_python_walk(built):
    hx, pr = np.zeros(108, np.int32), np.zeros(216, np.int32)
    end = ddn.lib().ddn_p25p1_layout_hdu(hx.ctypes.data, pr.ctypes.data)
    h, p, idx = _hdu_python_walk()
    assert end == idx and np.array_equal(hx.reshape(36, 3), h) and np.array_equal(pr.reshape(36, 6), p)
    assert hx[19 * 3] == 57 and not np.any(np.concatenate([hx, pr]) % 36 == 35)


@pytest.mark.gpu
def test_device_chain_hdu(built):
    """Synthetic header data units: 20 random hex words -> RS(36,20,17) -> Golay(24,6) per word -> C4FM; the device chain
    (front end, rx loop, framer gathers, Golay, RS) must return the words that were sent."""
    import torch
    import fecgen
    l = ddn.lib()
    rng = np.random.default_rng(31)
    B, FR, NF = 4, 396, 9
    N = 10 * FR * NF + 600
    h, p, _ = _hdu_python_walk()
    nac = 0x2A5
    data16 = [(nac >> (11 - k)) & 1 for k in range(12)] + [0, 0, 0, 0]              # DUID 0 = HDU
    cw = list(fecgen.bch_63_16_encode(data16)) + [0]
    nid = [(cw[2 * k] << 1) | cw[2 * k + 1] for k in range(32)]
    sent = np.zeros((B, NF, 20), np.int64)
    iq = np.zeros((B, N, 2), np.uint8)
    for c in range(B):
        frames = np.zeros((NF, FR), np.int8)
        for f in range(NF):
            d = rng.integers(0, 64, 20)
            sent[c, f] = d
            syms = np.concatenate([d, fecgen.rs63_encode(d, 8)])                     # hex_data[0..19], hex_parity[0..15]
            fr = np.full(FR, 2, np.int8)                                             # status / filler dibits
            fr[:24] = orc.P25_FS_DIBITS
            fr[[q for q in range(24, 57) if q != 35]] = nid
            for w in range(36):
                b6 = np.array([(int(syms[w]) >> (5 - k)) & 1 for k in range(6)], np.uint8)
                d12 = np.zeros(12, np.uint8)
                d12[6:] = b6
                par = np.array(fecgen.golay24_encode(d12), np.uint8)
                fr[h[w]] = (b6[0::2] << 1) | b6[1::2]
                fr[p[w]] = (par[0::2] << 1) | par[1::2]
            frames[f] = fr
        iq[c] = p25gen.modulate_cu8(frames.reshape(-1), N, lead=210 + 29 * c, seed=40 + c)
    fe = ddn.Batch(B, block_len=8192)
    disc = fe.run_host(iq, N)
    rx = ddn.P25Rx(B, lock_symbols=FR - 24, use_matched_filter=1)
    ms = l.ddn_p25_rx_max_symbols(rx.h, N)
    d_disc = _dev(disc)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert l.ddn_p25_rx_run(rx.h, d_disc.data_ptr(), N, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    F = NF + 3
    S = B * F
    fr_ = Framer(B, F)
    assert l.ddn_p25p1_framer_index(fr_.h, fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    hb = torch.zeros((S, 36, 6), dtype=torch.uint8, device="cuda")
    pb = torch.zeros((S, 36, 12), dtype=torch.uint8, device="cuda")
    hl = torch.zeros((S, 36, 6), dtype=torch.int16, device="cuda")
    pl = torch.zeros((S, 36, 12), dtype=torch.int16, device="cuda")
    vv = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_gather_hdu(fr_.h, rec.data_ptr(), cnt.data_ptr(), ms, hb.data_ptr(), pb.data_ptr(),
                                         hl.data_ptr(), pl.data_ptr(), vv.data_ptr(), None) == 0
    raw_h = hb.clone()
    gst = torch.zeros(S * 36, dtype=torch.uint8, device="cuda")
    gfx = torch.zeros(S * 36, dtype=torch.int32, device="cuda")
    assert l.ddn_fec_golay24_batch(6, hb.data_ptr(), pb.data_ptr(), S * 36, gst.data_ptr(), gfx.data_ptr(), None) == 0
    dd = torch.zeros((S, 20, 6), dtype=torch.uint8, device="cuda")
    pp = torch.zeros((S, 16, 6), dtype=torch.uint8, device="cuda")
    rst = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_pack_hdu_rs(fr_.h, hb.data_ptr(), dd.data_ptr(), pp.data_ptr(), None) == 0
    assert l.ddn_fec_p25_rs_batch(2, dd.data_ptr(), pp.data_ptr(), S, rst.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ns = np.zeros(B, np.int32)
    pos = np.zeros((B, F), np.int32)
    assert l.ddn_p25p1_framer_get_syncs(fr_.h, ns.ctypes.data, pos.ctypes.data) == 0
    vvh, ddh, rsh = vv.cpu().numpy().reshape(B, F), dd.cpu().numpy().reshape(B, F, 20, 6), rst.cpu().numpy().reshape(B, F)
    rawh, llh = raw_h.cpu().numpy().reshape(B, F, 36, 6), hl.cpu().numpy().reshape(B, F, 36, 6)
    rech, cnth = rec.cpu().numpy(), cnt.cpu().numpy()
    for c in range(B):
        r4, _ = orc.unpack_records10(rech[c, :cnth[c]])
        good = 0
        for k in range(int(ns[c])):
            if not vvh[c, k]:
                continue
            a = int(pos[c, k])
            w = r4[a - 23 + h]                                  # [36, 3, 4] host-side gather of the same dibits
            assert np.array_equal(rawh[c, k], np.stack([(w[:, :, 0] >> 1) & 1, w[:, :, 0] & 1], axis=2).reshape(36, 6))
            assert np.array_equal(llh[c, k], np.stack([w[:, :, 2], w[:, :, 3]], axis=2).reshape(36, 6))
            if k < 2:
                continue                                        # matched-filter / threshold warm-up frames
            assert rsh[c, k] == 0, (c, k)
            got = (ddh[c, k] * (1 << np.arange(5, -1, -1))).sum(axis=1)
            hits = [f for f in range(NF) if np.array_equal(sent[c, f], got)]
            assert len(hits) == 1, (c, k)
            good += 1
        assert good >= NF - 3, (c, good)


def _tdulc_python_walk():
    """p25p1_tdulc.c:199-207: dodeca_data[5..0] then dodeca_parity[5..0], each 6 data + 6 Golay-parity dibits."""
    idx = 57
    d, p = [None] * 12, [None] * 12

    def take(n):
        nonlocal idx
        out = []
        while len(out) < n:
            if idx % 36 != 35:
                out.append(idx)
            idx += 1
        return out
    for seq in range(12):
        word = 5 - seq if seq < 6 else 6 + (5 - (seq - 6))
        d[word] = take(6)
        p[word] = take(6)
    return np.array(d), np.array(p), idx


def test_tdulc_layout_matches_python_walk(built):
    a, b = np.zeros(72, np.int32), np.zeros(72, np.int32)
    end = ddn.lib().ddn_p25p1_layout_tdulc(a.ctypes.data, b.ctypes.data)
    d, p, idx = _tdulc_python_walk()
    assert end == idx and np.array_equal(a.reshape(12, 6), d) and np.array_equal(b.reshape(12, 6), p)


@pytest.mark.gpu
def test_device_chain_tdulc(built):
    """Synthetic terminator-with-link-control units: 12 hex words -> RS(24,12,13) -> pairs of hex words, halves swapped,
    as Golay(24,12) codewords -> C4FM; device chain: gathers, Golay, half swap + RS must return the 12 words sent."""
    import torch
    import fecgen
    l = ddn.lib()
    rng = np.random.default_rng(77)
    B, FR, NF = 3, 216, 12
    N = 10 * FR * NF + 600
    dpos, ppos, _ = _tdulc_python_walk()
    nac = 0x1B7
    data16 = [(nac >> (11 - k)) & 1 for k in range(12)] + [1, 1, 1, 1]              # DUID 15 = TDULC
    cw = list(fecgen.bch_63_16_encode(data16)) + [0]
    nid = [(cw[2 * k] << 1) | cw[2 * k + 1] for k in range(32)]
    sent = np.zeros((B, NF, 12), np.int64)
    iq = np.zeros((B, N, 2), np.uint8)
    for c in range(B):
        frames = np.zeros((NF, FR), np.int8)
        for f in range(NF):
            d = rng.integers(0, 64, 12)
            sent[c, f] = d
            hexw = np.concatenate([d, fecgen.rs63_encode(d, 6)])                     # hex data 0..11, hex parity 0..11
            fr = np.full(FR, 2, np.int8)
            fr[:24] = orc.P25_FS_DIBITS
            fr[[q for q in range(24, 57) if q != 35]] = nid
            for w in range(12):                                                      # dodeca word w = hex (2w+1, 2w)
                hi, lo = int(hexw[2 * w + 1]), int(hexw[2 * w])
                b12 = np.array([(hi >> (5 - k)) & 1 for k in range(6)] + [(lo >> (5 - k)) & 1 for k in range(6)], np.uint8)
                par = np.array(fecgen.golay24_encode(b12), np.uint8)
                fr[dpos[w]] = (b12[0::2] << 1) | b12[1::2]
                fr[ppos[w]] = (par[0::2] << 1) | par[1::2]
            frames[f] = fr
        iq[c] = p25gen.modulate_cu8(frames.reshape(-1), N, lead=190 + 31 * c, seed=70 + c)
    disc = ddn.Batch(B, block_len=8192).run_host(iq, N)
    rx = ddn.P25Rx(B, lock_symbols=FR - 24, use_matched_filter=1)
    ms = l.ddn_p25_rx_max_symbols(rx.h, N)
    d_disc = _dev(disc)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
    fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert l.ddn_p25_rx_run(rx.h, d_disc.data_ptr(), N, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    F = NF + 3
    S = B * F
    fr_ = Framer(B, F)
    assert l.ddn_p25p1_framer_index(fr_.h, fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    db = torch.zeros((S, 12, 12), dtype=torch.uint8, device="cuda")
    pb = torch.zeros((S, 12, 12), dtype=torch.uint8, device="cuda")
    vv = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_gather_tdulc(fr_.h, rec.data_ptr(), cnt.data_ptr(), ms, db.data_ptr(), pb.data_ptr(), None,
                                           None, vv.data_ptr(), None) == 0
    gst = torch.zeros(S * 12, dtype=torch.uint8, device="cuda")
    gfx = torch.zeros(S * 12, dtype=torch.int32, device="cuda")
    assert l.ddn_fec_golay24_batch(12, db.data_ptr(), pb.data_ptr(), S * 12, gst.data_ptr(), gfx.data_ptr(), None) == 0
    rd = torch.zeros((S, 12, 6), dtype=torch.uint8, device="cuda")
    rp = torch.zeros((S, 12, 6), dtype=torch.uint8, device="cuda")
    rst = torch.zeros(S, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p1_framer_pack_tdulc_rs(fr_.h, db.data_ptr(), rd.data_ptr(), rp.data_ptr(), None) == 0
    assert l.ddn_fec_p25_rs_batch(0, rd.data_ptr(), rp.data_ptr(), S, rst.data_ptr(), None) == 0
    torch.cuda.synchronize()
    vvh, rdh, rsh = vv.cpu().numpy().reshape(B, F), rd.cpu().numpy().reshape(B, F, 12, 6), rst.cpu().numpy().reshape(B, F)
    for c in range(B):
        good = 0
        for k in range(2, F):
            if not vvh[c, k]:
                continue
            assert rsh[c, k] == 0, (c, k)
            got = (rdh[c, k] * (1 << np.arange(5, -1, -1))).sum(axis=1)
            assert len([f for f in range(NF) if np.array_equal(sent[c, f], got)]) == 1, (c, k)
            good += 1
        assert good >= NF - 3, (c, good)
