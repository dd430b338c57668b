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
        rel = np.repeat(nd[:, 1], 2).astype(np.uint8)
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
