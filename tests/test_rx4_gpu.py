"""GPU parity of the DMR / NXDN48 receive loop (ddn_fsk4_rx_*, dsd-neo_amd/csrc/ddn_rx4.hip) against the CPU restatement
(oracle/ddn_oracle_rx4.c), bit for bit through the C-ABI: records, flags, payload dibits + reliabilities, per-sync hand-over,
carried state across ragged call splits; then the DMR known answers end to end on the device (burst gather -> Golay(20,8) ->
BPTC(196,96))."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import rx4

pytestmark = pytest.mark.gpu


def rec4_of(rec):
    """10-byte records -> [.., 4] int32 {dibit, rel, llr0, llr1} + float symbols"""
    r = rec.reshape(-1, 10)
    d = r[:, 0].astype(np.int32)
    rel = r[:, 1].astype(np.int32)
    l0 = r[:, 2:4].copy().view(np.int16).astype(np.int32).reshape(-1)
    l1 = r[:, 4:6].copy().view(np.int16).astype(np.int32).reshape(-1)
    sym = r[:, 6:10].copy().view(np.float32).reshape(-1)
    return np.stack([d, rel, l0, l1], axis=1), sym


def check_channel(got, c, want):
    k = int(got["cnt"][c])
    assert k == len(want["sym"]), (c, k, len(want["sym"]))
    r4, sym = rec4_of(got["rec"][c, :k])
    assert np.array_equal(sym.view(np.uint32), want["sym"].view(np.uint32)), c
    assert np.array_equal(r4, want["rec4"]), c
    assert np.array_equal(got["fl"][c, :k], want["fl"]), c
    assert np.array_equal(got["pay"][c, :k], want["pay"]), c
    ns = int(got["n_sync"][c])
    assert ns == len(want["sync_pos"]), c
    assert np.array_equal(got["sync_pos"][c, :ns], want["sync_pos"]) and np.array_equal(got["sync_pat"][c, :ns], want["sync_pat"])
    assert np.array_equal(got["pre"][c, :ns], want["pre"]) and np.array_equal(got["pre_rel"][c, :ns], want["pre_rel"])


CASES = [("iq_dmr_t3_ras_cc.npz", 2, rx4.PROTO_DMR, 0, 0), ("iq_dmr_t3_ras_cc.npz", 2, rx4.PROTO_DMR, 2, 0),
         ("iq_dmr_voice.npz", 2, rx4.PROTO_DMR, 0, 1), ("iq_dmr_t3_cc.npz", 2, rx4.PROTO_DMR, 2, 1),
         ("iq_nxdn48.npz", 1, rx4.PROTO_NXDN48, 0, 0), ("iq_nxdn48.npz", 1, rx4.PROTO_NXDN48, 2, 0),
         ("iq_nxdn96.npz", 2, rx4.PROTO_NXDN96, 0, 0), ("iq_nxdn96.npz", 2, rx4.PROTO_NXDN96, 2, 0)]
GPU_PROTO = {rx4.PROTO_DMR: ddn.FSK4_DMR, rx4.PROTO_NXDN48: ddn.FSK4_NXDN48, rx4.PROTO_NXDN96: ddn.FSK4_NXDN96, rx4.PROTO_M17: ddn.FSK4_M17}


@pytest.mark.parametrize("cap,lpf,proto,rf_mod,inv", CASES)
@pytest.mark.parametrize("use_filter", [1, 0])
def test_captures_bit_exact_with_call_splits(built, cap, lpf, proto, rf_mod, inv, use_filter):
    disc = rx4.capture_disc(cap, lpf)[:96000]
    n = len(disc)
    # channel c = the capture delayed by 37 * c samples behind noise, channel 3 negated, channel 4 = silence then signal
    B = 6
    rng = np.random.default_rng(3)
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        d = 37 * c
        x[c, :d] = rng.standard_normal(d) * 500
        x[c, d:] = disc[:n - d]
    x[3] = -x[3]
    x[4, :25000] = 0
    gpu = ddn.Fsk4Rx(B, GPU_PROTO[proto], rf_mod=rf_mod, inverted=inv,
                     use_matched_filter=use_filter)
    cpu = [rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=rf_mod, use_filter=use_filter, inverted=inv)) for _ in range(B)]
    cuts = [0, 4097, 4097 + 63, 30000, 30001, 61000, n]
    total_sync = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        got = gpu.run_host(x[:, a:b])
        for c in range(B):
            want = cpu[c].run(x[c, a:b], max_sync=got["sync_pos"].shape[1])
            check_channel(got, c, want)
            total_sync += len(want["sync_pos"])
            assert np.array_equal(gpu.thresholds(c).view(np.uint32), cpu[c].thresholds().view(np.uint32)), (c, a)
    assert total_sync > 60


def test_carrier_loss_and_custom_lock_lengths(built):
    """gaps long enough for the 1800-symbol timeout (filter memory goes stale, slicer and timing reset), per-channel handler
    lengths incl. 0 (sync reported, no in-frame symbols)"""
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)
    rng = np.random.default_rng(8)
    gap = (rng.standard_normal(26000) * 300).astype(np.float32)
    one = np.concatenate([disc[:30000], gap, disc[5000:45000], np.zeros(21000, np.float32), disc[:20000]])
    B = 5
    x = np.stack([np.roll(one, 11 * c) for c in range(B)])
    lock = np.array([[120, 1782, 0, 0], [0, 0, 0, 0], [54, 54, 0, 0], [700, 1, 0, 0], [2000, 2000, 0, 0]], np.int32)
    gpu = ddn.Fsk4Rx(B, ddn.FSK4_DMR, rf_mod=2)
    assert ddn.lib().ddn_fsk4_rx_set_lock_symbols(gpu.h, lock.ctypes.data) == 0
    cpu = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=2, lock=[int(v) for v in lock[c]])) for c in range(B)]
    half = x.shape[1] // 2 + 5
    for part in (slice(0, half), slice(half, None)):
        got = gpu.run_host(x[:, part])
        for c in range(B):
            check_channel(got, c, cpu[c].run(x[c, part], max_sync=got["sync_pos"].shape[1]))
    assert int(got["n_sync"][1]) > 20            # lock 0: every burst of the second half reports its sync


def test_wide_batch_both_wave_shapes(built):
    """more channels than one workgroup row, and the 32-channel shape (> 8192 channels)"""
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)[:20000]
    for B in (70, 8200):
        x = np.stack([np.roll(disc, 3 * (c % 97)) * (1.0 if c % 5 else -1.0) for c in range(B)]).astype(np.float32)
        got = ddn.Fsk4Rx(B, ddn.FSK4_DMR).run_host(x)
        for c in list(range(0, B, max(1, B // 9))) + [B - 1]:
            check_channel(got, c, rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR)).run(x[c], max_sync=got["sync_pos"].shape[1]))


@pytest.mark.parametrize("cap,lpf,proto", [("iq_dmr_t3_ras_cc.npz", 2, rx4.PROTO_DMR), ("iq_nxdn48.npz", 1, rx4.PROTO_NXDN48)])
@pytest.mark.parametrize("cpw", [2, 4])
def test_hunting_pass_of_all_rows_at_once_equals_one_owner_at_a_time(built, monkeypatch, cap, lpf, proto, cpw):
    """two / four channels per wavefront: the lanes that hunt take the bulk hunting pass together, one row of the wavefront
    each (ddn_rx4.hip); same records as the pass taken one owner after the other (ddn_fsk4_rx_set_debug_flags bit 65536) and as the
    oracle"""
    disc = rx4.capture_disc(cap, lpf)[:30000]
    B = 41
    x = np.stack([np.roll(disc, 211 * (c % 23)) * (0.35 + 0.05 * (c % 9)) for c in range(B)]).astype(np.float32)
    x[5, 9000:] = 0.0                                  # a channel whose carrier goes away hunts for the rest of the call
    x[6, :15000] = 0.0
    outs = []
    for dbg in (0, 65536):
        rx = ddn.Fsk4Rx(B, GPU_PROTO[proto])
        assert ddn.lib().ddn_fsk4_rx_set_channels_per_wave(rx.h, cpw) == 0
        assert ddn.lib().ddn_fsk4_rx_set_debug_flags(rx.h, dbg) == 0
        outs.append([rx.run_host(x[:, a:b]) for a, b in ((0, 14000), (14000, 30000))])
    for part in range(2):
        for k in outs[0][part]:
            assert np.array_equal(outs[0][part][k], outs[1][part][k]), k
    cpu = [rx4.OracleFsk4Rx(rx4.profile(proto)) for _ in range(B)]
    for part, (a, b) in enumerate(((0, 14000), (14000, 30000))):
        for c in (0, 5, 6, 7, 22, B - 1):
            check_channel(outs[0][part], c, cpu[c].run(x[c, a:b], max_sync=outs[0][part]["sync_pos"].shape[1]))


def test_dmr_known_answers_on_device(built):
    """RAS control-channel capture, everything after the front end on the device: receive loop -> burst gather -> Golay(20,8)
    slot type -> BPTC(196,96): colour code 0 on every burst ("Color Code=00") and C_ALOHA system identity Large / net 1 /
    site 1 (tests/CMakeLists.txt:8936-8947)."""
    import torch
    l = ddn.lib()
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)
    B, n = 4, len(disc)
    x = torch.from_numpy(np.stack([np.roll(disc, 17 * c) for c in range(B)])).cuda()
    rx = ddn.Fsk4Rx(B, ddn.FSK4_DMR, rf_mod=2)
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
    cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
    spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
    p = lambda t: t.data_ptr()
    assert l.ddn_fsk4_rx_run(rx.h, p(x), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
    S = B * my
    st, info, cach, valid = z((S, 20), torch.uint8), z((S, 196), torch.uint8), z((S, 24), torch.uint8), z((S,), torch.uint8)
    assert l.ddn_dmr_burst_gather(p(rec), p(cnt), ms, p(spos), p(pre), p(ns), B, my, 0, p(st), p(info), p(cach), p(valid), None) == 0
    ok = z((S,), torch.uint8)
    assert l.ddn_fec_block_code_batch(5, p(st), S, 1, None, p(ok), None) == 0
    out96, r3, errs = z((S, 96), torch.uint8), z((S, 3), torch.uint8), z((S,), torch.int32)
    assert l.ddn_fec_bptc_196x96_batch(p(info), 1, S, p(out96), p(r3), p(errs), None) == 0
    torch.cuda.synchronize()
    valid, ok, st, out96, errs, ns_h = (t.cpu().numpy() for t in (valid, ok, st, out96, errs, ns))
    n_aloha = 0
    for c in range(B):
        rows = [c * my + k for k in range(1, int(ns_h[c])) if valid[c * my + k]]
        assert len(rows) >= 55
        for r in rows:
            assert ok[r] == 1 and rx4.bits_int(st[r][:4]) == 0 and rx4.bits_int(st[r][4:8]) == 3 and errs[r] == 0
            pdu = out96[r]
            if rx4.bits_int(pdu[2:8]) == 25:
                n_aloha += 1
                assert rx4.bits_int(pdu[40:42]) == 2 and rx4.bits_int(pdu[42:46]) == 1 and rx4.bits_int(pdu[46:54]) == 1
    assert n_aloha >= 40


def test_trellis_decode_batch_and_dropin(built):
    l = ddn.lib()
    rng = np.random.default_rng(3)
    for ln in (32, 92):
        src = rng.integers(0, 2, (500, 2 * ln + 6), dtype=np.uint8)
        for i in range(0, 500, 2):
            bits = rng.integers(0, 2, ln + 3, dtype=np.uint8)
            reg, enc = 0, []
            for b in bits:
                reg = ((reg << 1) | int(b)) & 0x1F
                enc += [bin(reg & 0x19).count("1") & 1, bin(reg & 0x17).count("1") & 1]
            src[i] = np.array(enc, np.uint8)
            src[i, rng.choice(2 * ln, int(rng.integers(0, 5)), replace=False)] ^= 1
        want = rx4.oracle_trellis_decode(src, ln)
        got = np.zeros((500, ln), np.uint8)
        assert l.ddn_fec_trellis_decode_host(src.ctypes.data, src.shape[1], 500, ln, got.ctypes.data, ln) == 0
        assert np.array_equal(got, want)
        one = np.zeros(ln, np.uint8)
        row = np.ascontiguousarray(src[7])
        l.trellis_decode(one.ctypes.data, row.ctypes.data, ln)
        assert np.array_equal(one, want[7])
    assert l.ddn_fec_trellis_decode_host(src.ctypes.data, 10, 1, 32, got.ctypes.data, 32) != 0       # row too short for the lookahead


def test_nxdn48_known_answer_on_device(built):
    """NXDN48 capture, everything after the front end on the device: receive loop -> frame gather (de-scramble, LICH, SACCH /
    FACCH1 de-interleave + de-puncture) -> K=5 soft decode -> CRC6 -> greedy retry for the rows that failed -> CRC6; the host
    only strings the four SACCH parts together: VCALL source unit 901 ("Src=901", tests/CMakeLists.txt:8948).  The gather is also
    checked field by field against the python restatement of nxdn_frame()'s unpacking."""
    import torch
    l = ddn.lib()
    disc = rx4.capture_disc("iq_nxdn48.npz", 1)
    B, n = 2, len(disc)
    x = torch.from_numpy(np.stack([disc, np.roll(disc, 333)])).cuda()
    rx = ddn.Fsk4Rx(B, ddn.FSK4_NXDN48)
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    u8, i32 = torch.uint8, torch.int32
    rec, fl, pay = z((B, ms, 10), u8), z((B, ms), u8), z((B, ms, 2), u8)
    cnt, ns, spos = z((B,), i32), z((B,), i32), z((B, my), i32)
    spat, pre, prel = z((B, my), u8), z((B, my, 90), u8), z((B, my, 90), u8)
    p = lambda t: t.data_ptr()
    assert l.ddn_fsk4_rx_run(rx.h, p(x), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
    S = B * my
    lich, valid = z((S,), u8), z((S,), u8)
    ss, sr, fs, fr = z((S, 36, 2), u8), z((S, 36, 2), u8), z((S, 2, 96, 2), u8), z((S, 2, 96, 2), u8)
    assert l.ddn_nxdn_frame_gather(p(rec), p(cnt), ms, p(spos), p(ns), B, my, p(lich), p(ss), p(sr), p(fs), p(fr), p(valid), None) == 0
    soft, ok1 = z((S, 4), u8), z((S,), u8)
    assert l.ddn_fec_nxdn_conv_batch(p(ss), p(sr), S, 36, 32, None, p(soft), 4, None) == 0
    assert l.ddn_nxdn_crc_check_batch(p(soft), 4, S, 0, p(ok1), None) == 0
    hard_in = (ss.reshape(S, 72) >> 1).contiguous()                      # de-punctured hard bits, 72 >= 2 * 32 + 6
    hard, ok2 = z((S, 32), u8), z((S,), u8)
    assert l.ddn_fec_trellis_decode_batch(p(hard_in), 72, S, 32, p(hard), 32, None) == 0
    assert l.ddn_nxdn_crc_check_batch(p(hard), 32, S, 2, p(ok2), None) == 0      # kind 2: SACCH on bit rows
    torch.cuda.synchronize()
    hb = np.packbits(hard.cpu().numpy(), axis=1)
    rec_h, spos_h, ns_h, valid_h, lich_h = (t.cpu().numpy() for t in (rec, spos, ns, valid, lich))
    ss_h, sr_h, fs_h, fr_h = (t.cpu().numpy() for t in (ss, sr, fs, fr))
    soft_h, ok1_h, ok2_h = soft.cpu().numpy(), ok1.cpu().numpy(), ok2.cpu().numpy()
    for c in range(B):
        rows = []
        for k in range(int(ns_h[c])):
            s = c * my + k
            if not valid_h[s]:
                continue
            pos = int(spos_h[c, k])
            wl, wp, wss, wsr, wfs, wfr = rx4.nxdn_frame_fields(rec_h[c, pos + 1:pos + 183, 0] & 3, rec_h[c, pos + 1:pos + 183, 1])
            assert lich_h[s] == (wl | (0x80 if wp else 0))
            assert np.array_equal(ss_h[s], wss) and np.array_equal(sr_h[s], wsr)
            assert np.array_equal(fs_h[s], wfs) and np.array_equal(fr_h[s], wfr)
            t_soft = np.unpackbits(soft_h[s])[:32]
            assert bool(ok1_h[s]) == rx4.nxdn_crc_ok(t_soft, 0)
            t = t_soft if ok1_h[s] else np.unpackbits(hb[s])[:32]
            if ok1_h[s] or ok2_h[s]:
                rows.append(t)
        msgs = rx4.nxdn_superframes(rows)
        vcall = [m for ran, m in msgs if rx4.bits_int(m[2:8]) == 1]
        assert len(rows) >= 50 and len(vcall) >= 4
        assert all(rx4.bits_int(m[24:40]) == 901 for m in vcall)


# ---- the reference's handlers inside the loop (ddn_fsk4_rx_set_handlers) ----------------------------------------------------
def _oracle_handlers(x, proto, rf_mod):
    rx = rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=rf_mod, handler=1))
    out = rx.run(x)
    out["events"] = np.array([[e[0], e[1], e[2], (e[3] & 0xFFFF) | ((e[4] & 0xFFFF) << 16)] for e in rx.events.rows()],
                             np.int64).reshape(-1, 4)
    return out


HCASES = [("iq_dmr_t3_cc.npz", 2, rx4.PROTO_DMR, 2), ("iq_dmr_voice.npz", 2, rx4.PROTO_DMR, 2),
          ("iq_dmr_t3_ras_cc.npz", 2, rx4.PROTO_DMR, 2), ("iq_dmr_t3_ras_cc.npz", 2, rx4.PROTO_DMR, 0),
          ("iq_nxdn48.npz", 1, rx4.PROTO_NXDN48, 0), ("iq_nxdn96.npz", 2, rx4.PROTO_NXDN96, 2)]


@pytest.mark.parametrize("cap,lpf,proto,rf_mod", HCASES)
def test_handlers_in_the_loop_equal_the_oracle(built, cap, lpf, proto, rf_mod):
    """records, flags, payload, hand-overs AND the handlers' decisions (events) bit for bit, in one call and across call splits;
    channels = the capture delayed, negated, behind silence"""
    disc = rx4.capture_disc(cap, lpf)[:96000]
    n = len(disc)
    B = 6
    rng = np.random.default_rng(5)
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        d = 41 * c
        x[c, :d] = rng.standard_normal(d) * 500
        x[c, d:] = disc[:n - d]
    x[3] = -x[3]
    x[4, :5000] = 0.0
    want = [_oracle_handlers(x[c], proto, rf_mod) for c in range(B)]
    gproto = GPU_PROTO[proto]
    for splits in ([0, n], [0, 9, 4000, 4001, 30000, 52345, n]):
        rx = ddn.Fsk4Rx(B, gproto, rf_mod=rf_mod, handlers=True)
        parts = [rx.run_host(x[:, a:b]) for a, b in zip(splits[:-1], splits[1:])]
        for c in range(B):
            got = dict(cnt=[0] * B, n_sync=[0] * B)
            rec = np.concatenate([p["rec"][c, :p["cnt"][c]] for p in parts])
            fl = np.concatenate([p["fl"][c, :p["cnt"][c]] for p in parts])
            pay = np.concatenate([p["pay"][c, :p["cnt"][c]] for p in parts])
            base = np.cumsum([0] + [int(p["cnt"][c]) for p in parts])
            ev = []
            for k, p in enumerate(parts):
                e = p["events"][c, :p["n_events"][c]].astype(np.int64)
                e[:, 0] += base[k]
                ev.append(e)
            ev = np.concatenate(ev)
            r4, sym = rec4_of(rec)
            w = want[c]
            assert len(sym) == len(w["sym"]) and np.array_equal(sym.view(np.uint32), w["sym"].view(np.uint32)), (c, splits)
            bad = np.flatnonzero(fl != w["fl"])
            assert bad.size == 0, (c, splits, bad[:4], fl[bad[:4]], w["fl"][bad[:4]])
            assert np.array_equal(r4, w["rec4"]) and np.array_equal(pay, w["pay"]), (c, splits)
            m = np.array([-1, -1, -1, 0xFFFFFFFF])
            assert len(ev) == len(w["events"]) and np.array_equal(ev & m, w["events"] & m), (c, splits, ev[:5], w["events"][:5])


def test_dmr_color_code_02_on_the_device(built):
    """DECODE_IQ_DMR_VOICE / DECODE_IQ_DMR_T3_CC: the reference prints "Color Code=02" on these captures under plain -fs
    (tests/CMakeLists.txt:8925-8930); with its handlers inside the loop the device reports the same (event kind 6)"""
    for cap in ("iq_dmr_t3_cc.npz", "iq_dmr_voice.npz"):
        disc = rx4.capture_disc(cap, 2)
        rx = ddn.Fsk4Rx(1, ddn.FSK4_DMR, rf_mod=2, handlers=True)
        out = rx.run_host(disc[None])
        ev = out["events"][0, :out["n_events"][0]]
        printed = ev[ev[:, 1] == 6][:, 2]
        assert (printed == 2).sum() >= 4, (cap, printed)
    disc = rx4.capture_disc("iq_dmr_t3_ras_cc.npz", 2)
    out = ddn.Fsk4Rx(1, ddn.FSK4_DMR, rf_mod=2, handlers=True).run_host(disc[None])
    ev = out["events"][0, :out["n_events"][0]]
    printed = ev[ev[:, 1] == 6][:, 2]
    assert len(printed) >= 58 and set(printed) == {0}                      # DECODE_IQ_DMR_T3_RAS_CC_COLOR_CODE


def test_nxdn96_capture_ran_00_through_the_chain_object(built):
    """The reference's NXDN96 capture (4800 baud, tests/fixtures/iq/nxdn96.iq; DECODE_IQ_NXDN96 asserts "RAN 00",
    tests/CMakeLists.txt:8949) through ddn_fsk4_chain with protocol NXDN96 - front end (12.5 kHz filter), receive loop with the LICH gate
    inside, frame gather, SACCH K=5 decode + CRC6 - on the device: the frames' LICHs pass their parity, the SACCH parts pass their CRC
    and carry Radio Access Number 0 in every superframe part; the loop's records equal the CPU restatement's"""
    import ctypes as C
    from conftest import golden
    l = ddn.lib()
    iq = np.ascontiguousarray(golden("iq_nxdn96.npz")["iq"])
    n = len(iq)
    ch = ddn.Fsk4ChainC(1, n, ddn.FSK4_NXDN96, rf_mod=2, handlers=1, vocoder=0)
    d = C.c_void_p()
    assert l.ddn_device_alloc(iq.nbytes, C.byref(d)) == 0 and l.ddn_device_upload(d, iq.ctypes.data, iq.nbytes) == 0
    rans, n_valid, n_lich = [], 0, 0
    for step in ("run", "flush"):
        ch.run(d) if step == "run" else ch.flush()
        r = ch.results()
        S = int(r.max_syncs)
        valid = ch.fetch(r.d_valid, np.uint8, (S,))
        lich = ch.fetch(r.d_nxdn_lich, np.uint8, (S,))
        ok_soft, ok_hard = ch.fetch(r.d_nxdn_sacch_ok, np.uint8, (S,)), ch.fetch(r.d_nxdn_sacch_hard_ok, np.uint8, (S,))
        sacch = ch.fetch(r.d_nxdn_sacch, np.uint8, (S, 4))
        sacch_hard = ch.fetch(r.d_nxdn_sacch_hard, np.uint8, (S, 32))
        rows = np.flatnonzero(valid)
        n_valid += len(rows)
        n_lich += int(np.sum((lich[rows] & 0x80) != 0))
        for s in rows:
            bits = np.unpackbits(sacch[s])[:32] if ok_soft[s] else (sacch_hard[s] if ok_hard[s] else None)
            if bits is not None:
                rans.append(rx4.bits_int(bits[2:8]))
    assert n_valid >= 30 and n_lich >= n_valid - 6      # (the first frames fall in the filter's cold start)
    assert len(rans) >= 25 and all(v == 0 for v in rans)         # "RAN 00"
    ch.close()
    l.ddn_device_free(d)


def _m17_out_of(got, c, sync_thr):
    """the device loop's arrays of channel c in the shape tests/m17.py decodes (the oracle loop's output dict)"""
    k, ns = int(got["cnt"][c]), int(got["n_sync"][c])
    r4, sym = rec4_of(got["rec"][c, :k])
    return dict(sym=sym, rec4=r4, fl=got["fl"][c, :k], sync_pos=got["sync_pos"][c, :ns], sync_pat=got["sync_pat"][c, :ns], sync_thr=sync_thr)


@pytest.mark.parametrize("cpw", [0, 4])
def test_m17_loop_bit_exact_and_src_n0call_from_the_device(built, cpw):
    """DDN_FSK4_M17: frame_sync_try_m17()'s matcher inside the loop kernel (eight-symbol words, one error allowed, each word only after
    the sync type that may precede it, polarity from the preamble, EOT ends the transmission) = the restatement bit for bit on the
    reference's M17 capture - delayed, negated, behind silence, across ragged call splits - and the stream frames behind the DEVICE
    loop's syncs give the capture's known answer: the LSF reassembled from the LICH chunks passes its CRC16 and names N0CALL"""
    import m17
    disc = rx4.capture_disc("iq_m17.npz", 2)
    n = len(disc)
    B = 5
    rng = np.random.default_rng(4)
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        d = 41 * c
        x[c, :d] = rng.standard_normal(d) * 500
        x[c, d:] = disc[:n - d]
    x[3] = -x[3]
    x[4, :20000] = 0
    gpu = ddn.Fsk4Rx(B, ddn.FSK4_M17)
    if cpw:
        assert ddn.lib().ddn_fsk4_rx_set_channels_per_wave(gpu.h, cpw) == 0
    cpu = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)) for _ in range(B)]
    cuts = [0, 4097, 4097 + 63, 30000, 30001, n]
    n_sync = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        got = gpu.run_host(x[:, a:b])
        for c in range(B):
            want = cpu[c].run(x[c, a:b], max_sync=got["sync_pos"].shape[1])
            check_channel(got, c, want)
            n_sync += len(want["sync_pos"])
            assert np.array_equal(gpu.thresholds(c).view(np.uint32), cpu[c].thresholds().view(np.uint32)), (c, a)
    assert n_sync > 100
    # one call over the whole capture: the frames behind the device's syncs
    gpu = ddn.Fsk4Rx(2, ddn.FSK4_M17)
    got = gpu.run_host(np.stack([disc, -disc]))
    want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(disc, max_sync=got["sync_pos"].shape[1])
    check_channel(got, 0, want)
    fr = m17.decode_stream(_m17_out_of(got, 0, want["sync_thr"]))
    srcs = [m17.callsign(int.from_bytes(bytes(f["lich_lsf30"][6:12].tolist()), "big"))[1] for f in fr if f.get("lich_crc_ok")]
    assert len(srcs) >= 5 and set(srcs) == {"N0CALL"}
    st = [f for f in fr if f["kind"] == "str" and f["lich_err"] == 0]
    assert len(st) >= 35 and any(f["kind"] == "eot" for f in fr)


def test_m17_synthetic_transmissions_on_the_device(built):
    """preamble -> LSF -> 14 stream frames -> EOT built by the reference's own encoder, twice (with and without a gap: the second
    preamble is then matched on the other phase and taken inverted): device loop = restatement bit for bit, and the LSF / stream
    payloads decode from the device's records (the LSF's soft costs with the thresholds the sync left)"""
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    import m17
    import p25gen
    from test_oracle_m17 import two_transmissions
    for gap, seed in (([1, 3, 1], 5), ([], 1)):
        d, by, sent = two_transmissions(gap, seed)
        iq = p25gen.modulate_cu8(d, len(d) * 10 + 1200, lead=20, seed=seed, noise=0.02)
        disc = orc.OracleFrontEnd(profile=2).run_cu8(iq, 8192)
        gpu = ddn.Fsk4Rx(3, ddn.FSK4_M17)
        x = np.stack([disc, np.roll(disc, 7), -disc])
        got = gpu.run_host(x)
        wants = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(x[c], max_sync=got["sync_pos"].shape[1]) for c in range(3)]
        for c in range(3):
            check_channel(got, c, wants[c])
        fr = m17.decode_stream(_m17_out_of(got, 0, wants[0]["sync_thr"]))
        lsf = [f for f in fr if f["kind"] == "lsf" and f["crc_ok"]]
        assert len(lsf) >= 1 and np.array_equal(lsf[0]["lsf30"], by)
        st = [f for f in fr if f["kind"] == "str" and f["lich_err"] == 0]
        assert [(f["fn"], bytes(f["payload"].tolist())) for f in st[:14]] == [(fn, bytes(p.tolist())) for fn, p in sent]
