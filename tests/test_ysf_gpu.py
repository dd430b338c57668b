"""Yaesu System Fusion on the device: DDN_FSK4_YSF as the fsk4 loop's fifth protocol and the frame information channel behind its
syncs (ddn_ysf_fich_decode_batch: the K = 5 decoder's second consumer), against the CPU restatement on the reference's capture; then the
capture from cu8 I/Q through the chain object - "V/D2 RID Mode Repeater CC" (DECODE_IQ_YSF, tests/CMakeLists.txt:8953-8957)."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import rx4
import ysf
from test_rx4_gpu import check_channel, rec4_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cpw", [0, 4])
def test_ysf_loop_bit_exact_with_call_splits(built, cpw):
    disc = rx4.capture_disc("iq_ysf.npz", 2)[:120000]
    n = len(disc)
    B = 5
    rng = np.random.default_rng(6)
    x = np.zeros((B, n), np.float32)
    for c in range(B):
        d = 43 * c
        x[c, :d] = rng.standard_normal(d) * 500
        x[c, d:] = disc[:n - d]
    x[3] = -x[3]            # -YSF: the inverted word, negative polarity
    x[4, :20000] = 0
    for use_filter in (1, 0):
        gpu = ddn.Fsk4Rx(B, ddn.FSK4_YSF, use_matched_filter=use_filter)
        if cpw:
            assert ddn.lib().ddn_fsk4_rx_set_channels_per_wave(gpu.h, cpw) == 0
        cpu = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF, use_filter=use_filter)) for _ in range(B)]
        cuts = [0, 4097, 4097 + 63, 30000, 30001, 90000, n]
        n_sync = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            got = gpu.run_host(x[:, a:b])
            for c in range(B):
                want = cpu[c].run(x[c, a:b], max_sync=got["sync_pos"].shape[1])
                check_channel(got, c, want)
                n_sync += len(want["sync_pos"])
                assert np.array_equal(gpu.thresholds(c).view(np.uint32), cpu[c].thresholds().view(np.uint32)), (c, a)
        assert n_sync > 60
    assert int(np.sum(got["sync_pat"][3, :int(got["n_sync"][3])] == 1)) > 0      # the negated channel locks on -YSF


def test_fich_on_the_device_equals_the_restatement(built):
    import torch
    l = ddn.lib()
    disc = rx4.capture_disc("iq_ysf.npz", 2)
    x = np.stack([disc, -disc, np.roll(disc, 3)])
    B, n = x.shape
    d = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rx = ddn.Fsk4Rx(B, ddn.FSK4_YSF)
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
    cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
    spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
    p = lambda t: t.data_ptr()
    assert l.ddn_fsk4_rx_run(rx.h, p(d), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
    f4, st, ve = z((B, my, 4), torch.uint8), z((B, my), torch.uint8), z((B, my), torch.int32)
    assert l.ddn_ysf_fich_decode_batch(p(rec), ms, p(cnt), p(spos), p(ns), B, my, p(f4), p(st), p(ve), None) == 0, l.ddn_last_error()
    torch.cuda.synchronize()
    f4, st, ve = f4.cpu().numpy(), st.cpu().numpy(), ve.cpu().numpy().view(np.uint32)
    nsy, pos, cn, rc = ns.cpu().numpy(), spos.cpu().numpy(), cnt.cpu().numpy(), rec.cpu().numpy()
    good = 0
    for c in range(B):
        r4, _ = rec4_of(rc[c, :cn[c]])
        for k in range(int(nsy[c])):
            q = int(pos[c, k])
            if q + 101 > cn[c]:
                assert st[c, k] == 0
                continue
            err, bits, cost = ysf.fich(r4[q + 1:q + 101, 0])
            assert st[c, k] == {0: 1, -1: 2, -2: 3}[err], (c, k, err, st[c, k])
            assert np.array_equal(np.unpackbits(f4[c, k]), bits) and int(ve[c, k]) == cost, (c, k)
            good += err == 0
    assert good >= 40


def test_ysf_capture_through_the_chain_object(built):
    """cu8 I/Q in four calls + the flush -> front end -> loop -> FICH of every sync in the call that holds its last dibit: the same
    frames as the CPU pipeline over the whole stream, and every FICH that passes its CRC reads "V/D2 RID Mode Repeater CC" """
    from conftest import golden
    iq = np.ascontiguousarray(golden("iq_ysf.npz")["iq"], np.uint8)
    n = 60000
    calls = len(iq) // n
    B = 2
    x = np.stack([iq[:calls * n], np.roll(iq[:calls * n], 2 * 91)])
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_YSF, rf_mod=0, handlers=0, vocoder=0)
    l = ddn.lib()
    got = [[] for _ in range(B)]
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        S, T = r.max_syncs, r.carry_symbols
        f = ch.fetch
        ns, pos, pat = f(r.d_n_sync, np.int32, (B,)), f(r.d_sync_pos, np.int32, (B, S)), f(r.d_sync_pat, np.uint8, (B, S))
        f4, st, ve = f(r.d_ysf_fich4, np.uint8, (B, S, 4)), f(r.d_ysf_fich_status, np.uint8, (B, S)), f(r.d_ysf_fich_cost, np.uint32, (B, S))
        new = f(r.d_new, np.int32, (B,))
        for c in range(B):
            for k in range(int(ns[c])):
                got[c].append(dict(pos=int(base[c]) + int(pos[c, k]) - int(T), f4=f4[c, k].copy(), st=int(st[c, k]), ve=int(ve[c, k])))
            base[c] += int(new[c])

    for k in range(calls):
        part = np.ascontiguousarray(x[:, k * n:(k + 1) * n])
        p = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, part.ctypes.data, part.nbytes) == 0
        ch.run(p)
        take()
        l.ddn_device_free(p)
    ch.flush()
    take()
    ch.close()
    named = 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(x[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=4096)
        fr = ysf.decode_frames(want)
        mine = [g for g in got[c] if g["st"] != 0]
        assert [g["pos"] for g in mine] == [f["pos"] for f in fr], (c, len(mine), len(fr))
        for g, f in zip(mine, fr):
            assert g["st"] == {0: 1, -1: 2, -2: 3}[f["err"]] and np.array_equal(np.unpackbits(g["f4"]), f["bits"]) and g["ve"] == f["cost"]
            if f["err"] == 0:
                assert ysf.summary(ysf.fields(np.unpackbits(g["f4"]))) == "V/D2 RID Mode Repeater CC"
                named += 1
    assert named >= 30


def _payload_equal(info, dch, dst, dcost, ambe, errs, want, where, fr=None, nfr=None):
    pl = want["payload"]
    flags = want["fi"] | (want["dt"] << 2) | (16 if want["err"] != 0 else 0) | 32 | (128 if (pl is not None and pl["csd3"]) or want.get("csd3") else 0)
    assert int(info[1]) == flags, (where, int(info[1]), flags)
    if pl is None:
        assert int(info[0]) == 0, where
        return 0
    assert int(info[0]) == pl["kind"], (where, int(info[0]), pl["kind"])
    assert np.array_equal(dst, pl["dch_status"]) and np.array_equal(dcost, pl["dch_cost"]), (where, dst, pl["dch_status"])
    assert np.array_equal(dch, pl["dch"]), where
    if pl["kind"] == 2:
        assert np.array_equal(ambe, pl["ambe_d"]) and np.array_equal(errs, pl["errs2"]), where
    if fr is not None:
        assert int(nfr) == pl["n_frames"] and np.array_equal(fr, pl["frames"]), (where, int(nfr), pl["n_frames"])
    return int(pl["dch_status"][0] == 1) + int(pl["dch_status"][1] == 1)


def test_payload_on_the_device_equals_the_restatement(built):
    """the capture (V/D mode 2) + the same records read as the other frame types: a second and a third channel whose FICH bytes are
    replaced before the payload call (FI / DT of a V/D mode 1 frame, of a header) - every branch of ysf_dispatch_payload()"""
    import torch
    l = ddn.lib()
    disc = rx4.capture_disc("iq_ysf.npz", 2)
    x = np.stack([disc, disc, disc, np.roll(disc, 5)])
    B, n = x.shape
    d = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rx = ddn.Fsk4Rx(B, ddn.FSK4_YSF)
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
    cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
    spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
    p = lambda t: t.data_ptr()
    assert l.ddn_fsk4_rx_run(rx.h, p(d), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
    f4, st, ve = z((B, my, 4), torch.uint8), z((B, my), torch.uint8), z((B, my), torch.int32)
    assert l.ddn_ysf_fich_decode_batch(p(rec), ms, p(cnt), p(spos), p(ns), B, my, p(f4), p(st), p(ve), None) == 0, l.ddn_last_error()
    # channel 1: every good FICH says FI 1 / DT 0 (V/D mode 1); channel 2: FI 0 / DT 1 and FI 2 (header, terminator: full-rate data),
    # one frame in seven DT 3 (full-rate voice: named, not decoded)
    f4h = f4.cpu().numpy()
    f4h[1, :, 0] = (f4h[1, :, 0] & 0x3F) | (1 << 6)
    f4h[1, :, 2] &= 0xFC
    f4h[2, :, 0] = (f4h[2, :, 0] & 0x3F) | np.where(np.arange(my) % 3 == 0, 2 << 6, 0).astype(np.uint8)
    f4h[2, :, 2] = (f4h[2, :, 2] & 0xFC) | 1
    k7 = np.arange(my) % 7 == 3
    f4h[2, k7, 0] = (f4h[2, k7, 0] & 0x3F) | (1 << 6)
    f4h[2, k7, 2] |= 3
    k14 = np.arange(my) % 14 == 3                       # every other of those in the CSD3 layout: FN = 0, FT = 1
    f4h[2, k14, 1] = (f4h[2, k14, 1] & 0xC0) | 1
    f4 = torch.from_numpy(f4h).cuda()
    last = z((B, 2), torch.uint8)
    info, dch, dst = z((B, my, 2), torch.uint8), z((B, my, 2, 20), torch.uint8), z((B, my, 2), torch.uint8)
    dcost, ambe, errs = z((B, my, 2), torch.int32), z((B, my, 5, 49), torch.uint8), z((B, my, 5), torch.uint8)
    fr, nfr = z((B, my, 5, 184), torch.uint8), z((B, my), torch.uint8)
    assert l.ddn_ysf_payload_decode_batch(p(rec), ms, p(cnt), p(spos), p(ns), B, my, p(f4), p(st), p(last), p(info), p(dch), p(dst), p(dcost),
                                          p(ambe), p(errs), p(fr), p(nfr), None) == 0, l.ddn_last_error()
    torch.cuda.synchronize()
    g = lambda t: t.cpu().numpy()
    st, info, dch, dst, dcost, ambe, errs, last = g(st), g(info), g(dch), g(dst), g(dcost).view(np.uint32), g(ambe), g(errs), g(last)
    fr, nfr = g(fr), g(nfr)
    n_csd3 = 0
    nsy, pos, cn, rc = g(ns), g(spos), g(cnt), g(rec)
    good, kinds = 0, set()
    for c in range(B):
        r4, _ = rec4_of(rc[c, :cn[c]])
        dt, fi = 0, 0
        for k in range(int(nsy[c])):
            q = int(pos[c, k])
            if st[c, k] == 0:
                assert not info[c, k].any()
                continue
            csd3 = False
            if st[c, k] == 1:
                dt, fi = int(f4h[c, k, 2] & 3), int(f4h[c, k, 0] >> 6)
                csd3 = dt == 3 and fi == 1 and (int(f4h[c, k, 1]) & 7) == 1 and ((int(f4h[c, k, 1]) >> 3) & 7) == 0
            pl = ysf.payload(r4[q + 101:q + 461, 0], fi, dt, csd3) if q + 461 <= cn[c] else None
            want = dict(fi=fi, dt=dt, err=0 if st[c, k] == 1 else -1, payload=pl, csd3=csd3)
            good += _payload_equal(info[c, k], dch[c, k], dst[c, k], dcost[c, k], ambe[c, k], errs[c, k], want, (c, k), fr[c, k], nfr[c, k])
            n_csd3 += int(csd3 and pl is not None)
            kinds.add(int(info[c, k, 0]))
        assert (int(last[c, 0]), int(last[c, 1])) == (dt, fi), c
    assert kinds >= {1, 2, 4, 8} and good >= 16 and n_csd3 >= 1, (kinds, good, n_csd3)


def test_ysf_payload_through_the_chain_object(built):
    """the capture in four calls + flush: every frame's payload is decoded in the call that holds its last dibit, the frame type is
    carried from call to call - the same as processYSF() over the whole stream on the CPU"""
    from conftest import golden
    iq = np.ascontiguousarray(golden("iq_ysf.npz")["iq"], np.uint8)
    n = 60000
    calls = len(iq) // n
    B = 2
    x = np.stack([iq[:calls * n], np.roll(iq[:calls * n], 2 * 91)])
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_YSF, rf_mod=0, handlers=0, vocoder=0)
    l = ddn.lib()
    got = [[] for _ in range(B)]
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        S, T = r.max_syncs, r.carry_symbols
        assert T >= 461
        f = ch.fetch
        ns, pos, st = f(r.d_n_sync, np.int32, (B,)), f(r.d_sync_pos, np.int32, (B, S)), f(r.d_ysf_fich_status, np.uint8, (B, S))
        info, dch, dst = f(r.d_ysf_info2, np.uint8, (B, S, 2)), f(r.d_ysf_dch40, np.uint8, (B, S, 2, 20)), f(r.d_ysf_dch_status2, np.uint8, (B, S, 2))
        dcost, ambe, errs = f(r.d_ysf_dch_cost2, np.uint32, (B, S, 2)), f(r.d_ysf_ambe49x5, np.uint8, (B, S, 5, 49)), f(r.d_ysf_errs2x5, np.uint8, (B, S, 5))
        fr, nfr = f(r.d_ysf_frames184x5, np.uint8, (B, S, 5, 184)), f(r.d_ysf_n_frames, np.uint8, (B, S))
        new = f(r.d_new, np.int32, (B,))
        for c in range(B):
            for k in range(int(ns[c])):
                if st[c, k] != 0:
                    got[c].append(dict(pos=int(base[c]) + int(pos[c, k]) - int(T), info=info[c, k].copy(), dch=dch[c, k].copy(),
                                       dst=dst[c, k].copy(), dcost=dcost[c, k].copy(), ambe=ambe[c, k].copy(), errs=errs[c, k].copy(),
                                       fr=fr[c, k].copy(), nfr=int(nfr[c, k])))
            base[c] += int(new[c])

    for k in range(calls):
        part = np.ascontiguousarray(x[:, k * n:(k + 1) * n])
        p = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, part.ctypes.data, part.nbytes) == 0
        ch.run(p)
        take()
        l.ddn_device_free(p)
    ch.flush()
    take()
    ch.close()
    good = 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(x[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=4096)
        fr, _ = ysf.decode_payloads(want)
        assert [g["pos"] for g in got[c]] == [f["pos"] for f in fr], (c, len(got[c]), len(fr))
        for g, f in zip(got[c], fr):
            good += _payload_equal(g["info"], g["dch"], g["dst"], g["dcost"], g["ambe"], g["errs"], f, (c, f["pos"]), g["fr"], g["nfr"])
    assert good >= 16


def test_ysf_vd2_voice_to_pcm_through_the_chain_object(built):
    """vocoder = 1: the V/D mode 2 voice sub-frames of every channel go through AMBE 3600x2450 synthesis in stream order, talk path =
    channel, the history carried across calls - PCM and result rows bit for bit against the CPU vocoder restatement fed with the
    restated frames (the vocoder itself is unpinned: DESIGN 1, row a19)"""
    import mbe
    from conftest import golden
    iq = np.ascontiguousarray(golden("iq_ysf.npz")["iq"], np.uint8)
    n = 60000
    calls = len(iq) // n
    B = 2
    x = np.stack([iq[:calls * n], np.roll(iq[:calls * n], 2 * 137)])
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_YSF, rf_mod=0, handlers=0, vocoder=1)
    l = ddn.lib()
    got = [[] for _ in range(B)]
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        S, T, F = r.max_syncs, r.carry_symbols, r.ysf_voice_frames
        assert F >= 2
        f = ch.fetch
        pos, new = f(r.d_sync_pos, np.int32, (B, S)), f(r.d_new, np.int32, (B,))
        nv, slot = f(r.d_ysf_n_voice, np.int32, (B,)), f(r.d_ysf_voice_slot, np.int32, (B, F))
        res, pcm = f(r.d_ysf_voice_result, np.int32, (B, F * 5, 5)), f(r.d_ysf_pcm, np.float32, (B, F * 5, 160))
        for c in range(B):
            assert not pcm[c, 5 * nv[c]:].any()
            for j in range(int(nv[c])):
                got[c].append((int(base[c]) + int(pos[c, slot[c, j]]) - int(T), pcm[c, 5 * j:5 * j + 5].copy(), res[c, 5 * j:5 * j + 5].copy()))
            base[c] += int(new[c])

    for k in range(calls):
        part = np.ascontiguousarray(x[:, k * n:(k + 1) * n])
        p = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, part.ctypes.data, part.nbytes) == 0
        ch.run(p)
        take()
        l.ddn_device_free(p)
    ch.flush()
    take()
    ch.close()
    total = 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(x[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=4096)
        fr, _ = ysf.decode_payloads(want)
        voiced = [f for f in fr if f["payload"] is not None and f["payload"]["kind"] == 2]
        assert [g[0] for g in got[c]] == [f["pos"] for f in voiced], (c, len(got[c]), len(voiced))
        voc = mbe.OracleVocoder(ddn.MBE_AMBE, 1)
        for g, f in zip(got[c], voiced):
            bits = np.ascontiguousarray(f["payload"]["ambe_d"][None])
            ri = np.zeros((1, 5, 5), np.int32)
            ri[0, :, 3] = f["payload"]["errs2"]
            ri[0, :, 4] = f["payload"]["errs2"]
            pcm, ro = np.zeros((1, 5, 160), np.float32), np.zeros((1, 5, 5), np.int32)
            assert mbe._o().om_process_batch(ddn.MBE_AMBE, C.addressof(voc.tab), bits.ctypes.data, ri.ctypes.data, 0, c, 1, 5, pcm.ctypes.data,
                                             ro.ctypes.data, C.addressof(voc.cur), C.addressof(voc.prev), C.addressof(voc.enh)) == 0
            assert np.array_equal(g[1].view(np.uint32), pcm[0].view(np.uint32)), (c, f["pos"], float(np.abs(g[1] - pcm[0]).max()))
            assert np.array_equal(g[2], ro[0]), (c, f["pos"])
            total += 5
    assert total >= 200


def test_generated_frames_of_every_type_through_loop_fich_and_payload(built):
    """frames built by tests/ysfgen.py (header, V/D mode 1, V/D mode 2, full-rate voice plain and in the CSD3 layout, a data frame, a
    frame whose FICH is broken - read as the type before it -, terminator) as 4-level discriminator samples, two channels (the second
    inverted and delayed), in two calls: device loop -> FICH -> payload with the type carried = the CPU pipeline, frame for frame"""
    import torch
    import p25gen
    import ysfgen
    l = ddn.lib()
    rng = np.random.default_rng(31)
    plan = [(0, 1, {}), (1, 0, dict(fn=1, ft=6)), (1, 0, dict(fn=2, ft=6)), (1, 2, dict(fn=3, ft=6)), (1, 2, dict(fn=4, ft=6, break_fich=True)),
            (1, 3, dict(fn=0, ft=1)), (1, 3, dict(fn=1, ft=1)), (1, 3, dict(fn=2, ft=1, break_fich=True)), (1, 1, dict(fn=0, ft=0)),
            (1, 2, dict(fn=5, ft=6)), (2, 1, {})]
    dib = np.concatenate([ysfgen.frame(rng, fi, dt, **kw) for fi, dt, kw in plan] + [rng.integers(0, 4, 60).astype(np.uint8)])
    d0 = p25gen.modulate_disc(dib, lead=260, noise=250.0, seed=3)
    n = len(d0)
    x = np.zeros((2, n), np.float32)
    x[0] = d0
    x[1, 57:] = -d0[:n - 57]
    B = 2
    rx = ddn.Fsk4Rx(B, ddn.FSK4_YSF)
    cpu = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)) for _ in range(B)]
    last = torch.zeros((B, 2), dtype=torch.uint8, device="cuda")
    want_last = [(0, 0)] * B
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    p = lambda t: t.data_ptr()
    seen, csd3_seen, carried = set(), 0, 0
    for a, b in ((0, 30011), (30011, n)):          # (the cut lies inside a frame: that frame's payload is not in either call's records)
        part = np.ascontiguousarray(x[:, a:b])
        m = b - a
        d = torch.from_numpy(part).cuda()
        ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, m), l.ddn_fsk4_rx_max_syncs(rx.h, m)
        rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
        cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
        spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
        assert l.ddn_fsk4_rx_run(rx.h, p(d), m, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
        f4, st, ve = z((B, my, 4), torch.uint8), z((B, my), torch.uint8), z((B, my), torch.int32)
        assert l.ddn_ysf_fich_decode_batch(p(rec), ms, p(cnt), p(spos), p(ns), B, my, p(f4), p(st), p(ve), None) == 0
        info, dch, dst = z((B, my, 2), torch.uint8), z((B, my, 2, 20), torch.uint8), z((B, my, 2), torch.uint8)
        dcost, ambe, errs = z((B, my, 2), torch.int32), z((B, my, 5, 49), torch.uint8), z((B, my, 5), torch.uint8)
        fr, nfr = z((B, my, 5, 184), torch.uint8), z((B, my), torch.uint8)
        assert l.ddn_ysf_payload_decode_batch(p(rec), ms, p(cnt), p(spos), p(ns), B, my, p(f4), p(st), p(last), p(info), p(dch), p(dst),
                                              p(dcost), p(ambe), p(errs), p(fr), p(nfr), None) == 0, l.ddn_last_error()
        torch.cuda.synchronize()
        g = lambda t: t.cpu().numpy()
        st, info, dch, dst, dcost, ambe, errs, fr, nfr = g(st), g(info), g(dch), g(dst), g(dcost).view(np.uint32), g(ambe), g(errs), g(fr), g(nfr)
        nsy, pos = g(ns), g(spos)
        for c in range(B):
            want = cpu[c].run(part[c], max_sync=my)
            frames, want_last[c] = ysf.decode_payloads(want, last=want_last[c])
            mine = [k for k in range(int(nsy[c])) if st[c, k] != 0]
            assert [int(pos[c, k]) for k in mine] == [f["pos"] for f in frames], (c, a)
            for k, f in zip(mine, frames):
                _payload_equal(info[c, k], dch[c, k], dst[c, k], dcost[c, k], ambe[c, k], errs[c, k], f, (c, a, k), fr[c, k], nfr[c, k])
                seen.add(int(info[c, k, 0]))
                csd3_seen += int(info[c, k, 1]) >> 7
                carried += (int(info[c, k, 1]) >> 4) & 1
        assert [tuple(v) for v in g(last)] == want_last
    assert seen >= {1, 2, 4, 8} and csd3_seen >= 2 and carried >= 2, (seen, csd3_seen, carried)


def test_generated_voice_of_every_mode_to_pcm_through_the_chain_object(built):
    """generated frames as cu8 C4FM I/Q through the chain object with vocoder = 1, in three calls + flush: V/D mode 1 (four AMBE frames
    through the frame FEC) and V/D mode 2 (five sub-frames) share the AMBE talk path in stream order, full-rate voice (five IMBE frames,
    two in the CSD3 layout) goes down the IMBE one - positions, result rows and PCM bit for bit against the CPU pipeline + CPU vocoder"""
    import mbe
    import p25gen
    import ysfgen
    rng = np.random.default_rng(77)
    plan = [(0, 1, {}), (1, 0, dict(fn=1, ft=6)), (1, 2, dict(fn=2, ft=6)), (1, 0, dict(fn=3, ft=6)), (1, 2, dict(fn=4, ft=6, break_fich=True)),
            (1, 3, dict(fn=0, ft=1)), (1, 3, dict(fn=1, ft=1)), (1, 3, dict(fn=2, ft=1)), (1, 2, dict(fn=5, ft=6)), (1, 0, dict(fn=6, ft=6)),
            (2, 1, {})]
    dib = np.concatenate([ysfgen.frame(rng, fi, dt, **kw) for fi, dt, kw in plan] + [rng.integers(0, 4, 80).astype(np.uint8)])
    n = 20000
    calls = 3
    B = 2
    x = np.stack([p25gen.modulate_cu8(dib, calls * n, lead=230 + 170 * c, seed=5 + c) for c in range(B)])               # [B][calls * n][2]
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_YSF, rf_mod=0, handlers=0, vocoder=1)
    l = ddn.lib()
    got = {"a": [[] for _ in range(B)], "i": [[] for _ in range(B)]}
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        S, T, F = r.max_syncs, r.carry_symbols, r.ysf_voice_frames
        f = ch.fetch
        pos, new = f(r.d_sync_pos, np.int32, (B, S)), f(r.d_new, np.int32, (B,))
        for key, nv_p, slot_p, skip_p, res_p, pcm_p in (("a", r.d_ysf_n_voice, r.d_ysf_voice_slot, r.d_ysf_voice_skip, r.d_ysf_voice_result, r.d_ysf_pcm),
                                                        ("i", r.d_ysf_imbe_n_voice, r.d_ysf_imbe_voice_slot, r.d_ysf_imbe_voice_skip,
                                                         r.d_ysf_imbe_voice_result, r.d_ysf_imbe_pcm)):
            nv, slot, skip = f(nv_p, np.int32, (B,)), f(slot_p, np.int32, (B, F)), f(skip_p, np.uint8, (B, F * 5))
            res, pcm = f(res_p, np.int32, (B, F * 5, 5)), f(pcm_p, np.float32, (B, F * 5, 160))
            for c in range(B):
                assert not pcm[c, 5 * nv[c]:].any() and skip[c, 5 * nv[c]:].all()
                for j in range(int(nv[c])):
                    got[key][c].append((int(base[c]) + int(pos[c, slot[c, j]]) - int(T), pcm[c, 5 * j:5 * j + 5].copy(), res[c, 5 * j:5 * j + 5].copy(),
                                        skip[c, 5 * j:5 * j + 5].copy()))
        base[:] += new

    for k in range(calls):
        part = np.ascontiguousarray(x[:, k * n:(k + 1) * n])
        p = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, part.ctypes.data, part.nbytes) == 0
        ch.run(p)
        take()
        l.ddn_device_free(p)
    ch.flush()
    take()
    ch.close()
    total = {"a": 0, "i": 0}
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(x[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=4096)
        fr, _ = ysf.decode_payloads(want)
        for key, codec, kinds, shape in (("a", ddn.MBE_AMBE, (1, 2), (4, 24)), ("i", ddn.MBE_IMBE, (4,), (8, 23))):
            mine = [f for f in fr if f["payload"] is not None and f["payload"]["kind"] in kinds]
            assert [g[0] for g in got[key][c]] == [f["pos"] for f in mine], (key, c, len(got[key][c]), len(mine))
            voc = mbe.OracleVocoder(codec, 1)
            for g, f in zip(got[key][c], mine):
                pl = f["payload"]
                if pl["kind"] == 2:
                    nf, bits = 5, np.ascontiguousarray(pl["ambe_d"][None])
                    ri = np.zeros((1, 5, 5), np.int32)
                    ri[0, :, 3] = ri[0, :, 4] = pl["errs2"]
                else:
                    nf = pl["n_frames"]
                    frames = np.ascontiguousarray(pl["frames"][:nf, :shape[0] * shape[1]].reshape(nf, *shape))
                    bits, ri, rc = mbe.oracle_frame_decode(codec, frames)
                    bits, ri = np.ascontiguousarray(bits[None]), np.ascontiguousarray(ri[None])
                assert list(g[3]) == [0] * nf + [1] * (5 - nf), (key, c, f["pos"], g[3])
                pcm, ro = np.zeros((1, nf, 160), np.float32), np.zeros((1, nf, 5), np.int32)
                assert mbe._o().om_process_batch(codec, C.addressof(voc.tab), bits.ctypes.data, ri.ctypes.data, 0, c, 1, nf, pcm.ctypes.data,
                                                 ro.ctypes.data, C.addressof(voc.cur), C.addressof(voc.prev), C.addressof(voc.enh)) == 0
                assert np.array_equal(g[1][:nf].view(np.uint32), pcm[0].view(np.uint32)), (key, c, f["pos"])
                assert not g[1][nf:].any() and np.array_equal(g[2][:nf], ro[0]), (key, c, f["pos"])
                total[key] += nf
    assert total["a"] >= 2 * (3 * 4 + 2 * 5) - 9 and total["i"] >= 2 * (2 + 5), total
