"""GPU: AMBE 3600x2450 voice-frame gathers (ddn_ambe2450_deinterleave_batch, ddn_nxdn_voice_gather, ddn_dmr_voice_burst_gather)
against the python restatement of the reference's unpacking (tests/rx4.py), and the reference's NXDN48 capture taken all the
way to voice: receive loop -> de-scramble + de-interleave -> AMBE frame FEC (Golay words clean on real traffic) -> synthesis."""
import numpy as np
import pytest

import ddn
import rx4

pytestmark = pytest.mark.gpu


def test_ambe2450_schedule_matches_the_compiled_reference(built):
    import ctypes as C
    import orc
    m = rx4.ambe2450_map()
    cells = {(int(a), int(b)) for a, b, _, _ in m} | {(int(c), int(d)) for _, _, c, d in m}
    assert len(cells) == 72
    if orc.have_ref():           # (the build container; the GPU box carries the generated header only)
        r = C.CDLL(orc.REF_SO)
        if hasattr(r, "refh_ambe2450_map"):
            r.refh_ambe2450_map.argtypes = [C.c_int, C.c_void_p]
            w = np.zeros(4, np.int32)
            for i in range(36):
                assert r.refh_ambe2450_map(i, w.ctypes.data) == 0 and np.array_equal(w, m[i])


def test_deinterleave_and_burst_gathers_equal_the_restatement(built):
    import torch
    l = ddn.lib()
    rng = np.random.default_rng(5)
    p = lambda t: t.data_ptr()
    # plain frames
    n = 300
    dib = rng.integers(0, 4, (n, 36)).astype(np.uint8)
    rel = rng.integers(0, 256, (n, 36)).astype(np.uint8)
    d_fr = torch.full((n, 4, 24), 7, dtype=torch.uint8, device="cuda")
    d_rl = torch.full((n, 4, 24), 7, dtype=torch.uint8, device="cuda")
    assert l.ddn_ambe2450_deinterleave_batch(p(torch.from_numpy(dib).cuda()), p(torch.from_numpy(rel).cuda()), n, p(d_fr), p(d_rl), None) == 0
    fr, rl = d_fr.cpu().numpy(), d_rl.cpu().numpy()
    for i in range(n):
        w = rx4.ambe2450_deinterleave(dib[i], rel[i])
        assert np.array_equal(fr[i], w[0]) and np.array_equal(rl[i], w[1])
    # DMR voice bursts and NXDN voice frames out of random records (ragged counts, slots past the end)
    B, ms, mb = 5, 900, 7
    rec = rng.integers(0, 256, (B, ms, 10)).astype(np.uint8)
    cnt = np.array([900, 650, 143, 400, 0], np.int32)
    start = rng.integers(-20, 800, (B, mb)).astype(np.int32)
    start[0, 0], start[1, 1] = 900 - 144, 650 - 143      # last one that fits / first that does not
    d_rec, d_cnt, d_start = torch.from_numpy(rec).cuda(), torch.from_numpy(cnt).cuda(), torch.from_numpy(start).cuda()
    for inv in (0, 1):
        o_fr = torch.full((B * mb, 3, 4, 24), 9, dtype=torch.uint8, device="cuda")
        o_rl = torch.full((B * mb, 3, 4, 24), 9, dtype=torch.uint8, device="cuda")
        o_sy, o_ca = torch.zeros((B * mb, 48), dtype=torch.uint8, device="cuda"), torch.zeros((B * mb, 24), dtype=torch.uint8, device="cuda")
        o_v = torch.zeros((B * mb,), dtype=torch.uint8, device="cuda")
        assert l.ddn_dmr_voice_burst_gather(p(d_rec), p(d_cnt), ms, p(d_start), B, mb, inv, p(o_fr), p(o_rl), p(o_sy), p(o_ca), p(o_v), None) == 0
        fr, rl, sy, ca, v = (t.cpu().numpy() for t in (o_fr, o_rl, o_sy, o_ca, o_v))
        seen = 0
        for c in range(B):
            for k in range(mb):
                s, so = int(start[c, k]), c * mb + k
                ok = s >= 0 and s + 144 <= cnt[c]
                assert v[so] == ok
                if ok:
                    w = rx4.dmr_voice_burst_fields(rec[c, s:s + 144, 0] & 3, rec[c, s:s + 144, 1], inv)
                    assert np.array_equal(fr[so], w[0]) and np.array_equal(rl[so], w[1]) and np.array_equal(sy[so], w[2]) and np.array_equal(ca[so], w[3])
                    seen += 1
                else:
                    assert not fr[so].any() and not rl[so].any()
        assert seen >= 8
    my = 6
    spos = np.sort(rng.integers(0, 700, (B, my)).astype(np.int32), axis=1)
    nsy = np.array([6, 4, 1, 6, 0], np.int32)
    o_fr = torch.full((B * my, 4, 4, 24), 9, dtype=torch.uint8, device="cuda")
    o_rl = torch.full((B * my, 4, 4, 24), 9, dtype=torch.uint8, device="cuda")
    o_v = torch.zeros((B * my,), dtype=torch.uint8, device="cuda")
    assert l.ddn_nxdn_voice_gather(p(d_rec), p(d_cnt), ms, p(torch.from_numpy(spos).cuda()), p(torch.from_numpy(nsy).cuda()), B, my,
                                   p(o_fr), p(o_rl), p(o_v), None) == 0
    fr, rl, v = (t.cpu().numpy() for t in (o_fr, o_rl, o_v))
    seen = 0
    for c in range(B):
        for k in range(my):
            so, pos = c * my + k, int(spos[c, k])
            ok = k < nsy[c] and pos + 182 < cnt[c]
            assert v[so] == ok
            if ok:
                w = rx4.nxdn_voice_frames(rec[c, pos + 1:pos + 183, 0] & 3, rec[c, pos + 1:pos + 183, 1])
                assert np.array_equal(fr[so], w[0]) and np.array_equal(rl[so], w[1])
                seen += 1
            else:
                assert not fr[so].any()
    assert seen >= 6


def test_nxdn48_capture_voice_frames_decode_clean(built):
    """The reference's NXDN48 capture is a voice call (VCALL, Src=901: tests/test_rx4_gpu.py).  Its voice frames, as the LICH
    announces them, through de-scramble + AMBE de-interleave + the AMBE 3600x2450 frame FEC: on real traffic a wrong
    interleave or scrambler phase would show as Golay corrections in nearly every frame; here most frames need none, and the
    synthesised PCM is not silence.  (Quantiser tables are the library's default blob: the waveform itself is unpinned, DESIGN
    5d; the FEC words are not.)"""
    import torch
    l = ddn.lib()
    disc = rx4.capture_disc("iq_nxdn48.npz", 1)
    B, n = 1, len(disc)
    x = torch.from_numpy(disc[None, :].copy()).cuda()
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
    a_fr, a_rl, a_v = z((S, 4, 4, 24), u8), z((S, 4, 4, 24), u8), z((S,), u8)
    assert l.ddn_nxdn_voice_gather(p(rec), p(cnt), ms, p(spos), p(ns), B, my, p(a_fr), p(a_rl), p(a_v), None) == 0
    V = S * 4
    bits, res = z((V, 49), u8), z((V, 5), i32)
    assert l.ddn_mbe_frame_decode_batch(ddn.MBE_AMBE, p(a_fr), p(a_rl), V, p(bits), p(res), None) == 0
    torch.cuda.synchronize()
    lich_h, valid_h, res_h, ns_h = lich.cpu().numpy(), valid.cpu().numpy(), res.cpu().numpy().reshape(S, 4, 5), int(ns.cpu().numpy()[0])
    tot, frames = [], []
    for k in range(ns_h):
        if not valid_h[k] or not (lich_h[k] & 0x80):
            continue
        vo = rx4.nxdn_lich_voice(int(lich_h[k] & 0x7F))
        for v in range(4):
            if (vo == 3) or (vo == 1 and v < 2) or (vo == 2 and v >= 2):
                tot.append(int(res_h[k, v, 3]))
                frames.append(k * 4 + v)
    tot = np.array(tot)
    assert len(tot) >= 100, len(tot)
    assert np.mean(tot == 0) > 0.6 and np.mean(tot <= 2) > 0.9, (np.mean(tot == 0), np.mean(tot <= 2), np.bincount(tot)[:8])
    # the frames the LICH does not announce as voice (FACCH1 payload) look nothing like codewords
    other = np.array([int(res_h[k, v, 3]) for k in range(ns_h) if valid_h[k] and (lich_h[k] & 0x80) and rx4.nxdn_lich_voice(int(lich_h[k] & 0x7F)) == 0
                      for v in range(4)])
    if other.size >= 8:
        assert np.mean(other) > np.mean(tot) + 2
    # one talk path: the voice frames in order through the synthesiser
    idx = torch.tensor(frames, dtype=torch.long, device="cuda")
    vb, vr = bits[idx].contiguous(), res[idx].contiguous()
    nv = len(frames)
    import ctypes as C
    mbe = C.c_void_p()
    assert l.ddn_mbe_batch_create(ddn.MBE_AMBE, 1, C.byref(mbe)) == 0
    pcm, rout = z((1, nv, 160), torch.float32), z((nv, 5), i32)
    assert l.ddn_mbe_synth_batch(mbe, p(vb), p(vr), nv, p(pcm), p(rout), None) == 0
    torch.cuda.synchronize()
    w = pcm.cpu().numpy().reshape(-1)
    assert np.isfinite(w).all() and np.abs(w).max() > 0 and np.mean(np.abs(w) > 1e-3) > 0.3


def test_nxdn_chain_voice_stage_on_the_capture(built):
    """Fsk4Chain (bindings/ddn_chain_fsk4.py) on the NXDN48 capture from cu8 I/Q: the voice stage mutes the frames the LICH
    does not announce, the announced ones decode with few corrections and the talk path's PCM is audio."""
    import torch
    import ddn_chain_fsk4
    from conftest import golden
    g = golden("iq_nxdn48.npz")
    iq = np.ascontiguousarray(g["iq"], np.uint8).reshape(1, -1, 2)
    n = iq.shape[1]
    ch = ddn_chain_fsk4.Fsk4Chain(torch, 1, n, ddn.FSK4_NXDN48)
    ch.run(torch.from_numpy(iq).cuda())
    torch.cuda.synchronize()
    skip = ch.voice_skip.cpu().numpy().astype(bool)
    res = ch.ambe_res.cpu().numpy()
    assert (~skip).sum() >= 100
    tot = res[~skip, 3]
    assert np.mean(tot <= 2) > 0.9
    assert np.all(res[skip, 0].view(np.uint32) & 0x80000000)          # muted rows carry the skip flag
    assert int(ch.ns.cpu().numpy()[0]) <= ch.vf                      # the voice stage's slot bound holds on real traffic
    pcm = ch.pcm.cpu().numpy().reshape(-1, 160)
    assert np.isfinite(pcm).all() and np.abs(pcm[~skip]).max() > 0 and not pcm[skip].any()
