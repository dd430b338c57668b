"""GPU: vocoder stage through the C-ABI (include/ddn_mbe.h) against the CPU restatement and the reference-held vectors.

Tolerances: frame FEC decode is integer -> bit-exact.  PCM: the device evaluates the same IEEE binary32 operations in the
same order as the restatement (fixed polynomials instead of libm, counter-based generator instead of rand()), so PCM and
the carried decoder history are compared BIT FOR BIT (tolerance 0) against the in-repo restatement; against mbelib-neo
itself the stage is parity unpinned (source absent) - the RMS figure BASELINE configs[4] asks for cannot be taken here."""
import ctypes as C

import numpy as np
import pytest

import ddn
import mbe

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def gpu_frame_decode(codec, frames, soft=None):
    import torch
    n = frames.shape[0]
    nb = 88 if codec == ddn.MBE_IMBE else 49
    d_f = _dev(frames)
    d_s = _dev(soft) if soft is not None else None
    d_b = torch.full((n, nb), 9, dtype=torch.uint8, device="cuda")
    d_r = torch.zeros((n, 5), dtype=torch.int32, device="cuda")
    assert ddn.lib().ddn_mbe_frame_decode_batch(codec, d_f.data_ptr(), d_s.data_ptr() if d_s is not None else None, n,
                                                d_b.data_ptr(), d_r.data_ptr(), None) == 0
    torch.cuda.synchronize()
    return d_b.cpu().numpy(), d_r.cpu().numpy()


class GpuVocoder:
    def __init__(self, codec, S, tail_rule=0):
        self.h = C.c_void_p()
        assert ddn.lib().ddn_mbe_batch_create(codec, S, C.byref(self.h)) == 0
        assert ddn.lib().ddn_mbe_batch_set_p25p1_tail_rule(self.h, tail_rule) == 0
        self.S, self.codec = S, codec

    def run(self, bits, res_in=None):
        import torch
        S, F = bits.shape[0], bits.shape[1]
        d_b = _dev(bits)
        d_ri = _dev(res_in.astype(np.int32)) if res_in is not None else None
        d_p = torch.full((S, F, 160), 7.0, dtype=torch.float32, device="cuda")
        d_ro = torch.zeros((S, F, 5), dtype=torch.int32, device="cuda")
        assert ddn.lib().ddn_mbe_synth_batch(self.h, d_b.data_ptr(), d_ri.data_ptr() if d_ri is not None else None, F,
                                             d_p.data_ptr(), d_ro.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return d_p.cpu().numpy(), d_ro.cpu().numpy()

    def state(self, s):
        c, p, e = ddn.MbeParms(), ddn.MbeParms(), ddn.MbeParms()
        assert ddn.lib().ddn_mbe_batch_get_state(self.h, s, C.byref(c), C.byref(p), C.byref(e)) == 0
        return c, p, e

    def __del__(self):
        if self.h:
            ddn.lib().ddn_mbe_batch_destroy(self.h)


def test_imbe_reference_held_vectors_batch_and_dropin(built):
    """tests/core/test_core_mbe_transform_context.c:134-152 through ddn_mbe_frame_decode_batch and through the mbelib-neo
    named entry points, hard and soft (all-255 reliabilities, :606-613)."""
    kat = mbe.load_kat()
    frames = np.stack([k["frame"] for k in kat])
    bits, res = gpu_frame_decode(ddn.MBE_IMBE, frames)
    l = ddn.lib()
    for i, k in enumerate(kat):
        hx = mbe.bits_hex(bits[i])
        assert res[i, 3] == k["total_errors"] and hx.startswith(k.get("imbe_d_hex", k.get("imbe_d_hex_prefix", ""))), k["name"]
        out = np.zeros(88, np.uint8)
        r = ddn.MbeProcessResult()
        assert l.mbe_decodeImbe7200x4400Frame(k["frame"].ctypes.data, out.ctypes.data, C.byref(r)) == 0
        assert np.array_equal(out, bits[i]) and r.total_errors == k["total_errors"] and r.flags & 1
        soft = np.stack([k["frame"], np.full((8, 23), 255, np.uint8)], axis=-1).astype(np.uint8)   # {bit, reliability}
        out2 = np.zeros(88, np.uint8)
        r2 = ddn.MbeProcessResult()
        assert l.mbe_decodeImbe7200x4400SoftFrame(np.ascontiguousarray(soft).ctypes.data, out2.ctypes.data, C.byref(r2)) == 0
        assert np.array_equal(out2, bits[i]) and r2.total_errors == k["total_errors"] and r2.flags & 4
        assert r2.protected_errors == r2.total_errors - r2.c0_errors
    bad = kat[0]["frame"].copy()
    bad[2, 2] = 3
    assert l.mbe_decodeImbe7200x4400Frame(bad.ctypes.data, np.zeros(88, np.uint8).ctypes.data, None) == -2


@pytest.mark.parametrize("codec,n", [(ddn.MBE_IMBE, 8192), (ddn.MBE_IMBE, 77), (ddn.MBE_AMBE, 8192), (ddn.MBE_AMBE, 1)])
def test_frame_decode_batch_bit_exact(built, codec, n):
    """random code words with 0..4 flipped bits per word (beyond t too: mis-corrections must match as well)."""
    rng = np.random.default_rng(100 + n + codec)
    if codec == ddn.MBE_IMBE:
        frames = np.stack([mbe.imbe_encode(d) for d in rng.integers(0, 2, size=(min(n, 256), 88), dtype=np.uint8)])
    else:
        frames = np.stack([mbe.ambe_encode(d) for d in rng.integers(0, 2, size=(min(n, 256), 49), dtype=np.uint8)])
    frames = np.tile(frames, (n // frames.shape[0] + 1, 1, 1))[:n].copy()
    flips = rng.random(frames.shape) < 0.035
    frames ^= flips.astype(np.uint8)
    if n > 10:
        frames[5, 0, 0] = 2                                 # one invalid frame in the middle of a batch
    want_bits, want_res, rc = mbe.oracle_frame_decode(codec, frames)
    got_bits, got_res = gpu_frame_decode(codec, frames)
    ok = rc == 0
    assert np.array_equal(got_bits[ok], want_bits[ok]) and np.array_equal(got_res[ok], want_res[ok])
    assert np.all(got_res[~ok, 0].view(np.uint32) == 0x80000000) and np.all(got_bits[~ok] == 0)


@pytest.mark.parametrize("codec", [ddn.MBE_IMBE, ddn.MBE_AMBE])
def test_synth_batch_bit_exact_with_history(built, codec):
    """S talk paths x F frames in two calls (history carried on the device), error counts driving repeat / mute, special
    frames mixed in: PCM, result flags and the {cur, prev, prev_enhanced} triple equal the restatement bit for bit."""
    rng = np.random.default_rng(7 + codec)
    S, F = 24, 40
    gen = mbe.random_imbe_bits if codec == ddn.MBE_IMBE else mbe.random_ambe_bits
    bits = gen(rng, (S, F), 215 if codec == ddn.MBE_IMBE else 127)        # a few invalid / special fundamentals too
    res_in = np.zeros((S, F, 5), np.int32)
    res_in[..., 3] = rng.choice([0, 1, 2, 4, 6, 9], size=(S, F), p=[0.5, 0.15, 0.1, 0.1, 0.1, 0.05])
    res_in[3, 10:20, 3] = 7                                                 # a run of repeats -> mute
    res_in[..., 1] = np.minimum(res_in[..., 3], 3)
    res_in[..., 4] = res_in[..., 3] - res_in[..., 1]
    res_in[..., 0] = 1
    o = mbe.OracleVocoder(codec, S)
    g = GpuVocoder(codec, S)
    seen = 0
    for (a, b) in ((0, 25), (25, F)):
        want_pcm, want_res, rc = o.run(bits[:, a:b], res_in[:, a:b])
        assert rc == 0
        seen |= int(np.bitwise_or.reduce(want_res[..., 0].reshape(-1)))
        got_pcm, got_res = g.run(bits[:, a:b], res_in[:, a:b])
        assert np.array_equal(got_res, want_res)
        assert np.array_equal(got_pcm.view(np.uint32), want_pcm.view(np.uint32)), float(np.abs(got_pcm - want_pcm).max())
        assert np.all(np.isfinite(got_pcm)) and np.abs(got_pcm).max() > 0
        for s in (0, 3, S - 1):
            c, p, e = g.state(s)
            assert mbe.parms_equal(c, o.cur[s]) and mbe.parms_equal(p, o.prev[s]) and mbe.parms_equal(e, o.enh[s]), s
    assert (seen & 0x10) and (seen & 0x8)    # the case exercised repeat and mute


def test_imbe_invalid_fundamental_repeats_three_times_then_mutes(built):
    """mbelib 1.3 mbe_processImbe4400Dataf: b0 > 207 frames repeat the last good frame (synthesized) up to three times in a row, the
    fourth mutes and re-initialises the talk path; flags, PCM and state equal the restatement"""
    rng = np.random.default_rng(77)
    S, F = 5, 12
    bits = mbe.random_imbe_bits(rng, (S, F))
    bits[0, 2:6, :6] = 1
    bits[1, 3:5, :6] = 1
    bits[2, 1:9, :6] = 1
    bits[3, 0:2, :6] = 1
    o = mbe.OracleVocoder(ddn.MBE_IMBE, S)
    g = GpuVocoder(ddn.MBE_IMBE, S)
    want_pcm, want_res, rc = o.run(bits)
    got_pcm, got_res = g.run(bits)
    assert rc == 0 and np.array_equal(got_res, want_res)
    assert np.array_equal(got_pcm.view(np.uint32), want_pcm.view(np.uint32))
    REPEAT, MUTE = 0x8, 0x10
    assert [int(f) & 0x18 for f in got_res[0, :8, 0]] == [0, 0, REPEAT, REPEAT, REPEAT, REPEAT | MUTE, 0, 0]
    assert all(np.abs(got_pcm[0, k]).max() > 0 for k in (2, 3, 4)) and np.all(got_pcm[0, 5] == 0)
    assert [int(f) & 0x18 for f in got_res[1, 2:6, 0]] == [0, REPEAT, REPEAT, 0]
    for s in range(S):
        c, p, e = g.state(s)
        assert mbe.parms_equal(c, o.cur[s]) and mbe.parms_equal(p, o.prev[s]) and mbe.parms_equal(e, o.enh[s]), s


def test_c5_shape_8192_frames_properties(built):
    """BASELINE configs[4] / SURVEY §8d C5: 8192 voice frames as 64 talk paths x 128 frames.  A sample of talk paths is
    compared with the restatement; the whole batch is checked through size-independent properties: a talk path's output
    does not depend on which batch slot neighbours it, and splitting the call leaves the PCM unchanged."""
    rng = np.random.default_rng(2024)
    S, F = 64, 128
    bits = mbe.random_imbe_bits(rng, (S, F))
    g = GpuVocoder(ddn.MBE_IMBE, S)
    pcm, res = g.run(bits)
    assert np.all(np.isfinite(pcm)) and not (res[..., 0] & 0x10).any()
    # the generator is keyed by the talk path's index: run the restatement for three slots, each with its own seed
    for s in (0, 17, 63):
        v = mbe.OracleVocoder(ddn.MBE_IMBE, 1)
        want = np.zeros((F, 160), np.float32)
        rc = mbe._o().om_process_batch(ddn.MBE_IMBE, C.addressof(v.tab), np.ascontiguousarray(bits[s]).ctypes.data, None, 0, s, 1, F,
                                       want.ctypes.data, None, C.addressof(v.cur), C.addressof(v.prev), C.addressof(v.enh))
        assert rc == 0 and np.array_equal(want.view(np.uint32), pcm[s].view(np.uint32)), s
    g2 = GpuVocoder(ddn.MBE_IMBE, S)
    a, _ = g2.run(bits[:, :50])
    b, _ = g2.run(bits[:, 50:])
    assert np.array_equal(np.concatenate([a, b], axis=1).view(np.uint32), pcm.view(np.uint32))


def test_dropin_process_matches_batch_and_tail_rule(built):
    """mbe_processImbe4400Dataf / mbe_processAmbe2450Dataf with a caller-owned mbe_parms triple reproduce a one-talk-path
    batch frame by frame; ddn_mbe_batch_set_p25p1_tail_rule mutes the teardown frame without touching the history
    (dsd_mbe.c:447-463,540-566; test_core_mbe_transform_context.c:847-873)."""
    l = ddn.lib()
    rng = np.random.default_rng(5)
    for codec, fn, gen in ((ddn.MBE_IMBE, l.mbe_processImbe4400Dataf, mbe.random_imbe_bits),
                           (ddn.MBE_AMBE, l.mbe_processAmbe2450Dataf, mbe.random_ambe_bits)):
        bits = gen(rng, (1, 6))
        want, _, _ = mbe.OracleVocoder(codec, 1).run(bits)
        c, p, e = ddn.MbeParms(), ddn.MbeParms(), ddn.MbeParms()
        l.mbe_initMbeParms(C.byref(c), C.byref(p), C.byref(e))
        for f in range(6):
            out = np.full(160, 5.0, np.float32)
            r = ddn.MbeProcessResult()
            assert fn(out.ctypes.data, C.byref(r), np.ascontiguousarray(bits[0, f]).ctypes.data, C.byref(c), C.byref(p), C.byref(e)) == 0
            assert np.array_equal(out.view(np.uint32), want[0, f].view(np.uint32)), (codec, f)
        assert c.un == 6
    kat = {k["name"]: k for k in mbe.load_kat()}
    fb, fr = gpu_frame_decode(ddn.MBE_IMBE, np.stack([kat["p25p1_corrected_speech"]["frame"], kat["p25p1_tail_erasure"]["frame"],
                                                      kat["p25p1_dense_fc"]["frame"]]))
    g = GpuVocoder(ddn.MBE_IMBE, 1, tail_rule=1)
    o = mbe.OracleVocoder(ddn.MBE_IMBE, 1, tail_rule=1)
    pcm, res = g.run(fb[None], fr[None])
    wpcm, wres, _ = o.run(fb[None], fr[None])
    assert np.array_equal(pcm.view(np.uint32), wpcm.view(np.uint32)) and np.array_equal(res, wres)
    assert np.all(pcm[0, 1] == 0) and np.all(res[0, 1] == 0) and res[0, 2, 3] == 12
    assert g.state(0)[0].un == 2                               # the muted teardown frame did not advance the history
    buf = C.create_string_buffer(64)
    r = ddn.MbeProcessResult(flags=0x8, total_errors=3)
    l.mbe_formatProcessResult(buf, 64, C.byref(r))
    assert buf.value == b"===R"
    sil = np.ones(160, np.float32)
    l.mbe_synthesizeSilencef(sil.ctypes.data)
    assert np.all(sil == 0)


def test_dropin_table_blob_is_loadable(built):
    """ddn_mbe_dropin_set_tables: an integrator's blob (here the default with the IMBE gain levels halved and an AMBE gain
    delta changed) reaches the single-stream mbe_* entry points - PCM equals the CPU restatement run on the same blob and
    differs from the default blob's; a blob that fails validation is refused and leaves the loaded one in place."""
    l = ddn.lib()
    rng = np.random.default_rng(17)
    t = mbe.tables()
    for i in range(64):
        t.imbe_gain_b2[i] *= 0.5
    for i in range(32):
        t.ambe_dg[i] += 0.25
    t.synthetic = 0
    try:
        assert l.ddn_mbe_dropin_set_tables(C.byref(t)) == 0
        bad = mbe.tables()
        bad.magic = 0
        assert l.ddn_mbe_dropin_set_tables(C.byref(bad)) != 0
        for codec, fn, gen in ((ddn.MBE_IMBE, l.mbe_processImbe4400Dataf, mbe.random_imbe_bits),
                               (ddn.MBE_AMBE, l.mbe_processAmbe2450Dataf, mbe.random_ambe_bits)):
            bits = gen(rng, (1, 4))
            want, _, _ = mbe.OracleVocoder(codec, 1, tab=t).run(bits)
            dflt, _, _ = mbe.OracleVocoder(codec, 1).run(bits)
            assert not np.array_equal(want, dflt)
            c, p, e = ddn.MbeParms(), ddn.MbeParms(), ddn.MbeParms()
            l.mbe_initMbeParms(C.byref(c), C.byref(p), C.byref(e))
            for f in range(4):
                out = np.zeros(160, np.float32)
                r = ddn.MbeProcessResult()
                assert fn(out.ctypes.data, C.byref(r), np.ascontiguousarray(bits[0, f]).ctypes.data, C.byref(c), C.byref(p), C.byref(e)) == 0
                assert np.array_equal(out.view(np.uint32), want[0, f].view(np.uint32)), (codec, f)
    finally:
        d = mbe.tables()
        assert l.ddn_mbe_dropin_set_tables(C.byref(d)) == 0


def test_table_blob_file_flips_the_synthetic_flag(built, tmp_path):
    """ddn_mbe_batch_tables_synthetic: 1 on the built-in placeholder blob, 0 once a synthetic = 0 blob file is loaded - and the loaded
    tables are the ones synthesis runs on (PCM follows the changed gain table, state and flags stay those of the restatement)"""
    l = ddn.lib()
    g = GpuVocoder(ddn.MBE_AMBE, 2)
    assert l.ddn_mbe_batch_tables_synthetic(g.h) == 1
    t = mbe.tables()
    t2 = ddn.MbeTables.from_buffer_copy(bytes(t))
    t2.synthetic = 0
    for k in range(32):
        t2.ambe_dg[k] = t.ambe_dg[k] * 0.5
    path = str(tmp_path / "tables.ddnmbet").encode()
    assert l.ddn_mbe_tables_save_file(path, C.byref(t2)) == 0
    rng = np.random.default_rng(9)
    bits = mbe.random_ambe_bits(rng, (2, 6))
    before, _ = g.run(bits)
    g2 = GpuVocoder(ddn.MBE_AMBE, 2)
    assert l.ddn_mbe_batch_load_tables_file(g2.h, path) == 0 and l.ddn_mbe_batch_tables_synthetic(g2.h) == 0
    after, res = g2.run(bits)
    o = mbe.OracleVocoder(ddn.MBE_AMBE, 2, tab=t2)
    want, want_res, rc = o.run(bits)
    assert rc == 0 and np.array_equal(after.view(np.uint32), want.view(np.uint32)) and np.array_equal(res, want_res)
    assert not np.array_equal(after, before)
    assert l.ddn_mbe_batch_load_tables_file(g2.h, b"/nonexistent/tables") != 0 and l.ddn_mbe_batch_tables_synthetic(g2.h) == 0


def test_agf_batch_bit_exact(built):
    """the voice-frame auto gain on the device == the CPU restatement (itself pinned to the compiled gain.c): samples and the
    carried gain state, several talk paths, across two calls"""
    from test_oracle_audio import oracle_agf, voice_like
    l = ddn.lib()
    rng = np.random.default_rng(6)
    S, F = 70, 24
    pcm = voice_like(rng, S, F)
    for audio_gain, a21 in ((0.0, 0), (30.0, 0), (0.0, 1)):
        g_cpu = np.full(S, 25.0, np.float32)
        g_gpu = g_cpu.copy()
        for part in (slice(0, 10), slice(10, F)):
            want, g_cpu = oracle_agf(pcm[:, part], g_cpu, audio_gain, a21)
            got = np.ascontiguousarray(pcm[:, part]).copy()
            assert l.ddn_audio_agf_host(got.ctypes.data, S, got.shape[1], audio_gain, a21, g_gpu.ctypes.data) == 0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            assert np.array_equal(g_gpu, g_cpu)
    one = np.ascontiguousarray(pcm[3, 0]).copy()
    g = np.array([25.0], np.float32)
    assert l.ddn_agf_frame(one.ctypes.data, 0.0, 0, g.ctypes.data) == 0
    w, gw = oracle_agf(pcm[3:4, 0:1], np.array([25.0], np.float32))
    assert np.array_equal(one.view(np.uint32), w[0, 0].view(np.uint32)) and g[0] == gw[0]


@pytest.mark.parametrize("audio_gain,hpf,agsm", [(0.0, 1, 0), (0.0, 0, 0), (-1.0, 1, 1), (20.0, 1, 0), (0.0, 1, 1)])
def test_audio_s16_batch_bit_exact(built, audio_gain, hpf, agsm):
    """the short-integer voice path (processAudio -> hpf_dL -> agsm) on 70 talk paths (a full wave + a ragged one), in two
    calls so the carried state (gain, peak history, filter memory) is exercised, against the restatement"""
    import ddn
    from test_oracle_audio import oracle_s16, s16_state, voice_like
    l = ddn.lib()
    rng = np.random.default_rng(77)
    S, F = 70, 45
    pcm = voice_like(rng, S, F)
    pcm[3] *= 0.02
    st_cpu = s16_state(S)
    st_cpu[:, 0] = rng.choice([25.0, 7.0, 50.0], S)
    st_gpu = np.zeros((S, 32), np.float32)
    assert l.ddn_audio_s16_state_init(st_gpu.ctypes.data, S) == 0 and np.array_equal(st_gpu, s16_state(S))
    st_gpu[:, 0] = st_cpu[:, 0]
    ga_gpu = np.zeros(S, np.float32)
    for part in (slice(0, 17), slice(17, F)):
        x = np.ascontiguousarray(pcm[:, part])
        want, st_cpu, ga_cpu = oracle_s16(x, st_cpu, audio_gain, hpf, agsm)
        got = np.zeros(x.shape, np.int16)
        assert l.ddn_audio_s16_host(x.ctypes.data, S, x.shape[1], audio_gain, hpf, agsm, got.ctypes.data, st_gpu.ctypes.data,
                                    ga_gpu.ctypes.data) == 0
        assert np.array_equal(got, want)
        assert np.array_equal(st_gpu.view(np.uint32), st_cpu.view(np.uint32))
        if agsm:
            assert np.array_equal(ga_gpu, ga_cpu)
    assert np.abs(want.astype(np.int32)).max() > 1000
    assert l.ddn_audio_s16_host(None, S, 1, 0.0, 1, 0, None, None, None) != 0
