"""f4 on the CPU: the voice-frame auto gain restatement against the compiled reference (gain.c), and the two output-file
writers (symbol capture, WAV) - host code, no GPU."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import ddn
import orc


def oracle_agf(pcm, aout, audio_gain=0.0, algid21=0):
    o = orc.oracle()
    o.orc_agf.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    o.orc_agf.restype = None
    pcm = np.ascontiguousarray(pcm, np.float32).copy()
    g = np.ascontiguousarray(aout, np.float32).copy()
    for s in range(pcm.shape[0]):
        o.orc_agf(pcm[s].ctypes.data, pcm.shape[1], audio_gain, algid21, g[s:s + 1].ctypes.data)
    return pcm, g


def voice_like(rng, S, F):
    """int16-scale float frames with level changes, silences and hot passages"""
    t = np.arange(F * 160)
    out = np.zeros((S, F * 160), np.float32)
    for s in range(S):
        env = 3000.0 * (1.0 + np.sin(t / (900.0 + 77 * s))) * (rng.random() * 4 + 0.1)
        x = env * np.sin(t * (0.05 + 0.01 * s)) + rng.standard_normal(t.size) * 200
        x[(t // 480) % 7 == 3] = 0.0
        x[(t // 800) % 11 == 5] *= 40.0
        out[s] = x
    return out.reshape(S, F, 160)


@pytest.mark.skipif(not orc.have_ref(), reason="needs oracle/_ref")
@pytest.mark.parametrize("audio_gain,algid21", [(0.0, 0), (0.0, 1), (37.0, 0), (12.5, 1)])
def test_agf_restatement_equals_compiled_reference(built, audio_gain, algid21):
    r = orc.ref()
    r.refh_agf_run.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    r.refh_agf_run.restype = None
    rng = np.random.default_rng(2)
    pcm = voice_like(rng, 5, 40)
    for g0 in (25.0, 1.0, 46.0, 10.5):
        got, gg = oracle_agf(pcm, np.full(5, g0, np.float32), audio_gain, algid21)
        for s in range(5):
            want = pcm[s].copy()
            g = np.array([g0], np.float32)
            r.refh_agf_run(want.ctypes.data, 40, audio_gain, algid21, g.ctypes.data)
            assert np.array_equal(got[s].view(np.uint32), want.view(np.uint32)), (g0, s)
            assert gg[s] == g[0]
    assert np.abs(got).max() <= 0.9 * max(audio_gain / 25.0 if audio_gain else (1.75 if algid21 else 1.0), 1e-9) * 0.8 + 1e-6


def test_symbol_capture_file(built, tmp_path):
    """header + records as dsd-neo's -c file: in-frame records verbatim, hunting records with the fallback soft decision of their
    sign dibit (reliability 255, LLR +-255) - src/core/file/dsd_file.c:876-890, src/core/frames/dsd_dibit.c:592-602,794-818"""
    rng = np.random.default_rng(1)
    n = 300
    rec = rng.integers(0, 256, (n, 10), dtype=np.uint8)
    fl = (rng.random(n) < 0.6).astype(np.uint8)
    rec[fl == 0, 0] = rng.choice([1, 3], int((fl == 0).sum()))
    rec[fl == 0, 1:6] = 0
    path = str(tmp_path / "cap.bin").encode()
    l = ddn.lib()
    assert l.ddn_symbol_capture_write(path, rec.ctypes.data, fl.ctypes.data, 200, 0) == 0
    assert l.ddn_symbol_capture_write(path, rec[200:].ctypes.data, fl[200:].ctypes.data, 100, 1) == 0
    raw = open(path, "rb").read()
    assert raw[:16] == b"DSDNSYM2" + bytes([2, 10, 0, 0, 0, 0, 0, 0]) and len(raw) == 16 + 10 * n
    got = np.frombuffer(raw[16:], np.uint8).reshape(n, 10)
    assert np.array_equal(got[fl == 1], rec[fl == 1])
    h = got[fl == 0]
    assert np.array_equal(h[:, 0], rec[fl == 0, 0]) and np.array_equal(h[:, 6:], rec[fl == 0, 6:]) and np.all(h[:, 1] == 255)
    l0 = h[:, 2:4].copy().view(np.int16).reshape(-1)
    l1 = h[:, 4:6].copy().view(np.int16).reshape(-1)
    assert np.array_equal(l0, np.where(h[:, 0] >> 1, 255, -255)) and np.array_equal(l1, np.where(h[:, 0] & 1, 255, -255))
    assert l.ddn_symbol_capture_write(None, rec.ctypes.data, fl.ctypes.data, 1, 0) != 0


def test_wav_writer(built, tmp_path):
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 0.25], np.float32)
    path = str(tmp_path / "a.wav").encode()
    assert ddn.lib().ddn_wav_write_s16(path, 8000, 2, x.ctypes.data, 4, 1.0) == 0
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    fmt, ch, rate, brate, align, bits = struct.unpack("<HHIIHH", raw[20:36])
    assert (fmt, ch, rate, brate, align, bits) == (1, 2, 8000, 32000, 4, 16)
    assert struct.unpack("<I", raw[4:8])[0] == len(raw) - 8 and struct.unpack("<I", raw[40:44])[0] == 16
    s = np.frombuffer(raw[44:], np.int16)
    assert list(s) == [0, 16384, -16384, 32767, -32767, 32767, -32768, 8192]
    assert ddn.lib().ddn_wav_write_s16(path, 8000, 3, x.ctypes.data, 2, 1.0) != 0


# ---- the short-integer voice path (processAudio -> hpf_dL -> agsm) ------------------------------------------------------------
S16_STATE = 32


def s16_state(n, aout_gain=25.0):
    """fresh per-talk-path state: aout_gain 25 (src/core/util/dsd_init.c:580), empty peak history, filter at rest"""
    st = np.zeros((n, S16_STATE), np.float32)
    st[:, 0] = aout_gain
    return st


def oracle_s16(pcm, state, audio_gain=0.0, hpf=1, agsm=0):
    o = orc.oracle()
    o.orc_audio_s16.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_audio_s16.restype = None
    pcm = np.ascontiguousarray(pcm, np.float32)
    st = state.copy()
    out = np.zeros(pcm.shape, np.int16)
    ga = np.zeros(pcm.shape[0], np.float32)
    for s in range(pcm.shape[0]):
        o.orc_audio_s16(pcm[s].ctypes.data, pcm.shape[1], audio_gain, hpf, agsm, out[s].ctypes.data, st[s].ctypes.data, ga[s:s + 1].ctypes.data)
    return out, st, ga


@pytest.mark.skipif(not orc.have_ref(), reason="needs oracle/_ref")
@pytest.mark.parametrize("hpf,agsm", [(1, 0), (0, 1), (1, 1)])
def test_s16_hpf_and_agsm_equal_compiled_reference(built, hpf, agsm):
    """audio_gain < 0 makes stage 1 a plain clamp + truncation, so stages 2 and 3 see exactly the shorts the compiled hpf_dL /
    agsm are given"""
    r = orc.ref()
    r.refh_s16_post_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    r.refh_s16_post_run.restype = C.c_float
    rng = np.random.default_rng(12)
    pcm = voice_like(rng, 6, 60)
    pcm[5] *= 0.01                                                    # quiet: agsm's 3x cap
    got, st, ga = oracle_s16(pcm, s16_state(6), -1.0, hpf, agsm)
    for s in range(6):
        want = np.clip(pcm[s], -32768, 32767).astype(np.int16).reshape(-1).copy()
        g = r.refh_s16_post_run(want.ctypes.data, 60, hpf, agsm)
        assert np.array_equal(got[s].reshape(-1), want), s
        if agsm:
            assert ga[s] == np.float32(g)
    assert np.abs(got.astype(np.int32)).max() > 1000 and np.abs(pcm).max() > 32767


def test_s16_agsm_reference_vectors():
    """tests/core/test_core_audio_gain.c:251-302: 1000 -> 3000 at the 3x cap, silence stays silence"""
    pcm = np.full((1, 1, 160), 1000.0, np.float32)
    out, _, ga = oracle_s16(pcm, s16_state(1), -1.0, 0, 1)
    assert (out == 3000).all() and ga[0] == 3.0
    out, _, ga = oracle_s16(np.zeros((1, 2, 160), np.float32), s16_state(1), -1.0, 1, 1)
    assert (out == 0).all() and np.isfinite(ga[0])


def test_s16_auto_gain_walk():
    """stage 1 by its own rules: the gain drops at once to 30000 / peak, climbs back by at most 5 % per frame, never passes 50,
    and a loud frame stays in the 25-frame history"""
    x = np.zeros((1, 40, 160), np.float32)
    x[0, :, 80] = 100.0
    x[0, 3, 80] = 6000.0
    out, st, _ = oracle_s16(x, s16_state(1), 0.0, 0, 0)
    assert out[0, 3, 80] == 30000 and abs(int(out[0, 4, 80]) - 500) <= 1     # 30000 / 6000 = 5, held by the history
    assert abs(int(out[0, 30, 80]) - int(out[0, 29, 80]) * 1.05) <= 2         # frame 28 drops the loud block: +5 % per frame
    assert st[0, 0] <= 50.0 and st[0, 1] == 40 % 25
    out2, _, _ = oracle_s16(x, s16_state(1, 7.0), 12.0, 0, 0)                 # manual gain: aout_gain as handed in, no ramp
    assert out2[0, 5, 80] == 700
