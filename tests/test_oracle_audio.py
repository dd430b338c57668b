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
