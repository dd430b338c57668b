"""The YSF frame information channel on the CPU: the restatement's stages against the reference's own compiled code (oracle/_ref:
dsd_ysf_soft_viterbi_decode of ysf_frame.c, Golay_24_12_decode of fec.c), and the known answer of the reference's capture through the
loop's YSF profile: every frame 480 symbols after the one before, FICH CRC good, "V/D2 RID Mode Repeater CC"
(DECODE_IQ_YSF, tests/CMakeLists.txt:8953-8957)."""
import ctypes as C

import numpy as np
import pytest

import orc
import rx4
import ysf


def test_viterbi_and_golay_stages_equal_the_compiled_reference(built):
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref not built")
    o = orc.oracle()
    r.dsd_ysf_soft_viterbi_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    r.dsd_ysf_soft_viterbi_decode.restype = C.c_uint32
    o.orc_ysf_soft_viterbi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    o.orc_ysf_soft_viterbi.restype = C.c_uint32
    r.Golay_24_12_decode.argtypes = [C.c_void_p]
    r.Golay_24_12_decode.restype = C.c_bool
    o.orc_golay_dmr_decode.argtypes = [C.c_int, C.c_void_p]
    o.orc_golay_dmr_decode.restype = C.c_int
    rng = np.random.default_rng(12)
    for n, nby, off, nb in ((100, 13, 8, 96), (180, 23, 8, 176)):       # the FICH and the DCH shapes (ysf.c:265,321,380)
        for k in range(200):
            d = rng.integers(0, 4, n).astype(np.uint8)
            a, b, by = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros(nb // 8, np.uint8)
            ea = r.dsd_ysf_soft_viterbi_decode(d.ctypes.data, n, nby, off, nb, a.ctypes.data, by.ctypes.data)
            eb = o.orc_ysf_soft_viterbi(d.ctypes.data, n, nby, off, nb, b.ctypes.data)
            assert ea == eb and np.array_equal(a, b), (n, k)
    r.Golay_24_12_init()      # (the reference builds its syndrome table at start-up: InitAllFecFunction())
    for s in range(4096):     # the decoder's answer depends on the syndrome alone: a word per syndrome is every case
        w = np.zeros(24, np.uint8)
        w[12:] = [(s >> (11 - i)) & 1 for i in range(12)]
        a, b = w.copy(), w.copy()
        assert bool(r.Golay_24_12_decode(a.ctypes.data)) == bool(o.orc_golay_dmr_decode(24, b.ctypes.data)) and np.array_equal(a, b), s
    for k in range(500):
        w = rng.integers(0, 2, 24).astype(np.uint8)
        a, b = w.copy(), w.copy()
        assert bool(r.Golay_24_12_decode(a.ctypes.data)) == bool(o.orc_golay_dmr_decode(24, b.ctypes.data)) and np.array_equal(a, b), k


def test_ysf_capture_known_answer(built):
    disc = rx4.capture_disc("iq_ysf.npz", 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=2048)
    fr = ysf.decode_frames(out)
    good = [f for f in fr if f["err"] == 0]
    # a weak capture: the Viterbi path costs say 8 .. 25 of a frame's 200 coded FICH bits are wrong, and about half of the frames pass
    # Golay + CRC16 - every one of those says what the reference's test asserts
    assert len(fr) >= 40 and len(good) >= 15, (len(good), len(fr))
    assert {ysf.summary(f["fields"]) for f in good} == {"V/D2 RID Mode Repeater CC"}
    # frames 480 symbols apart (20 sync + 100 FICH + 360), frame numbers counting round their total
    gaps = [b["pos"] - a["pos"] for a, b in zip(fr[:-1], fr[1:])]
    assert sum(1 for v in gaps if v % 480 <= 1 or v % 480 >= 479) >= 0.9 * len(gaps), gaps
    for a, b in zip(good[:-1], good[1:]):
        if b["pos"] - a["pos"] == 480:
            assert b["fields"]["fn"] == (a["fields"]["fn"] + 1) % (a["fields"]["ft"] + 1), (a["pos"], a["fields"], b["fields"])
