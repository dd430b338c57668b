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


def test_payload_primitives_equal_the_compiled_reference_and_its_test_numbers(built):
    o = orc.oracle()
    # what tests/protocol/ysf/test_ysf_frame.c asserts (:73-80, :85)
    assert [o.orc_ysf_vd2_index(k) for k in (0, 1, 2, 3, 4, 99, 100, 103)] == [0, 26, 52, 78, 1, 102, 25, 103]
    assert [o.orc_ysf_pn95_bit(i) for i in range(16)] == [1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1, 1, 1]
    assert o.orc_ysf_pn95_bit(512) == 1
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref not built")
    r.dsd_ysf_vd2_interleave_index.argtypes = [C.c_size_t]
    r.dsd_ysf_vd2_interleave_index.restype = C.c_uint8
    r.dsd_ysf_pn95_bit.argtypes = [C.c_size_t]
    r.dsd_ysf_pn95_bit.restype = C.c_uint8
    r.dsd_ysf_dewhiten_bits.argtypes = [C.c_void_p, C.c_size_t]
    o.orc_ysf_dewhiten.argtypes = [C.c_void_p, C.c_int]
    assert [o.orc_ysf_vd2_index(k) for k in range(104)] == [r.dsd_ysf_vd2_interleave_index(k) for k in range(104)]
    assert [o.orc_ysf_pn95_bit(k) for k in range(1100)] == [r.dsd_ysf_pn95_bit(k) for k in range(1100)]
    rng = np.random.default_rng(3)
    for n in (80, 160, 511, 512, 513, 1200):
        a = rng.integers(0, 2, n).astype(np.uint8)
        b = a.copy()
        r.dsd_ysf_dewhiten_bits(a.ctypes.data, n)
        o.orc_ysf_dewhiten(b.ctypes.data, n)
        assert np.array_equal(a, b), n


def test_ysf_capture_payloads(built):
    """the capture is V/D mode 2: behind every frame five voice sub-frames and the 100-dibit data channel; the data channel of the
    frames with a good FICH passes CRC16 in most of them (a weak capture), and the blocks that pass repeat their text fields"""
    disc = rx4.capture_disc("iq_ysf.npz", 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_YSF)).run(disc, max_sync=2048)
    fr, last = ysf.decode_payloads(out)
    full = [f for f in fr if f["payload"] is not None]
    assert len(full) >= 40 and all(f["payload"]["kind"] == 2 for f in full if f["dt"] == 2 and f["fi"] == 1)
    vd2 = [f for f in full if f["payload"]["kind"] == 2]
    assert len(vd2) >= 0.9 * len(full)                                   # frames with a failed FICH are read as the last good type
    good = [f for f in vd2 if f["payload"]["dch_status"][0] == 1]
    assert len(good) >= 8, (len(good), len(vd2))
    assert last == (2, 1)
    # a block that passes CRC16 is ten bytes of text or addresses: the same frame number of the next round carries the same block
    seen = {}
    for f in good:
        seen.setdefault(bytes(f["payload"]["dch"][0, :10]), 0)
        seen[bytes(f["payload"]["dch"][0, :10])] += 1
    assert max(seen.values()) >= 2, seen


def test_full_rate_unpack_equals_the_compiled_reference(built):
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref not built")
    o = orc.oracle()
    r.dsd_ysf_unpack_full_rate_imbe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_ysf_fr_unpack.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(9)
    for k in range(50):
        d = rng.integers(0, 4, 72).astype(np.uint8)
        raw = np.zeros(144, np.uint8)
        raw[0::2], raw[1::2] = d >> 1, d & 1
        vch, a = np.zeros(144, np.uint8), np.zeros((8, 23), np.uint8)
        r.dsd_ysf_unpack_full_rate_imbe(raw.ctypes.data, vch.ctypes.data, a.ctypes.data)
        b = np.zeros((8, 23), np.uint8)
        o.orc_ysf_fr_unpack(d.ctypes.data, b.ctypes.data)
        assert np.array_equal(a, b), k


def test_vd1_frames_are_the_shared_ambe_schedule(built):
    """ysf_ehr() uses the map nxdn_voice.c and dmr_bs.c use (the generated schedule, pinned in tests/test_oracle_rx4.py)"""
    rng = np.random.default_rng(10)
    p = rng.integers(0, 4, 360).astype(np.uint8)
    pl = ysf.payload(p, 1, 0)
    assert pl["kind"] == 1 and pl["n_frames"] == 4
    for sf in range(4):
        fr, _ = rx4.ambe2450_deinterleave(p[72 * sf + 36:72 * sf + 72])
        assert np.array_equal(pl["frames"][sf, :96].reshape(4, 24), fr), sf


def test_generated_fich_round_trip(built):
    """tests/ysfgen.py is the inverse of the pinned decoder stages: every field comes back with a path cost of 0"""
    import ysfgen
    rng = np.random.default_rng(4)
    for k in range(12):
        kw = dict(cm=int(rng.integers(0, 4)), bn=int(rng.integers(0, 4)), bt=int(rng.integers(0, 4)), fn=int(rng.integers(0, 8)),
                  ft=int(rng.integers(0, 8)), mr=int(rng.integers(0, 8)), vp=int(rng.integers(0, 2)), st=int(rng.integers(0, 2)),
                  sc=int(rng.integers(0, 128)))
        fi, dt = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        f = ysfgen.frame(rng, fi, dt, **kw)
        err, bits, cost = ysf.fich(f[20:120])
        got = ysf.fields(bits)
        assert err == 0 and cost == 0 and got["fi"] == fi and got["dt"] == dt and all(got[k2] == v for k2, v in kw.items()), (k, got)
    assert ysf.fich(ysfgen.frame(rng, 1, 2, break_fich=True)[20:120])[0] != 0
