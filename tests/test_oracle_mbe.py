"""CPU: the vocoder restatement (oracle/ddn_oracle_mbe.c) against what the reference holds for this stage, the table
blob's invariants, and the decoder's control flow.  Past the frame FEC the stage is PARITY UNPINNED (mbelib-neo source
absent), see oracle/ddn_oracle_mbe.c."""
import ctypes as C

import numpy as np
import pytest

import ddn
import mbe


def test_imbe_frame_decode_reference_held_vectors(built):
    """tests/core/test_core_mbe_transform_context.c:134-152: the four capture-derived frames decode to the data words and
    correction counts that file asserts (:869, :886, :1014, :1020, :1043)."""
    for k in mbe.load_kat():
        bits, res, rc = mbe.oracle_frame_decode(ddn.MBE_IMBE, k["frame"][None])
        assert rc[0] == 0
        hx = mbe.bits_hex(bits[0])
        assert res[0, 3] == k["total_errors"], k["name"]
        if "c0_errors" in k:
            assert res[0, 1] == k["c0_errors"]
        if "imbe_d_hex" in k:
            assert hx == k["imbe_d_hex"], (k["name"], hx)
        if "imbe_d_hex_prefix" in k:
            assert hx.startswith(k["imbe_d_hex_prefix"]), (k["name"], hx)
        assert res[0, 4] == res[0, 3] - res[0, 1]          # protected = total - c0 (test_core_mbe_transform_context.c:268)
        assert res[0, 0] & 1 and res[0, 0] & 2              # C0_VALID, C4_VALID


@pytest.mark.parametrize("codec", [ddn.MBE_IMBE, ddn.MBE_AMBE])
def test_frame_round_trip_and_correction_capacity(built, codec):
    """encode -> flip up to t bits per code word -> decode returns the data and mbelib's correction counts."""
    rng = np.random.default_rng(11 + codec)
    n = 300
    if codec == ddn.MBE_IMBE:
        data = rng.integers(0, 2, size=(n, 88), dtype=np.uint8)
        frames = np.stack([mbe.imbe_encode(d) for d in data])
        # Golay words count corrected DATA bits only (mbe_golay2312), Hamming words any corrected bit (mbe_hamming1511)
        words = [(r, 23, 3, 11) for r in range(4)] + [(r, 15, 1, 0) for r in range(4, 7)]
    else:
        data = rng.integers(0, 2, size=(n, 49), dtype=np.uint8)
        frames = np.stack([mbe.ambe_encode(d) for d in data])
        words = [(0, 23, 3, 11), (1, 23, 3, 11)]
    clean_bits, clean_res, rc = mbe.oracle_frame_decode(codec, frames)
    assert np.all(rc == 0) and np.array_equal(clean_bits, data) and np.all(clean_res[:, 3] == 0)
    noisy = frames.copy()
    want_errs = np.zeros(n, np.int64)
    for i in range(n):
        for (r, ln, t, first_data) in words:
            off = 1 if (codec == ddn.MBE_AMBE and r == 0) else 0
            k = int(rng.integers(0, t + 1))
            pos = rng.choice(ln, size=k, replace=False)
            for p in pos:
                noisy[i, r, p + off] ^= 1
                want_errs[i] += int(p >= first_data)
    bits, res, rc = mbe.oracle_frame_decode(codec, noisy)
    assert np.all(rc == 0) and np.array_equal(bits, data)
    assert np.array_equal(res[:, 3], want_errs)
    bad = frames[:1].copy()
    bad[0, 1, 3] = 2
    assert mbe.oracle_frame_decode(codec, bad)[2][0] == -2      # MBE_STATUS_INVALID_BITS


def test_default_tables_are_flagged_synthetic_and_consistent(built):
    t = mbe.tables()
    l = ddn.lib()
    assert t.synthetic == 1 and l.ddn_mbe_validate_tables(C.byref(t)) == 0
    for L in range(9, 57):
        K = (L + 2) // 3 if L < 37 else 12
        assert sum(t.imbe_bits[L - 9][:]) == 73 - K
        assert sum(t.ambe_blocks[L][:]) == L
    broken = mbe.tables()
    broken.imbe_bits[5][4] += 1
    assert l.ddn_mbe_validate_tables(C.byref(broken)) < 0
    broken = mbe.tables()
    broken.imbe_bit_order[0][85][0] = 3                        # the fundamental's LSBs must stay at imbe_d[85..86]
    assert l.ddn_mbe_validate_tables(C.byref(broken)) < 0
    broken = mbe.tables()
    broken.magic = 0
    assert l.ddn_mbe_validate_tables(C.byref(broken)) < 0


@pytest.mark.parametrize("codec", [ddn.MBE_IMBE, ddn.MBE_AMBE])
def test_process_control_flow(built, codec):
    """repeat after too many corrections, mute after four repeats in a row, an invalid IMBE fundamental repeats (synthesized) and
    only the fourth in a row mutes, AMBE erasure / tone frames mute; audio is finite and non-trivial on valid frames."""
    rng = np.random.default_rng(3)
    F = 12
    bits = (mbe.random_imbe_bits if codec == ddn.MBE_IMBE else mbe.random_ambe_bits)(rng, (1, F))
    res_in = np.zeros((1, F, 5), np.int32)
    limit = 5 if codec == ddn.MBE_IMBE else 3
    res_in[0, 3:8, 3] = limit + 1                              # frames 3..7 exceed the correction limit
    v = mbe.OracleVocoder(codec, 1)
    pcm, res, rc = v.run(bits, res_in)
    assert rc == 0 and np.all(np.isfinite(pcm))
    REPEAT, MUTE = 0x8, 0x10
    assert [bool(f & REPEAT) for f in res[0, :, 0]] == [False] * 3 + [True] * 5 + [False] * 4
    assert [bool(f & MUTE) for f in res[0, :, 0]] == [False] * 6 + [True] + [False] * 5   # repeat count 4 -> mute at frame 6
    assert np.all(pcm[0, 6] == 0) and np.abs(pcm[0, 1]).max() > 0 and np.abs(pcm[0, 9]).max() > 0
    assert v.cur[0].un == F
    # invalid fundamental / special frames
    v = mbe.OracleVocoder(codec, 1)
    if codec == ddn.MBE_IMBE:
        # mbelib 1.3 mbe_processImbe4400Dataf: an invalid fundamental (b0 > 207) repeats the last good frame - synthesized, not muted -
        # up to three times in a row; the fourth mutes and re-initialises the talk path (the `bad == 0` gate is AMBE's only)
        special = bits[:, :8].copy()
        special[0, 2:6, :6] = 1                                # frames 2..5: b0 >= 252 > 207
        want = [0, 0, REPEAT, REPEAT, REPEAT, REPEAT | MUTE, 0, 0]
        pcm, res, rc = v.run(special)
        assert rc == 0 and [int(f) & 0x78 for f in res[0, :, 0]] == want
        assert all(np.abs(pcm[0, k]).max() > 0 for k in (2, 3, 4)) and np.all(pcm[0, 5] == 0) and np.abs(pcm[0, 6]).max() > 0
        assert v.prev[0].L == bits_L(special[0, 7])
        # the repeated frames carry the last good frame's parameters: same L, and the history survives the bad frame
        v2 = mbe.OracleVocoder(codec, 1)
        v2.run(special[:, :3])
        assert v2.prev[0].L == bits_L(special[0, 1]) and v2.cur[0].repeat == 1
    else:
        special = bits[:, :3].copy()
        for k, p in enumerate((0, 1, 2, 3, 37, 38, 39)):
            special[0, 1, p] = (120 >> (6 - k)) & 1            # erasure
            special[0, 2, p] = (126 >> (6 - k)) & 1            # tone
        want = [0, MUTE | 0x40, MUTE | 0x20]
        pcm, res, rc = v.run(special)
        assert rc == 0 and [int(f) & 0x78 for f in res[0, :, 0]] == want
        assert np.all(pcm[0, 1] == 0)


def bits_L(b):
    b0 = 0
    for k in range(6):
        b0 = (b0 << 1) | int(b[k])
    b0 = (b0 << 2) | (int(b[85]) << 1) | int(b[86])
    return (9254 * ((2 * b0 + 81) // 8)) // 10000


def test_imbe_L_formula_matches_published_float_form(built):
    """L = (int)(0.9254 * (int)(pi / w0 + 0.25)), w0 = 4 pi / (b0 + 39.5): the integer form used is the same function."""
    for b0 in range(208):
        w0 = 4.0 * np.pi / (b0 + 39.5)
        L = int(0.9254 * int(np.pi / w0 + 0.25))
        assert L == (9254 * ((2 * b0 + 81) // 8)) // 10000 and 9 <= L <= 56


def test_p25p1_tail_erasure_rule(built):
    """dsd_mbe.c:447-463,540-566: FC.. frame with >= 10 corrections is muted without touching the history; the dense FC
    frame (12 corrections, > 24 set bits) is not (test_core_mbe_transform_context.c:1040-1047)."""
    kat = {k["name"]: k for k in mbe.load_kat()}
    for name, muted in (("p25p1_tail_erasure", True), ("p25p1_dense_fc", False)):
        bits, res, _ = mbe.oracle_frame_decode(ddn.MBE_IMBE, kat[name]["frame"][None])
        v = mbe.OracleVocoder(ddn.MBE_IMBE, 1, tail_rule=1)
        before = mbe.parms_tuple(v.enh[0])
        pcm, out, rc = v.run(bits[None], res[None])
        assert rc == 0
        if muted:
            assert np.all(pcm == 0) and np.all(out == 0) and v.cur[0].un == 0
            assert all(np.array_equal(a, b) for a, b in zip(before, mbe.parms_tuple(v.enh[0])))
        else:
            assert out[0, 0, 3] == 12 and v.cur[0].un == 1


def test_table_blob_file_round_trip(built, tmp_path):
    """a blob marked synthetic = 0 survives save -> load byte for byte and keeps the flag; damaged files and blobs that fail
    validation are refused (host only: no device)"""
    import ctypes as C
    l = ddn.lib()
    t = ddn.MbeTables()
    assert l.ddn_mbe_default_tables(C.byref(t)) == 0 and t.synthetic == 1
    t.synthetic = 0                                     # what an integrator's blob says
    t.ambe_dg[3] = 0.25                                 # ... and some content of its own
    path = str(tmp_path / "tables.ddnmbet").encode()
    assert l.ddn_mbe_tables_save_file(path, C.byref(t)) == 0
    u = ddn.MbeTables()
    assert l.ddn_mbe_tables_load_file(path, C.byref(u)) == 0
    assert bytes(u) == bytes(t) and u.synthetic == 0 and u.ambe_dg[3] == 0.25
    raw = bytearray(open(path, "rb").read())
    assert raw[:8] == b"DDNMBET1" and len(raw) == 8 + 4 + C.sizeof(ddn.MbeTables) + 4
    raw[100] ^= 1                                       # one flipped bit: the checksum no longer fits
    bad = str(tmp_path / "bad.ddnmbet").encode()
    open(bad, "wb").write(raw)
    assert l.ddn_mbe_tables_load_file(bad, C.byref(u)) != 0
    open(bad, "wb").write(bytes(raw[:-9]))              # truncated
    assert l.ddn_mbe_tables_load_file(bad, C.byref(u)) != 0
    t.ambe_L[5] = 3                                     # out of range: refused at save time
    assert l.ddn_mbe_tables_save_file(bad, C.byref(t)) != 0
