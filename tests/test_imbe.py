"""IMBE de-interleave (process_IMBE, src/protocol/p25/phase1/p25p1_ldu.c:89-120).
CPU: the oracle's schedule formula vs the reference's own interleave tables and soft-bit conversion (oracle/_ref harness),
every status-counter phase.  GPU: ddn_p25p1_imbe_deinterleave_* vs the oracle, reading capture records in place."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def _frame(rng, n=80):
    d = rng.integers(0, 4, n).astype(np.uint8)
    l0 = rng.integers(-400, 400, n).astype(np.int16)
    l1 = rng.integers(-32768, 32768, n).astype(np.int16)
    return d, l0, l1


@needs_ref
def test_oracle_imbe_deinterleave_vs_reference_tables():
    rng = np.random.default_rng(FZ + 7)
    for sc in list(range(0, 36)) * 3:
        d, l0, l1 = _frame(rng)
        fr, soft, flag, sc_out, used = orc.oracle_imbe_deinterleave(d, l0, l1, sc)
        rfr, rsoft, rsc, rused = orc.ref_imbe_deinterleave(d, l0, l1, sc)
        assert np.array_equal(fr, rfr) and np.array_equal(soft, rsoft), sc
        assert (sc_out, used) == (rsc, rused), sc
        assert flag == 0
    # every dibit reaches exactly two distinct cells: 144 cells written, the other 40 stay zero
    d = np.full(80, 3, np.uint8)
    fr, soft, _, _, _ = orc.oracle_imbe_deinterleave(d, np.zeros(80, np.int16), np.zeros(80, np.int16), 0)
    assert int(fr.sum()) == 144 and fr[4:7, 15:].sum() == 0 and fr[7, 7:].sum() == 0


def test_oracle_imbe_non_standard_c0_and_short_input():
    rng = np.random.default_rng(FZ + 8)
    d, l0, l1 = _frame(rng)
    fr, _, _, _, _ = orc.oracle_imbe_deinterleave(d, l0, l1, 3)
    # rebuild the dibit stream so that c0 becomes the word the reference skips (bits 15..17 set, all else clear)
    want = np.zeros(23, np.uint8)
    want[15:18] = 1
    for j in range(80):
        for hi in (1, 0):
            t = d.copy()
            t[j] ^= (2 if hi else 1)
            f2, _, _, _, _ = orc.oracle_imbe_deinterleave(t, l0, l1, 3)
            diff = np.argwhere(f2 != fr)
            if len(diff) == 1 and diff[0][0] == 0 and fr[0, diff[0][1]] != want[diff[0][1]]:
                d = t
                fr = f2
    assert np.array_equal(fr[0], want)
    assert orc.oracle_imbe_deinterleave(d, l0, l1, 3)[2] == 1
    assert orc.oracle_imbe_deinterleave(d[:60], l0[:60], l1[:60], 3)[2] == -1


@pytest.mark.gpu
def test_imbe_deinterleave_gpu_vs_oracle(built):
    import ddn
    rng = np.random.default_rng(FZ + 9)
    n_rec, n_frames = 5000, 300
    rec = np.zeros((n_rec, 10), np.uint8)
    rec[:, 0] = rng.integers(0, 4, n_rec)
    rec[:, 1] = rng.integers(0, 256, n_rec)
    llr = rng.integers(-600, 600, (n_rec, 2)).astype(np.int16)
    rec[:, 2:6] = llr.view(np.uint8).reshape(n_rec, 4)
    rec[:, 6:10] = rng.integers(0, 256, (n_rec, 4))
    first = rng.integers(0, n_rec - 75, n_frames).astype(np.int64)
    first[-1] = n_rec - 40       # runs past the end -> flag 0xFF
    first[-2] = n_rec - 75       # status_count 35 skips before steps 0, 35 and 70: 72 + 3 records, exactly fits
    sc = rng.integers(0, 36, n_frames).astype(np.int32)
    sc[-2] = 35
    # one frame whose c0 is the non-standard word: all-zero dibits except the ones feeding c0[15..17]
    probe = np.zeros(80, np.uint8)
    base = orc.oracle_imbe_deinterleave(probe, np.zeros(80, np.int16), np.zeros(80, np.int16), 0)[0]
    for j in range(80):
        for v in (1, 2):
            t = np.zeros(80, np.uint8)
            t[j] = v
            f = orc.oracle_imbe_deinterleave(t, np.zeros(80, np.int16), np.zeros(80, np.int16), 0)[0]
            w = np.argwhere(f != base)
            if len(w) == 1 and w[0][0] == 0 and 15 <= w[0][1] <= 17:
                probe[j] |= v
    first[0], sc[0] = 100, 0
    rec[100:180, 0] = probe
    fr = np.zeros((n_frames, 8, 23), np.uint8)
    soft = np.zeros((n_frames, 8, 23, 2), np.uint8)
    fl = np.zeros(n_frames, np.uint8)
    sco = np.zeros(n_frames, np.int32)
    rc = ddn.lib().ddn_p25p1_imbe_deinterleave_host(rec.ctypes.data, n_rec, first.ctypes.data, sc.ctypes.data, n_frames,
                                                    fr.ctypes.data, soft.ctypes.data, fl.ctypes.data, sco.ctypes.data)
    assert rc == 0, ddn.lib().ddn_last_error()
    assert fl[0] == 1 and fl[-1] == 0xFF and fl[-2] == 0
    for f in range(n_frames - 1):
        a = int(first[f])
        ofr, osoft, flag, osc, used = orc.oracle_imbe_deinterleave(rec[a:, 0], llr[a:, 0], llr[a:, 1], int(sc[f]))
        assert flag == fl[f], f
        assert np.array_equal(fr[f], ofr) and np.array_equal(soft[f], osoft), f
        assert sco[f] == osc, f
