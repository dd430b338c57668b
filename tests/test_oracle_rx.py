"""CPU: the P25p1 receive-loop oracle (oracle/ddn_oracle_rx.c).

Pinned pieces are compared bit for bit with the reference's compiled units (frame_sync_level.c, sync_calibration.c,
dsd_dibit.c); the sample / hunting loop itself is unpinned (dsd_symbol.c / dsd_frame_sync.c cannot be built here) and
is checked functionally on synthetic P25p1 frame streams.
"""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


@needs_ref
def test_level_estimate_matches_reference():
    o, r = orc.oracle(), orc.ref()
    rng = np.random.default_rng(FZ + 5)
    for count in list(range(0, 26)) * 4:
        v = np.sort(rng.normal(0, 9000, max(count, 1)).astype(np.float32))
        a = (C.c_float * 2)()
        b = (C.c_float * 2)()
        o.orc_level_estimate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        r.refh_level_estimate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        o.orc_level_estimate(v.ctypes.data, count, C.addressof(a), C.addressof(a) + 4)
        r.refh_level_estimate(v.ctypes.data, count, C.addressof(b), C.addressof(b) + 4)
        assert bytes(a) == bytes(b), count


@needs_ref
def test_warm_start_then_slicer_matches_reference():
    """Warm start from a sync window, then keep slicing: thresholds, ring refill and the rebuilt binary64 sums."""
    o, r = orc.oracle(), orc.ref()
    r.refh_slicer_create.restype = C.c_void_p
    r.refh_slicer_create.argtypes = [C.c_int]
    r.refh_slicer_destroy.argtypes = [C.c_void_p]
    r.refh_slicer_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    r.refh_slicer_warm_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    r.refh_sync_p25p1_pos.restype = C.c_int
    o.orc_slicer_init.argtypes = [C.c_void_p, C.c_int]
    o.orc_slicer_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    o.orc_slicer_warm_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    o.orc_slicer_warm_start.restype = C.c_int
    rng = np.random.default_rng(FZ + 11)
    cases = []
    fs = np.where(orc.P25_FS_DIBITS == 1, 1.0, -1.0)
    cases.append((fs * 6500.0 + rng.normal(0, 300, 24)).astype(np.float32))
    cases.append((fs * 0.3).astype(np.float32))                     # span < 1 -> degenerate
    cases.append(np.abs(rng.normal(0, 3000, 24)).astype(np.float32) + 1)  # all positive -> degenerate
    cases.append((fs * 21000.0 + 900.0 + rng.normal(0, 2500, 24)).astype(np.float32))
    for sync in cases:
        h = r.refh_slicer_create(r.refh_sync_p25p1_pos())
        st = C.create_string_buffer(20000)
        o.orc_slicer_init(st, 0)
        pre = orc.synth_c4fm_symbols(3, 300)
        post = orc.synth_c4fm_symbols(4, 1500, scale=0.8)
        ra, rb = np.zeros((300, 4), np.int32), np.zeros((300, 4), np.int32)
        r.refh_slicer_run(h, pre.ctypes.data, 300, ra.ctypes.data, None)
        o.orc_slicer_run(st, pre.ctypes.data, 300, rb.ctypes.data, None)
        assert np.array_equal(ra, rb)
        t_ref = np.zeros(7, np.float32)
        rc_ref = r.refh_slicer_warm_start(h, sync.ctypes.data, 24, 24, t_ref.ctypes.data)
        nf = np.ascontiguousarray(sync[::-1])
        rc = o.orc_slicer_warm_start(st, nf.ctypes.data, 24)
        assert rc == rc_ref
        ra, rb = np.zeros((1500, 4), np.int32), np.zeros((1500, 4), np.int32)
        ta, tb = np.zeros((1500, 5), np.float32), np.zeros((1500, 5), np.float32)
        r.refh_slicer_run(h, post.ctypes.data, 1500, ra.ctypes.data, ta.ctypes.data)
        o.orc_slicer_run(st, post.ctypes.data, 1500, rb.ctypes.data, tb.ctypes.data)
        assert np.array_equal(ra, rb)
        assert ta.tobytes() == tb.tobytes()
        r.refh_slicer_destroy(h)


@pytest.mark.parametrize("use_filter,negative", [(0, False), (1, False), (1, True)])
def test_rx_finds_sync_and_recovers_payload(use_filter, negative):
    frame = 864
    x, dib, starts = orc.synth_p25_disc(21, 3, 60000, frame_dibits=frame, negative=negative)
    for c in range(3):
        rx = orc.OracleP25Rx(lock_symbols=frame - 24, use_filter=use_filter)
        sym, rec, fl = rx.run(x[c])
        acc = np.flatnonzero(fl & 2)
        assert len(acc) >= 4
        assert all(bool(fl[a] & 4) == negative for a in acc)
        # every accepted sync (after the filter turn-on transient) is followed by the frame's payload, error-free
        good = 0
        for a in acc[2:-1]:
            got = rec[a + 1:a + 1 + frame - 24, 0]
            assert np.all(fl[a + 1:a + 1 + frame - 24] & 1)
            # locate which frame this is from the payload itself
            hit = [f for f in range(dib.shape[1] // frame) if np.array_equal(dib[c, f * frame + 24:(f + 1) * frame], got)]
            good += len(hit) == 1
        assert good == len(acc[2:-1])
        # in steady state consecutive syncs are exactly one frame apart
        assert np.all(np.diff(acc[2:]) == frame)


def test_rx_block_split_invariance():
    x, _, _ = orc.synth_p25_disc(22, 1, 30000)
    a = orc.OracleP25Rx()
    s0, r0, f0 = a.run(x[0])
    b = orc.OracleP25Rx()
    parts = [b.run(x[0][i:j]) for i, j in [(0, 7), (7, 4096), (4096, 4097), (4097, 20001), (20001, 30000)]]
    s1 = np.concatenate([p[0] for p in parts])
    r1 = np.concatenate([p[1] for p in parts])
    f1 = np.concatenate([p[2] for p in parts])
    assert s0.tobytes() == s1.tobytes() and np.array_equal(r0, r1) and np.array_equal(f0, f1)
    assert a.thresholds().tobytes() == b.thresholds().tobytes()
