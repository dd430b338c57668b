"""GPU parity: P25p1 NID (BCH + Chase) and Hamming(10,6,3) batch kernels vs reference goldens and the oracle."""
import ctypes as C

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import ddn
import fecgen
from conftest import golden
from test_oracle_block import oracle_nid

pytestmark = pytest.mark.gpu


def gpu_nid(bits, rel, obs, par, prel, thr=64):
    n = bits.shape[0]
    out = np.zeros((n, 4), np.int32)
    rc = ddn.lib().ddn_p25p1_nid_decode_host(bits.ctypes.data, rel.ctypes.data if rel is not None else None,
                                             obs.ctypes.data, par.ctypes.data, prel.ctypes.data, thr, n,
                                             out.ctypes.data)
    assert rc == 0, ddn.lib().ddn_last_error()
    return out


def test_nid_golden_and_fresh(built):
    g = golden("fec_p25p1_nid.npz")
    obs = np.ascontiguousarray(g["obs"], np.int32)
    assert np.array_equal(gpu_nid(g["bits"], g["rel"], obs, g["parity"], g["parity_rel"]), g["out_soft"])
    assert np.array_equal(gpu_nid(g["bits"], None, obs, g["parity"], g["parity_rel"]), g["out_hard"])
    rng = np.random.default_rng(FZ + 31)
    # up to 16384 NIDs a call takes the wavefront-per-NID route (the receive loops' nid_decode_wave), larger ones lane-per-NID + Chase list
    for n, thr in ((1, 64), (65, 64), (3000, 40), (500, 200), (17000, 64)):
        bits, rel, obs, par, prel = fecgen.gen_nid(rng, n, max_err=16)
        rel[: n // 4] = rng.integers(0, 256, (n // 4, 63))  # ties / everything below threshold
        want = oracle_nid(bits, rel, obs, par, prel, thr)
        assert np.array_equal(gpu_nid(bits, rel, obs, par, prel, thr), want)
    o4 = np.zeros(4, np.int32)
    b0 = np.ascontiguousarray(g["bits"][0])
    r0 = np.ascontiguousarray(g["rel"][0])
    rc = ddn.lib().ddn_p25p1_nid_decode(b0.ctypes.data, r0.ctypes.data, int(g["obs"][0]), int(g["parity"][0]),
                                        int(g["parity_rel"][0]), 64, o4.ctypes.data)
    assert rc == 0 and np.array_equal(o4, g["out_soft"][0])


def test_hamming_all_words(built):
    t = golden("fec_hamming_10_6_3.npz")["table"]
    words = np.arange(1024)
    bits = ((words[:, None] >> (9 - np.arange(10))[None, :]) & 1).astype(np.uint8)
    errs = np.zeros(1024, np.uint8)
    rc = ddn.lib().ddn_fec_hamming_10_6_3_host(bits.ctypes.data, 1024, errs.ctypes.data)
    assert rc == 0
    assert np.array_equal(errs, t[:, 6]) and np.array_equal(bits[:, :6], t[:, :6])
    d = np.array([1, 0, 1, 1, 0, 0], np.int8)
    p = np.array([0, 1, 1, 0], np.int8)
    w = int("".join(map(str, d.tolist() + p.tolist())), 2)
    e = ddn.lib().hamming_10_6_3_decode(d.ctypes.data, p.ctypes.data)
    assert e == t[w, 6] and d.tolist() == t[w, :6].tolist()
