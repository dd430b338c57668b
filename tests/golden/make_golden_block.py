#!/usr/bin/env python3
"""Golden vectors for the P25p1 block codes from the REFERENCE'S OWN compiled decoders
(p25p1_nid_decode via oracle/ref_harness.cpp:refh_nid_decode, BCH_63_16_11::decode_with_result via
refh_bch_63_16_decode, hamming_10_6_3_decode).  Run in the build container."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import fecgen  # noqa: E402
import orc  # noqa: E402

VP = C.c_void_p


def main():
    r = orc.ref()
    r.refh_nid_decode.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP]
    r.refh_bch_63_16_decode.argtypes = [VP, VP, VP]
    r.hamming_10_6_3_decode.argtypes = [VP, VP]
    rng = np.random.default_rng(20260929)
    n = 1500
    bits, rel, obs, par, prel = fecgen.gen_nid(rng, n, max_err=15)
    soft = np.zeros((n, 4), np.int32)
    hard = np.zeros((n, 4), np.int32)
    bch = np.zeros((n, 18), np.int32)  # success, count, 16 data bits
    for i in range(n):
        r.refh_nid_decode(bits[i].ctypes.data, rel[i].ctypes.data, int(obs[i]), int(par[i]), int(prel[i]),
                          soft[i].ctypes.data)
        r.refh_nid_decode(bits[i].ctypes.data, None, int(obs[i]), int(par[i]), int(prel[i]), hard[i].ctypes.data)
        d = np.zeros(16, np.uint8)
        e = C.c_int(0)
        bch[i, 0] = r.refh_bch_63_16_decode(bits[i].ctypes.data, d.ctypes.data, C.byref(e))
        bch[i, 1] = e.value
        bch[i, 2:] = d if bch[i, 0] else 0
    np.savez_compressed(os.path.join(HERE, "fec_p25p1_nid.npz"), bits=bits, rel=rel, obs=obs, parity=par,
                        parity_rel=prel, out_soft=soft, out_hard=hard, bch=bch, threshold=np.int32(64))
    ham = np.zeros((1024, 8), np.uint8)  # 6 corrected data bits, error count, 0
    for w in range(1024):
        d = np.array([(w >> (9 - k)) & 1 for k in range(6)], np.int8)
        p = np.array([(w >> (3 - k)) & 1 for k in range(4)], np.int8)
        e = r.hamming_10_6_3_decode(d.ctypes.data, p.ctypes.data)
        ham[w, :6] = d
        ham[w, 6] = e
    np.savez_compressed(os.path.join(HERE, "fec_hamming_10_6_3.npz"), table=ham)
    print("nid status histogram (soft):", np.bincount(soft[:, 0], minlength=3), "(hard):",
          np.bincount(hard[:, 0], minlength=3))


if __name__ == "__main__":
    main()
