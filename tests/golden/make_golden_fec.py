#!/usr/bin/env python3
"""Golden vectors for the trellis / Viterbi decoders, produced by the REFERENCE'S OWN compiled decoders
(oracle/_ref/libdsdneo_ref.so: p25_12_soft_llr, dmr_r34_viterbi_decode[_soft], CNXDNConvolution_*,
viterbi_decode[_punctured]).  Run in the build container:  python tests/golden/make_golden_fec.py"""
import ctypes as C
import json
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
    r.p25_12_soft_llr.argtypes = [VP, VP, VP]
    r.dmr_r34_viterbi_decode.argtypes = [VP, VP]
    r.dmr_r34_viterbi_decode_soft.argtypes = [VP, VP, VP]
    r.CNXDNConvolution_decode.argtypes = [C.c_uint8] * 2
    r.CNXDNConvolution_decode_soft.argtypes = [C.c_uint8] * 4
    r.CNXDNConvolution_chainback.argtypes = [VP, C.c_uint]
    r.viterbi_decode.restype = C.c_uint32
    r.viterbi_decode.argtypes = [VP, VP, C.c_uint16]
    r.viterbi_decode_punctured.restype = C.c_uint32
    r.viterbi_decode_punctured.argtypes = [VP, VP, VP, C.c_uint16, C.c_uint16]
    rng = np.random.default_rng(20260928)

    n = 192
    llr, _ = fecgen.gen_p25_half_rate(rng, n)
    out = np.zeros((n, 12), np.uint8)
    met = np.zeros(n, np.int32)
    for i in range(n):
        met[i] = r.p25_12_soft_llr(None, llr[i].ctypes.data, out[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "fec_p25_half_rate.npz"), llr=llr, out=out, metric=met)

    d, rel, _ = fecgen.gen_r34(rng, n)
    oh = np.zeros((n, 18), np.uint8)
    os_ = np.zeros((n, 18), np.uint8)
    for i in range(n):
        r.dmr_r34_viterbi_decode(d[i].ctypes.data, oh[i].ctypes.data)
        r.dmr_r34_viterbi_decode_soft(d[i].ctypes.data, rel[i].ctypes.data, os_[i].ctypes.data)
    np.savez_compressed(os.path.join(HERE, "fec_r34.npz"), dibits=d, reliab=rel, out_hard=oh, out_soft=os_)

    cases = {}
    for name, steps, nbits, soft in [("facch", 96, 92, False), ("sacch", 36, 32, True), ("udch", 208, 204, True),
                                     ("long", 300, 296, False)]:
        sym, rl = fecgen.gen_nxdn(rng, 96, steps)
        o = np.zeros((96, (nbits + 7) // 8), np.uint8)
        for i in range(96):
            r.CNXDNConvolution_init()
            r.CNXDNConvolution_start()
            for t in range(steps):
                if soft:
                    r.CNXDNConvolution_decode_soft(int(sym[i, 2 * t]), int(sym[i, 2 * t + 1]), int(rl[i, 2 * t]),
                                                   int(rl[i, 2 * t + 1]))
                else:
                    r.CNXDNConvolution_decode(int(sym[i, 2 * t]), int(sym[i, 2 * t + 1]))
            r.CNXDNConvolution_chainback(o[i].ctypes.data, nbits)
        cases[name + "_sym"] = sym
        cases[name + "_rel"] = rl
        cases[name + "_out"] = o
        cases[name + "_cfg"] = np.array([steps, nbits, int(soft)], np.int32)
    np.savez_compressed(os.path.join(HERE, "fec_nxdn_conv.npz"), **cases)

    cases = {}
    p2 = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1], np.uint8)  # M17 stream puncture pattern P2 (11/12)
    for name, in_len, punct in [("lsf", 488, None), ("ysf", 200, None), ("stream", 272, p2)]:
        soft = fecgen.gen_m17(rng, 96, in_len)
        _, _, stride = fecgen.oracle_m17(soft[:1], punct)
        o = np.zeros((96, stride), np.uint8)
        cost = np.zeros(96, np.uint32)
        for i in range(96):
            if punct is None:
                cost[i] = r.viterbi_decode(o[i].ctypes.data, soft[i].ctypes.data, in_len)
            else:
                cost[i] = r.viterbi_decode_punctured(o[i].ctypes.data, soft[i].ctypes.data, punct.ctypes.data, in_len,
                                                     len(punct))
        cases[name + "_soft"] = soft
        cases[name + "_out"] = o
        cases[name + "_cost"] = cost
        cases[name + "_punct"] = punct if punct is not None else np.zeros(0, np.uint8)
    np.savez_compressed(os.path.join(HERE, "fec_viterbi_k5.npz"), **cases)

    # the reference's own 3/4-rate known-answer vectors (data table of tests/protocol/dmr/dmr_r34_reference_vectors.h)
    src = open("/root/reference/tests/protocol/dmr/dmr_r34_reference_vectors.h").read()
    import re
    blocks = re.findall(r"\{\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\}", src)
    kat = [dict(payload=[int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]+", a)],
                dibits=[int(x) for x in re.findall(r"\d+", b)]) for a, b in blocks]
    assert len(kat) == 3 and all(len(k["payload"]) == 18 and len(k["dibits"]) == 98 for k in kat)
    json.dump(kat, open(os.path.join(HERE, "kat_r34_reference_vectors.json"), "w"))
    print("ok", os.listdir(HERE))


if __name__ == "__main__":
    main()
