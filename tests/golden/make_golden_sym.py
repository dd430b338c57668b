#!/usr/bin/env python3
"""Golden vectors for the P25p1 slicer / soft decisions and the P25 matched filter, produced by the reference's own
compiled dsd_dibit.c / dsd_filters.c (oracle/_ref; getSymbol supplied by oracle/ref_harness.cpp)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import orc  # noqa: E402

VP = C.c_void_p


def main():
    r = orc.ref()
    r.refh_slicer_create.restype = VP
    r.refh_slicer_create.argtypes = [C.c_int]
    r.refh_slicer_run.argtypes = [VP, VP, C.c_long, VP, VP]
    r.refh_slicer_destroy.argtypes = [VP]
    r.refh_p25_filter_run.argtypes = [VP, C.c_long, C.c_int, C.c_int, VP]
    cases = {}
    for name, neg, seed in (("pos", 0, 1), ("neg", 1, 2)):
        sym = orc.synth_c4fm_symbols(seed, 5000)
        h = r.refh_slicer_create(r.refh_sync_p25p1_neg() if neg else r.refh_sync_p25p1_pos())
        rec = np.zeros((5000, 4), np.int32)
        thr = np.zeros((5000, 5), np.float32)
        r.refh_slicer_run(h, sym.ctypes.data, 5000, rec.ctypes.data, thr.ctypes.data)
        r.refh_slicer_destroy(h)
        cases[name + "_sym"] = sym
        cases[name + "_rec"] = rec
        cases[name + "_thr_last"] = thr[-1]
    np.savez_compressed(os.path.join(HERE, "sym_p25_slicer.npz"), **cases)
    x = orc.synth_c4fm_symbols(7, 6000)
    y = np.zeros(6000, np.float32)
    r.refh_p25_filter_run(x.ctypes.data, 6000, 10, 1, y.ctypes.data)
    np.savez_compressed(os.path.join(HERE, "sym_p25_matched_filter.npz"), x=x, y=y)
    print("ok")


if __name__ == "__main__":
    main()
