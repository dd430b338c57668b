#!/usr/bin/env python3
"""Golden vectors for the Gardner timing recovery from the REFERENCE'S OWN op25_gardner_cc (oracle/_ref)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import orc  # noqa: E402

if __name__ == "__main__":
    cases = {}
    for name, sps, rate in (("p25_cqpsk_48k", 10, 4800), ("p25_cqpsk_24k", 5, 4800), ("p25p2_48k", 8, 6000)):
        iq = orc.synth_qpsk_f32(1234 + sps, 1, 1200, sps)[0]
        blocks = [2000, 3, 4096, 10 ** 9]
        outs, st = orc.ref_ted_blocks(iq, sps, rate, 0.0, blocks)
        cases[name + "_iq"] = iq
        cases[name + "_sym"] = np.concatenate(outs, axis=0)
        cases[name + "_blocks"] = np.array(blocks[:3] + [iq.shape[0]], np.int64)
        cases[name + "_state"] = st
        cases[name + "_cfg"] = np.array([sps, rate], np.int32)
        print(name, iq.shape, cases[name + "_sym"].shape, st)
    np.savez_compressed(os.path.join(HERE, "ted_gardner.npz"), **cases)
