#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libdsdneo_ref.so = the reference's own
sources compiled in place by oracle/Makefile).  Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Each file holds inputs + the reference's outputs for one boundary of the hot path; the inputs are either
excerpts of the reference's own test fixtures (tests/fixtures/iq/*.iq, data files) or synthetic I/Q from
tests/orc.py.  Manifest fields: block_len (the fixed full_demod block), impl (simd_fir_get_impl_name()).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import orc  # noqa: E402


def fe_case(name, iq, block_len, squelch=0.0, profile=4, rate=48000, sym=4800):
    out, taps, st = orc.ref_front_end_cu8(iq, block_len, rate=rate, sym=sym, profile=profile, squelch=squelch)
    np.savez_compressed(os.path.join(HERE, name), iq=iq, disc=out, taps=taps, state=st,
                        block_len=np.int32(block_len), squelch=np.float32(squelch), profile=np.int32(profile),
                        rate=np.int32(rate), impl=np.bytes_(orc.ref().simd_fir_get_impl_name()))
    print(name, iq.shape, out.shape, "zeros:", int((out == 0).sum()))


def main():
    vc = orc.load_fixture_cu8("p25p1_c4fm_vc.iq")
    cc = orc.load_fixture_cu8("p25p1_c4fm_cc.iq")
    nx = orc.load_fixture_cu8("nxdn48.iq")
    fe_case("fe_p25p1_vc_b8192.npz", vc[44000:44000 + 24576], 8192)
    fe_case("fe_p25p1_vc_b3000_sq.npz", vc[44000:64000], 3000, squelch=0.001)
    fe_case("fe_p25p1_cc_b8192.npz", cc[40000:40000 + 20000], 8192)
    fe_case("fe_nxdn48_b4096.npz", nx[38000:38000 + 12288], 4096, profile=1, sym=2400)
    syn = orc.synth_c4fm_cu8(0, 1, 20000)[0]
    fe_case("fe_synth_ch0_b8192.npz", syn, 8192)


if __name__ == "__main__":
    main()
