"""CPU: the oracle restatement reproduces the reference's golden vectors bit-for-bit (and, where the compiled
reference is present, the reference itself on fresh inputs).  These pin the oracle (task rule ③)."""
import numpy as np
import pytest

import orc
from conftest import golden

FE_CASES = ["fe_p25p1_vc_b8192.npz", "fe_p25p1_vc_b3000_sq.npz", "fe_p25p1_cc_b8192.npz", "fe_nxdn48_b4096.npz",
            "fe_synth_ch0_b8192.npz"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", FE_CASES)
def test_front_end_oracle_matches_golden(built, name):
    g = golden(name)
    fe = orc.OracleFrontEnd(rate=int(g["rate"]), profile=int(g["profile"]), squelch=float(g["squelch"]))
    out = fe.run_cu8(g["iq"], int(g["block_len"]))
    assert out.shape == g["disc"].shape
    assert np.array_equal(bits(out), bits(g["disc"]))
    # first sample of a stream is exactly 0.0 (src/dsp/fsk_modem.c:148-153)
    assert out[0] == 0.0


@pytest.mark.parametrize("name", FE_CASES)
def test_channel_lpf_design_matches_golden(built, name):
    g = golden(name)
    taps = np.zeros(144, np.float32)
    n = orc.oracle().orc_channel_lpf_design(int(g["rate"]), int(g["profile"]), taps.ctypes.data, 144)
    assert n == len(g["taps"]) == 135
    assert np.array_equal(bits(taps[:n]), bits(g["taps"]))


def test_synth_generator_is_stable(built):
    g = golden("fe_synth_ch0_b8192.npz")
    iq = orc.synth_c4fm_cu8(0, 1, 20000)[0]
    # cos/sin are libm-dependent: allow a handful of +-1 LSB quantisation flips, not a different signal
    d = np.abs(iq.astype(np.int16) - g["iq"].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_discriminator_contract(built):
    """Properties the reference's own unit test pins (tests/dsp/test_fsk_modem.c:31-91): sign follows the
    rotation direction, output is scaled towards +-30000 and clipped to the int16 range."""
    o = orc.oracle()
    n = 4000
    ph = np.cumsum(np.full(n, 0.2))
    for sgn in (+1.0, -1.0):
        iq = np.stack([np.cos(sgn * ph), np.sin(sgn * ph)], axis=1).astype(np.float32)
        st = np.zeros(5, np.float32)
        out = np.zeros(n, np.float32)
        w = o.orc_fsk_discriminator(st.ctypes.data, iq.ctypes.data, 2 * n, out.ctypes.data, n)
        assert w == n and out[0] == 0.0
        assert np.sign(out[1]) == sgn
        assert out.max() <= 32767.0 and out.min() >= -32768.0
        assert abs(abs(out[1]) - 30000.0) < 1.0


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("block_len,squelch,profile", [(8192, 0.0, 4), (4000, 0.0, 2), (300, 0.0, 4), (100, 0.0, 4),
                                                       (131072, 0.001, 4), (5000, 0.001, 5)])
def test_front_end_oracle_matches_reference(built, block_len, squelch, profile):
    iq = orc.synth_c4fm_cu8(7, 1, 30000)[0]
    want, taps, st = orc.ref_front_end_cu8(iq, block_len, profile=profile, squelch=squelch)
    fe = orc.OracleFrontEnd(profile=profile, squelch=squelch)
    got = fe.run_cu8(iq, block_len)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_matches_reference_on_full_fixture(built):
    import os
    if not os.path.exists(orc.REFERENCE_ROOT):
        pytest.skip("reference fixtures not present")
    iq = orc.load_fixture_cu8("p25p1_c4fm_vc.iq")
    want, _, _ = orc.ref_front_end_cu8(iq, 8192)
    got = orc.OracleFrontEnd().run_cu8(iq, 8192)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("passes,block_len", [(1, 8192), (2, 8192), (1, 600), (3, 4096), (2, 1000)])
def test_halfband_cascade_oracle_matches_reference(built, passes, block_len):
    """downsample_passes > 0: 31-tap + 15-tap half-band stages in front of the channel LPF (full_demod)."""
    iq = orc.synth_c4fm_cu8(9, 1, 40000, sps=10 << passes)[0]
    want, _, _ = orc.ref_front_end_cu8(iq, block_len, passes=passes)
    fe = orc.OracleFrontEnd(downsample_passes=passes)
    got = fe.run_cu8(iq, block_len)
    assert len(want) == 40000 >> passes or block_len % (1 << passes)
    assert np.array_equal(bits(got[:len(want)]), bits(want))
