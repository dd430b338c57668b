"""CPU: the CQPSK chain restatement (channel LPF -> RMS AGC -> FLL band-edge -> Gardner -> diff phasor -> Costas ->
phase extractor, oracle/ddn_oracle_cqpsk.c) pinned bit for bit against full_demod(cqpsk_enable) of the compiled
reference (oracle/_ref)."""
import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


@needs_ref
@pytest.mark.parametrize("rate,sps,block_len,lpf", [(24000, 5, 4096, 1), (48000, 10, 8192, 1), (24000, 5, 1000, 1),
                                                    (24000, 5, 333, 0), (48000, 10, 20000, 1)])
def test_cqpsk_chain_matches_reference(built, rate, sps, block_len, lpf):
    iq = orc.synth_dqpsk_f32(3 + sps, 1, 5000, sps)[0]
    want, st = orc.ref_cqpsk_f32(iq, block_len, rate=rate, lpf=lpf)
    got = orc.OracleCqpskFe(rate=rate, lpf_enable=lpf).run(iq, block_len)
    assert len(got) == len(want) and len(want) > 4000
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the loops did something: the FLL moved off zero and symbols sit near the 4 levels
    assert abs(st[1]) > 1e-5
    tail = want[-2000:]
    assert np.mean(np.minimum(np.abs(np.abs(tail) - 1.0), np.abs(np.abs(tail) - 3.0)) < 0.5) > 0.9


@needs_ref
def test_cqpsk_noise_and_silence(built):
    rng = np.random.default_rng(FZ + 8)
    noise = (rng.standard_normal((20000, 2)) * 0.3).astype(np.float32)
    noise[5000:9000] = 0.0                      # dead air: AGC floor, detector confidence 0, zero-magnitude paths
    noise[12000:12100] *= 1e4
    want, _ = orc.ref_cqpsk_f32(noise, 4096)
    got = orc.OracleCqpskFe().run(noise, 4096)
    assert len(got) == len(want)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
