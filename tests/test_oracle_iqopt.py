"""CPU: optional IQ conditioning between the channel LPF and the discriminator (iq_dc_block and
full_demod_apply_iq_balance, src/dsp/demod_pipeline.cpp:948-978,1131-1171) - oracle restatement pinned bit for bit
against the compiled reference's full_demod with the switches on."""
import numpy as np
import pytest

import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")
CASES = [  # (dc_enable, dc_shift, bal_enable, bal_thr, bal_ema_a, squelch)
    (1, 11, 0, 0.0, 0.0, 0.0),
    (1, 3, 0, 0.0, 0.0, 0.0),      # shift clamped up to 6
    (1, 20, 0, 0.0, 0.0, 0.0),     # shift clamped down to 15
    (0, 11, 1, 0.0, 0.0, 0.0),     # default threshold 0.02 / EMA 0.2
    (0, 11, 1, 0.001, 0.5, 0.0),
    (1, 9, 1, 0.0005, 0.3, 0.0),
    (1, 9, 1, 0.0005, 0.3, 0.02),  # with the squelch gate closing some blocks
]


def impaired_cu8(seed, n):
    """C4FM-like carrier with a DC offset and gain / phase imbalance so that both stages have something to do."""
    iq = orc.synth_c4fm_cu8(seed, 1, n)[0].astype(np.float64)
    x = (iq[:, 0] - 127.5) / 127.5 + 1j * (iq[:, 1] - 127.5) / 127.5
    x = x * np.where(np.arange(n) // 3000 % 4 == 3, 0.02, 1.0)       # quiet stretches for the squelch case
    y = 0.08 + 0.05j + x.real * 1.15 + 1j * (x.imag * 0.9 + 0.2 * x.real)
    out = np.stack([y.real, y.imag], -1) * 110.0 + 127.5
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


@needs_ref
@pytest.mark.parametrize("case", CASES)
def test_iq_options_vs_reference(case):
    dc, sh, bal, thr, ema, sq = case
    iq = impaired_cu8(11, 30000)
    want, _, _ = orc.ref_front_end_cu8(iq, 4096, squelch=sq, iq_options=(dc, sh, bal, thr, ema))
    fe = orc.OracleFrontEnd(squelch=sq).set_iq_options(dc, sh, bal, thr, ema)
    got = fe.run_cu8(iq, 4096)
    plain = orc.OracleFrontEnd(squelch=sq).run_cu8(iq, 4096)
    assert len(got) == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32)), case
    assert not np.array_equal(got.view(np.uint32), plain.view(np.uint32))   # the switches did change the output
