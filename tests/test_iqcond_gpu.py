"""GPU parity: the FSK path with the optional IQ conditioning stages on (iq_dc_block,
full_demod_apply_iq_balance) through ddn_batch_set_iq_conditioning -> ddn_front_end_run, vs the CPU oracle (which
tests/test_oracle_iqopt.py pins to the compiled reference).  Bit-exact discriminator samples across call splits,
ragged blocks, the squelch gate, cf32 input and a half-band stage."""
import numpy as np
import pytest

import ddn
import orc
from test_oracle_iqopt import CASES, impaired_cu8

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", CASES)
def test_iq_conditioning_vs_oracle(built, case):
    dc, sh, bal, thr, ema, sq = case
    B, n, blk = 70, 21000, 4096            # 70: a ragged second wavefront; 21000: ragged last block
    iq = np.stack([impaired_cu8(100 + c, n) for c in range(B)])
    b = ddn.Batch(B, block_len=blk, squelch_level=sq)
    b.set_iq_conditioning(dc, sh, bal, thr, ema)
    cuts = [0, 4096, 4096 + 300, 12588, n]  # call boundaries = block boundaries of the reference run
    got = np.concatenate([b.run_host(iq[:, a:e], e - a) for a, e in zip(cuts[:-1], cuts[1:])], axis=1)
    for c in range(B):
        fe = orc.OracleFrontEnd(squelch=sq).set_iq_options(dc, sh, bal, thr, ema)
        want = np.concatenate([fe.run_cu8(iq[c, a:e], blk) for a, e in zip(cuts[:-1], cuts[1:])])
        bad = np.flatnonzero(bits(got[c]) != bits(want))
        assert len(bad) == 0, (case, c, bad[:5])
    st = np.zeros(5, np.float32)
    assert ddn.lib().ddn_batch_get_fsk_state(b.h, 0, st.ctypes.data) == 0 and st[2] in (0.0, 1.0)


def test_iq_conditioning_cf32_decimated_and_off_again(built):
    B, n, blk = 5, 16384, 2048
    rng = np.random.default_rng(3)
    iq8 = np.stack([impaired_cu8(300 + c, n) for c in range(B)])
    iq = ((iq8.astype(np.float32) - 127.5) / 127.5).astype(np.float32)
    b = ddn.Batch(B, sample_rate_hz=48000, input_format=ddn.IN_CF32, block_len=blk)
    b.set_decimation(1)
    b.set_iq_conditioning(1, 8, 1, 0.0, 0.0)
    got = b.run_host(iq, n)
    for c in range(B):
        fe = orc.OracleFrontEnd(downsample_passes=1).set_iq_options(1, 8, 1, 0.0, 0.0)
        want = fe.run_f32(iq[c], blk)
        assert np.array_equal(bits(got[c]), bits(want)), c
    # switches off again -> the fused kernel's output
    b.set_iq_conditioning(0, 0, 0, 0.0, 0.0)
    got = b.run_host(iq, n)
    for c in range(B):
        want = orc.OracleFrontEnd(downsample_passes=1).run_f32(iq[c], blk)
        assert np.array_equal(bits(got[c]), bits(want)), c
