"""GPU parity: batched CQPSK front end (ddn_cqpsk_*: channel LPF -> RMS AGC -> FLL -> Gardner -> diff phasor -> Costas
-> phase extractor) vs the CPU oracle, which is pinned bit for bit to full_demod(cqpsk_enable) of the compiled
reference (tests/test_oracle_cqpsk.py).  Symbols, counts and carried loop state identical, across calls."""
import numpy as np
import pytest

import ddn
import orc

pytestmark = pytest.mark.gpu


def _oracle(iq, rate, blk, lpf, splits):
    out = []
    for c in range(iq.shape[0]):
        fe = orc.OracleCqpskFe(rate=rate, lpf_enable=lpf)
        parts = [fe.run(iq[c, a:b], blk) for a, b in zip(splits[:-1], splits[1:])]
        s = np.zeros(8, np.float32)
        import ctypes as C
        fe.o.orc_cqpsk_fe_get_state.argtypes = [C.c_void_p, C.c_void_p]
        fe.o.orc_cqpsk_fe_get_state(fe.st, s.ctypes.data)
        out.append((parts, s))
    return out


@pytest.mark.parametrize("rate,sps,blk,lpf", [(24000, 5, 4096, 1), (48000, 10, 8192, 1), (24000, 5, 1000, 0)])
def test_cqpsk_vs_oracle(built, rate, sps, blk, lpf):
    B = 70
    iq = orc.synth_dqpsk_f32(70 + sps, B, 3000, sps)
    iq[3] = 0.0                                                          # dead channel
    iq[4] = np.random.default_rng(2).standard_normal(iq[4].shape).astype(np.float32) * 0.2   # noise only
    n = iq.shape[1]
    splits = [0, 3 * blk, n]                                             # second call ends in a ragged block
    if (n - 3 * blk) % blk in (1, 2, 3):
        splits[-1] -= 4
    want = _oracle(iq, rate, blk, lpf, splits)
    b = ddn.CqpskBatch(B, rate=rate, lpf_enable=lpf, block_len=blk)
    for k, (a, e) in enumerate(zip(splits[:-1], splits[1:])):
        sym, cnt = b.run(iq[:, a:e])
        for c in range(B):
            w = want[c][0][k]
            assert cnt[c] == len(w), (c, k, cnt[c], len(w))
            assert np.array_equal(sym[c, :cnt[c]].view(np.uint32), w.view(np.uint32)), (c, k)
    for c in range(B):
        assert np.array_equal(b.state(c).view(np.uint32), want[c][1].view(np.uint32)), c


def test_cqpsk_cu8_input(built):
    B, sps, blk = 8, 5, 2048
    iqf = orc.synth_dqpsk_f32(91, B, 2500, sps, amp=0.5)
    u8 = np.clip(np.rint(127.5 + 127.5 * iqf), 0, 255).astype(np.uint8)
    wid = ((u8.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    n = u8.shape[1]
    b = ddn.CqpskBatch(B, rate=24000, block_len=blk, input_format=ddn.IN_CU8)
    sym, cnt = b.run(u8)
    for c in range(B):
        w = orc.OracleCqpskFe(rate=24000).run(wid[c], blk)
        assert cnt[c] == len(w)
        assert np.array_equal(sym[c, :cnt[c]].view(np.uint32), w.view(np.uint32)), c
