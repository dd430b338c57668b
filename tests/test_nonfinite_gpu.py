"""Non-finite input must stay contained: NaN / Inf samples in some channels neither hang a kernel (every data-dependent
loop is bounded) nor leak into the other channels of the batch, which still equal the oracle bit for bit."""
import numpy as np
import pytest

import ddn
import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_front_end_and_rx_with_nan_inf_channels(built):
    B, n = 20, 12000
    iq8 = orc.synth_c4fm_cu8(50, B, n)
    iq = ((iq8.astype(np.float32) - 127.5) / 127.5).astype(np.float32)
    bad = [3, 7, 16]
    iq[3, 1000:1100] = np.nan
    iq[7, 5000:5003, 0] = np.inf
    iq[16, ::97, 1] = -np.inf
    fe = ddn.Batch(B, input_format=ddn.IN_CF32, block_len=4096)
    disc = fe.run_host(iq, n)
    for c in range(B):
        if c in bad:
            continue
        want = orc.OracleFrontEnd().run_f32(iq[c], 4096)
        assert np.array_equal(bits(disc[c]), bits(want)), c
    rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1)
    x, _, _ = orc.synth_p25_disc(3, B, n, frame_dibits=432)
    x[2, 3000:3050] = np.nan
    x[9, 7000] = np.inf
    x[11] = np.nan                                           # a channel that is NaN throughout
    rec, fl, cnt = rx.run(x)
    for c in range(B):
        if c in (2, 9, 11):
            assert 0 < cnt[c] <= rec.shape[1]
            continue
        o = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
        sym, rec4, flo = o.run(x[c])
        k = int(cnt[c])
        r4, sy = orc.unpack_records10(rec[c, :k])
        assert k == len(sym) and np.array_equal(sy.view(np.uint32), sym.view(np.uint32)) and np.array_equal(r4, rec4), c


def test_gardner_and_cqpsk_with_nan_inf_channels(built):
    import ctypes as C
    B, sps = 12, 5
    iq = orc.synth_dqpsk_f32(5, B, 1200, sps)
    n = iq.shape[1]
    iq[1, 400:420] = np.nan                                  # the timing loop zeroes NaN components like the reference
    iq[4, 2000, 0] = np.inf                                  # Inf is not sanitised: that channel's loop state goes non-finite
    iq[6] = np.inf
    b = ddn.CqpskBatch(B, rate=24000, block_len=2048)
    sym, cnt = b.run(iq)
    for c in range(B):
        if c in (1, 4, 6):
            assert 0 <= cnt[c] <= sym.shape[1]
            continue
        want = orc.OracleCqpskFe(rate=24000).run(iq[c], 2048)
        assert cnt[c] == len(want) and np.array_equal(bits(sym[c, :cnt[c]]), bits(want)), c
