"""GPU differential fuzz: random front-end configurations (block length, call splits, channel counts, input format,
decimation passes, squelch, LPF profile) against the oracle, bit-exact; random receive-loop call splits and lock lengths."""
import os

import numpy as np
import pytest

import ddn
import orc

pytestmark = pytest.mark.gpu

# DDN_FUZZ_BASE=<k> shifts every generator seed, for longer sweeps than the default suite
BASE = int(os.environ.get("DDN_FUZZ_BASE", "0"))


@pytest.mark.parametrize("seed", range(48))
def test_front_end_random_configs(built, seed):
    rng = np.random.default_rng(1000 + seed + 7919 * BASE)
    passes = int(rng.choice([0, 0, 0, 1, 2]))
    profile = int(rng.choice([2, 4, 4, 5, 1]))
    blk = int(rng.choice([135, 200, 1000, 2048, 4096, 8192, 8191, 12345])) if passes == 0 else int(rng.choice([1024, 2048, 8192]))
    blk = max(blk, 135 << passes)
    squelch = float(rng.choice([0.0, 0.0, 0.0005, 0.02]))
    B = int(rng.integers(1, 40))
    fmt_cf32 = bool(rng.integers(0, 2))
    n_calls = int(rng.integers(1, 4))
    # every call but the last ends on a block boundary (the reference's blocks do not straddle calls)
    lens = [int(rng.integers(1, 4)) * blk for _ in range(n_calls - 1)] + [int(rng.integers(1, 3 * blk))]
    if passes:
        lens[-1] = max(1 << passes, (lens[-1] >> passes) << passes)
    n = sum(lens)
    iq = orc.synth_c4fm_cu8(int(rng.integers(0, 1000)), B, n, sps=10 << passes)
    if squelch > 0:
        iq[:, n // 3: n // 2] = 127                                  # a quiet stretch that the gate closes on
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32) if fmt_cf32 else iq
    b = ddn.Batch(B, lpf_profile=profile, block_len=blk, squelch_level=squelch,
                  input_format=ddn.IN_CF32 if fmt_cf32 else ddn.IN_CU8)
    if passes:
        b.set_decimation(passes)
    got, pos = [], 0
    for ln in lens:
        got.append(b.run_host(x[:, pos:pos + ln], ln))
        pos += ln
    got = np.concatenate(got, axis=1)
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=profile, squelch=squelch, downsample_passes=passes)
        want, pos = [], 0
        for ln in lens:
            want.append(fe.run_cu8(iq[c, pos:pos + ln], blk))
            pos += ln
        want = np.concatenate(want)
        assert got.shape[1] == len(want), (seed, c)
        bad = np.flatnonzero(got[c].view(np.uint32) != want.view(np.uint32))
        assert len(bad) == 0, (seed, c, passes, profile, blk, squelch, fmt_cf32, lens, bad[:5])


@pytest.mark.parametrize("seed", range(48))
def test_rx_random_splits(built, seed):
    rng = np.random.default_rng(2000 + seed + 7919 * BASE)
    B = int(rng.integers(1, 24))
    frame = int(rng.choice([180, 360, 432, 864]))
    lock = frame - 24 if rng.random() < 0.7 else int(rng.integers(0, frame))
    use_filter = int(rng.integers(0, 2))
    n = int(rng.integers(3000, 30000))
    # seeds < 24 keep the default 48 kHz / 10 samples per symbol; the rest walk the samples-per-symbol cases of
    # getSymbol(): 5 (single-sample window), 8, 12, 20 (wide window), and a fractional rate (accumulator carries)
    out_rate, sps = 48000, 10
    if seed >= 24:
        out_rate, sps = [(24000, 5), (38400, 8), (57600, 12), (96000, 20), (50000, 10), (48000, 10)][seed % 6]
        use_filter = use_filter if out_rate == 48000 else 0  # the matched filter is designed for 48 kHz only
    x, _, _ = orc.synth_p25_disc(int(rng.integers(0, 999)), B, n, frame_dibits=frame, sps=sps,
                                 noise=float(rng.choice([200, 800, 2500])))
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(1, n, int(rng.integers(0, 5)))]))
    rx = ddn.P25Rx(B, out_rate=out_rate, lock_symbols=lock, use_matched_filter=use_filter,
                   channels_per_wave=int(rng.choice([0, 8, 16, 32, 64])))
    recs, fls = [[] for _ in range(B)], [[] for _ in range(B)]
    for a, e in zip(cuts[:-1], cuts[1:]):
        rec, fl, cnt = rx.run(x[:, a:e])
        for c in range(B):
            recs[c].append(rec[c, :cnt[c]])
            fls[c].append(fl[c, :cnt[c]])
    for c in range(B):
        o = orc.OracleP25Rx(out_rate=out_rate, lock_symbols=lock, use_filter=use_filter)
        sym, rec4, fl = o.run(x[c])
        r4, sy = orc.unpack_records10(np.concatenate(recs[c]))
        assert np.array_equal(sy.view(np.uint32), sym.view(np.uint32)), (seed, c)
        assert np.array_equal(r4, rec4) and np.array_equal(np.concatenate(fls[c]), fl), (seed, c)
        assert np.array_equal(rx.thresholds(c).view(np.uint32), o.thresholds().view(np.uint32)), (seed, c)


@pytest.mark.parametrize("seed", range(12))
def test_cqpsk_random_configs(built, seed):
    rng = np.random.default_rng(3000 + seed + 7919 * BASE)
    sps = int(rng.choice([4, 5, 5, 10, 8]))
    sym_rate = 6000 if sps == 4 else 4800
    rate = sps * sym_rate
    blk = int(rng.choice([333, 1000, 2048, 4096, 8192]))
    lpf = int(rng.integers(0, 2))
    B = int(rng.integers(1, 20))
    iq = orc.synth_dqpsk_f32(int(rng.integers(0, 999)), B, int(rng.integers(600, 2500)), sps, cfo=float(rng.choice([0.0, 0.002, 0.01])))
    n = iq.shape[1]
    n_calls = int(rng.integers(1, 4))
    lens = [int(rng.integers(1, 3)) * blk for _ in range(n_calls - 1)]
    if sum(lens) >= n - 8:
        lens = []
    last = n - sum(lens)
    if last % blk in (1, 2, 3):
        last -= 4
    lens.append(last)
    b = ddn.CqpskBatch(B, rate=rate, sym_rate=sym_rate, lpf_enable=lpf, block_len=blk)
    got = [[] for _ in range(B)]
    pos = 0
    for ln in lens:
        sym, cnt = b.run(iq[:, pos:pos + ln])
        for c in range(B):
            got[c].append(sym[c, :cnt[c]])
        pos += ln
    for c in range(B):
        fe = orc.OracleCqpskFe(rate=rate, sym_rate=sym_rate, lpf_enable=lpf)
        want, pos = [], 0
        for ln in lens:
            want.append(fe.run(iq[c, pos:pos + ln], blk))
            pos += ln
        want = np.concatenate(want)
        g = np.concatenate(got[c])
        assert len(g) == len(want), (seed, c, sps, blk, lens)
        assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), (seed, c, sps, blk, lpf, lens)
