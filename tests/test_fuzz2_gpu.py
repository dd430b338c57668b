"""GPU differential fuzz, second set: the Gardner loop, the rational resampler, the P25 slicer and the IQ-conditioned
front end on random shapes and call splits, bit-exact against the oracle.  DDN_FUZZ_BASE=<k> shifts the seeds."""
import os

import numpy as np
import pytest

import ddn
import orc
from test_iqcond_gpu import impaired_cu8
from test_resampler_gpu import GpuResampler
from test_slicer_gpu import GpuSlicer
from test_ted_gpu import GpuTed

pytestmark = pytest.mark.gpu

BASE = int(os.environ.get("DDN_FUZZ_BASE", "0"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_cuts(rng, n, k, lo=1):
    """k call boundaries inside (0, n), each call at least `lo` long."""
    if n <= 2 * lo:
        return [0, n]
    inner = sorted(set(int(v) for v in rng.integers(lo, n - lo, k)))
    cuts = [0]
    for v in inner:
        if v - cuts[-1] >= lo:
            cuts.append(v)
    if n - cuts[-1] < lo:
        cuts.pop()
    return cuts + [n]


@pytest.mark.parametrize("seed", range(16))
def test_gardner_random(built, seed):
    rng = np.random.default_rng(4000 + seed + 7919 * BASE)
    sps = int(rng.choice([2, 3, 4, 5, 8, 10, 12, 20, 23, 24, 25]))
    rate = int(rng.choice([4800, 6000]))
    B = int(rng.integers(1, 150))
    n_sym = int(rng.integers(30, 400))
    iq = orc.synth_qpsk_f32(int(rng.integers(0, 10000)), B, n_sym, sps, drift=float(rng.choice([1.0005, 0.9993, 1.002])),
                            noise=float(rng.choice([0.02, 0.1, 0.4])))
    n = iq.shape[1]
    if rng.integers(0, 3) == 0:
        iq[int(rng.integers(0, B)), n // 3: n // 3 + 25] = np.nan
    cuts = random_cuts(rng, n, int(rng.integers(0, 5)))
    t = GpuTed(B, sps, rate)
    got = [t.run(iq[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for c in range(B):
        o = orc.OracleTed(sps, rate)
        for (a, b), g in zip(zip(cuts[:-1], cuts[1:]), got):
            want = o.block(iq[c, a:b])
            assert np.array_equal(bits(g[c]), bits(want)), (seed, sps, rate, B, c, a, b, cuts)


@pytest.mark.parametrize("seed", range(16))
def test_resampler_random(built, seed):
    rng = np.random.default_rng(5000 + seed + 7919 * BASE)
    L = int(rng.integers(1, 40))
    M = int(rng.integers(1, 40))
    B = int(rng.integers(1, 50))
    n = int(rng.integers(1, 6000))
    x = (rng.normal(0, 9000, (B, n)) + 12000 * np.sin(np.arange(n) * 0.07)).astype(np.float32)
    cuts = random_cuts(rng, n, int(rng.integers(0, 6)))
    g = GpuResampler(B, L, M)
    got = [g.run(x[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    for c in range(B):
        o = orc.OracleResampler(L, M)
        for (a, b), gg in zip(zip(cuts[:-1], cuts[1:]), got):
            want = o.run(x[c, a:b])
            assert gg.shape[1] == len(want), (seed, L, M, c, a, b)
            assert np.array_equal(bits(gg[c]), bits(want)), (seed, L, M, c, a, b, cuts)


@pytest.mark.parametrize("seed", range(12))
def test_slicer_random(built, seed):
    rng = np.random.default_rng(6000 + seed + 7919 * BASE)
    B = int(rng.integers(1, 80))
    n = int(rng.integers(1, 4500))
    neg = int(rng.integers(0, 2))
    sym = np.stack([orc.synth_c4fm_symbols(int(rng.integers(0, 10000)), n, scale=float(rng.uniform(0.05, 1.4)),
                                           noise=float(rng.choice([50, 500, 3000]))) for _ in range(B)])
    if rng.integers(0, 3) == 0:
        sym[int(rng.integers(0, B)), n // 2:] = 0.0
    cuts = random_cuts(rng, n, int(rng.integers(0, 7)))
    s = GpuSlicer(B, neg)
    got = np.concatenate([s.run(sym[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
    rec, _ = orc.unpack_records10(got)
    want, thr = orc.oracle_slicer(sym, negative=neg)
    assert np.array_equal(rec, want), (seed, B, n, neg, cuts)
    for c in (0, B - 1):
        assert np.array_equal(s.thresholds(c).view(np.uint32), np.asarray(thr[c], np.float32).view(np.uint32)), (seed, c)


@pytest.mark.parametrize("seed", range(10))
def test_iq_conditioning_random(built, seed):
    rng = np.random.default_rng(7000 + seed + 7919 * BASE)
    dc = int(rng.integers(0, 2))
    bal = int(rng.integers(0, 2)) if dc else 1
    sh = int(rng.choice([6, 9, 11, 14]))
    thr = float(rng.choice([0.0, 0.02, 0.2]))
    ema = float(rng.choice([0.05, 0.2, 1.0]))
    sq = float(rng.choice([0.0, 0.0, 0.004]))
    blk = int(rng.choice([135, 200, 256, 1000, 4096, 8192]))
    B = int(rng.integers(1, 100))
    n_calls = int(rng.integers(1, 4))
    lens = [int(rng.integers(1, 4)) * blk for _ in range(n_calls - 1)] + [int(rng.integers(1, 3 * blk))]
    n = sum(lens)
    iq = np.stack([impaired_cu8(int(rng.integers(0, 10000)), n) for _ in range(B)])
    if sq > 0:
        iq[:, n // 3: n // 2] = 127
    b = ddn.Batch(B, block_len=blk, squelch_level=sq)
    b.set_iq_conditioning(dc, sh, bal, thr, ema)
    got, pos = [], 0
    for ln in lens:
        got.append(b.run_host(iq[:, pos:pos + ln], ln))
        pos += ln
    got = np.concatenate(got, axis=1)
    for c in range(B):
        fe = orc.OracleFrontEnd(squelch=sq).set_iq_options(dc, sh, bal, thr, ema)
        want, pos = [], 0
        for ln in lens:
            want.append(fe.run_cu8(iq[c, pos:pos + ln], blk))
            pos += ln
        want = np.concatenate(want)
        bad = np.flatnonzero(bits(got[c]) != bits(want))
        assert len(bad) == 0, (seed, c, dc, sh, bal, thr, ema, sq, blk, lens, bad[:5])


@pytest.mark.parametrize("seed", range(10))
def test_rx_random_wide_batches(built, seed):
    """The receive loop on batches wide enough for several wavefronts / workgroups of every kernel shape, with
    per-channel in-frame lengths and channels that differ in framing, level and noise."""
    rng = np.random.default_rng(8000 + seed + 7919 * BASE)
    B = int(rng.integers(24, 300))
    use_filter = int(rng.integers(0, 2))
    n = int(rng.integers(2500, 9000))
    parts, locks = [], []
    left = B
    while left > 0:
        k = int(min(left, rng.integers(1, 64)))
        frame = int(rng.choice([180, 360, 432, 864]))
        xs, _, _ = orc.synth_p25_disc(int(rng.integers(0, 9999)), k, n, frame_dibits=frame,
                                      noise=float(rng.choice([200, 800, 2500, 7000])))
        parts.append(xs * np.float32(rng.choice([0.3, 1.0, 1.0, 1.6])))
        locks += [frame - 24 if rng.random() < 0.6 else int(rng.integers(0, 900)) for _ in range(k)]
        left -= k
    x = np.concatenate(parts).astype(np.float32)
    if rng.integers(0, 2):
        x[int(rng.integers(0, B))] = 0.0
    lock = np.asarray(locks, np.int32)
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(1, n, int(rng.integers(0, 4)))]))
    rx = ddn.P25Rx(B, lock_symbols=156, use_matched_filter=use_filter, channels_per_wave=int(rng.choice([0, 8, 16, 32, 64])))
    assert ddn.lib().ddn_p25_rx_set_lock_symbols(rx.h, lock.ctypes.data) == 0
    recs, fls = [[] for _ in range(B)], [[] for _ in range(B)]
    for a, e in zip(cuts[:-1], cuts[1:]):
        rec, fl, cnt = rx.run(x[:, a:e])
        for c in range(B):
            recs[c].append(rec[c, :cnt[c]])
            fls[c].append(fl[c, :cnt[c]])
    for c in range(B):
        o = orc.OracleP25Rx(lock_symbols=int(lock[c]), use_filter=use_filter)
        sym, rec4, fl = o.run(x[c])
        r4, sy = orc.unpack_records10(np.concatenate(recs[c]))
        assert np.array_equal(sy.view(np.uint32), sym.view(np.uint32)), (seed, c, B, cuts)
        assert np.array_equal(r4, rec4) and np.array_equal(np.concatenate(fls[c]), fl), (seed, c, B, cuts)
        assert np.array_equal(rx.thresholds(c).view(np.uint32), o.thresholds().view(np.uint32)), (seed, c)


@pytest.mark.parametrize("seed", range(6))
def test_front_end_random_configs_16_channel_workgroups(built, seed):
    """More than 2048 channels select the 16-channels-per-workgroup shape of k_front_end_fused (the bench's shape);
    same random block lengths / call splits / formats / profiles as the small-batch fuzz, a sample of channels checked."""
    rng = np.random.default_rng(9000 + seed + 7919 * BASE)
    profile = int(rng.choice([2, 4, 4, 5, 1]))
    blk = int(rng.choice([135, 200, 256, 300, 1000, 2048, 8191]))
    B = int(rng.integers(2049, 2400))
    fmt_cf32 = bool(rng.integers(0, 2))
    n_calls = int(rng.integers(1, 4))
    lens = [int(rng.integers(1, 3)) * blk for _ in range(n_calls - 1)] + [int(rng.integers(1, 2 * blk))]
    lens = [min(v, 4000) // blk * blk or blk for v in lens[:-1]] + [min(lens[-1], 3000)]
    n = sum(lens)
    base = orc.synth_c4fm_cu8(int(rng.integers(0, 1000)), 67, n)
    iq = np.ascontiguousarray(np.tile(base, (B // 67 + 1, 1, 1))[:B])
    iq[:, :, 0] ^= (np.arange(B, dtype=np.uint8) & 1)[:, None]      # neighbouring channels differ
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32) if fmt_cf32 else iq
    b = ddn.Batch(B, lpf_profile=profile, block_len=blk, input_format=ddn.IN_CF32 if fmt_cf32 else ddn.IN_CU8)
    got, pos = [], 0
    for ln in lens:
        got.append(b.run_host(x[:, pos:pos + ln], ln))
        pos += ln
    got = np.concatenate(got, axis=1)
    pick = sorted(set([0, 1, 15, 16, 17, B - 17, B - 16, B - 1] + [int(v) for v in rng.integers(0, B, 40)]))
    for c in pick:
        fe = orc.OracleFrontEnd(profile=profile)
        want, pos = [], 0
        for ln in lens:
            want.append(fe.run_cu8(iq[c, pos:pos + ln], blk))
            pos += ln
        want = np.concatenate(want)
        bad = np.flatnonzero(bits(got[c]) != bits(want))
        assert len(bad) == 0, (seed, c, B, profile, blk, fmt_cf32, lens, bad[:5])


@pytest.mark.parametrize("seed", range(6))
def test_cqpsk_random_wide_batches(built, seed):
    """The CQPSK chain (lane = channel kernels) on batches of several wavefronts with a ragged last one."""
    rng = np.random.default_rng(10000 + seed + 7919 * BASE)
    sps = int(rng.choice([4, 5, 5, 10, 8]))
    sym_rate = 6000 if sps == 4 else 4800
    rate = sps * sym_rate
    blk = int(rng.choice([333, 1000, 2048, 4096]))
    lpf = int(rng.integers(0, 2))
    B = int(rng.integers(60, 260))
    k = 12
    base = orc.synth_dqpsk_f32(int(rng.integers(0, 999)), k, int(rng.integers(500, 1400)), sps,
                               cfo=float(rng.choice([0.0, 0.002, 0.01])))
    scale = (0.4 + 0.1 * (np.arange(B) % 13)).astype(np.float32)        # channels differ in level
    iq = np.ascontiguousarray(np.tile(base, (B // k + 1, 1, 1))[:B] * scale[:, None, None])
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
        assert len(g) == len(want), (seed, c, B, sps, blk, lens)
        assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), (seed, c, B, sps, blk, lpf, lens)


@pytest.mark.parametrize("seed", range(4))
def test_decimated_front_end_16_channel_workgroups(built, seed):
    """Half-band cascade in front of the fused kernel, above 2048 channels (the 16-channel workgroup shape on the
    decimated stream), two calls."""
    rng = np.random.default_rng(11000 + seed + 7919 * BASE)
    passes = int(rng.choice([1, 2]))
    blk = int(rng.choice([1024, 2048, 4096]))
    B = int(rng.integers(2049, 2300))
    fmt_cf32 = bool(rng.integers(0, 2))
    lens = [blk, max(1 << passes, (int(rng.integers(1, 2 * blk)) >> passes) << passes)]
    n = sum(lens)
    base = orc.synth_c4fm_cu8(int(rng.integers(0, 1000)), 33, n, sps=10 << passes)
    iq = np.ascontiguousarray(np.tile(base, (B // 33 + 1, 1, 1))[:B])
    iq[:, :, 1] ^= (np.arange(B, dtype=np.uint8) & 1)[:, None]
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32) if fmt_cf32 else iq
    b = ddn.Batch(B, block_len=blk, input_format=ddn.IN_CF32 if fmt_cf32 else ddn.IN_CU8)
    b.set_decimation(passes)
    got = np.concatenate([b.run_host(x[:, :lens[0]], lens[0]), b.run_host(x[:, lens[0]:], lens[1])], axis=1)
    for c in sorted(set([0, 16, B - 1] + [int(v) for v in rng.integers(0, B, 20)])):
        fe = orc.OracleFrontEnd(downsample_passes=passes)
        want = np.concatenate([fe.run_cu8(iq[c, :lens[0]], blk), fe.run_cu8(iq[c, lens[0]:], blk)])
        assert got.shape[1] == len(want), (seed, c)
        bad = np.flatnonzero(bits(got[c]) != bits(want))
        assert len(bad) == 0, (seed, c, B, passes, blk, fmt_cf32, lens, bad[:5])
