"""GPU parity tests proper: the HIP front end (through the C-ABI) against the reference's golden vectors and the
CPU oracle on the same inputs.

Bar: BIT-IDENTICAL, every sample and the carried modem state.  The channel LPF, dc/peak recurrences, AGC scaling and clipping
run the reference's IEEE-754 op sequence; the small-angle phase polynomial (src/dsp/fsk_modem.c:23-33) likewise; the large-angle
branch calls libm atan2f in the reference, and the kernel evaluates glibc 2.35's binary32 algorithm for it operation by operation
(dsd-neo_amd/csrc/ddn_atan2f.h), so there is no tolerance anywhere in this file.
"""
import numpy as np
import pytest

import ddn
import orc
from conftest import golden

pytestmark = pytest.mark.gpu

def check(got, want, exact=True):
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), "mismatch at %s" % (np.argwhere(~same)[:5],)


@pytest.mark.parametrize("name", ["fe_p25p1_vc_b8192.npz", "fe_p25p1_vc_b3000_sq.npz", "fe_p25p1_cc_b8192.npz",
                                  "fe_nxdn48_b4096.npz", "fe_synth_ch0_b8192.npz"])
def test_golden_vectors(built, name):
    g = golden(name)
    iq = g["iq"][None]
    b = ddn.Batch(1, sample_rate_hz=int(g["rate"]), lpf_profile=int(g["profile"]), block_len=int(g["block_len"]),
                  squelch_level=float(g["squelch"]))
    assert np.array_equal(b.taps().view(np.uint32), g["taps"].view(np.uint32))
    got = b.run_host(iq, iq.shape[1])[0]
    check(got, g["disc"])
    st = b.fsk_state(0)
    # {prev_i, prev_q, have_prev, dc, peak} vs the reference's modem state after the same stream
    want = np.asarray(g["state"], np.float32)[:5]
    assert np.array_equal(np.asarray(st, np.float32)[:5].view(np.uint32), want.view(np.uint32)), (st, want)


def test_batch_matches_oracle_bitexact_on_synthetic(built):
    """Synthetic C4FM stays inside the polynomial branch -> bit-exact for every sample."""
    B, n, blk = 96, 20000, 8192
    iq = orc.synth_c4fm_cu8(100, B, n)
    got = ddn.Batch(B, block_len=blk).run_host(iq, n)
    want = orc.oracle_batch_cu8(iq, blk)
    check(got, want, exact=True)
    assert (got[:, 0] == 0.0).all()


@pytest.mark.parametrize("blk,n", [(2048, 6000), (3000, 9001), (135, 1000), (8192, 100), (50000, 50000)])
def test_block_sizes_and_ragged_tails(built, blk, n):
    B = 5
    iq = orc.synth_c4fm_cu8(3, B, n)
    got = ddn.Batch(B, block_len=blk).run_host(iq, n)
    if n < 135:
        # shorter than one tap set: the reference routes such blocks to its non-fused unit
        pytest.skip("covered by drop-in test")
    want = orc.oracle_batch_cu8(iq, blk)
    check(got, want, exact=(n >= 270))


def test_streaming_calls_carry_state(built):
    """Two calls (block-aligned split) == one call: FIR look-back and modem state persist in the batch."""
    B, blk = 7, 4096
    n1, n2 = 3 * blk, 2 * blk + 777
    iq = orc.synth_c4fm_cu8(11, B, n1 + n2)
    b = ddn.Batch(B, block_len=blk)
    a = b.run_host(iq[:, :n1], n1)
    c = b.run_host(iq[:, n1:], n2)
    want = orc.oracle_batch_cu8(iq, blk)
    check(np.concatenate([a, c], axis=1), want, exact=True)
    b.reset()
    again = b.run_host(iq[:, :n1], n1)
    check(again, a, exact=True)


def test_cf32_input(built):
    B, n, blk = 3, 12000, 4096
    u8 = orc.synth_c4fm_cu8(21, B, n)
    f32 = ((u8.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)).astype(np.float32)
    got = ddn.Batch(B, block_len=blk, input_format=ddn.IN_CF32).run_host(f32, n)
    want = orc.oracle_batch_cu8(u8, blk)
    check(got, want, exact=True)


def test_squelch_gate_mixed_blocks(built):
    B, n, blk = 4, 16384, 2048
    iq = orc.synth_c4fm_cu8(31, B, n)
    iq[:, 4096:8192] = 127  # dead air in the middle -> squelched blocks, then modem restart
    iq[1, 12000:] = 128
    got = ddn.Batch(B, block_len=blk, squelch_level=0.01).run_host(iq, n)
    want = orc.oracle_batch_cu8(iq, blk, squelch=0.01)
    assert (want[:, 4096 + 2048:8192] == 0).all()
    check(got, want)


@pytest.mark.parametrize("blk", [135, 200, 256, 257, 512])
def test_squelch_gate_one_tile_blocks(built, blk):
    """Blocks no longer than the kernel's 256-sample tile: every tile starts a block, so the block-power sum of one
    tile runs beside the filter pass of the next (found by the DDN_FUZZ_BASE sweep: loud samples of the following
    block reached the sum of a quiet one).  Quiet / loud edges on every block phase, two calls."""
    B = 11
    n1, n2 = 5 * blk, 7 * blk + 33
    iq = orc.synth_c4fm_cu8(77, B, n1 + n2)
    for c in range(B):
        iq[c, 2 * blk + 17 * c: 6 * blk + 9 * c] = 127
        iq[c, 9 * blk: 10 * blk] = 127 + (c & 1)
    b = ddn.Batch(B, block_len=blk, squelch_level=0.02)
    got = np.concatenate([b.run_host(iq[:, :n1], n1), b.run_host(iq[:, n1:], n2)], axis=1)
    for c in range(B):
        fe = orc.OracleFrontEnd(squelch=0.02)
        want = np.concatenate([fe.run_cu8(iq[c, :n1], blk), fe.run_cu8(iq[c, n1:], blk)])
        assert (want[4 * blk:5 * blk] == 0).all()
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), (blk, c)


def test_other_rates_and_profiles(built):
    # 24 kHz -> 67 taps (unrolled centre 33); 32 kHz -> 89 taps (generic kernel)
    for rate, prof in [(24000, 1), (32000, 2), (48000, 5), (48000, 0)]:
        B, n, blk = 2, 9000, 4096
        iq = orc.synth_c4fm_cu8(41, B, n)
        got = ddn.Batch(B, sample_rate_hz=rate, lpf_profile=prof, block_len=blk).run_host(iq, n)
        want = orc.oracle_batch_cu8(iq, blk, rate=rate, profile=prof)
        check(got, want)


def test_noise_input_takes_atan2_branch(built):
    rng = np.random.default_rng(5)
    B, n, blk = 4, 20000, 8192
    iq = rng.integers(0, 256, size=(B, n, 2), dtype=np.uint8)
    got = ddn.Batch(B, block_len=blk).run_host(iq, n)
    want = orc.oracle_batch_cu8(iq, blk)
    check(got, want)


def test_dropin_symbols_match_oracle(built):
    import ctypes as C
    l = ddn.lib()
    o = orc.oracle()
    rng = np.random.default_rng(9)
    taps = np.zeros(144, np.float32)
    nt = o.orc_channel_lpf_design(48000, 4, taps.ctypes.data, 144)
    for n in (2000, 60):  # fused order / short-block non-fused order
        x = rng.standard_normal(2 * n).astype(np.float32)
        hi = rng.standard_normal(nt - 1).astype(np.float32)
        hq = rng.standard_normal(nt - 1).astype(np.float32)
        hi2, hq2 = hi.copy(), hq.copy()
        y1 = np.zeros(2 * n, np.float32)
        y2 = np.zeros(2 * n, np.float32)
        l.simd_fir_complex_apply(x.ctypes.data, 2 * n, y1.ctypes.data, hi.ctypes.data, hq.ctypes.data,
                                 taps.ctypes.data, nt)
        o.orc_fir_complex_apply(x.ctypes.data, 2 * n, y2.ctypes.data, hi2.ctypes.data, hq2.ctypes.data,
                                taps.ctypes.data, nt, 1)
        check(y1, y2, exact=True)
        check(hi, hi2, exact=True)
        check(hq, hq2, exact=True)
    # half-band /2, 31 and 15 taps
    hb31 = np.ctypeslib.as_array((C.c_float * 31).in_dll(o, "orc_hb31_taps")).copy()
    hb15 = np.ctypeslib.as_array((C.c_float * 15).in_dll(o, "orc_hb15_taps")).copy()
    # 23 taps: the third length include/dsd-neo/dsp/simd_fir.h:54,68 names (no prototype ships; a Hamming-windowed one here)
    k = np.arange(23) - 11
    hb23 = np.where(k == 0, 0.5, np.where(k % 2 != 0, np.sin(np.pi * k / 2) / (np.pi * np.where(k == 0, 1, k)), 0.0))
    hb23 = (hb23 * np.hamming(23)).astype(np.float32)
    for hb in (hb31, hb15, hb23):
        n = 3001
        x = rng.standard_normal(2 * n).astype(np.float32)
        hi = rng.standard_normal(len(hb) - 1).astype(np.float32)
        hq = rng.standard_normal(len(hb) - 1).astype(np.float32)
        hi2, hq2 = hi.copy(), hq.copy()
        y1 = np.zeros(2 * n, np.float32)
        y2 = np.zeros(2 * n, np.float32)
        r1 = l.simd_hb_decim2_complex(x.ctypes.data, 2 * n, y1.ctypes.data, hi.ctypes.data, hq.ctypes.data,
                                      hb.ctypes.data, len(hb))
        r2 = o.orc_hb_decim2_complex(x.ctypes.data, 2 * n, y2.ctypes.data, hi2.ctypes.data, hq2.ctypes.data,
                                     hb.ctypes.data, len(hb), 1)
        assert r1 == r2 == 2 * (n // 2)
        check(y1[:r1], y2[:r2], exact=True)
        check(hi, hi2, exact=True)
    # widen
    u = rng.integers(0, 256, 5001, dtype=np.uint8)
    w1 = np.zeros(5001, np.float32)
    w2 = np.zeros(5001, np.float32)
    l.widen_u8_to_f32_bias127(u.ctypes.data, w1.ctypes.data, 5001)
    o.orc_widen_u8(u.ctypes.data, w2.ctypes.data, 5001)
    check(w1, w2, exact=True)
    # widen + raw byte moments, rotate-by-j^n variants (block sizes that split waves; phases 0..3; an odd trailing byte)
    def np_moments(b):
        b = b.astype(np.uint64)
        return (int(b.size), int(b.sum()), int((b * b).sum()), int(((b <= 1) | (b >= 254)).sum()), int(b.min()), int(b.max()))

    def rot(u8, phase):
        z = ((u8[:u8.size & ~1].astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)).reshape(-1, 2)
        ph = (phase + np.arange(len(z))) & 3
        out = np.empty_like(z)
        out[:, 0] = np.select([ph == 0, ph == 1, ph == 2], [z[:, 0], -z[:, 1], -z[:, 0]], z[:, 1])
        out[:, 1] = np.select([ph == 0, ph == 1, ph == 2], [z[:, 1], z[:, 0], -z[:, 1]], -z[:, 0])
        return out.reshape(-1)

    for ln, phase in [(5001, 0), (130, 1), (4096, 2), (2, 3), (777, 7)]:
        u = rng.integers(0, 256, ln, dtype=np.uint8)
        u[:2] = (0, 255)
        m = ddn.Cu8Moments(0, 0, 0, 0, 255, 0)
        w = np.zeros(ln, np.float32)
        l.widen_u8_to_f32_bias127_moments(u.ctypes.data, w.ctypes.data, ln, C.byref(m))
        o.orc_widen_u8(u.ctypes.data, w2.ctypes.data, ln)
        check(w, w2[:ln], exact=True)
        assert (m.count, m.sum, m.sum_sq, m.clipped, m.min_sample, m.max_sample) == np_moments(u)
        l.widen_u8_to_f32_bias127_moments(u.ctypes.data, w.ctypes.data, ln, C.byref(m))         # second block merges
        assert (m.count, m.sum, m.clipped) == (2 * ln, 2 * int(u.astype(np.uint64).sum()), 2 * np_moments(u)[3])
        m = ddn.Cu8Moments(0, 0, 0, 0, 255, 0)
        w[:] = 7.0
        nxt = l.widen_rotate90_u8_to_f32_bias127_phase_moments(u.ctypes.data, w.ctypes.data, ln, phase, C.byref(m))
        assert nxt == ((phase & 3) + ln // 2) & 3
        even = ln & ~1
        check(w[:even], rot(u, phase & 3), exact=True)
        assert np.all(w[even:] == 7.0)
        assert (m.count, m.sum, m.sum_sq, m.clipped, m.min_sample, m.max_sample) == np_moments(u[:even])
        w[:] = 0
        assert l.widen_rotate90_u8_to_f32_bias127_phase(u.ctypes.data, w.ctypes.data, ln, phase) == nxt
        check(w[:even], rot(u, phase & 3), exact=True)
        if orc.have_ref():
            r = orc.ref()
            r.widen_rotate90_u8_to_f32_bias127_phase_moments.restype = C.c_uint32
            r.widen_rotate90_u8_to_f32_bias127_phase_moments.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
            mr = ddn.Cu8Moments(0, 0, 0, 0, 255, 0)
            wr = np.zeros(ln, np.float32)
            assert r.widen_rotate90_u8_to_f32_bias127_phase_moments(u.ctypes.data, wr.ctypes.data, ln, phase, C.byref(mr)) == nxt
            check(w[:even], wr[:even], exact=True)
            assert bytes(m)[:34] == bytes(mr)[:34]
    # discriminator with carried state
    n = 3000
    ph = np.cumsum(rng.normal(0.0, 0.15, n))
    iq = np.stack([np.cos(ph), np.sin(ph)], axis=1).astype(np.float32)
    st = ddn.FskModemState(48000, 4800, 4, 4, 0, 0, 0, 0, 0)
    ost = np.zeros(5, np.float32)
    d1 = np.zeros(n, np.float32)
    d2 = np.zeros(n, np.float32)
    for a, bnd in ((0, 1000), (1000, n)):
        c1 = l.ddn_fsk_modem_discriminator_process(C.byref(st), iq[a:bnd].ctypes.data, 2 * (bnd - a),
                                                   d1[a:].ctypes.data, bnd - a)
        c2 = o.orc_fsk_discriminator(ost.ctypes.data, iq[a:bnd].ctypes.data, 2 * (bnd - a), d2[a:].ctypes.data,
                                     bnd - a)
        assert c1 == c2 == bnd - a
    check(d1, d2, exact=True)
    assert st.dc_est == ost[3] and st.discriminator_peak_est == ost[4]


def test_full_size_c2_properties(built):
    """BASELINE config 2 shape: 4096 channels x 48000 samples on one GPU.  Whole-batch properties + a sample of
    channels against the oracle."""
    import torch
    B, n, blk = 4096, 48000, 8192
    parts = [torch.from_numpy(orc.synth_c4fm_cu8(c, 256, n)) for c in range(0, B, 256)]
    iq = torch.cat(parts, 0)
    d_in = iq.to("cuda:0")
    d_out = torch.empty((B, n), dtype=torch.float32, device="cuda:0")
    b = ddn.Batch(B, block_len=blk)
    st = torch.cuda.current_stream().cuda_stream
    b.run_device(d_in.data_ptr(), n, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    first = d_out.clone()
    assert torch.isfinite(first).all()
    assert float(first.abs().max()) <= 32768.0
    assert (first[:, 0] == 0).all()
    # determinism / idempotence after reset: identical bits
    b.reset(st)
    b.run_device(d_in.data_ptr(), n, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(first.view(torch.int32), d_out.view(torch.int32))
    pick = [0, 1, 255, 256, 1023, 2048, 4095] + list(range(17, 4096, 409))
    want = orc.oracle_batch_cu8(iq[pick].numpy(), blk)
    check(first[pick].cpu().numpy(), want, exact=True)


@pytest.mark.parametrize("passes,blk,fmt", [(1, 8192, "cu8"), (2, 8192, "cu8"), (3, 4096, "cu8"), (1, 2048, "cf32"),
                                            (2, 1200, "cu8")])
def test_halfband_decimation_cascade(built, passes, blk, fmt):
    """ddn_batch_set_decimation: 31-tap + 15-tap half-band stages ahead of the LPF, carried across calls, ragged last
    block, both input formats — bit-exact with the oracle (itself pinned to full_demod with downsample_passes)."""
    B = 6
    n1, n2 = 3 * blk, blk + (blk // 2)            # second call ends in a half block
    iq = orc.synth_c4fm_cu8(40, B, n1 + n2, sps=10 << passes)
    if fmt == "cf32":
        x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
        b = ddn.Batch(B, block_len=blk, input_format=ddn.IN_CF32)
    else:
        x = iq
        b = ddn.Batch(B, block_len=blk)
    b.set_decimation(passes)
    got = np.concatenate([b.run_host(x[:, :n1], n1), b.run_host(x[:, n1:], n2)], axis=1)
    for c in range(B):
        fe = orc.OracleFrontEnd(downsample_passes=passes)
        want = np.concatenate([fe.run_cu8(iq[c, :n1], blk), fe.run_cu8(iq[c, n1:], blk)])
        assert got.shape[1] == len(want)
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c


def test_single_stream_adapter_full_demod_and_gardner(built):
    """B4: full_demod(struct demod_state*) / op25_gardner_cc(struct demod_state*) under the reference's names
    (include/ddn_demod_adapter.h): block after block on one stream, FSK-discriminator and CQPSK output kinds, equal to the
    oracle run with the same block boundaries (carried state included); the Gardner entry writes its symbols back into
    lowpassed like the reference."""
    import ctypes as C
    l = ddn.lib()
    rng = np.random.default_rng(21)
    # --- FSK discriminator output
    iq = orc.synth_c4fm_cu8(5, 1, 9000)[0]
    x = ((iq.astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)).astype(np.float32)
    fe = orc.OracleFrontEnd()
    s = ddn.DemodState(rate_in=48000, rate_out=48000, output_kind=1, symbol_rate_hz=4800, symbol_levels=4,
                       channel_lpf_enable=1, channel_lpf_profile=ddn.LPF_P25_C4FM)
    pos = 0
    for ln in (4000, 137, 3000, 1863):
        blk = np.ascontiguousarray(x[pos:pos + ln])
        s.lowpassed = blk.ctypes.data_as(C.POINTER(C.c_float))
        s.lp_len = 2 * ln
        l.full_demod(C.byref(s))
        want = fe.run_f32(blk, ln)
        assert s.result_len == ln
        out = np.frombuffer(s.result, np.float32, ln)      # s->result[i], the reference's in-struct array
        check(out, want, exact=True)
        pos += ln
    l.ddn_demod_state_release(C.byref(s))
    assert not s.ddn_adapter
    # --- CQPSK symbol output (24 ksps, 5 samples per symbol)
    sig = orc.synth_dqpsk_f32(3, 1, 1500, 5)[0]
    cq = orc.OracleCqpskFe(rate=24000, sym_rate=4800, profile=5, lpf_enable=1)
    s2 = ddn.DemodState(rate_in=24000, rate_out=24000, output_kind=2, symbol_rate_hz=4800, symbol_levels=4, channel_lpf_enable=1,
                        channel_lpf_profile=ddn.LPF_P25_CQPSK, cqpsk_enable=1, ted_enabled=1, ted_sps=5)
    pos = 0
    for ln in (3000, 2048, 2400):
        blk = np.ascontiguousarray(sig[pos:pos + ln])
        s2.lowpassed = blk.ctypes.data_as(C.POINTER(C.c_float))
        s2.lp_len = 2 * ln
        l.full_demod(C.byref(s2))
        want = cq.run(blk, ln)
        assert s2.result_len == len(want) and len(want) > ln // 6
        out = np.frombuffer(s2.result, np.float32, len(want))
        check(out, want, exact=True)
        pos += ln
    l.ddn_demod_state_release(C.byref(s2))
    # --- op25_gardner_cc: symbols written back into lowpassed
    ted = orc.OracleTed(5, 4800)
    s3 = ddn.DemodState(cqpsk_enable=1, ted_sps=5, symbol_rate_hz=4800)
    pos = 0
    for ln in (2500, 1201, 3):
        blk = np.ascontiguousarray(sig[pos:pos + ln]).copy()
        keep = blk.copy()
        s3.lowpassed = blk.ctypes.data_as(C.POINTER(C.c_float))
        s3.lp_len = 2 * ln
        l.op25_gardner_cc(C.byref(s3))
        if ln < 4:
            assert s3.lp_len == 2 * ln and np.array_equal(blk, keep)      # the reference returns early below four samples
        else:
            want = ted.block(keep)
            assert s3.lp_len == 2 * len(want)
            check(blk.reshape(-1)[:2 * len(want)], want.reshape(-1), exact=True)
        pos += ln
    l.ddn_demod_state_release(C.byref(s3))


@pytest.mark.parametrize("counts", [(21, 19, 10), (5, 30), (16, 16, 16), (1, 1, 45)])
def test_segments_one_launch_equals_a_batch_per_group(built, counts):
    """ddn_batch_set_segments / ddn_front_end_run_segments: the protocol groups of a mixed batch (a channel low-pass profile, an input
    array and an output array each) behind one launch of ceil(total / 16) workgroups - workgroups that hold channels of two groups
    included - give every group what a batch object of its own gives it, bit for bit, over two calls (carried look-back and modem
    state in the shared batch) and with sixteen or eight channels per workgroup"""
    import ctypes as C
    l = ddn.lib()
    profiles = [ddn.LPF_P25_C4FM, ddn.LPF_12K5, ddn.LPF_6K25][:len(counts)]
    n1, n2, blk = 10000, 6000, 4096
    iqs = [orc.synth_c4fm_cu8(40 + k, B, n1 + n2) for k, B in enumerate(counts)]
    want = []
    for iq, B, prof in zip(iqs, counts, profiles):
        b = ddn.Batch(B, lpf_profile=prof, block_len=blk)
        want.append(np.concatenate([b.run_host(iq[:, :n1], n1), b.run_host(iq[:, n1:], n2)], axis=1))
        b.close()

    def dev(a):
        p = C.c_void_p()
        assert l.ddn_device_alloc(a.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
        return p

    for group in (0, 8):
        b = ddn.Batch(sum(counts), lpf_profile=profiles[0], block_len=blk)
        cnt, prof = np.array(counts, np.int32), np.array(profiles, np.int32)
        assert l.ddn_batch_set_segments(b.h, len(counts), cnt.ctypes.data, prof.ctypes.data) == 0
        assert l.ddn_batch_set_channels_per_workgroup(b.h, group) == 0
        got = [[] for _ in counts]
        for lo, n in ((0, n1), (n1, n2)):
            ins = [dev(np.ascontiguousarray(iq[:, lo:lo + n])) for iq in iqs]
            outs = [dev(np.zeros((B, n), np.float32)) for B in counts]
            ia, oa = (C.c_void_p * len(counts))(*ins), (C.c_void_p * len(counts))(*outs)
            assert l.ddn_front_end_run_segments(b.h, ia, n, oa, None) == 0
            for k, B in enumerate(counts):
                h = np.zeros((B, n), np.float32)
                assert l.ddn_device_download(h.ctypes.data, outs[k], h.nbytes) == 0
                got[k].append(h)
            for p in ins + outs:
                l.ddn_device_free(p)
        for k in range(len(counts)):
            a = np.concatenate(got[k], axis=1)
            assert np.array_equal(a.view(np.uint32), want[k].view(np.uint32)), (group, k)
        b.close()
    # the segments must add up, keep the batch's profile first, and design one tap count
    b = ddn.Batch(10, block_len=blk)
    bad = np.array([4, 5], np.int32)
    assert l.ddn_batch_set_segments(b.h, 2, bad.ctypes.data, np.array([ddn.LPF_P25_C4FM, ddn.LPF_12K5], np.int32).ctypes.data) == -1
    b.close()


def test_large_angle_branch_on_noise_of_every_kind(built):
    """the discriminator's atan2f as straight-line code (ddn_atan2f.h: one division for atanf's four argument reductions, every rare
    case through one branch to the branchy original): ~5 million evaluations against the host libm the oracle calls - noise at full
    scale, noise of a few counts around mid-scale (zeros in I or Q: the iy == 0 / ix == 0 cases), a constant input, and cf32 samples
    whose magnitudes span sixty binades between neighbours (the exponent-gap and |q| >= 2^25 / < 2^-29 cases)"""
    rng = np.random.default_rng(77)
    B, n, blk = 48, 50000, 8192
    iq = rng.integers(0, 256, size=(B, n, 2), dtype=np.uint8)
    iq[16:32] = rng.integers(125, 131, size=(16, n, 2), dtype=np.uint8)
    iq[32:40] = rng.integers(127, 129, size=(8, n, 2), dtype=np.uint8)
    iq[40:44, :, 0] = 128
    iq[44:48] = 127
    got = ddn.Batch(B, block_len=blk).run_host(iq, n)
    want = orc.oracle_batch_cu8(iq, blk)
    check(got, want)
    assert np.count_nonzero(np.abs(want[:16]) > 20000) > 1000        # (large angles were there)
    # cf32: every sample's magnitude drawn from 2^-40 .. 2^40, and runs of exact zeros
    Bf, nf = 6, 30000
    mag = np.exp2(rng.integers(-40, 41, size=(Bf, nf, 1))).astype(np.float32)
    x = (rng.standard_normal((Bf, nf, 2)).astype(np.float32) * mag).astype(np.float32)
    x[2, 1000:1400] = 0.0
    x[3, ::7, 1] = 0.0
    x[4, ::5, 0] = 0.0
    gotf = ddn.Batch(Bf, block_len=4096, input_format=ddn.IN_CF32).run_host(x, nf)
    wantf = np.stack([orc.OracleFrontEnd().run_f32(x[c], 4096) for c in range(Bf)])
    check(gotf, wantf)
