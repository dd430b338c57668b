#!/usr/bin/env python3
"""Per-stage throughput of the batched FEC / timing kernels on one MI355X, next to the CPU oracle on one core.
Prints one JSON line per stage.  (bench.py stays the BASELINE metric; this is the per-row measurement of SURVEY §8.)

    python tools/bench_stages.py [--reps 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch
    import ddn
    import fecgen
    import orc
    l = ddn.lib()
    rng = np.random.default_rng(1)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    def cpu(fn, n_items):
        t0 = time.perf_counter()
        fn()
        return n_items / (time.perf_counter() - t0)

    def nid_case(tag, inv):
        from test_oracle_block import oracle_nid
        n2 = 4096 * 8
        bits, rel63, obs, par, prel = fecgen.gen_nid(rng, 4096, max_err=9, invalid_duids=inv)
        db = torch.from_numpy(np.tile(bits, (8, 1))).cuda()
        drl = torch.from_numpy(np.tile(rel63, (8, 1))).cuda()
        dob = torch.from_numpy(np.tile(obs, 8)).cuda()
        dp = torch.from_numpy(np.tile(par, 8)).cuda()
        dpr = torch.from_numpy(np.tile(prel, 8)).cuda()
        dres = torch.zeros((n2, 4), dtype=torch.int32, device="cuda")
        ms = timeit(lambda: l.ddn_p25p1_nid_decode_batch(db.data_ptr(), drl.data_ptr(), dob.data_ptr(), dp.data_ptr(),
                                                         dpr.data_ptr(), 64, n2, dres.data_ptr(), st))
        report(tag, n2, ms, 63 * 2 + 16, cpu(lambda: oracle_nid(bits[:512], rel63[:512], obs[:512], par[:512],
                                                                  prel[:512]), 512), "NIDs")

    def report(stage, n, ms, bytes_per_item, cpu_items_s, unit):
        gps = n / (ms * 1e-3)
        print(json.dumps({"stage": stage, "n": n, "ms": round(ms, 4), "items_per_s": round(gps, 1), "unit": unit,
                          "algorithmic_GBps": round(gps * bytes_per_item / 1e9, 2),
                          "cpu_oracle_items_per_s_1core": round(cpu_items_s, 1),
                          "speedup_vs_1core": round(gps / cpu_items_s, 1)}))

    # P25 1/2-rate trellis: 4096 channels x 26 blocks
    n = 4096 * 26
    llr, _ = fecgen.gen_p25_half_rate(rng, 4096, sigma=400.0)
    d_in = torch.from_numpy(np.tile(llr, (26, 1))).cuda()
    d_out = torch.zeros((n, 12), dtype=torch.uint8, device="cuda")
    d_met = torch.zeros(n, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_fec_p25_12_soft_batch(d_in.data_ptr(), n, d_out.data_ptr(), d_met.data_ptr(), st))
    report("p25_half_rate_trellis", n, ms, 196 * 2 + 12 + 4, cpu(lambda: fecgen.oracle_p25_half_rate(llr[:512]), 512),
           "codewords")

    # 3/4-rate trellis (soft)
    d, rel, _ = fecgen.gen_r34(rng, 4096)
    dd = torch.from_numpy(np.tile(d, (26, 1))).cuda()
    dr = torch.from_numpy(np.tile(rel, (26, 1))).cuda()
    do = torch.zeros((n, 18), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_fec_r34_batch(dd.data_ptr(), dr.data_ptr(), n, do.data_ptr(), st))
    report("r34_trellis_soft", n, ms, 98 * 2 + 18, cpu(lambda: fecgen.oracle_r34(d[:512], rel[:512]), 512), "codewords")

    # K=5 NXDN (FACCH-sized) and M17 LSF
    sym, rl = fecgen.gen_nxdn(rng, 4096, 96)
    ds = torch.from_numpy(np.tile(sym, (26, 1))).cuda()
    do2 = torch.zeros((n, 12), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_fec_nxdn_conv_batch(ds.data_ptr(), None, n, 96, 92, None, do2.data_ptr(), 12, st))
    report("k5_nxdn_96steps", n, ms, 192 + 12, cpu(lambda: fecgen.oracle_nxdn(sym[:512], None, 96, 92), 512),
           "codewords")
    soft = fecgen.gen_m17(rng, 4096, 488)
    dsf = torch.from_numpy(np.tile(soft, (8, 1))).cuda()
    n2 = 4096 * 8
    do3 = torch.zeros((n2, 32), dtype=torch.uint8, device="cuda")
    dc = torch.zeros(n2, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_fec_viterbi_k5_batch(dsf.data_ptr(), n2, 488, None, 0, do3.data_ptr(), 32, dc.data_ptr(),
                                                   st))
    report("k5_m17_244steps", n2, ms, 488 * 2 + 32 + 4, cpu(lambda: fecgen.oracle_m17(soft[:256]), 256), "codewords")

    # NID: first the common case (every NID decodes on the hard path), then a stress mix where ~9 % carry an
    # undefined DUID and fall into the 93..186-trial Chase search
    for tag, inv in (("p25p1_nid_hard_path", False), ("p25p1_nid_9pct_chase", True)):
        nid_case(tag, inv)

    # Gardner: 4096 channels x 48000 samples
    B, sps = 4096, 10
    iq1 = orc.synth_qpsk_f32(9, 8, 4900, sps)
    nn = iq1.shape[1]
    d_iq = torch.from_numpy(np.tile(iq1, (B // 8, 1, 1))).cuda()
    h = C.c_void_p()
    assert l.ddn_ted_batch_create(B, sps, 4800, 0.0, C.byref(h)) == 0
    stride = nn // sps + 64
    d_sym = torch.zeros((B, stride, 2), dtype=torch.float32, device="cuda")
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_gardner_run(h, d_iq.data_ptr(), nn, d_sym.data_ptr(), stride, d_cnt.data_ptr(), st))
    o = orc.OracleTed(sps, 4800)
    report("gardner_ted_sps10", B * nn, ms, 8 + 0.8, cpu(lambda: o.block(iq1[0]), nn), "complex samples")

    # P25p1 slicer + soft decisions (4096 channels x 4800 symbols) and the per-sample P25 matched filter
    B, ns = 4096, 4800
    sy1 = np.stack([orc.synth_c4fm_symbols(50 + c, ns) for c in range(8)])
    d_sy = torch.from_numpy(np.tile(sy1, (B // 8, 1))).cuda()
    hs = C.c_void_p()
    assert l.ddn_slicer_batch_create(B, 0, C.byref(hs)) == 0
    d_rec = torch.zeros((B, ns, 10), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_p25_slicer_run(hs, d_sy.data_ptr(), ns, d_rec.data_ptr(), st))
    report("p25_slicer_soft", B * ns, ms, 4 + 10, cpu(lambda: orc.oracle_slicer(sy1[:1]), ns), "symbols")
    nn = 48000
    x1, _, _ = orc.synth_p25_disc(5, 8, nn, frame_dibits=864)
    d_x = torch.from_numpy(np.tile(x1, (B // 8, 1))).cuda()
    d_y = torch.zeros_like(d_x)
    ms = timeit(lambda: l.ddn_p25_matched_filter_run(hs, d_x.data_ptr(), nn, d_y.data_ptr(), st))
    report("p25_matched_filter_91tap", B * nn, ms, 8, cpu(lambda: orc.oracle_p25_filter(x1[:1]), nn), "samples")

    # P25p1 receive loop: discriminator samples -> capture records (symbolizer + sync hunt + warm start + slicer)
    rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1)
    msym = l.ddn_p25_rx_max_symbols(rx.h, nn)
    d_rec2 = torch.zeros((B, msym, 10), dtype=torch.uint8, device="cuda")
    d_fl = torch.zeros((B, msym), dtype=torch.uint8, device="cuda")
    d_cn = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_p25_rx_run(rx.h, d_x.data_ptr(), nn, d_rec2.data_ptr(), d_fl.data_ptr(), d_cn.data_ptr(),
                                         msym, st))
    orx = orc.OracleP25Rx(lock_symbols=840, use_filter=1)
    report("p25_rx_loop_sps10", B * nn, ms, 4 + 1.1, cpu(lambda: orx.run(x1[0]), nn), "samples")

    # CQPSK front end: 4096 channels x 24000 samples @24 ksps (sps 5), cf32 in -> symbols
    B, sps = 4096, 5
    q1 = orc.synth_dqpsk_f32(12, 8, 4808, sps)
    nq = q1.shape[1]
    d_q = torch.from_numpy(np.tile(q1, (B // 8, 1, 1))).cuda()
    cq = ddn.CqpskBatch(B, rate=24000, block_len=4096)
    strd = l.ddn_cqpsk_max_symbols(cq.h, nq)
    d_s = torch.zeros((B, strd), dtype=torch.float32, device="cuda")
    d_c = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_cqpsk_run(cq.h, d_q.data_ptr(), nq, d_s.data_ptr(), strd, d_c.data_ptr(), st))
    ocq = orc.OracleCqpskFe(rate=24000)
    report("cqpsk_front_end_sps5", B * nq, ms, 8 + 0.8, cpu(lambda: ocq.run(q1[0], 4096), nq), "complex samples")

    # rational resampler 5/4 (e.g. 38.4 -> 48 ksps): 4096 channels x 48000 discriminator samples
    B, nn = 4096, 48000
    xr = (rng.normal(0, 9000, (8, nn))).astype(np.float32)
    d_x = torch.from_numpy(np.tile(xr, (B // 8, 1))).cuda()
    h = C.c_void_p()
    assert l.ddn_resampler_create(B, 5, 4, C.byref(h)) == 0
    no = nn * 5 // 4 + 4
    d_y = torch.zeros((B, no), dtype=torch.float32, device="cuda")
    ms = timeit(lambda: l.ddn_resampler_run(h, d_x.data_ptr(), nn, d_y.data_ptr(), no, st))
    ors = orc.OracleResampler(5, 4)
    report("resampler_5_4", B * nn, ms, 4 + 5, cpu(lambda: ors.run(xr[0]), nn), "input samples")
    l.ddn_resampler_destroy(h)

    # front end with the optional IQ conditioning on (unfused route): 4096 channels x 48000 cu8 samples
    iq8 = orc.synth_c4fm_cu8(0, 8, nn)
    d_i = torch.from_numpy(np.tile(iq8, (B // 8, 1, 1))).cuda()
    d_o = torch.zeros((B, nn), dtype=torch.float32, device="cuda")
    fb = ddn.Batch(B, block_len=8192)
    fb.set_iq_conditioning(1, 11, 1, 0.0, 0.0)
    ms = timeit(lambda: fb.run_device(d_i.data_ptr(), nn, d_o.data_ptr(), st))
    ofe = orc.OracleFrontEnd().set_iq_options(1, 11, 1, 0.0, 0.0)
    report("front_end_iq_conditioned", B * nn, ms, 2 + 4, cpu(lambda: ofe.run_cu8(iq8[0], 8192), nn), "complex samples")

    # BASELINE configs[3] mix: the same fused front end with the DMR (12.5 kHz) and NXDN48 (6.25 kHz) channel filters
    for tag, prof in (("front_end_dmr_12k5", ddn.LPF_12K5), ("front_end_nxdn48_6k25", ddn.LPF_6K25)):
        fb2 = ddn.Batch(B, lpf_profile=prof, block_len=8192)
        ms = timeit(lambda: fb2.run_device(d_i.data_ptr(), nn, d_o.data_ptr(), st))
        of2 = orc.OracleFrontEnd(profile=prof)
        report(tag, B * nn, ms, 2 + 4, cpu(lambda: of2.run_cu8(iq8[0], 8192), nn), "complex samples")

    # wide capture: two half-band passes (192 ksps -> 48 ksps) in front of the same channel filter; 1024 channels
    Bw = 1024
    d_w = torch.from_numpy(np.tile(iq8, (Bw // 8, 4, 1))).cuda()          # [1024][192000][2] cu8
    d_ow = torch.zeros((Bw, nn), dtype=torch.float32, device="cuda")
    fw = ddn.Batch(Bw, block_len=32768)
    fw.set_decimation(2)
    ms = timeit(lambda: fw.run_device(d_w.data_ptr(), 4 * nn, d_ow.data_ptr(), st))
    ofw = orc.OracleFrontEnd(downsample_passes=2)
    wide1 = np.tile(iq8[0], (4, 1))
    report("front_end_halfband_x4", Bw * 4 * nn, ms, 2 + 1, cpu(lambda: ofw.run_cu8(wide1, 32768), 4 * nn),
           "input complex samples")

    # IMBE de-interleave: 4096 channels x 9 voice frames out of an LDU's records
    nf = 4096 * 9
    recs = torch.from_numpy(rng.integers(0, 256, (4096 * 900, 10), dtype=np.uint8)).cuda()
    first = torch.from_numpy((np.arange(nf, dtype=np.int64) // 9) * 900 + (np.arange(nf) % 9) * 76 + 60).cuda()
    sc = torch.from_numpy(rng.integers(0, 36, nf).astype(np.int32)).cuda()
    o1 = torch.zeros((nf, 184), dtype=torch.uint8, device="cuda")
    o2 = torch.zeros((nf, 368), dtype=torch.uint8, device="cuda")
    o3 = torch.zeros(nf, dtype=torch.uint8, device="cuda")
    o4 = torch.zeros(nf, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_p25p1_imbe_deinterleave_batch(recs.data_ptr(), 4096 * 900, first.data_ptr(), sc.data_ptr(),
                                                            nf, o1.data_ptr(), o2.data_ptr(), o3.data_ptr(),
                                                            o4.data_ptr(), st))
    d80 = rng.integers(0, 4, 80).astype(np.uint8)
    z80 = np.zeros(80, np.int16)

    def cpu_imbe():
        for _ in range(2000):
            orc.oracle_imbe_deinterleave(d80, z80, z80, 7)
    report("imbe_deinterleave", nf, ms, 72 * 10 + 184 * 3, cpu(cpu_imbe, 2000), "voice frames")

    # P25 Phase 2 RS(63,35) FACCH sections with erasures: 9 punctured parity symbols erased + 4 errors + 2 more erasures
    import rs28
    nrs = 32768
    cases = [rs28.make_case(rng, 1, 4, 2, 1) for _ in range(512)]
    pl = np.tile(np.stack([c[0] for c in cases]).astype(np.uint8), (nrs // 512, 1))
    pa = np.tile(np.stack([c[1] for c in cases]).astype(np.uint8), (nrs // 512, 1))
    er = np.zeros((512, 28), np.int8)
    ne = np.zeros(512, np.uint8)
    for i, c in enumerate(cases):
        er[i, :c[2].size], ne[i] = c[2], c[2].size
    d_pl0, d_pa = torch.from_numpy(pl).cuda(), torch.from_numpy(pa).cuda()
    d_er, d_ne = torch.from_numpy(np.tile(er, (nrs // 512, 1))).cuda(), torch.from_numpy(np.tile(ne, nrs // 512)).cuda()
    d_pl, d_st = d_pl0.clone(), torch.zeros(nrs, dtype=torch.int32, device="cuda")

    def run_rs28():
        d_pl.copy_(d_pl0)
        l.ddn_fec_rs28_batch(1, d_pl.data_ptr(), d_pa.data_ptr(), d_er.data_ptr(), d_ne.data_ptr(), nrs, d_st.data_ptr(), st)
    ms = timeit(run_rs28)
    assert int((d_st > 0).sum()) == nrs

    def cpu_rs28():
        for c in cases:
            rs28.oracle_rs28(1, c[0], c[1], c[2])
    report("p25p2_rs28_facch", nrs, ms, 45 * 6 + 28 + 4, cpu(cpu_rs28, 512), "sections")

    # short-integer voice path (processAudio -> hpf_dL) and the I-ISCH lookup
    from test_oracle_audio import oracle_s16, s16_state, voice_like
    S, F = 4096, 50
    x1 = voice_like(rng, 64, F)
    d_x = torch.from_numpy(np.tile(x1, (S // 64, 1, 1))).cuda()
    d_o = torch.zeros(S, F, 160, dtype=torch.int16, device="cuda")
    d_s0 = torch.from_numpy(s16_state(S)).cuda()
    d_s, d_g = d_s0.clone(), torch.zeros(S, device="cuda")

    def run_s16():
        l.ddn_audio_s16_batch(d_x.data_ptr(), S, F, 0.0, 1, 0, d_o.data_ptr(), d_s.data_ptr(), d_g.data_ptr(), st)
    ms = timeit(run_s16)
    report("audio_s16_agc_hpf", S * F, ms, 160 * 6, cpu(lambda: oracle_s16(x1, s16_state(64), 0.0, 1, 0), 64 * F), "voice frames")

    from test_oracle_isch import oracle_hard, words
    ws = words(rng, 4096)[:4096]
    nw = 1 << 20
    d_w = torch.from_numpy(np.tile(np.array(ws, np.uint64).view(np.int64), nw // 4096)).cuda()
    d_v = torch.zeros(nw, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_fec_isch_lookup_batch(d_w.data_ptr(), None, nw, d_v.data_ptr(), st))
    report("p25p2_isch_lookup", nw, ms, 12, cpu(lambda: [oracle_hard(w) for w in ws], 4096), "words")

    # P25 Phase 2 burst layer: a timeslot's DUID + I-ISCH, its FACCH decode (gather, fixed erasures, ranked retries), the ESS section
    from test_oracle_p25p2_xcch import oracle_duid, oracle_ess, oracle_xcch
    nb = 65536
    bursts = [rs28.make_xcch_burst(rng, 0, int(rng.integers(0, 9)), 3, 2) for _ in range(512)]
    d_bits = torch.from_numpy(np.tile(np.stack([b[0] for b in bursts]), (nb // 512, 1))).cuda()
    d_llr = torch.from_numpy(np.tile(np.stack([b[1] for b in bursts]), (nb // 512, 1))).cuda()
    d_pay, d_ec, d_used = torch.zeros(nb, 156, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.int32, device="cuda"), torch.zeros(nb, dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_p25p2_xcch_batch(0, d_bits.data_ptr(), d_llr.data_ptr(), nb, 64, d_pay.data_ptr(), d_ec.data_ptr(), d_used.data_ptr(), st))
    report("p25p2_facch_burst", nb, ms, 360 * 3 + 156 + 5, cpu(lambda: [oracle_xcch(0, b[0], b[1]) for b in bursts], 512), "bursts")
    d_du, d_is = torch.zeros(nb, dtype=torch.int32, device="cuda"), torch.zeros(nb, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: l.ddn_p25p2_burst_fields_batch(d_bits.data_ptr(), d_llr.data_ptr(), nb, 64, d_du.data_ptr(), d_is.data_ptr(), st))
    report("p25p2_duid_isch", nb, ms, 48 * 3 + 8, cpu(lambda: [oracle_duid(0x17, np.minimum(np.abs(b[1][[0, 1, 74, 75, 244, 245, 318, 319]].astype(np.int32)), 255).astype(np.uint8)) for b in bursts], 512), "timeslots")
    ess = [rs28.make_ess_case(rng, int(rng.integers(0, 18)), 4, 2) for _ in range(512)]
    t = lambda k, dt: torch.from_numpy(np.tile(np.stack([e[k] for e in ess]).astype(dt), (nb // 512, 1))).cuda()
    e_pl, e_pll, e_pa, e_pal = t(0, np.uint8), t(1, np.int16), t(2, np.uint8), t(3, np.int16)
    e_out = torch.zeros(nb, 96, dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_p25p2_ess_batch(e_pl.data_ptr(), e_pll.data_ptr(), e_pa.data_ptr(), e_pal.data_ptr(), nb, 64, e_out.data_ptr(), d_ec.data_ptr(), d_used.data_ptr(), st))
    report("p25p2_ess", nb, ms, 264 * 3 + 96 + 5, cpu(lambda: [oracle_ess(*e[:4]) for e in ess], 512), "sections")

    # P25 Phase 2 above the bursts: processP2() on the groups behind each sync (ISCH -> offset, de-scrambling, DUID dispatch, burst
    # decodes, ESS), 4096 channels x 6 groups = one second of a TDMA channel each; and the dibit-level sync cut in front of it
    import p2seq
    Cn, G = 4096, 6
    base = [p2seq.make_stream(rng, G, 0xBEE00 + k, 0x164, 0x161, start_sf=int(rng.integers(0, 12)), noise=0.002) for k in range(16)]
    gb = np.stack([base[c % 16][0] for c in range(Cn)])
    gl = np.stack([base[c % 16][1] for c in range(Cn)])
    obj = ddn.P25P2Groups([((0xBEE00 + c % 16) << 24) | (0x164 << 12) | 0x161 for c in range(Cn)])
    tb, tl = torch.from_numpy(gb).cuda(), torch.from_numpy(gl).cuda()
    nr = Cn * G * 4
    o_info, o_pay = torch.zeros((nr, 8), dtype=torch.int32, device="cuda"), torch.zeros((nr, 180), dtype=torch.uint8, device="cuda")
    o_fr, o_rel = torch.zeros((nr, 384), dtype=torch.uint8, device="cuda"), torch.zeros((nr, 384), dtype=torch.uint8, device="cuda")
    o_ess = torch.zeros((nr, 96), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: l.ddn_p25p2_groups_batch(tb.data_ptr(), tl.data_ptr(), Cn, G, None, obj.seed.data_ptr(), obj.state.data_ptr(), 64,
                                                 o_info.data_ptr(), o_pay.data_ptr(), o_fr.data_ptr(), o_rel.data_ptr(), o_ess.data_ptr(), st))
    def cpu_groups():
        stt = p2seq.new_state()
        p2seq.run_groups(base[0][0], base[0][1], 0xBEE00, 0x164, 0x161, stt)
    report("p25p2_groups", nr, ms, 360 * 3 + 32 + 180, cpu(cpu_groups, G * 4), "timeslots")
    nd = 6000
    dib = rng.integers(0, 4, (Cn, nd)).astype(np.uint8)
    for c in range(Cn):
        for k in range(int(rng.integers(0, 720)), nd - 20, 720):
            dib[c, k:k + 20] = p2seq.SYNC20
    llr2 = rng.integers(-300, 300, (Cn, nd, 2)).astype(np.int16)
    td, tl2 = torch.from_numpy(dib).cuda(), torch.from_numpy(llr2).cuda()
    ng, gp, co = (torch.zeros(Cn * k, dtype=torch.int32, device="cuda") for k in (1, 8, 1))
    cb, cl = torch.zeros((Cn, 8, 1400), dtype=torch.uint8, device="cuda"), torch.zeros((Cn, 8, 1400), dtype=torch.int16, device="cuda")
    ms = timeit(lambda: l.ddn_p25p2_sync_cut_batch(td.data_ptr(), tl2.data_ptr(), Cn, nd, nd, None, 8, ng.data_ptr(), gp.data_ptr(), co.data_ptr(),
                                                   cb.data_ptr(), cl.data_ptr(), st))
    report("p25p2_sync_cut", Cn * nd, ms, 5 + 3, cpu(lambda: p2seq.sync_cut(dib[0], llr2[0]), nd), "dibits")


if __name__ == "__main__":
    main()
