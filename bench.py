#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on its end-to-end configuration.

    python bench.py --gpus N --steps K --warmup W
    N > 1: one rank per GPU over RCCL.  Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment) it is one of the ranks; started from a plain shell it launches the N ranks itself (torch.distributed.run on
    127.0.0.1, a free port) and passes rank 0's JSON line through.  Fewer than N visible devices: one line on stderr, exit 2.

metric  "I/Q Msamples/s end-to-end (demod->FEC->MBE) per GPU; % HBM roofline"
step    BASELINE configs[2] with the vocoder on: B = 4096 P25 Phase 1 channels per GPU x n = 48000 complex cu8 samples (1 s
        of air time each), resident in HBM, through the whole chain on the device - ONE C call per step
        (ddn_p25_chain_run_pipelined, include/ddn_chain.h):
          widen + 135-tap channel LPF + FSK discriminator -> P25 matched filter -> symbolizer + frame sync + slicer with the
          reference's per-DUID handlers deciding every in-frame length inside the loop (NID BCH + Chase, TSDU last-block flag,
          PDU header; no per-channel lock length is configured) -> bit-exact 10-byte dibit records -> framer -> NID BCH(63,16) ->
          TSDU blocks 0..2: 1/2-rate list-8 trellis + CRC16 selection / LDU1-LDU2: 24 x Hamming(10,6,3) + RS(24,12,13) /
          RS(24,16,9) + low speed data (16,8) / HDU: 36 x Golay(24,6) + RS(36,20,17) / TDULC: 12 x Golay(24,12) + RS(24,12,13) ->
          9 IMBE frames per LDU: de-interleave -> Golay / Hamming / PN frame decode -> parameter unpack -> enhancement ->
          synthesis to f32 PCM.
        Traffic: even channels carry voice calls (HDU, LDU1 / LDU2 ..., TDULC, TDU), odd channels a control channel (TSDUs of one
        to three blocks); 64 distinct channels of each kind tiled to B.  Carried state streams from step to step; frames that
        cross a call boundary are decoded whole in the next call (the chain object's carry).
Weak scaling: every rank owns its own B channels (independent streams, no data-path collective).

Rank 0 prints ONE JSON line.  `roofline` describes the chain's dominant kernel (k_p25_rxw, timed with HIP events on the
launch stream inside the C-ABI) against the chain's algorithmic bytes (2 B cu8 in + 10 B per symbol record
out = 3.0 B/sample, + 640 B per synthesized voice frame).  `cpu_baseline` is the same chain on the host, stage by stage with
its own `kind` (compiled reference where oracle/_ref holds the stage, the C restatement elsewhere), 1 core and all cores.
`front_end_stage` keeps BASELINE configs[1] as a named sub-object: the fused FIR + discriminator kernel with its own roofline, and (round
6) its third leg - the batched Gardner kernel on the same B x n - as `gardner` / `configs1_all_three_legs`.  Further informational
sub-objects: `configs3_mixed` (the configs[3] mix through ddn_mixed_chain), `cqpsk_p2_chains` (the P25 CQPSK chain and the Phase 2 chain at
1365 and 4096 channels), `m17_ysf_chains`, `vocoder_c5`, `batch_sweep`, `pcie_inclusive`."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

B_PER_GPU = 4096
N_SAMPLES = 48000
BLOCK = 8192
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s
N_BASE = 64                  # distinct channels of each traffic kind


def rxw_traffic_from_profiles():
    """HBM bytes per launch of the dominant kernel from the newest committed counter passes (profiles/rNN_bench_pmc_FETCH_SIZE.txt +
    _WRITE_SIZE.txt, written by tools/prof_rNN.sh: separate rocprofv3 --pmc runs of this command on this shape).  rocprofv3 reports
    both in KiB; gfx950 tallies a 128-B read request as 64 B (MI355X_MICROARCH.md, HBM), so FETCH_SIZE counts double.
    -> (bytes or None, the files read)"""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_FETCH_SIZE.txt")), reverse=True):
        w = f.replace("FETCH_SIZE", "WRITE_SIZE")
        if not os.path.exists(w):
            continue
        vals = []
        for path in (f, w):
            m = [re.search(r"mean\s+([0-9.]+)", ln) for ln in open(path) if ln.startswith("k_p25_rxw")]
            vals.append(float(m[0].group(1)) if m and m[0] else None)
        if None not in vals:
            return (2.0 * vals[0] + vals[1]) * 1024.0, [os.path.relpath(f, ROOT), os.path.relpath(w, ROOT)]
    return None, []


def make_base_traffic(n):
    """-> (voice cu8 [N_BASE][n][2], control cu8 [N_BASE][n][2]); deterministic.  Whole frames only: a voice channel is one call
    (header, voice frames, terminators) inside the second, a control channel back-to-back TSDUs of one to three blocks, so the
    replayed buffer never cuts a frame at its seam."""
    import numpy as np
    import mbe
    import p25gen
    rng = np.random.default_rng(20260928)
    n_sym = n // 10
    voice = np.zeros((N_BASE, n, 2), np.uint8)
    ctrl = np.zeros((N_BASE, n, 2), np.uint8)
    for c in range(N_BASE):
        lead = 200 + 11 * c
        room = n_sym - lead // 10 - 40
        head = [p25gen.make_hdu(rng, 0x293)[0]] if c % 2 == 0 else []
        tail = [p25gen.make_tdulc(rng, 0x293)[0], p25gen.make_tdu(0x293)]
        n_ldus = (room - sum(len(p) for p in head + tail)) // p25gen.LDU
        bits = mbe.random_imbe_bits(rng, (n_ldus * 9,))
        frames = np.stack([mbe.imbe_encode(b) for b in bits])
        dib = np.concatenate(head + [p25gen.make_ldus(rng, n_ldus, 0x293, frames)[0]] + tail)
        voice[c] = p25gen.modulate_cu8(dib, n, lead=lead, seed=c)
        lead = 200 + 13 * c
        room, parts = n_sym - lead // 10 - 40, []
        while True:
            fr = p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=int(rng.integers(1, 4)))[0]
            if sum(len(p) for p in parts) + len(fr) > room:
                break
            parts.append(fr)
        ctrl[c] = p25gen.modulate_cu8(np.concatenate(parts), n, lead=lead, seed=1000 + c)
    return voice, ctrl


def channel_source(ch):
    """global channel index -> (kind, base index)"""
    return ("voice" if ch % 2 == 0 else "ctrl"), (ch // 2) % N_BASE


def usable_cores():
    """cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a container can see 256
    logical CPUs and be allowed a handful)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(args):
    """One process of the CPU baseline: run the chain on `iq` (k channels) `reps` times; returns stage seconds."""
    iq, reps = args
    import chain_stream
    T = {}
    t0 = time.perf_counter()
    for _ in range(reps):
        for c in range(iq.shape[0]):
            chain_stream.run_stream(iq[c], iq.shape[1], seed=c, timers=T, use_ref=True)
    T["wall"] = time.perf_counter() - t0
    # the handlers' NID / half-rate decodes run inside "rx" through the restatement; where the compiled reference timed the same
    # decodes ("nid", "trellis") the restatement's share is taken out of "rx", otherwise it simply stays in
    port = T.pop("rx_fec_port", 0.0)
    if "nid" in T or "trellis" in T:
        T["rx"] = max(T["rx"] - port, 0.0)
    return T


def cpu_baseline(voice, ctrl, n):
    """The same chain on the host (tests/chain_stream.py), timed INSIDE the C calls (Python glue excluded), stage by stage: the
    compiled reference (oracle/_ref: its own sources built where they lie) for the stages it holds - front end (AVX2 unit), NID
    decode, half-rate list decode (the handlers' decodes, timed outside the loop on the same inputs; the restatement's time for them
    is taken out of the loop's) - and the C restatement for the rest (the receive loop needs the reference's global decoder state
    and its I/O layer; mbelib-neo is absent).  Hamming(10,6,3) / Reed-Solomon / Golay words of the LDU / HDU / TDULC frames
    are not in the CPU figure - in the CPU's favour."""
    import numpy as np
    import multiprocessing as mp
    import orc
    iq = np.stack([voice[0], ctrl[0], voice[1], ctrl[1]])
    _cpu_worker((iq, 1))                      # warm-up (page-in, table builds)
    reps, T = 0, {}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 10.0:
        t = _cpu_worker((iq, 1))
        for k, v in t.items():
            T[k] = T.get(k, 0.0) + v
        reps += 1
    stages = {k: v for k, v in T.items() if k != "wall"}
    c_seconds = sum(stages.values())
    samples = reps * iq.shape[0] * n
    one = samples / c_seconds / 1e6
    cores = usable_cores()
    all_cores = None
    if cores > 1:
        per = max(1, int(4.0 / (T["wall"] / reps)))          # ~4 s of work per worker
        ctx = mp.get_context("fork")
        with ctx.Pool(cores) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [(iq, per)] * cores)
            wall = time.perf_counter() - t0
        # aggregate rate over the pool, again counting only time inside the C calls of the slowest worker
        slow = max(sum(v for k, v in r.items() if k != "wall") for r in res)
        all_cores = {"value": round(cores * per * iq.shape[0] * n / slow / 1e6, 3), "unit": "Msamples/s", "cores": cores,
                     "wall_s": round(wall, 2)}
    ref = orc.have_ref()
    kinds = {"front_end": "reference" if ref else "port", "rx": "port", "nid": "reference", "trellis": "reference",
             "imbe_deint": "port", "imbe_fec": "port", "mbe_synth": "port"}
    return {"value": round(one, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d channels (2 voice + 2 control) x %d samples x %d reps of the bench traffic through the whole chain on "
                      "one core, time inside the C calls only; per-stage kind below (reference = oracle/_ref, the reference's own "
                      "sources compiled; port = oracle C restatement)" % (iq.shape[0], n, reps),
            "stages": {k: {"share": round(v / c_seconds, 3), "kind": kinds.get(k, "port")} for k, v in sorted(stages.items())},
            "all_cores": all_cores}


def parity_gate(chain, d_iq_ptr, voice, ctrl, ch_first, B, n):
    """Before any timing: a sample of this rank's channels, the first two calls of a fresh stream + the flush, through the same C
    call the timed loop makes, against the whole-stream CPU oracle (tests/chain_stream.py) - dibit records, flags, the handlers'
    decisions, NIDs, TSDU blocks, voice parameter bits and PCM all bit-exact, every sync decoded exactly once."""
    import numpy as np
    import chain_stream
    pick = sorted(set([0, 1, 2, 3, B // 2, B // 2 + 1, B - 2, B - 1]))
    col = chain_stream.Collector(chain, channels=pick)
    for _ in range(2):
        chain.run_pipelined(d_iq_ptr)
        chain.wait()
        col.take()
    chain.flush()
    col.take()
    out = {"checked_channels": len(pick), "calls": "2 + flush", "bit_exact": True, "nids": 0, "tsbk_blocks": 0, "voice_frames": 0}
    for i, c in enumerate(pick):
        kind, bi = channel_source(ch_first + c)
        iq_c = voice[bi] if kind == "voice" else ctrl[bi]
        want = chain_stream.run_stream(np.concatenate([iq_c, iq_c]), n, seed=c)
        try:
            a, b, v = chain_stream.check_channel(col, i, want)
        except AssertionError as e:
            out["bit_exact"] = False
            out["first_difference"] = "channel %d: %s" % (c, str(e)[:200])
            break
        out["nids"] += a
        out["tsbk_blocks"] += b
        out["voice_frames"] += v
    return out


def m17_ysf_chains(torch, ddn, np, n):
    """The two consumers of the libM17-style K = 5 decoder as chain objects (ddn_fsk4_chain, protocols M17 and YSF; informational):
    1365 channels (a third of the headline batch, as a configs[3] group) x n cu8 samples of the reference's own captures, every channel
    a different rotation; I/Q resident -> front end -> loop -> every frame behind every sync (M17: LSF + stream frames + LICH
    reassembly; YSF: the frame information channel, the payload behind it - V/D mode 2 voice bits -> AMBE synthesis, the DCH2 data
    channel).  Known answers on the device outputs before the timed steps."""
    import ctypes as C
    from conftest import golden
    dev = torch.device("cuda")
    out = {"workload": "1365 channels x %d cu8 samples of the reference's M17 / YSF capture (rotated per channel), one C call per step" % n}
    for name, proto, cap in (("m17", ddn.FSK4_M17, "iq_m17.npz"), ("ysf", ddn.FSK4_YSF, "iq_ysf.npz")):
        B = 1365
        iq = torch.from_numpy(np.ascontiguousarray(golden(cap)["iq"], np.uint8)).to(dev)
        m = iq.shape[0]
        off = (torch.arange(B, device=dev) * 37) % (m - n)
        x = iq[off[:, None] + torch.arange(n, device=dev)[None, :]].contiguous()
        ch = ddn.Fsk4ChainC(B, n, proto, rf_mod=0, handlers=0, vocoder=1 if name == "ysf" else 0)
        ch.run(x.data_ptr())          # (the known answers are read off this first call: replaying the buffer puts a seam into the stream)
        torch.cuda.synchronize()
        r = ch.results()
        S = B * r.max_syncs
        if name == "m17":
            st = ch.fetch(r.d_m17_str_status, np.uint8, (S,))
            lls = ch.fetch(r.d_m17_lich_status, np.uint8, (S,))
            ll = ch.fetch(r.d_m17_lich_lsf30, np.uint8, (S, 30))
            good = np.flatnonzero(lls == 2)
            src = {bytes(ll[k, 6:12].tolist()) for k in good}
            # N0CALL in base 40 (m17_address_decode_csd): the six address bytes every CRC-good reassembled LSF carries
            out[name] = {"stream_frames_decoded": int((st == 2).sum()), "lich_lsf_crc_good": int(len(good)),
                         "distinct_source_addresses": len(src), "source_address_hex": sorted(v.hex() for v in src)[:2]}
        else:
            fs = ch.fetch(r.d_ysf_fich_status, np.uint8, (S,))
            f4 = ch.fetch(r.d_ysf_fich4, np.uint8, (S, 4))
            good = np.flatnonzero(fs == 1)
            bits = np.unpackbits(f4[good], axis=1)
            v = lambda a, k: (bits[:, a:a + k] * (1 << np.arange(k - 1, -1, -1))).sum(axis=1)
            out[name] = {"fich_crc_good": int(len(good)), "fich_crc_bad": int((fs == 3).sum()),
                         "all_good_fich_read_vd2_rid_repeater_cc": bool(len(good) > 0 and np.all(v(22, 2) == 2) and np.all(v(4, 2) == 1)
                                                                        and np.all(bits[:, 21] == 1) and np.all(v(0, 2) == 1))}
            info = ch.fetch(r.d_ysf_info2, np.uint8, (S, 2))
            dst = ch.fetch(r.d_ysf_dch_status2, np.uint8, (S, 2))
            nv = ch.fetch(r.d_ysf_n_voice, np.int32, (B,))
            out[name].update({"frames_with_payload_decoded": int((info[:, 0] != 0).sum()), "vd2_frames": int((info[:, 0] == 2).sum()),
                              "dch2_crc_good": int((dst[:, 0] == 1).sum()), "dch2_crc_bad": int((dst[:, 0] == 3).sum()),
                              "voice_subframes_synthesized": int(nv.sum()) * 5})
        ch.run(x.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            ch.run(x.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        out[name].update({"channels": B, "ms_per_step": round(ms, 3), "Msamples_per_s": round(B * n / ms / 1e3, 1)})
        ch.close()
    return out


def cqpsk_p2_chains(torch, ddn, np, n):
    """What round 5 built, at batch scale (informational): the P25 Phase 1 chain with modulation = CQPSK (CQPSK demodulator -> symbol-rate
    loop with the handlers inside -> the same decode) and the P25 Phase 2 chain (demodulator at 6000 symbols/s -> loop -> processP2 ->
    AMBE synthesis of both logical channels), 1365 and 4096 channels x n cu8 samples of the reference's own captures, every channel a
    different rotation, I/Q resident, one C call per step."""
    from conftest import golden
    dev = torch.device("cuda")
    out = {"workload": "channels x %d cu8 samples of the reference's P25p1 CQPSK control-channel / voice captures and its Phase 2 capture "
                       "(rotated per channel), one C call per step" % n}

    def tile(cap, B):
        iq = torch.from_numpy(np.ascontiguousarray(golden(cap)["iq"], np.uint8)).to(dev)
        m = iq.shape[0]
        reps = (n + 40 * B + m - 1) // m + 1
        iq = iq.repeat(reps, 1)
        off = (torch.arange(B, device=dev) * 37) % (iq.shape[0] - n)
        return iq[off[:, None] + torch.arange(n, device=dev)[None, :]].contiguous()

    def timed(run, reps=5):
        run()
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for B in (1365, 4096):
        row = {}
        x = torch.cat([tile("iq_p25p1_cqpsk_cc.npz", B - B // 2), tile("iq_p25p1_cqpsk_vc.npz", B // 2)])
        ch = ddn.P25ChainC(B, n, block_len=8192, modulation=1)
        ms = timed(lambda: ch.run(x.data_ptr()))
        r = ch.results()
        nid = ch.fetch(r.d_nid4, np.int32, (B * ch.F, 4))
        row["p25p1_cqpsk"] = {"ms_per_step": round(ms, 3), "Msamples_per_s": round(B * n / ms / 1e3, 1),
                              "nids_decoded_last_call": int((nid[:, 0] > 0).sum())}
        ch.close()
        del x
        x2 = tile("iq_p25p2_cc.npz", B)
        seed = (0xBEE00 << 24) | (0x001 << 12) | 0x293
        try:
            import p2capture
            seed = (p2capture.WACN << 24) | (p2capture.SYSID << 12) | p2capture.NAC
        except Exception:
            pass
        c2 = ddn.P25P2ChainC([seed] * B, n, vocoder=1)
        ms2 = timed(lambda: c2.run(x2.data_ptr()))
        r2 = c2.results()
        ng = c2.fetch(r2.d_n_groups, np.int32, (B,))
        row["p25p2"] = {"ms_per_step": round(ms2, 3), "Msamples_per_s": round(B * n / ms2 / 1e3, 1), "sync_groups_last_call": int(ng.sum())}
        c2.close()
        del x2
        out["%d_channels" % B] = row
    return out


def vocoder_c5(torch, ddn, np, steps):
    """BASELINE configs[4] / SURVEY 8d C5: 8192 voice frames as 64 talk paths x 128 frames, uniform random valid parameter
    fields (fixed seed), through frame FEC (the encoded 144 / 72-bit frames) -> parameter decode -> synthesis, IMBE 7200x4400
    and AMBE 3600x2450.  Three talk paths per codec are checked bit-exact against the CPU restatement first.  Roofline term of
    SURVEY 8d: 11 B of parameter bits in + 640 B of f32 PCM out per frame."""
    import ctypes as C
    import mbe
    l = ddn.lib()
    S, F = 64, 128
    out = {"workload": "configs[4] shape: %d talk paths x %d frames per codec, frame FEC -> parameters -> synthesis" % (S, F)}
    for name, codec, gen, enc, nb, shape in (("imbe_7200x4400", ddn.MBE_IMBE, mbe.random_imbe_bits, mbe.imbe_encode, 88, (8, 23)),
                                              ("ambe_3600x2450", ddn.MBE_AMBE, mbe.random_ambe_bits, mbe.ambe_encode, 49, (4, 24))):
        rng = np.random.default_rng(2024)
        bits = gen(rng, (S, F))
        frames = np.stack([enc(b) for b in bits.reshape(S * F, nb)]).astype(np.uint8).reshape(S * F, *shape)
        d_fr = torch.from_numpy(frames).cuda()
        d_bits = torch.zeros((S * F, nb), dtype=torch.uint8, device="cuda")
        d_res = torch.zeros((S * F, 5), dtype=torch.int32, device="cuda")
        d_pcm = torch.zeros((S, F, 160), dtype=torch.float32, device="cuda")
        d_ro = torch.zeros((S * F, 5), dtype=torch.int32, device="cuda")
        h = C.c_void_p()
        assert l.ddn_mbe_batch_create(codec, S, C.byref(h)) == 0
        st = torch.cuda.current_stream().cuda_stream

        def run():
            assert l.ddn_mbe_frame_decode_batch(codec, d_fr.data_ptr(), None, S * F, d_bits.data_ptr(), d_res.data_ptr(), st) == 0
            assert l.ddn_mbe_synth_batch(h, d_bits.data_ptr(), d_res.data_ptr(), F, d_pcm.data_ptr(), d_ro.data_ptr(), st) == 0
        run()
        torch.cuda.synchronize()
        got_bits, pcm = d_bits.cpu().numpy().reshape(S, F, nb), d_pcm.cpu().numpy()
        ok = bool(np.array_equal(got_bits, bits))
        for sidx in (0, 17, 63):
            v = mbe.OracleVocoder(codec, 1)
            want = np.zeros((F, 160), np.float32)
            rc = mbe._o().om_process_batch(codec, C.addressof(v.tab), np.ascontiguousarray(bits[sidx]).ctypes.data, None, 0, sidx, 1, F,
                                           want.ctypes.data, None, C.addressof(v.cur), C.addressof(v.prev), C.addressof(v.enh))
            ok = ok and rc == 0 and bool(np.array_equal(want.view(np.uint32), pcm[sidx].view(np.uint32)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        fps = S * F / (ms * 1e-3)
        out[name] = {"frames": S * F, "ms": round(ms, 4), "frames_per_s": round(fps, 1), "realtime_talk_paths": round(fps / 50.0, 1),
                     "algorithmic_GBps": round(fps * 651.0 / 1e9, 3), "bit_exact_vs_cpu_restatement": ok}
        l.ddn_mbe_batch_destroy(h)
    out["note"] = ("64 talk paths are 64 wavefronts of the parameter kernel (one per talk path, the frame-to-frame recurrence): the "
                   "shape is latency-bound, the headline step's 81 664 frames over 4096 talk paths take 2.5 ms")
    return out


def front_end_stage(torch, ddn, chain, d_iq, B, n, steps):
    """BASELINE configs[1] (FIR + discriminator only) as a stage figure: the fused front-end kernel alone (the chain object's own
    front-end batch; its carried state just streams on)."""
    import numpy as np
    l = ddn.lib()
    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty((B, n), dtype=torch.float32, device=d_iq.device)
    fe = chain.fe
    assert l.ddn_batch_set_timing(fe, 1) == 0
    ms = []
    t = np.zeros(3, np.float32)
    for _ in range(steps):
        assert l.ddn_front_end_run(fe, d_iq.data_ptr(), n, out.data_ptr(), st) == 0
        assert l.ddn_batch_get_timing(fe, t.ctypes.data) == 0
        ms.append(float(t[0]))
    assert l.ddn_batch_set_timing(fe, 0) == 0
    ms = sorted(ms)[:max(1, len(ms) // 2)]
    avg = sum(ms) / len(ms)
    alg = 6.0 * B * n
    out_d = {"workload": "configs[1]: widen + 135-tap channel LPF + FSK discriminator, cu8 in -> f32 out",
             "kernel": "k_front_end_fused", "launch_ms": round(avg, 4), "Msamples_per_s": round(B * n / (avg * 1e-3) / 1e6, 1),
             "roofline": {"bound": "hbm", "achieved": round(alg / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_sample": 6.0}}
    # configs[1]'s third leg, Gardner symbol timing (the reference's bench shape, tests/dsp/bench_dsp.cpp:1101-1114: op25_gardner_cc on
    # the complex baseband of the same B x n): the batched Gardner kernel on B channels of complex samples at 10 samples per symbol -
    # complex f32 in (8 B per sample), one complex symbol per ten samples out
    import ctypes as C
    import orc
    sps = 10
    iq1 = orc.synth_qpsk_f32(9, 8, n // sps, sps)
    nn = iq1.shape[1]
    d_c = torch.from_numpy(np.tile(iq1, ((B + 7) // 8, 1, 1))[:B].copy()).to(d_iq.device)
    h = C.c_void_p()
    assert l.ddn_ted_batch_create(B, sps, 4800, 0.0, C.byref(h)) == 0
    stride = nn // sps + 64
    d_sym = torch.zeros((B, stride, 2), dtype=torch.float32, device=d_iq.device)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=d_iq.device)
    for _ in range(2):
        assert l.ddn_gardner_run(h, d_c.data_ptr(), nn, d_sym.data_ptr(), stride, d_cnt.data_ptr(), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        assert l.ddn_gardner_run(h, d_c.data_ptr(), nn, d_sym.data_ptr(), stride, d_cnt.data_ptr(), st) == 0
    e1.record()
    torch.cuda.synchronize()
    g_ms = e0.elapsed_time(e1) / 6
    l.ddn_ted_batch_destroy(h)
    g_alg = 8.8 * B * nn
    out_d["gardner"] = {"workload": "Gardner + MMSE interpolator, %d channels x %d complex f32 samples at %d samples per symbol" % (B, nn, sps),
                        "kernel": "k_gardner_ring", "launch_ms": round(g_ms, 4), "Msamples_per_s": round(B * nn / (g_ms * 1e-3) / 1e6, 1),
                        "symbols_out_per_channel": int(d_cnt.float().mean().item()),
                        "roofline": {"bound": "hbm", "achieved": round(g_alg / (g_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(g_alg / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_sample": 8.8}}
    out_d["configs1_all_three_legs"] = {"ms": round(avg + g_ms, 4), "Msamples_per_s": round(B * n / ((avg + g_ms) * 1e-3) / 1e6, 1),
                                        "note": "front end and Gardner kernel one after the other on the same B x n (the fused front end "
                                                "keeps no complex baseband in HBM for the C4FM path; the CQPSK chain, where the Gardner "
                                                "loop runs in a dsd-neo decode, is timed in cqpsk_p2_chains)"}
    return out_d


def self_launch(args, argv):
    """--gpus N from a plain shell: start the N ranks (one per GPU) and hand their output through"""
    import socket
    import subprocess
    if args.dry_run_cpu:
        have = args.gpus
    else:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node - nothing run\n" % (args.gpus, have))
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run_cpu(args, rank, world, voice, ctrl):
    """TEST MODE (tests/test_bench_launch.py), never a measurement: the launch, sharding, timing and reporting logic of this file with
    gloo between the ranks and the CPU oracle as each rank's compute - what can be checked of the N > 1 path without N GPUs.  The
    line it prints says so ("dry_run_cpu": true, "data": "synthetic, CPU oracle - not a measurement")."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import chain_stream
    import ddn_shard
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n = args.samples
    desc = ddn_shard.broadcast_descriptor({"B_total": args.channels * world, "n": n, "blk": BLOCK} if rank == 0 else None)
    ch_first, B = ddn_shard.channel_range(rank, world, desc["B_total"])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_sym = n_frames = 0
    for _ in range(args.steps):
        for c in range(B):
            kind, bi = channel_source(ch_first + c)
            got = chain_stream.run_stream((voice if kind == "voice" else ctrl)[bi], n, seed=ch_first + c, vocoder=False)
            n_sym += len(got["sym"])
            n_frames += len(got["frames"])
    if world > 1:
        dist.barrier()
    dt = ddn_shard.reduce_max_seconds(time.perf_counter() - t0, torch.device("cpu"))
    counts = torch.tensor([B, n_sym, n_frames], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(counts)
    if rank == 0:
        print(json.dumps({"metric": "I/Q Msamples/s end-to-end (demod->FEC->MBE) per GPU; % HBM roofline", "dry_run_cpu": True,
                          "value": round(float(counts[0]) * n * args.steps / dt / 1e6, 4), "unit": "Msamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic, CPU oracle - not a measurement",
                          "config": {"workload": "launch / sharding self-test", "channels_per_gpu": args.channels,
                                     "channels_total": int(counts[0]), "samples_per_channel": n, "parallelism": "channel-sharded x%d" % world},
                          "work": {"symbols": int(counts[1]), "syncs": int(counts[2])}}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=B_PER_GPU)
    ap.add_argument("--samples", type=int, default=N_SAMPLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage breakdown / PCIe / front-end sub-objects")
    ap.add_argument("--no-pipeline", action="store_true", help="run every stage of a step on one stream (no overlap of the frame "
                    "FEC / vocoder stages of step k with the front end / receive loop of step k + 1)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="test mode, never a measurement: gloo between the ranks and the CPU "
                    "oracle as compute (the launch / sharding / reporting logic without GPUs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    n = args.samples
    voice, ctrl = make_base_traffic(n)
    if args.dry_run_cpu:
        raise SystemExit(dry_run_cpu(args, rank, world, voice, ctrl))

    # the CPU leg runs first: its worker pool forks, which must happen before this process holds a HIP context
    cpu = None
    # (N > 1: rank 0 alone, before the process group forms - the other ranks wait in init_process_group meanwhile, blocked, not
    # spinning; the CPU figure belongs in the same run as the GPU figure it is compared with)
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(voice, ctrl, n)

    import torch
    import torch.distributed as dist
    import ddn
    import ddn_shard

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # rank 0 owns the batch descriptor; every rank takes its block of the channel index
    desc = ddn_shard.broadcast_descriptor({"B_total": args.channels * world, "n": n, "blk": BLOCK} if rank == 0 else None)
    ch_first, B = ddn_shard.channel_range(rank, world, desc["B_total"])
    d_voice = torch.from_numpy(voice).to(dev)
    d_ctrl = torch.from_numpy(ctrl).to(dev)
    kinds = [channel_source(ch_first + c) for c in range(B)]
    idx_v = torch.tensor([bi for (k, bi) in kinds if k == "voice"], device=dev, dtype=torch.long)
    idx_c = torch.tensor([bi for (k, bi) in kinds if k == "ctrl"], device=dev, dtype=torch.long)
    is_v = torch.tensor([k == "voice" for (k, _) in kinds], device=dev)
    d_iq = torch.empty((B, n, 2), dtype=torch.uint8, device=dev)
    d_iq[is_v] = d_voice[idx_v]
    d_iq[~is_v] = d_ctrl[idx_c]
    del d_voice, d_ctrl
    torch.cuda.synchronize()
    # the whole chain as one C object (include/ddn_chain.h); the handlers inside the receive loop decide every in-frame length
    chain = ddn.P25ChainC(B, n, block_len=BLOCK)
    l = ddn.lib()
    iq_ptr = d_iq.data_ptr()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.no_pipeline:
        def step():
            chain.run(iq_ptr, None)
    else:
        def step():
            chain.run_pipelined(iq_ptr)

    parity = parity_gate(chain, iq_ptr, voice, ctrl, ch_first, B, n)
    if not parity["bit_exact"] and not os.environ.get("DDN_BENCH_NOPARITY"):
        raise SystemExit("parity gate failed: %s" % json.dumps(parity))

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the dominant kernel on its own stream, recorded inside the C-ABI for every launch of the timed loop
    # (no synchronisation between launches; read back after the closing barrier)
    l.ddn_p25_rx_set_timing(chain.rx, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()          # every stage of every step inside the timed region: torch.cuda.synchronize() covers every stream
    dt = time.perf_counter() - t0
    rx_timed = np.zeros(2, np.float32)
    rx_timed_n = C.c_int(0)
    l.ddn_p25_rx_get_timing_avg(chain.rx, rx_timed.ctypes.data, C.byref(rx_timed_n))
    l.ddn_p25_rx_set_timing(chain.rx, 0)
    dt = ddn_shard.reduce_max_seconds(dt, dev)

    # what one step decoded (the last one): counted from the chain's result arrays
    F, Fv, S = chain.F, chain.Fv, B * chain.F
    r = chain.results()
    n_sym = int(chain.fetch(r.d_new, np.int32, (B,)).sum())
    ns = chain.fetch(r.d_n_syncs, np.int32, (B,))
    used = (np.arange(F)[None, :] < ns[:, None]).reshape(S)
    nidh = chain.fetch(r.d_nid4, np.int32, (S, 4))
    ok = used & (nidh[:, 0] > 0)
    duid = nidh[:, 2]
    tcrc = chain.fetch(r.d_tsbk_crc, np.uint8, (3, S))
    tsbk = chain.fetch(r.d_tsbk, np.uint8, (3, S, 12))
    tsdu = ok & (duid == 7)
    last = (tsbk[:, :, 0] & 0x80) != 0
    in_tsdu = np.stack([tsdu, tsdu & ~last[0], tsdu & ~last[0] & ~last[1]])       # block k belongs to the TSDU
    rs = [chain.fetch(r.d_ldu_rs_status[i], np.uint8, (S,)) for i in range(2)]
    lsd_ok = chain.fetch(r.d_lsd_ok, np.uint8, (S, 2))
    hdu_rs = chain.fetch(r.d_hdu_rs_status, np.uint8, (S,))
    td_rs = chain.fetch(r.d_tdulc_rs_status, np.uint8, (S,))
    skipv = (chain.fetch(r.d_imbe_result, np.int32, (B * Fv * 9, 5))[:, 0].view(np.uint32) & 0x80000000) != 0
    res = chain.fetch(r.d_synth_result, np.int32, (B * Fv * 9, 5))
    voice_frames = int((~skipv).sum())
    nev = chain.fetch(r.d_n_events, np.int32, (B,))
    work = {"symbols": n_sym, "syncs": int(ns.sum()), "frames_with_valid_nid": int(ok.sum()),
            "handler_decisions": int(nev.sum()),
            "tsdu": int(tsdu.sum()), "tsbk_blocks": int(in_tsdu.sum()), "tsbk_blocks_crc_ok": int((in_tsdu & (tcrc != 0)).sum()),
            "ldu1": int((ok & (duid == 5)).sum()), "ldu1_rs_ok": int((ok & (duid == 5) & (rs[0] == 0)).sum()),
            "ldu2": int((ok & (duid == 10)).sum()), "ldu2_rs_ok": int((ok & (duid == 10) & (rs[1] == 0)).sum()),
            "lsd_words_ok": int(lsd_ok[ok & ((duid == 5) | (duid == 10))].sum()),
            "hdu": int((ok & (duid == 0)).sum()), "hdu_rs_ok": int((ok & (duid == 0) & (hdu_rs == 0)).sum()),
            "tdulc": int((ok & (duid == 15)).sum()), "tdulc_rs_ok": int((ok & (duid == 15) & (td_rs == 0)).sum()),
            "tdu": int((ok & (duid == 3)).sum()),
            "voice_frames_synthesized": voice_frames,
            "voice_frames_muted_or_repeated": int((((res[:, 0] & 0x18) != 0) & ~skipv).sum())}

    serial_ms = None
    if not args.no_pipeline and rank == 0 and not args.no_extras:
        for _ in range(2):
            chain.run(iq_ptr, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 10)):
            chain.run(iq_ptr, None)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / min(args.steps, 10) * 1e3

    # per-stage / per-kernel times: a separate instrumented loop outside the timed region (one stream, stage boundaries marked
    # with HIP events inside the C object)
    chain.set_timing(True)
    l.ddn_p25_rx_set_timing(chain.rx, 1)
    l.ddn_mbe_batch_set_timing(chain.mbe, 1)
    stage_ms = np.zeros(4)
    rx_ms = np.zeros(2)
    mbe_ms = np.zeros(2)
    reps = min(args.steps, 8)
    for _ in range(reps):
        chain.run(iq_ptr, None)
        stage_ms += chain.stage_ms()
        t2 = np.zeros(2, np.float32)
        l.ddn_p25_rx_get_timing(chain.rx, t2.ctypes.data)
        rx_ms += t2
        l.ddn_mbe_batch_get_timing(chain.mbe, t2.ctypes.data)
        mbe_ms += t2
    stage_ms /= reps
    rx_ms /= reps
    mbe_ms /= reps
    chain.set_timing(False)
    l.ddn_p25_rx_set_timing(chain.rx, 0)
    l.ddn_mbe_batch_set_timing(chain.mbe, 0)

    # BASELINE configs[3] (mixed protocols), sharded over all ranks: every rank runs its part, rank 0 reports
    mixed = None
    if not args.no_extras:
        mixed = configs3_mixed(torch, ddn, np, d_iq, args.channels, n, 6, rank, world, dev)

    if rank == 0:
        total_samples = float(world) * B * n * args.steps
        msps = total_samples / dt / 1e6
        # algorithmic bytes of one step: cu8 in, one 10-byte record per symbol out, 640 B per voice frame
        alg_bytes = 2.0 * B * n + 10.0 * n_sym + 640.0 * voice_frames
        dom_ms = float(rx_timed[1]) if rx_timed_n.value > 0 else float(rx_ms[1])
        traffic, traffic_src = rxw_traffic_from_profiles()
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        line = {
            "metric": "I/Q Msamples/s end-to-end (demod->FEC->MBE) per GPU; % HBM roofline",
            "value": round(msps, 3),
            "value_per_gpu": round(msps / world, 3),      # `value` is the whole-job aggregate the bench contract asks for
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[2] + vocoder: %d P25 Phase 1 channels/GPU @48 ksps x %d cu8 samples, half voice calls (HDU, "
                                   "LDU1/LDU2, TDULC, TDU) half control (TSDUs of 1-3 blocks); front end -> matched filter -> "
                                   "symbolizer/sync/slicer with the reference's per-DUID handlers inside the loop (no configured lock "
                                   "length) -> framer -> NID BCH -> TSDU blocks 0-2 list-8 trellis + CRC16 | Hamming(10,6,3) + RS + LSD | "
                                   "HDU Golay(24,6) + RS(36,20,17) | TDULC Golay(24,12) + RS(24,12,13) -> IMBE de-interleave + "
                                   "Golay/Hamming/PN decode -> MBE synthesis (PCM f32)" % (B, n),
                       "channels_per_gpu": B, "samples_per_channel": n, "block_len": BLOCK,
                       "parallelism": "channel-sharded x%d" % world,
                       "host": "one C call per step (ddn_p25_chain_run%s, include/ddn_chain.h)" % ("" if args.no_pipeline else "_pipelined"),
                       "pipelining": ("none (one stream)" if args.no_pipeline else
                                      "2 HIP streams inside the C object: frame FEC + vocoder of step k overlap front end + receive "
                                      "loop of step k+1 (double-buffered loop outputs)") + "; every stage of every step inside the "
                                     "timed region",
                       "vocoder_tables": ("synthetic = %d as ddn_mbe_batch_tables_synthetic reports it: the built-in placeholder blob "
                                          "(include/ddn_mbe.h; a real blob loads with ddn_mbe_batch_load_tables_file); mbelib-neo absent -> "
                                          "vocoder parity unpinned" % l.ddn_mbe_batch_tables_synthetic(chain.mbe))},
            "parity": parity,
            "work_per_step": work,
            "stages_ms": {"front_end": round(float(stage_ms[0]), 3), "receive_loop": round(float(stage_ms[1]), 3),
                          "framer_nid_trellis_hamming_golay_rs": round(float(stage_ms[2]), 3),
                          "imbe_deinterleave_decode_synthesis": round(float(stage_ms[3]), 3)},
            "kernels_ms": {"k_p25_matched_filter": round(float(rx_ms[0]), 4), "k_p25_rxw": round(float(rx_ms[1]), 4),
                           "k_mbe_params": round(float(mbe_ms[0]), 4), "k_mbe_synth": round(float(mbe_ms[1]), 4)},
            "roofline": {"bound": "hbm", "kernel": "k_p25_rxw", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         # HBM bytes per launch of k_p25_rxw from separate rocprofv3 --pmc passes of this command on this shape
                         # (2 x FETCH_SIZE [gfx950 tallies 128-B requests as 64 B] + WRITE_SIZE, KB -> B; profiles/r04_bench_pmc_*.txt)
                         "traffic": traffic if (B == B_PER_GPU and n == N_SAMPLES) else None, "traffic_source": traffic_src,
                         "algorithmic_bytes": alg_bytes, "bytes_per_sample": round(alg_bytes / (B * n), 3),
                         "launch_ms": round(dom_ms, 4), "launches_averaged": int(rx_timed_n.value),
                         "launch_ms_alone": round(float(rx_ms[1]), 4),   # the same kernel with nothing else on the device
                         "chain_frac": round(alg_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if serial_ms is not None:
            line["one_stream_ms_per_step"] = round(serial_ms, 4)
        if world == 1 and not args.no_extras:
            line["front_end_stage"] = front_end_stage(torch, ddn, chain, d_iq, B, n, 12)
            line["pcie_inclusive"] = pcie_inclusive(torch, ddn, chain, d_iq, B, n)
            # `value` is the bench contract's figure (inputs resident in HBM when the timed region starts); SURVEY 8(d)'s wall time with the
            # I/Q coming from pinned host memory and the results going back is reported beside it, both forms
            line["value_pcie_inclusive"] = {"compact_results_Msamples_per_s": line["pcie_inclusive"]["compact"]["Msamples_per_s"],
                                            "compact_results_ms_per_step": line["pcie_inclusive"]["compact"]["ms_per_step"],
                                            "all_results_Msamples_per_s": line["pcie_inclusive"]["Msamples_per_s"],
                                            "all_results_ms_per_step": line["pcie_inclusive"]["ms_per_step"],
                                            "resident_ms_per_step": line["ms_per_step"]}
            line["vocoder_c5"] = vocoder_c5(torch, ddn, np, 10)
            line["batch_sweep"] = batch_sweep(torch, ddn, np, d_iq, n, B, dt / args.steps * 1e3, dom_ms)
            line["m17_ysf_chains"] = m17_ysf_chains(torch, ddn, np, n)
            line["cqpsk_p2_chains"] = cqpsk_p2_chains(torch, ddn, np, n)
        if mixed is not None:
            line["configs3_mixed"] = mixed
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["speedup_vs_cpu_1core"] = round(msps / world / cpu["value"], 1)
            if cpu.get("all_cores"):
                line["speedup_vs_cpu_all_cores"] = round(msps / world / cpu["all_cores"]["value"], 1)
        print(json.dumps(line))
    chain.close()
    if world > 1:
        dist.destroy_process_group()


def batch_sweep(torch, ddn, np, d_iq, n, base_B, base_ms, base_loop_ms, steps=4):
    """The same resident step at larger batches on ONE GPU (informational; `value` stays at BASELINE's 4096 channels): the receive
    loop is a latency chain with 4 live lanes per recurrence wavefront at 4096 channels, so per-GPU throughput keeps rising with
    the batch until the wavefronts are full - this is where it saturates, and what configs[3]'s 32 768 channels cost on one device."""
    l = ddn.lib()
    rows = [{"channels": base_B, "ms_per_step": round(base_ms, 3), "k_p25_rxw_ms": round(base_loop_ms, 3),
             "Msamples_per_s": round(base_B * n / base_ms / 1e3, 1)}]
    for mult in (2, 4, 8):
        Bs = base_B * mult
        try:
            big = d_iq.repeat(mult, 1, 1).contiguous()
            ch = ddn.P25ChainC(Bs, n, block_len=BLOCK)
        except Exception as e:          # (an allocation that does not fit is reported, not hidden)
            rows.append({"channels": Bs, "error": str(e)[:120]})
            break
        for _ in range(2):
            ch.run_pipelined(big.data_ptr())
        ch.wait()
        l.ddn_p25_rx_set_timing(ch.rx, 1)
        t0 = time.perf_counter()
        for _ in range(steps):
            ch.run_pipelined(big.data_ptr())
        ch.wait()
        ms = (time.perf_counter() - t0) / steps * 1e3
        t2 = np.zeros(2, np.float32)
        k = C.c_int(0)
        l.ddn_p25_rx_get_timing_avg(ch.rx, t2.ctypes.data, C.byref(k))
        l.ddn_p25_rx_set_timing(ch.rx, 0)
        rows.append({"channels": Bs, "ms_per_step": round(ms, 3), "k_p25_rxw_ms": round(float(t2[1]), 3),
                     "Msamples_per_s": round(Bs * n / ms / 1e3, 1)})
        ch.close()
        del big
        torch.cuda.empty_cache()
    return {"workload": "the headline step (same traffic tiled) at B = %s channels on one GPU, resident, %d pipelined steps each"
                        % (", ".join(str(r["channels"]) for r in rows), steps), "rows": rows}


def configs3_mixed(torch, ddn, np, d_iq_p25, B_per_gpu, n, steps, rank, world, dev):
    """BASELINE configs[3] shape (informational; never `value`): B_per_gpu x world channels, a third each P25 Phase 1 (the headline
    traffic), DMR and NXDN48, every receive loop with the reference's handlers inside it (no configured lock length), as ONE C object
    per GPU (ddn_mixed_chain, include/ddn_chain.h).  The global channel index [P25 | DMR | NXDN48] is block-partitioned over the
    ranks (ddn_mixed_partition): a rank owns a contiguous range of each group, no data-path collective.  DMR / NXDN48 traffic = the
    committed regression captures (tests/golden), every channel a different rotation.  Parity: 3 channels per DMR / NXDN48 group
    of this rank against the CPU restatement, bit-exact records / payload / sync hand-over."""
    import ddn_shard
    import rx4
    import orc
    from conftest import golden
    total = B_per_gpu * world
    third = total // 3
    groups = (total - 2 * third, third, third)
    part = ddn.mixed_partition(*groups, rank, world)
    (fp, Bp), (fd, Bd), (fn, Bn) = part

    def tile(name, lo, hi, first, Bc):
        iq = torch.from_numpy(np.ascontiguousarray(golden(name)["iq"][lo:hi], np.uint8)).to(dev)       # [m][2]
        m = iq.shape[0]
        off = ((torch.arange(Bc, device=dev) + first) * 37) % (m - n)
        idx = off[:, None] + torch.arange(n, device=dev)[None, :]
        return (iq[idx].contiguous() if Bc else iq[:0]), off.cpu().numpy(), iq.cpu().numpy()

    d_dmr, off_d, iq_d = tile("iq_dmr_t3_ras_cc.npz", 0, 96000, fd, Bd)
    # every fourth DMR channel carries voice: the reference's dmr_voice capture (its BS voice handlers pass 11 bursts of it on to the
    # vocoder); the others the Tier III control channel
    if Bd:
        d_dv, off_v, iq_dv = tile("iq_dmr_voice.npz", 0, 96000, fd, Bd)
        is_voice_ch = ((np.arange(Bd) + fd) % 4) == 3
        d_dmr[torch.from_numpy(is_voice_ch).to(dev)] = d_dv[torch.from_numpy(is_voice_ch).to(dev)]
        del d_dv
    d_nx, off_n, iq_n = tile("iq_nxdn48.npz", 60000, 288000, fn, Bn)
    d_p25 = d_iq_p25[:Bp].contiguous() if Bp else d_iq_p25[:0]
    m = ddn.MixedChainC(Bp, Bd, Bn, n, block_len=BLOCK)
    ptr = lambda t, k: t.data_ptr() if k else None
    args = (ptr(d_p25, Bp), ptr(d_dmr, Bd), ptr(d_nx, Bn))

    # parity (first call of fresh batches = fresh CPU states)
    m.run(*args)
    m.wait()
    par_ok, checked = True, 0
    res = {}
    for which, proto, lpf, iq, offs, rf, Bc in ((1, rx4.PROTO_DMR, 2, iq_d, off_d, 2, Bd), (2, rx4.PROTO_NXDN48, 1, iq_n, off_n, 0, Bn)):
        if not Bc:
            continue
        ch = m.part(which)
        r = ch.results()
        st, T, my = r.stride_symbols, r.carry_symbols, r.max_syncs
        f = ch.fetch
        rec, fl, pay = f(r.d_records10, np.uint8, (Bc, st, 10)), f(r.d_flags, np.uint8, (Bc, st)), f(r.d_payload2, np.uint8, (Bc, st, 2))
        new, ns = f(r.d_new, np.int32, (Bc,)), f(r.d_n_sync, np.int32, (Bc,))
        spos, pre = f(r.d_sync_pos, np.int32, (Bc, my)), f(r.d_pre, np.uint8, (Bc, my, 90))
        res[which] = (ch, r, ns, my)
        for c in sorted(set([0, Bc // 2, Bc - 1])):
            x = iq[offs[c]:offs[c] + n]
            if which == 1 and is_voice_ch[c]:
                x = iq_dv[off_v[c]:off_v[c] + n]
            disc = orc.OracleFrontEnd(profile=lpf).run_cu8(np.ascontiguousarray(x), BLOCK)
            want = rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=rf, handler=1)).run(disc, max_sync=512)
            k = int(new[c])
            rr = rec[c, T:T + k]
            # the syncs decoded in this first call: the ones whose burst / frame the call's records complete (the rest wait in the carry)
            first = [j for j, p in enumerate(want["sync_pos"]) if int(p) + T < k]
            ok = (k == len(want["sym"]) and np.array_equal(rr[:, 6:10].copy().view(np.uint32).reshape(-1), want["sym"].view(np.uint32))
                  and np.array_equal(rr[:, 0].astype(np.int32), want["rec4"][:, 0]) and np.array_equal(fl[c, T:T + k], want["fl"])
                  and np.array_equal(pay[c, T:T + k], want["pay"]) and int(ns[c]) == len(first)
                  and np.array_equal(spos[c, :int(ns[c])] - T, want["sync_pos"][first]) and np.array_equal(pre[c, :int(ns[c])], want["pre"][first]))
            par_ok = par_ok and bool(ok)
            checked += 1
    cc_ok = nx_ok = None
    nx_frac = None
    if 1 in res:   # DMR known answer on the device outputs: colour code 0, CSBK, BPTC clean on every complete burst
        ch, r, ns, my = res[1]
        S = Bd * my
        valid, st_ok = ch.fetch(r.d_valid, np.uint8, (S,)), ch.fetch(r.d_dmr_slot_type_ok, np.uint8, (S,))
        stb, errs = ch.fetch(r.d_dmr_slot_type, np.uint8, (S, 20)), ch.fetch(r.d_dmr_bptc_errs, np.uint32, (S,))
        rows = np.flatnonzero(valid)
        rows = rows[(rows % my) != 0]        # a fresh stream's first burst falls in the filter's cold start
        rows = rows[~is_voice_ch[rows // my]]   # (the known answer of the control-channel capture; the voice channels carry colour code 2)
        cc_ok = bool(len(rows) > 0 and np.all(st_ok[rows] == 1) and np.all(stb[rows][:, :4] == 0) and np.mean(errs[rows] == 0) > 0.99)
    if 2 in res:   # NXDN48: LICH parity and SACCH CRC6 (soft decode or the greedy retry) on the complete frames
        ch, r, ns, my = res[2]
        S = Bn * my
        nv, nl = ch.fetch(r.d_valid, np.uint8, (S,)), ch.fetch(r.d_nxdn_lich, np.uint8, (S,))
        ns1, ns2 = ch.fetch(r.d_nxdn_sacch_ok, np.uint8, (S,)), ch.fetch(r.d_nxdn_sacch_hard_ok, np.uint8, (S,))
        nrows = np.flatnonzero(nv)
        nx_frac = [round(float(np.mean((nl[nrows] & 0x80) != 0)), 3), round(float(np.mean(ns1[nrows] != 0)), 3),
                   round(float(np.mean((ns1[nrows] | ns2[nrows]) != 0)), 3), int(len(nrows))]
        # a fresh stream's first frames fall in the filter's cold start; with the handlers in the loop a frame whose LICH fails is
        # dropped at once (nxdn_frame.c), so the complete frames are the ones whose LICH parity held
        nx_ok = bool(len(nrows) > 0 and nx_frac[0] > 0.9 and nx_frac[2] > 0.5)

    l = ddn.lib()
    for _ in range(3):
        m.run(*args)
    m.wait()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.run(*args)
    m.wait()
    dt = ddn_shard.reduce_max_seconds((time.perf_counter() - t0) / steps, dev)
    out = {"workload": "configs[3] shape: %d channels over %d GPU(s) = %d P25 Phase 1 + %d DMR (Tier III control channel capture, GFSK rules) "
                       "+ %d NXDN48 (capture), %d cu8 samples each, handlers inside every receive loop; per protocol front end -> matched "
                       "filter -> receive loop -> frame FEC (P25: the headline chain; DMR: burst gather + Golay(20,8) + BPTC(196,96), every fourth channel a voice capture whose bursts go through AMBE frame FEC + synthesis; NXDN48: "
                       "frame gather + SACCH / FACCH1 K=5 decode + CRC + greedy retry, and the voice frames the LICHs announce through "
                       "AMBE de-interleave + frame FEC + synthesis)" % (total, world, groups[0], groups[1], groups[2], n),
           "host": "one C object per GPU (ddn_mixed_chain), one C call per step; ranks take contiguous blocks of the global channel index "
                   "(ddn_mixed_partition), no data-path collective",
           "this_rank": {"p25p1": Bp, "dmr": Bd, "nxdn48": Bn},
           "ms_per_step": round(dt * 1e3, 3), "Msamples_per_s": round(total * n / dt / 1e6, 1),
           "streams": "one HIP stream per protocol group (independent channel sets)",
           "parity": {"channels_checked": checked, "bit_exact": par_ok, "dmr_colour_code_0_csbk_bptc_clean": cc_ok,
                      "nxdn_lich_parity_and_sacch_crc": nx_ok, "nxdn_fractions": nx_frac}}
    if rank == 0:
        # each group alone on its stream, and its receive-loop kernel
        alone = {}
        for which, name, a in ((0, "p25p1", (args[0], None, None)), (1, "dmr", (None, args[1], None)), (2, "nxdn48", (None, None, args[2]))):
            if not part[which][1]:
                continue
            h = l.ddn_mixed_chain_part(m.h, which)
            st = torch.cuda.current_stream().cuda_stream
            run1 = (lambda: l.ddn_p25_chain_run(h, a[0], st)) if which == 0 else (lambda: l.ddn_fsk4_chain_run(h, a[which], st))
            run1()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                assert run1() == 0
            e1.record()
            torch.cuda.synchronize()
            alone[name] = round(e0.elapsed_time(e1) / 4, 3)
        out["chain_ms_alone"] = alone
        k4 = {}
        for which, name in ((1, "dmr"), (2, "nxdn48")):
            if which in res:
                rxh = res[which][0].rx
                l.ddn_fsk4_rx_set_timing(rxh, 1)
                l.ddn_fsk4_chain_run(l.ddn_mixed_chain_part(m.h, which), args[which], None)
                t2 = np.zeros(2, np.float32)
                l.ddn_fsk4_rx_get_timing(rxh, t2.ctypes.data)
                l.ddn_fsk4_rx_set_timing(rxh, 0)
                k4[name] = round(float(t2[1]), 3)
        out["k_fsk4_rx_ms"] = k4
        out["work_per_step"] = {"dmr_syncs": int(res[1][2].sum()) if 1 in res else 0, "nxdn_syncs": int(res[2][2].sum()) if 2 in res else 0}
        if world == 1:
            # the overlapped schedule (ddn_mixed_chain_config.overlap = 1: front ends on streams of their own, two discriminator buffers per group, the
            # fsk4 loops one channel per wavefront) needs more hardware queues than HIP's default four, and more queues cost the
            # headline chain (8.07 -> 8.64 ms at six): a process of its own, same shape, same library
            import subprocess
            env = dict(os.environ, MIX_OVERLAP="1", GPU_MAX_HW_QUEUES="6")
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_mixed.py"), str(total), "10"], env=env, capture_output=True,
                                   text=True, timeout=300)
                ms = float(r.stdout.strip().splitlines()[-1].split(":")[1].split()[0])
                out["overlapped_schedule"] = {"ms_per_step": ms, "Msamples_per_s": round(total * n / ms / 1e3, 1),
                                              "config": {"ddn_mixed_chain_config.overlap": 1}, "env": {"GPU_MAX_HW_QUEUES": "6"},
                                              "note": "opt-in (tests/test_chain_mixed_gpu.py passes in both modes); separate process"}
            except Exception as ex:  # informational only
                out["overlapped_schedule"] = {"error": str(ex)[:200]}
        if 1 in res:
            r1 = res[1][0].results()
            out["work_per_step"]["dmr_voice_bursts_synthesized"] = int(res[1][0].fetch(r1.d_dmr_n_voice, np.int32, (2 * Bd,)).sum())
    torch.cuda.synchronize()
    m.close()
    return out


def pcie_inclusive(torch, ddn, chain, d_iq, B, n):
    """the same step with the raw I/Q coming from pinned host memory and the results going back to it -
    ddn_p25_chain_run_host: the H2D copy of step k + 1 and the D2H copies of step k - 1 run on two copy streams beside the kernels of
    step k (the result copies beside its receive loop).  Two result sets: everything (10-byte records + flags + counts + handler decisions + NIDs + TSDU blocks + PCM) and the
    compact one (records as {dibit | flags, reliability} pairs - `records2` - instead of the 10-byte records and the flag bytes).
    Never `value`."""
    l = ddn.lib()
    S, V, st, E = B * chain.F, B * chain.Fv * 9, chain.stride, chain.E
    full = {"records10": B * st * 10, "flags": B * st, "counts": B * 4, "events": B * E * 16, "n_events": B * 4, "nid4": S * 16,
            "tsbk": 3 * S * 12, "pcm": V * 640}
    compact = {k: v for k, v in full.items() if k not in ("records10", "flags", "pcm")}
    compact["records2"] = B * st * 2
    dense_frames = V // 3          # host-side capacity for the dense PCM: a third of the slots (the traffic fills 28 %)
    compact["pcm_dense"], compact["pcm_slot"], compact["pcm_count"] = dense_frames * 640, dense_frames * 4, 4
    iq_bytes = B * n * 2
    pinned = []

    def pin(nbytes):
        p = C.c_void_p()
        assert l.ddn_host_alloc_pinned(nbytes, C.byref(p)) == 0
        pinned.append(p)
        return p
    h_iq = [pin(iq_bytes) for _ in range(2)]
    for p in h_iq:
        assert l.ddn_device_download(p, d_iq.data_ptr(), iq_bytes) == 0

    def run(sizes, steps=12):
        outs = []
        for _ in range(3):          # three output sets used in turn (include/ddn_chain.h: call k's results are complete when call k + 3 returns)
            o = ddn.P25ChainHostOut()
            for k, nb in sizes.items():
                setattr(o, k, pin(nb).value)
            if "pcm_dense" in sizes:
                o.pcm_dense_frames = dense_frames
            outs.append(o)
        for k in range(3):
            chain.run_host(h_iq[k & 1], outs[k % 3])
        chain.wait()
        t0 = time.perf_counter()
        for k in range(steps):
            chain.run_host(h_iq[(k + 1) & 1], outs[k % 3])
        chain.wait()
        t = (time.perf_counter() - t0) / steps
        moved = iq_bytes + sum(sizes.values())
        return {"ms_per_step": round(t * 1e3, 3), "Msamples_per_s": round(B * n / t / 1e6, 1), "bytes_over_pcie": moved,
                "GB_per_s_over_pcie": round(moved / t / 1e9, 1)}
    out = run(full)
    out["d2h_route"] = "0x%x" % l.ddn_p25_chain_d2h_route(chain.h)
    out["note"] = ("pinned host I/Q in; records + flags + counts + handler decisions + NIDs + TSDU blocks + PCM out (ddn_p25_chain_run_host), "
                   "12 steps between two waits (the drain of the last results included).  H2D: hipMemcpyAsync on a copy stream (an SDMA "
                   "engine); D2H: d2h_route != 0x0 = an SDMA engine driven below HIP (hsa_amd_memory_async_copy_on_engine, the engine bit "
                   "shown) from a worker thread that waits for the call's decode - runs beside any kernel and duplex with the input copy; "
                   "0x0 = hipMemcpyAsync (shader blit kernels on this ROCm).  Three device-side buffer sets, three host output sets in turn")
    out["compact"] = run(compact)
    out["compact"]["note"] = ("the same with the records as {dibit | flags << 2, reliability} pairs (records2) instead of records10 + flags "
                              "and the synthesized PCM frames dense (pcm_dense / pcm_slot / pcm_count, host capacity a third of the slots) "
                              "instead of every slot's 640 bytes")
    for p in pinned:
        l.ddn_host_free_pinned(p)
    return out


if __name__ == "__main__":
    main()
