"""GPU: the timed kernels at the sizes bench.py times them, inside the suite.
 * the P25 Phase 1 chain object in handler mode (the reference's per-DUID handlers inside the loop, k_p25_rxw<8, true>) at BASELINE
   configs[2]'s full shape, 4096 channels x 48000 samples per call, two calls + the flush: every channel that replays the same source
   gives the same records, flags, decisions, decoded blocks and voice parameter bits as its first replica (no channel sees its neighbours, whatever workgroup, wave and
   lane it lands on), and a sample of channels - first, last, one voice and one control channel from the middle - equals the
   whole-stream CPU oracle record for record, decision for decision, frame for frame;
 * ddn_mixed_chain at configs[3]'s per-GPU share (1366 P25 + 1365 DMR + 1365 NXDN48 channels): bench.py's own parity block (three
   channels per DMR / NXDN48 group against the CPU restatement, the DMR colour-code / CSBK / BPTC and NXDN LICH / SACCH known answers)."""
import os
import sys

import numpy as np
import pytest

import chain_stream
import ddn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench_traffic(n, n_base):
    import bench
    keep = bench.N_BASE
    bench.N_BASE = n_base
    try:
        voice, ctrl = bench.make_base_traffic(n)
        src = [bench.channel_source(c) for c in range(4096)]
    finally:
        bench.N_BASE = keep
    return voice, ctrl, src


def test_p25_handler_chain_at_4096_x_48000(built):
    import torch
    B, n, n_base, calls = 4096, 48000, 4, 2
    voice, ctrl, src = _bench_traffic(n * calls, n_base)
    period = 2 * n_base                                    # channel c and c + period replay the same source
    dev = torch.device("cuda:0")
    dv, dc = torch.from_numpy(voice).to(dev), torch.from_numpy(ctrl).to(dev)
    d_iq = torch.empty((B, n * calls, 2), dtype=torch.uint8, device=dev)
    is_v = torch.tensor([k == "voice" for k, _ in src], device=dev)
    d_iq[is_v] = dv[torch.tensor([b for k, b in src if k == "voice"], device=dev)]
    d_iq[~is_v] = dc[torch.tensor([b for k, b in src if k == "ctrl"], device=dev)]
    del dv, dc
    ch = ddn.P25ChainC(B, n)
    assert ddn.lib().ddn_p25_rx_channels_per_wave(ch.rx) == 8 if hasattr(ddn.lib(), "ddn_p25_rx_channels_per_wave") else True
    sample = [0, 1, 2050, 2051, B - 2, B - 1]
    col = chain_stream.Collector(ch, channels=sample)
    st, F, Fv, E = ch.stride, ch.F, ch.Fv, ch.E

    def replicas_agree(ptr, dtype, row_shape, used=None, lead=1):
        """array [lead][B][row] -> every channel's row (its first used[c] entries) equals that of channel c % period"""
        a = ch.fetch(ptr, dtype, (lead, B) + tuple(row_shape))
        g = a.reshape((lead, B // period, period) + tuple(row_shape))
        if used is None:
            return bool(np.array_equal(g, np.broadcast_to(g[:, :1], g.shape)))
        u = used.reshape(B // period, period)
        if not np.array_equal(u, np.broadcast_to(u[:1], u.shape)):
            return False
        k = np.arange(row_shape[0])[None, :] < u[0][:, None]                                  # [period][row0]
        mask = k.reshape((1, 1, period, row_shape[0]) + (1,) * (len(row_shape) - 1))
        return bool(np.array_equal(np.where(mask, g, 0), np.where(mask, np.broadcast_to(g[:, :1], g.shape), 0)))

    totals = np.zeros(3, np.int64)
    for k in range(calls + 1):
        if k < calls:
            part = d_iq[:, k * n:(k + 1) * n].contiguous()
            torch.cuda.synchronize()         # (torch's copy runs on torch's stream, the chain on its own)
            ch.run_pipelined(part.data_ptr())
            ch.wait()
        else:
            ch.flush()
        r = ch.results()
        new = ch.fetch(r.d_new, np.int32, (B,))
        cnt = ch.fetch(r.d_counts, np.int32, (B,))
        nev = ch.fetch(r.d_n_events, np.int32, (B,))
        ns = ch.fetch(r.d_n_syncs, np.int32, (B,))
        assert int(ch.fetch(r.d_dropped_syncs, np.int32, (B,)).sum()) == 0
        assert replicas_agree(r.d_records10, np.uint8, (st, 10), used=cnt), ("records", k)
        assert replicas_agree(r.d_flags, np.uint8, (st,), used=cnt), ("flags", k)
        assert replicas_agree(r.d_events, np.int32, (E, 4), used=nev), ("events", k)
        assert replicas_agree(r.d_event_data, np.int32, (E, 4), used=nev), ("event data", k)
        assert replicas_agree(r.d_nid4, np.int32, (F, 4), used=ns), ("nid", k)
        assert replicas_agree(r.d_tsbk, np.uint8, (F, 12), used=ns, lead=3), ("tsbk", k)
        nldu = ch.fetch(r.d_n_ldu, np.int32, (B,))
        assert replicas_agree(r.d_imbe_bits, np.uint8, (Fv * 9, 88), used=nldu * 9), ("imbe", k)
        # (the PCM is not replica-invariant by design: the unvoiced excitation is counter-based noise keyed by the talk path's index;
        # it is compared with the oracle on the sampled channels below)
        totals += [int(new.sum()), int(ns.sum()), int(nev.sum())]
        col.take()
    assert totals[0] > 4096 * 9000 and totals[1] > 4096 * 20 and totals[2] > 4096 * 30, totals
    h_iq = {c: (voice if src[c][0] == "voice" else ctrl)[src[c][1]] for c in sample}
    tot = np.zeros(3, np.int64)
    for i, c in enumerate(sample):
        tot += chain_stream.check_channel(col, i, chain_stream.run_stream(h_iq[c], n, seed=c))
    assert tot[0] > 60 and tot[1] > 60 and tot[2] > 150, tot
    ch.close()


def test_mixed_chain_at_the_per_gpu_share_of_configs3(built):
    import torch
    import bench
    n = 48000
    voice, ctrl, src = _bench_traffic(n, 4)
    dev = torch.device("cuda:0")
    dv, dc = torch.from_numpy(voice).to(dev), torch.from_numpy(ctrl).to(dev)
    Bp = 1366
    d_iq = torch.empty((Bp, n, 2), dtype=torch.uint8, device=dev)
    is_v = torch.tensor([k == "voice" for k, _ in src[:Bp]], device=dev)
    d_iq[is_v] = dv[torch.tensor([b for k, b in src[:Bp] if k == "voice"], device=dev)]
    d_iq[~is_v] = dc[torch.tensor([b for k, b in src[:Bp] if k == "ctrl"], device=dev)]
    out = bench.configs3_mixed(torch, ddn, np, d_iq, 4096, n, 1, 0, 1, dev)
    assert out["this_rank"] == {"p25p1": 1366, "dmr": 1365, "nxdn48": 1365}
    par = out["parity"]
    assert par["bit_exact"] is True and par["channels_checked"] == 6, par
    assert par["dmr_colour_code_0_csbk_bptc_clean"] is True and par["nxdn_lich_parity_and_sacch_crc"] is True, par
    assert out["work_per_step"]["dmr_syncs"] > 1365 * 20 and out["work_per_step"]["nxdn_syncs"] > 1365 * 5, out["work_per_step"]


def test_gardner_full_c2_shape_properties(built):
    """BASELINE configs[1]'s Gardner leg at its full shape: 4096 channels x 48000 complex samples (sps 10) through ddn_gardner_run in
    ONE call.  Size-independent properties over the whole batch - every count within the loop's bounds, every symbol finite, the
    same bits after a reset (determinism), tiled channels identical to their source channel - and a sample of channels bit-exact
    against the oracle (pinned to the compiled costas.cpp)."""
    import ctypes as C
    import torch
    import ddn
    import orc
    l = ddn.lib()
    B, sps, n, base = 4096, 10, 48000, 16
    iq1 = np.ascontiguousarray(orc.synth_qpsk_f32(9, base, 4900, sps, noise=0.05)[:, :n])
    assert iq1.shape[1] == n
    d_iq = torch.from_numpy(iq1).cuda().repeat(B // base, 1, 1).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    stride = n // sps + 64
    d_cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    outs = []
    for rep in range(2):            # two fresh batch objects = the determinism check (a fresh object is a reset state)
        h = C.c_void_p()
        assert l.ddn_ted_batch_create(B, sps, 4800, 0.0, C.byref(h)) == 0
        d_sym = torch.zeros((B, stride, 2), dtype=torch.float32, device="cuda")
        assert l.ddn_gardner_run(h, d_iq.data_ptr(), n, d_sym.data_ptr(), stride, d_cnt.data_ptr(), st) == 0
        torch.cuda.synchronize()
        outs.append((d_sym, d_cnt.clone()))
        l.ddn_ted_batch_destroy(h)
    (sym, cnt), (sym2, cnt2) = outs
    assert torch.equal(cnt, cnt2) and torch.equal(sym.view(torch.int32), sym2.view(torch.int32))
    assert int(cnt.min()) >= n // sps - 60 and int(cnt.max()) <= n // sps + 60
    assert bool(torch.isfinite(sym).all())
    # channel c carries source channel c % base
    tiled = sym.view(B // base, base, stride, 2)
    assert torch.equal(tiled.view(torch.int32), tiled[:1].expand_as(tiled).contiguous().view(torch.int32))
    for c in (0, 5, base - 1, 2048 + 3, B - 1):
        want = orc.OracleTed(sps, 4800).block(iq1[c % base])
        k = int(cnt[c])
        assert k == len(want), (c, k, len(want))
        got = sym[c, :k].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32)), c
