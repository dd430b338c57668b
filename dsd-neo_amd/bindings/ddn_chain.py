"""BASELINE configs[2] wired through the C-ABI: B P25 Phase 1 channels, cu8 I/Q in HBM -> front end -> receive loop ->
framer -> NID BCH -> per-DUID FEC (TSDU: 1/2-rate trellis + CRC16; LDU1/LDU2: 24 Hamming(10,6,3) words + Reed-Solomon,
nine IMBE voice frames: de-interleave -> Golay / Hamming / PN frame decode -> parameter unpack -> synthesis) -> PCM,
without leaving the device.

Plumbing only (ctypes calls + torch allocations) - what a host application's batch scheduler would do around
libdsdneo_hip.so; used by bench.py and the end-to-end tests.  Every stage is a library call on the caller's stream."""
import ctypes as C
import os

import ddn


class P25Chain:
    def __init__(self, torch, B, n, lock_symbols, block_len=8192, max_frames=None, max_ldu=None, vocoder=True):
        l = ddn.lib()
        self.torch, self.l, self.B, self.n = torch, l, B, n
        dev = "cuda"
        self.fe = ddn.Batch(B, block_len=block_len)
        self.rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1, channels_per_wave=int(os.environ.get("DDN_RX_CPW", "0")))
        if lock_symbols is not None:
            import numpy as np
            ls = np.ascontiguousarray(lock_symbols, np.int32)
            assert ls.shape == (B,)
            assert l.ddn_p25_rx_set_lock_symbols(self.rx.h, ls.ctypes.data) == 0
        self.ms = l.ddn_p25_rx_max_symbols(self.rx.h, n)
        self.F = max_frames or (n // 1800 + 4)
        self.Fv = max_ldu or (n // 8640 + 2)
        S, ms, F, Fv = B * self.F, self.ms, self.F, self.Fv
        self.S = S
        self.fr = C.c_void_p()
        assert l.ddn_p25p1_framer_create(B, F, C.byref(self.fr)) == 0
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        u8, i16, i32, f32, i64 = torch.uint8, torch.int16, torch.int32, torch.float32, torch.int64
        self.disc = z((B, n), f32)
        # the receive loop's outputs are double-buffered so that run_pipelined() can decode batch k on a second stream while the
        # loop of batch k + 1 writes the other set; run() stays on set 0
        self.sets = [(z((B, ms, 10), u8), z((B, ms), u8), z((B,), i32)) for _ in range(2)]
        self.rec, self.fl, self.cnt = self.sets[0]
        self.step = 0
        self._pending = None
        self.ev_produced = [torch.cuda.Event() for _ in range(2)]
        self.ev_consumed = [torch.cuda.Event() for _ in range(2)]
        self.bits, self.rel, self.par, self.prel, self.v_nid = z((S, 63), u8), z((S, 63), u8), z((S,), u8), z((S,), u8), z((S,), u8)
        self.obs, self.nid = z((S,), i32), z((S, 4), i32)
        self.llr, self.v_blk = z((S, 196), i16), z((S,), u8)
        self.tsbk, self.met, self.crc_ok = z((S, 12), u8), z((S,), i32), z((S,), u8)
        self.words = [z((S, 240), u8), z((S, 240), u8)]
        self.wrel = z((S, 240), u8)
        self.werrs = [z((S * 24,), u8), z((S * 24,), u8)]
        self.v_ldu = z((S,), u8)
        self.rs_d = [z((S, 12, 6), u8), z((S, 16, 6), u8)]
        self.rs_p = [z((S, 12, 6), u8), z((S, 8, 6), u8)]
        self.rs_st = [z((S,), u8), z((S,), u8)]
        self.vocoder = vocoder
        V = B * Fv * 9
        self.V = V
        self.first, self.sc, self.n_ldu = z((V,), i64), z((V,), i32), z((B,), i32)
        self.imbe_fr, self.imbe_soft = z((V, 8, 23), u8), z((V, 8, 23, 2), u8)
        self.imbe_fl, self.sc_out = z((V,), u8), z((V,), i32)
        self.imbe_d, self.imbe_res = z((V, 88), u8), z((V, 5), i32)
        self.res_out = z((V, 5), i32)
        self.pcm = z((B, Fv * 9, 160), f32)
        self.mbe = C.c_void_p()
        if vocoder:
            assert l.ddn_mbe_batch_create(ddn.MBE_IMBE, B, C.byref(self.mbe)) == 0
            assert l.ddn_mbe_batch_set_p25p1_tail_rule(self.mbe, 1) == 0

    def close(self):
        if self.fr:
            self.l.ddn_p25p1_framer_destroy(self.fr)
            self.fr = C.c_void_p()
        if self.mbe:
            self.l.ddn_mbe_batch_destroy(self.mbe)
            self.mbe = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages (each returns nothing; all asynchronous on `st`) ----
    def front_end(self, d_iq, st):
        self.fe.run_device(d_iq.data_ptr(), self.n, self.disc.data_ptr(), st)

    def receive(self, st, disc=None):
        l = self.l
        disc = self.disc if disc is None else disc
        assert l.ddn_p25_rx_run(self.rx.h, disc.data_ptr(), self.n, self.rec.data_ptr(), self.fl.data_ptr(),
                                self.cnt.data_ptr(), self.ms, st) == 0

    def frame_fec(self, st):
        l, ms, S = self.l, self.ms, self.S
        p = lambda t: t.data_ptr()
        assert l.ddn_p25p1_framer_index(self.fr, p(self.fl), p(self.cnt), ms, st) == 0
        assert l.ddn_p25p1_framer_gather_nid(self.fr, p(self.rec), p(self.cnt), ms, p(self.bits), p(self.rel), p(self.par),
                                             p(self.prel), p(self.v_nid), st) == 0
        assert l.ddn_p25p1_nid_decode_batch(p(self.bits), p(self.rel), p(self.obs), p(self.par), p(self.prel), 64, S,
                                            p(self.nid), st) == 0
        # TSDU: first trellis block + CRC16
        assert l.ddn_p25p1_framer_gather_trellis_block(self.fr, 0, p(self.rec), p(self.cnt), ms, p(self.llr), None,
                                                       p(self.v_blk), st) == 0
        assert l.ddn_fec_p25_12_soft_batch(p(self.llr), S, p(self.tsbk), p(self.met), st) == 0
        assert l.ddn_fec_p25_crc16_batch(p(self.tsbk), 12, S, p(self.crc_ok), st) == 0
        # LDU1 / LDU2: Hamming words -> Reed-Solomon (24,12,13) / (24,16,9)
        for i, ldu in enumerate((1, 2)):
            assert l.ddn_p25p1_framer_gather_ldu_words(self.fr, ldu, p(self.rec), p(self.cnt), ms, p(self.words[i]),
                                                       p(self.wrel), p(self.v_ldu), st) == 0
            assert l.ddn_fec_hamming_10_6_3_batch(p(self.words[i]), S * 24, p(self.werrs[i]), st) == 0
            assert l.ddn_p25p1_framer_pack_ldu_rs(self.fr, ldu, p(self.words[i]), p(self.rs_d[i]), p(self.rs_p[i]), st) == 0
            assert l.ddn_fec_p25_rs_batch(0 if ldu == 1 else 1, p(self.rs_d[i]), p(self.rs_p[i]), S, p(self.rs_st[i]), st) == 0

    def voice(self, st):
        l, ms, V = self.l, self.ms, self.V
        p = lambda t: t.data_ptr()
        assert l.ddn_p25p1_framer_voice_index(self.fr, p(self.nid), p(self.cnt), self.Fv, ms, p(self.first), p(self.sc), p(self.n_ldu), st) == 0
        assert l.ddn_p25p1_imbe_deinterleave_batch(p(self.rec), self.B * ms, p(self.first), p(self.sc), V, p(self.imbe_fr),
                                                   p(self.imbe_soft), p(self.imbe_fl), p(self.sc_out), st) == 0
        assert l.ddn_mbe_frame_decode_batch(ddn.MBE_IMBE, p(self.imbe_fr), None, V, p(self.imbe_d), p(self.imbe_res), st) == 0
        assert l.ddn_mbe_result_skip_batch(p(self.imbe_fl), V, p(self.imbe_res), st) == 0
        if self.vocoder:
            assert l.ddn_mbe_synth_batch(self.mbe, p(self.imbe_d), p(self.imbe_res), self.Fv * 9, p(self.pcm),
                                         p(self.res_out), st) == 0

    def run(self, d_iq, st=None):
        self.front_end(d_iq, st)
        self.receive(st)
        self.frame_fec(st)
        self.voice(st)

    def run_pipelined3(self, d_iq, s_fe, s_rx, s_aux):
        """One batch interval over three torch streams: the front end of this batch on s_fe, its receive loop on s_rx, its
        frame FEC + voice on s_aux.  Called back to back, the front end of batch k + 1 and the decode of batch k - 1 run beside
        the receive loop of batch k: the loop is one latency-bound wavefront per CU for its whole duration, so the other three
        SIMDs of every CU (and the loop SIMD's idle issue slots) are free for the VALU-bound front end.  The discriminator
        buffer and the loop's output set are double-buffered; every stage of every batch still runs, in order per stage, and
        carried state stays per stage."""
        k = self.step
        cur = k & 1
        if not hasattr(self, "discs"):
            self.discs = [self.disc, self.torch.zeros_like(self.disc)]
            self.ev_fe = [self.torch.cuda.Event() for _ in range(2)]
        disc = self.discs[cur]
        self.rec, self.fl, self.cnt = self.sets[cur]
        if k >= 2:
            s_fe.wait_event(self.ev_produced[cur])        # the loop of batch k - 2 has read this discriminator buffer
        self.fe.run_device(d_iq.data_ptr(), self.n, disc.data_ptr(), s_fe.cuda_stream)
        self.ev_fe[cur].record(s_fe)
        s_rx.wait_event(self.ev_fe[cur])
        if k >= 2:
            s_rx.wait_event(self.ev_consumed[cur])        # batch k - 2 has been decoded out of this output set
        self.receive(s_rx.cuda_stream, disc)
        self.ev_produced[cur].record(s_rx)
        s_aux.wait_event(self.ev_produced[cur])
        self.frame_fec(s_aux.cuda_stream)
        self.voice(s_aux.cuda_stream)
        self.ev_consumed[cur].record(s_aux)
        self.step = k + 1

    def run_pipelined_deferred(self, d_iq, s_main, s_aux):
        """run_pipelined with the decode of batch k held back until the front end of batch k + 1 has run: the front end is
        a throughput kernel that wants the whole GPU, the receive loop is one latency-bound wavefront per SIMD - so the frame
        FEC + voice of batch k go beside the matched filter and the loop of batch k + 1 instead of beside its front end.  The
        decode of the last batch is queued by flush()."""
        k = self.step
        cur = k & 1
        rec, fl, cnt = self.sets[cur]
        if k >= 2:
            s_main.wait_event(self.ev_consumed[cur])      # batch k - 2 has been decoded out of this set
        self.front_end(d_iq, s_main.cuda_stream)
        if k >= 1:
            self._decode_pending(s_main, s_aux)           # batch k - 1, after this batch's front end
        self.rec, self.fl, self.cnt = rec, fl, cnt
        self.receive(s_main.cuda_stream)
        self.ev_produced[cur].record(s_main)
        self._pending = cur
        self.step = k + 1

    def _decode_pending(self, s_main, s_aux):
        cur = self._pending
        if cur is None:
            return
        if not hasattr(self, "ev_gate"):
            self.ev_gate = self.torch.cuda.Event()
        self.ev_gate.record(s_main)                       # (the front end queued just before)
        keep = (self.rec, self.fl, self.cnt)
        self.rec, self.fl, self.cnt = self.sets[cur]
        s_aux.wait_event(self.ev_produced[cur])
        s_aux.wait_event(self.ev_gate)
        self.frame_fec(s_aux.cuda_stream)
        self.voice(s_aux.cuda_stream)
        self.ev_consumed[cur].record(s_aux)
        self.rec, self.fl, self.cnt = keep
        self._pending = None

    def flush(self, s_main, s_aux):
        """queue the decode that run_pipelined_deferred still holds back (call before reading the last batch's results)"""
        if getattr(self, "_pending", None) is not None:
            cur = self._pending
            self._decode_pending(s_main, s_aux)
            self.rec, self.fl, self.cnt = self.sets[cur]

    def run_pipelined(self, d_iq, s_main, s_aux):
        """One batch interval, software-pipelined over two torch streams: front end + receive loop of this batch on s_main,
        frame FEC + voice of this batch on s_aux - which overlaps the NEXT call's front end + receive loop (the loop is a
        per-channel latency chain on one wavefront per CU; the FEC / vocoder kernels fill the idle SIMDs beside it).  Every
        stage of every batch still runs, in order per stage; results of batch k are complete when s_aux reaches the end of
        call k.  Carried state stays per stage (front end / loop on s_main, framer / vocoder on s_aux)."""
        k = self.step
        cur = k & 1
        self.rec, self.fl, self.cnt = self.sets[cur]
        if k >= 2:
            s_main.wait_event(self.ev_consumed[cur])      # batch k - 2 has been decoded out of this set
        self.front_end(d_iq, s_main.cuda_stream)
        self.receive(s_main.cuda_stream)
        self.ev_produced[cur].record(s_main)
        s_aux.wait_event(self.ev_produced[cur])
        self.frame_fec(s_aux.cuda_stream)
        self.voice(s_aux.cuda_stream)
        self.ev_consumed[cur].record(s_aux)
        self.step = k + 1
