"""BASELINE configs[3]'s other two protocols wired through the C-ABI: B DMR or NXDN48 channels, cu8 I/Q in HBM -> front end
(12.5 kHz / 6.25 kHz channel filter) -> matched filter + receive loop (ddn_fsk4_rx_run) -> for DMR: burst gather -> slot-type
Golay(20,8) -> BPTC(196,96); for NXDN48: frame gather -> SACCH / FACCH1 decode + CRC, and the four voice frames of every frame
through AMBE de-interleave -> AMBE 3600x2450 frame FEC -> synthesis (frames the LICH does not announce as voice are muted),
without leaving the device.  Plumbing only (ctypes calls + torch allocations), like ddn_chain.py."""
import ctypes as C

import ddn


class Fsk4Chain:
    def __init__(self, torch, B, n, protocol, rf_mod=0, inverted=0, block_len=8192, lock=None, handlers=False):
        l = ddn.lib()
        self.torch, self.l, self.B, self.n, self.protocol, self.inverted = torch, l, B, n, protocol, inverted
        dmr = protocol == ddn.FSK4_DMR
        self.fe = ddn.Batch(B, symbol_rate_hz=4800 if dmr else 2400, lpf_profile=ddn.LPF_12K5 if dmr else ddn.LPF_6K25,
                            block_len=block_len)
        self.rx = ddn.Fsk4Rx(B, protocol, rf_mod=rf_mod, inverted=inverted, lock=lock, handlers=handlers)
        self.ms = l.ddn_fsk4_rx_max_symbols(self.rx.h, n)
        self.my = l.ddn_fsk4_rx_max_syncs(self.rx.h, n)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
        u8, i32, f32 = torch.uint8, torch.int32, torch.float32
        ms, my = self.ms, self.my
        self.disc = z((B, n), f32)
        self.rec, self.fl, self.pay = z((B, ms, 10), u8), z((B, ms), u8), z((B, ms, 2), u8)
        self.cnt, self.ns, self.spos = z((B,), i32), z((B,), i32), z((B, my), i32)
        self.spat, self.pre, self.prel = z((B, my), u8), z((B, my, 90), u8), z((B, my, 90), u8)
        S = B * my
        self.S = S
        if dmr:
            self.st, self.info, self.cach, self.valid = z((S, 20), u8), z((S, 196), u8), z((S, 24), u8), z((S,), u8)
            self.st_ok, self.pdu, self.r3, self.errs = z((S,), u8), z((S, 96), u8), z((S, 3), u8), z((S,), i32)
        else:
            self.lich, self.valid = z((S,), u8), z((S,), u8)
            self.ss, self.sr = z((S, 36, 2), u8), z((S, 36, 2), u8)
            self.fs, self.fr = z((S, 2, 96, 2), u8), z((S, 2, 96, 2), u8)
            self.sacch, self.sacch_ok = z((S, 4), u8), z((S,), u8)
            self.sacch_hard, self.sacch_hard_ok = z((S, 32), u8), z((S,), u8)
            self.facch, self.facch_ok = z((S * 2, 12), u8), z((S * 2,), u8)
            # voice: four AMBE frames per NXDN frame, one talk path per channel.  With the default handler length (182 symbols
            # after the sync word) two syncs are at least a 192-symbol frame apart, so a call holds n / (192 * 20) + 2 frames at
            # most - far fewer than the sync capacity the loop is sized for; the voice stage works on that many slots per channel
            self.vf = my if lock is not None else min(my, n // (192 * 20) + 2)
            V = B * self.vf
            self.v_spos, self.v_ns = z((B, self.vf), i32), z((B,), i32)
            self.ambe_fr, self.ambe_rel = z((V, 4, 4, 24), u8), z((V, 4, 4, 24), u8)
            self.ambe_d, self.ambe_res = z((V * 4, 49), u8), z((V * 4, 5), i32)
            self.voice_skip = z((V * 4,), u8)
            self.pcm, self.res_out = z((B, self.vf * 4, 160), f32), z((V * 4, 5), i32)
            self.mbe = C.c_void_p()
            assert l.ddn_mbe_batch_create(ddn.MBE_AMBE, B, C.byref(self.mbe)) == 0

    def front_end(self, d_iq, st):
        self.fe.run_device(d_iq.data_ptr(), self.n, self.disc.data_ptr(), st)

    def receive(self, st):
        p = lambda t: t.data_ptr()
        assert self.l.ddn_fsk4_rx_run(self.rx.h, p(self.disc), self.n, p(self.rec), p(self.fl), p(self.pay), p(self.cnt), self.ms,
                                      p(self.spos), p(self.spat), p(self.pre), p(self.prel), p(self.ns), self.my, st) == 0

    def burst_fec(self, st):
        l, p = self.l, (lambda t: t.data_ptr())
        if self.protocol != ddn.FSK4_DMR:
            # NXDN48: frame gather -> SACCH / FACCH1 K=5 soft decode -> CRC6 / CRC12 -> the reference's greedy retry for the SACCH
            S = self.S
            assert l.ddn_nxdn_frame_gather(p(self.rec), p(self.cnt), self.ms, p(self.spos), p(self.ns), self.B, self.my, p(self.lich),
                                           p(self.ss), p(self.sr), p(self.fs), p(self.fr), p(self.valid), st) == 0
            assert l.ddn_fec_nxdn_conv_batch(p(self.ss), p(self.sr), S, 36, 32, None, p(self.sacch), 4, st) == 0
            assert l.ddn_nxdn_crc_check_batch(p(self.sacch), 4, S, 0, p(self.sacch_ok), st) == 0
            hard_in = self.ss.view(S, 72) >> 1
            assert l.ddn_fec_trellis_decode_batch(p(hard_in), 72, S, 32, p(self.sacch_hard), 32, st) == 0
            assert l.ddn_nxdn_crc_check_batch(p(self.sacch_hard), 32, S, 2, p(self.sacch_hard_ok), st) == 0
            assert l.ddn_fec_nxdn_conv_batch(p(self.fs), p(self.fr), S * 2, 96, 92, None, p(self.facch), 12, st) == 0
            assert l.ddn_nxdn_crc_check_batch(p(self.facch), 12, S * 2, 1, p(self.facch_ok), st) == 0
            # voice (nxdn_voice(): the LICH's profile says which of the four 36-dibit frames are voice), on the first vf sync
            # slots of every channel (torch ops: the caller keeps torch's current stream == st, as bench.py does)
            vf, V = self.vf, self.B * self.vf
            self.v_spos.copy_(self.spos[:, :vf])
            self.v_ns.copy_(self.torch.clamp(self.ns, max=vf))
            assert l.ddn_nxdn_voice_gather(p(self.rec), p(self.cnt), self.ms, p(self.v_spos), p(self.v_ns), self.B, vf,
                                           p(self.ambe_fr), p(self.ambe_rel), None, st) == 0
            assert l.ddn_mbe_frame_decode_batch(ddn.MBE_AMBE, p(self.ambe_fr), p(self.ambe_rel), V * 4, p(self.ambe_d),
                                                p(self.ambe_res), st) == 0
            lich = self.lich.view(self.B, self.my)[:, :vf].reshape(V, 1)
            l7, good = lich & 0x7F, ((lich & 0x80) != 0) & (self.valid.view(self.B, self.my)[:, :vf].reshape(V, 1) != 0)
            both = (l7 == 0x36) | (l7 == 0x37) | (l7 == 0x56) | (l7 == 0x57) | (l7 == 0x46) | (l7 == 0x76) | (l7 == 0x77)
            first = (l7 == 0x34) | (l7 == 0x35) | (l7 == 0x54) | (l7 == 0x55) | (l7 == 0x75)
            last = (l7 == 0x32) | (l7 == 0x33) | (l7 == 0x52) | (l7 == 0x53) | (l7 == 0x72) | (l7 == 0x73)
            v = self.torch.arange(4, device=lich.device).view(1, 4)
            voiced = good & (both | (first & (v < 2)) | (last & (v >= 2)))
            self.voice_skip.copy_((~voiced).to(self.torch.uint8).view(-1))
            assert l.ddn_mbe_result_skip_batch(p(self.voice_skip), V * 4, p(self.ambe_res), st) == 0
            assert l.ddn_mbe_synth_batch(self.mbe, p(self.ambe_d), p(self.ambe_res), vf * 4, p(self.pcm), p(self.res_out), st) == 0
            return
        assert l.ddn_dmr_burst_gather(p(self.rec), p(self.cnt), self.ms, p(self.spos), p(self.pre), p(self.ns), self.B, self.my,
                                      self.inverted, p(self.st), p(self.info), p(self.cach), p(self.valid), st) == 0
        assert l.ddn_fec_block_code_batch(5, p(self.st), self.S, 1, None, p(self.st_ok), st) == 0        # DDN_CODE_GOLAY_20_8
        assert l.ddn_fec_bptc_196x96_batch(p(self.info), 1, self.S, p(self.pdu), p(self.r3), p(self.errs), st) == 0

    def run(self, d_iq, st=None):
        self.front_end(d_iq, st)
        self.receive(st)
        self.burst_fec(st)
