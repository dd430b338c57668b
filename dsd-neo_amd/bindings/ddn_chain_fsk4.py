"""BASELINE configs[3]'s other two protocols wired through the C-ABI: B DMR or NXDN48 channels, cu8 I/Q in HBM -> front end
(12.5 kHz / 6.25 kHz channel filter) -> matched filter + receive loop (ddn_fsk4_rx_run) -> for DMR: burst gather -> slot-type
Golay(20,8) -> BPTC(196,96), without leaving the device.  Plumbing only (ctypes calls + torch allocations), like ddn_chain.py."""
import ctypes as C

import ddn


class Fsk4Chain:
    def __init__(self, torch, B, n, protocol, rf_mod=0, inverted=0, block_len=8192, lock=None):
        l = ddn.lib()
        self.torch, self.l, self.B, self.n, self.protocol, self.inverted = torch, l, B, n, protocol, inverted
        dmr = protocol == ddn.FSK4_DMR
        self.fe = ddn.Batch(B, symbol_rate_hz=4800 if dmr else 2400, lpf_profile=ddn.LPF_12K5 if dmr else ddn.LPF_6K25,
                            block_len=block_len)
        self.rx = ddn.Fsk4Rx(B, protocol, rf_mod=rf_mod, inverted=inverted, lock=lock)
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
            return
        assert l.ddn_dmr_burst_gather(p(self.rec), p(self.cnt), self.ms, p(self.spos), p(self.pre), p(self.ns), self.B, self.my,
                                      self.inverted, p(self.st), p(self.info), p(self.cach), p(self.valid), st) == 0
        assert l.ddn_fec_block_code_batch(5, p(self.st), self.S, 1, None, p(self.st_ok), st) == 0        # DDN_CODE_GOLAY_20_8
        assert l.ddn_fec_bptc_196x96_batch(p(self.info), 1, self.S, p(self.pdu), p(self.r3), p(self.errs), st) == 0

    def run(self, d_iq, st=None):
        self.front_end(d_iq, st)
        self.receive(st)
        self.burst_fec(st)
