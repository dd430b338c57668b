"""Test-side helpers: ctypes bindings of the CPU oracle (oracle/libddn_oracle.so), of the compiled reference
(oracle/_ref/libdsdneo_ref.so, only present where /root/reference was available at build time) and the synthetic
I/Q generators shared by tests/ and bench.py.  TEST INFRASTRUCTURE — never imported by the product."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ORACLE_SO = os.path.join(ROOT, "oracle", "libddn_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libdsdneo_ref.so")
REFERENCE_ROOT = "/root/reference"

_orc = None
_ref = None


def oracle():
    global _orc
    if _orc is None:
        if not os.path.exists(ORACLE_SO):
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        o = C.CDLL(ORACLE_SO)
        o.orc_fe_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        o.orc_fe_run_cu8.restype = C.c_long
        o.orc_fe_run_cu8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        o.orc_fe_run_f32.restype = C.c_long
        o.orc_fe_run_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        o.orc_fe_run_batch_cu8.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p]
        o.orc_channel_lpf_design.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        o.orc_fir_complex_apply.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int]
        o.orc_hb_decim2_complex.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int]
        o.orc_fsk_discriminator.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        o.orc_widen_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        o.orc_mean_power.restype = C.c_float
        o.orc_mean_power.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _orc = o
    return _orc


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        r = C.CDLL(REF_SO)
        r.refh_fe_create.restype = C.c_void_p
        r.refh_fe_create.argtypes = [C.c_int] * 5 + [C.c_float, C.c_int]
        r.refh_fe_destroy.argtypes = [C.c_void_p]
        r.refh_fe_run_cu8.restype = C.c_long
        r.refh_fe_run_cu8.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        r.refh_fe_run_f32.restype = C.c_long
        r.refh_fe_run_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        r.refh_fe_get_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        r.refh_fe_get_state.argtypes = [C.c_void_p, C.c_void_p]
        r.simd_fir_get_impl_name.restype = C.c_char_p
        _ref = r
    return _ref


# ---------------------------------------------------------------------------------------------------------
# oracle front end

ORC_FE_BYTES = 16384  # >= sizeof(orc_front_end)


class OracleFrontEnd:
    def __init__(self, rate=48000, profile=4, lpf_enable=1, squelch=0.0, downsample_passes=0, fma_order=1):
        self.buf = C.create_string_buffer(ORC_FE_BYTES)
        oracle().orc_fe_init(self.buf, rate, profile, lpf_enable, squelch, downsample_passes, fma_order)

    def set_iq_options(self, dc_enable=0, dc_shift=11, bal_enable=0, bal_thr=0.0, bal_ema_a=0.0):
        o = oracle()
        o.orc_fe_set_iq_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        o.orc_fe_set_iq_options(self.buf, dc_enable, dc_shift, bal_enable, bal_thr, bal_ema_a)
        return self

    def run_cu8(self, iq_u8, block_len):
        iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8).reshape(-1)
        n = iq_u8.size // 2
        out = np.zeros(n, np.float32)
        w = oracle().orc_fe_run_cu8(self.buf, iq_u8.ctypes.data, n, block_len, out.ctypes.data)
        return out[:w]

    def run_f32(self, iq, block_len):
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        n = iq.size // 2
        out = np.zeros(n, np.float32)
        w = oracle().orc_fe_run_f32(self.buf, iq.ctypes.data, n, block_len, out.ctypes.data)
        return out[:w]

    def fsk_state(self):
        # orc_front_end layout: 6 ints/floats header, taps[144], hist_i[144], hist_q[144], hb hists, fsk ...
        raise NotImplementedError


def oracle_batch_cu8(iq_u8, block_len, rate=48000, profile=4, squelch=0.0):
    """iq_u8: [B, n, 2] uint8 -> float32 [B, n] (each channel an independent stream)."""
    iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8)
    B, n = iq_u8.shape[0], iq_u8.shape[1]
    out = np.zeros((B, n), np.float32)
    oracle().orc_fe_run_batch_cu8(B, iq_u8.ctypes.data, n, block_len, rate, profile, squelch, out.ctypes.data)
    return out


def ref_front_end_cu8(iq_u8, block_len, rate=48000, sym=4800, levels=4, profile=4, lpf=1, squelch=0.0, passes=0,
                      iq_options=None):
    iq_u8 = np.ascontiguousarray(iq_u8, dtype=np.uint8).reshape(-1)
    n = iq_u8.size // 2
    h = ref().refh_fe_create(rate, sym, levels, profile, lpf, squelch, passes)
    if iq_options:
        ref().refh_fe_set_iq_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
        ref().refh_fe_set_iq_options(h, *iq_options)
    out = np.zeros(n, np.float32)
    w = ref().refh_fe_run_cu8(h, iq_u8.ctypes.data, n, block_len, out.ctypes.data)
    taps = np.zeros(160, np.float32)
    nt = ref().refh_fe_get_taps(h, taps.ctypes.data, 160)
    st = np.zeros(7, np.float32)
    ref().refh_fe_get_state(h, st.ctypes.data)
    ref().refh_fe_destroy(h)
    return out[:w], taps[:nt].copy(), st


# ---------------------------------------------------------------------------------------------------------
# synthetic C4FM-like I/Q (SURVEY.md §8d "C2 synthetic input"; modelled on the reference bench's 4-level FSK
# generator tests/dsp/bench_dsp.cpp:183-226): per channel c an LCG x <- 1664525 x + 1013904223 seeded
# 0xC0FFEE11 + c picks one of {-3,-1,+1,+3} per symbol from (x>>30)&3, the phase advances 0.028*level rad per
# sample (sps = 10), amplitude 0.85, initial phase 0.19, additive uniform noise at -20 dBc from a
# counter-based 32-bit hash, quantised to cu8 with round-half-even of 127.5 + 127.5*x clipped to [0,255].

LEVELS = np.array([-3.0, -1.0, 1.0, 3.0])


def _hash32(x):
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return x


def synth_symbols(ch_first, n_ch, n_sym):
    x = (np.uint64(0xC0FFEE11) + np.arange(ch_first, ch_first + n_ch, dtype=np.uint64)) & np.uint64(0xFFFFFFFF)
    sym = np.empty((n_ch, n_sym), np.int8)
    for s in range(n_sym):
        x = (x * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        sym[:, s] = ((x >> np.uint64(30)) & np.uint64(3)).astype(np.int8)
    return sym


def synth_c4fm_cu8(ch_first, n_ch, n, sps=10, noise_dbc=-20.0):
    """Returns uint8 [n_ch, n, 2]."""
    n_sym = (n + sps - 1) // sps
    sym = synth_symbols(ch_first, n_ch, n_sym)
    lv = LEVELS[sym]                                  # [n_ch, n_sym]
    step = np.repeat(lv * 0.028, sps, axis=1)[:, :n]   # rad / sample
    ph = 0.19 + np.cumsum(step, axis=1)
    i = 0.85 * np.cos(ph)
    q = 0.85 * np.sin(ph)
    a = np.sqrt(3.0 * 0.5 * (0.85 ** 2) * 10.0 ** (noise_dbc / 10.0))
    idx = (np.arange(ch_first, ch_first + n_ch, dtype=np.uint64)[:, None] * np.uint64(2 * n)
           + np.arange(n, dtype=np.uint64)[None, :] * np.uint64(2))
    ni = (_hash32(idx).astype(np.float64) / 4294967296.0 * 2.0 - 1.0) * a
    nq = (_hash32(idx + np.uint64(1)).astype(np.float64) / 4294967296.0 * 2.0 - 1.0) * a
    out = np.empty((n_ch, n, 2), np.uint8)
    out[:, :, 0] = np.clip(np.rint(127.5 + 127.5 * (i + ni)), 0, 255).astype(np.uint8)
    out[:, :, 1] = np.clip(np.rint(127.5 + 127.5 * (q + nq)), 0, 255).astype(np.uint8)
    return out


def load_fixture_cu8(name):
    """Reference IQ fixture bytes -> uint8 [n, 2]; from tests/golden (committed excerpt) or /root/reference."""
    p = os.path.join(REFERENCE_ROOT, "tests", "fixtures", "iq", name)
    return np.fromfile(p, dtype=np.uint8).reshape(-1, 2)


# ---------------------------------------------------------------------------------------------------------
# Gardner timing recovery helpers

def synth_qpsk_f32(seed, n_ch, n_sym, sps, drift=1.0005, noise=0.05):
    """Band-limited random QPSK at `sps` samples/symbol with a per-channel timing offset and a small clock drift,
    amplitude 0.85 (the RMS-AGC reference level of the CQPSK chain).  -> float32 [n_ch, n, 2]."""
    rng = np.random.default_rng(seed)
    n = int(n_sym * sps / drift) - 40
    out = np.empty((n_ch, n, 2), np.float32)
    for c in range(n_ch):
        ph = (rng.integers(0, 4, n_sym) * 2 + 1) * np.pi / 4
        t = np.arange(n) * drift / sps + rng.random()
        x = np.zeros(n, complex)
        for off in range(-3, 4):
            idx = np.clip(np.floor(t).astype(int) + off, 0, n_sym - 1)
            tau = t - idx
            x += np.exp(1j * ph[idx]) * np.sinc(tau) * np.cos(np.pi * 0.35 * tau) / (1 - (0.7 * tau) ** 2 + 1e-9)
        x += noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        out[c, :, 0] = 0.85 * x.real
        out[c, :, 1] = 0.85 * x.imag
    return out


class OracleTed:
    def __init__(self, sps, symbol_rate_hz, ted_gain=0.0):
        o = oracle()
        o.orc_ted_init.argtypes = [C.c_void_p]
        o.orc_gardner_block.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self.st = C.create_string_buffer(4096)
        o.orc_ted_init(self.st)
        self.sps, self.rate, self.gain = sps, symbol_rate_hz, ted_gain

    def block(self, iq):
        iq = np.ascontiguousarray(iq, np.float32).reshape(-1)
        n = iq.size // 2
        out = np.zeros(2 * n + 8, np.float32)
        w = oracle().orc_gardner_block(self.st, self.sps, self.gain, self.rate, iq.ctypes.data, n, out.ctypes.data)
        return out[:w].reshape(-1, 2).copy()


def ref_ted_blocks(iq, sps, symbol_rate_hz, ted_gain, blocks):
    """Run the compiled reference's op25_gardner_cc over `iq` [n,2] cut into the given block lengths."""
    r = ref()
    r.refh_ted_create.restype = C.c_void_p
    r.refh_ted_create.argtypes = [C.c_int, C.c_float, C.c_int]
    r.refh_ted_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    r.refh_ted_state.argtypes = [C.c_void_p, C.c_void_p]
    r.refh_ted_destroy.argtypes = [C.c_void_p]
    h = r.refh_ted_create(sps, ted_gain, symbol_rate_hz)
    iq = np.ascontiguousarray(iq, np.float32)
    outs, pos = [], 0
    for b in blocks:
        m = min(b, iq.shape[0] - pos)
        if m <= 0:
            break
        o = np.zeros(2 * m + 8, np.float32)
        seg = np.ascontiguousarray(iq[pos:pos + m])
        w = r.refh_ted_block(h, seg.ctypes.data, m, o.ctypes.data)
        outs.append(o[:w].reshape(-1, 2).copy())
        pos += m
    st = np.zeros(8, np.float32)
    r.refh_ted_state(h, st.ctypes.data)
    r.refh_ted_destroy(h)
    return outs, st


# ---------------------------------------------------------------------------------------------------------
# slicer helpers

def synth_c4fm_symbols(seed, n, scale=1.0, noise=1500.0, drift=0.05):
    """4-level symbol stream at the discriminator's +-30000 scale with noise, slow level drift and a few outliers."""
    rng = np.random.default_rng(seed)
    lv = np.array([-3.0, -1.0, 1.0, 3.0])[rng.integers(0, 4, n)]
    s = lv * 8000.0 * scale + rng.normal(0.0, noise, n) + drift * np.arange(n)
    s[n // 3:n // 3 + 12] = 0.0
    s[n // 2] = 45000.0
    return s.astype(np.float32)


def oracle_slicer(sym, negative=0):
    """sym: float32 [B, n] -> (records int32 [B, n, 4], last thresholds [B, 5])."""
    o = oracle()
    o.orc_slicer_init.argtypes = [C.c_void_p, C.c_int]
    o.orc_slicer_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    sym = np.ascontiguousarray(sym, np.float32)
    B, n = sym.shape
    rec = np.zeros((B, n, 4), np.int32)
    thr = np.zeros((B, 5), np.float32)
    for c in range(B):
        st = C.create_string_buffer(20000)
        o.orc_slicer_init(st, negative)
        t = np.zeros((n, 5), np.float32)
        o.orc_slicer_run(st, sym[c].ctypes.data, n, rec[c].ctypes.data, t.ctypes.data)
        thr[c] = t[-1]
    return rec, thr


def oracle_p25_filter(x):
    o = oracle()
    o.orc_p25_filter_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    x = np.ascontiguousarray(x, np.float32)
    B, n = x.shape
    y = np.zeros((B, n), np.float32)
    for c in range(B):
        h = np.zeros(90, np.float32)
        o.orc_p25_filter_run(h.ctypes.data, x[c].ctypes.data, n, y[c].ctypes.data)
    return y


def unpack_records10(rec):
    """uint8 [..., 10] capture records -> (int32 [..., 4] {dibit, rel, llr0, llr1}, float32 [...] symbol)."""
    rec = np.ascontiguousarray(rec, np.uint8)
    d = rec[..., 0].astype(np.int32)
    rl = rec[..., 1].astype(np.int32)
    l0 = rec[..., 2:4].copy().view(np.int16)[..., 0].astype(np.int32)
    l1 = rec[..., 4:6].copy().view(np.int16)[..., 0].astype(np.int32)
    sym = rec[..., 6:10].copy().view(np.float32)[..., 0]
    return np.stack([d, rl, l0, l1], axis=-1), sym


# ---- fixed-protocol P25p1 receive loop (oracle/ddn_oracle_rx.c) ---------------------------------------------------
P25_FS_DIBITS = np.array([int(c) for c in "111113113311333313133333"], np.int8)
_DIBIT_LEVEL = {0: 1.0, 1: 3.0, 2: -1.0, 3: -3.0}


def synth_p25_disc(seed, n_ch, n, frame_dibits=864, sps=10, amp=7000.0, noise=400.0, negative=False):
    """Discriminator-scale C4FM stream [n_ch, n] made of back-to-back frames (24-dibit P25p1 frame sync + random
    payload), each channel with its own start offset and an idle (noise only) lead-in.  Returns (x float32,
    dibits int8 [n_ch, n_sym], first_frame_start_sample int [n_ch])."""
    rng = np.random.default_rng(seed)
    n_sym = n // sps + 2
    nfr = n_sym // frame_dibits + 2
    dib = np.empty((n_ch, nfr * frame_dibits), np.int8)
    for c in range(n_ch):
        for f in range(nfr):
            dib[c, f * frame_dibits:f * frame_dibits + 24] = P25_FS_DIBITS
            dib[c, f * frame_dibits + 24:(f + 1) * frame_dibits] = rng.integers(0, 4, frame_dibits - 24)
    lv = np.vectorize(_DIBIT_LEVEL.get)(dib).astype(np.float64) * (-1.0 if negative else 1.0)
    win = np.hanning(sps + 3)[1:-1]
    win /= win.sum()
    x = np.zeros((n_ch, n), np.float64)
    starts = rng.integers(3 * sps, 40 * sps, n_ch)
    for c in range(n_ch):
        nrz = np.repeat(lv[c], sps)
        shaped = np.convolve(nrz, win, mode="same") * amp
        m = n - starts[c]
        x[c, starts[c]:] = shaped[:m]
    x += rng.normal(0.0, noise, x.shape)
    return x.astype(np.float32), dib, starts


class HEvent(C.Structure):
    _fields_ = [("pos", C.c_int32), ("kind", C.c_int16), ("a", C.c_int16), ("b", C.c_int16), ("c", C.c_int16), ("data", C.c_int32 * 4)]


class HEvents(C.Structure):
    """orc_hevents (oracle/ddn_oracle.h): the protocol handlers' event log"""
    _fields_ = [("n", C.c_int), ("ev", HEvent * 4096)]

    def rows(self):
        return [(e.pos, e.kind, e.a, e.b, e.c) for e in self.ev[:min(self.n, 4096)]]

    def data(self):
        """the decoded payload of every event, int32 [n][4] (orc_hevent.data)"""
        import numpy as np
        return np.array([list(e.data) for e in self.ev[:min(self.n, 4096)]], np.int32).reshape(-1, 4)


HEV_P25_NID, HEV_P25_TSBK, HEV_P25_MPDU, HEV_NXDN_LICH, HEV_DMR_DATA, HEV_DMR_CC_PRINT, HEV_DMR_VOICE_BURST, HEV_DMR_VOICE_END = range(1, 9)


class OracleP25Rx:
    """lock_symbols = -1: the reference's per-DUID handlers decide the in-frame length (oracle/ddn_oracle_handlers.c);
    .events.rows() then lists what they decoded"""

    def __init__(self, out_rate=48000, sym_rate=4800, lock_symbols=840, use_filter=1):
        o = oracle()
        o.orc_p25rx_sizeof.restype = C.c_size_t
        o.orc_p25rx_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        o.orc_p25rx_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        o.orc_p25rx_run.restype = C.c_long
        o.orc_p25rx_get_thresholds.argtypes = [C.c_void_p, C.c_void_p]
        self.o = o
        self.st = C.create_string_buffer(o.orc_p25rx_sizeof())
        o.orc_p25rx_init(self.st, out_rate, sym_rate, lock_symbols, use_filter)
        self.events = HEvents()
        o.orc_p25rx_set_events.argtypes = [C.c_void_p, C.c_void_p]
        o.orc_p25rx_set_events(self.st, C.byref(self.events))

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        cap = x.size // 2 + 8
        sym = np.zeros(cap, np.float32)
        rec = np.zeros((cap, 4), np.int32)
        fl = np.zeros(cap, np.uint8)
        k = self.o.orc_p25rx_run(self.st, x.ctypes.data, x.size, sym.ctypes.data, rec.ctypes.data, fl.ctypes.data, cap)
        assert k <= cap
        return sym[:k].copy(), rec[:k].copy(), fl[:k].copy()

    def thresholds(self):
        t = np.zeros(7, np.float32)
        self.o.orc_p25rx_get_thresholds(self.st, t.ctypes.data)
        return t


# ---- symbol-rate receive loop behind the CQPSK demodulator (oracle/ddn_oracle_cqrx.c) ---------------------------------
CQ_P25P1, CQ_P25P2 = 0, 1


class OracleCqRx:
    """protocol CQ_P25P1: the per-DUID handlers decide the in-frame length (lock_symbols = -1) unless a count is given;
    CQ_P25P2: 700 dibits per sync.  run(symbols) -> (rec4 [n][4] {dibit, reliability, llr0, llr1}, flags [n])"""

    def __init__(self, protocol=CQ_P25P1, lock_symbols=-1, snr_db=-100.0):
        o = oracle()
        o.orc_cqrx_sizeof.restype = C.c_size_t
        o.orc_cqrx_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        o.orc_cqrx_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        o.orc_cqrx_run.restype = C.c_long
        o.orc_cqrx_set_events.argtypes = [C.c_void_p, C.c_void_p]
        o.orc_cqrx_get_state.argtypes = [C.c_void_p, C.c_void_p]
        self.o = o
        self.st = C.create_string_buffer(o.orc_cqrx_sizeof())
        o.orc_cqrx_init(self.st, protocol, lock_symbols, snr_db)
        self.events = HEvents()
        o.orc_cqrx_set_events(self.st, C.byref(self.events))

    def run(self, sym):
        sym = np.ascontiguousarray(sym, np.float32)
        rec = np.zeros((len(sym), 4), np.int32)
        fl = np.zeros(len(sym), np.uint8)
        self.o.orc_cqrx_run(self.st, sym.ctypes.data, len(sym), rec.ctypes.data, fl.ctypes.data)
        return rec, fl

    def state(self):
        t = np.zeros(8, np.float32)
        self.o.orc_cqrx_get_state(self.st, t.ctypes.data)
        return t


def oracle_cq_inframe(sym, map_idx=0, negative=0, snr_db=-100.0):
    """the in-frame path alone from initState(): -> (rec4 [n][4], thr5 [n][5] {centre, umid, lmid, max, min})"""
    o = oracle()
    o.orc_cq_inframe_step.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_double, C.c_void_p]
    o.orc_cqrx_sizeof.restype = C.c_size_t
    o.orc_cqrx_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    st = C.create_string_buffer(o.orc_cqrx_sizeof())
    o.orc_cqrx_init(st, CQ_P25P1, -1, snr_db)
    # orc_cqrx.sl sits behind the loop's own words: take its address from a probe of the struct layout
    off = _cqrx_slicer_offset()
    sl = C.addressof(st) + off
    sym = np.ascontiguousarray(sym, np.float32)
    rec = np.zeros((len(sym), 4), np.int32)
    thr = np.zeros((len(sym), 5), np.float32)
    f7 = (C.c_float * 7)
    for i, v in enumerate(sym):
        o.orc_cq_inframe_step(sl, float(v), map_idx, negative, snr_db, rec[i].ctypes.data)
        t = f7.from_address(sl + 4)         # orc_slicer: int negative; float center, umid, lmid, max, min, maxref, minref
        thr[i] = (t[0], t[1], t[2], t[3], t[4])
    return rec, thr


def _cqrx_slicer_offset():
    o = oracle()
    o.orc_cqrx_slicer_offset.restype = C.c_size_t
    return o.orc_cqrx_slicer_offset()


# ---- CQPSK front end (oracle/ddn_oracle_cqpsk.c) --------------------------------------------------------------------
def synth_dqpsk_f32(seed, n_ch, n_sym, sps, cfo=0.002, noise=0.03, amp=0.6):
    """pi/4-DQPSK-like stream (differential +-pi/4, +-3pi/4 steps) with a carrier offset (rad/sample), band-limited."""
    rng = np.random.default_rng(seed)
    n = n_sym * sps - 40
    out = np.empty((n_ch, n, 2), np.float32)
    steps = np.array([1, 3, -1, -3]) * np.pi / 4
    for c in range(n_ch):
        ph = np.cumsum(steps[rng.integers(0, 4, n_sym)])
        t = np.arange(n) / sps + rng.random()
        x = np.zeros(n, complex)
        for off in range(-3, 4):
            idx = np.clip(np.floor(t).astype(int) + off, 0, n_sym - 1)
            tau = t - idx
            x += np.exp(1j * ph[idx]) * np.sinc(tau) * np.cos(np.pi * 0.2 * tau) / (1 - (0.4 * tau) ** 2 + 1e-9)
        x *= np.exp(1j * (cfo * (1 + 0.3 * c) * np.arange(n) + rng.random() * 6.28))
        x += noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        out[c, :, 0] = amp * x.real
        out[c, :, 1] = amp * x.imag
    return out


def modulate_dqpsk_cu8(dibits, sps, seed=0, cfo=0.002, noise=0.02, amp=0.6, lead=40):
    """a dibit stream as pi/4-DQPSK (phase step = level * pi/4, levels +1 / +3 / -1 / -3 for dibits 0 / 1 / 2 / 3), band-limited like
    synth_dqpsk_f32, `sps` samples per symbol, a small carrier offset -> cu8 [n][2]"""
    rng = np.random.default_rng(seed)
    d = np.concatenate([rng.integers(0, 4, lead), np.asarray(dibits, np.int64) & 3, rng.integers(0, 4, 12)])
    n_sym = len(d)
    steps = np.array([1, 3, -1, -3])[d] * np.pi / 4
    ph = np.cumsum(steps)
    n = n_sym * sps
    t = np.arange(n) / sps + 0.37
    x = np.zeros(n, complex)
    for off in range(-3, 4):
        idx = np.clip(np.floor(t).astype(int) + off, 0, n_sym - 1)
        tau = t - idx
        x += np.exp(1j * ph[idx]) * np.sinc(tau) * np.cos(np.pi * 0.2 * tau) / (1 - (0.4 * tau) ** 2 + 1e-9)
    x *= np.exp(1j * (cfo * np.arange(n) + 1.1))
    x += noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    out = np.empty((n, 2), np.float64)
    out[:, 0], out[:, 1] = amp * x.real, amp * x.imag
    return np.clip(np.round(127.5 + 127.5 * out), 0, 255).astype(np.uint8)


class OracleCqpskFe:
    def __init__(self, rate=24000, sym_rate=4800, profile=5, lpf_enable=1, ted_gain=0.0):
        o = oracle()
        o.orc_cqpsk_fe_sizeof.restype = C.c_size_t
        o.orc_cqpsk_fe_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        o.orc_cqpsk_fe_run_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
        o.orc_cqpsk_fe_run_f32.restype = C.c_long
        self.o = o
        self.st = C.create_string_buffer(o.orc_cqpsk_fe_sizeof())
        o.orc_cqpsk_fe_init(self.st, rate, sym_rate, profile, lpf_enable, ted_gain)

    def run(self, iq, block_len):
        iq = np.ascontiguousarray(iq, np.float32)
        n = iq.shape[0]
        out = np.zeros(n + 16, np.float32)
        scr = np.zeros(4 * block_len + 16, np.float32)
        k = self.o.orc_cqpsk_fe_run_f32(self.st, iq.ctypes.data, n, block_len, out.ctypes.data, scr.ctypes.data)
        return out[:k].copy()


def ref_cqpsk_f32(iq, block_len, rate=24000, sym_rate=4800, profile=5, lpf=1):
    r = ref()
    r.refh_cqpsk_create.restype = C.c_void_p
    r.refh_cqpsk_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    r.refh_cqpsk_run_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_long]
    r.refh_cqpsk_run_f32.restype = C.c_long
    r.refh_cqpsk_get_state.argtypes = [C.c_void_p, C.c_void_p]
    r.refh_fe_destroy.argtypes = [C.c_void_p]
    iq = np.ascontiguousarray(iq, np.float32)
    n = iq.shape[0]
    h = r.refh_cqpsk_create(rate, sym_rate, profile, lpf)
    out = np.zeros(n + 16, np.float32)
    k = r.refh_cqpsk_run_f32(h, iq.ctypes.data, n, block_len, out.ctypes.data, len(out))
    st = np.zeros(8, np.float32)
    r.refh_cqpsk_get_state(h, st.ctypes.data)
    r.refh_fe_destroy(h)
    return out[:k].copy(), st


# ---- IMBE de-interleave (oracle/ddn_oracle_block.c; reference harness refh_imbe_deinterleave) --------------------------
def oracle_imbe_deinterleave(dibits, llr0, llr1, status_count):
    """One voice frame: dibits uint8 [>= 75], llr int16 -> (fr uint8 [8,23], soft uint8 [8,23,2], flag, status_out,
    records consumed)."""
    o = oracle()
    o.orc_p25p1_imbe_deinterleave.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
    d = np.ascontiguousarray(dibits, np.uint8)
    a = np.ascontiguousarray(llr0, np.int16)
    b = np.ascontiguousarray(llr1, np.int16)
    fr = np.zeros((8, 23), np.uint8)
    soft = np.zeros((8, 23, 2), np.uint8)
    sc, used = C.c_int(0), C.c_int(0)
    flag = o.orc_p25p1_imbe_deinterleave(d.ctypes.data, a.ctypes.data, b.ctypes.data, d.size, status_count,
                                         fr.ctypes.data, soft.ctypes.data, C.byref(sc), C.byref(used))
    return fr, soft, flag, sc.value, used.value


def ref_imbe_deinterleave(dibits, llr0, llr1, status_count):
    r = ref()
    r.refh_imbe_deinterleave.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    d = np.ascontiguousarray(dibits, np.uint8)
    a = np.ascontiguousarray(llr0, np.int16)
    b = np.ascontiguousarray(llr1, np.int16)
    fr = np.zeros((8, 23), np.uint8)
    soft = np.zeros((8, 23, 2), np.uint8)
    sc = C.c_int(0)
    used = r.refh_imbe_deinterleave(d.ctypes.data, a.ctypes.data, b.ctypes.data, status_count, fr.ctypes.data,
                                    soft.ctypes.data, C.byref(sc))
    return fr, soft, sc.value, used


# ---- rational resampler (oracle/ddn_oracle_resamp.c; reference dsd_resampler_* called directly) ------------------------
class RefResamplerState(C.Structure):  # include/dsd-neo/dsp/resampler.h:30-42
    _fields_ = [("enabled", C.c_int), ("target_hz", C.c_int), ("L", C.c_int), ("M", C.c_int), ("phase", C.c_int),
                ("taps_len", C.c_int), ("taps_per_phase", C.c_int), ("hist_head", C.c_int),
                ("taps", C.POINTER(C.c_float)), ("hist", C.POINTER(C.c_float)), ("internal_cookie", C.c_uint64)]


class RefResampler:
    def __init__(self, L, M):
        r = ref()
        r.dsd_resampler_design.argtypes = [C.c_void_p, C.c_int, C.c_int]
        r.dsd_resampler_process_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        r.dsd_resampler_reset.argtypes = [C.c_void_p]
        self.r = r
        self.st = RefResamplerState()
        assert r.dsd_resampler_design(C.byref(self.st), L, M) == 1

    def taps(self):
        return np.ctypeslib.as_array(self.st.taps, (self.st.taps_len,)).copy()

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(x.size * self.st.L // self.st.M + 4, np.float32)
        k = self.r.dsd_resampler_process_block(C.byref(self.st), x.ctypes.data, x.size, out.ctypes.data, out.size)
        assert k >= 0
        return out[:k].copy()

    def __del__(self):
        self.r.dsd_resampler_reset(C.byref(self.st))


class OracleResampler:
    def __init__(self, L, M):
        o = oracle()
        o.orc_resamp_sizeof.restype = C.c_size_t
        o.orc_resamp_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
        o.orc_resamp_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
        o.orc_resamp_run.restype = C.c_long
        o.orc_resamp_design.argtypes = [C.c_int, C.c_int, C.c_void_p]
        self.o, self.L, self.M = o, L, M
        self.st = C.create_string_buffer(o.orc_resamp_sizeof())
        assert o.orc_resamp_init(self.st, L, M) == 16 * L

    def taps(self):
        t = np.zeros(16 * self.L, np.float32)
        self.o.orc_resamp_design(self.L, self.M, t.ctypes.data)
        return t

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(x.size * self.L // self.M + 4, np.float32)
        k = self.o.orc_resamp_run(self.st, x.ctypes.data, x.size, out.ctypes.data, out.size)
        assert k >= 0
        return out[:k].copy()
