"""Known answers the reference's own unit tests hold for the loops that cannot be compiled here (a10 / a12), replayed against
the restatement: tests/dsp/test_rtl_symbol_cache_generation.c (getSymbol()'s RTL-FSK sample loop on ramp inputs)."""
import ctypes as C

import numpy as np

import orc


class Sym(C.Structure):
    _fields_ = [("out_rate", C.c_int), ("sym_rate", C.c_int), ("rf_mod", C.c_int), ("l_edge", C.c_int), ("r_edge", C.c_int),
                ("sps_accum", C.c_int), ("jitter", C.c_int), ("last_sps", C.c_int), ("last_centre", C.c_int),
                ("lastsample", C.c_float), ("center", C.c_float), ("min", C.c_float), ("max", C.c_float),
                ("minref", C.c_float), ("maxref", C.c_float)]


def symbolizer(rate, sym_rate, rf_mod, l, r):
    o = orc.oracle()
    o.orc_symbolizer_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    o.orc_symbolizer_symbol.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    o.orc_symbolizer_symbol.restype = C.c_long
    s = Sym()
    o.orc_symbolizer_init(C.byref(s), rate, sym_rate, rf_mod, l, r)
    return s


def take(s, ramp, pos, have_sync):
    out = C.c_float(0)
    x = np.ascontiguousarray(ramp[pos:], np.float32)
    k = orc.oracle().orc_symbolizer_symbol(C.byref(s), x.ctypes.data, len(x), have_sync, C.byref(out))
    assert k > 0
    return out.value, pos + k


def ramp(base, n=200):
    # fake_rtl_read (test_rtl_symbol_cache_generation.c:134-183): four samples base + {0,1,2,3} per read, base += step (4)
    return (base + np.arange(n)).astype(np.float32)


def test_symbolizer_nominal_sps10(built):
    """:414-419 48 kHz / 4800 sym/s (decoder fixture rf_mod = 2, GFSK window centre -+ 1): symbol 1004, sps 10, centre 4, no
    crossing latched, ten samples consumed (three reads of four, two left in the cache)"""
    s = symbolizer(48000, 4800, 2, 1, 1)
    v, pos = take(s, ramp(1000.0), 0, 1)
    assert v == 1004.0 and s.last_sps == 10 and s.last_centre == 4 and s.jitter == -1 and pos == 10
    # the P25 C4FM window (l = r = 2) gives the same value on a ramp
    s = symbolizer(48000, 4800, 0, 2, 2)
    assert take(s, ramp(1000.0), 0, 1)[0] == 1004.0


def test_symbolizer_nxdn48_double_window(built):
    """:421-436 2400 sym/s at 48 kHz: sps 20, centre 9; samples 7..13 are added first and the GFSK pair (8, 10) again:
    (70 + 18) / 9 above the base = 2009.7778"""
    s = symbolizer(48000, 2400, 2, 1, 1)
    v, pos = take(s, ramp(2000.0), 0, 1)
    assert abs(v - 2009.7778) < 0.01 and v == np.float32(2000.0 + 88.0 / 9.0) and s.last_sps == 20 and s.last_centre == 9 and pos == 20
    # C4FM window at sps 20: 7..13 plus 7..11 again -> (70 + 45) / 12
    s = symbolizer(48000, 2400, 0, 2, 2)
    assert take(s, ramp(2000.0), 0, 1)[0] == np.float32(np.float32(24000.0 + 115.0) / np.float32(12.0))


def test_symbolizer_fractional_sps_accumulator(built):
    """:438-468 9600 sym/s at 24 kHz: 2.5 samples per symbol -> 2, 3, 2 ... with the remainder accumulator 4800, 0, 4800;
    GFSK at sps <= 4 takes the centre sample"""
    s = symbolizer(24000, 9600, 2, 1, 1)
    v, pos = take(s, ramp(5000.0), 0, 0)
    assert v == 5000.0 and s.last_sps == 2 and s.last_centre == 0 and s.jitter == -1
    s = symbolizer(24000, 9600, 2, 1, 1)
    r = ramp(5000.0)
    v, pos = take(s, r, 0, 1)
    assert v == 5000.0 and s.last_sps == 2 and s.sps_accum == 4800
    v, pos = take(s, r, pos, 1)
    assert v == 5003.0 and s.last_sps == 3 and s.last_centre == 1 and s.sps_accum == 0
    v, pos = take(s, r, pos, 1)
    assert v == 5005.0 and s.last_sps == 2 and s.sps_accum == 4800


def test_symbolizer_slip_by_latched_crossing(built):
    """:470-493 a crossing latched at the symbol centre moves the next hunting symbol one sample late: 7004, then (with
    jitter = centre, have_sync = 0) eleven samples and 7015; the latch is cleared"""
    s = symbolizer(48000, 4800, 2, 1, 1)
    r = ramp(7000.0)
    v, pos = take(s, r, 0, 1)
    assert v == 7004.0 and pos == 10
    s.jitter = s.last_centre
    v, pos = take(s, r, pos, 0)
    assert abs(v - 7015.0) < 0.01 and s.jitter == -1 and pos == 21
    # in frame the same latch does nothing (symbol_adjust_timing_index returns at have_sync != 0, dsd_symbol.c:498-502)
    s = symbolizer(48000, 4800, 0, 2, 2)
    v, pos = take(s, r, 0, 1)
    s.jitter = 4
    v, pos = take(s, r, pos, 1)
    assert v == 7014.0 and pos == 20 and s.jitter == 4
    # C4FM rule: 0 < jitter <= centre -> one sample late; centre < jitter < sps -> one sample early
    for jit, want, used in ((4, 7015.0, 11), (7, 7013.0, 9), (0, 7014.0, 10)):
        s = symbolizer(48000, 4800, 0, 2, 2)
        take(s, r, 0, 1)
        s.jitter = jit
        v, p2 = take(s, r, 10, 0)
        assert v == want and p2 - 10 == used, (jit, v, p2)


def test_p25_receive_loop_restatement_agrees_with_the_general_symbolizer(built):
    """ddn_oracle_rx.c hard-wires the C4FM / sps 10 case of the loop above: while hunting (no sync ever found, so no matched
    filter, no clipping) its symbols must equal the general restatement's sample for sample."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(6000) * 6000).astype(np.float32)
    sym, rec4, fl = orc.OracleP25Rx(lock_symbols=840, use_filter=1).run(x)
    assert not (fl & 3).any()
    s = symbolizer(48000, 4800, 0, 2, 2)
    pos, got = 0, []
    while True:
        out = C.c_float(0)
        seg = np.ascontiguousarray(x[pos:])
        k = orc.oracle().orc_symbolizer_symbol(C.byref(s), seg.ctypes.data, len(seg), 0, C.byref(out))
        if k < 0:
            break
        # the hunting loop copies max / min into the reference levels from the eighth symbol on (dsd_frame_sync.c:2316-2336)
        if len(got) + 1 >= 8:
            s.maxref, s.minref = s.max, s.min
        got.append(out.value)
        pos += k
    assert len(got) == len(sym) and np.array_equal(np.array(got, np.float32).view(np.uint32), sym.view(np.uint32))


def test_division_by_five_as_mul_and_two_fma_is_the_ieee_quotient(tmp_path):
    """the lean run's x / 5.0f (dsd_symbol.c's window mean) is computed as q = x * RN(1/5), r = fma(-5, q, x), q + r * RN(1/5)
    (ddn_rx.hip: div5_exact).  That equals the correctly rounded quotient for every finite binary32 x except -0 - checked over all
    2^32 patterns once (64 s); here every 251st pattern plus the neighbourhood of every power of two, compiled with the host's gcc"""
    import subprocess
    src = tmp_path / "div5.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static int check(uint32_t u) {
    float x; memcpy(&x, &u, 4);
    if (!isfinite(x) || u == 0x80000000u) return 0;
    volatile float want = x / 5.0f;
    const float y = 0.2f, q0 = x * y, r = fmaf(-5.0f, q0, x), q1 = fmaf(r, y, q0);
    float w = want; uint32_t a, c; memcpy(&a, &w, 4); memcpy(&c, &q1, 4);
    return a != c;
}
int main(void) {
    unsigned long bad = 0, n = 0;
    for (uint64_t b = 0; b < (1ull << 32); b += 251) { bad += check((uint32_t)b); n++; }
    for (uint32_t e = 0; e < 256; e++) for (uint32_t s = 0; s < 2; s++) for (int d = -4096; d <= 4096; d++) { bad += check(((s << 31) | (e << 23)) + (uint32_t)d); n++; }
    printf("%lu %lu\n", n, bad);
    return bad != 0;
}
''')
    exe = tmp_path / "div5"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    n, bad = map(int, out.stdout.split())
    assert n > 2 ** 24 and bad == 0
