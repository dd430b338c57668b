"""The demod thread's mode matrix (ddn_mode_config, dsd-neo_amd/csrc/ddn_host_mode.c) against the expectations the reference's
own test holds for rtl_demod_init_for_mode() + the channel-LPF defaults (tests/io/test_io_rtl_demod_config.cpp:950-1090: every
single-protocol mode at 48 kHz and 24 kHz, the all-protocols AUTO start, the CQPSK samples-per-symbol cases) and against the
documented table (docs/rtl-demod-pipeline-audit.md:36-51).  Host logic: runs without a GPU."""
import ctypes as C

import pytest

import ddn

FSK, CQPSK = 1, 2
WIDE, K6, K12, PV, C4FM, CQ = ddn.LPF_WIDE, ddn.LPF_6K25, ddn.LPF_12K5, ddn.LPF_PROVOICE, ddn.LPF_P25_C4FM, ddn.LPF_P25_CQPSK
# (flags, kind, symbol rate, levels, profile) - test_io_rtl_demod_config.cpp:977-1068
CASES = [
    (dict(p25p1=1), FSK, 4800, 4, C4FM),
    (dict(p25p1=1, mod_qpsk=1), CQPSK, 4800, 4, CQ),
    (dict(p25p2=1, mod_qpsk=1), CQPSK, 6000, 4, CQ),
    (dict(nxdn48=1), FSK, 2400, 4, K6),
    (dict(nxdn96=1), FSK, 4800, 4, K12),
    (dict(dmr=1), FSK, 4800, 4, K12),
    (dict(dstar=1), FSK, 4800, 2, K6),
    (dict(x2tdma=1), FSK, 6000, 4, K12),
    (dict(ysf=1), FSK, 4800, 4, K12),
    (dict(dpmr=1), FSK, 2400, 4, K6),
    (dict(m17=1), FSK, 4800, 4, K12),
    (dict(provoice=1), FSK, 9600, 2, PV),
    (dict(p25p1=1, p25p2=1, dmr=1, nxdn48=1, nxdn96=1, x2tdma=1, ysf=1, dstar=1, dpmr=1, provoice=1, m17=1), FSK, 4800, 4, K12),
]


def run(flags, rate):
    f, r = ddn.ModeFlags(**flags), ddn.ModeResult()
    assert ddn.lib().ddn_mode_config(C.byref(f), rate, C.byref(r)) == 0
    return r


@pytest.mark.parametrize("flags,kind,sym,levels,profile", CASES)
@pytest.mark.parametrize("rate", [48000, 24000])
def test_mode_matrix(built, flags, kind, sym, levels, profile, rate):
    r = run(flags, rate)
    assert (r.output_kind, r.symbol_rate_hz, r.symbol_levels, r.lpf_profile) == (kind, sym, levels, profile)
    assert r.channel_lpf_enable == 1 and r.cqpsk_enable == (kind == CQPSK) and r.ted_enabled == (kind == CQPSK)


def test_cqpsk_samples_per_symbol_and_edges(built):
    # expect_sps cases, test_io_rtl_demod_config.cpp:955-971
    assert run(dict(p25p2=1, mod_qpsk=1), 48000).samples_per_symbol == 8
    assert run(dict(p25p2=1, mod_qpsk=1), 24000).samples_per_symbol == 4
    assert run(dict(p25p1=1, mod_qpsk=1), 48000).samples_per_symbol == 10
    assert run(dict(p25p1=1, mod_qpsk=1), 24000).samples_per_symbol == 5
    assert run(dict(p25p1=1, p25p2=1, mod_qpsk=1), 48000).samples_per_symbol == 10      # trunking starts on the control-channel rate
    # no digital mode / analog only: audio monitor; below 20 kHz the channel LPF is off (profile WIDE)
    assert run(dict(), 48000).output_kind == 0 and run(dict(dmr=1, analog_only=1), 48000).output_kind == 0
    r = run(dict(dmr=1), 16000)
    assert r.channel_lpf_enable == 0 and r.lpf_profile == WIDE
    assert ddn.lib().ddn_mode_config(None, 48000, C.byref(ddn.ModeResult())) != 0
