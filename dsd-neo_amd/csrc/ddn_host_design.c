/*
 * ddn_host_design.c — host-side (C) filter planning for the MI355X front end.
 *
 * Mirrors the host logic the reference runs once per (rate, profile) before its SIMD FIR:
 *   channel_lpf_ensure_plan / channel_lpf_design_low_pass / channel_lpf_cutoff_for_profile
 *     (reference src/dsp/demod_pipeline.cpp:443-524, constants :129-149)
 *   dsd_firdes_low_pass with DSD_WIN_BLACKMAN (reference src/dsp/firdes.cpp, GNU Radio firdes::low_pass rule)
 * Taps are designed on the host in the same float/double mix as the reference so that the device kernel
 * sees bit-identical coefficients, then uploaded once per batch.
 */
#include "ddn_internal.h"

#include <math.h>

static const double k_pi = 3.14159265358979323846;

static double
profile_transition_centre_hz(int profile) {
    /* protected channel edge + half the 1200 Hz transition band as guard (P25 CQPSK: fixed 7250 Hz) */
    switch (profile) {
        case DDN_LPF_6K25: return 3125.0 + 600.0;
        case DDN_LPF_12K5:
        case DDN_LPF_PROVOICE:
        case DDN_LPF_P25_C4FM: return 6250.0 + 600.0;
        case DDN_LPF_P25_CQPSK: return 7250.0;
        default: return 8000.0 + 600.0;
    }
}

int
ddn_design_channel_lpf(int rate_hz, int profile, float* taps, int max_taps) {
    if (rate_hz <= 0 || !taps) {
        return DDN_EINVAL;
    }
    const double fs = (double)rate_hz;
    const double tw = 1200.0;
    double fc = profile_transition_centre_hz(profile);
    if (fc < 100.0) {
        fc = 100.0;
    }
    if (fc > fs * 0.5 * 0.90) {
        fc = fs * 0.5 * 0.90;
    }
    /* tap count: Blackman attenuation 74 dB -> ntaps = 74*fs/(22*tw), forced odd */
    int nt = (int)(74.0 * fs / (22.0 * tw));
    nt |= 1;
    if (nt > max_taps || nt > DDN_MAX_TAPS) {
        return DDN_ERANGE;
    }
    const int half = (nt - 1) / 2;
    const float span = (float)(nt - 1);
    const double wc = 2.0 * k_pi * fc / fs;
    for (int i = 0; i < nt; i++) {
        const float fi = (float)i;
        const float w = 0.42f - 0.5f * cosf((2.0f * (float)k_pi * fi) / span)
                        + 0.08f * cosf((4.0f * (float)k_pi * fi) / span);
        const int m = i - half;
        const double ideal = (m == 0) ? (wc / k_pi) : (sin(m * wc) / (m * k_pi));
        taps[i] = (float)(ideal * w);
    }
    double dc_gain = taps[half];
    for (int m = 1; m <= half; m++) {
        dc_gain += 2.0 * taps[half + m];
    }
    const float g = (float)(1.0 / dc_gain);
    for (int i = 0; i < nt; i++) {
        taps[i] *= g;
    }
    return nt;
}
