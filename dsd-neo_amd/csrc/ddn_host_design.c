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

/* FLL band-edge filter pair + loop gains for `sps` samples per symbol (host-side design, like the reference's
 * fll_band_edge_design_filter / fll_configure_loop_params, src/dsp/costas.cpp:620-630,979-1070: GNU Radio's
 * fll_band_edge_cc design with rolloff 0.2 and 2*sps+1 taps, loop bandwidth 2*pi/sps/350, damping sqrt(2)/2).
 * taps4 = lower_r | lower_i | upper_r | upper_i, each DDN_FLL_MAX_TAPS long, stored reversed like the reference.
 * Returns n_taps. */
int
ddn_design_fll_band_edge(int sps, float* taps4, float* alpha, float* beta) {
    const float two_pi = 6.28318530717958647692f, pi = 3.14159265358979323846f;
    const float rolloff = 0.2f;
    int n_taps = 2 * sps + 1;
    if (n_taps > DDN_FLL_MAX_TAPS) {
        n_taps = DDN_FLL_MAX_TAPS;
    }
    if (n_taps < 3) {
        n_taps = 3;
    }
    const float M = roundf((float)n_taps / (float)sps);
    const int N = (n_taps - 1) / 2;
    float bb[DDN_FLL_MAX_TAPS];
    float power = 0.0f;
    for (int i = 0; i < n_taps; i++) {
        const float k = -M + (float)i * 2.0f / (float)sps;
        const float am = rolloff * k - 0.5f, ap = rolloff * k + 0.5f;
        const float sm = (fabsf(am) < 1e-6f) ? 1.0f : sinf(pi * am) / (pi * am);
        const float sp = (fabsf(ap) < 1e-6f) ? 1.0f : sinf(pi * ap) / (pi * ap);
        bb[i] = sm + sp;
        power += bb[i] * bb[i];
    }
    if (power > 0.0f) {
        const float norm = 1.0f / power;
        for (int i = 0; i < n_taps; i++) {
            bb[i] *= norm;
        }
    }
    for (int i = 0; i < 4 * DDN_FLL_MAX_TAPS; i++) {
        taps4[i] = 0.0f;
    }
    for (int i = 0; i < n_taps; i++) {
        const float freq = (float)(-N + i) / (2.0f * (float)sps);
        const float phase = two_pi * (1.0f + rolloff) * freq;
        const int r = n_taps - 1 - i;
        taps4[r] = bb[i] * cosf(-phase);
        taps4[DDN_FLL_MAX_TAPS + r] = bb[i] * sinf(-phase);
        taps4[2 * DDN_FLL_MAX_TAPS + r] = bb[i] * cosf(phase);
        taps4[3 * DDN_FLL_MAX_TAPS + r] = bb[i] * sinf(phase);
    }
    const float loop_bw = two_pi / (float)sps / 350.0f;
    const float damping = 0.70710678118654752440f;
    const float denom = 1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw;
    *alpha = (4.0f * damping * loop_bw) / denom;
    *beta = (4.0f * loop_bw * loop_bw) / denom;
    return n_taps;
}


/* Polyphase prototype of the rational resampler (resampler_design_taps, src/dsp/resampler.cpp:166-190; 16 taps per
 * phase :253-260): Hamming-windowed sinc at fc = 0.45 / max(L, M) evaluated in binary64, unity DC gain, times L,
 * stored per phase with the oldest tap first.  taps holds 16 * L floats; returns 16 * L. */
static double
rs_sinc(double x) {
    const double pi = 3.14159265358979323846;
    return x == 0.0 ? 1.0 : sin(pi * x) / (pi * x);
}

int
ddn_design_resampler(int L, int M, float* taps) {
    const double pi = 3.14159265358979323846;
    const int K = 16, total = K * L, mid = (total - 1) / 2;
    const double fc = 0.45 / (double)((L > M) ? L : M);
    double gain = 0.0;
    for (int n = 0; n < total; n++) {
        const double w = 0.54 - 0.46 * cos(2.0 * pi * (double)n / (double)(total - 1));
        gain += 2.0 * fc * rs_sinc(2.0 * fc * (double)(n - mid)) * w;
    }
    if (gain == 0.0) {
        gain = 1.0;
    }
    for (int phase = 0; phase < L; phase++) {
        for (int k = 0; k < K; k++) {
            const int src = phase + ((K - 1 - k) * L);
            const double w = 0.54 - 0.46 * cos(2.0 * pi * (double)src / (double)(total - 1));
            const double h = 2.0 * fc * rs_sinc(2.0 * fc * (double)(src - mid));
            taps[phase * K + k] = (float)((h * w / gain) * (double)L);
        }
    }
    return total;
}
