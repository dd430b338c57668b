/*
 * oracle/ddn_oracle_cqpsk.c — CPU restatement of the P25 CQPSK/LSM chain that full_demod() runs after the channel
 * LPF when cqpsk_enable is set (TEST INFRASTRUCTURE ONLY):
 *
 *   RMS AGC (alpha 0.45, reference 0.85)          src/dsp/demod_pipeline.cpp:796-842
 *   FLL band-edge (sample rate)                   src/dsp/costas.cpp:636-781 (loop), :979-1070 (filter design),
 *                                                 :1176-1207 (driver); NCO sin/cos polynomial :80-133
 *   Gardner + MMSE                                oracle/ddn_oracle_ted.c (src/dsp/costas.cpp:352-534,804-858)
 *   differential phasor                           src/dsp/costas.cpp:871-901
 *   Costas loop (symbol rate)                     src/dsp/costas.cpp:536-603,934-962 (+ :179-259 detector helpers)
 *   phase extractor theta*4/pi                    src/dsp/demod_pipeline.cpp:63-98,742-764
 *   order of the chain                            src/dsp/demod_pipeline.cpp:1100-1118,1250-1257
 *
 * Everything is float, compiled -ffp-contract=off like the reference's non-AVX units; the band-edge taps use libm
 * sinf/cosf exactly where the reference does (host-side design code).  Pinned bit for bit against the compiled
 * reference (full_demod with cqpsk_enable, oracle/_ref) by tests/test_oracle_cqpsk.py.
 */
#include "ddn_oracle.h"

#include <math.h>
#include <string.h>

static const float kTwoPi = 6.28318530717958647692f;
static const float kPi = 3.14159265358979323846f;

static float
clipf_limit(float x, float lim) {
    if (x > lim) {
        return lim;
    }
    if (x < -lim) {
        return -lim;
    }
    return x;
}

static float
clampf_range(float v, float lo, float hi) {
    if (v < lo) {
        return lo;
    }
    if (v > hi) {
        return hi;
    }
    return v;
}

static void
sincos_half_pi(float phase, float* s, float* c) {
    const float x2 = phase * phase;
    *s = phase
         * (1.0f
            + x2
                  * (-0.16666666666666666667f
                     + x2
                           * (0.00833333333333333333f
                              + x2 * (-0.00019841269841269841f + x2 * (0.00000275573192239859f + x2 * -0.00000002505210838544f)))));
    *c = 1.0f
         + x2
               * (-0.5f
                  + x2
                        * (0.04166666666666666667f
                           + x2 * (-0.00138888888888888889f + x2 * (0.00002480158730158730f + x2 * -0.00000027557319223986f))));
}

static void
sincos_two_pi(float phase, float* s, float* c) {
    if (!isfinite(phase) || phase < -kTwoPi || phase > kTwoPi) {
        *s = sinf(phase);
        *c = cosf(phase);
        return;
    }
    if (phase > kPi) {
        phase -= kTwoPi;
    } else if (phase < -kPi) {
        phase += kTwoPi;
    }
    if (phase > (kPi / 2.0f)) {
        float sv, cv;
        sincos_half_pi(kPi - phase, &sv, &cv);
        *s = sv;
        *c = -cv;
        return;
    }
    if (phase < (-kPi / 2.0f)) {
        float sv, cv;
        sincos_half_pi(-kPi - phase, &sv, &cv);
        *s = sv;
        *c = -cv;
        return;
    }
    sincos_half_pi(phase, s, c);
}

/* ---- RMS AGC ---------------------------------------------------------------------------------------------- */
void
orc_cqpsk_rms_agc(float* avg_io, float* iq, int pairs) {
    const float alpha = 0.45f, beta = 1.0f - alpha, gain = 0.85f;
    float avg = *avg_io;
    if (avg <= 0.0f) {
        avg = 1.0f;
    }
    for (int i = 0; i < pairs; i++) {
        const float I = iq[2 * i], Q = iq[2 * i + 1];
        const float m2 = I * I + Q * Q;
        avg = beta * avg + alpha * m2;
        if (avg > 0.0f) {
            const float sc = gain / sqrtf(avg);
            iq[2 * i] = I * sc;
            iq[2 * i + 1] = Q * sc;
        }
    }
    *avg_io = avg;
}

/* ---- FLL band-edge ---------------------------------------------------------------------------------------- */
void
orc_fll_design(orc_fll* f, int sps) {
    const float rolloff = 0.2f;
    int n_taps = 2 * sps + 1;
    if (n_taps > ORC_FLL_MAX_TAPS) {
        n_taps = ORC_FLL_MAX_TAPS;
    }
    if (n_taps < 3) {
        n_taps = 3;
    }
    f->n_taps = n_taps;
    f->sps = sps;
    const float M = roundf((float)n_taps / (float)sps);
    const int N = (n_taps - 1) / 2;
    float bb[ORC_FLL_MAX_TAPS];
    float power = 0.0f;
    for (int i = 0; i < n_taps; i++) {
        const float k = -M + (float)i * 2.0f / (float)sps;
        const float am = rolloff * k - 0.5f, ap = rolloff * k + 0.5f;
        const float sm = (fabsf(am) < 1e-6f) ? 1.0f : sinf(kPi * am) / (kPi * am);
        const float sp = (fabsf(ap) < 1e-6f) ? 1.0f : sinf(kPi * ap) / (kPi * ap);
        bb[i] = sm + sp;
        power += bb[i] * bb[i];
    }
    if (power > 0.0f) {
        const float norm = 1.0f / power;
        for (int i = 0; i < n_taps; i++) {
            bb[i] *= norm;
        }
    }
    for (int i = 0; i < n_taps; i++) {
        const float freq = (float)(-N + i) / (2.0f * (float)sps);
        const float phase = kTwoPi * (1.0f + rolloff) * freq;
        const int r = n_taps - 1 - i;
        f->tlr[r] = bb[i] * cosf(-phase);
        f->tli[r] = bb[i] * sinf(-phase);
        f->tur[r] = bb[i] * cosf(phase);
        f->tui[r] = bb[i] * sinf(phase);
    }
    const float loop_bw = kTwoPi / (float)sps / 350.0f;
    const float damping = 0.70710678118654752440f;
    const float denom = 1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw;
    f->alpha = (4.0f * damping * loop_bw) / denom;
    f->beta = (4.0f * loop_bw * loop_bw) / denom;
    f->max_freq = 1.0f;
    f->min_freq = -1.0f;
    f->initialized = 1;
}

void
orc_fll_block(orc_fll* f, int sps, float* iq, int pairs) {
    if (!f->initialized || (f->sps > 0 && f->sps != sps)) {
        const int first = !f->initialized;
        orc_fll_design(f, sps);
        f->phase = 0.0f;
        if (first) {
            f->freq = 0.0f;
        }
        f->delay_idx = 0;
        memset(f->dr, 0, sizeof(f->dr));
        memset(f->di, 0, sizeof(f->di));
    }
    const int nt = f->n_taps;
    float phase = f->phase, freq = f->freq;
    int idx = f->delay_idx;
    for (int n = 0; n < pairs; n++) {
        const float ir = iq[2 * n], ii = iq[2 * n + 1];
        float ns, nc;
        sincos_two_pi(phase, &ns, &nc);
        const float orr = ir * nc - ii * ns;
        const float oi = ir * ns + ii * nc;
        f->dr[idx] = orr;
        f->di[idx] = oi;
        f->dr[idx + nt] = orr;
        f->di[idx + nt] = oi;
        float lr = 0.0f, li = 0.0f, ur = 0.0f, ui = 0.0f;
        const int base = idx + nt;
        for (int k = 0; k < nt; k++) {
            const float dr = f->dr[base - k], di = f->di[base - k];
            lr += dr * f->tlr[k] - di * f->tli[k];
            li += dr * f->tli[k] + di * f->tlr[k];
            ur += dr * f->tur[k] - di * f->tui[k];
            ui += dr * f->tui[k] + di * f->tur[k];
        }
        idx++;
        if (idx == nt) {
            idx = 0;
        }
        const float lm = lr * lr + li * li, um = ur * ur + ui * ui;
        const float err = clipf_limit(um - lm, 1.0f);
        freq += f->beta * err;
        freq = clampf_range(freq, f->min_freq, f->max_freq);
        phase += freq + f->alpha * err;
        while (phase > kTwoPi) {
            phase -= kTwoPi;
        }
        while (phase < -kTwoPi) {
            phase += kTwoPi;
        }
        iq[2 * n] = orr;
        iq[2 * n + 1] = oi;
    }
    f->phase = phase;
    f->freq = freq;
    f->delay_idx = idx;
}

/* ---- differential phasor ---------------------------------------------------------------------------------- */
void
orc_diff_phasor(float* prev_r, float* prev_j, float* iq, int pairs) {
    float pr = *prev_r, pj = *prev_j;
    for (int n = 0; n < pairs; n++) {
        const float cr = iq[2 * n], cj = iq[2 * n + 1];
        iq[2 * n] = cr * pr + cj * pj;
        iq[2 * n + 1] = cj * pr - cr * pj;
        pr = cr;
        pj = cj;
    }
    *prev_r = pr;
    *prev_j = pj;
}

/* ---- Costas loop ------------------------------------------------------------------------------------------ */
static float
smoothstep(float e0, float e1, float x) {
    if (x <= e0) {
        return 0.0f;
    }
    if (x >= e1) {
        return 1.0f;
    }
    const float t = (x - e0) / (e1 - e0);
    return t * t * (3.0f - 2.0f * t);
}

static float
detector_normalize(float re, float im, float* o_re, float* o_im) {
    const float floor2 = 0.10f * 0.10f, full2 = 0.35f * 0.35f, target = 0.85f * 0.85f;
    const float mag2 = re * re + im * im;
    if (!isfinite(mag2)) {
        *o_re = 0.0f;
        *o_im = 0.0f;
        return 0.0f;
    }
    if (mag2 <= floor2) {
        *o_re = re;
        *o_im = im;
        return 0.0f;
    }
    const float mag = sqrtf(mag2);
    const float conf = (mag2 >= full2) ? 1.0f : (isfinite(mag) ? smoothstep(0.10f, 0.35f, mag) : 0.0f);
    const float scale = target / mag;
    if (!isfinite(scale)) {
        *o_re = 0.0f;
        *o_im = 0.0f;
        return 0.0f;
    }
    *o_re = re * scale;
    *o_im = im * scale;
    return conf;
}

void
orc_costas_block(orc_costas* c, float* iq, int pairs) {
    if (!c->initialized) {
        const float loop_bw = 0.008f, damping = 0.70710678118654752440f;
        const float denom = 1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw;
        c->alpha = (4.0f * damping * loop_bw) / denom;
        c->beta = (4.0f * loop_bw * loop_bw) / denom;
        c->initialized = 1;
    }
    const float max_phase = kPi / 2.0f, min_phase = -max_phase;
    float phase = isfinite(c->phase) ? clampf_range(c->phase, min_phase, max_phase) : 0.0f;
    float freq = c->freq;
    float es = isfinite(c->error_smooth) ? c->error_smooth : 0.0f;
    float last_error = 0.0f;
    for (int n = 0; n < pairs; n++) {
        const float ir = iq[2 * n], ij = iq[2 * n + 1];
        float nj, nr;
        sincos_half_pi(-phase, &nj, &nr);
        const float rr = ir * nr - ij * nj;
        const float rj = ir * nj + ij * nr;
        float dr, dj;
        const float conf = detector_normalize(rr, rj, &dr, &dj);
        float error = 0.0f;
        if (conf <= 0.0f || !isfinite(conf)) {
            es = 0.0f;
        } else {
            const float pd = ((dr > 0.0f ? 1.0f : -1.0f) * dj - (dj > 0.0f ? 1.0f : -1.0f) * dr);
            const float raw = clipf_limit(pd * conf, 1.0f);
            float a;
            if (!isfinite(raw) || !isfinite(es) || fabsf(es) <= 1.0e-6f) {
                a = 0.25f;
            } else {
                const float kick = smoothstep(0.02f, 0.18f, fabsf(raw - es));
                a = 0.25f + (0.10f - 0.25f) * kick;
            }
            es += a * (raw - es);
            error = clipf_limit(es, 1.0f);
        }
        last_error = error;
        freq += c->beta * error;
        phase += freq + c->alpha * error;
        phase = clampf_range(phase, min_phase, max_phase);
        freq = clampf_range(freq, -1.0f, 1.0f);
        iq[2 * n] = dr;
        iq[2 * n + 1] = dj;
    }
    c->phase = phase;
    c->freq = freq;
    c->error = last_error;
    c->error_smooth = es;
}

/* ---- phase extractor --------------------------------------------------------------------------------------- */
static float
atan_unit(float x) {
    const float ax = fabsf(x);
    return x * (0.78539816339744830962f - (ax - 1.0f) * (0.2447f + 0.0663f * ax));
}

float
orc_atan2_qpsk(float y, float x) {
    if (x == 0.0f && y == 0.0f) {
        return 0.0f;
    }
    const float ax = fabsf(x), ay = fabsf(y);
    if (ax >= ay) {
        float a = atan_unit(y / x);
        if (x < 0.0f) {
            a += (y < 0.0f) ? -3.14159265358979323846f : 3.14159265358979323846f;
        }
        return a;
    }
    const float a = atan_unit(x / y);
    return (y > 0.0f) ? (1.57079632679489661923f - a) : (-1.57079632679489661923f - a);
}

/* ---- the chain: post-LPF complex block -> symbols ----------------------------------------------------------- */
void
orc_cqpsk_init(orc_cqpsk* c, int sps, int symbol_rate_hz, float ted_gain) {
    memset(c, 0, sizeof(*c));
    c->sps = sps;
    c->sym_rate = symbol_rate_hz;
    c->ted_gain = ted_gain;
    orc_ted_init(&c->ted);
}

/* iq: n complex samples (modified in place); work: >= 2n floats (the reference's separate timing_buf: the Gardner
 * stage may emit a symbol before it has read this block's first sample); out: symbols; returns the count */
int
orc_cqpsk_block(orc_cqpsk* c, float* iq, int n, float* out, float* work) {
    if (n < 1) {
        return 0;
    }
    orc_cqpsk_rms_agc(&c->agc_avg, iq, n);
    orc_fll_block(&c->fll, c->sps, iq, n);
    /* blocks shorter than 4 samples leave the reference's Gardner stage without output (costas.cpp:808-812 returns
     * early); callers keep blocks >= 4 samples */
    const int ns = orc_gardner_block(&c->ted, c->sps, c->ted_gain, c->sym_rate, iq, n, work) / 2;
    iq = work;
    orc_diff_phasor(&c->diff_prev_r, &c->diff_prev_j, iq, ns);
    orc_costas_block(&c->cos, iq, ns);
    const float k = 4.0f / 3.14159265358979323846f;
    for (int i = 0; i < ns; i++) {
        out[i] = orc_atan2_qpsk(iq[2 * i + 1], iq[2 * i]) * k;
    }
    return ns;
}

size_t
orc_cqpsk_sizeof(void) {
    return sizeof(orc_cqpsk);
}

/* ---- one channel's CQPSK front end == full_demod() with cqpsk_enable ------------------------------------------ */
void
orc_cqpsk_fe_init(orc_cqpsk_fe* fe, int rate_hz, int symbol_rate_hz, int profile, int lpf_enable, float ted_gain) {
    memset(fe, 0, sizeof(*fe));
    fe->taps_len = lpf_enable ? orc_channel_lpf_design(rate_hz, profile, fe->taps, ORC_MAX_TAPS) : 0;
    orc_cqpsk_init(&fe->chain, symbol_rate_hz > 0 ? rate_hz / symbol_rate_hz : 5, symbol_rate_hz, ted_gain);
}

/* scratch >= 4 * block_len floats; returns symbols written */
long
orc_cqpsk_fe_run_f32(orc_cqpsk_fe* fe, const float* iq, long n_complex, int block_len, float* out, float* scratch) {
    long done = 0, w = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        float* a = scratch;
        if (fe->taps_len >= 3) {
            orc_fir_complex_apply(iq + 2 * done, (int)(2 * n), a, fe->hist_i, fe->hist_q, fe->taps, fe->taps_len, 1);
        } else {
            memcpy(a, iq + 2 * done, sizeof(float) * 2 * (size_t)n);
        }
        w += orc_cqpsk_block(&fe->chain, a, (int)n, out + w, scratch + 2 * (size_t)block_len);
        done += n;
    }
    return w;
}

size_t
orc_cqpsk_fe_sizeof(void) {
    return sizeof(orc_cqpsk_fe);
}

/* out8 = {agc_avg, fll.freq, fll.phase, costas.phase, costas.freq, costas.error_smooth, ted.mu, ted.omega} */
void
orc_cqpsk_fe_get_state(const orc_cqpsk_fe* fe, float out8[8]) {
    out8[0] = fe->chain.agc_avg;
    out8[1] = fe->chain.fll.freq;
    out8[2] = fe->chain.fll.phase;
    out8[3] = fe->chain.cos.phase;
    out8[4] = fe->chain.cos.freq;
    out8[5] = fe->chain.cos.error_smooth;
    out8[6] = fe->chain.ted.mu;
    out8[7] = fe->chain.ted.omega;
}
