/*
 * oracle/ddn_oracle_dsp.c — CPU restatement of the dsd-neo DSP front end (TEST INFRASTRUCTURE ONLY).
 *
 * Restates, for the FSK-discriminator digital path only (full_demod, src/dsp/demod_pipeline.cpp:1330-1350):
 *   cu8 widen            src/dsp/simd_widen.cpp:139-149
 *   half-band cascade    src/dsp/simd_fir.cpp:139-222, src/dsp/demod_pipeline.cpp:983-1001
 *   channel LPF design   src/dsp/firdes.cpp (Blackman low_pass), src/dsp/demod_pipeline.cpp:443-524
 *   channel LPF apply    src/dsp/simd_fir.cpp:55-133 (scalar), src/dsp/simd_fir_avx2.cpp:119-143,400-447 (FMA)
 *   power / squelch      src/dsp/demod_pipeline.cpp:926-945,1003-1020
 *   IQ DC block / balance src/dsp/demod_pipeline.cpp:948-978,1131-1171 (optional, default off)
 *   FSK discriminator    src/dsp/fsk_modem.c:23-35,89-164
 *
 * Pinned against the reference's own objects by tests/test_oracle_vs_ref.py (bit-exact on this container's
 * libm) and against tests/golden/ vectors elsewhere.  Compile with -ffp-contract=off (oracle/Makefile).
 */
#include "ddn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------- */
/* widen: (u8 - 127.5) * (1/127.5), src/dsp/simd_widen.cpp:139-149                                    */
void
orc_widen_u8(const uint8_t* src, float* dst, size_t len) {
    const float inv = 1.0f / 127.5f;
    for (size_t i = 0; i < len; i++) {
        dst[i] = ((float)src[i] - 127.5f) * inv;
    }
}

/* ------------------------------------------------------------------------------------------------- */
/* GNU-Radio style windowed-sinc low-pass, Blackman window (src/dsp/firdes.cpp: window :~95-104,      */
/* ntaps rule :~205-216, low_pass body :~218-280).  Window in float/cosf, sinc in double, gain         */
/* normalised by the float-rounded taps summed in double.                                              */
#define ORC_PI 3.14159265358979323846

int
orc_firdes_low_pass_blackman(double gain, double fs, double cutoff, double transition, float* taps_out,
                             int max_taps) {
    if (fs <= 0.0 || cutoff <= 0.0 || cutoff > fs / 2.0 || transition <= 0.0) {
        return -1;
    }
    int ntaps = (int)(74.0 * fs / (22.0 * transition)); /* Blackman max attenuation = 74 dB */
    if ((ntaps & 1) == 0) {
        ntaps++;
    }
    if (ntaps > max_taps || ntaps > 1024) {
        return -1;
    }
    float win[1024];
    const float Mf = (float)(ntaps - 1);
    for (int n = 0; n < ntaps; n++) {
        win[n] = 0.42f - 0.5f * cosf((2.0f * (float)ORC_PI * (float)n) / Mf)
                 + 0.08f * cosf((4.0f * (float)ORC_PI * (float)n) / Mf);
    }
    const int M = (ntaps - 1) / 2;
    const double w0 = 2.0 * ORC_PI * cutoff / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0) {
            taps_out[M] = (float)((w0 / ORC_PI) * win[M]);
        } else {
            taps_out[n + M] = (float)((sin(n * w0) / (n * ORC_PI)) * win[n + M]);
        }
    }
    double dc = taps_out[M];
    for (int n = 1; n <= M; n++) {
        dc += 2.0 * taps_out[n + M];
    }
    gain /= dc;
    for (int i = 0; i < ntaps; i++) {
        taps_out[i] *= (float)gain;
    }
    return ntaps;
}

/* profile -> cutoff (transition-band centre), src/dsp/demod_pipeline.cpp:130-149,478-490; clamp :443-460.
 * The 63-tap static fallback tables (:150-440) are only reached when the design fails (ntaps > 144, i.e.
 * rate > ~51 kHz); the oracle reports that as an error instead of restating the tables. */
int
orc_channel_lpf_design(int rate_hz, int profile, float* taps_out, int max_taps) {
    const double transition = 1200.0, guard = 600.0;
    double cutoff;
    switch (profile) {
        case ORC_LPF_6K25: cutoff = 3125.0 + guard; break;
        case ORC_LPF_12K5:
        case ORC_LPF_PROVOICE:
        case ORC_LPF_P25_C4FM: cutoff = 6250.0 + guard; break;
        case ORC_LPF_P25_CQPSK: cutoff = 7250.0; break;
        case ORC_LPF_WIDE:
        default: cutoff = 8000.0 + guard; break;
    }
    if (rate_hz <= 0) {
        return -1;
    }
    const double max_cutoff = (double)rate_hz * 0.5 * 0.90;
    if (cutoff < 100.0) {
        cutoff = 100.0;
    }
    if (cutoff > max_cutoff) {
        cutoff = max_cutoff;
    }
    if (max_taps > ORC_MAX_TAPS) {
        max_taps = ORC_MAX_TAPS;
    }
    return orc_firdes_low_pass_blackman(1.0, (double)rate_hz, cutoff, transition, taps_out, max_taps);
}

/* ------------------------------------------------------------------------------------------------- */
/* Zero-latency symmetric complex FIR with block-edge replication.                                     */
/* Indexing: src/dsp/simd_fir.cpp:55-133.  For output n the window is centred on input n; samples      */
/* before the block come from hist (last taps_len-1 inputs of the previous block), samples past the     */
/* block end are the block's last sample.  Order per output: centre tap, then k = 0..centre-1 with      */
/* zero taps skipped.  fma_order selects acc = fmaf(h, xm+xp, acc) (AVX2 unit, src/dsp/                 */
/* simd_fir_avx2.cpp:119-143) vs acc += h*(xm+xp) with separate rounding (scalar unit).                 */
typedef struct {
    const float* in;
    const float* hi;
    const float* hq;
    int hist_len;
    int n;
    float last_i, last_q;
} orc_iq_src;

static inline void
orc_fetch(const orc_iq_src* s, int idx, float* xi, float* xq) {
    if (idx < s->hist_len) {
        *xi = s->hi[idx];
        *xq = s->hq[idx];
    } else {
        int rel = idx - s->hist_len;
        if (rel < s->n) {
            *xi = s->in[2 * rel];
            *xq = s->in[2 * rel + 1];
        } else {
            *xi = s->last_i;
            *xq = s->last_q;
        }
    }
}

static void
orc_update_hist(const float* in, int n, float* hi, float* hq, int hist_len) {
    if (n >= hist_len) {
        for (int k = 0; k < hist_len; k++) {
            hi[k] = in[2 * (n - hist_len + k)];
            hq[k] = in[2 * (n - hist_len + k) + 1];
        }
    } else {
        int keep = hist_len - n;
        memmove(hi, hi + n, (size_t)keep * sizeof(float));
        memmove(hq, hq + n, (size_t)keep * sizeof(float));
        for (int k = 0; k < n; k++) {
            hi[keep + k] = in[2 * k];
            hq[keep + k] = in[2 * k + 1];
        }
    }
}

static inline void
orc_sym_taps(const orc_iq_src* s, const float* taps, int center, int center_idx, int tap_step, int fma_order,
             float* oi, float* oq) {
    float ci, cq;
    orc_fetch(s, center_idx, &ci, &cq);
    float ai, aq;
    if (fma_order) {
        ai = fmaf(taps[center], ci, 0.0f);
        aq = fmaf(taps[center], cq, 0.0f);
    } else {
        ai = 0.0f + taps[center] * ci;
        aq = 0.0f + taps[center] * cq;
    }
    for (int k = 0; k < center; k += tap_step) {
        float h = taps[k];
        if (h == 0.0f) {
            continue;
        }
        int d = center - k;
        float mi, mq, pi, pq;
        orc_fetch(s, center_idx - d, &mi, &mq);
        orc_fetch(s, center_idx + d, &pi, &pq);
        float si = mi + pi, sq = mq + pq;
        if (fma_order) {
            ai = fmaf(h, si, ai);
            aq = fmaf(h, sq, aq);
        } else {
            ai += h * si;
            aq += h * sq;
        }
    }
    *oi = ai;
    *oq = aq;
}

void
orc_fir_complex_apply(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                      int taps_len, int fma_order) {
    if (taps_len < 3 || (taps_len & 1) == 0 || in_len < 2) {
        return;
    }
    /* blocks shorter than 2*taps_len floats always take the scalar unit (src/dsp/simd_fir.cpp:303-306,350-356) */
    if (in_len < taps_len * 2) {
        fma_order = 0;
    }
    const int n = in_len >> 1;
    const int hist_len = taps_len - 1;
    const int center = hist_len >> 1;
    orc_iq_src s = {in, hist_i, hist_q, hist_len, n, in[2 * (n - 1)], in[2 * (n - 1) + 1]};
    for (int i = 0; i < n; i++) {
        orc_sym_taps(&s, taps, center, hist_len + i, 1, fma_order, &out[2 * i], &out[2 * i + 1]);
    }
    orc_update_hist(in, n, hist_i, hist_q, hist_len);
}

/* half-band /2: only even taps + centre are non-zero (src/dsp/simd_fir.cpp:139-222); window centred on
 * input 2n; same edge rules. */
const float orc_hb15_taps[15] = {-108.0f / 32768.0f, 0.0f, 1800.0f / 32768.0f, 0.0f, -500.0f / 32768.0f, 0.0f,
                                 7000.0f / 32768.0f, 0.5f, 7000.0f / 32768.0f, 0.0f, -500.0f / 32768.0f, 0.0f,
                                 1800.0f / 32768.0f, 0.0f, -108.0f / 32768.0f};
const float orc_hb31_taps[31] = {
    0.0f, 0.0f, 13.0f / 32768.0f, 0.0f, -73.0f / 32768.0f, 0.0f, 233.0f / 32768.0f, 0.0f, -587.0f / 32768.0f,
    0.0f, 1314.0f / 32768.0f, 0.0f, -2953.0f / 32768.0f, 0.0f, 10244.0f / 32768.0f, 16386.0f / 32768.0f,
    10244.0f / 32768.0f, 0.0f, -2953.0f / 32768.0f, 0.0f, 1314.0f / 32768.0f, 0.0f, -587.0f / 32768.0f, 0.0f,
    233.0f / 32768.0f, 0.0f, -73.0f / 32768.0f, 0.0f, 13.0f / 32768.0f, 0.0f, 0.0f};

int
orc_hb_decim2_complex(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                      int taps_len, int fma_order) {
    if (taps_len < 3 || (taps_len & 1) == 0) {
        return 0;
    }
    const int n = in_len >> 1;
    if (n <= 0) {
        return 0;
    }
    if (in_len < taps_len * 2) {
        fma_order = 0;
    }
    const int n_out = n >> 1;
    const int hist_len = taps_len - 1;
    const int center = hist_len >> 1;
    orc_iq_src s = {in, hist_i, hist_q, hist_len, n, in[in_len - 2], in[in_len - 1]};
    for (int i = 0; i < n_out; i++) {
        orc_sym_taps(&s, taps, center, hist_len + 2 * i, 2, fma_order, &out[2 * i], &out[2 * i + 1]);
    }
    orc_update_hist(in, n, hist_i, hist_q, hist_len);
    return n_out << 1;
}

/* DC-corrected mean power, double accumulation, src/dsp/demod_pipeline.cpp:926-945 */
float
orc_mean_power(const float* samples, int len, int step) {
    double p = 0.0, t = 0.0;
    for (int i = 0; i < len; i += step) {
        double s = (double)samples[i];
        t += s;
        p += s * s;
    }
    double dc = (len > 0) ? (t * t) / (double)len : 0.0;
    double e = p - dc;
    if (e < 0.0) {
        e = 0.0;
    }
    return (float)(e / (double)(len > 0 ? len : 1));
}

/* ------------------------------------------------------------------------------------------------- */
/* FSK discriminator, src/dsp/fsk_modem.c:23-35 (phase), :98-105 (centring), :116-133 (AGC), :135-164   */
void
orc_fsk_reset(orc_fsk_state* st) {
    memset(st, 0, sizeof(*st));
}

float
orc_fsk_phase_delta(float cur_i, float cur_q, float prev_i, float prev_q) {
    float re = cur_i * prev_i + cur_q * prev_q;
    float im = cur_q * prev_i - cur_i * prev_q;
    if (re > 1.0e-7f && fabsf(im) <= (0.35f * re)) {
        float x = im / re;
        float x2 = x * x;
        return x * (1.0f + x2 * (-0.3333333333333333f + x2 * 0.2f));
    }
    return atan2f(im, re);
}

int
orc_fsk_discriminator(orc_fsk_state* st, const float* iq, int len_interleaved, float* out, int max_out) {
    if (!st || !out || max_out <= 0 || !iq || len_interleaved < 2) {
        return 0;
    }
    const int pairs = len_interleaved >> 1;
    int w = 0;
    for (int n = 0; n < pairs && w < max_out; n++) {
        float ci = iq[2 * n], cq = iq[2 * n + 1];
        if (!st->have_prev) {
            st->prev_i = ci;
            st->prev_q = cq;
            st->have_prev = 1;
            out[w++] = 0.0f;
            continue;
        }
        float f = orc_fsk_phase_delta(ci, cq, st->prev_i, st->prev_q);
        st->dc_est += 0.00025f * (f - st->dc_est);
        float c = f - st->dc_est;
        float mag = fabsf(c);
        if (mag > 1.0e-7f) {
            if (st->peak_est <= 1.0e-7f) {
                st->peak_est = mag;
            } else if (mag > st->peak_est) {
                st->peak_est += 0.125f * (mag - st->peak_est);
            } else {
                st->peak_est += 0.00005f * (mag - st->peak_est);
            }
        }
        float peak = st->peak_est;
        if (peak <= 1.0e-7f) {
            peak = 1.0f;
        }
        float y = c * (30000.0f / peak);
        if (y > 32767.0f) {
            y = 32767.0f;
        } else if (y < -32768.0f) {
            y = -32768.0f;
        }
        out[w++] = y;
        st->prev_i = ci;
        st->prev_q = cq;
    }
    return w;
}

/* ------------------------------------------------------------------------------------------------- */
/* one-channel front end == full_demod() for output_kind FSK_DISCRIMINATOR                             */
void
orc_fe_init(orc_front_end* fe, int rate_hz, int profile, int lpf_enable, float squelch_level, int downsample_passes,
            int fma_order) {
    memset(fe, 0, sizeof(*fe));
    fe->rate_hz = rate_hz;
    fe->lpf_enable = lpf_enable;
    fe->squelch_level = squelch_level;
    fe->downsample_passes = downsample_passes;
    fe->fma_order = fma_order;
    fe->taps_len = lpf_enable ? orc_channel_lpf_design(rate_hz, profile, fe->taps, ORC_MAX_TAPS) : 0;
}

int
orc_fe_block_f32(orc_front_end* fe, const float* iq, int n_complex, float* out, float* scratch) {
    float* a = scratch;
    float* b = scratch + 2 * (size_t)n_complex;
    const float* cur = iq;
    int len = 2 * n_complex;
    /* half-band cascade: 31-tap first stage, 15-tap afterwards (src/dsp/demod_pipeline.cpp:983-1001) */
    for (int i = 0; i < fe->downsample_passes; i++) {
        float* dst = (cur == a) ? b : a;
        len = orc_hb_decim2_complex(cur, len, dst, fe->hb_hist_i[i], fe->hb_hist_q[i],
                                    i == 0 ? orc_hb31_taps : orc_hb15_taps, i == 0 ? 31 : 15, fe->fma_order);
        cur = dst;
    }
    /* channel LPF (:526-555) */
    if (fe->lpf_enable && len >= 2 && fe->taps_len >= 3) {
        float* dst = (cur == a) ? b : a;
        orc_fir_complex_apply(cur, len, dst, fe->hist_i, fe->hist_q, fe->taps, fe->taps_len, fe->fma_order);
        cur = dst;
    }
    /* channel power over the first <=512 floats, squelch gate (:1003-1020) */
    if (len >= 2) {
        fe->channel_pwr = orc_mean_power(cur, len > 512 ? 512 : len, 1);
    }
    const int pairs = len >> 1;
    if (len > 0 && fe->squelch_level > 0.0f && fe->channel_pwr < fe->squelch_level) {
        fe->channel_squelched = 1;
        orc_fsk_reset(&fe->fsk); /* :1179-1184 */
        for (int i = 0; i < pairs; i++) {
            out[i] = 0.0f;
        }
        return pairs;
    }
    fe->channel_squelched = 0;
    if ((fe->iq_dc_enable || fe->iqbal_enable) && len >= 2) {
        float* w = (cur == a) ? a : b;
        if (cur != a && cur != b) { /* no stage ran yet: work on a copy, the caller's input stays untouched */
            memcpy(b, cur, sizeof(float) * (size_t)len);
        }
        cur = w;
        if (fe->iq_dc_enable) { /* iq_dc_block(), :948-978: leaky integrator, alpha = 2^-k, k clamped to 6..15 */
            int k = fe->iq_dc_shift;
            k = k < 6 ? 6 : (k > 15 ? 15 : k);
            const float alpha = 1.0f / (float)(1 << k);
            float dcI = fe->iq_dc_r, dcQ = fe->iq_dc_i;
            for (int n = 0; n < pairs; n++) {
                const float I = w[2 * n], Q = w[2 * n + 1];
                dcI += (I - dcI) * alpha;
                dcQ += (Q - dcQ) * alpha;
                w[2 * n] = I - dcI;
                w[2 * n + 1] = Q - dcQ;
            }
            fe->iq_dc_r = dcI;
            fe->iq_dc_i = dcQ;
        }
        if (fe->iqbal_enable) { /* full_demod_apply_iq_balance(), :1131-1171 */
            double s2r = 0.0, s2i = 0.0, p2 = 0.0;
            for (int n = 0; n < pairs; n++) {
                const double I = (double)w[2 * n], Q = (double)w[2 * n + 1];
                s2r += I * I - Q * Q;
                s2i += 2.0 * I * Q;
                p2 += I * I + Q * Q;
            }
            if (p2 <= 1e-9) {
                p2 = 1e-9;
            }
            float er = fe->iqbal_er, ei = fe->iqbal_ei;
            const float ar = (float)(s2r / p2), ai = (float)(s2i / p2);
            const float ema = fe->iqbal_ema_a > 0.0f ? fe->iqbal_ema_a : 0.2f;
            er += ema * (ar - er);
            ei += ema * (ai - ei);
            fe->iqbal_er = er;
            fe->iqbal_ei = ei;
            const float thr = fe->iqbal_thr > 0.0f ? fe->iqbal_thr : 0.02f;
            if (!((er * er + ei * ei) < (thr * thr))) {
                for (int n = 0; n < pairs; n++) {
                    const float I = w[2 * n], Q = w[2 * n + 1];
                    const float tI = er * I + ei * Q;
                    const float tQ = -er * Q + ei * I;
                    w[2 * n] = I - tI;
                    w[2 * n + 1] = Q - tQ;
                }
            }
        }
    }
    return orc_fsk_discriminator(&fe->fsk, cur, len, out, pairs > 0 ? pairs : 1);
}

void
orc_fe_set_iq_options(orc_front_end* fe, int dc_enable, int dc_shift, int bal_enable, float bal_thr, float bal_ema_a) {
    fe->iq_dc_enable = dc_enable;
    fe->iq_dc_shift = dc_shift;
    fe->iqbal_enable = bal_enable;
    fe->iqbal_thr = bal_thr;
    fe->iqbal_ema_a = bal_ema_a;
}

long
orc_fe_run_f32(orc_front_end* fe, const float* iq, long n_complex, int block_len, float* out) {
    float* scratch = (float*)malloc(sizeof(float) * 4 * (size_t)block_len);
    long done = 0, w = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        w += orc_fe_block_f32(fe, iq + 2 * done, (int)n, out + w, scratch);
        done += n;
    }
    free(scratch);
    return w;
}

long
orc_fe_run_cu8(orc_front_end* fe, const uint8_t* iq, long n_complex, int block_len, float* out) {
    float* wide = (float*)malloc(sizeof(float) * 2 * (size_t)block_len);
    float* scratch = (float*)malloc(sizeof(float) * 4 * (size_t)block_len);
    long done = 0, w = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        orc_widen_u8(iq + 2 * done, wide, (size_t)(2 * n));
        w += orc_fe_block_f32(fe, wide, (int)n, out + w, scratch);
        done += n;
    }
    free(wide);
    free(scratch);
    return w;
}

void
orc_fe_run_batch_cu8(int n_channels, const uint8_t* iq, long n_complex, int block_len, int rate_hz, int profile,
                     float squelch_level, float* out) {
    orc_front_end* fe = (orc_front_end*)malloc(sizeof(orc_front_end));
    for (int c = 0; c < n_channels; c++) {
        orc_fe_init(fe, rate_hz, profile, 1, squelch_level, 0, 1);
        orc_fe_run_cu8(fe, iq + (size_t)c * 2 * (size_t)n_complex, n_complex, block_len,
                       out + (size_t)c * (size_t)n_complex);
    }
    free(fe);
}
