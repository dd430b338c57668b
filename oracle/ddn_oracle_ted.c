/*
 * oracle/ddn_oracle_ted.c — CPU restatement of the OP25-style Gardner timing recovery used on the CQPSK branch
 * (TEST INFRASTRUCTURE ONLY).
 *
 *   op25_gardner_cc + helpers      src/dsp/costas.cpp:352-534,804-858
 *   8-tap MMSE interpolator        src/dsp/mmse_interp.cpp:17-99 (GNU Radio interpolator_taps.h, every 8th row of
 *                                  the 128-step table, linearly interpolated between rows, taps applied reversed)
 *   state                          include/dsd-neo/dsp/ted.h:22-45
 * Float ops in the reference's order (compile with -ffp-contract=off).  Pinned bit-exact against the compiled
 * reference by tests/test_oracle_ted.py.
 */
#include "ddn_oracle.h"

#include <math.h>
#include <string.h>

static const float k_mmse[17][8] = {
    {0.00000e+00f, 0.00000e+00f, 0.00000e+00f, 0.00000e+00f, 1.00000e+00f, 0.00000e+00f, 0.00000e+00f, 0.00000e+00f},
    {-1.23337e-03f, 6.84261e-03f, -2.24178e-02f, 6.57852e-02f, 9.83392e-01f, -4.04519e-02f, 9.56876e-03f, -1.54221e-03f},
    {-2.43121e-03f, 1.35716e-02f, -4.49929e-02f, 1.36968e-01f, 9.55956e-01f, -7.43154e-02f, 1.80759e-02f, -2.94361e-03f},
    {-3.55283e-03f, 1.99599e-02f, -6.70018e-02f, 2.12443e-01f, 9.18329e-01f, -1.01501e-01f, 2.53295e-02f, -4.16581e-03f},
    {-4.55932e-03f, 2.57844e-02f, -8.77011e-02f, 2.91006e-01f, 8.71305e-01f, -1.22047e-01f, 3.11866e-02f, -5.17776e-03f},
    {-5.41467e-03f, 3.08323e-02f, -1.06342e-01f, 3.71376e-01f, 8.15826e-01f, -1.36111e-01f, 3.55525e-02f, -5.95620e-03f},
    {-6.08674e-03f, 3.49066e-02f, -1.22185e-01f, 4.52218e-01f, 7.52958e-01f, -1.43968e-01f, 3.83800e-02f, -6.48585e-03f},
    {-6.54823e-03f, 3.78315e-02f, -1.34515e-01f, 5.32164e-01f, 6.83875e-01f, -1.45993e-01f, 3.96678e-02f, -6.75943e-03f},
    {-6.77751e-03f, 3.94578e-02f, -1.42658e-01f, 6.09836e-01f, 6.09836e-01f, -1.42658e-01f, 3.94578e-02f, -6.77751e-03f},
    {-6.73929e-03f, 3.95900e-02f, -1.46043e-01f, 6.92808e-01f, 5.22267e-01f, -1.33190e-01f, 3.75341e-02f, -6.50285e-03f},
    {-6.48585e-03f, 3.83800e-02f, -1.43968e-01f, 7.52958e-01f, 4.52218e-01f, -1.22185e-01f, 3.49066e-02f, -6.08674e-03f},
    {-5.95620e-03f, 3.55525e-02f, -1.36111e-01f, 8.15826e-01f, 3.71376e-01f, -1.06342e-01f, 3.08323e-02f, -5.41467e-03f},
    {-5.17776e-03f, 3.11866e-02f, -1.22047e-01f, 8.71305e-01f, 2.91006e-01f, -8.77011e-02f, 2.57844e-02f, -4.55932e-03f},
    {-4.16581e-03f, 2.53295e-02f, -1.01501e-01f, 9.18329e-01f, 2.12443e-01f, -6.70018e-02f, 1.99599e-02f, -3.55283e-03f},
    {-2.94361e-03f, 1.80759e-02f, -7.43154e-02f, 9.55956e-01f, 1.36968e-01f, -4.49929e-02f, 1.35716e-02f, -2.43121e-03f},
    {-1.54221e-03f, 9.56876e-03f, -4.04519e-02f, 9.83392e-01f, 6.57852e-02f, -2.24178e-02f, 6.84261e-03f, -1.23337e-03f},
    {0.00000e+00f, 0.00000e+00f, 0.00000e+00f, 1.00000e+00f, 0.00000e+00f, 0.00000e+00f, 0.00000e+00f, 0.00000e+00f}
};

const float*
orc_mmse_table(void) {
    return &k_mmse[0][0];
}

static float
mmse_real(const float* s /* 8 samples, stride 2 */, float mu) {
    float pos = mu * 16.0f;
    int lo = (int)pos;
    float fr = pos - (float)lo;
    if (lo < 0) {
        lo = 0;
        fr = 0.0f;
    }
    if (lo >= 16) {
        lo = 15;
        fr = 1.0f;
    }
    const float lw = 1.0f - fr;
    float acc = 0.0f;
    for (int i = 0; i < 8; i++) {
        const float tap = lw * k_mmse[lo][i] + fr * k_mmse[lo + 1][i];
        acc += tap * s[2 * (7 - i)];
    }
    return acc;
}

void
orc_mmse_interp_complex(const float* samples, float mu, float* re, float* im) {
    *re = mmse_real(samples, mu);
    *im = mmse_real(samples + 1, mu);
}

static inline float
clipf(float x, float lim) {
    return x > lim ? lim : (x < -lim ? -lim : x);
}

void
orc_ted_init(orc_ted_state* t) {
    memset(t, 0, sizeof(*t));
}

/* gain_mu: op25_gardner_gain_mu_for_state (src/dsp/costas.cpp:143-168) with no env/API override:
 * 0.025 unless symbol rate >= 5500 and the Yair-Linn lock metric over >= 240 symbols is >= 0.05 (then 0.018). */
static float
gain_mu_for(const orc_ted_state* t, float ted_gain, int symbol_rate_hz) {
    const float req = ted_gain > 0.0f ? ted_gain : 0.025f;
    if (symbol_rate_hz < 5500 || t->lock_count < 240) {
        return req;
    }
    if (t->lock_accum / (float)t->lock_count < 0.05f) {
        return req;
    }
    return 0.018f;
}

/* One block: iq = n complex samples in, out = symbol-rate complex samples; returns floats written (0 if < 2). */
int
orc_gardner_block(orc_ted_state* t, int sps, float ted_gain, int symbol_rate_hz, const float* iq, int n, float* out) {
    if (n < 4) {
        return 0;
    }
    float omega = t->omega;
    const int first = (t->omega_mid == 0.0f || t->twice_sps < 2);
    if (first || (t->sps > 0 && t->sps != sps)) {
        t->mu = (float)sps;
        omega = (float)sps;
        t->omega = omega;
        t->omega_rel = 0.002f;
        t->omega_mid = omega;
        t->omega_min = omega * (1.0f - t->omega_rel);
        t->omega_max = omega * (1.0f + t->omega_rel);
        const int a = 2 * (int)ceilf(t->omega_max);
        const int b = (int)ceilf(t->omega_max / 2.0f) + 8 + 1;
        const int need = a > b ? a : b;
        if (need > ORC_TED_DL) {
            return 0;
        }
        t->twice_sps = need;
        t->dl_index = 0;
        t->sps = sps;
        t->dl[0] = 0.0f;
        t->dl[1] = 0.0f;
    }
    const float gain_mu = gain_mu_for(t, ted_gain, symbol_rate_hz);
    const float gain_omega = 0.1f * gain_mu * gain_mu;
    float mu = t->mu, last_r = t->last_r, last_j = t->last_j, lock = t->lock_accum;
    int lock_n = t->lock_count, dli = t->dl_index;
    const int tw = t->twice_sps;
    int i = 0, o = 0;
    while (o < 2 * n && i < n) {
        while (mu > 1.0f && i < n) {
            mu -= 1.0f;
            float r = iq[2 * i], j = iq[2 * i + 1];
            if (r != r) {
                r = 0.0f;
            }
            if (j != j) {
                j = 0.0f;
            }
            t->dl[2 * dli] = r;
            t->dl[2 * dli + 1] = j;
            t->dl[2 * (dli + tw)] = r;
            t->dl[2 * (dli + tw) + 1] = j;
            if (++dli >= tw) {
                dli = 0;
            }
            i++;
        }
        if (i >= n) {
            break;
        }
        const float half_omega = omega / 2.0f;
        int hs = (int)floorf(half_omega);
        float hmu = mu + half_omega - (float)hs;
        if (hmu > 1.0f) {
            hmu -= 1.0f;
            hs += 1;
        }
        if (hs < 0) {
            hs = 0;
        }
        if (dli + 7 >= 2 * tw || dli + hs + 7 >= 2 * tw) {
            mu += omega;
            continue;
        }
        float mr, mj, sr, sj;
        orc_mmse_interp_complex(t->dl + 2 * dli, mu, &mr, &mj);
        orc_mmse_interp_complex(t->dl + 2 * (dli + hs), hmu, &sr, &sj);
        float err = (last_r - sr) * mr + (last_j - sj) * mj;
        if (err != err) {
            err = 0.0f;
        }
        err = clipf(err, 1.0f);
        const float ie2 = sr * sr, io2 = mr * mr, qe2 = sj * sj, qo2 = mj * mj;
        const float yi = ((ie2 + io2) != 0.0f) ? (ie2 - io2) / (ie2 + io2) : 0.0f;
        const float yq = ((qe2 + qo2) != 0.0f) ? (qe2 - qo2) / (qe2 + qo2) : 0.0f;
        lock += yi + yq;
        lock_n++;
        const float mag = sqrtf(sr * sr + sj * sj);
        omega += gain_omega * err * mag;
        omega = t->omega_mid + clipf(omega - t->omega_mid, t->omega_rel);
        mu += omega + gain_mu * err;
        last_r = sr;
        last_j = sj;
        out[o++] = sr;
        out[o++] = sj;
    }
    t->mu = mu;
    t->omega = omega;
    t->dl_index = dli;
    t->last_r = last_r;
    t->last_j = last_j;
    t->lock_accum = lock;
    t->lock_count = lock_n;
    return o >= 2 ? o : 0;
}
