/*
 * oracle/ddn_oracle_resamp.c — CPU restatement of the rational L/M polyphase resampler
 * (TEST INFRASTRUCTURE ONLY; binary32 arithmetic in the reference's operation order).
 *
 *   prototype design   src/dsp/resampler.cpp:166-190 (Hamming-windowed sinc at fc = 0.45 / max(L, M), evaluated in
 *                      binary64, normalised to unity DC gain, scaled by L, stored per phase oldest tap first),
 *                      16 taps per phase :253-260
 *   block processing   src/dsp/resampler.cpp:205-233,318-356: push one input, then emit one output per polyphase
 *                      index phase, phase + M, ... < L; phase -= L
 *   dot product        src/dsp/resampler.cpp:84-99: four interleaved partial sums over the 16 taps, combined as
 *                      (a0 + a1) + (a2 + a3); products and sums are separate roundings (the file is built without FMA)
 *
 * Pinned by tests/test_oracle_resamp.py against the compiled reference (dsd_resampler_design /
 * dsd_resampler_process_block of oracle/_ref).
 */
#include "ddn_oracle.h"

#include <math.h>
#include <string.h>

#define RS_K 16

static double
rs_sinc(double x) {
    const double pi = 3.14159265358979323846;
    if (x == 0.0) {
        return 1.0;
    }
    return sin(pi * x) / (pi * x);
}

int
orc_resamp_design(int L, int M, float* taps) {
    const double pi = 3.14159265358979323846;
    if (L < 1 || M < 1 || L > ORC_RESAMP_MAX_L) {
        return -1;
    }
    const int total = RS_K * L;
    const double fc = 0.45 / (double)((L > M) ? L : M);
    const int mid = (total - 1) / 2;
    double gain = 0.0;
    for (int n = 0; n < total; n++) {
        const int m = n - mid;
        const double w = 0.54 - 0.46 * cos(2.0 * pi * (double)n / (double)(total - 1));
        const double h = 2.0 * fc * rs_sinc(2.0 * fc * (double)m);
        gain += h * w;
    }
    if (gain == 0.0) {
        gain = 1.0;
    }
    for (int phase = 0; phase < L; phase++) {
        for (int k = 0; k < RS_K; k++) {
            const int src = phase + ((RS_K - 1 - k) * L);
            const int m = src - mid;
            const double w = 0.54 - 0.46 * cos(2.0 * pi * (double)src / (double)(total - 1));
            const double h = 2.0 * fc * rs_sinc(2.0 * fc * (double)m);
            taps[phase * RS_K + k] = (float)((h * w / gain) * (double)L);
        }
    }
    return total;
}

size_t
orc_resamp_sizeof(void) {
    return sizeof(orc_resamp);
}

int
orc_resamp_init(orc_resamp* r, int L, int M) {
    memset(r, 0, sizeof(*r));
    r->L = L;
    r->M = M;
    return orc_resamp_design(L, M, r->taps);
}

long
orc_resamp_run(orc_resamp* r, const float* in, long n, float* out, long cap) {
    long o = 0;
    for (long i = 0; i < n; i++) {
        /* window = the 16 most recent inputs, oldest first */
        memmove(r->win, r->win + 1, sizeof(float) * (RS_K - 1));
        r->win[RS_K - 1] = in[i];
        while (r->phase < r->L) {
            const float* t = r->taps + (size_t)r->phase * RS_K;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            for (int k = 0; k < RS_K; k += 4) {
                a0 += r->win[k + 0] * t[k + 0];
                a1 += r->win[k + 1] * t[k + 1];
                a2 += r->win[k + 2] * t[k + 2];
                a3 += r->win[k + 3] * t[k + 3];
            }
            if (o >= cap) {
                return -1;
            }
            out[o++] = (a0 + a1) + (a2 + a3);
            r->phase += r->M;
        }
        r->phase -= r->L;
    }
    return o;
}
