/*
 * oracle/ddn_oracle_sym.c — CPU restatement of the P25 Phase 1 C4FM slicer / soft-decision path and of the
 * per-sample matched filter (TEST INFRASTRUCTURE ONLY).
 *
 *   per-symbol slicer state update   src/core/frames/dsd_dibit.c:194-241 (two-smallest / two-largest window mean),
 *                                    :243-275 (use_symbol), include/dsd-neo/core/state.h:1399-1455 (1024-deep moving
 *                                    average of window extrema kept as double sums)
 *   4-level slice                    src/core/frames/dsd_dibit.c:963-976,1018-1041
 *   soft metrics                     :456-500 (threshold-distance reliability), :506-547 (SNR weight: hooks unset ->
 *                                    -100 dB -> scale 204/256), :609-656 (ideal levels, per-bit magnitude), :688-721
 *   matched filter                   src/dsp/dsd_filters.c:173-200 (circular FIR, oldest-first mul+add), taps measured
 *                                    from the compiled reference (oracle/ddn_tables_p25.h)
 *   slicer reset values              src/dsp/dsd_symbol.c:1306-1326
 *
 * Pinned bit-exact against the reference's own compiled dsd_dibit.c / dsd_filters.c by tests/test_oracle_sym.py.
 */
#include "ddn_oracle.h"

#include <math.h>
#include <string.h>

#include "ddn_tables_p25.h"

void
orc_slicer_init(orc_slicer* s, int negative_polarity) {
    memset(s, 0, sizeof(*s));
    s->negative = negative_polarity ? 1 : 0;
    s->center = 0.0f;
    s->min = -30000.0f;
    s->max = 30000.0f;
    s->lmid = -20000.0f;
    s->umid = 20000.0f;
    s->minref = -24000.0f;
    s->maxref = 24000.0f;
    for (int i = 0; i < ORC_SLICER_MSIZE; i++) {
        s->minbuf[i] = s->min;
        s->maxbuf[i] = s->max;
    }
    s->sums_valid = 0;
}

static void
window_extrema(const float* v, int n, float* lo, float* hi) {
    float a1 = v[0], a2 = v[1], b1 = v[0], b2 = v[1];
    if (a2 < a1) {
        float t = a1;
        a1 = a2;
        a2 = t;
    }
    if (b2 > b1) {
        float t = b1;
        b1 = b2;
        b2 = t;
    }
    for (int i = 2; i < n; i++) {
        const float x = v[i];
        if (x < a1) {
            a2 = a1;
            a1 = x;
        } else if (x < a2) {
            a2 = x;
        }
        if (x > b1) {
            b2 = b1;
            b1 = x;
        } else if (x > b2) {
            b2 = x;
        }
    }
    *lo = (a1 + a2) * 0.5f;
    *hi = (b1 + b2) * 0.5f;
}

static int
clamp255(int v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

static int
bit_magnitude(float sym, const float ideal[4], int bit_index) {
    float best0 = 3.4028234663852886e38f, best1 = 3.4028234663852886e38f, spacing = 3.4028234663852886e38f;
    for (int i = 0; i < 4; i++) {
        const float d = (sym - ideal[i]) * (sym - ideal[i]);
        if ((i >> (1 - bit_index)) & 1) {
            if (d < best1) {
                best1 = d;
            }
        } else if (d < best0) {
            best0 = d;
        }
        for (int j = i + 1; j < 4; j++) {
            const float sp = fabsf(ideal[i] - ideal[j]);
            if (sp > 1e-6f && sp < spacing) {
                spacing = sp;
            }
        }
    }
    if (spacing == 3.4028234663852886e38f) {
        spacing = 2.0f;
    }
    const float scale = 255.0f / (spacing * spacing);
    return clamp255((int)lrintf(fabsf(best0 - best1) * scale));
}

static int
threshold_reliability(const orc_slicer* s, float sym) {
    const float eps = 1e-6f;
    int rel;
    if (sym > s->umid) {
        float span = s->max - s->umid;
        if (span < eps) {
            span = eps;
        }
        rel = (int)lrintf(((sym - s->umid) * 255.0f) / span);
    } else if (sym > s->center) {
        const float d1 = sym - s->center, d2 = s->umid - sym;
        float span = s->umid - s->center;
        if (span < eps) {
            span = eps;
        }
        rel = (int)lrintf(((d1 < d2 ? d1 : d2) * 510.0f) / span);
    } else if (sym >= s->lmid) {
        const float d1 = s->center - sym, d2 = sym - s->lmid;
        float span = s->center - s->lmid;
        if (span < eps) {
            span = eps;
        }
        rel = (int)lrintf(((d1 < d2 ? d1 : d2) * 510.0f) / span);
    } else {
        float span = s->lmid - s->min;
        if (span < eps) {
            span = eps;
        }
        rel = (int)lrintf(((s->lmid - sym) * 255.0f) / span);
    }
    rel = clamp255(rel);
    /* SNR weight with every metrics hook unset: snr = -100 dB -> w256 = 0 -> scale 204/256 */
    return clamp255((rel * 204) >> 8);
}

/* One symbol through get_dibit_and_analog_signal()'s slicer path.  rec4 = {dibit, reliability, llr0, llr1}. */
void
orc_slicer_step(orc_slicer* s, float sym, int rec4[4]) {
    s->sbuf[s->sidx] = sym;
    /* use_symbol(): P25p1 keeps the thresholds live */
    float lo, hi;
    window_extrema(s->sbuf, ORC_SLICER_SSIZE, &lo, &hi);
    if (!s->sums_valid) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < ORC_SLICER_MSIZE; i++) {
            a += (double)s->minbuf[i];
            b += (double)s->maxbuf[i];
        }
        s->min_sum = a;
        s->max_sum = b;
        s->sums_valid = 1;
        if (s->midx < 0 || s->midx >= ORC_SLICER_MSIZE) {
            s->midx = 0;
        }
    }
    s->min_sum += (double)lo - (double)s->minbuf[s->midx];
    s->max_sum += (double)hi - (double)s->maxbuf[s->midx];
    s->minbuf[s->midx] = lo;
    s->maxbuf[s->midx] = hi;
    s->midx = (s->midx + 1 >= ORC_SLICER_MSIZE) ? 0 : s->midx + 1;
    s->min = (float)(s->min_sum / (double)ORC_SLICER_MSIZE);
    s->max = (float)(s->max_sum / (double)ORC_SLICER_MSIZE);
    s->center = (s->max + s->min) / 2.0f;
    s->umid = ((s->max - s->center) * 5.0f / 8.0f) + s->center;
    s->lmid = ((s->min - s->center) * 5.0f / 8.0f) + s->center;
    s->maxref = s->max * 0.80f;
    s->minref = s->min * 0.80f;
    s->sidx = (s->sidx >= ORC_SLICER_SSIZE - 1) ? 0 : s->sidx + 1;

    orc_slicer_digitize(s, sym, rec4);
}

/* digitize() + compute_dibit_soft_metric() with the thresholds as they stand (dsd_dibit.c:963-1043,690-721) */
void
orc_slicer_digitize(const orc_slicer* s, float sym, int rec4[4]) {
    const int neg = s->negative;
    int dibit;
    if (sym > s->center) {
        dibit = (sym > s->umid) ? (neg ? 3 : 1) : (neg ? 2 : 0);
    } else {
        dibit = (sym < s->lmid) ? (neg ? 1 : 3) : (neg ? 0 : 2);
    }
    const float plus_one = 0.5f * (s->center + s->umid), minus_one = 0.5f * (s->lmid + s->center);
    float ideal[4];
    if (neg) {
        ideal[0] = minus_one;
        ideal[1] = s->min;
        ideal[2] = plus_one;
        ideal[3] = s->max;
    } else {
        ideal[0] = plus_one;
        ideal[1] = s->max;
        ideal[2] = minus_one;
        ideal[3] = s->min;
    }
    int mag0 = bit_magnitude(sym, ideal, 0), mag1 = bit_magnitude(sym, ideal, 1);
    const int rel = threshold_reliability(s, sym);
    const int mn = mag0 < mag1 ? mag0 : mag1;
    if (mn > 0 && rel < mn) {
        mag0 = (mag0 * rel) / mn;
        mag1 = (mag1 * rel) / mn;
    }
    const int l0 = ((dibit >> 1) & 1) ? clamp255(mag0) : -clamp255(mag0);
    const int l1 = (dibit & 1) ? clamp255(mag1) : -clamp255(mag1);
    const int a0 = l0 < 0 ? -l0 : l0, a1 = l1 < 0 ? -l1 : l1;
    rec4[0] = dibit;
    rec4[1] = clamp255(a1 < a0 ? a1 : a0);
    rec4[2] = l0;
    rec4[3] = l1;
}


/* dmr_compute_reliability() for rf_mod != 1 (dsd_dibit.c:548-568) */
int
orc_slicer_reliability(const orc_slicer* s, float sym) {
    return threshold_reliability(s, sym);
}

/* get_dibit_and_analog_signal() when use_symbol() takes its "no continuous update" branch (dsd_dibit.c:264-276: any
 * protocol but P25p1 on a C4FM / GFSK stream): the window slot is written, the crossing references follow max / min,
 * the thresholds stay as the last sync left them. */
void
orc_slicer_step_static(orc_slicer* s, float sym, int rec4[4]) {
    s->sbuf[s->sidx] = sym;
    s->maxref = s->max;
    s->minref = s->min;
    s->sidx = (s->sidx >= ORC_SLICER_SSIZE - 1) ? 0 : s->sidx + 1;
    orc_slicer_digitize(s, sym, rec4);
}

void
orc_slicer_run(orc_slicer* s, const float* sym, long n, int* rec4, float* thr5) {
    for (long i = 0; i < n; i++) {
        orc_slicer_step(s, sym[i], rec4 + 4 * i);
        if (thr5) {
            thr5[5 * i] = s->center;
            thr5[5 * i + 1] = s->umid;
            thr5[5 * i + 2] = s->lmid;
            thr5[5 * i + 3] = s->max;
            thr5[5 * i + 4] = s->min;
        }
    }
}

/* P25 matched filter at sps = 10: y[n] = sum_{i=0..90} taps[i] * x[n-90+i], products added oldest-first, history
 * zero at start.  hist carries the last 90 inputs between calls. */
void
orc_p25_filter_run(float hist[90], const float* in, long n, float* out) {
    float taps[DDN_P25_FILTER_TAPS];
    memcpy(taps, ddn_p25_filter_bits, sizeof(taps));
    for (long k = 0; k < n; k++) {
        float acc = 0.0f;
        for (int i = 0; i < DDN_P25_FILTER_TAPS; i++) {
            const long j = k - 90 + i;
            const float x = (j >= 0) ? in[j] : hist[90 + j];
            acc += taps[i] * x;
        }
        out[k] = acc;
    }
    float nh[90];
    for (int i = 0; i < 90; i++) {
        const long j = n - 90 + i;
        nh[i] = (j >= 0) ? in[j] : hist[90 + j];
    }
    memcpy(hist, nh, sizeof(nh));
}
