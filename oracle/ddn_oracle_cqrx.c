/*
 * oracle/ddn_oracle_cqrx.c - CPU restatement of the SYMBOL-RATE receive loop behind the CQPSK demodulator (TEST INFRASTRUCTURE
 * ONLY): full_demod()'s CQPSK branch hands the consumer one float per symbol (levels +-1 / +-3); this is what the consumer thread does
 * with them for P25 Phase 1 (CQPSK / LSM) and P25 Phase 2 until they are dibits + soft decisions in capture records.
 *
 *   getSymbol(), symbol-rate fast path   src/dsp/dsd_symbol.c:1581-1624 (symbol_try_rtl_symbol_rate_fast_path): samplesPerSymbol 1,
 *                                        jitter -1, apply_rtl_symbol_thresholds() :744-765 EVERY call (centre 0, max / min +-3,
 *                                        umid / lmid +-2, maxref / minref +-2.4), the symbol straight from the demodulator's output,
 *                                        history push - no matched filter (:302-306), no window, no clip
 *   getFrameSync() hunting loop          src/dsp/dsd_frame_sync.c:3098-3148; ring update :1746-1764 (level ring of t_max = 24, 19 for
 *                                        the Phase 2 profile on a QPSK stream :1729-1744; the slicer window slot is written too);
 *                                        4-level slice against the centre :2061-2075,2110-2116 (frame_sync_cqpsk_4level_enabled
 *                                        :2043-2058: rf_mod 1, RTL input, P25 enabled, the demodulator reports CQPSK + timing);
 *                                        level window :2319-2336 - on a QPSK profile it feeds the 1024-deep extrema average
 *                                        (dsd_state_push_minmax_window) and moves centre / maxref / minref; exact sync compare
 *                                        :698-716 (24 dibits) / :801-816 (20 dibits); rotated-constellation retries :668-696 with
 *                                        the maps of include/dsd-neo/core/p25_cqpsk_dibit.h (X2400, N1200, P1200, in that order, the
 *                                        positive pattern before the inverted one), the raw-level fit :417-500 and its application
 *                                        :520-548 (seeds the extrema average and the slicer window); accept :385-392, :603-625,
 *                                        :639-666; no-sync timeout :2753-2760, :3037-3053
 *   in-frame symbol                      src/core/frames/dsd_dibit.c:1043-1075 (window slot, use_symbol, digitize):
 *                                        use_symbol :243-275 (QPSK: two-smallest / two-largest mean of the 128-symbol window into the
 *                                        extrema average every symbol; centre, umid, lmid, references follow), CQPSK slice
 *                                        :330-350,951-1000 (fixed +-2 around the centre, then the rotation map, then the polarity
 *                                        inversion), soft metrics :376-405 (distance to the ideal level), :408-430 (SNR weight),
 *                                        :548-556, :609-641 (per-bit magnitude), :658-721 (ideals = centre + the level the map sends to
 *                                        each dibit)
 *   what reads in frame                  P25 Phase 1: the per-DUID handlers (oracle/ddn_oracle_handlers.c); Phase 2: p2_dibit_buffer()
 *                                        src/protocol/p25/phase2/p25p2_frame.c:352-370 - 700 dibits per sync
 *
 * PARITY STATUS.  The in-frame path (use_symbol + digitize + soft metrics with the CQPSK hooks reporting active) is pinned against the
 * reference's compiled dsd_dibit.c (oracle/_ref, tests/test_oracle_cqrx.py); the level estimator was pinned before
 * (frame_sync_level.c).  dsd_symbol.c and dsd_frame_sync.c cannot be compiled here (<sndfile.h>, the protocol tree): the fast path, the
 * hunting loop, the rotated retries and the raw fit are restated from the lines above, PARITY UNPINNED, anchored on the reference's
 * three P25 Phase 1 CQPSK captures and its Phase 2 capture (known answers of tests/CMakeLists.txt:8901-8923).  Deliberate limits:
 *   - modulation locked to QPSK (as -mq does): no C4FM / QPSK / GFSK voting (frame_sync_maybe_auto_switch_modulation);
 *   - one protocol per stream (-f1 or -f2): the other matchers of frame_sync_try_protocol_matches never fire;
 *   - the CQPSK SNR the reliability weight reads (dsd_rtl_stream_metrics_hook_snr_cqpsk_db) is an asynchronous estimate of the
 *     radio thread (src/io/radio/rtl_sdr_fm.cpp:2752,3088), not a function of the symbol stream: it is a caller-given constant
 *     here, default -100 dB = "not available", which leaves the reliability unweighted (dsd_dibit.c:411-413);
 *   - debug switches (cqpsk_sync_inv / _neg) off, warm start enabled (their defaults).
 */
#include "ddn_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

static const uint8_t k_map[5][4] = {{0, 1, 2, 3}, {2, 3, 0, 1}, {3, 2, 1, 0}, {1, 3, 0, 2}, {2, 0, 3, 1}};

static int
invert_dibit(int d) {
    return d ^ 2; /* 0<->2, 1<->3 */
}

static int
cq_slice(float s) {
    return s >= 2.0f ? 1 : (s >= 0.0f ? 0 : (s >= -2.0f ? 2 : 3));
}

static void
fast_path_thresholds(orc_slicer* s) { /* apply_rtl_symbol_thresholds(state, 4) */
    s->center = 0.0f;
    s->min = -3.0f;
    s->max = 3.0f;
    s->lmid = -2.0f;
    s->umid = 2.0f;
    s->minref = -2.4f;
    s->maxref = 2.4f;
}

/* dsd_state_push_minmax_window(state, 1024, lo, hi): state->min / max = the running means */
static void
push_minmax(orc_slicer* s, float lo, float hi) {
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
}

static void
window_extrema128(const float* v, float* lo, float* hi) {
    float a1 = v[0], a2 = v[1], b1 = v[0], b2 = v[1];
    if (a2 < a1) {
        const float t = a1;
        a1 = a2;
        a2 = t;
    }
    if (b2 > b1) {
        const float t = b1;
        b1 = b2;
        b2 = t;
    }
    for (int i = 2; i < ORC_SLICER_SSIZE; i++) {
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
bit_magnitude(float sym, const float ideal[4], int bit_index) { /* soft_metric_for_bit() */
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

/* dmr_compute_reliability() for rf_mod == 1: distance to the ideal level + the SNR weight */
int
orc_cq_reliability(float sym_c, double snr_db) {
    const float ideal = sym_c >= 2.0f ? 3.0f : (sym_c >= 0.0f ? 1.0f : (sym_c >= -2.0f ? -1.0f : -3.0f));
    float err = fabsf(sym_c - ideal);
    if (err > 1.0f) {
        err = 1.0f;
    }
    int rel = clamp255((int)((1.0f - err) * 255.0f + 0.5f));
    if (snr_db <= -50.0) {
        return rel;
    }
    int w256 = 0;
    if (snr_db >= 25.0) {
        w256 = 255;
    } else if (snr_db > 0.0) {
        w256 = (int)((snr_db / 25.0) * 255.0 + 0.5);
    }
    return clamp255((rel * (204 + (w256 >> 2))) >> 8);
}

/* digitize() + compute_dibit_soft_metric() on the CQPSK slice, thresholds as they stand.  rec4 = {dibit, reliability, llr0, llr1} */
void
orc_cq_digitize(const orc_slicer* s, float sym, int map_idx, int negative, double snr_db, int rec4[4]) {
    int dibit = k_map[map_idx][cq_slice(sym - s->center)];
    if (negative) {
        dibit = invert_dibit(dibit);
    }
    /* build_cqpsk_dibit_ideals(): the level whose raw dibit the map sends to (the polarity-corrected) dibit d */
    static const float base_ideal[4] = {1.0f, 3.0f, -1.0f, -3.0f};
    float ideal[4];
    for (int d = 0; d < 4; d++) {
        const int corrected = negative ? invert_dibit(d) : d;
        int raw = corrected;
        for (int q = 0; q < 4; q++) {
            if (k_map[map_idx][q] == corrected) {
                raw = q;
                break;
            }
        }
        ideal[d] = s->center + base_ideal[raw];
    }
    int mag0 = bit_magnitude(sym, ideal, 0), mag1 = bit_magnitude(sym, ideal, 1);
    const int rel = orc_cq_reliability(sym - s->center, snr_db);
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

/* one in-frame symbol: get_dibit_and_analog_signal() behind the fast path */
void
orc_cq_inframe_step(orc_slicer* s, float sym, int map_idx, int negative, double snr_db, int rec4[4]) {
    fast_path_thresholds(s);
    s->sbuf[s->sidx] = sym;
    float lo, hi;
    window_extrema128(s->sbuf, &lo, &hi);
    push_minmax(s, lo, hi);
    s->center = (s->max + s->min) / 2.0f;
    s->umid = ((s->max - s->center) * 5.0f / 8.0f) + s->center;
    s->lmid = ((s->min - s->center) * 5.0f / 8.0f) + s->center;
    s->maxref = s->max * 0.80f;
    s->minref = s->min * 0.80f;
    s->sidx = (s->sidx >= ORC_SLICER_SSIZE - 1) ? 0 : s->sidx + 1;
    orc_cq_digitize(s, sym, map_idx, negative, snr_db, rec4);
}

void
orc_cqrx_init(orc_cqrx* r, int protocol, int lock_symbols, double snr_db) {
    memset(r, 0, sizeof(*r));
    r->protocol = protocol;
    r->sync_len = protocol == ORC_CQ_P25P2 ? 20 : 24;
    r->t_max = protocol == ORC_CQ_P25P2 ? 19 : 24;
    r->lock_symbols = protocol == ORC_CQ_P25P2 ? (lock_symbols > 0 ? lock_symbols : 700) : lock_symbols;
    r->snr_db = snr_db;
    /* initState(): src/core/util/dsd_init.c:519-539 */
    orc_slicer* s = &r->sl;
    s->center = 0.0f;
    s->min = -15000.0f;
    s->max = 15000.0f;
    s->lmid = 0.0f;
    s->umid = 0.0f;
    for (int i = 0; i < ORC_SLICER_MSIZE; i++) {
        s->minbuf[i] = -15000.0f;
        s->maxbuf[i] = 15000.0f;
    }
    r->lmin = s->min;
    r->lmax = s->max;
    orc_p25h_init(&r->h, 64);
    /* the sync words as raw-dibit strings under each map: raw window == inverse map of the pattern */
    static const char* pat[2][2] = {{"111113113311333313133333", "333331331133111131311111"}, {"11131131111333133333", "33313313333111311111"}};
    const int order[4] = {0, 2, 3, 4}; /* identity, then X2400, N1200, P1200 */
    for (int pol = 0; pol < 2; pol++) {
        const char* p = pat[protocol == ORC_CQ_P25P2][pol];
        for (int m = 0; m < 4; m++) {
            uint64_t w = 0;
            for (int i = 0; i < r->sync_len; i++) {
                const int want = p[i] - '0';
                int raw = want;
                for (int q = 0; q < 4; q++) {
                    if (k_map[order[m]][q] == want) {
                        raw = q;
                    }
                }
                w = (w << 2) | (uint64_t)raw;
            }
            r->target[pol][m] = w;
        }
    }
}

void
orc_cqrx_set_events(orc_cqrx* r, orc_hevents* ev) {
    r->ev = ev;
    if (ev) {
        ev->n = 0;
    }
}

static void
hunt_enter(orc_cqrx* r) { /* a fresh getFrameSync(): frame_sync_runtime_init() */
    r->hunt_pos = 0;
    r->have_sync = 0;
    r->lidx = 0;
    r->level_count = 0;
    r->hist_count = 0;
    r->hist = 0;
    r->lmin = r->sl.min;
    r->lmax = r->sl.max;
}

static void
no_carrier(orc_cqrx* r) { /* engine.c:1836-1851 as far as this loop sees it */
    r->lastsync = 0;
    r->sl.max = 15000.0f;
    r->sl.min = -15000.0f;
    r->sl.center = 0.0f;
    orc_p25h_no_carrier(&r->h);
}

static void
sort_small(float* v, int n) {
    for (int i = 1; i < n; i++) {
        const float x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) {
            v[j + 1] = v[j];
            j--;
        }
        v[j + 1] = x;
    }
}

/* frame_sync_fit_p25_cqpsk_raw_sync(): the sync symbols' mean level per raw dibit against the units {+1, +3, -1, -3}; 1 = fit found */
static int
raw_fit(const orc_cqrx* r, float* centre, float* gain) {
    static const int unit[4] = {1, 3, -1, -3};
    float sum[4] = {0, 0, 0, 0};
    int cnt[4] = {0, 0, 0, 0};
    if (r->scount < r->sync_len) {
        return 0;
    }
    for (int i = 0; i < r->sync_len; i++) {
        const int raw = (int)((r->hist >> (2 * (r->sync_len - 1 - i))) & 3u);
        const int back = r->sync_len - 1 - i;
        sum[raw] += r->shist[(r->shead - 1 - back + 2 * 24) % 24];
        cnt[raw]++;
    }
    float sx = 0.0f, sy = 0.0f, sxx = 0.0f, sxy = 0.0f;
    int n = 0;
    for (int q = 0; q < 4; q++) {
        if (!cnt[q]) {
            continue;
        }
        const float x = (float)unit[q], y = sum[q] / (float)cnt[q];
        sx += x;
        sy += y;
        sxx += x * x;
        sxy += x * y;
        n++;
    }
    if (n < 2) {
        return 0;
    }
    const float den = ((float)n * sxx) - (sx * sx);
    if (fabsf(den) < 1.0e-6f) {
        return 0;
    }
    const float g = (((float)n * sxy) - (sx * sy)) / den;
    if (fabsf(g) * 2.0f < 1.0f) {
        return 0;
    }
    *centre = (sy - (g * sx)) / (float)n;
    *gain = g;
    return 1;
}

/* One symbol of the demodulator's output through the loop.  rec4 = {dibit, reliability, llr0, llr1}; returns the flag bits:
 * 1 = in frame, 2 = a sync completed on this symbol, 4 = negative polarity, and (map index) << 4 with a completed sync */
int
orc_cqrx_symbol(orc_cqrx* r, float sym, int rec4[4]) {
    orc_slicer* s = &r->sl;
    r->shist[r->shead] = sym;
    r->shead = (r->shead + 1) % 24;
    if (r->scount < 24) {
        r->scount++;
    }
    if (r->have_sync) {
        const int neg = r->lastsync == 2;
        orc_cq_inframe_step(s, sym, r->map_idx, neg, r->snr_db, rec4);
        const int flags = 1 | (neg ? 4 : 0);
        if (r->lock_symbols < 0) {
            if (!orc_p25h_symbol(&r->h, r->n_sym, rec4[0], rec4[2], rec4[3], r->ev)) {
                hunt_enter(r);
            }
        } else if (--r->lock_left <= 0) {
            hunt_enter(r);
        }
        r->n_sym++;
        return flags;
    }
    /* hunting */
    fast_path_thresholds(s);
    r->lbuf[r->lidx] = sym;
    if (r->level_count < r->t_max) {
        r->level_count++;
    }
    s->sbuf[s->sidx] = sym;
    r->lidx = (r->lidx == r->t_max - 1) ? 0 : r->lidx + 1;
    s->sidx = (s->sidx == ORC_SLICER_SSIZE - 1) ? 0 : s->sidx + 1;
    const int raw = cq_slice(sym - s->center);
    const uint64_t mask = (r->sync_len == 24) ? 0xFFFFFFFFFFFFull : 0xFFFFFFFFFFull;
    r->hist = ((r->hist << 2) | (uint64_t)raw) & mask;
    if (r->hist_count < 24) {
        r->hist_count++;
    }
    rec4[0] = raw;
    rec4[1] = 0;
    rec4[2] = 0;
    rec4[3] = 0;
    int flags = 0;
    if (r->hist_count >= 8) {
        float tmp[24];
        memcpy(tmp, r->lbuf, sizeof(float) * (size_t)r->level_count);
        sort_small(tmp, r->level_count);
        orc_level_estimate(tmp, r->level_count, &r->lmin, &r->lmax);
        push_minmax(s, r->lmin, r->lmax); /* QPSK profile: the hunting levels feed the extrema average too */
        s->center = (s->max + s->min) / 2.0f;
        s->maxref = s->max * 0.80f;
        s->minref = s->min * 0.80f;
        if (r->hist_count >= r->sync_len) {
            int pol = 0, m = -1;
            /* exact + then - ; rotated (X2400, N1200, P1200) + then - */
            if (r->hist == r->target[0][0]) {
                pol = 1, m = 0;
            } else if (r->hist == r->target[1][0]) {
                pol = 2, m = 0;
            } else {
                for (int p = 0; p < 2 && m < 0; p++) {
                    for (int k = 1; k < 4; k++) {
                        if (r->hist == r->target[p][k]) {
                            pol = p + 1;
                            m = k;
                            break;
                        }
                    }
                }
            }
            float fc = 0.0f, fg = 0.0f;
            int fit = 0;
            if (m > 0) {
                fit = raw_fit(r, &fc, &fg);
                if ((m == 2 || m == 3) && !fit) { /* N1200 / P1200 need the centre fit */
                    pol = 0;
                }
            }
            if (pol) {
                static const int map_of[4] = {0, 2, 3, 4};
                r->map_idx = map_of[m];
                s->max = (s->max + r->lmax) / 2;
                s->min = (s->min + r->lmin) / 2;
                r->lastsync = pol;
                if (m > 0 && fit) { /* frame_sync_apply_p25_cqpsk_raw_fit() */
                    const float half = fabsf(fg) * 3.0f;
                    s->center = fc;
                    s->min = fc - half;
                    s->max = fc + half;
                    s->umid = s->center + (s->max - s->center) * 0.625f;
                    s->lmid = s->center + (s->min - s->center) * 0.625f;
                    s->maxref = s->max * 0.80f;
                    s->minref = s->min * 0.80f;
                    for (int i = 0; i < ORC_SLICER_MSIZE; i++) {
                        s->minbuf[i] = s->min;
                        s->maxbuf[i] = s->max;
                    }
                    s->sums_valid = 0;
                    for (int i = 0; i < ORC_SLICER_SSIZE; i++) {
                        s->sbuf[i] = (i & 1) ? s->max : s->min;
                    }
                }
                /* (m == 0: dsd_sync_warm_start_center_outer_only() moves the centre only, and the next getSymbol() resets it) */
                r->have_sync = 1;
                r->lock_left = r->lock_symbols;
                flags |= 2 | (pol == 2 ? 4 : 0) | (r->map_idx << 4);
                if (r->lock_symbols < 0) {
                    (void)orc_p25h_begin(&r->h);
                } else if (r->lock_left <= 0) {
                    hunt_enter(r);
                }
                r->n_sym++;
                return flags;
            }
        }
    }
    if (r->hunt_pos < 10200) {
        r->hunt_pos++;
    } else {
        r->hunt_pos = 0;
        no_carrier(r);
    }
    if (!(r->protocol == ORC_CQ_P25P1 && r->lastsync == 2) && r->hunt_pos >= 1800) {
        no_carrier(r);
        hunt_enter(r);
    }
    r->n_sym++;
    return flags;
}

long
orc_cqrx_run(orc_cqrx* r, const float* sym, long n, int* rec4, uint8_t* flags) {
    for (long k = 0; k < n; k++) {
        flags[k] = (uint8_t)orc_cqrx_symbol(r, sym[k], rec4 + 4 * k);
    }
    return n;
}

size_t
orc_cqrx_sizeof(void) {
    return sizeof(orc_cqrx);
}

size_t
orc_cqrx_slicer_offset(void) { /* where the slicer words sit inside the loop's state (the tests drive it alone) */
    return offsetof(orc_cqrx, sl);
}

void
orc_cqrx_get_state(const orc_cqrx* r, float out8[8]) {
    out8[0] = r->sl.center;
    out8[1] = r->sl.max;
    out8[2] = r->sl.min;
    out8[3] = (float)r->map_idx;
    out8[4] = (float)r->lastsync;
    out8[5] = (float)r->have_sync;
    out8[6] = (float)r->hunt_pos;
    out8[7] = (float)r->sl.midx;
}
