/*
 * oracle/ddn_oracle_rx.c — CPU restatement of the fixed-protocol P25 Phase 1 C4FM receive loop between the
 * discriminator stream and the capture records (TEST INFRASTRUCTURE ONLY): symbol extraction with jitter timing,
 * frame-sync hunting, threshold warm start, and the in-frame slicer.
 *
 *   getSymbol() RTL-FSK path        src/dsp/dsd_symbol.c:1343-1387 (whole + fractional samples/symbol accumulator),
 *                                   :197-211 (C4FM window l = r = 2), :347-358 (in-sync clip), :360-397 (jitter =
 *                                   first centre crossing), :436-460 (window accumulation), :489-517 (one-sample slip
 *                                   at i == 0 while hunting), :1769-1805 (per-sample order: matched filter, clip,
 *                                   jitter, accumulate; symbol = sum / count), :1839-1851 (history push)
 *   matched-filter gating           src/dsp/dsd_symbol.c:301-338 (only once lastsynctype is P25p1)
 *   getFrameSync() hunting loop     src/dsp/dsd_frame_sync.c:3098-3148; ring update :1747-1764; sign-only dibit
 *                                   :2110-2127; level window :2316-2336 + src/dsp/frame_sync_level.c:10-44; pattern
 *                                   compare :698-716 (include/dsd-neo/core/sync_patterns.h:33-34); accept :385-392,
 *                                   :603-625
 *   threshold warm start            src/dsp/sync_calibration.c:156-233
 *   in-frame symbol                 src/core/frames/dsd_dibit.c (orc_slicer_step, oracle/ddn_oracle_sym.c)
 *
 * PARITY STATUS.  The level estimator, the warm start and the slicer are pinned against the reference's compiled
 * frame_sync_level.c / sync_calibration.c / dsd_dibit.c (tests/test_oracle_rx.py).  dsd_symbol.c and
 * dsd_frame_sync.c cannot be compiled here (they include <sndfile.h> and the whole protocol tree), so the sample loop
 * and the hunting loop are restated from the source lines above with PARITY UNPINNED.  Deliberate simplifications,
 * each a documented deviation from a full dsd-neo run:
 *   - one protocol (P25p1), modulation locked to C4FM (opts->mod_cli_lock), so no modulation voting / SPS hunting;
 *   - after a sync the loop stays in-frame for a caller-given number of symbols (`lock_symbols`) instead of running
 *     the per-DUID handlers; the reference's handlers decide that count frame by frame;
 * Carrier loss IS modelled: after 1800 symbols of one hunt without a sync (10200 when the last sync was the inverted
 * pattern) noCarrier() runs - src/dsp/dsd_frame_sync.c:2753-2760,3037-3053, src/engine/engine.c:1838-1858 - which for this
 * loop means: crossing latch cleared, lastsynctype NONE (matched filter off, its memory kept as it stands), timing ratio
 * zeroed so that the next getSymbol() re-initialises the accumulator and the slicer (dsd_symbol.c:1306-1341).
 */
#include "ddn_oracle.h"

#include <string.h>

#include "ddn_tables_p25.h"

/* P25P1_SYNC "111113113311333313133333" as sign bits (dibit 1 -> 1, dibit 3 -> 0), oldest symbol in bit 23 */
#define P25_SYNC_BITS 0xFB30A0u /* 1111 1011 0011 0000 1010 0000 */

void
orc_level_estimate(const float* sorted, int count, float* lo, float* hi) {
    if (count <= 0) {
        *lo = 0.0f;
        *hi = 0.0f;
        return;
    }
    if (count < 3) {
        float sum = 0.0f;
        for (int i = 0; i < count; i++) {
            sum += sorted[i];
        }
        *lo = *hi = sum / (float)count;
        return;
    }
    int a = 0, b = count - 3;
    if (count >= 13) {
        a = 2;
        b = count - 5;
    }
    *lo = (sorted[a] + sorted[a + 1] + sorted[a + 2]) / 3.0f;
    *hi = (sorted[b] + sorted[b + 1] + sorted[b + 2]) / 3.0f;
}

/* dsd_sync_warm_start_thresholds_outer_only(opts, state, 24): `newest_first[i]` = symbol i steps back.
 * Returns the reference's result code (0 = applied, 3 = degenerate). */
int
orc_slicer_warm_start(orc_slicer* s, const float* newest_first, int sync_len) {
    float sp = 0.0f, sn = 0.0f;
    int np = 0, nn = 0;
    for (int i = 0; i < sync_len; i++) {
        const float v = newest_first[i];
        if (v > 0.0f) {
            sp += v;
            np++;
        } else {
            sn += v;
            nn++;
        }
    }
    if (np == 0 || nn == 0) {
        return 3;
    }
    const float mp = sp / (float)np, mn = sn / (float)nn;
    const float span = mp - mn;
    if ((span < 0.0f ? -span : span) < 1.0f) {
        return 3;
    }
    s->max = mp;
    s->min = mn;
    s->center = (s->max + s->min) / 2.0f;
    s->umid = s->center + (s->max - s->center) * 0.625f;
    s->lmid = s->center + (s->min - s->center) * 0.625f;
    s->maxref = s->max * 0.80f;
    s->minref = s->min * 0.80f;
    for (int i = 0; i < ORC_SLICER_MSIZE; i++) {
        s->maxbuf[i] = s->max;
        s->minbuf[i] = s->min;
    }
    s->sums_valid = 0;
    return 0;
}

void
orc_p25rx_init(orc_p25rx* r, int out_rate_hz, int sym_rate_hz, int lock_symbols, int use_matched_filter) {
    memset(r, 0, sizeof(*r));
    r->out_rate = out_rate_hz;
    r->sym_rate = sym_rate_hz;
    r->lock_symbols = lock_symbols;
    r->use_filter = use_matched_filter;
    r->jitter = -1;
    orc_slicer_init(&r->sl, 0);
    r->lmin = r->sl.min;
    r->lmax = r->sl.max;
    orc_p25h_init(&r->h, 64); /* p25p1_get_erasure_threshold() default */
}

void
orc_p25rx_set_events(orc_p25rx* r, orc_hevents* ev) {
    r->ev = ev;
    if (ev) {
        ev->n = 0;
    }
}

/* noCarrier() as far as this loop can see it (engine.c:1838-1847) */
static void
no_carrier(orc_p25rx* r) {
    r->jitter = -1;
    r->lastsync = 0;
    r->filter_on = 0; /* the filter is gated by lastsynctype (dsd_symbol.c:301-338); its memory is not touched */
    r->sl.max = 15000.0f;
    r->sl.min = -15000.0f;
    r->sl.center = 0.0f;
    r->need_reset = 1; /* rtl_fsk_sps_num / _den = 0 */
    orc_p25h_no_carrier(&r->h);
}

static void
hunt_enter(orc_p25rx* r) {
    /* a fresh getFrameSync() call: frame_sync_runtime_init() */
    r->hunt_pos = 0;
    r->have_sync = 0;
    r->lidx = 0;
    r->level_count = 0;
    r->hist_count = 0;
    r->hist_bits = 0;
    r->lmin = r->sl.min;
    r->lmax = r->sl.max;
}

static float
matched_filter(orc_p25rx* r, float x) {
    /* p25_filter(): shift in, products added oldest first (mul and add rounded separately) */
    float taps[DDN_P25_FILTER_TAPS];
    memcpy(taps, ddn_p25_filter_bits, sizeof(taps));
    float acc = 0.0f;
    for (int i = 0; i < DDN_P25_FILTER_TAPS - 1; i++) {
        acc += taps[i] * r->fhist[i];
    }
    acc += taps[DDN_P25_FILTER_TAPS - 1] * x;
    memmove(r->fhist, r->fhist + 1, sizeof(float) * (DDN_P25_FILTER_TAPS - 2));
    r->fhist[DDN_P25_FILTER_TAPS - 2] = x;
    return acc;
}

static void
symbol_begin(orc_p25rx* r) {
    if (r->need_reset) { /* symbol_reset_rtl_fsk_timing_if_needed() + symbol_reset_rtl_fsk_discriminator_slicer() */
        orc_slicer* s = &r->sl;
        r->need_reset = 0;
        r->sps_accum = 0;
        r->jitter = -1;
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
        s->midx = 0;
        s->sums_valid = 0;
    }
    int whole = r->out_rate / r->sym_rate, rem = r->out_rate % r->sym_rate;
    if (whole < 2) {
        whole = 2;
        rem = 0;
    }
    if (whole > 64) {
        whole = 64;
        rem = 0;
    }
    int sps = whole;
    if (rem > 0) {
        int acc = r->sps_accum + rem;
        if (acc >= r->sym_rate) {
            sps++;
            acc -= r->sym_rate;
        }
        r->sps_accum = acc;
        if (sps > 64) {
            sps = 64;
        }
    }
    r->span = sps;
    r->centre = (sps - 1) / 2;
    r->i = 0;
    r->sum = 0.0f;
    r->count = 0;
    r->in_symbol = 1;
    /* symbol_adjust_timing_index(): only at i == 0, only while hunting */
    if (sps > 1 && r->have_sync == 0 && r->jitter >= 0) {
        if (r->jitter > 0 && r->jitter <= r->centre) {
            r->i--;
        } else if (r->jitter > r->centre && r->jitter < sps) {
            r->i++;
        }
        r->jitter = -1;
    }
}

static void
sample_step(orc_p25rx* r, float x) {
    orc_slicer* s = &r->sl;
    if (r->filter_on) {
        x = matched_filter(r, x);
    }
    if (r->have_sync) {
        if (x > s->max) {
            x = s->max;
        } else if (x < s->min) {
            x = s->min;
        }
    }
    const int i = r->i;
    if (x > s->center) {
        if (!(x > s->maxref * 1.25f)) {
            if (r->jitter < 0 && r->lastsample < s->center) {
                r->jitter = i;
            }
        }
    } else {
        if (!(x < s->minref * 1.25f)) {
            if (r->jitter < 0 && r->lastsample > s->center) {
                r->jitter = i;
            }
        }
    }
    /* symbol_accumulate_sample(), rf_mod == 0 */
    if (r->span == 20 && i >= 7 && i <= 13) {
        r->sum += x;
        r->count++;
    }
    if (r->span == 5 && i == 2) {
        r->sum += x;
        r->count++;
    } else if (i >= r->centre - 2 && i <= r->centre + 2) {
        r->sum += x;
        r->count++;
    }
    r->lastsample = x;
    r->i++;
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

/* One finished symbol through the hunting / in-frame logic.  rec4 as orc_slicer_step; returns flag bits:
 * 1 = in frame (have_sync == 1 when the symbol was read), 2 = sync accepted on this symbol, 4 = negative polarity */
static int
symbol_commit(orc_p25rx* r, float sym, int rec4[4]) {
    orc_slicer* s = &r->sl;
    /* dsd_symbol_history_push() */
    r->shist[r->shead] = sym;
    r->shead = (r->shead + 1) % 24;
    if (r->scount < 24) {
        r->scount++;
    }
    if (r->have_sync) {
        s->negative = (r->lastsync == 2);
        orc_slicer_step(s, sym, rec4);
        int flags = 1 | (s->negative ? 4 : 0);
        if (r->lock_symbols < 0) {
            if (!orc_p25h_symbol(&r->h, r->n_sym, rec4[0], rec4[2], rec4[3], r->ev)) {
                hunt_enter(r);
            }
        } else if (--r->lock_left <= 0) {
            hunt_enter(r);
        }
        return flags;
    }
    /* hunting */
    r->lbuf[r->lidx] = sym;
    if (r->level_count < 24) {
        r->level_count++;
    }
    s->sbuf[s->sidx] = sym;
    r->lidx = (r->lidx == 23) ? 0 : r->lidx + 1;
    s->sidx = (s->sidx == ORC_SLICER_SSIZE - 1) ? 0 : s->sidx + 1;
    const int bit = sym > 0 ? 1 : 0;
    r->hist_bits = ((r->hist_bits << 1) | (uint32_t)bit) & 0xFFFFFFu;
    if (r->hist_count < 24) {
        r->hist_count++;
    }
    rec4[0] = bit ? 1 : 3;
    rec4[1] = 0;
    rec4[2] = 0;
    rec4[3] = 0;
    int flags = 0;
    if (r->hist_count >= 8) {
        float tmp[24];
        memcpy(tmp, r->lbuf, sizeof(float) * (size_t)r->level_count);
        sort_small(tmp, r->level_count);
        orc_level_estimate(tmp, r->level_count, &r->lmin, &r->lmax);
        s->maxref = s->max;
        s->minref = s->min;
        if (r->hist_count >= 24) {
            int pol = 0;
            if (r->hist_bits == P25_SYNC_BITS) {
                pol = 1;
            } else if (r->hist_bits == (~P25_SYNC_BITS & 0xFFFFFFu)) {
                pol = 2;
            }
            if (pol) {
                s->max = (s->max + r->lmax) / 2;
                s->min = (s->min + r->lmin) / 2;
                r->lastsync = pol;
                if (r->use_filter) {
                    r->filter_on = 1;
                }
                float nf[24];
                for (int k = 0; k < 24; k++) {
                    nf[k] = r->shist[(r->shead - 1 - k + 48) % 24];
                }
                if (r->scount >= 24) {
                    (void)orc_slicer_warm_start(s, nf, 24);
                }
                r->have_sync = 1;
                r->lock_left = r->lock_symbols;
                flags |= 2 | (pol == 2 ? 4 : 0);
                if (r->lock_symbols < 0) {
                    (void)orc_p25h_begin(&r->h);
                } else if (r->lock_left <= 0) {
                    hunt_enter(r);
                }
                return flags;
            }
        }
    }
    /* frame_sync_advance_sync_window() then frame_sync_handle_no_sync_timeout() */
    if (r->hunt_pos < 10200) {
        r->hunt_pos++;
    } else {
        r->hunt_pos = 0;
        no_carrier(r);
    }
    if (r->lastsync != 2 && r->hunt_pos >= 1800) {
        no_carrier(r);
        hunt_enter(r); /* getFrameSync() returns -1, the engine calls it again */
    }
    return flags;
}

/* Consume n discriminator samples; emit finished symbols.  out_sym[k], rec4[4k..], flags[k]; returns the count. */
long
orc_p25rx_run(orc_p25rx* r, const float* in, long n, float* out_sym, int* rec4, uint8_t* flags, long max_out) {
    long o = 0;
    for (long k = 0; k < n; k++) {
        if (!r->in_symbol) {
            symbol_begin(r);
        }
        sample_step(r, in[k]);
        if (r->i >= r->span) {
            const float sym = (r->count > 0) ? (r->sum / (float)r->count) : 0.0f;
            r->in_symbol = 0;
            int rr[4];
            const int f = symbol_commit(r, sym, rr);
            r->n_sym++;
            if (o < max_out) {
                out_sym[o] = sym;
                memcpy(rec4 + 4 * o, rr, sizeof(rr));
                flags[o] = (uint8_t)f;
            }
            o++;
        }
    }
    return o;
}

void
orc_p25rx_get_thresholds(const orc_p25rx* r, float out7[7]) {
    out7[0] = r->sl.center;
    out7[1] = r->sl.umid;
    out7[2] = r->sl.lmid;
    out7[3] = r->sl.max;
    out7[4] = r->sl.min;
    out7[5] = r->sl.maxref;
    out7[6] = r->sl.minref;
}

size_t
orc_p25rx_sizeof(void) {
    return sizeof(orc_p25rx);
}
