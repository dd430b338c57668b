/*
 * oracle/ddn_oracle_rx4.c - TEST INFRASTRUCTURE ONLY: CPU restatement of the fixed-protocol receive loop between the
 * discriminator stream and the dibits for the 4-level FSK protocols of BASELINE configs[3]: P25 Phase 1, DMR, NXDN48.
 * One loop, driven by a profile (sync patterns, matched filter, window / timing rules, what an accepted sync does).
 *
 *   getSymbol() RTL-FSK path       src/dsp/dsd_symbol.c:1343-1387,1769-1805 as oracle/ddn_oracle_rx.c, plus the pieces
 *                                  that differ by modulation lock / protocol: window selection :197-224 (C4FM left edge 1
 *                                  once a DMR type is the last sync; GFSK l = r = 1), accumulation :404-460 (the sps-20
 *                                  7..13 window added on top, GFSK = the two edge samples only), slip rules :462-517
 *                                  (sps 20, GFSK, C4FM), in-sync clip only on C4FM :347-358, matched-filter family by
 *                                  lastsynctype :301-338 (dmr_filter / nxdn_filter / p25_filter, src/dsp/dsd_filters.c)
 *   getFrameSync()                 src/dsp/dsd_frame_sync.c:3098-3148; ring :1729-1764 (level ring 12 long on the
 *                                  2400-baud profile); sign dibit :2110-2127; 4-level payload dibit + reliability stored
 *                                  while hunting :2161-2189; level window :2316-2336; timeouts :2753-2760,3037-3053
 *       P25p1 accept               :603-625,698-716 (as ddn_oracle_rx.c)
 *       DMR accept                 :1102-1106,1150-1260 (+ MS / direct-mode :1108-1314): basic lock :385-392, then
 *                                  dmr_resample_on_sync() src/dsp/dmr_sync.c:109-131 = warm start over the 24 sync
 *                                  symbols + re-digitisation of the 66 payload dibits before the sync :63-103
 *       NXDN accept                :1507-1556: 10-symbol window, five patterns per polarity, a match first only becomes
 *                                  lastsynctype (which turns the NXDN filter on); the next match of the same type is
 *                                  accepted with a 10-symbol warm start
 *   in-frame symbol                get_dibit_and_analog_signal, src/core/frames/dsd_dibit.c:1045-1076; use_symbol :243-299
 *                                  (continuous threshold update only for P25p1); digitize :1018-1043 (stored dibit =
 *                                  un-inverted for negative types)
 *
 * PARITY STATUS: as ddn_oracle_rx.c - slicer, reliability, warm start, level window and the three FIRs are pinned to
 * compiled reference objects (the taps are measured from them, ddn_tables_fsk4.h); dsd_symbol.c / dsd_frame_sync.c /
 * dmr_sync.c cannot be compiled here, their loops are restated.  With the P25p1 profile this loop must equal
 * ddn_oracle_rx.c symbol for symbol (tests/test_oracle_rx4.py); the DMR profile is anchored on the reference's
 * full-chain known answers for its own captures ("Color Code=02", tests/CMakeLists.txt:8925-8930).
 * Deviations from a full dsd-neo run: one protocol, modulation locked (-mc or -mg) so no modulation voting / SPS
 * hunting; non-inverted DMR only; the handler of a sync type consumes a configured number of symbols
 * (lock_symbols[class]: DMR data 54 + 66 = 120, src/protocol/dmr/dmr_data.c:213-302; NXDN 182; DMR voice = the host's
 * call length) instead of running the protocol decoders.
 *
 * M17 (profile.m17, round 5): -fz = C4FM lock at 4800 symbols/s with NO matched filter (decode_mode_apply_m17(),
 * src/runtime/decode_mode.c:486-510: use_cosine_filter = 0), static in-frame thresholds (use_symbol() :269-283 keeps them live for
 * P25p1 / QPSK only), 8-symbol warm start.  The matcher is frame_sync_try_m17() (src/dsp/dsd_frame_sync.c:1060-1100): Hamming
 * distance of the last eight sign dibits to the preamble / EOT / LSF / BERT / stream / packet words, each accepted only after the
 * sync type that may precede it, polarity learnt from the preamble; the handlers consume fixed counts (dispatch_m17.c:25-68:
 * preamble skipDibit(8), EOT skipDibit(184) then lastsynctype = NONE, every frame type 184 dibits).
 */
#include <string.h>

#include "ddn_oracle.h"

void
orc_fsk4rx_init(orc_fsk4rx* r, const orc_fsk4_profile* p) {
    memset(r, 0, sizeof(*r));
    r->p = *p;
    r->jitter = -1;
    orc_slicer_init(&r->sl, 0);
    r->lmin = r->sl.min;
    r->lmax = r->sl.max;
    orc_p25h_init(&r->hp25, 64);
    orc_dmrh_init(&r->hdmr);
    orc_nxdnh_init(&r->hnxdn);
}

void
orc_fsk4rx_set_events(orc_fsk4rx* r, orc_hevents* ev) {
    r->ev = ev;
    if (ev) {
        ev->n = 0;
    }
}

static void
no_carrier(orc_fsk4rx* r) { /* engine.c:1838-1848 as far as this loop sees it */
    r->m17_pol = 0;
    r->jitter = -1;
    r->lastsync = 0;
    r->filter_on = 0;
    r->sl.max = 15000.0f;
    r->sl.min = -15000.0f;
    r->sl.center = 0.0f;
    r->need_reset = 1;
    orc_p25h_no_carrier(&r->hp25);
    orc_dmrh_no_carrier(&r->hdmr);
}

static void
hunt_enter(orc_fsk4rx* r) {
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
matched_filter(orc_fsk4rx* r, float x) { /* apply_sps_fir(): products added oldest first, mul and add rounded apart */
    const int nt = r->p.nt;
    float acc = 0.0f;
    for (int i = 0; i < nt - 1; i++) {
        float t;
        memcpy(&t, &r->p.taps[i], 4);
        acc += t * r->fhist[i];
    }
    float t;
    memcpy(&t, &r->p.taps[nt - 1], 4);
    acc += t * x;
    memmove(r->fhist, r->fhist + 1, sizeof(float) * (size_t)(nt - 2));
    r->fhist[nt - 2] = x;
    return acc;
}

/* symbol_adjust_timing_index(), src/dsp/dsd_symbol.c:462-517 (the shape of the reference's own test hook
 * dsd_symbol_test_adjust_timing_index, :539-553): returns the sample index the symbol starts at (start_i, one less or one
 * more), *jitter_after = the latch afterwards.  sps 20 has its own rule, then QPSK (rf_mod 1), GFSK (2), C4FM (0). */
int
orc_fsk4_adjust_timing(int sps, int centre, int rf_mod, int jitter, int have_sync, int span, int start_i, int* jitter_after) {
    int i = start_i;
    *jitter_after = jitter;
    if (span <= 1 || i != 0 || have_sync != 0 || jitter < 0) {
        return i;
    }
    const int j = jitter, c = centre;
    if (sps == 20) {
        if (j >= 7 && j <= 10) {
            i--;
        } else if (j >= 11 && j <= 14) {
            i++;
        }
    } else if (rf_mod == 1) {
        if (j >= 0 && j < c) {
            i++;
        } else if (j > c && j < 10) {
            i--;
        }
    } else if (rf_mod == 2) {
        if (j >= c - 1 && j <= c) {
            i--;
        } else if (j >= c + 1 && j <= c + 2) {
            i++;
        }
    } else if (rf_mod == 0) {
        if (j > 0 && j <= c) {
            i--;
        } else if (j > c && j < sps) {
            i++;
        }
    }
    *jitter_after = -1;
    return i;
}

/* select_window_c4fm / _qpsk / _gfsk, src/dsp/dsd_symbol.c:197-224: the samples either side of the centre that enter the symbol.
 * narrow = the C4FM left edge moved in (YSF sync type, or a DMR type as last sync). */
void
orc_fsk4_window(int rf_mod, int narrow, int* l_edge, int* r_edge) {
    if (rf_mod == 0) {
        *l_edge = narrow ? 1 : 2;
        *r_edge = 2;
    } else if (rf_mod == 1) {
        *l_edge = 1;
        *r_edge = 2;
    } else {
        *l_edge = 1;
        *r_edge = 1;
    }
}

static void
symbol_begin(orc_fsk4rx* r) {
    const orc_fsk4_profile* p = &r->p;
    if (r->need_reset) {
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
    int whole = p->out_rate / p->sym_rate, rem = p->out_rate % p->sym_rate;
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
        if (acc >= p->sym_rate) {
            sps++;
            acc -= p->sym_rate;
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
    /* symbol_adjust_timing_index(): at i == 0, hunting only, one of the modulation's rules */
    r->i = orc_fsk4_adjust_timing(sps, r->centre, p->rf_mod, r->jitter, r->have_sync, sps, 0, &r->jitter);
}

static void
sample_step(orc_fsk4rx* r, float x) {
    const orc_fsk4_profile* p = &r->p;
    orc_slicer* s = &r->sl;
    if (r->filter_on) {
        x = matched_filter(r, x);
    }
    if (r->have_sync && p->rf_mod == 0) {
        if (x > s->max) {
            x = s->max;
        } else if (x < s->min) {
            x = s->min;
        }
    }
    const int i = r->i, c = r->centre;
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
    /* symbol_accumulate_sample() */
    int take = 0;
    if (r->span == 20 && i >= 7 && i <= 13) {
        r->sum += x;
        r->count++;
    }
    if (r->span == 5 && i == 2) {
        take = 1;
    } else {
        int l, rr;
        orc_fsk4_window(p->rf_mod, p->dmr_window && r->lastsync != 0, &l, &rr);
        if (p->rf_mod == 0) {
            take = (i >= c - l && i <= c + rr);
        } else {
            take = (r->span <= 4) ? (i == c) : (i == c - l || i == c + rr);
        }
    }
    if (take) {
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

static int
slice4(const orc_slicer* s, float sym) { /* dmr_digitize_symbol / frame_sync_store_dmr_payload_symbol */
    if (sym > s->center) {
        return (sym > s->umid) ? 1 : 0;
    }
    return (sym < s->lmid) ? 3 : 2;
}

static void
warm_start(orc_fsk4rx* r, int len) {
    if (r->scount < len) {
        return;
    }
    float nf[24];
    for (int k = 0; k < len; k++) {
        nf[k] = r->shist[(r->shead - 1 - k + 4 * ORC_FSK4_HIST) % ORC_FSK4_HIST];
    }
    (void)orc_slicer_warm_start(&r->sl, nf, len);
}

/* frame_sync_try_m17(), src/dsp/dsd_frame_sync.c:865-1100, with only M17 enabled (max_hamming 1 for the preamble, no repeated-marker
 * rule: that one is D-STAR's).  w8 = the last eight sign dibits ('1' -> 1), oldest first.  Returns the outcome's row in the M17 table
 * (tests/rx4.py: 0 / 1 preamble + / -, 2 / 3 EOT, 4 / 5 LSF, 6 / 7 BERT, 8 / 9 stream, 10 / 11 packet) or -1. */
enum { M17_W_LSF = 0xF2, M17_W_STR = 0x0D, M17_W_PRE = 0x55, M17_W_PIV = 0xAA, M17_W_BRT = 0x4F, M17_W_PKT = 0xB0, M17_W_EOT = 0xFD, M17_W_EOT_INV = 0x02 };
static int
m17_ham(uint32_t w8, uint32_t word) {
    return __builtin_popcount((w8 ^ word) & 0xFFu);
}
static int
m17_match(orc_fsk4rx* r, uint32_t w8) {
    const orc_fsk4_profile* p = &r->p;
    const int last = r->lastsync;
    const int t_pre_p = p->pat_type[0], t_pre_n = p->pat_type[1], t_lsf_p = p->pat_type[4], t_lsf_n = p->pat_type[5];
    const int t_brt_p = p->pat_type[6], t_brt_n = p->pat_type[7], t_str_p = p->pat_type[8], t_str_n = p->pat_type[9];
    const int t_pkt_p = p->pat_type[10], t_pkt_n = p->pat_type[11];
    const int inv = r->m17_pol == 2; /* opts->inverted_m17 = 0 */
    /* preamble first (:865-903) */
    if (m17_ham(w8, M17_W_PRE) <= 1) {
        r->m17_pol = 1;
        return 0;
    }
    if (m17_ham(w8, M17_W_PIV) <= 1) {
        r->m17_pol = 2;
        return 1;
    }
    /* EOT, only after a frame type (:905-933) */
    {
        const int ham = inv ? m17_ham(w8, M17_W_EOT_INV) : m17_ham(w8, M17_W_EOT);
        const int after_frame = last == t_lsf_p || last == t_lsf_n || last == t_str_p || last == t_str_n || last == t_pkt_p
                                || last == t_pkt_n || last == t_brt_p || last == t_brt_n;
        if (ham <= 1 && after_frame) {
            r->m17_pol = 0;
            return inv ? 3 : 2;
        }
    }
    /* LSF after the preamble, BERT after the preamble or a BERT frame (:1005-1058) */
    {
        const int after_pre = (!inv && last == t_pre_p) || (inv && last == t_pre_n);
        const int after_brt = (!inv && last == t_brt_p) || (inv && last == t_brt_n);
        if (after_pre || after_brt) {
            if (after_pre && (inv ? m17_ham(w8, M17_W_STR) : m17_ham(w8, M17_W_LSF)) <= 1) {
                return inv ? 5 : 4;
            }
            if ((inv ? m17_ham(w8, M17_W_PKT) : m17_ham(w8, M17_W_BRT)) <= 1) {
                return inv ? 7 : 6;
            }
        }
    }
    /* stream frames after an LSF or a stream frame (:965-995) */
    if (m17_ham(w8, M17_W_STR) <= 1 && !inv) {
        if (last == t_lsf_p || last == t_str_p) {
            return 8;
        }
    } else if (m17_ham(w8, M17_W_LSF) <= 1 && inv) {
        if (last == t_lsf_n || last == t_str_n) {
            return 9;
        }
    }
    /* packet frames after an LSF or a packet frame (:935-963) */
    if (m17_ham(w8, M17_W_PKT) <= 1 && !inv) {
        if (last == t_lsf_p || last == t_pkt_p) {
            return 10;
        }
    } else if (m17_ham(w8, M17_W_BRT) <= 1 && inv) {
        if (last == t_lsf_n || last == t_pkt_n) {
            return 11;
        }
    }
    return -1;
}

/* One finished symbol.  pay2 = {payload dibit, reliability}.  Returns the flag bits. */
static int
symbol_commit(orc_fsk4rx* r, float sym, int rec4[4], uint8_t pay2[2]) {
    const orc_fsk4_profile* p = &r->p;
    orc_slicer* s = &r->sl;
    const int slot = r->shead;
    r->shist[slot] = sym; /* dsd_symbol_history_push() */
    r->shead = (r->shead + 1) % ORC_FSK4_HIST;
    if (r->scount < ORC_FSK4_HIST) {
        r->scount++;
    }
    if (r->have_sync) {
        const int neg = p->pat_neg[r->cur_pat];
        s->negative = neg;
        if (p->live_thresholds) {
            orc_slicer_step(s, sym, rec4);
        } else {
            orc_slicer_step_static(s, sym, rec4);
        }
        const int d = rec4[0];
        pay2[0] = (uint8_t)(neg ? (d ^ 2) : d); /* stored_dibit: invert_dibit() swaps 0<->2, 1<->3 */
        pay2[1] = (uint8_t)rec4[1];
        r->phist[slot] = pay2[0];
        r->rhist[slot] = pay2[1];
        const int flags = 1 | (neg ? 4 : 0);
        if (p->handler) {
            int on;
            if (p->proto == 0) {
                on = orc_p25h_symbol(&r->hp25, r->n_sym, d, rec4[2], rec4[3], r->ev);
            } else if (p->proto == 1) {
                on = orc_dmrh_symbol(&r->hdmr, r->n_sym, pay2[0], pay2[1], r->ev);
            } else {
                int bad = 0;
                on = orc_nxdnh_symbol(&r->hnxdn, r->n_sym, d, &bad, r->ev);
                if (bad) { /* nxdn_mark_bad_sync(): lastsynctype NONE - the NXDN filter gate and the two-match rule see it */
                    r->lastsync = 0;
                    r->filter_on = 0;
                }
            }
            if (!on) {
                hunt_enter(r);
            }
        } else if (--r->lock_left <= 0) {
            if (p->m17 && (r->cur_pat == 2 || r->cur_pat == 3)) { /* dsd_dispatch_handle_m17(): the EOT handler ends the transmission */
                r->lastsync = 0;
                r->m17_pol = 0;
            }
            hunt_enter(r);
        }
        return flags;
    }
    /* hunting */
    r->lbuf[r->lidx] = sym;
    if (r->level_count < p->t_max) {
        r->level_count++;
    }
    s->sbuf[s->sidx] = sym;
    r->lidx = (r->lidx == p->t_max - 1) ? 0 : r->lidx + 1;
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
    pay2[0] = (uint8_t)slice4(s, sym);
    pay2[1] = (uint8_t)orc_slicer_reliability(s, sym);
    r->phist[slot] = pay2[0];
    r->rhist[slot] = pay2[1];
    int flags = 0;
    if (r->hist_count >= 8) {
        float tmp[24];
        memcpy(tmp, r->lbuf, sizeof(float) * (size_t)r->level_count);
        sort_small(tmp, r->level_count);
        orc_level_estimate(tmp, r->level_count, &r->lmin, &r->lmax);
        s->maxref = s->max;
        s->minref = s->min;
        if (r->hist_count >= p->win_len) {
            const uint32_t mask = (p->win_len >= 24) ? 0xFFFFFFu : ((1u << p->win_len) - 1u);
            const uint32_t w = r->hist_bits & mask;
            int hit = -1;
            if (p->m17) {
                hit = m17_match(r, w);
            } else {
                for (int k = 0; k < p->n_pat; k++) {
                    if (w == p->pat_bits[k]) {
                        hit = k;
                        break;
                    }
                }
            }
            if (hit >= 0) {
                const int type = p->pat_type[hit];
                s->max = (s->max + r->lmax) / 2;
                s->min = (s->min + r->lmin) / 2;
                int accept = 1;
                if (p->confirm && r->lastsync != type) {
                    accept = 0; /* frame_sync_try_nxdn(): remembered, not yet returned */
                }
                r->lastsync = type;
                if (p->use_filter) {
                    r->filter_on = 1;
                }
                if (accept) {
                    if (p->redigitize) {
                        /* dmr_resample_on_sync(): nothing at all without 24 symbols of history; the CACH rewrite needs 90 */
                        if (r->scount >= 24) {
                            warm_start(r, 24);
                            if (r->scount >= 90) {
                                for (int i = 0; i < 66; i++) {
                                    const int back = 89 - i;
                                    const int sl = (r->shead - 1 - back + 4 * ORC_FSK4_HIST) % ORC_FSK4_HIST;
                                    r->phist[sl] = (uint8_t)slice4(s, r->shist[sl]);
                                }
                            }
                        }
                    } else if (!(p->m17 && (hit == 2 || hit == 3))) { /* (the EOT marker takes the basic lock only, :905-933) */
                        warm_start(r, p->warm_len);
                    }
                    if (r->sync_thr && r->sync_thr_n < r->sync_thr_max) {
                        float* t = r->sync_thr + 5 * (size_t)r->sync_thr_n;
                        t[0] = s->center, t[1] = s->umid, t[2] = s->lmid, t[3] = s->max, t[4] = s->min;
                    }
                    r->sync_thr_n++;
                    r->have_sync = 1;
                    r->cur_pat = hit;
                    r->lock_left = p->lock_symbols[p->pat_class[hit] & 3];
                    flags |= 2 | (p->pat_neg[hit] ? 4 : 0) | (hit << 3);
                    if (p->handler) {
                        int on;
                        if (p->proto == 0) {
                            on = orc_p25h_begin(&r->hp25);
                        } else if (p->proto == 1) {
                            if (hit < 2) { /* BS data / BS voice words; the other types keep the configured count */
                                uint8_t pre[ORC_FSK4_PRE], prel[ORC_FSK4_PRE];
                                for (int i = 0; i < ORC_FSK4_PRE; i++) {
                                    const int sl = (r->shead - ORC_FSK4_PRE + i + 4 * ORC_FSK4_HIST) % ORC_FSK4_HIST;
                                    const int have = (ORC_FSK4_PRE - i) <= r->scount;
                                    pre[i] = have ? r->phist[sl] : 0;
                                    prel[i] = have ? r->rhist[sl] : 0;
                                }
                                on = orc_dmrh_begin(&r->hdmr, r->n_sym, p->pat_class[hit] & 1, pre, prel, r->ev);
                            } else {
                                on = orc_dmrh_begin_fixed(&r->hdmr, r->lock_left);
                            }
                        } else {
                            on = orc_nxdnh_begin(&r->hnxdn);
                        }
                        if (!on) {
                            hunt_enter(r);
                        }
                    } else if (r->lock_left <= 0) {
                        hunt_enter(r);
                    }
                    return flags;
                }
            }
        }
    }
    if (r->hunt_pos < 10200) {
        r->hunt_pos++;
    } else {
        r->hunt_pos = 0;
        no_carrier(r);
    }
    if (!(p->slow_type && r->lastsync == p->slow_type) && r->hunt_pos >= 1800) {
        no_carrier(r);
        hunt_enter(r);
    }
    return flags;
}

long
orc_fsk4rx_run(orc_fsk4rx* r, const float* in, long n, float* out_sym, int* rec4, uint8_t* flags, uint8_t* pay2,
               long max_out, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre, uint8_t* pre_rel, int max_sync,
               int* n_sync) {
    long o = 0;
    int ns = 0;
    for (long k = 0; k < n; k++) {
        if (!r->in_symbol) {
            symbol_begin(r);
        }
        sample_step(r, in[k]);
        if (r->i >= r->span) {
            const float sym = (r->count > 0) ? (r->sum / (float)r->count) : 0.0f;
            r->in_symbol = 0;
            int rr[4];
            uint8_t pp[2];
            const int f = symbol_commit(r, sym, rr, pp);
            r->n_sym++;
            if (o < max_out) {
                out_sym[o] = sym;
                memcpy(rec4 + 4 * o, rr, sizeof(rr));
                flags[o] = (uint8_t)f;
                pay2[2 * o] = pp[0];
                pay2[2 * o + 1] = pp[1];
            }
            if (f & 2) {
                if (ns < max_sync) {
                    sync_pos[ns] = (int32_t)o;
                    sync_pat[ns] = (uint8_t)((f >> 3) & 31);
                    for (int i = 0; i < ORC_FSK4_PRE; i++) {
                        const int sl = (r->shead - ORC_FSK4_PRE + i + 4 * ORC_FSK4_HIST) % ORC_FSK4_HIST;
                        const int have = (ORC_FSK4_PRE - i) <= r->scount;
                        pre[(size_t)ns * ORC_FSK4_PRE + i] = have ? r->phist[sl] : 0;
                        pre_rel[(size_t)ns * ORC_FSK4_PRE + i] = have ? r->rhist[sl] : 0;
                    }
                }
                ns++;
            }
            o++;
        }
    }
    *n_sync = ns;
    return o;
}

void
orc_fsk4rx_set_sync_thresholds(orc_fsk4rx* r, float* buf, int max_syncs) { /* restarts the count */
    r->sync_thr = buf;
    r->sync_thr_max = max_syncs;
    r->sync_thr_n = 0;
}

size_t
orc_fsk4rx_sizeof(void) {
    return sizeof(orc_fsk4rx);
}
size_t
orc_fsk4_profile_sizeof(void) {
    return sizeof(orc_fsk4_profile);
}
void
orc_fsk4rx_get_thresholds(const orc_fsk4rx* r, float out7[7]) {
    out7[0] = r->sl.center;
    out7[1] = r->sl.umid;
    out7[2] = r->sl.lmid;
    out7[3] = r->sl.max;
    out7[4] = r->sl.min;
    out7[5] = r->sl.maxref;
    out7[6] = r->sl.minref;
}
