/*
 * oracle/ddn_oracle_symbolizer.c - TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * getSymbol()'s RTL-FSK discriminator sample loop restated for every window / timing variant it has, not just the P25
 * Phase 1 one ddn_oracle_rx.c hard-wires (src/dsp/dsd_symbol.c):
 *   samples per symbol      :1343-1374  whole = rate / symbol rate clamped to [2, 64], Bresenham remainder accumulator
 *   symbol centre           include/dsd-neo/core/opts.h:751-753  (sps - 1) / 2
 *   window edges            :197-225    C4FM l = r = 2 (l = 1 after a YSF / DMR sync), QPSK l = 1 r = 2, GFSK l = r = 1
 *   accumulation            :398-460    sps 20: samples 7..13 first (NXDN48), sps 5: sample 2 only, rf_mod 0: every sample
 *                                       in [centre - l, centre + r], otherwise the two samples centre - l and centre + r
 *                                       (GFSK at sps <= 4: the centre sample)
 *   timing slip             :462-516    only at i == 0 and while hunting: NXDN (sps 20) 7..10 -> -1, 11..14 -> +1; C4FM
 *                                       0 < j <= centre -> -1, centre < j < sps -> +1; GFSK centre-1..centre -> -1,
 *                                       centre+1..centre+2 -> +1; QPSK 0 <= j < centre -> +1, centre < j < 10 -> -1
 *   crossing latch          :360-397    first sample that crosses `center` with |x| within 1.25 x the reference level
 *   in-sync clip            :347-358    rf_mod 0 only
 *
 * PARITY: dsd_symbol.c cannot be compiled here (<sndfile.h>), so this is pinned by the reference's own assertions on
 * that loop - tests/dsp/test_rtl_symbol_cache_generation.c:414-493 (ramp inputs: symbol values, samples per symbol,
 * centre, accumulator, the slip by a latched crossing) - replayed in tests/test_oracle_rx_kat.py.
 */
#include <string.h>

#include "ddn_oracle.h"

void
orc_symbolizer_init(orc_symbolizer* s, int out_rate_hz, int sym_rate_hz, int rf_mod, int l_edge, int r_edge) {
    memset(s, 0, sizeof(*s));
    s->out_rate = out_rate_hz;
    s->sym_rate = sym_rate_hz;
    s->rf_mod = rf_mod;
    s->l_edge = l_edge;
    s->r_edge = r_edge;
    s->jitter = -1;
    s->center = 0.0f;
    s->min = -30000.0f; /* symbol_reset_rtl_fsk_discriminator_slicer(), :1306-1326 */
    s->max = 30000.0f;
    s->minref = -24000.0f;
    s->maxref = 24000.0f;
}

static int
next_sps(orc_symbolizer* s) {
    int whole = s->out_rate / s->sym_rate, rem = s->out_rate % s->sym_rate;
    if (whole < 2) {
        whole = 2;
        rem = 0;
    }
    if (whole > 64) {
        whole = 64;
        rem = 0;
    }
    if (rem <= 0) {
        return whole;
    }
    int acc = s->sps_accum + rem;
    if (acc >= s->sym_rate) {
        whole++;
        acc -= s->sym_rate;
    }
    s->sps_accum = acc;
    return whole > 64 ? 64 : whole;
}

/* One symbol: consumes what it needs from in[0..n), returns the number of samples consumed, or -1 when fewer than a
 * whole symbol is available (nothing is consumed then). */
long
orc_symbolizer_symbol(orc_symbolizer* s, const float* in, long n, int have_sync, float* out_sym) {
    orc_symbolizer t = *s;
    const int sps = next_sps(&t);
    const int centre = (sps - 1) / 2;
    int i = 0;
    if (sps > 1 && have_sync == 0 && t.jitter >= 0) {
        if (sps == 20) {
            if (t.jitter >= 7 && t.jitter <= 10) {
                i--;
            } else if (t.jitter >= 11 && t.jitter <= 14) {
                i++;
            }
        } else if (t.rf_mod == 1) {
            if (t.jitter >= 0 && t.jitter < centre) {
                i++;
            } else if (t.jitter > centre && t.jitter < 10) {
                i--;
            }
        } else if (t.rf_mod == 2) {
            if (t.jitter >= centre - 1 && t.jitter <= centre) {
                i--;
            } else if (t.jitter >= centre + 1 && t.jitter <= centre + 2) {
                i++;
            }
        } else {
            if (t.jitter > 0 && t.jitter <= centre) {
                i--;
            } else if (t.jitter > centre && t.jitter < sps) {
                i++;
            }
        }
        t.jitter = -1;
    }
    if (n < (long)(sps - i)) {
        return -1;
    }
    float sum = 0.0f;
    int count = 0;
    long k = 0;
    for (; i < sps; i++, k++) {
        float x = in[k];
        if (have_sync && t.rf_mod == 0) {
            if (x > t.max) {
                x = t.max;
            } else if (x < t.min) {
                x = t.min;
            }
        }
        if (x > t.center) {
            if (!(x > t.maxref * 1.25f) && t.jitter < 0 && t.lastsample < t.center) {
                t.jitter = i;
            }
        } else if (!(x < t.minref * 1.25f) && t.jitter < 0 && t.lastsample > t.center) {
            t.jitter = i;
        }
        if (sps == 20 && i >= 7 && i <= 13) {
            sum += x;
            count++;
        }
        if (sps == 5 && i == 2) {
            sum += x;
            count++;
        } else if (t.rf_mod == 0) {
            if (i >= centre - t.l_edge && i <= centre + t.r_edge) {
                sum += x;
                count++;
            }
        } else if (t.rf_mod == 2 && sps <= 4) {
            if (i == centre) {
                sum += x;
                count++;
            }
        } else if (i == centre - t.l_edge || i == centre + t.r_edge) {
            sum += x;
            count++;
        }
        t.lastsample = x;
    }
    t.last_sps = sps;
    t.last_centre = centre;
    *s = t;
    *out_sym = count > 0 ? sum / (float)count : 0.0f;
    return k;
}
