/*
 * oracle/ddn_oracle_fec.c — CPU restatement of the trellis / Viterbi decoders on the hot path
 * (TEST INFRASTRUCTURE ONLY; integer arithmetic, bit-exact target).
 *
 *   P25 1/2-rate 4-state trellis, LLR costs      src/protocol/p25/p25_12.c:204-283 (best path), :144-202 (list-8)
 *   3/4-rate 8-state trellis hard / soft          src/protocol/dmr/dmr_34_viterbi.c:205-255,365-407
 *   K=5 R=1/2 16-state, u16 metrics (NXDN)        src/protocol/nxdn/nxdn_convolution.c:57-99 (hard), :124-160 (soft)
 *   K=5 R=1/2 16-state, u32 metrics (M17/YSF)     src/core/util/dsd_misc.c:118-283
 *   4-bit look-ahead trellis decode               src/core/util/dsd_misc.c:29-74
 * Tables are the TIA-102 / ETSI TS 102 361 trellis constants the reference carries in src/fec/trellis34.c and
 * src/protocol/p25/p25_12.c:19.
 *
 * Pinned bit-exact against the compiled reference by tests/test_oracle_fec.py.
 */
#include "ddn_oracle.h"

#include <stdlib.h>
#include <string.h>

/* 98-dibit block interleaver: 13+12+12+12 dibit pairs taken with stride 4 pairs (src/fec/trellis34.c:7-12) */
void
orc_trellis_interleave_98(uint8_t tbl[98]) {
    int n = 0;
    for (int lane = 0; lane < 4; lane++) {
        for (int pair = lane; pair < 49; pair += 4) {
            tbl[n++] = (uint8_t)(2 * pair);
            tbl[n++] = (uint8_t)(2 * pair + 1);
        }
    }
}

/* dibit-pair expected for a (state -> state) transition of the 1/2-rate FSM (src/protocol/p25/p25_12.c:19) */
static const uint8_t k_p25_half_rate_nibble[16] = {2, 12, 1, 15, 14, 0, 13, 3, 9, 7, 10, 4, 5, 11, 6, 8};
/* 3/4-rate: constellation point per (state, tribit), point <-> dibit-pair maps (src/fec/trellis34.c:14-21) */
static const uint8_t k_r34_point_to_nibble[16] = {2, 10, 7, 15, 14, 6, 11, 3, 13, 5, 8, 0, 1, 9, 4, 12};
static const uint8_t k_r34_nibble_to_point[16] = {11, 12, 0, 7, 14, 9, 5, 2, 10, 13, 1, 6, 15, 8, 4, 3};
static const uint8_t k_r34_fsm[64] = {0, 8,  4, 12, 2, 10, 6, 14, 4, 12, 2, 10, 6, 14, 0, 8, 1, 9,  5, 13, 3, 11,
                                      7, 15, 5, 13, 3, 11, 7, 15, 1, 9,  3, 11, 7, 15, 1, 9, 5, 13, 7, 15, 1, 9,
                                      5, 13, 3, 11, 2, 10, 6, 14, 0, 8,  4, 12, 6, 14, 0, 8, 4, 12, 2, 10};

const uint8_t*
orc_tbl_p25_half_rate_nibble(void) {
    return k_p25_half_rate_nibble;
}
const uint8_t*
orc_tbl_r34_point_to_nibble(void) {
    return k_r34_point_to_nibble;
}
const uint8_t*
orc_tbl_r34_nibble_to_point(void) {
    return k_r34_nibble_to_point;
}
const uint8_t*
orc_tbl_r34_fsm(void) {
    return k_r34_fsm;
}

static inline uint32_t
llr_disagreement(int16_t llr, int bit) {
    /* positive llr favours bit 1; cost = how strongly the observation contradicts `bit` */
    if (bit) {
        return llr < 0 ? (uint32_t)(-llr) : 0u;
    }
    return llr > 0 ? (uint32_t)llr : 0u;
}

/* P25 1/2-rate best path.  out12 = first 48 state dibits MSB-first; returns final metric >> 8. */
int
orc_p25_12_soft_llr(const int16_t* llr196, uint8_t out12[12]) {
    uint8_t il[98];
    int16_t d[196];
    orc_trellis_interleave_98(il);
    memset(d, 0, sizeof(d));
    for (int i = 0; i < 98; i++) {
        d[2 * il[i]] = llr196[2 * i];
        d[2 * il[i] + 1] = llr196[2 * i + 1];
    }
    uint32_t prev[4] = {0, 256, 256, 256}, cur[4];
    uint8_t back[49][4];
    for (int t = 0; t < 49; t++) {
        for (int ns = 0; ns < 4; ns++) {
            uint32_t best = 0xFFFFFFFFu;
            uint8_t arg = 0;
            for (int ps = 0; ps < 4; ps++) {
                const uint8_t e = k_p25_half_rate_nibble[(ps << 2) | ns];
                uint32_t c = 0;
                for (int b = 0; b < 4; b++) {
                    c += llr_disagreement(d[4 * t + b], (e >> (3 - b)) & 1);
                }
                const uint32_t m = prev[ps] + c;
                if (m < best) {
                    best = m;
                    arg = (uint8_t)ps;
                }
            }
            cur[ns] = best;
            back[t][ns] = arg;
        }
        memcpy(prev, cur, sizeof(prev));
    }
    int st = 0;
    uint32_t best = cur[0];
    for (int j = 1; j < 4; j++) {
        if (cur[j] < best) {
            best = cur[j];
            st = j;
        }
    }
    uint8_t path[49];
    for (int t = 48; t >= 0; t--) {
        path[t] = (uint8_t)st;
        st = back[t][st];
    }
    for (int i = 0; i < 12; i++) {
        out12[i] = (uint8_t)((path[4 * i] << 6) | (path[4 * i + 1] << 4) | (path[4 * i + 2] << 2) | path[4 * i + 3]);
    }
    return (int)(best >> 8);
}

/* P25 1/2-rate list decoder, src/protocol/p25/p25_12.c:31-202: 8 survivors per state.  A state's survivor list is the
 * 8 smallest of the 32 extensions under the order (metric, predecessor state, predecessor rank) — which is what the
 * reference's "insert before the first strictly larger metric" gives when extensions arrive predecessor by
 * predecessor, rank by rank.  The 32 final paths are traced back in (state, rank) order, duplicates (same 12 bytes)
 * dropped, and kept sorted by metric with the same strict-less rule.  Returns the candidate count (<= max). */
int
orc_p25_12_soft_llr_list(const int16_t* llr196, uint8_t* out_bytes /* [max][12] */, uint32_t* out_metric, int max) {
    enum { K = 8 };
    if (max <= 0) {
        return 0;
    }
    if (max > K) {
        max = K;
    }
    uint8_t il[98];
    int16_t d[196];
    orc_trellis_interleave_98(il);
    memset(d, 0, sizeof(d));
    for (int i = 0; i < 98; i++) {
        d[2 * il[i]] = llr196[2 * i];
        d[2 * il[i] + 1] = llr196[2 * i + 1];
    }
    static uint8_t back[49][4][K];
    uint32_t prev[4][K], cur[4][K];
    for (int s = 0; s < 4; s++) {
        for (int r = 0; r < K; r++) {
            prev[s][r] = 0xFFFFFFFFu;
        }
        prev[s][0] = (s == 0) ? 0u : 256u;
    }
    memset(back, 0, sizeof(back));
    for (int t = 0; t < 49; t++) {
        for (int ns = 0; ns < 4; ns++) {
            uint32_t* cm = cur[ns];
            uint8_t* cb = back[t][ns];
            int n = 0; /* filled entries, sorted ascending; the rest is UINT32_MAX */
            for (int r = 0; r < K; r++) {
                cm[r] = 0xFFFFFFFFu;
            }
            for (int ps = 0; ps < 4; ps++) {
                const uint8_t e = k_p25_half_rate_nibble[(ps << 2) | ns];
                uint32_t c = 0;
                for (int b = 0; b < 4; b++) {
                    c += llr_disagreement(d[4 * t + b], (e >> (3 - b)) & 1);
                }
                for (int r = 0; r < K; r++) {
                    if (prev[ps][r] == 0xFFFFFFFFu) {
                        continue;
                    }
                    const uint32_t m = prev[ps][r] + c;
                    int at = -1;
                    for (int i = 0; i < K; i++) {
                        if (m < cm[i]) {
                            at = i;
                            break;
                        }
                    }
                    if (at < 0) {
                        continue;
                    }
                    for (int i = K - 1; i > at; i--) {
                        cm[i] = cm[i - 1];
                        cb[i] = cb[i - 1];
                    }
                    cm[at] = m;
                    cb[at] = (uint8_t)((ps << 3) | r);
                    (void)n;
                }
            }
        }
        memcpy(prev, cur, sizeof(prev));
    }
    int count = 0;
    for (int st = 0; st < 4; st++) {
        for (int rk = 0; rk < K; rk++) {
            if (prev[st][rk] == 0xFFFFFFFFu) {
                continue;
            }
            uint8_t path[49], bytes[12];
            int s = st, r = rk;
            for (int t = 48; t >= 0; t--) {
                path[t] = (uint8_t)s;
                const uint8_t p = back[t][s][r];
                s = (p >> 3) & 3;
                r = p & 7;
            }
            for (int i = 0; i < 12; i++) {
                bytes[i] = (uint8_t)((path[4 * i] << 6) | (path[4 * i + 1] << 4) | (path[4 * i + 2] << 2) | path[4 * i + 3]);
            }
            int dup = 0;
            for (int i = 0; i < count; i++) {
                if (memcmp(out_bytes + 12 * i, bytes, 12) == 0) {
                    dup = 1;
                    break;
                }
            }
            if (dup) {
                continue;
            }
            const uint32_t m = prev[st][rk];
            int at = count;
            for (int i = 0; i < count; i++) {
                if (m < out_metric[i]) {
                    at = i;
                    break;
                }
            }
            if (count < max) {
                count++;
            } else if (at >= max) {
                continue;
            }
            for (int i = count - 1; i > at; i--) {
                memcpy(out_bytes + 12 * i, out_bytes + 12 * (i - 1), 12);
                out_metric[i] = out_metric[i - 1];
            }
            memcpy(out_bytes + 12 * at, bytes, 12);
            out_metric[at] = m;
        }
    }
    return count;
}

/* 3/4-rate trellis, hard (reliab98 == NULL) or reliability-weighted; traceback from state 0. */
int
orc_r34_decode(const uint8_t* dibits98, const uint8_t* reliab98, uint8_t out18[18]) {
    enum { T = 49, S = 8, INF = 1000000000 };
    uint8_t il[98], dd[98], rr[98];
    orc_trellis_interleave_98(il);
    for (int i = 0; i < 98; i++) {
        dd[il[i]] = dibits98[i] & 3u;
        rr[il[i]] = reliab98 ? reliab98[i] : 1;
    }
    int prev[S], cur[S];
    uint8_t back[T][S];
    memset(back, 0, sizeof(back));
    for (int s = 0; s < S; s++) {
        prev[s] = INF;
    }
    prev[0] = 0;
    for (int t = 0; t < T; t++) {
        const uint8_t nib = (uint8_t)((dd[2 * t] << 2) | dd[2 * t + 1]);
        const uint8_t point = k_r34_nibble_to_point[nib];
        for (int s = 0; s < S; s++) {
            cur[s] = INF;
        }
        for (int ps = 0; ps < S; ps++) {
            if (prev[ps] >= INF) {
                continue;
            }
            for (int ns = 0; ns < S; ns++) {
                const uint8_t e = k_r34_fsm[ps * 8 + ns];
                int c;
                if (reliab98) {
                    const uint8_t x = (uint8_t)(k_r34_point_to_nibble[e] ^ nib);
                    c = ((x >> 3) & 1) * rr[2 * t] + ((x >> 2) & 1) * rr[2 * t] + ((x >> 1) & 1) * rr[2 * t + 1]
                        + (x & 1) * rr[2 * t + 1];
                } else {
                    c = __builtin_popcount((unsigned)((e ^ point) & 15));
                }
                const int m = prev[ps] + c;
                if (m < cur[ns]) {
                    cur[ns] = m;
                    back[t][ns] = (uint8_t)ps;
                }
            }
        }
        memcpy(prev, cur, sizeof(prev));
    }
    uint8_t path[T];
    int s = 0;
    for (int t = T - 1; t >= 0; t--) {
        path[t] = (uint8_t)s;
        s = back[t][s];
    }
    for (int g = 0; g < 6; g++) {
        uint32_t v = 0;
        for (int k = 0; k < 8; k++) {
            v = (v << 3) | (path[8 * g + k] & 7u);
        }
        out18[3 * g] = (uint8_t)(v >> 16);
        out18[3 * g + 1] = (uint8_t)(v >> 8);
        out18[3 * g + 2] = (uint8_t)v;
    }
    return 0;
}

/* 3/4-rate LIST decoder, src/protocol/dmr/dmr_34_viterbi.c:255-362,446-474: 32 survivors per state.  The reference
 * inserts an extension before the first survivor whose metric is >= its own, feeding extensions predecessor by
 * predecessor and rank by rank: the list therefore holds the 32 smallest extensions under the total order
 * (metric ascending, arrival index ps*32+pr DESCENDING).  Here each extension carries the 32-bit key
 * (metric << 8) | (255 - arrival index) and the 32 smallest keys are kept - all keys of a step are distinct, so the
 * order is unambiguous; metrics stay below 2^24 (49 steps x <= 1020).  Candidates are the survivors of state 0 in
 * list order (already non-decreasing in metric, so the reference's stable sort is the identity); no de-duplication.
 * out_metric[i], out_bytes[18*i..]; returns the count (<= max). */
int
orc_r34_decode_list(const uint8_t* dibits98, const uint8_t* reliab98, int max, int32_t* out_metric, uint8_t* out_bytes) {
    enum { T = 49, S = 8, K = 32 };
    const uint32_t EMPTY = 0xFFFFFFFFu;
    uint8_t il[98], dd[98], rr[98];
    orc_trellis_interleave_98(il);
    for (int i = 0; i < 98; i++) {
        dd[il[i]] = dibits98[i] & 3u;
        rr[il[i]] = reliab98 ? reliab98[i] : 1;
    }
    static uint32_t prev[S][K], cur[S][K];
    static uint8_t back[T][S][K];
    for (int s = 0; s < S; s++) {
        for (int r = 0; r < K; r++) {
            prev[s][r] = EMPTY;
        }
    }
    prev[0][0] = 0u << 8 | 255u; /* metric 0; arrival index is irrelevant for the seed */
    for (int t = 0; t < T; t++) {
        const uint8_t nib = (uint8_t)((dd[2 * t] << 2) | dd[2 * t + 1]);
        for (int ns = 0; ns < S; ns++) {
            uint32_t* c = cur[ns];
            for (int r = 0; r < K; r++) {
                c[r] = EMPTY;
            }
            for (int ps = 0; ps < S; ps++) {
                const uint8_t x = (uint8_t)(k_r34_point_to_nibble[k_r34_fsm[ps * 8 + ns]] ^ nib);
                int cost;
                if (reliab98) {
                    cost = ((x >> 3) & 1) * rr[2 * t] + ((x >> 2) & 1) * rr[2 * t] + ((x >> 1) & 1) * rr[2 * t + 1]
                           + (x & 1) * rr[2 * t + 1];
                } else {
                    cost = x ? 256 + __builtin_popcount(x) : 0;
                }
                for (int pr = 0; pr < K; pr++) {
                    if (prev[ps][pr] == EMPTY) {
                        continue;
                    }
                    const uint32_t m = (prev[ps][pr] >> 8) + (uint32_t)cost;
                    uint32_t key = (m << 8) | (uint32_t)(255 - (ps * K + pr));
                    /* sorted insert, keep the 32 smallest */
                    for (int i = 0; i < K; i++) {
                        if (key < c[i]) {
                            const uint32_t tmp = c[i];
                            c[i] = key;
                            key = tmp;
                        }
                    }
                }
            }
            for (int r = 0; r < K; r++) {
                back[t][ns][r] = (c[r] == EMPTY) ? 0 : (uint8_t)(255 - (c[r] & 255u));
            }
        }
        memcpy(prev, cur, sizeof(prev));
    }
    int count = 0;
    for (int r = 0; r < K && count < max; r++) {
        if (prev[0][r] == EMPTY) {
            continue;
        }
        uint8_t path[T];
        int s = 0, rk = r;
        for (int t = T - 1; t >= 0; t--) {
            path[t] = (uint8_t)s;
            const int idx = back[t][s][rk];
            s = idx >> 5;
            rk = idx & 31;
        }
        uint8_t* o = out_bytes + 18 * count;
        for (int g = 0; g < 6; g++) {
            uint32_t v = 0;
            for (int k = 0; k < 8; k++) {
                v = (v << 3) | (path[8 * g + k] & 7u);
            }
            o[3 * g] = (uint8_t)(v >> 16);
            o[3 * g + 1] = (uint8_t)(v >> 8);
            o[3 * g + 2] = (uint8_t)v;
        }
        out_metric[count] = (int32_t)(prev[0][r] >> 8);
        count++;
    }
    return count;
}

/* K=5 R=1/2, NXDN flavour: uint16 metrics that wrap, 16 decision bits per step, chainback from state 0.
 * sym = n_steps pairs (s0, s1) with values 0..2; rel (optional) = pairs (r0, r1).  metrics16 carries the
 * decoder's path metrics in and out (the reference keeps them in file-static storage across decodes).
 * out = ceil(n_bits/8) bytes, MSB-first, bit index i = chainback position (src/protocol/nxdn/nxdn_convolution.c:85-99). */
void
orc_nxdn_conv_decode(const uint8_t* sym, const uint8_t* rel, int n_steps, uint16_t metrics16[16], uint8_t* out,
                     int n_bits) {
    static const uint8_t B1[8] = {0, 0, 0, 0, 2, 2, 2, 2};
    static const uint8_t B2[8] = {0, 2, 2, 0, 0, 2, 2, 0};
    uint16_t a[16], b[16];
    uint16_t* oldm = a;
    uint16_t* newm = b;
    memcpy(a, metrics16, sizeof(a));
    uint16_t* dec = (uint16_t*)calloc((size_t)(n_steps > 0 ? n_steps : 1), sizeof(uint16_t));
    for (int t = 0; t < n_steps; t++) {
        const int s0 = sym[2 * t], s1 = sym[2 * t + 1];
        uint16_t word = 0;
        for (int i = 0; i < 8; i++) {
            const int j = 2 * i;
            int d0, d1;
            if (rel) {
                const uint32_t full = 8;
                uint32_t metric = ((uint32_t)abs((int)B1[i] - s0) * rel[2 * t] + (uint32_t)abs((int)B2[i] - s1) * rel[2 * t + 1]) / 128u;
                if (metric > full) {
                    metric = full;
                }
                uint32_t m0 = oldm[i] + metric, m1 = oldm[i + 8] + (full - metric);
                d0 = m0 >= m1;
                newm[j] = (uint16_t)(d0 ? m1 : m0);
                m0 = oldm[i] + (full - metric);
                m1 = oldm[i + 8] + metric;
                d1 = m0 >= m1;
                newm[j + 1] = (uint16_t)(d1 ? m1 : m0);
            } else {
                const uint16_t metric = (uint16_t)(abs((int)B1[i] - s0) + abs((int)B2[i] - s1));
                uint16_t m0 = (uint16_t)(oldm[i] + metric), m1 = (uint16_t)(oldm[i + 8] + (4u - metric));
                d0 = m0 >= m1;
                newm[j] = d0 ? m1 : m0;
                m0 = (uint16_t)(oldm[i] + (4u - metric));
                m1 = (uint16_t)(oldm[i + 8] + metric);
                d1 = m0 >= m1;
                newm[j + 1] = d1 ? m1 : m0;
            }
            word |= (uint16_t)((d1 << (j + 1)) | (d0 << j));
        }
        dec[t] = word;
        uint16_t* tmp = oldm;
        oldm = newm;
        newm = tmp;
    }
    memcpy(metrics16, oldm, 16 * sizeof(uint16_t));
    uint32_t state = 0;
    int t = n_steps;
    for (int nb = n_bits; nb-- > 0;) {
        --t;
        const uint32_t i = state >> 4;
        const uint8_t bit = (uint8_t)((dec[t] >> i) & 1);
        state = ((uint32_t)bit << 7) | (state >> 1);
        if (bit) {
            out[nb >> 3] |= (uint8_t)(0x80u >> (nb & 7));
        } else {
            out[nb >> 3] &= (uint8_t)~(0x80u >> (nb & 7));
        }
    }
    free(dec);
}

/* K=5 R=1/2, libM17 flavour: uint16 soft symbols (0 = strong 0 ... 0xFFFF = strong 1), uint32 metrics from 0,
 * history word per step, chainback from state 0 writing len/2 + 4 bit positions
 * (src/core/util/dsd_misc.c:118-145,180-270).  Returns the minimum final metric. */
uint32_t
orc_m17_viterbi_decode(uint8_t* out, const uint16_t* in, int len) {
    static const uint16_t C0[8] = {0, 0, 0, 0, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF};
    static const uint16_t C1[8] = {0, 0xFFFF, 0xFFFF, 0, 0, 0xFFFF, 0xFFFF, 0};
    uint32_t prev[16] = {0}, cur[16] = {0};
    uint16_t hist[244];
    memset(hist, 0, sizeof(hist));
    int pos = 0;
    for (int i = 0; i + 1 < len; i += 2) {
        const uint16_t s0 = in[i], s1 = in[i + 1];
        for (int k = 0; k < 8; k++) {
            const uint32_t metric = (uint32_t)(C0[k] > s0 ? C0[k] - s0 : s0 - C0[k])
                                    + (uint32_t)(C1[k] > s1 ? C1[k] - s1 : s1 - C1[k]);
            const uint32_t m0 = prev[k] + metric, m1 = prev[k + 8] + (0x1FFFEu - metric);
            const uint32_t m2 = prev[k] + (0x1FFFEu - metric), m3 = prev[k + 8] + metric;
            if (m0 >= m1) {
                hist[pos] |= (uint16_t)(1u << (2 * k));
                cur[2 * k] = m1;
            } else {
                cur[2 * k] = m0;
            }
            if (m2 >= m3) {
                hist[pos] |= (uint16_t)(1u << (2 * k + 1));
                cur[2 * k + 1] = m3;
            } else {
                cur[2 * k + 1] = m2;
            }
        }
        memcpy(prev, cur, sizeof(prev));
        pos++;
    }
    const int nbits = len / 2;
    uint8_t state = 0;
    int bitpos = nbits + 4;
    memset(out, 0, (size_t)((nbits - 1) / 8 + 1));
    while (pos > 0) {
        bitpos--;
        pos--;
        const int bit = hist[pos] & (1 << (state >> 4));
        state >>= 1;
        if (bit) {
            state |= 0x80;
            out[bitpos / 8] |= (uint8_t)(1 << (7 - (bitpos % 8)));
        }
    }
    uint32_t best = prev[0];
    for (int i = 1; i < 16; i++) {
        if (prev[i] < best) {
            best = prev[i];
        }
    }
    return best;
}

/* punctured wrapper: erased positions get the neutral value 0x7FFF (src/core/util/dsd_misc.c:147-178) */
uint32_t
orc_m17_viterbi_decode_punctured(uint8_t* out, const uint16_t* in, const uint8_t* punct, int in_len, int p_len) {
    uint16_t um[488];
    memset(um, 0, sizeof(um));
    int p = 0, u = 0, i = 0;
    while (i < in_len && u < 488) {
        if (punct[p]) {
            um[u] = in[i++];
        } else {
            um[u] = 0x7FFF;
        }
        u++;
        p = (p + 1) % p_len;
    }
    return orc_m17_viterbi_decode(out, um, u) - (uint32_t)(u - in_len) * 0x7FFFu;
}


/* ---- P25 Phase 1 confirmed data, rate 3/4 blocks with LLRs (src/protocol/p25/phase1/p25p1_mbf34.c) -------------------------------
 * TEST INFRASTRUCTURE.  p25_mbf34_decode_soft_list(:189-213): the 98 LLR pairs are de-interleaved (dsd_trellis_interleave_98), every
 * state starts with one survivor (metric 0 for state 0, 1024 otherwise), and at each of the 49 steps every target state keeps the 8
 * cheapest of the up to 64 extensions, an extension going in front of the first strictly more expensive one (:46-63) - so the order
 * is (metric, arrival), arrival = previous state major, rank minor (:139-151).  Branch cost = the four LLRs' disagreement with the
 * expected dibit pair of the transition (:123-127).  At the end the up to 64 paths are packed to 18 bytes from their first 48 states
 * (:65-75; the 49th is not part of the bytes, so paths can coincide) and merged into one list by the same insertion rule, a path whose
 * bytes are already IN the list being skipped (:85-109).  orc_p25_mbf34_best = p25_mbf34_decode_soft (:215-257): plain Viterbi, lowest
 * previous state on a tie, lowest final state on a tie; returns metric >> 8. */
typedef struct {
    uint32_t metric;
    uint8_t st[49];
    uint8_t on;
} mbf_path;

static uint32_t
mbf_cost(const int16_t* d, int step, int prev, int next) {
    const int expect = k_r34_point_to_nibble[k_r34_fsm[prev * 8 + next] & 15] & 15;
    uint32_t c = 0;
    for (int b = 0; b < 4; b++) {
        c += llr_disagreement(d[4 * step + b], (expect >> (3 - b)) & 1);
    }
    return c;
}

static void
mbf_deinterleave(const int16_t* llr196, int16_t* d) {
    uint8_t il[98];
    orc_trellis_interleave_98(il);
    memset(d, 0, 196 * sizeof(int16_t));
    for (int i = 0; i < 98; i++) {
        d[2 * il[i]] = llr196[2 * i];
        d[2 * il[i] + 1] = llr196[2 * i + 1];
    }
}

static void
mbf_pack(const uint8_t* st, uint8_t out[18]) {
    for (int g = 0; g < 6; g++) {
        uint32_t w = 0;
        for (int k = 0; k < 8; k++) {
            w = (w << 3) | st[8 * g + k];
        }
        out[3 * g] = (uint8_t)(w >> 16);
        out[3 * g + 1] = (uint8_t)(w >> 8);
        out[3 * g + 2] = (uint8_t)w;
    }
}

int
orc_p25_mbf34_list(const int16_t* llr196, int max, uint8_t* out_bytes, uint32_t* out_metric) {
    enum { K = 8 };
    if (!llr196 || !out_bytes || !out_metric || max <= 0) {
        return 0;
    }
    max = max > K ? K : max;
    int16_t d[196];
    mbf_deinterleave(llr196, d);
    static mbf_path a[8][K], b[8][K]; /* (not re-entrant: test infrastructure) */
    memset(a, 0, sizeof(a));
    for (int s = 0; s < 8; s++) {
        a[s][0].on = 1;
        a[s][0].metric = s == 0 ? 0u : 1024u;
    }
    for (int t = 0; t < 49; t++) {
        memset(b, 0, sizeof(b));
        for (int ps = 0; ps < 8; ps++) {
            for (int r = 0; r < K; r++) {
                if (!a[ps][r].on) {
                    continue;
                }
                for (int ns = 0; ns < 8; ns++) {
                    mbf_path c = a[ps][r];
                    c.metric += mbf_cost(d, t, ps, ns);
                    c.st[t] = (uint8_t)ns;
                    int at = -1;
                    for (int i = 0; i < K; i++) {
                        if (!b[ns][i].on || c.metric < b[ns][i].metric) {
                            at = i;
                            break;
                        }
                    }
                    if (at < 0) {
                        continue;
                    }
                    for (int i = K - 1; i > at; i--) {
                        b[ns][i] = b[ns][i - 1];
                    }
                    b[ns][at] = c;
                }
            }
        }
        memcpy(a, b, sizeof(a));
    }
    int n = 0;
    for (int s = 0; s < 8; s++) {
        for (int r = 0; r < K; r++) {
            if (!a[s][r].on) {
                continue;
            }
            uint8_t by[18];
            mbf_pack(a[s][r].st, by);
            int dup = 0;
            for (int i = 0; i < n; i++) {
                dup |= memcmp(out_bytes + 18 * i, by, 18) == 0;
            }
            if (dup) {
                continue;
            }
            int at = n;
            for (int i = 0; i < n; i++) {
                if (a[s][r].metric < out_metric[i]) {
                    at = i;
                    break;
                }
            }
            if (n < max) {
                n++;
            } else if (at >= max) {
                continue;
            }
            for (int i = n - 1; i > at; i--) {
                memcpy(out_bytes + 18 * i, out_bytes + 18 * (i - 1), 18);
                out_metric[i] = out_metric[i - 1];
            }
            memcpy(out_bytes + 18 * at, by, 18);
            out_metric[at] = a[s][r].metric;
        }
    }
    return n;
}

int
orc_p25_mbf34_best(const int16_t* llr196, uint8_t out18[18]) {
    int16_t d[196];
    mbf_deinterleave(llr196, d);
    uint32_t pm[8], cm[8];
    uint8_t bp[49][8];
    for (int s = 0; s < 8; s++) {
        pm[s] = s == 0 ? 0u : 1024u;
    }
    for (int t = 0; t < 49; t++) {
        for (int ns = 0; ns < 8; ns++) {
            uint32_t best = 0xFFFFFFFFu;
            int bpv = 0;
            for (int ps = 0; ps < 8; ps++) {
                const uint32_t m = pm[ps] + mbf_cost(d, t, ps, ns);
                if (m < best) {
                    best = m;
                    bpv = ps;
                }
            }
            cm[ns] = best;
            bp[t][ns] = (uint8_t)bpv;
        }
        memcpy(pm, cm, sizeof(pm));
    }
    int st = 0;
    for (int s = 1; s < 8; s++) {
        if (cm[s] < cm[st]) {
            st = s;
        }
    }
    const uint32_t best_final = cm[st];
    uint8_t tri[49];
    for (int t = 48; t >= 0; t--) {
        tri[t] = (uint8_t)st;
        st = bp[t][st];
    }
    mbf_pack(tri, out18);
    return (int)(best_final >> 8);
}
