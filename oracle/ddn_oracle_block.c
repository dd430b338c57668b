/*
 * oracle/ddn_oracle_block.c — CPU restatement of the block codes on the P25 Phase 1 path
 * (TEST INFRASTRUCTURE ONLY; integer arithmetic, bit-exact target).
 *
 *   BCH(63,16,11) over GF(2^6), x^6+x+1      include/dsd-neo/fec/BCH_63_16.hpp:47-330
 *   P25p1 NID decode: hard + NAC retry + Chase   src/protocol/p25/phase1/p25p1_check_nid.cpp:200-354
 *   Hamming(10,6,3)                          src/fec/hamming_10_6_3.cpp:20-105
 *   IMBE de-interleave of one LDU voice frame    src/protocol/p25/phase1/p25p1_ldu.c:27-48,89-120
 *   low-speed-data cyclic (16,8) code, hard + soft  src/protocol/p25/p25_lsd.c:31-160
 *
 * The BCH decoder is a bounded-distance decoder: for a received word within 11 bits of a codeword every correct
 * Berlekamp-Massey formulation returns that codeword and the same error count; beyond that the connection
 * polynomial of length L <= 11 is still unique (2L <= 22 syndromes) so "number of roots == L" fails or succeeds
 * identically.  This file uses Massey's polynomial-domain form (not the reference's index-form table walk);
 * tests/test_oracle_block.py pins it to the compiled reference on 0..16-error patterns.
 */
#include "ddn_oracle.h"

#include <stdlib.h>
#include <string.h>

static uint8_t gexp[128], glog[64];
static int gf_ready = 0;

static void
gf_init(void) {
    if (gf_ready) {
        return;
    }
    int x = 1;
    for (int i = 0; i < 63; i++) {
        gexp[i] = (uint8_t)x;
        gexp[i + 63] = (uint8_t)x;
        glog[x] = (uint8_t)i;
        x <<= 1;
        if (x & 64) {
            x ^= 0x43; /* x^6 = x + 1 */
        }
    }
    gexp[126] = gexp[0];
    gf_ready = 1;
}

static inline int
gmul(int a, int b) {
    return (a && b) ? gexp[glog[a] + glog[b]] : 0;
}

static inline int
gdiv(int a, int b) { /* b != 0 */
    return a ? gexp[glog[a] + 63 - glog[b]] : 0;
}

/* in63: bit per byte, data bits first (positions 0..15), parity 16..62.  Returns 1 on success. */
int
orc_bch_63_16_decode(const uint8_t in63[63], uint8_t out16[16], int* err_count) {
    gf_init();
    uint8_t r[63]; /* r[j] multiplies alpha^(i*j): the reference reverses the input (BCH_63_16.hpp:84-90) */
    for (int j = 0; j < 63; j++) {
        r[j] = in63[62 - j] ? 1 : 0;
    }
    int S[23];
    int any = 0;
    for (int i = 1; i <= 22; i++) {
        int s = 0;
        for (int j = 0; j < 63; j++) {
            if (r[j]) {
                s ^= gexp[(i * j) % 63];
            }
        }
        S[i] = s;
        any |= s;
    }
    *err_count = 0;
    if (any) {
        int C[24] = {1}, B[24] = {1}, T[24];
        int L = 0, m = 1, b = 1;
        for (int n = 0; n < 22; n++) {
            int d = S[n + 1];
            for (int i = 1; i <= L; i++) {
                d ^= gmul(C[i], S[n + 1 - i]);
            }
            if (d == 0) {
                m++;
                continue;
            }
            const int f = gdiv(d, b);
            if (2 * L <= n) {
                memcpy(T, C, sizeof(T));
                for (int i = 0; i + m < 24; i++) {
                    C[i + m] ^= gmul(f, B[i]);
                }
                L = n + 1 - L;
                memcpy(B, T, sizeof(B));
                b = d;
                m = 1;
            } else {
                for (int i = 0; i + m < 24; i++) {
                    C[i + m] ^= gmul(f, B[i]);
                }
                m++;
            }
            if (L > 11) {
                return 0;
            }
        }
        int count = 0;
        int loc[11];
        for (int i = 1; i <= 63; i++) { /* Chien: root alpha^i <-> error at position 63 - i */
            int q = 0;
            for (int k = 0; k <= L; k++) {
                if (C[k]) {
                    q ^= gexp[(glog[C[k]] + i * k) % 63];
                }
            }
            if (q == 0) {
                if (count >= 11) {
                    break;
                }
                loc[count++] = (63 - i) % 63;
            }
        }
        if (count != L) {
            return 0;
        }
        for (int k = 0; k < count; k++) {
            r[loc[k]] ^= 1;
        }
        *err_count = count;
    }
    for (int i = 0; i < 16; i++) {
        out16[i] = r[62 - i];
    }
    return 1;
}

/* ---- P25p1 NID -------------------------------------------------------------------------------------- */
enum { NID_FAIL = 0, NID_OK = 1, NID_PARITY_OVERRIDE = 2 };

typedef struct {
    int status, nac, duid, errs;
} nid_res;

static nid_res
nid_codeword(const uint8_t code[63], int parity, int* bch_failed) {
    /* DUIDs defined by TIA-102.BAAA-A table 8-4; parity bit is 1 only for LDU1 (5) and LDU2 (0xA) */
    static const uint8_t duid_ok[16] = {1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 0, 1};
    nid_res r = {NID_FAIL, 0, 0, 0};
    uint8_t d[16];
    int errs = 0;
    if (bch_failed) {
        *bch_failed = 0;
    }
    if (!orc_bch_63_16_decode(code, d, &errs)) {
        if (bch_failed) {
            *bch_failed = 1;
        }
        return r;
    }
    r.errs = errs;
    for (int i = 0; i < 12; i++) {
        r.nac = (r.nac << 1) | d[i];
    }
    r.duid = (d[12] << 3) | (d[13] << 2) | (d[14] << 1) | d[15];
    if (!duid_ok[r.duid]) {
        r.errs = 0;
        return r;
    }
    const int want = (r.duid == 5 || r.duid == 10) ? 1 : 0;
    r.status = (want == parity) ? NID_OK : NID_PARITY_OVERRIDE;
    return r;
}

static int
rx_nac(const uint8_t code[63]) {
    int n = 0;
    for (int i = 0; i < 12; i++) {
        n = (n << 1) | (code[i] ? 1 : 0);
    }
    return n;
}

static void
put_nac(uint8_t code[63], int nac) {
    for (int i = 0; i < 12; i++) {
        code[i] = (uint8_t)((nac >> (11 - i)) & 1);
    }
}

static int
nac_usable(int nac) {
    return nac > 0 && nac < 0xFFF;
}

typedef struct {
    int found;
    nid_res dec;
    int score, changes;
} chase_best;

static void
chase_from(const uint8_t base[63], const uint8_t* rel, const int* pool, int np, int parity, int parity_rel,
           int threshold, chase_best* best) {
    for (int mask = 0; mask < (1 << np); mask++) {
        if (__builtin_popcount((unsigned)mask) > 3) {
            continue;
        }
        uint8_t cand[63];
        memcpy(cand, base, 63);
        int changed = 0, score = 0;
        for (int b = 0; b < np; b++) {
            if (mask & (1 << b)) {
                cand[pool[b]] ^= 1;
                changed++;
                score += rel[pool[b]];
            }
        }
        if (changed && score > threshold * changed) {
            continue;
        }
        nid_res dec = nid_codeword(cand, parity, 0);
        if (dec.status <= 0) {
            continue;
        }
        int sc = score + (dec.status == NID_PARITY_OVERRIDE ? parity_rel : 0);
        int better = !best->found || sc < best->score
                     || (sc == best->score && dec.status == NID_OK && best->dec.status != NID_OK)
                     || (sc == best->score && dec.status == best->dec.status && dec.errs < best->dec.errs)
                     || (sc == best->score && dec.status == best->dec.status && dec.errs == best->dec.errs
                         && changed < best->changes);
        if (better) {
            best->found = 1;
            best->dec = dec;
            best->score = sc;
            best->changes = changed;
        }
    }
}

/* out4 = {status, nac, duid, error_count}; rel63 may be NULL (hard only).  threshold = erasure threshold
 * (p25p1_get_erasure_threshold(), default 64: src/protocol/p25/phase1/p25p1_soft.cpp:21-40). */
void
orc_p25p1_nid_decode(const uint8_t code[63], const uint8_t* rel63, int observed_nac, int parity, int parity_rel,
                     int threshold, int out4[4]) {
    int failed = 0;
    nid_res hard = nid_codeword(code, parity, &failed);
    if (hard.status == NID_FAIL && failed && nac_usable(observed_nac) && rx_nac(code) != observed_nac) {
        uint8_t retry[63];
        memcpy(retry, code, 63);
        put_nac(retry, observed_nac);
        hard = nid_codeword(retry, parity, 0);
    }
    nid_res res = hard;
    if (hard.status <= 0 && rel63) {
        /* pool: up to 8 least-reliable positions below the threshold, topped up to 6 (stable order by
         * (reliability, index)) — p25p1_check_nid.cpp:118-150 */
        int order[63];
        for (int i = 0; i < 63; i++) {
            order[i] = i;
        }
        for (int i = 0; i < 63; i++) {
            for (int j = i + 1; j < 63; j++) {
                const int a = order[j], b = order[i];
                const int c = (rel63[a] != rel63[b]) ? ((int)rel63[a] - (int)rel63[b]) : (a - b);
                if (c < 0) {
                    order[i] = a;
                    order[j] = b;
                }
            }
        }
        int pool[8], np = 0, sel[63] = {0};
        for (int i = 0; i < 63 && np < 8; i++) {
            if (rel63[order[i]] < threshold) {
                pool[np++] = order[i];
                sel[i] = 1;
            }
        }
        for (int i = 0; i < 63 && np < 6; i++) {
            if (!sel[i]) {
                pool[np++] = order[i];
            }
        }
        if (np > 0) {
            chase_best best = {0, {NID_FAIL, 0, 0, 0}, 0, 0};
            chase_from(code, rel63, pool, np, parity, parity_rel, threshold, &best);
            if (nac_usable(observed_nac) && rx_nac(code) != observed_nac) {
                uint8_t retry[63];
                memcpy(retry, code, 63);
                put_nac(retry, observed_nac);
                chase_from(retry, rel63, pool, np, parity, parity_rel, threshold, &best);
            }
            if (best.found) {
                res = best.dec;
            }
        }
    }
    out4[0] = res.status;
    out4[1] = res.nac;
    out4[2] = res.duid;
    out4[3] = res.errs;
}

/* ---- Hamming(10,6,3), src/fec/hamming_10_6_3.cpp:20-58,90-105 ------------------------------------------ */
/* word10 = (6 data bits << 4) | 4 parity bits.  Returns error count 0/1/2; *fixed6 = corrected data. */
int
orc_hamming_10_6_3(int word10, int* fixed6) {
    static const int8_t bad_bit[16] = {-2, 0, 1, 5, 2, -1, -1, 6, 3, -1, -1, 7, 4, 8, 9, -1};
    static const int mask[4] = {0x398, 0x354, 0x2E2, 0x1E1};
    int syn = 0;
    for (int k = 0; k < 4; k++) {
        syn = (syn << 1) | (__builtin_popcount((unsigned)(word10 & mask[k])) & 1);
    }
    int fixed = word10, errs = 0;
    if (syn) {
        const int bb = bad_bit[syn];
        if (bb < 0) {
            errs = 2;
        } else {
            errs = 1;
            if (bb >= 4) {
                fixed ^= 1 << bb;
            }
        }
    }
    *fixed6 = fixed >> 4;
    return errs;
}


/* ---- IMBE de-interleave (process_IMBE, src/protocol/p25/phase1/p25p1_ldu.c:89-120) ---------------------------------
 * The 144 bits of the eight IMBE code vectors c0..c3 (23 bits), c4..c6 (15), c7 (7), laid end to end most significant
 * index first, are sent twelve bits per row r = 0..11: columns 0..5 carry stream bits 24k + 2r (k = column), columns
 * 6..11 carry 24k + 2r + 1 for k = 1,0,3,2,5,4 (the schedule behind the p25p1_imbe_interleave_{w,x,y,z} tables of
 * include/dsd-neo/protocol/p25/p25p1_const.h:30-47; tests pin this formula to those tables).  Dibit j holds
 * transmitted bits 2j (high) and 2j+1 (low).  A status symbol is skipped whenever the running dibit counter shows 35
 * (p25p1_ldu.c:27-39).  Soft bit = {hard bit, min(|llr|, 255)} (include/dsd-neo/core/vocoder.h:30-38). */
static void
imbe_cell(int p, int* vec, int* idx) {
    static const int len[8] = {23, 23, 23, 23, 15, 15, 15, 7};
    const int r = p / 12, m = p % 12;
    int L = (m < 6) ? 24 * m + 2 * r : 24 * ((m - 6) ^ 1) + 2 * r + 1;
    int v = 0;
    while (L >= len[v]) {
        L -= len[v];
        v++;
    }
    *vec = v;
    *idx = len[v] - 1 - L;
}

int
orc_p25p1_imbe_deinterleave(const uint8_t* dibits, const int16_t* llr0, const int16_t* llr1, long n_avail,
                            int status_count, uint8_t fr[8][23], uint8_t soft[8][23][2], int* status_count_out,
                            int* consumed) {
    memset(fr, 0, 8 * 23);
    memset(soft, 0, 8 * 23 * 2);
    long pos = 0;
    for (int j = 0; j < 72; j++) {
        if (status_count == 35) {
            pos++; /* mid-frame status symbol */
            status_count = 1;
        } else {
            status_count++;
        }
        if (pos >= n_avail) {
            return -1;
        }
        const int d = dibits[pos];
        const int l[2] = {llr0[pos], llr1[pos]};
        pos++;
        for (int h = 0; h < 2; h++) {
            int v, i;
            imbe_cell(2 * j + h, &v, &i);
            const int bit = h == 0 ? ((d >> 1) & 1) : (d & 1);
            int rel = l[h] < 0 ? -l[h] : l[h];
            rel = rel > 255 ? 255 : rel;
            fr[v][i] = (uint8_t)bit;
            soft[v][i][0] = (uint8_t)bit;
            soft[v][i][1] = (uint8_t)rel;
        }
    }
    *status_count_out = status_count;
    *consumed = (int)pos;
    /* is_non_standard_c0_word(), p25p1_ldu.c:55-66: c0 == 0x000070 read index 0 first */
    int ns = 1;
    for (int i = 0; i < 23; i++) {
        ns &= (fr[0][i] == ((i >= 15 && i <= 17) ? 1 : 0));
    }
    return ns;
}


/* ---- P25p1 low speed data: (16,8) cyclic code (src/protocol/p25/p25_lsd.c) --------------------------------------------
 * Systematic cyclic code with g(x) = x^8 + x^5 + x^4 + x^3 + 1: parity byte = (data(x) * x^8) mod g(x) (this is what the
 * reference's 256-entry lsd_parity table holds; tests pin all 256 values).  Decoder: zero syndrome, a one-hot syndrome
 * (single parity-bit error) or the syndrome of a single data-bit error are accepted / corrected, anything else is
 * uncorrectable.  bits16 = data bits MSB first, then parity bits MSB first.  Soft variant: hard decode first, then every
 * non-empty subset of the <= 6 least reliable bits with |llr| below the erasure threshold (64; ties by position), the
 * cheapest flip set (strictly smaller penalty wins, subsets in mask order) that hard-decodes is taken. */
static int
lsd_parity_of(int data) {
    int v = data << 8;
    for (int i = 15; i >= 8; i--) {
        if (v & (1 << i)) {
            v ^= 0x139 << (i - 8);
        }
    }
    return v & 0xFF;
}

int
orc_p25_lsd_parity(int data) {
    return lsd_parity_of(data & 0xFF);
}

int
orc_p25_lsd_fec_16x8(uint8_t* bits16) {
    int data = 0, parity = 0;
    for (int i = 0; i < 8; i++) {
        data = (data << 1) | (bits16[i] & 1);
        parity = (parity << 1) | (bits16[8 + i] & 1);
    }
    const int synd = parity ^ lsd_parity_of(data);
    if (synd == 0) {
        return 1;
    }
    if ((synd & (synd - 1)) == 0) {
        int b = 7;
        while (!(synd & (1 << b))) {
            b--;
        }
        bits16[8 + (7 - b)] = (uint8_t)(1 - (bits16[8 + (7 - b)] & 1));
        return 1;
    }
    for (int pos = 0; pos < 8; pos++) {
        if (lsd_parity_of(1 << (7 - pos)) == synd) {
            bits16[pos] = (uint8_t)(1 - (bits16[pos] & 1));
            return 1;
        }
    }
    return 0;
}

int
orc_p25_lsd_fec_16x8_soft(uint8_t* bits16, const int16_t* llr16) {
    if (orc_p25_lsd_fec_16x8(bits16)) {
        return 1;
    }
    if (!llr16) {
        return 0;
    }
    int cand[16], nc = 0;
    for (int i = 0; i < 16; i++) {
        const int r = llr16[i] < 0 ? -(int)llr16[i] : (int)llr16[i];
        if (r < 64) {
            cand[nc++] = i;
        }
    }
    for (int i = 0; i < nc; i++) {
        for (int j = i + 1; j < nc; j++) {
            const int ri = abs((int)llr16[cand[i]]), rj = abs((int)llr16[cand[j]]);
            if (rj < ri || (rj == ri && cand[j] < cand[i])) {
                const int t = cand[i];
                cand[i] = cand[j];
                cand[j] = t;
            }
        }
    }
    if (nc > 6) {
        nc = 6;
    }
    if (nc <= 0) {
        return 0;
    }
    uint8_t best[16];
    int best_pen = 999999, found = 0;
    for (int mask = 1; mask < (1 << nc); mask++) {
        uint8_t tmp[16];
        memcpy(tmp, bits16, 16);
        int pen = 0;
        for (int b = 0; b < nc; b++) {
            if (mask & (1 << b)) {
                tmp[cand[b]] ^= 1u;
                pen += abs((int)llr16[cand[b]]);
            }
        }
        if (pen >= best_pen) {
            continue;
        }
        if (orc_p25_lsd_fec_16x8(tmp)) {
            memcpy(best, tmp, 16);
            best_pen = pen;
            found = 1;
        }
    }
    if (!found) {
        return 0;
    }
    memcpy(bits16, best, 16);
    return 1;
}


/* ---- CRC-CCITT16 of TSBK / LCCH blocks (src/protocol/p25/p25_crc.c:18-76): polynomial 0x1021, zero start, one bit per
 * step MSB first, inverted at the end; the block is good when the 16 bits that follow the payload equal it.
 * bytes = payload_bytes of payload followed by the two CRC bytes; returns 0 good / 65535 bad like crc16_lb_bridge
 * (whose helper returns (uint16_t)-1). */
int
orc_p25_crc16_ok(const uint8_t* bytes, int payload_bytes) {
    unsigned crc = 0;
    for (int i = 0; i < payload_bytes * 8; i++) {
        const unsigned bit = (bytes[i >> 3] >> (7 - (i & 7))) & 1u;
        crc = (((crc >> 15) & 1u) ^ bit) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
    }
    crc ^= 0xFFFFu;
    const unsigned rx = ((unsigned)bytes[payload_bytes] << 8) | bytes[payload_bytes + 1];
    return crc == rx ? 0 : 65535;
}
