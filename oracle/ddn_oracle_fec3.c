/*
 * oracle/ddn_oracle_fec3.c - TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU restatement of the DMR / NXDN block codes
 * of src/fec/fec.c (Hamming (7,4) (12,8) (13,9) (15,11) (16,11,4), Golay (20,8) (24,12), QR (16,7,6)), src/fec/bptc.c
 * (BPTC (196,96)) and src/fec/rs-12-9.c.  Pinned: tests/test_oracle_fec3.py compares every function with the reference's
 * own objects (oracle/_ref) - exhaustively over all 2^n words for n <= 16, over all error patterns of weight <= 4 plus random
 * words for the 20 / 24-bit codes, and on the KATs of tests/fec/test_fec_block_codes.c / test_fec_bptc_rs.c.
 */
#include <string.h>

#include "ddn_oracle.h"
#include "ddn_tables_fec3.h"

static int
syn_bits(const uint8_t* rx, const uint32_t* H, int n, int r) {
    int s = 0;
    for (int is = 0; is < r; is++) {
        int acc = 0;
        for (int j = 0; j < n; j++) {
            acc += rx[j] * (int)((H[is] >> j) & 1u);
        }
        s += (acc % 2) << (r - 1 - is);
    }
    return s;
}

static int
syn_mask(uint32_t w, const uint32_t* H, int r) {
    int s = 0;
    for (int is = 0; is < r; is++) {
        s |= (__builtin_popcount(w & H[is]) & 1) << (r - 1 - is);
    }
    return s;
}

/* single-error tables: position = the column equal to the syndrome (the explicit lists of Hamming_*_init, fec.c:133-143,
 * 183-198, 248-264, 312-330, 384-403) */
static void
hamming_table(uint8_t* tab, int size, const uint32_t* H, int n, int r) {
    memset(tab, 0xFF, (size_t)size);
    for (int p = 0; p < n; p++) {
        tab[syn_mask(1u << p, H, r)] = (uint8_t)p;
    }
}

/* fec.c:145-170 */
int
orc_hamming_7_4_decode(uint8_t* rx) {
    uint8_t tab[8];
    hamming_table(tab, 8, ddn_hamming_7_4_H, 7, 3);
    const int s = syn_bits(rx, ddn_hamming_7_4_H, 7, 3);
    if (s > 0) {
        if (tab[s] == 0xFF) {
            return 0;
        }
        rx[tab[s]] ^= 1;
    }
    return 1;
}

/* Hamming_12_8 / 13_9 / 15_11 / 16_11_4 _decode (fec.c:200-431).  which = 0..3.  As in the reference the corrected
 * position indexes rxBits without the code word's offset, (12,8) does not stop at an uncorrectable word, the others do. */
int
orc_hamming_multi_decode(int which, uint8_t* rx, uint8_t* dec, int nb) {
    static const int N[4] = {12, 13, 15, 16}, K[4] = {8, 9, 11, 11}, R[4] = {4, 4, 4, 5};
    const uint32_t* Hs[4] = {ddn_hamming_12_8_H, ddn_hamming_13_9_H, ddn_hamming_15_11_H, ddn_hamming_16_11_4_H};
    const int n = N[which], k = K[which], r = R[which];
    uint8_t tab[32];
    hamming_table(tab, 1 << r, Hs[which], n, r);
    int ok = 1;
    for (int ic = 0; ic < nb; ic++) {
        const int s = syn_bits(rx + n * ic, Hs[which], n, r);
        if (s > 0) {
            if (tab[s] == 0xFF) {
                ok = 0;
                if (which != 0) {
                    break;
                }
            } else {
                rx[tab[s]] ^= 1;
            }
        }
        if (dec) {
            memcpy(dec + k * ic, rx + n * ic, (size_t)k);
        }
    }
    return ok;
}

/* Golay_20_8_init / Golay_24_12_init (fec.c:437-512, 566-640): the reference's loops, slot by slot */
static void
golay_table(uint8_t tab[4096][3], const uint32_t* H, int kd) {
    memset(tab, 0xFF, 4096 * 3);
    for (int i1 = 0; i1 < kd; i1++) {
        for (int i2 = i1 + 1; i2 < kd; i2++) {
            for (int i3 = i2 + 1; i3 < kd; i3++) {
                const int s = syn_mask((1u << i1) | (1u << i2) | (1u << i3), H, 12);
                tab[s][0] = (uint8_t)i1;
                tab[s][1] = (uint8_t)i2;
                tab[s][2] = (uint8_t)i3;
            }
            const int s = syn_mask((1u << i1) | (1u << i2), H, 12);
            tab[s][0] = (uint8_t)i1;
            tab[s][1] = (uint8_t)i2;
            for (int ip = 0; ip < 12; ip++) {
                const int sp = s ^ (1 << (11 - ip));
                tab[sp][0] = (uint8_t)i1;
                tab[sp][1] = (uint8_t)i2;
                tab[sp][2] = (uint8_t)(kd + ip);
            }
        }
        const int s = syn_mask(1u << i1, H, 12);
        tab[s][0] = (uint8_t)i1;
        for (int ip1 = 0; ip1 < 12; ip1++) {
            const int s1 = s ^ (1 << (11 - ip1));
            tab[s1][0] = (uint8_t)i1;
            tab[s1][1] = (uint8_t)(kd + ip1);
            for (int ip2 = ip1 + 1; ip2 < 12; ip2++) {
                const int s2 = s1 ^ (1 << (11 - ip2));
                tab[s2][0] = (uint8_t)i1;
                tab[s2][1] = (uint8_t)(kd + ip1);
                tab[s2][2] = (uint8_t)(kd + ip2);
            }
        }
    }
    for (int ip1 = 0; ip1 < 12; ip1++) {
        const int s1 = 1 << (11 - ip1);
        tab[s1][0] = (uint8_t)(kd + ip1);
        for (int ip2 = ip1 + 1; ip2 < 12; ip2++) {
            const int s2 = s1 ^ (1 << (11 - ip2));
            tab[s2][0] = (uint8_t)(kd + ip1);
            tab[s2][1] = (uint8_t)(kd + ip2);
            for (int ip3 = ip2 + 1; ip3 < 12; ip3++) {
                const int s3 = s2 ^ (1 << (11 - ip3));
                tab[s3][0] = (uint8_t)(kd + ip1);
                tab[s3][1] = (uint8_t)(kd + ip2);
                tab[s3][2] = (uint8_t)(kd + ip3);
            }
        }
    }
}

/* Golay_20_8_decode (fec.c:514-561) / Golay_24_12_decode (fec.c:656-690); n = 20 or 24 */
int
orc_golay_dmr_decode(int n, uint8_t* rx) {
    static uint8_t t20[4096][3], t24[4096][3];
    static int ready;
    if (!ready) {
        golay_table(t20, ddn_golay_20_8_H, 8);
        golay_table(t24, ddn_golay_24_12_H, 12);
        ready = 1;
    }
    uint8_t(*tab)[3] = n == 20 ? t20 : t24;
    const int s = syn_bits(rx, n == 20 ? ddn_golay_20_8_H : ddn_golay_24_12_H, n, 12);
    if (s > 0) {
        int i = 0, corrections = 0;
        for (; i < 3; i++) {
            if (tab[s][i] == 0xFF) {
                break;
            }
            rx[tab[s][i]] ^= 1;
            corrections++;
        }
        if (i == 0) {
            return 0;
        }
        if (n == 20 && corrections > 2) {
            return 0;
        }
    }
    return 1;
}

/* QR_16_7_6_init + _decode (fec.c:744-822) */
int
orc_qr_16_7_6_decode(uint8_t* rx) {
    static uint8_t tab[512][2];
    static int ready;
    const uint32_t* H = ddn_qr_16_7_6_H;
    if (!ready) {
        memset(tab, 0xFF, sizeof(tab));
        for (int i1 = 0; i1 < 7; i1++) {
            for (int i2 = i1 + 1; i2 < 7; i2++) {
                const int s = syn_mask((1u << i1) | (1u << i2), H, 9);
                tab[s][0] = (uint8_t)i1;
                tab[s][1] = (uint8_t)i2;
            }
            const int s = syn_mask(1u << i1, H, 9);
            tab[s][0] = (uint8_t)i1;
            for (int ip = 0; ip < 9; ip++) {
                const int sp = s ^ (1 << (8 - ip));
                tab[sp][0] = (uint8_t)i1;
                tab[sp][1] = (uint8_t)(7 + ip);
            }
        }
        for (int ip1 = 0; ip1 < 9; ip1++) {
            const int s1 = 1 << (8 - ip1);
            tab[s1][0] = (uint8_t)(7 + ip1);
            for (int ip2 = ip1 + 1; ip2 < 9; ip2++) {
                const int s2 = s1 ^ (1 << (8 - ip2));
                tab[s2][0] = (uint8_t)(7 + ip1);
                tab[s2][1] = (uint8_t)(7 + ip2);
            }
        }
        ready = 1;
    }
    const int s = syn_bits(rx, H, 16, 9);
    if (s > 0) {
        int i = 0;
        for (; i < 2; i++) {
            if (tab[s][i] == 0xFF) {
                break;
            }
            rx[tab[s][i]] ^= 1;
        }
        if (i == 0) {
            return 0;
        }
    }
    return 1;
}

/* BPTCDeInterleaveDMRData + BPTC_196x96_Extract_Data (bptc.c:27-160).  col_corrected[] is uninitialised in the reference
 * when the very first column fails; zeros here. */
static int g_bptc_col0_failed; /* the last call hit the reference's uninitialised read (first column uncorrectable) */
int
orc_bptc_last_col0_failed(void) {
    return g_bptc_col0_failed;
}

uint32_t
orc_bptc_196x96(const uint8_t* in196, int deinterleave, uint8_t out96[96], uint8_t r3[3]) {
    uint8_t d[196], m[13][15];
    g_bptc_col0_failed = 0;
    if (deinterleave) {
        for (int i = 0; i < 196; i++) {
            d[(i * 13) % 196] = in196[i] & 1; /* BPTCDeInterleavingIndex[i] = 13 i mod 196 (bptc.c:21-31) */
        }
    } else {
        memcpy(d, in196, 196);
    }
    int k = 1;
    for (int i = 0; i < 13; i++) {
        for (int j = 0; j < 15; j++) {
            m[i][j] = d[k++] & 1;
        }
    }
    uint32_t errs = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t e = 0;
        uint8_t line[15], lc[11];
        memset(lc, 0, sizeof(lc));
        for (int i = 0; i < 9; i++) {
            memcpy(line, m[i], 15);
            if (!orc_hamming_multi_decode(2, line, lc, 1)) {
                e++;
            }
            memcpy(m[i], lc, 11);
        }
        uint8_t col[13], cc[9];
        memset(cc, 0, sizeof(cc));
        for (int i = 0; i < 15; i++) {
            for (int j = 0; j < 13; j++) {
                col[j] = m[j][i];
            }
            if (!orc_hamming_multi_decode(1, col, cc, 1)) {
                e++;
                if (i == 0) {
                    g_bptc_col0_failed = 1;
                }
            }
            for (int j = 0; j < 9; j++) {
                m[j][i] = cc[j];
            }
        }
        if (pass == 1) {
            errs = e;
        }
    }
    k = 0;
    for (int i = 3; i < 11; i++) {
        out96[k++] = m[0][i];
    }
    for (int i = 1; i < 9; i++) {
        for (int j = 0; j < 11; j++) {
            out96[k++] = m[i][j];
        }
    }
    r3[0] = m[0][2];
    r3[1] = m[0][1];
    r3[2] = m[0][0];
    return errs;
}

/* rs-12-9.c: GF(256) mod x^8+x^4+x^3+x^2+1; exp[255] = 1 and log[0] = 0 as in its tables (:33-62) */
static uint8_t g_exp[256], g_log[256];
static void
gf_init(void) {
    if (g_exp[0]) {
        return;
    }
    unsigned x = 1;
    for (int i = 0; i < 255; i++) {
        g_exp[i] = (uint8_t)x;
        g_log[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) {
            x ^= 0x11D;
        }
    }
    g_exp[255] = 1;
    g_log[0] = 0;
}
static uint8_t
gm(uint8_t a, uint8_t b) {
    return (a == 0 || b == 0) ? 0 : g_exp[(g_log[a] + g_log[b]) % 255];
}

/* rs_12_9_calc_syndrome + rs_12_9_check_syndrome + rs_12_9_correct_errors (:247-323); returns the result code (0 also when
 * the syndrome is zero), *found = roots the Chien search counted */
int
orc_rs_12_9(uint8_t cw[12], uint8_t syn3[3], uint8_t* found) {
    gf_init();
    uint8_t S[6] = {0};
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < 12; i++) {
            S[j] = cw[i] ^ gm(g_exp[j + 1], S[j]);
        }
    }
    memcpy(syn3, S, 3);
    *found = 0;
    if (!(S[0] | S[1] | S[2])) {
        return 0;
    }
    uint8_t loc[6] = {1, 0, 0, 0, 0, 0}, D[6] = {0, 1, 0, 0, 0, 0}, psi2[6] = {0};
    int L = 0, k = -1;
    for (int n = 0; n < 3; n++) {
        uint8_t d = 0;
        for (int i = 0; i <= L; i++) {
            d ^= gm(loc[i], S[n - i]);
        }
        if (d != 0) {
            for (int i = 0; i < 6; i++) {
                psi2[i] = loc[i] ^ gm(d, D[i]);
            }
            if (L < (n - k)) {
                const int L2 = n - k;
                k = n - L;
                for (int i = 0; i < 6; i++) {
                    D[i] = gm(loc[i], g_exp[255 - g_log[d]]);
                }
                L = L2;
            }
            memcpy(loc, psi2, 6);
        }
        for (int i = 5; i > 0; i--) {
            D[i] = D[i - 1];
        }
        D[0] = 0;
    }
    uint8_t prod[12] = {0}, ev[6] = {0};
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) {
            prod[i + j] ^= gm(S[j], loc[i]);
        }
    }
    memcpy(ev, prod, 3);
    uint8_t locs[256];
    int nr = 0;
    for (int r = 1; r < 256; r++) {
        uint8_t sum = 0;
        for (int q = 0; q < 4; q++) {
            sum ^= gm(g_exp[(q * r) % 255], loc[q]);
        }
        if (sum == 0) {
            locs[nr++] = (uint8_t)(255 - r);
        }
    }
    *found = (uint8_t)nr;
    if (nr == 0) {
        return 0;
    }
    if (nr > 3) {
        return 2;
    }
    for (int r = 0; r < nr; r++) {
        if (locs[r] >= 12) {
            return 2;
        }
    }
    for (int r = 0; r < nr; r++) {
        const int i = locs[r];
        uint8_t num = 0, den = 0;
        for (int j = 0; j < 6; j++) {
            num ^= gm(ev[j], g_exp[((255 - i) * j) % 255]);
        }
        for (int j = 1; j < 6; j += 2) {
            den ^= gm(loc[j], g_exp[((255 - i) * (j - 1)) % 255]);
        }
        cw[12 - i - 1] ^= gm(num, g_exp[255 - g_log[den]]);
    }
    return 1;
}

/* trellis_decode(), src/core/util/dsd_misc.c:24-71: greedy decode of the K = 5 rate-1/2 code, four bits of lookahead.
 * source holds 2 * result_len + 6 bits (the last output bit compares source[2p .. 2p + 7]). */
void
orc_trellis_decode(uint8_t* result, const uint8_t* source, int result_len) {
    static const int par5[32] = {0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 1, 0, 1, 0, 0, 1};
    unsigned reg = 0;
    int min_d = 9999, min_bt = 0;
    for (int p = 0; p < result_len; p++) {
        for (int c = 0; c < 16; c++) {
            unsigned r = reg;
            int sum = 0;
            for (int b = 0; b < 4; b++) {
                r = ((r << 1) | (unsigned)((c >> (3 - b)) & 1)) & 0x1Fu;
                sum += par5[r & 0x19u] ^ source[p * 2 + 2 * b];
                sum += par5[r & 0x17u] ^ source[p * 2 + 2 * b + 1];
            }
            if (c == 0 || sum < min_d) {
                min_d = sum;
                min_bt = (c >> 3) & 1;
            }
        }
        result[p] = (uint8_t)min_bt;
        reg = ((reg << 1) | (unsigned)min_bt) & 0x1Fu;
    }
}

/* ---- BPTC_128x77_Extract_Data (src/fec/bptc.c:167-258): embedded-signalling 8 x 16 matrix.  Rows 0..6 through Hamming(16,11,4);
 * a row that cannot be corrected gets the eleven bits the PREVIOUS row decoded to (the reference's line buffer is only
 * written on success, fec.c:432-446; for row 0 it is uninitialised there - zeros here, *row0_failed says so); 77 bits out
 * (2 x 11, 5 x 10, then the 5 CRC bits of rows 2..6); return = failed rows + columns whose parity over rows 0..6 differs
 * from row 7. */
uint32_t
orc_bptc_128x77(const uint8_t in128[128], uint8_t out77[77], int* row0_failed) {
    uint8_t m[8][16], tab[32], line[11] = {0};
    hamming_table(tab, 32, ddn_hamming_16_11_4_H, 16, 5);
    uint32_t bad = 0;
    if (row0_failed) {
        *row0_failed = 0;
    }
    for (int i = 0; i < 128; i++) {
        m[i / 16][i % 16] = in128[i] & 1;
    }
    for (int i = 0; i < 7; i++) {
        uint8_t rx[16];
        memcpy(rx, m[i], 16);
        const int sy = syn_bits(rx, ddn_hamming_16_11_4_H, 16, 5);
        int ok = 1;
        if (sy > 0) {
            if (tab[sy] == 0xFF) {
                ok = 0;
            } else {
                rx[tab[sy]] ^= 1;
            }
        }
        if (ok) {
            memcpy(line, rx, 11);
        } else {
            bad++;
            if (i == 0 && row0_failed) {
                *row0_failed = 1;
            }
        }
        memcpy(m[i], line, 11);
    }
    int k = 0;
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 11; j++) {
            out77[k++] = m[i][j];
        }
    }
    for (int i = 2; i < 7; i++) {
        for (int j = 0; j < 10; j++) {
            out77[k++] = m[i][j];
        }
    }
    for (int i = 2; i < 7; i++) {
        out77[k++] = m[i][10];
    }
    for (int c = 0; c < 16; c++) {
        int ones = 0;
        for (int j = 0; j < 7; j++) {
            ones += m[j][c];
        }
        if ((ones & 1) != m[7][c]) {
            bad++;
        }
    }
    return bad;
}

/* ---- BPTC_16x2_Extract_Data (src/fec/bptc.c:278-336): reverse-channel single-burst BPTC.  De-interleave (measured from the
 * compiled reference, tools/gen_tables_bptc.py), row 0 through Hamming(16,11,4), 32 bits out (11 corrected data bits, the 5
 * Hamming bits and the 16 parity-row bits as received); return = failed row + positions whose parity-row bit is equal to
 * (odd) / different from (even) the row-0 bit.  When the row cannot be corrected the reference reads its uninitialised line
 * buffer; here the received bits stay and *hamming_failed is set. */
#include "ddn_tables_bptc.h"
uint32_t
orc_bptc_16x2(const uint8_t in32[32], uint8_t out32[32], uint32_t parity_odd, int* hamming_failed) {
    static const uint8_t perm[32] = DDN_BPTC_RC_PERM_INIT;
    uint8_t tab[32], rx[16];
    hamming_table(tab, 32, ddn_hamming_16_11_4_H, 16, 5);
    for (int i = 0; i < 32; i++) {
        out32[perm[i]] = in32[i] & 1;
    }
    memcpy(rx, out32, 16);
    const int sy = syn_bits(rx, ddn_hamming_16_11_4_H, 16, 5);
    uint32_t bad = 0;
    if (sy > 0) {
        if (tab[sy] == 0xFF) {
            bad = 1;
        } else {
            rx[tab[sy]] ^= 1;
        }
    }
    if (hamming_failed) {
        *hamming_failed = (int)bad;
    }
    if (!bad) {
        memcpy(out32, rx, 11);
    }
    for (int i = 0; i < 16; i++) {
        const int same = out32[i] == out32[i + 16];
        bad += parity_odd ? (same ? 1u : 0u) : (same ? 0u : 1u);
    }
    return bad;
}
