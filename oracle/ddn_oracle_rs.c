/*
 * oracle/ddn_oracle_rs.c — CPU restatement of the P25 Phase 1 Golay(24,12,8) and Reed-Solomon GF(64) hard-decision
 * decoders (TEST INFRASTRUCTURE ONLY).
 *
 *   Golay: include/dsd-neo/fec/Golay24.hpp:19-197 (generator 0xAE3, [23,12] word = data bits 0..11 + 11 check bits
 *          12..22, overall parity in bit 23; systematic-search corrector), DSD adapters :262-389
 *          (check_and_fix_golay_24_6 / _24_12, src/protocol/p25/phase1/p25p1_check_hdu.cpp:28-37)
 *   RS:    include/dsd-neo/fec/ReedSolomon.hpp:61-105,334-582,685-772 (GF(64) from x^6+x+1, syndromes r(alpha^i)
 *          i = 1..2t, Berlekamp iteration, Chien search, error values), DSD adapters :836-866,933-963,1028-1058
 *          (parity symbols first, then data, zero padding up to 63; only the data symbols are written back)
 *
 * The restatement decodes the same codes by the textbook route instead of transliterating the search loops:
 *   - the [23,12,7] Golay code is perfect, so "the codeword the reference's search converges to" is the unique
 *     codeword within distance 3, found here from a 2048-entry syndrome -> error-pattern table.  The reference's
 *     *errs is the weight of the syndrome at the shift where its search stops: the full error weight when the errors
 *     fit inside one cyclic window of 11 consecutive positions (first pass), one less when a trial bit had to be
 *     flipped first (Golay24.hpp:128-161); with <= 3 errors those are the only two outcomes.
 *   - the RS decoders succeed exactly when a codeword of the length-63 mother code lies within t symbols of the
 *     zero-padded received word (locator degree <= t and as many roots as its degree); that codeword is unique, so
 *     Massey's form of the iteration + Chien + Forney returns the same symbols; on failure the data is untouched.
 * Both claims are what tests/test_oracle_rs.py pins bit for bit against the compiled reference (random codewords
 * with 0..t+3 symbol errors incl. errors that land in the zero padding, all 2^24 Golay syndromes/parities sampled).
 */
#include "ddn_oracle.h"

#include <string.h>

/* ---- Golay ------------------------------------------------------------------------------------------------ */
static uint32_t g_gol_tab[2048];
static int g_gol_ready;

static uint32_t
golay_syndrome11(uint32_t cw) {
    cw &= 0x7fffffu;
    for (int i = 0; i < 12; i++) {
        if (cw & 1u) {
            cw ^= 0xAE3u;
        }
        cw >>= 1;
    }
    return cw; /* 11 bits */
}

static void
golay_init(void) {
    if (g_gol_ready) {
        return;
    }
    g_gol_tab[0] = 0;
    for (int a = 0; a < 23; a++) {
        const uint32_t ea = 1u << a;
        g_gol_tab[golay_syndrome11(ea)] = ea;
        for (int b = a + 1; b < 23; b++) {
            const uint32_t eb = ea | (1u << b);
            g_gol_tab[golay_syndrome11(eb)] = eb;
            for (int c = b + 1; c < 23; c++) {
                const uint32_t ec = eb | (1u << c);
                g_gol_tab[golay_syndrome11(ec)] = ec;
            }
        }
    }
    g_gol_ready = 1;
}

static int
popc(uint32_t v) {
    int n = 0;
    while (v) {
        n += (int)(v & 1u);
        v >>= 1;
    }
    return n;
}

static int
fits_check_window(uint32_t e) {
    /* some left-rotation of the 23-bit pattern has every set bit in positions 12..22 */
    for (int i = 0; i < 23; i++) {
        if ((e & 0xFFFu) == 0) {
            return 1;
        }
        e = ((e << 1) | (e >> 22)) & 0x7fffffu;
    }
    return 0;
}

/* check_and_fix_golay_24_6 (len 6) / _24_12 (len 12): data bits corrected in place, returns 0/1, *fixed as the ref. */
int
orc_golay_24_decode(uint8_t* data, int len, const uint8_t* parity, int* fixed) {
    golay_init();
    *fixed = 0;
    for (int i = 0; i < len; i++) {
        if (data[i] > 1) {
            return 1;
        }
    }
    for (int i = 0; i < 12; i++) {
        if (parity[i] > 1) {
            return 1;
        }
    }
    uint32_t cw = 0;
    for (int k = 0; k < 12; k++) {
        cw |= (uint32_t)parity[k] << (12 + k);
    }
    for (int k = 0; k < len; k++) {
        cw |= (uint32_t)data[k] << (12 - len + k);
    }
    const uint32_t pbit = cw & 0x800000u;
    uint32_t w23 = cw & 0x7fffffu;
    const uint32_t e = g_gol_tab[golay_syndrome11(w23)];
    if (e) {
        const int wt = popc(e);
        *fixed = fits_check_window(e) ? wt : wt - 1;
        w23 ^= e;
    }
    cw = w23 | pbit;
    const int odd = popc(cw) & 1;
    if (odd && (cw & 0x3fu) != 0) {
        return 1;
    }
    for (int i = 0; i < len; i++) {
        data[i] = (uint8_t)((cw >> (12 - len + i)) & 1u);
    }
    return 0;
}

/* ---- Reed-Solomon over GF(64) ---------------------------------------------------------------------------- */
static uint8_t g_ex[128], g_lg[64];
static int g_gf_ready;

static void
gf_init(void) {
    if (g_gf_ready) {
        return;
    }
    int x = 1;
    for (int i = 0; i < 63; i++) {
        g_ex[i] = (uint8_t)x;
        g_ex[i + 63] = (uint8_t)x;
        g_lg[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x40) {
            x ^= 0x43;
        }
    }
    g_gf_ready = 1;
}

static int
gmul(int a, int b) {
    return (a && b) ? g_ex[g_lg[a] + g_lg[b]] : 0;
}
static int
gdiv(int a, int b) {
    return a ? g_ex[g_lg[a] + 63 - g_lg[b]] : 0;
}

/* word[63] symbols (coefficient of x^j at j); t <= 8.  Returns 0 and corrects in place, or 1 and leaves it alone. */
int
orc_rs63_decode(int* word, int t) {
    gf_init();
    const int n2 = 2 * t;
    int S[17] = {0};
    int any = 0;
    for (int i = 1; i <= n2; i++) {
        int s = 0;
        for (int j = 0; j < 63; j++) {
            if (word[j]) {
                s ^= g_ex[(g_lg[word[j]] + i * j) % 63];
            }
        }
        S[i] = s;
        any |= s;
    }
    if (!any) {
        return 0;
    }
    /* Massey: C(x) connection polynomial, L its LFSR length */
    int C[18] = {1}, Bp[18] = {1}, T[18];
    int L = 0, m = 1, b = 1;
    for (int n = 0; n < n2; n++) {
        int d = S[n + 1];
        for (int i = 1; i <= L; i++) {
            d ^= gmul(C[i], S[n + 1 - i]);
        }
        if (d == 0) {
            m++;
        } else {
            const int f = gdiv(d, b);
            memcpy(T, C, sizeof(T));
            for (int i = 0; i + m < 18; i++) {
                C[i + m] ^= gmul(f, Bp[i]);
            }
            if (2 * L <= n) {
                L = n + 1 - L;
                memcpy(Bp, T, sizeof(T));
                b = d;
                m = 1;
            } else {
                m++;
            }
        }
    }
    if (L > t) {
        return 1;
    }
    int deg = 0;
    for (int i = 17; i >= 0; i--) {
        if (C[i]) {
            deg = i;
            break;
        }
    }
    if (deg != L) {
        return 1;
    }
    /* Chien: error at position p <-> root alpha^(-p) */
    int pos[8], np = 0;
    for (int p = 0; p < 63; p++) {
        int v = 0;
        for (int i = 0; i <= L; i++) {
            if (C[i]) {
                v ^= g_ex[(g_lg[C[i]] + i * (63 - p)) % 63];
            }
        }
        if (v == 0) {
            if (np < 8) {
                pos[np] = p;
            }
            np++;
        }
    }
    if (np != L) {
        return 1;
    }
    /* Forney with roots alpha^1..: Omega(x) = S(x) C(x) mod x^2t, S(x) = sum S_{i+1} x^i;
     * e_p = X^0 * Omega(X^-1) / C'(X^-1), X = alpha^p (first consecutive root exponent is 1) */
    int Om[17] = {0};
    for (int i = 0; i < n2; i++) {
        int v = 0;
        for (int j = 0; j <= i && j <= L; j++) {
            v ^= gmul(C[j], S[i - j + 1]);
        }
        Om[i] = v;
    }
    for (int k = 0; k < L; k++) {
        const int p = pos[k];
        const int xi = (63 - p) % 63; /* exponent of X^-1 */
        int num = 0, den = 0;
        for (int i = 0; i < n2; i++) {
            if (Om[i]) {
                num ^= g_ex[(g_lg[Om[i]] + i * xi) % 63];
            }
        }
        for (int i = 1; i <= L; i += 2) {
            if (C[i]) {
                den ^= g_ex[(g_lg[C[i]] + (i - 1) * xi) % 63];
            }
        }
        if (den == 0) {
            return 1;
        }
        word[p] ^= gdiv(num, den);
    }
    return 0;
}

/* check_and_fix_reedsolomon_24_12_13 (n_par 12, n_data 12, t 6), _24_16_9 (8, 16, 4), redsolomon_36_20_17 (16, 20, 8):
 * data6 = n_data six-bit words, one bit per byte MSB first, corrected in place; parity6 likewise. */
int
orc_p25_rs_decode(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t) {
    int w[63] = {0};
    for (int i = 0; i < n_par; i++) {
        int v = 0;
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (parity6[6 * i + b] != 0);
        }
        w[i] = v;
    }
    for (int i = 0; i < n_data; i++) {
        int v = 0;
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (data6[6 * i + b] != 0);
        }
        w[n_par + i] = v;
    }
    const int rc = orc_rs63_decode(w, t);
    for (int i = 0; i < n_data; i++) {
        for (int b = 0; b < 6; b++) {
            data6[6 * i + b] = (uint8_t)((w[n_par + i] >> (5 - b)) & 1);
        }
    }
    return rc;
}

/* ---- soft (Chase) variants, src/protocol/p25/phase1/p25p1_soft.cpp ------------------------------------------ */
/* Reliabilities are clamped to 0..255; "least reliable k" = the first k positions in (reliability, position) order
 * (p25p1_soft.cpp:175-205: the erasure threshold only reorders below-threshold entries first, which an ascending sort
 * already does).  A candidate's penalty is the clamped reliability summed over the bits where the re-encoded
 * codeword differs from the received word; ties go to fewer differing bits; masks are tried in increasing order and
 * only strict improvements replace the incumbent.  Hard-decision override margin 8 (p25p1_soft.cpp:21,462-470). */
static int
clamp255i(int v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

static void
least_reliable(const int* reliab, int n, int k, int* out) {
    int key[32], used[32] = {0};
    for (int i = 0; i < n; i++) {
        key[i] = clamp255i(reliab[i]) * 64 + i;
    }
    for (int j = 0; j < k; j++) {
        int best = -1;
        for (int i = 0; i < n; i++) {
            if (!used[i] && (best < 0 || key[i] < key[best])) {
                best = i;
            }
        }
        used[best] = 1;
        out[j] = best;
    }
}

static void
hamming_parity(const uint8_t d[6], uint8_t p[4]) {
    p[0] = d[0] ^ d[1] ^ d[2] ^ d[5];
    p[1] = d[0] ^ d[1] ^ d[3] ^ d[5];
    p[2] = d[0] ^ d[2] ^ d[3] ^ d[4];
    p[3] = d[1] ^ d[2] ^ d[3] ^ d[4];
}

/* == hamming_10_6_3_decode on a bit-per-byte word: returns 0/1/2, corrects data bits in place on 1 */
static int
hamming_hard_bits(uint8_t b[10]) {
    int w = 0;
    for (int i = 0; i < 10; i++) {
        if (b[i] > 1) {
            return 2;
        }
        w = (w << 1) | b[i];
    }
    int fixed6 = 0;
    const int e = orc_hamming_10_6_3(w, &fixed6);
    if (e == 1) {
        for (int i = 0; i < 6; i++) {
            b[i] = (uint8_t)((fixed6 >> (5 - i)) & 1);
        }
    }
    return e;
}

int
orc_hamming_10_6_3_soft(const uint8_t* bits, const int* reliab, uint8_t* out) {
    int best_pen = 999999, best_flips = 99, found = 0, hard_valid = 0, hard_corr = 0, hard_pen = 999999;
    uint8_t best[10], hard[10];
    memset(best, 0, sizeof(best));
    memset(hard, 0, sizeof(hard));
    {
        uint8_t t[10];
        memcpy(t, bits, 10);
        const int r = hamming_hard_bits(t);
        if (r == 0 || r == 1) {
            memcpy(hard, t, 6);
            hamming_parity(hard, hard + 6);
            hard_valid = 1;
            hard_corr = (r == 1);
            int pen = 0, nd = 0;
            for (int i = 0; i < 10; i++) {
                if (bits[i] != hard[i]) {
                    pen += clamp255i(reliab[i]);
                    nd++;
                }
            }
            hard_pen = best_pen = pen;
            best_flips = nd;
            memcpy(best, hard, 10);
            found = 1;
        }
    }
    int lr[5];
    least_reliable(reliab, 10, 5, lr);
    for (int mask = 0; mask < 32; mask++) {
        if (__builtin_popcount((unsigned)mask) > 2) {
            continue;
        }
        uint8_t c[10];
        memcpy(c, bits, 10);
        for (int b = 0; b < 5; b++) {
            if (mask & (1 << b)) {
                c[lr[b]] ^= 1;
            }
        }
        if (hamming_hard_bits(c) != 0) {
            continue;
        }
        int pen = 0;
        for (int i = 0; i < 10; i++) {
            if (bits[i] != c[i]) {
                pen += clamp255i(reliab[i]);
            }
        }
        const int nf = __builtin_popcount((unsigned)mask);
        if (pen < best_pen || (pen == best_pen && nf < best_flips)) {
            best_pen = pen;
            best_flips = nf;
            memcpy(best, c, 10);
            found = 1;
        }
    }
    if (found) {
        if (hard_valid && hard_corr && memcmp(best, hard, 10) != 0 && best_pen + 8 >= hard_pen) {
            memcpy(out, hard, 10);
            return 1;
        }
        memcpy(out, best, 10);
        return memcmp(bits, best, 10) == 0 ? 0 : 1;
    }
    memcpy(out, bits, 10);
    return 2;
}

static void
golay_reencode(const uint8_t* data, int len, uint8_t* word /* len + 12 */) {
    uint8_t d12[12] = {0};
    memcpy(d12 + 12 - len, data, (size_t)len);
    uint32_t d = 0;
    for (int k = 0; k < 12; k++) {
        d |= (uint32_t)d12[k] << k;
    }
    uint32_t cw = d;
    for (int i = 0; i < 12; i++) {
        if (cw & 1u) {
            cw ^= 0xAE3u;
        }
        cw >>= 1;
    }
    uint32_t w = (cw << 12) | d;
    if (popc(w) & 1) {
        w ^= 0x800000u;
    }
    memcpy(word, data, (size_t)len);
    for (int k = 0; k < 12; k++) {
        word[len + k] = (uint8_t)((w >> (12 + k)) & 1u);
    }
}

/* == check_and_fix_golay_24_6_soft (len 6) / _24_12_soft (len 12) */
int
orc_golay_24_soft(uint8_t* data, int len, const uint8_t* parity, const int* reliab, int* fixed) {
    const int n = len + 12;
    uint8_t orig[24], dec[24], best_data[12], hard_data[12];
    *fixed = 0;
    memcpy(orig, data, (size_t)len);
    memcpy(orig + len, parity, 12);
    int best_pen = 999999, best_fixed = 0, found = 0, hard_valid = 0, hard_corr = 0, hard_pen = 999999, hard_fixed = 0;
    {
        uint8_t t[12];
        memcpy(t, data, (size_t)len);
        if (orc_golay_24_decode(t, len, parity, &hard_fixed) == 0) {
            golay_reencode(t, len, dec);
            int pen = 0, nd = 0;
            for (int i = 0; i < n; i++) {
                if (orig[i] != dec[i]) {
                    pen += clamp255i(reliab[i]);
                    nd++;
                }
            }
            hard_valid = 1;
            hard_corr = hard_fixed > 0;
            hard_pen = best_pen = pen;
            best_fixed = nd;
            memcpy(hard_data, t, (size_t)len);
            memcpy(best_data, t, (size_t)len);
            found = 1;
        }
    }
    int lr[8];
    least_reliable(reliab, n, 8, lr);
    for (int mask = 0; mask < 256; mask++) {
        if (__builtin_popcount((unsigned)mask) > 4) {
            continue;
        }
        uint8_t c[24];
        memcpy(c, orig, (size_t)n);
        for (int b = 0; b < 8; b++) {
            if (mask & (1 << b)) {
                c[lr[b]] ^= 1;
            }
        }
        int cf = 0;
        if (orc_golay_24_decode(c, len, c + len, &cf) != 0) {
            continue;
        }
        golay_reencode(c, len, dec);
        int pen = 0, nd = 0;
        for (int i = 0; i < n; i++) {
            if (orig[i] != dec[i]) {
                pen += clamp255i(reliab[i]);
                nd++;
            }
        }
        if (pen < best_pen || (pen == best_pen && nd < best_fixed)) {
            best_pen = pen;
            best_fixed = nd;
            memcpy(best_data, c, (size_t)len);
            found = 1;
        }
    }
    if (!found) {
        return 1;
    }
    if (hard_valid && hard_corr && memcmp(best_data, hard_data, (size_t)len) != 0 && best_pen + 8 >= hard_pen) {
        memcpy(data, hard_data, (size_t)len);
        *fixed = hard_fixed;
        return 0;
    }
    memcpy(data, best_data, (size_t)len);
    *fixed = best_fixed;
    return 0;
}

/* ---- RS errors-and-erasures (ReedSolomon.hpp:175-262,585-690,774-802) and the ranked-erasure retry loop -------- */
/* word[63] in place; erasures = n distinct positions 0..62.  Same stages as the reference: erasure locator, modified
 * syndromes, Massey on the last 2t - n of them, degree bound 2v + n <= 2t, combined locator, Chien (root count must equal
 * the degree and cover every erasure), errata values, and a final all-syndromes-zero check.  The reference gets the
 * values from a Vandermonde solve over the first n_loc syndromes; with distinct locations that system has exactly one
 * solution, which Forney's formula on the combined locator gives as well.  Returns 0 (word corrected) or 1 (untouched). */
static int
rs63_syndromes(const int* word, int n2, int* S /* 1..n2 */) {
    int any = 0;
    for (int i = 1; i <= n2; i++) {
        int s = 0;
        for (int j = 0; j < 63; j++) {
            if (word[j]) {
                s ^= g_ex[(g_lg[word[j]] + i * j) % 63];
            }
        }
        S[i] = s;
        any |= s;
    }
    return any;
}

int
orc_rs63_decode_erasures(int* word, int t, const int* erasures, int n_er) {
    gf_init();
    const int n2 = 2 * t;
    if (n_er < 0 || n_er > n2) {
        return 1;
    }
    int S[17] = {0};
    if (n_er == 0) {
        return rs63_syndromes(word, n2, S) ? 1 : 0;
    }
    uint8_t seen[63] = {0};
    for (int i = 0; i < n_er; i++) {
        if (erasures[i] < 0 || erasures[i] >= 63 || seen[erasures[i]]) {
            return 1;
        }
        seen[erasures[i]] = 1;
    }
    if (!rs63_syndromes(word, n2, S)) {
        return 0;
    }
    int G[18] = {1}; /* erasure locator prod (1 + alpha^pos x) */
    for (int e = 0, deg = 0; e < n_er; e++, deg++) {
        const int f = g_ex[erasures[e] % 63];
        for (int i = deg; i >= 0; i--) {
            G[i + 1] ^= gmul(G[i], f);
        }
    }
    int M[17] = {0}; /* modified syndromes, index 0..n2-1 */
    for (int i = 0; i < n2; i++) {
        int v = 0;
        for (int j = 0; j <= n_er && j <= i; j++) {
            v ^= gmul(G[j], S[i - j + 1]);
        }
        M[i] = v;
    }
    /* Massey on M[n_er .. n2-1] */
    const int ns = n2 - n_er;
    const int* sy = M + n_er;
    int C[18] = {1}, B[18] = {1}, T[18];
    int L = 0, m = 1, bb = 1;
    for (int n = 0; n < ns; n++) {
        int d = sy[n];
        for (int i = 1; i <= L; i++) {
            d ^= gmul(C[i], sy[n - i]);
        }
        if (d == 0) {
            m++;
            continue;
        }
        memcpy(T, C, sizeof(T));
        const int co = gdiv(d, bb);
        for (int i = 0; i + m <= n2; i++) {
            if (B[i]) {
                C[i + m] ^= gmul(co, B[i]);
            }
        }
        if (2 * L <= n) {
            L = n + 1 - L;
            memcpy(B, T, sizeof(T));
            bb = d;
            m = 1;
        } else {
            m++;
        }
    }
    int udeg = 0;
    for (int i = n2; i >= 0; i--) {
        if (C[i]) {
            udeg = i;
            break;
        }
    }
    if (2 * udeg + n_er > n2) {
        return 1;
    }
    int gdeg = 0;
    for (int i = n2; i >= 0; i--) {
        if (G[i]) {
            gdeg = i;
            break;
        }
    }
    if (gdeg + udeg > n2) {
        return 1;
    }
    int Lam[18] = {0};
    for (int i = 0; i <= gdeg; i++) {
        for (int j = 0; j <= udeg; j++) {
            Lam[i + j] ^= gmul(G[i], C[j]);
        }
    }
    int ldeg = 0;
    for (int i = n2; i >= 0; i--) {
        if (Lam[i]) {
            ldeg = i;
            break;
        }
    }
    int loc[17], nl = 0;
    if (ldeg > 0) {
        for (int p = 0; p < 63; p++) {
            int v = 0;
            for (int i = 0; i <= ldeg; i++) {
                if (Lam[i]) {
                    v ^= g_ex[(g_lg[Lam[i]] + i * ((63 - p) % 63)) % 63];
                }
            }
            if (v == 0) {
                if (nl >= n2) {
                    nl++;
                    break;
                }
                loc[nl++] = p;
            }
        }
    }
    if (nl != ldeg || nl > n2) {
        return 1;
    }
    for (int i = 0; i < n_er; i++) {
        int ok = 0;
        for (int k = 0; k < nl; k++) {
            ok |= (loc[k] == erasures[i]);
        }
        if (!ok) {
            return 1;
        }
    }
    /* errata values: Omega = S(x) Lam(x) mod x^2t, e = Omega(X^-1) / Lam'(X^-1) */
    int Om[17] = {0};
    for (int i = 0; i < n2; i++) {
        int v = 0;
        for (int j = 0; j <= i && j <= ldeg; j++) {
            v ^= gmul(Lam[j], S[i - j + 1]);
        }
        Om[i] = v;
    }
    int out[63];
    memcpy(out, word, sizeof(out));
    for (int k = 0; k < nl; k++) {
        const int xi = (63 - loc[k]) % 63;
        int num = 0, den = 0;
        for (int i = 0; i < n2; i++) {
            if (Om[i]) {
                num ^= g_ex[(g_lg[Om[i]] + i * xi) % 63];
            }
        }
        for (int i = 1; i <= ldeg; i += 2) {
            if (Lam[i]) {
                den ^= g_ex[(g_lg[Lam[i]] + (i - 1) * xi) % 63];
            }
        }
        if (den == 0) {
            return 1;
        }
        out[loc[k]] ^= gdiv(num, den);
    }
    int S2[17];
    if (rs63_syndromes(out, n2, S2)) {
        return 1;
    }
    memcpy(word, out, sizeof(out));
    return 0;
}

static void
bits6_to_word(const uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int* w) {
    memset(w, 0, 63 * sizeof(int));
    for (int i = 0; i < n_par; i++) {
        int v = 0;
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (parity6[6 * i + b] != 0);
        }
        w[i] = v;
    }
    for (int i = 0; i < n_data; i++) {
        int v = 0;
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (data6[6 * i + b] != 0);
        }
        w[n_par + i] = v;
    }
}

/* == check_and_fix_*_soft(data, parity, erasures, n) */
int
orc_p25_rs_decode_soft(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t, const int* erasures,
                       int n_er) {
    int w[63];
    bits6_to_word(data6, parity6, n_par, n_data, w);
    /* the reference's hard attempt rewrites the data bits from the symbols whatever the outcome */
    int h[63];
    memcpy(h, w, sizeof(h));
    const int hard = orc_rs63_decode(h, t);
    for (int i = 0; i < n_data; i++) {
        for (int b = 0; b < 6; b++) {
            data6[6 * i + b] = (uint8_t)((h[n_par + i] >> (5 - b)) & 1);
        }
    }
    if (hard == 0) {
        return 0;
    }
    if (n_er <= 0 || n_er > 2 * t || erasures == NULL) {
        return 1;
    }
    const int rc = orc_rs63_decode_erasures(w, t, erasures, n_er);
    if (rc == 0) {
        for (int i = 0; i < n_data; i++) {
            for (int b = 0; b < 6; b++) {
                data6[6 * i + b] = (uint8_t)((w[n_par + i] >> (5 - b)) & 1);
            }
        }
    }
    return rc;
}

/* == p25p1_build_rs_ranked_erasures(data_reliab, n_data, parity_reliab, n_par, min_erasures, out, max) with the default
 * erasure threshold 64 (src/protocol/p25/phase1/p25p1_soft.cpp:20,141-168) */
int
orc_p25_rs_ranked_erasures(const uint8_t* data_rel, int n_data, const uint8_t* parity_rel, int n_par, int min_er,
                           int* out, int max_er) {
    int key[64], n = 0, hits = 0;
    for (int i = 0; i < n_par && n < 64; i++) {
        hits += parity_rel[i] < 64;
        key[n++] = parity_rel[i] * 64 + i;
    }
    for (int i = 0; i < n_data && n < 64; i++) {
        hits += data_rel[i] < 64;
        key[n++] = data_rel[i] * 64 + (n_par + i);
    }
    for (int i = 1; i < n; i++) { /* (reliability, position) ascending */
        const int k = key[i];
        int j = i - 1;
        while (j >= 0 && key[j] > k) {
            key[j + 1] = key[j];
            j--;
        }
        key[j + 1] = k;
    }
    int cnt = hits > min_er ? hits : min_er;
    if (cnt > n) {
        cnt = n;
    }
    if (cnt > max_er) {
        cnt = max_er;
    }
    for (int i = 0; i < cnt; i++) {
        out[i] = key[i] & 63;
    }
    return cnt;
}

/* == p25p1_rs_{24_12_13,24_16_9,36_20_17}_soft_reliability */
int
orc_p25_rs_soft_reliability(uint8_t* data6, const uint8_t* parity6, const uint8_t* data_rel, const uint8_t* parity_rel,
                            int n_par, int n_data, int t) {
    int er[16];
    const int nr = orc_p25_rs_ranked_erasures(data_rel, n_data, parity_rel, n_par, t, er, 2 * t);
    uint8_t orig[20 * 6], cand[20 * 6];
    memcpy(orig, data6, (size_t)n_data * 6);
    for (int n = 1; n <= nr; n++) {
        memcpy(cand, orig, (size_t)n_data * 6);
        if (orc_p25_rs_decode_soft(cand, parity6, n_par, n_data, t, er, n) == 0) {
            memcpy(data6, cand, (size_t)n_data * 6);
            return 0;
        }
    }
    return 1;
}

/* ---- P25 Phase 2 RS(63,35) over GF(64) with caller-given erasures (ESS / FACCH / SACCH sections) -------------------------
 * == ez_rs28_ess / _facch / _sacch (src/fec/ez.cpp:104-281) over the vendored ezpwd RS<63,35> (x^6+x+1, first consecutive
 * root alpha^1, 28 roots; src/third_party/ezpwd/rs:79, decoder rs_base:1380-1720).  That decoder is the classic
 * errors-and-erasures procedure, restated here on coefficient values:
 *   - block position p = 0..62 carries the coefficient of x^(62-p); a section's symbols sit at fixed block positions, the
 *     rest of the block is zero ("pad" in front of the data, punctured parity behind it);
 *   - syndromes S_i = c(alpha^(i+1)), i = 0..27; all zero -> 0 corrections;
 *   - the locator starts as the erasure locator prod (1 + alpha^(62-p) x) and is grown by Berlekamp's iteration from step
 *     n_erasures + 1 to 28 with the length rule 2 L <= r + n_erasures - 1;
 *   - Chien search over i = 1..63 (root alpha^i <-> block position i - 1), stopped when deg(lambda) roots are found; a root
 *     count short of the degree, or degree 0 with a non-zero syndrome, is a failure (-1), nothing touched;
 *   - errata values by Forney from omega = S lambda mod x^deg; a zero derivative, or a non-zero value inside the pad,
 *     fails the decode.  The reference decodes a masked copy of the caller's symbols (6-bit symbols in 8-bit storage,
 *     rs_base:1180-1235) and copies it back only when the count is positive, so a failed decode leaves the section as it
 *     was received;
 *   - the return value is the number of roots (errors + erasures, whether or not an erased symbol was actually wrong).
 * Pinned against the compiled reference by tests/test_oracle_rs28.py. */
#define RS28_ROOTS 28

static int
rs28_block_decode(uint8_t c[63], int pad, const int* eras_blockpos, int n_er) {
    gf_init();
    int S[RS28_ROOTS];
    int any = 0;
    for (int i = 0; i < RS28_ROOTS; i++) {
        int v = 0;
        for (int p = 0; p < 63; p++) { /* Horner from the highest coefficient */
            v = gmul(v, g_ex[i + 1]) ^ c[p];
        }
        S[i] = v;
        any |= v;
    }
    if (!any) {
        return 0;
    }
    int lam[RS28_ROOTS + 1] = {1}, b[RS28_ROOTS + 1], t[RS28_ROOTS + 1];
    for (int e = 0; e < n_er; e++) { /* times (1 + X x), X = alpha^(62 - p) */
        const int X = g_ex[(62 - eras_blockpos[e]) % 63];
        for (int j = e + 1; j > 0; j--) {
            lam[j] ^= gmul(lam[j - 1], X);
        }
    }
    memcpy(b, lam, sizeof(b));
    int el = n_er;
    for (int r = n_er + 1; r <= RS28_ROOTS; r++) {
        int d = 0;
        for (int i = 0; i < r; i++) {
            d ^= gmul(lam[i], S[r - i - 1]);
        }
        if (d == 0) {
            memmove(b + 1, b, sizeof(int) * RS28_ROOTS); /* b <- x b */
            b[0] = 0;
            continue;
        }
        t[0] = lam[0];
        for (int i = 0; i < RS28_ROOTS; i++) {
            t[i + 1] = lam[i + 1] ^ gmul(d, b[i]);
        }
        if (2 * el <= r + n_er - 1) {
            el = r + n_er - el;
            for (int i = 0; i <= RS28_ROOTS; i++) {
                b[i] = gdiv(lam[i], d);
            }
        } else {
            memmove(b + 1, b, sizeof(int) * RS28_ROOTS);
            b[0] = 0;
        }
        memcpy(lam, t, sizeof(lam));
    }
    int deg = 0;
    for (int i = 0; i <= RS28_ROOTS; i++) {
        if (lam[i]) {
            deg = i;
        }
    }
    int root[RS28_ROOTS], loc[RS28_ROOTS], count = 0;
    for (int i = 1; i <= 63 && count < deg; i++) {
        int q = 1;
        for (int j = 1; j <= deg; j++) {
            if (lam[j]) {
                q ^= g_ex[(g_lg[lam[j]] + i * j) % 63];
            }
        }
        if (q == 0) {
            root[count] = i;
            loc[count] = i - 1;
            count++;
        }
    }
    if (count != deg || deg == 0) {
        return -1;
    }
    int om[RS28_ROOTS];
    for (int i = 0; i < deg; i++) {
        int v = 0;
        for (int j = 0; j <= i; j++) {
            v ^= gmul(S[i - j], lam[j]);
        }
        om[i] = v;
    }
    const int top = (deg < RS28_ROOTS - 1 ? deg : RS28_ROOTS - 1) & ~1;
    for (int j = count - 1; j >= 0; j--) {
        int num = 0, den = 0;
        for (int i = 0; i < deg; i++) {
            if (om[i]) {
                num ^= g_ex[(g_lg[om[i]] + i * root[j]) % 63];
            }
        }
        for (int i = top; i >= 0; i -= 2) {
            if (lam[i + 1]) {
                den ^= g_ex[(g_lg[lam[i + 1]] + i * root[j]) % 63];
            }
        }
        if (den == 0) {
            return -1;
        }
        if (num != 0) {
            if (loc[j] < pad) {
                return -1;
            }
            c[loc[j]] ^= (uint8_t)gdiv(num, den); /* first consecutive root 1: no extra factor */
        }
    }
    return count;
}

/* kind 0 ESS (16 payload + 28 parity symbols, erasure positions 0..43 count from the first payload symbol), 1 FACCH (26 +
 * 19 at block positions 9.. / 35.., erasure positions are block positions), 2 SACCH (30 + 22 at 5.. / 35..).  Bit arrays
 * hold one bit per int, most significant bit of a symbol first.  At most 28 erasures are consumed.  Returns what the
 * reference returns; the payload changes only when that is positive. */
int
orc_ez_rs28(int kind, int* payload, const int* parity, const int* erasures, int n_erasures) {
    static const int n_data[3] = {16, 26, 30}, n_par[3] = {28, 19, 22}, first[3] = {19, 9, 5};
    if (kind < 0 || kind > 2) {
        return -2;
    }
    uint8_t c[63] = {0};
    for (int i = 0; i < n_data[kind]; i++) {
        int v = 0;
        for (int j = 0; j < 6; j++) {
            v = (v << 1) + payload[6 * i + j];
        }
        c[first[kind] + i] = (uint8_t)(v & 63);
    }
    for (int i = 0; i < n_par[kind]; i++) {
        int v = 0;
        for (int j = 0; j < 6; j++) {
            v = (v << 1) + parity[6 * i + j];
        }
        c[35 + i] = (uint8_t)(v & 63);
    }
    int ep[RS28_ROOTS], n = 0;
    for (int i = 0; erasures && i < n_erasures && i < RS28_ROOTS; i++) {
        ep[n++] = erasures[i] + (kind == 0 ? 19 : 0);
    }
    uint8_t w[63];
    memcpy(w, c, sizeof(w));
    const int ec = rs28_block_decode(w, kind == 0 ? 19 : 0, ep, n);
    if (ec > 0) {
        memcpy(c, w, sizeof(c));
    }
    for (int i = 0; i < n_data[kind]; i++) {
        for (int j = 0; j < 6; j++) {
            payload[6 * i + j] = (c[first[kind] + i] >> (5 - j)) & 1;
        }
    }
    return ec;
}

/* ---- P25 Phase 2 I-ISCH lookup ((40,9,16) code, 128 codewords + the S-ISCH word) -------------------------------------------
 * == isch_lookup / isch_lookup_soft (src/fec/ez.cpp:325-384).  Table measured from the compiled reference
 * (tools/gen_tables_isch.py -> ddn_tables_isch.h).  Hard: exact match, else the nearest entry within 7 bits; the code's
 * minimum distance is 16, so the only possible tie is between a codeword and the S-ISCH word 14 bits from it, which the
 * reference resolves by the order its unordered_map happens to be walked in - measured per codeword as well.  Soft: exact
 * matches stay authoritative, otherwise the entry within 7 bits with the least (sum of the reliabilities of the differing
 * bits, number of differing bits, answer value) - a total order, independent of the walk.  -2 = S-ISCH or nothing found. */
#include "ddn_tables_isch.h"
static const uint64_t g_isch[128] = DDN_ISCH_TABLE_INIT;
static const uint8_t g_isch_s_first[128] = DDN_ISCH_S_FIRST_INIT;

static int
popc64(uint64_t v) {
    int n = 0;
    while (v) {
        v &= v - 1;
        n++;
    }
    return n;
}

int
orc_isch_lookup(uint64_t isch) {
    int best = -2, bd = 40;
    for (int i = 0; i < 128; i++) {
        if (g_isch[i] == isch) {
            return i;
        }
        const int d = popc64(isch ^ g_isch[i]);
        if (d <= 7 && d < bd) {
            best = i;
            bd = d;
        }
    }
    if (isch == DDN_ISCH_S_WORD) {
        return -2;
    }
    const int ds = popc64(isch ^ DDN_ISCH_S_WORD);
    if (ds <= 7 && (ds < bd || (ds == bd && best >= 0 && g_isch_s_first[best]))) {
        return -2;
    }
    return best;
}

int
orc_isch_lookup_soft(uint64_t isch, const uint8_t* reliab40) {
    for (int i = 0; i < 128; i++) {
        if (g_isch[i] == isch) {
            return i;
        }
    }
    if (isch == DDN_ISCH_S_WORD) {
        return -2;
    }
    if (!reliab40) {
        return orc_isch_lookup(isch);
    }
    int best = -2, bc = 0x7fffffff, bp = 40;
    for (int i = -1; i < 128; i++) { /* -1: the S-ISCH word, answer -2 */
        const uint64_t w = i < 0 ? DDN_ISCH_S_WORD : g_isch[i];
        const int val = i < 0 ? -2 : i;
        const uint64_t diff = isch ^ w;
        const int p = popc64(diff);
        if (p > 7) {
            continue;
        }
        int cost = 0;
        for (int b = 0; b < 40; b++) {
            if (diff & (1ULL << (39 - b))) {
                cost += reliab40[b];
            }
        }
        if (cost < bc || (cost == bc && p < bp) || (cost == bc && p == bp && val < best)) {
            best = val;
            bc = cost;
            bp = p;
        }
    }
    return best;
}

/* ---- P25 Phase 2 FACCH / SACCH burst decode (test infrastructure, like everything in oracle/) --------------------------------
 * p25p2_process_facchc() / process_SACCHs() (src/protocol/p25/phase2/p25p2_frame.c:473-495,652-671): the burst's 360 bits (one
 * timeslot of p2bit / p2xbit, offsets relative to its start) -> payload + parity bits of the RS(63,35) section;
 * p25p2_decode_facch_ranked() / _sacch_ranked() (:408-470): the section with its fixed erasures (the punctured and the unsent
 * symbols), and when that fails the retries with 1, 2, ... more erasures from the ranked list of
 * p25p2_facch_soft_erasures() / p25p2_sacch_soft_erasures() (src/protocol/p25/phase2/p25p2_soft.c:40-108,255-329): a hexbit's
 * reliability = the least min(|LLR|, 255) of its six bits, candidates = the symbols not erased yet, ordered by (reliability,
 * position); as many as fall below the threshold, at least 5 / 8, at most 10 / 16.  Returns the reference's ec (>= 0 symbols
 * located, -1 beyond the code: payload as received); *used_dynamic as the reference's flag.  kind 0 = FACCH, 1 = SACCH. */
int
orc_p25p2_xcch(int kind, const uint8_t* bits360, const int16_t* llr360, int threshold, uint8_t* payload_out, int* used_dynamic) {
    static const int facch_fixed[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 54, 55, 56, 57, 58, 59, 60, 61, 62};
    static const int sacch_fixed[11] = {0, 1, 2, 3, 4, 57, 58, 59, 60, 61, 62};
    int pos_pl[180], pos_pa[132]; /* burst offset of every payload / parity bit */
    int n_pl = 0, n_pa = 0;
    if (kind == 0) {
        for (int i = 0; i < 72; i++) pos_pl[n_pl++] = i + 2;
        for (int i = 0; i < 62; i++) pos_pl[n_pl++] = i + 76;
        for (int i = 0; i < 22; i++) pos_pl[n_pl++] = i + 180;
        for (int i = 0; i < 42; i++) pos_pa[n_pa++] = i + 202;
        for (int i = 0; i < 72; i++) pos_pa[n_pa++] = i + 246;
    } else if (kind == 1) {
        for (int i = 0; i < 72; i++) pos_pl[n_pl++] = i + 2;
        for (int i = 0; i < 108; i++) pos_pl[n_pl++] = i + 76;
        for (int i = 0; i < 60; i++) pos_pa[n_pa++] = i + 184;
        for (int i = 0; i < 72; i++) pos_pa[n_pa++] = i + 246;
    } else {
        return -2;
    }
    const int rk = kind == 0 ? 1 : 2; /* orc_ez_rs28's section kind */
    const int n_fixed = kind == 0 ? 18 : 11, max_add = kind == 0 ? 10 : 16, min_add = kind == 0 ? 5 : 8;
    const int first = kind == 0 ? 9 : 5;
    const int* fixed = kind == 0 ? facch_fixed : sacch_fixed;
    int payload[180], parity[132];
    for (int i = 0; i < n_pl; i++) payload[i] = bits360[pos_pl[i]] & 1;
    for (int i = 0; i < n_pa; i++) parity[i] = bits360[pos_pa[i]] & 1;
    *used_dynamic = 0;
    int ec = orc_ez_rs28(rk, payload, parity, fixed, n_fixed);
    if (ec < 0) {
        /* ranked candidates */
        int c_rel[52], c_pos[52], nc = 0;
        for (int part = 0; part < 2; part++) {
            const int nh = (part == 0 ? n_pl : n_pa) / 6;
            const int* pos = part == 0 ? pos_pl : pos_pa;
            for (int hb = 0; hb < nh; hb++) {
                int r = 255;
                for (int b = 0; b < 6; b++) {
                    int v = llr360[pos[6 * hb + b]];
                    v = v < 0 ? -v : v;
                    v = v > 255 ? 255 : v;
                    r = v < r ? v : r;
                }
                const int rs_pos = (part == 0 ? first : 35) + hb;
                int is_fixed = 0;
                for (int k = 0; k < n_fixed; k++) is_fixed |= fixed[k] == rs_pos;
                if (!is_fixed) {
                    c_rel[nc] = r;
                    c_pos[nc] = rs_pos;
                    nc++;
                }
            }
        }
        for (int i = 0; i < nc; i++) { /* sort_candidates(): selection by (reliability, position) */
            for (int j = i + 1; j < nc; j++) {
                if (c_rel[j] < c_rel[i] || (c_rel[j] == c_rel[i] && c_pos[j] < c_pos[i])) {
                    int t = c_rel[i]; c_rel[i] = c_rel[j]; c_rel[j] = t;
                    t = c_pos[i]; c_pos[i] = c_pos[j]; c_pos[j] = t;
                }
            }
        }
        int add = 0;
        for (int i = 0; i < nc; i++) add += c_rel[i] < threshold;
        add = add < min_add ? min_add : add;
        add = add > max_add ? max_add : add;
        add = add > nc ? nc : add;
        int erasures[28];
        for (int k = 0; k < n_fixed; k++) erasures[k] = fixed[k];
        for (int k = 0; k < add; k++) erasures[n_fixed + k] = c_pos[k];
        const int n_er = n_fixed + add;
        for (int n = n_fixed + 1; n <= n_er; n++) {
            for (int i = 0; i < n_pl; i++) payload[i] = bits360[pos_pl[i]] & 1;
            ec = orc_ez_rs28(rk, payload, parity, erasures, n);
            if (ec >= 0) {
                *used_dynamic = 1;
                break;
            }
        }
        if (ec < 0) {
            for (int i = 0; i < n_pl; i++) payload[i] = bits360[pos_pl[i]] & 1;
        }
    }
    for (int i = 0; i < n_pl; i++) payload_out[i] = (uint8_t)payload[i];
    return ec;
}

/* p25p2_duid_lookup_soft() (src/protocol/p25/phase2/p25p2_frame.c:127-248; the file does not compile here - it needs the whole
 * protocol library - so this restatement is anchored on the reference's own answers for it, tests/protocol/p25/
 * test_p25_p2_reliability.c:970-974,1372-1440, and its hard table is checked entry by entry against the source's where the tree is
 * present: tests/test_oracle_p25p2_xcch.py).  The table: a canonical word of the (8,4) code or a word one bit from exactly one of
 * them, 0x80 withheld. */
static const uint8_t k_duid_canonical[16] = {0x00, 0x17, 0x2E, 0x39, 0x4B, 0x5C, 0x65, 0x72, 0x8D, 0x9A, 0xA3, 0xB4, 0xC6, 0xD1, 0xE8, 0xFF};
int
orc_p25p2_duid_hard(int r) {
    if ((r & 0xFF) == 0x80) {
        return -1;
    }
    int found = -1, n = 0;
    for (int d = 0; d < 16; d++) {
        if (__builtin_popcount((unsigned)((r ^ k_duid_canonical[d]) & 0xFF)) <= 1) {
            found = d;
            n++;
        }
    }
    return n == 1 ? found : -1;
}
int
orc_p25p2_duid_lookup_soft(int received, const uint8_t* reliab8, int threshold) {
    received &= 0xFF;
    const int hard = orc_p25p2_duid_hard(received);
    if (!reliab8 || (hard >= 0 && received == k_duid_canonical[hard])) {
        return hard;
    }
    if (received == 0x80) {
        int ok = reliab8[0] < threshold;
        for (int i = 1; i < 8; i++) {
            ok = ok && reliab8[i] >= threshold;
        }
        if (!ok) {
            return hard;
        }
    }
    int best = hard, best_cost = 999999, tied = 0;
    for (int d = 0; d < 16; d++) {
        const int diff = (received ^ k_duid_canonical[d]) & 0xFF, dist = __builtin_popcount((unsigned)diff);
        if (dist < 1 || dist > 2 || (received == 0x80 && d != 0)) {
            continue;
        }
        int cost = 0;
        for (int i = 0; i < 8 && cost < 999999; i++) {
            if ((diff >> (7 - i)) & 1) {
                cost = reliab8[i] >= threshold ? 999999 : cost + reliab8[i];
            }
        }
        if (cost >= 999999) {
            continue;
        }
        if (cost < best_cost) {
            best_cost = cost;
            best = d;
            tied = 0;
        } else if (cost == best_cost && d != best) {
            tied = 1;
        }
    }
    return tied ? hard : best;
}

/* P25 Phase 2 ESS (RS(44,16) over the four ESS-B fragments + ESS-A): p25p2_ess_decode_with_soft_erasures() (src/protocol/p25/phase2/
 * p25p2_frame.c:1061-1091) - the plain decode is taken when it located fewer than 15 symbols; otherwise the retries with the first
 * 1, 2, ... of p25p2_ess_soft_erasures_ranked()'s list (p25p2_soft.c:331-383: the 44 symbols by (least min(|LLR|, 255) of their six
 * bits, position), as many as fall below the threshold, at least 14, at most 28), each from the bits as received; the first that
 * decodes is taken.  Returns accepted (payload corrected) or 0 (payload as received); *ec = the last decode's return value. */
int
orc_p25p2_ess(const uint8_t* payload_bits96, const int16_t* payload_llr96, const uint8_t* parity_bits168, const int16_t* parity_llr168,
              int threshold, uint8_t* payload_out96, int* ec) {
    int payload[96], parity[168];
    for (int i = 0; i < 96; i++) payload[i] = payload_bits96[i] & 1;
    for (int i = 0; i < 168; i++) parity[i] = parity_bits168[i] & 1;
    int accepted = 0;
    *ec = orc_ez_rs28(0, payload, parity, 0, 0);
    if (*ec >= 0 && *ec < 15) {
        accepted = 1;
    } else {
        int rel[44], pos[44];
        for (int hb = 0; hb < 44; hb++) {
            const int16_t* l = hb < 16 ? payload_llr96 + 6 * hb : parity_llr168 + 6 * (hb - 16);
            int r = 255;
            for (int b = 0; b < 6; b++) {
                int v = l[b] < 0 ? -(int)l[b] : (int)l[b];
                v = v > 255 ? 255 : v;
                r = v < r ? v : r;
            }
            rel[hb] = r;
            pos[hb] = hb;
        }
        for (int i = 0; i < 44; i++) {
            for (int j = i + 1; j < 44; j++) {
                if (rel[j] < rel[i] || (rel[j] == rel[i] && pos[j] < pos[i])) {
                    int t = rel[i]; rel[i] = rel[j]; rel[j] = t;
                    t = pos[i]; pos[i] = pos[j]; pos[j] = t;
                }
            }
        }
        int n_er = 0;
        for (int i = 0; i < 44; i++) n_er += rel[i] < threshold;
        n_er = n_er < 14 ? 14 : n_er;
        n_er = n_er > 28 ? 28 : n_er;
        for (int n = 1; n <= n_er && !accepted; n++) {
            for (int i = 0; i < 96; i++) payload[i] = payload_bits96[i] & 1;
            *ec = orc_ez_rs28(0, payload, parity, pos, n);
            if (*ec >= 0) {
                accepted = 1;
            }
        }
        if (!accepted) {
            for (int i = 0; i < 96; i++) payload[i] = payload_bits96[i] & 1;
        }
    }
    for (int i = 0; i < 96; i++) payload_out96[i] = (uint8_t)payload[i];
    return accepted;
}
