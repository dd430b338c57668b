// ddn_rs.hip — batched P25 Phase 1 Golay(24,12,8)/(18,6,8) and Reed-Solomon GF(64) hard-decision decoders.
//
//   k_golay24   == check_and_fix_golay_24_6 / check_and_fix_golay_24_12
//                  (src/protocol/p25/phase1/p25p1_check_hdu.cpp:28-37; include/dsd-neo/fec/Golay24.hpp:19-197,262-389)
//   k_rs63      == check_and_fix_reedsolomon_24_12_13 / _24_16_9 / check_and_fix_redsolomon_36_20_17
//                  (p25p1_check_ldu.cpp, p25p1_check_hdu.cpp:39-46; include/dsd-neo/fec/ReedSolomon.hpp:334-582,738-772,
//                  adapters :836-866,933-963,1028-1058)
//
// One codeword per lane.  Golay: the [23,12,7] code is perfect, so the reference's systematic search always lands on the
// unique codeword within distance 3; here that is one lookup in a 2048-entry syndrome -> error-pattern table (LDS), and
// the reference's reported correction count is the pattern weight when the errors fit one cyclic window of 11 positions
// (its first pass) and one less otherwise (it had to flip a trial bit first).  RS: syndromes r(alpha^i), i = 1..2t, of
// the zero-padded length-63 word, Massey's iteration, Chien search over all 63 positions (a "correction" that lands in
// the zero padding is accepted exactly like the reference does), Forney error values.  The decoders are
// bounded-distance, so any correct formulation returns the reference's symbols; on failure the data is left untouched.
// All per-lane polynomial arrays live in LDS as [index][lane] (dynamic indexing of a private array would go to scratch).

#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_tables_isch.h"

namespace {
__device__ __forceinline__ uint32_t
golay_syndrome11(uint32_t cw) {
    cw &= 0x7fffffu;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        cw = (cw & 1u) ? ((cw ^ 0xAE3u) >> 1) : (cw >> 1);
    }
    return cw;
}
} // namespace

// builds the syndrome -> error pattern table (every pattern of weight <= 3 over 23 bits: 1+23+253+1771 = 2048)
__global__ void
k_golay_table(uint32_t* __restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x; // 0 .. 23*23*23-1, decoded as (a, b, c) with a <= b <= c
    if (idx == 0) {
        tab[0] = 0;
    }
    const int a = idx / 529, b = (idx / 23) % 23, c = idx % 23;
    if (a >= 23 || a > b || b > c) {
        return;
    }
    const uint32_t e = (1u << a) | (1u << b) | (1u << c); // equal indices collapse to lower weights
    tab[golay_syndrome11(e)] = e;
}

__global__ __launch_bounds__(256) void
k_golay24(uint8_t* __restrict__ data, const uint8_t* __restrict__ parity, int len, int n,
          const uint32_t* __restrict__ tab_g, uint8_t* __restrict__ status, int32_t* __restrict__ fixed, DdnSel sel) {
    const uint32_t* tab = tab_g; // (read in place - 8 KB, it stays in the L1: single-wave workgroups, no LDS; see DDN_WG)
    auto one = [&](long i) {
    uint8_t* d = data + (size_t)i * len;
    const uint8_t* p = parity + (size_t)i * 12;
    uint32_t cw = 0;
    bool valid = true;
    for (int k = 0; k < 12; k++) {
        const uint32_t v = p[k];
        valid &= v <= 1;
        cw |= (v & 1u) << (12 + k);
    }
    for (int k = 0; k < len; k++) {
        const uint32_t v = d[k];
        valid &= v <= 1;
        cw |= (v & 1u) << (12 - len + k);
    }
    int fx = 0, rc = 1;
    if (valid) {
        const uint32_t pbit = cw & 0x800000u;
        uint32_t w23 = cw & 0x7fffffu;
        const uint32_t e = tab[golay_syndrome11(w23)];
        if (e) {
            const int wt = __popc(e);
            bool fits = false;
            uint32_t r = e;
#pragma unroll
            for (int k = 0; k < 23; k++) {
                fits |= (r & 0xFFFu) == 0;
                r = ((r << 1) | (r >> 22)) & 0x7fffffu;
            }
            fx = fits ? wt : wt - 1;
            w23 ^= e;
        }
        cw = w23 | pbit;
        const bool odd = (__popc(cw) & 1) != 0;
        if (!(odd && (cw & 0x3fu) != 0)) {
            rc = 0;
            for (int k = 0; k < len; k++) {
                d[k] = (uint8_t)((cw >> (12 - len + k)) & 1u);
            }
        }
    }
    status[i] = (uint8_t)rc;
    if (fixed) {
        fixed[i] = fx;
    }
    };
    ddn_sel_for_each(sel, (long)n, one);
}

__global__ __launch_bounds__(64) void
k_rs63(uint8_t* __restrict__ data6, const uint8_t* __restrict__ parity6, int n_par, int n_data, int t, int n,
       uint8_t* __restrict__ status, DdnSel sel) {
    __shared__ uint8_t ex[128], lg[64];
    __shared__ uint8_t W[36][64];                   // received symbols
    __shared__ uint8_t S[17][64], Cc[18][64], Bb[18][64], Tt[18][64], Om[16][64], Pos[8][64];
    const int lane = threadIdx.x;
    if (lane == 0) {
        int x = 1;
        for (int i = 0; i < 63; i++) {
            ex[i] = (uint8_t)x;
            ex[i + 63] = (uint8_t)x;
            lg[x] = (uint8_t)i;
            x <<= 1;
            if (x & 0x40) {
                x ^= 0x43;
            }
        }
        ex[126] = ex[0];
        ex[127] = ex[1];
        lg[0] = 0;
    }
    __syncthreads();
    auto one = [&](long i) {
    auto gmul = [&](int a, int b) -> int { return (a && b) ? ex[lg[a] + lg[b]] : 0; };
    auto gdiv = [&](int a, int b) -> int { return a ? ex[lg[a] + 63 - lg[b]] : 0; };
    auto gpow = [&](int a, int e) -> int { return a ? ex[(lg[a] + e) % 63] : 0; }; // a * alpha^e, e >= 0
    const int nsym = n_par + n_data, n2 = 2 * t;
    uint8_t* dp = data6 + (size_t)i * n_data * 6;
    const uint8_t* pp = parity6 + (size_t)i * n_par * 6;
    for (int k = 0; k < nsym; k++) {
        const uint8_t* q = (k < n_par) ? (pp + 6 * k) : (dp + 6 * (k - n_par));
        int v = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (q[b] != 0);
        }
        W[k][lane] = (uint8_t)v;
    }
    // syndromes S_i = r(alpha^i), i = 1..2t
    int any = 0;
    for (int s = 1; s <= n2; s++) {
        int acc = 0;
        for (int j = 0; j < nsym; j++) {
            acc ^= gpow(W[j][lane], (s * j) % 63);
        }
        S[s][lane] = (uint8_t)acc;
        any |= acc;
    }
    int rc = 0;
    if (any) {
        // Massey
        for (int k = 0; k < 18; k++) {
            Cc[k][lane] = (k == 0);
            Bb[k][lane] = (k == 0);
        }
        int L = 0, m = 1, b = 1;
        for (int it = 0; it < n2; it++) {
            int d = S[it + 1][lane];
            for (int k = 1; k <= L; k++) {
                d ^= gmul(Cc[k][lane], S[it + 1 - k][lane]);
            }
            if (d == 0) {
                m++;
            } else {
                const int f = gdiv(d, b);
                const bool grow = 2 * L <= it;
                for (int k = 0; k < 18; k++) {
                    Tt[k][lane] = Cc[k][lane];
                }
                for (int k = 0; k + m < 18; k++) {
                    Cc[k + m][lane] ^= (uint8_t)gmul(f, Bb[k][lane]);
                }
                if (grow) {
                    L = it + 1 - L;
                    for (int k = 0; k < 18; k++) {
                        Bb[k][lane] = Tt[k][lane];
                    }
                    b = d;
                    m = 1;
                } else {
                    m++;
                }
            }
        }
        int deg = 0;
        for (int k = 17; k >= 0; k--) {
            if (Cc[k][lane]) {
                deg = k;
                break;
            }
        }
        if (L > t || deg != L) {
            rc = 1;
        } else {
            // Chien over the whole length-63 word: error at position p <-> C(alpha^-p) = 0
            int np = 0;
            for (int p = 0; p < 63; p++) {
                const int xi = (63 - p) % 63;
                int v = 0;
                for (int k = 0; k <= L; k++) {
                    v ^= gpow(Cc[k][lane], (k * xi) % 63);
                }
                if (v == 0) {
                    if (np < 8) {
                        Pos[np][lane] = (uint8_t)p;
                    }
                    np++;
                }
            }
            if (np != L) {
                rc = 1;
            } else {
                for (int k = 0; k < n2; k++) {
                    int v = 0;
                    for (int j = 0; j <= k && j <= L; j++) {
                        v ^= gmul(Cc[j][lane], S[k - j + 1][lane]);
                    }
                    Om[k][lane] = (uint8_t)v;
                }
                // all error values first (a zero derivative aborts with the word untouched), then apply
                int ev[8];
                for (int k = 0; k < 8; k++) {
                    ev[k] = 0;
                }
                for (int e = 0; e < L && rc == 0; e++) {
                    const int p = Pos[e][lane];
                    const int xi = (63 - p) % 63;
                    int num = 0, den = 0;
                    for (int k = 0; k < n2; k++) {
                        num ^= gpow(Om[k][lane], (k * xi) % 63);
                    }
                    for (int k = 1; k <= L; k += 2) {
                        den ^= gpow(Cc[k][lane], ((k - 1) * xi) % 63);
                    }
                    if (den == 0) {
                        rc = 1;
                    } else {
                        const int val = gdiv(num, den);
                        // static unrolled select keeps ev[] in registers
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            ev[k] = (k == e) ? val : ev[k];
                        }
                    }
                }
                if (rc == 0) {
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        if (e < L) {
                            const int p = Pos[e][lane];
                            if (p < nsym) {
                                W[p][lane] ^= (uint8_t)ev[e];
                            }
                        }
                    }
                    for (int k = 0; k < n_data; k++) {
                        const int v = W[n_par + k][lane];
#pragma unroll
                        for (int b = 0; b < 6; b++) {
                            dp[6 * k + b] = (uint8_t)((v >> (5 - b)) & 1);
                        }
                    }
                }
            }
        }
    } else {
        // clean word: the reference still rewrites the data bits from the symbols (normalises non-0/1 bytes)
        for (int k = 0; k < n_data; k++) {
            const int v = W[n_par + k][lane];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                dp[6 * k + b] = (uint8_t)((v >> (5 - b)) & 1);
            }
        }
    }
    if (rc != 0) {
        for (int k = 0; k < n_data; k++) {
            const int v = W[n_par + k][lane];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                dp[6 * k + b] = (uint8_t)((v >> (5 - b)) & 1);
            }
        }
    }
    status[i] = (uint8_t)rc;
    };
    ddn_sel_for_each(sel, (long)n, one);
}

// ---- RS errors-and-erasures with reliability-ranked erasures ------------------------------------------------------------
// == p25p1_rs_24_12_13_soft_reliability / _24_16_9_ / _36_20_17_ (src/protocol/p25/phase1/p25p1_rs_soft_reliability.cpp,
// p25p1_check_ldu.cpp:70-94, p25p1_check_hdu.cpp:48-69; DSDReedSolomon_*::decode_soft and ReedSolomon_63::decode_with_erasures,
// include/dsd-neo/fec/ReedSolomon.hpp:175-262,585-690,774-802; ranking p25p1_soft.cpp:86-168).
// Per codeword: hard decode; if that fails, the symbols are ranked by (reliability, position) and the n = 1, 2, ... weakest
// are declared erasures until an errors-and-erasures decode succeeds (n up to max(#below-threshold, t), capped at 2t).
// One attempt = erasure locator (grown by one factor per attempt), modified syndromes, Massey on the last 2t - n of
// them, 2v + n <= 2t, combined locator, Chien over the 63 positions (root count = degree, every erasure a root), errata
// values by Forney (the reference solves the equivalent Vandermonde system), and all 2t syndromes of the corrected word
// must vanish - updated from the received word's syndromes instead of recomputed.  On failure the data is untouched.
__global__ __launch_bounds__(64) void
k_rs63_soft(uint8_t* __restrict__ data6, const uint8_t* __restrict__ parity6, const uint8_t* __restrict__ data_rel,
            const uint8_t* __restrict__ parity_rel, int n_par, int n_data, int t, int threshold, int n,
            uint8_t* __restrict__ status) {
    __shared__ uint8_t ex[128], lg[64];
    __shared__ uint8_t W[36][64], S[17][64], G[18][64], M[17][64], Cc[18][64], Bb[18][64], Tt[18][64], Lam[18][64],
        Om[16][64], Loc[16][64], Val[16][64], Er[16][64];
    __shared__ uint16_t Key[36][64];
    const int lane = threadIdx.x;
    if (lane == 0) {
        int x = 1;
        for (int i = 0; i < 63; i++) {
            ex[i] = (uint8_t)x;
            ex[i + 63] = (uint8_t)x;
            lg[x] = (uint8_t)i;
            x <<= 1;
            if (x & 0x40) {
                x ^= 0x43;
            }
        }
        ex[126] = ex[0];
        ex[127] = ex[1];
        lg[0] = 0;
    }
    __syncthreads();
    const int i = blockIdx.x * 64 + lane;
    if (i >= n) {
        return;
    }
    auto gmul = [&](int a, int b) -> int { return (a && b) ? ex[lg[a] + lg[b]] : 0; };
    auto gdiv = [&](int a, int b) -> int { return a ? ex[lg[a] + 63 - lg[b]] : 0; };
    auto gpow = [&](int a, int e) -> int { return a ? ex[(lg[a] + e) % 63] : 0; };
    const int nsym = n_par + n_data, n2 = 2 * t;
    uint8_t* dp = data6 + (size_t)i * n_data * 6;
    const uint8_t* pp = parity6 + (size_t)i * n_par * 6;
    for (int k = 0; k < nsym; k++) {
        const uint8_t* q = (k < n_par) ? (pp + 6 * k) : (dp + 6 * (k - n_par));
        int v = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            v = (v << 1) | (q[b] != 0);
        }
        W[k][lane] = (uint8_t)v;
    }
    int any = 0;
    for (int s = 1; s <= n2; s++) {
        int acc = 0;
        for (int j = 0; j < nsym; j++) {
            acc ^= gpow(W[j][lane], (s * j) % 63);
        }
        S[s][lane] = (uint8_t)acc;
        any |= acc;
    }
    // Massey on sy[0..ns) (sy = S+1 for the hard attempt, M+n_er for an erasure attempt); returns the polynomial degree
    auto massey = [&](bool on_m, int off, int ns) -> int {
        for (int k = 0; k < 18; k++) {
            Cc[k][lane] = (k == 0);
            Bb[k][lane] = (k == 0);
        }
        auto sy = [&](int k) -> int { return on_m ? M[off + k][lane] : S[off + k][lane]; };
        int L = 0, m = 1, b = 1;
        for (int it = 0; it < ns; it++) {
            int d = sy(it);
            for (int k = 1; k <= L; k++) {
                d ^= gmul(Cc[k][lane], sy(it - k));
            }
            if (d == 0) {
                m++;
                continue;
            }
            const int f = gdiv(d, b);
            for (int k = 0; k < 18; k++) {
                Tt[k][lane] = Cc[k][lane];
            }
            for (int k = 0; k + m < 18; k++) {
                Cc[k + m][lane] ^= (uint8_t)gmul(f, Bb[k][lane]);
            }
            if (2 * L <= it) {
                L = it + 1 - L;
                for (int k = 0; k < 18; k++) {
                    Bb[k][lane] = Tt[k][lane];
                }
                b = d;
                m = 1;
            } else {
                m++;
            }
        }
        int deg = 0;
        for (int k = 17; k >= 0; k--) {
            if (Cc[k][lane]) {
                deg = k;
                break;
            }
        }
        return (L << 8) | deg;
    };
    // roots of Lam (degree ldeg) over all 63 positions -> Loc[]; returns the count
    auto chien = [&](int ldeg) -> int {
        int nl = 0;
        for (int p = 0; p < 63; p++) {
            const int xi = (63 - p) % 63;
            int v = 0;
            for (int k = 0; k <= ldeg; k++) {
                v ^= gpow(Lam[k][lane], (k * xi) % 63);
            }
            if (v == 0) {
                if (nl < 16) {
                    Loc[nl][lane] = (uint8_t)p;
                }
                nl++;
            }
        }
        return nl;
    };
    // errata values at Loc[0..nl) by Forney; false if a derivative vanishes
    auto forney = [&](int ldeg, int nl) -> bool {
        for (int k = 0; k < n2; k++) {
            int v = 0;
            for (int j = 0; j <= k && j <= ldeg; j++) {
                v ^= gmul(Lam[j][lane], S[k - j + 1][lane]);
            }
            Om[k][lane] = (uint8_t)v;
        }
        for (int e = 0; e < nl; e++) {
            const int xi = (63 - Loc[e][lane]) % 63;
            int num = 0, den = 0;
            for (int k = 0; k < n2; k++) {
                num ^= gpow(Om[k][lane], (k * xi) % 63);
            }
            for (int k = 1; k <= ldeg; k += 2) {
                den ^= gpow(Lam[k][lane], ((k - 1) * xi) % 63);
            }
            if (den == 0) {
                return false;
            }
            Val[e][lane] = (uint8_t)gdiv(num, den);
        }
        return true;
    };
    auto write_back = [&](int nl) {
        for (int e = 0; e < nl; e++) {
            const int p = Loc[e][lane];
            if (p < nsym) {
                W[p][lane] ^= Val[e][lane];
            }
        }
        for (int k = 0; k < n_data; k++) {
            const int v = W[n_par + k][lane];
#pragma unroll
            for (int b = 0; b < 6; b++) {
                dp[6 * k + b] = (uint8_t)((v >> (5 - b)) & 1);
            }
        }
    };
    int rc = 1;
    if (!any) {
        rc = 0;
        write_back(0);
    } else {
        // hard attempt
        const int r = massey(false, 1, n2);
        const int L = r >> 8, deg = r & 255;
        if (L <= t && deg == L) {
            for (int k = 0; k < 18; k++) {
                Lam[k][lane] = Cc[k][lane];
            }
            if (chien(L) == L && forney(L, L)) {
                rc = 0;
                write_back(L);
            }
        }
    }
    if (rc != 0) {
        // rank the symbols: (reliability, position) ascending; position = parity index, then n_par + data index
        int hits = 0;
        for (int k = 0; k < nsym; k++) {
            const int rel = (k < n_par) ? parity_rel[(size_t)i * n_par + k] : data_rel[(size_t)i * n_data + (k - n_par)];
            hits += rel < threshold;
            Key[k][lane] = (uint16_t)(rel * 64 + k);
        }
        int nr = hits > t ? hits : t;
        nr = nr > nsym ? nsym : nr;
        nr = nr > n2 ? n2 : nr;
        for (int j = 0; j < nr; j++) { // selection of the nr smallest keys, in order
            int bk = 0x7fffffff, bi = 0;
            for (int k = 0; k < nsym; k++) {
                const int kk = Key[k][lane];
                if (kk < bk) {
                    bk = kk;
                    bi = k;
                }
            }
            Key[bi][lane] = 0xFFFF;
            Er[j][lane] = (uint8_t)bi;
        }
        for (int k = 0; k < 18; k++) {
            G[k][lane] = (k == 0);
        }
        for (int ne = 1; ne <= nr && rc != 0; ne++) {
            // erasure locator grows by (1 + alpha^pos x)
            const int f = ex[Er[ne - 1][lane] % 63];
            for (int k = ne - 1; k >= 0; k--) {
                G[k + 1][lane] ^= (uint8_t)gmul(G[k][lane], f);
            }
            for (int k = 0; k < n2; k++) {
                int v = 0;
                for (int j = 0; j <= ne && j <= k; j++) {
                    v ^= gmul(G[j][lane], S[k - j + 1][lane]);
                }
                M[k][lane] = (uint8_t)v;
            }
            const int r = massey(true, ne, n2 - ne);
            const int udeg = r & 255;
            if (2 * udeg + ne > n2) {
                continue;
            }
            int gdeg = 0;
            for (int k = 17; k >= 0; k--) {
                if (G[k][lane]) {
                    gdeg = k;
                    break;
                }
            }
            if (gdeg + udeg > n2) {
                continue;
            }
            for (int k = 0; k < 18; k++) {
                Lam[k][lane] = 0;
            }
            for (int a = 0; a <= gdeg; a++) {
                for (int b = 0; b <= udeg; b++) {
                    Lam[a + b][lane] ^= (uint8_t)gmul(G[a][lane], Cc[b][lane]);
                }
            }
            int ldeg = 0;
            for (int k = 17; k >= 0; k--) {
                if (Lam[k][lane]) {
                    ldeg = k;
                    break;
                }
            }
            const int nl = (ldeg > 0) ? chien(ldeg) : 0;
            if (nl != ldeg || nl > n2) {
                continue;
            }
            bool cover = true;
            for (int e = 0; e < ne; e++) {
                bool ok = false;
                for (int k = 0; k < nl; k++) {
                    ok |= (Loc[k][lane] == Er[e][lane]);
                }
                cover &= ok;
            }
            if (!cover || !forney(ldeg, nl)) {
                continue;
            }
            // every syndrome of the corrected word must vanish: S_s ^ sum_k val_k alpha^(s loc_k)
            int resid = 0;
            for (int s2 = 1; s2 <= n2; s2++) {
                int v = S[s2][lane];
                for (int k = 0; k < nl; k++) {
                    v ^= gpow(Val[k][lane], (s2 * Loc[k][lane]) % 63);
                }
                resid |= v;
            }
            if (resid) {
                continue;
            }
            rc = 0;
            write_back(nl);
        }
    }
    status[i] = (uint8_t)rc;
}

// ---- soft (Chase) variants: src/protocol/p25/phase1/p25p1_soft.cpp:175-593 ------------------------------------------
// hamming_10_6_3_soft: seed with the hard decode, then flip every subset of <= 2 of the 5 least reliable bits and keep
// valid codewords; check_and_fix_golay_24_{6,12}_soft: seed with the hard decode, then every subset of <= 4 of the 8
// least reliable bits through the hard decoder.  Penalty = clamped reliability summed over the bits where the
// re-encoded codeword differs from the received word, ties to fewer differing bits, masks in increasing order with
// strict improvement only; a corrected hard decode wins unless the best candidate beats it by more than 8.
namespace {
__device__ __forceinline__ int
clamp255d(int v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// returns 0 and the corrected 24-bit word, or 1 (odd overall parity with non-zero low six bits: DSDGolay24::decode_*)
__device__ __forceinline__ int
golay_hard_word(uint32_t cw, const uint32_t* tab, uint32_t* out, int* fx) {
    const uint32_t pbit = cw & 0x800000u;
    uint32_t w23 = cw & 0x7fffffu;
    const uint32_t e = tab[golay_syndrome11(w23)];
    *fx = 0;
    if (e) {
        const int wt = __popc(e);
        bool fits = false;
        uint32_t r = e;
#pragma unroll
        for (int k = 0; k < 23; k++) {
            fits |= (r & 0xFFFu) == 0;
            r = ((r << 1) | (r >> 22)) & 0x7fffffu;
        }
        *fx = fits ? wt : wt - 1;
        w23 ^= e;
    }
    const uint32_t c2 = w23 | pbit;
    *out = c2;
    return ((__popc(c2) & 1) && (c2 & 0x3fu)) ? 1 : 0;
}

__device__ __forceinline__ uint32_t
golay_encode_word(uint32_t d12) {
    uint32_t cw = d12 & 0xFFFu;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        cw = (cw & 1u) ? ((cw ^ 0xAE3u) >> 1) : (cw >> 1);
    }
    uint32_t w = (cw << 12) | (d12 & 0xFFFu);
    if (__popc(w) & 1) {
        w ^= 0x800000u;
    }
    return w;
}
} // namespace

template <int L>
__global__ __launch_bounds__(256) void
k_golay24_soft(uint8_t* __restrict__ data, const uint8_t* __restrict__ parity, const int32_t* __restrict__ reliab, int n,
               const uint32_t* __restrict__ tab_g, uint8_t* __restrict__ status, int32_t* __restrict__ fixed) {
    constexpr int N = L + 12, SH = 12 - L;
    __shared__ uint32_t tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) {
        tab[i] = tab_g[i];
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint8_t* d = data + (size_t)i * L;
    const uint8_t* p = parity + (size_t)i * 12;
    const int32_t* rp = reliab + (size_t)i * N;
    uint32_t orig = 0;
    bool valid = true;
    int rel[N], key[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t v = (k < L) ? d[k] : p[k - L];
        valid &= v <= 1;
        orig |= (v & 1u) << (k + SH);
        rel[k] = clamp255d(rp[k]);
        key[k] = rel[k] * 64 + k;
    }
    int rc = 1, fx_out = 0;
    if (valid) {
        const uint32_t nmask = ((1u << N) - 1u) << SH;
        auto penalty = [&](uint32_t diff, int* nd) {
            int pen = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                pen += ((diff >> (k + SH)) & 1u) ? rel[k] : 0;
            }
            *nd = __popc(diff);
            return pen;
        };
        int best_pen = 999999, best_fixed = 0, hard_pen = 999999, hard_fixed = 0;
        bool found = false, hard_valid = false, hard_corr = false;
        uint32_t best_dw = 0, hard_dw = 0;
        {
            uint32_t c2;
            if (golay_hard_word(orig, tab, &c2, &hard_fixed) == 0) {
                hard_dw = c2 & (0xFFFu & ~((1u << SH) - 1u));
                int nd;
                hard_pen = penalty((orig ^ golay_encode_word(hard_dw)) & nmask, &nd);
                best_pen = hard_pen;
                best_fixed = nd;
                best_dw = hard_dw;
                hard_valid = true;
                hard_corr = hard_fixed > 0;
                found = true;
            }
        }
        // the 8 least reliable positions in (reliability, position) order -> the word bits they flip
        uint32_t flip[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int bk = key[0], bi = 0;
#pragma unroll
            for (int k = 1; k < N; k++) {
                const bool lt = key[k] < bk;
                bk = lt ? key[k] : bk;
                bi = lt ? k : bi;
            }
#pragma unroll
            for (int k = 0; k < N; k++) {
                key[k] = (k == bi) ? 0x7fffffff : key[k];
            }
            flip[j] = 1u << (bi + SH);
        }
        for (int mask = 0; mask < 256; mask++) {
            if (__popc((unsigned)mask) > 4) {
                continue;
            }
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                x ^= (mask & (1 << b)) ? flip[b] : 0u;
            }
            uint32_t c2;
            int cf;
            if (golay_hard_word(orig ^ x, tab, &c2, &cf) != 0) {
                continue;
            }
            const uint32_t dw = c2 & (0xFFFu & ~((1u << SH) - 1u));
            int nd;
            const int pen = penalty((orig ^ golay_encode_word(dw)) & nmask, &nd);
            if (pen < best_pen || (pen == best_pen && nd < best_fixed)) {
                best_pen = pen;
                best_fixed = nd;
                best_dw = dw;
                found = true;
            }
        }
        if (found) {
            rc = 0;
            uint32_t out_dw = best_dw;
            fx_out = best_fixed;
            if (hard_valid && hard_corr && best_dw != hard_dw && best_pen + 8 >= hard_pen) {
                out_dw = hard_dw;
                fx_out = hard_fixed;
            }
#pragma unroll
            for (int k = 0; k < L; k++) {
                d[k] = (uint8_t)((out_dw >> (k + SH)) & 1u);
            }
        }
    }
    status[i] = (uint8_t)rc;
    if (fixed) {
        fixed[i] = fx_out;
    }
}

__global__ __launch_bounds__(256) void
k_hamming_10_6_3_soft(const uint8_t* __restrict__ bits, const int32_t* __restrict__ reliab, int n,
                      uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* b = bits + (size_t)i * 10;
    const int32_t* rp = reliab + (size_t)i * 10;
    int rel[10], key[10];
    uint32_t orig = 0; // bit (9 - k) = element k: (data << 4) | check, data[0] the MSB
    bool valid = true;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const uint32_t v = b[k];
        valid &= v <= 1;
        orig |= (v & 1u) << (9 - k);
        rel[k] = clamp255d(rp[k]);
        key[k] = rel[k] * 64 + k;
    }
    // hamming_10_6_3_decode on a word: 0 valid, 1 corrected (data bits only are rewritten), 2 uncorrectable
    auto hard = [&](uint32_t w, uint32_t* fixed_w) -> int {
        const uint32_t m[4] = {0x398u, 0x354u, 0x2E2u, 0x1E1u};
        int syn = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            syn = (syn << 1) | (__popc(w & m[k]) & 1);
        }
        *fixed_w = w;
        if (syn == 0) {
            return 0;
        }
        // syndrome -> flipped word bit (4..9 = data bits), -1 = two or more errors; 0..3 = a parity bit
        // -1 encoded as 15 in a packed nibble table: {-, 0, 1, 5, 2, x, x, 6, 3, x, x, 7, 4, 8, 9, x}
        const uint32_t nib = (uint32_t)((0xF9847FF36FF2510Full >> (4 * syn)) & 15u);
        const int bb = (nib == 15u) ? -1 : (int)nib;
        if (bb < 0) {
            return 2;
        }
        if (bb >= 4) {
            *fixed_w = w ^ (1u << bb);
        }
        return 1;
    };
    auto encode = [&](uint32_t w) -> uint32_t {
        const uint32_t d0 = (w >> 9) & 1, d1 = (w >> 8) & 1, d2 = (w >> 7) & 1, d3 = (w >> 6) & 1, d4 = (w >> 5) & 1,
                       d5 = (w >> 4) & 1;
        const uint32_t p0 = d0 ^ d1 ^ d2 ^ d5, p1 = d0 ^ d1 ^ d3 ^ d5, p2 = d0 ^ d2 ^ d3 ^ d4, p3 = d1 ^ d2 ^ d3 ^ d4;
        return (w & 0x3F0u) | (p0 << 3) | (p1 << 2) | (p2 << 1) | p3;
    };
    auto penalty = [&](uint32_t diff) {
        int pen = 0;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            pen += ((diff >> (9 - k)) & 1u) ? rel[k] : 0;
        }
        return pen;
    };
    int rc = 2;
    uint32_t outw = orig;
    if (valid) {
        int best_pen = 999999, best_flips = 99, hard_pen = 999999;
        bool found = false, hard_valid = false, hard_corr = false;
        uint32_t best = 0, hardw = 0;
        {
            uint32_t fw;
            const int r = hard(orig, &fw);
            if (r == 0 || r == 1) {
                hardw = encode(fw);
                hard_valid = true;
                hard_corr = (r == 1);
                hard_pen = penalty(orig ^ hardw);
                best_pen = hard_pen;
                best_flips = __popc(orig ^ hardw);
                best = hardw;
                found = true;
            }
        }
        uint32_t flip[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            int bk = key[0], bi = 0;
#pragma unroll
            for (int k = 1; k < 10; k++) {
                const bool lt = key[k] < bk;
                bk = lt ? key[k] : bk;
                bi = lt ? k : bi;
            }
#pragma unroll
            for (int k = 0; k < 10; k++) {
                key[k] = (k == bi) ? 0x7fffffff : key[k];
            }
            flip[j] = 1u << (9 - bi);
        }
        for (int mask = 0; mask < 32; mask++) {
            const int nf = __popc((unsigned)mask);
            if (nf > 2) {
                continue;
            }
            uint32_t x = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                x ^= (mask & (1 << q)) ? flip[q] : 0u;
            }
            uint32_t fw;
            if (hard(orig ^ x, &fw) != 0) {
                continue;
            }
            const uint32_t c = orig ^ x;
            const int pen = penalty(orig ^ c);
            if (pen < best_pen || (pen == best_pen && nf < best_flips)) {
                best_pen = pen;
                best_flips = nf;
                best = c;
                found = true;
            }
        }
        if (found) {
            if (hard_valid && hard_corr && best != hardw && best_pen + 8 >= hard_pen) {
                outw = hardw;
                rc = 1;
            } else {
                outw = best;
                rc = (best == orig) ? 0 : 1;
            }
        }
    }
    uint8_t* o = out + (size_t)i * 10;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            o[k] = (uint8_t)((outw >> (9 - k)) & 1u);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            o[k] = b[k];
        }
    }
    status[i] = (uint8_t)rc;
}

// syndrome -> error pattern table, built once per DEVICE (immutable afterwards): the pointer is device memory, so a
// process that moves to another GPU with hipSetDevice() must not reuse the first device's table
static hipError_t
golay_table(uint32_t** out, hipStream_t st) {
    static std::mutex mu;
    static uint32_t* tabs[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        return e;
    }
    if (dev < 0 || dev >= 64) {
        return hipErrorInvalidDevice;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!tabs[dev]) {
        uint32_t* t = nullptr;
        e = hipMalloc(&t, 2048 * sizeof(uint32_t));
        if (e != hipSuccess) {
            return e;
        }
        hipLaunchKernelGGL(k_golay_table, dim3((23 * 23 * 23 + 255) / 256), dim3(256), 0, st, t);
        e = hipGetLastError();
        if (e == hipSuccess) {
            e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) {
            (void)hipFree(t);
            return e;
        }
        tabs[dev] = t;
    }
    *out = tabs[dev];
    return hipSuccess;
}

extern "C" hipError_t
ddn_dev_golay24_soft(uint8_t* data, const uint8_t* parity, const int32_t* reliab, int len, int n, uint8_t* status,
                     int32_t* fixed, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    uint32_t* tab = nullptr;
    hipError_t e = golay_table(&tab, st);
    if (e != hipSuccess) {
        return e;
    }
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (len == 6) {
        hipLaunchKernelGGL((k_golay24_soft<6>), grid, blk, 0, st, data, parity, reliab, n, (const uint32_t*)tab, status, fixed);
    } else {
        hipLaunchKernelGGL((k_golay24_soft<12>), grid, blk, 0, st, data, parity, reliab, n, (const uint32_t*)tab, status, fixed);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_hamming_10_6_3_soft(const uint8_t* bits, const int32_t* reliab, int n, uint8_t* out, uint8_t* status,
                            hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_hamming_10_6_3_soft, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bits, reliab, n, out,
                       status);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_golay24(uint8_t* data, const uint8_t* parity, int len, int n, uint8_t* status, int32_t* fixed, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    uint32_t* tab = nullptr;
    hipError_t e0 = golay_table(&tab, st);
    if (e0 != hipSuccess) {
        return e0;
    }
    const DdnSel sel = ddn_sel_for(len == 6 ? 36 : 12);
    hipLaunchKernelGGL(k_golay24, dim3(ddn_sel_grid(&sel, ((unsigned long)n + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, data, parity, len, n,
                       (const uint32_t*)tab, status, fixed, sel);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_rs63(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t, int n, uint8_t* status,
             hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(1);
    hipLaunchKernelGGL(k_rs63, dim3(ddn_sel_grid(&sel, ((unsigned long)n + 63) / 64)), dim3(64), 0, st, data6, parity6, n_par, n_data, t, n,
                       status, sel);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_rs63_soft(uint8_t* data6, const uint8_t* parity6, const uint8_t* data_rel, const uint8_t* parity_rel, int n_par,
                  int n_data, int t, int threshold, int n, uint8_t* status, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_rs63_soft, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, data6, parity6, data_rel, parity_rel,
                       n_par, n_data, t, threshold, n, status);
    return hipGetLastError();
}

// ---- P25 Phase 2 RS(63,35) sections with caller-given erasures --------------------------------------------------------------
// == ez_rs28_ess / _facch / _sacch (src/fec/ez.cpp:104-281) over the vendored ezpwd RS<63,35> (GF(64) from x^6+x+1, roots
// alpha^1..alpha^28; decoder src/third_party/ezpwd/rs_base:1380-1720): block position p carries the coefficient of
// x^(62-p); syndromes, erasure locator, Berlekamp's iteration from step n_erasures + 1 with the length rule
// 2 L <= r + n_erasures - 1, Chien search over the 63 positions stopped at deg(lambda) roots, Forney from the last root to
// the first.  A root count short of the degree, a zero derivative or a non-zero value inside the pad fails the decode (-1)
// and leaves the section as received (the reference decodes a copy and copies back on a positive count only).
// One section per thread, its polynomials in LDS columns.
enum { DDN_RS28_PROBE = -3, DDN_RS28_SYN = -5 };
__global__ __launch_bounds__(64) void
k_rs28(int kind, uint8_t* __restrict__ payload_bits, const uint8_t* __restrict__ parity_bits,
       const int8_t* __restrict__ erasures, const uint8_t* __restrict__ n_erasures, int n, int32_t* __restrict__ status,
       int attempt, int n_fixed, uint8_t* __restrict__ used_dynamic, int loop_to, int ess_rule, int probe_A,
       int32_t* __restrict__ probe_res, uint8_t* __restrict__ syn, uint8_t* __restrict__ stage) {
    // probe_A > 0: the retries of a section are independent of each other (attempt a decodes the received block with the first
    // n_fixed + a erasures of the ranked list; the reference takes the first that succeeds), so they run side by side instead of
    // one after the other: the PROBE pass (attempt == DDN_RS28_PROBE) gives every (section, attempt 1 .. probe_A) pair a thread that
    // decodes on its own and leaves its verdict in probe_res and, when it succeeded, the corrected data symbols in stage
    // [pair][n_data]; k_rs28_pick then takes, per failed section, the first attempt that succeeded - one decode deep instead of up to
    // probe_A.  The syndromes do not depend on the attempt: a pass of its own (attempt == DDN_RS28_SYN) leaves those of every failed
    // section in syn [n][28], and a probe thread starts from them - it reads the code word only when its decode reaches Forney's step.
    // attempt >= 0 (the Phase 2 burst stage's ranked retries, p25p2_decode_facch_ranked()): this launch decodes with the first
    // n_fixed + attempt erasures of each list - attempt 0 every section, attempt a > 0 only the sections that have failed so far
    // and whose list (n_erasures = its full length) reaches that far; a failed decode leaves the payload as received, so every
    // retry starts from the original like the reference's
    constexpr int R = 28;
    __shared__ uint8_t ex[128], lg[64];
    __shared__ uint8_t Cw[63][64], Sy[R][64], La[R + 1][64], Bp[R + 1][64], Tp[R + 1][64], Om[R][64], Rt[R][64];
    const int lane = threadIdx.x;
    if (lane == 0) {
        int x = 1;
        for (int i = 0; i < 63; i++) {
            ex[i] = (uint8_t)x;
            ex[i + 63] = (uint8_t)x;
            lg[x] = (uint8_t)i;
            x <<= 1;
            if (x & 0x40) {
                x ^= 0x43;
            }
        }
        ex[126] = ex[0];
        ex[127] = ex[1];
        lg[0] = 0;
    }
    __syncthreads();
    const long gid = (long)blockIdx.x * 64 + lane;
    const bool probe = probe_A > 0 && attempt == DDN_RS28_PROBE;
    const bool synp = probe_A > 0 && attempt == DDN_RS28_SYN;
    const int i = probe ? (int)(gid / probe_A) : (int)gid;
    if (i >= n || (probe && gid >= (long)n * probe_A)) {
        return;
    }
    if (probe) {
        attempt = 1 + (int)(gid % probe_A);
        loop_to = attempt;
        if (status[i] >= 0) {
            return;
        }
        if (n_fixed + attempt > (int)n_erasures[i]) {
            probe_res[gid] = -2;
            return;
        }
    } else if (synp) {
        if (status[i] >= 0) {
            return;
        }
    }
    if (attempt > 0 && (status[i] >= 0 || n_fixed + attempt > (int)n_erasures[i])) {
        return;
    }
    auto gmul = [&](int a, int b) -> int { return (a && b) ? ex[lg[a] + lg[b]] : 0; };
    auto gdiv = [&](int a, int b) -> int { return a ? ex[lg[a] + 63 - lg[b]] : 0; };
    auto gpow = [&](int a, int e) -> int { return a ? ex[(lg[a] + e) % 63] : 0; }; // a * alpha^e
    const int n_data = kind == 0 ? 16 : (kind == 1 ? 26 : 30), n_par = kind == 0 ? 28 : (kind == 1 ? 19 : 22);
    const int first = kind == 0 ? 19 : (kind == 1 ? 9 : 5), pad = kind == 0 ? 19 : 0;
    uint8_t* pl = payload_bits + (size_t)i * n_data * 6;
    const uint8_t* pa = parity_bits + (size_t)i * n_par * 6;
    auto load_cw = [&]() { // the block as received, one symbol per LDS row
        for (int p = 0; p < 63; p++) {
            int v = 0;
            const uint8_t* q = nullptr;
            if (p >= first && p < 35) {
                q = pl + 6 * (p - first);
            } else if (p >= 35 && p < 35 + n_par) {
                q = pa + 6 * (p - 35);
            }
            if (q) {
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    v = (v << 1) | (q[b] != 0);
                }
            }
            Cw[p][lane] = (uint8_t)v;
        }
    };
    int any = 0;
    if (probe) { // the section failed its plain decode: its syndromes are on file and not all zero
        for (int r = 0; r < R; r++) {
            Sy[r][lane] = syn[(size_t)i * R + r];
        }
        any = 1;
    } else {
        load_cw();
        for (int r = 0; r < R; r++) {
            int v = 0;
            for (int p = 0; p < 63; p++) {
                v = gpow(v, r + 1) ^ Cw[p][lane];
            }
            Sy[r][lane] = (uint8_t)v;
            any |= v;
        }
    }
    if (synp) {
        for (int r = 0; r < R; r++) {
            syn[(size_t)i * R + r] = Sy[r][lane];
        }
        return;
    }
    if (!any) {
        if (probe) {
            probe_res[gid] = 0;
        } else {
            status[i] = 0;
        }
        return;
    }
    // loop_to >= attempt: this thread goes on to the next attempt itself (n_fixed + attempt + 1 erasures, ...) until one decodes,
    // the list ends or loop_to is reached; ess_rule: a plain decode (attempt 0) that located 15 or more symbols is not taken
    // (p25p2_ess_decode_with_soft_erasures(), p25p2_frame.c:1063-1066) - the retries start from the received bits
    const int att_last = loop_to > attempt ? loop_to : attempt;
    int result = -1;
    for (int att = attempt;; att++) {
        if (att > attempt) { // a failed attempt may have touched the block: take it again as received
            load_cw();
        }
        int n_er = att >= 0 ? n_fixed + att : (n_erasures ? n_erasures[i] : 0);
        n_er = n_er > R ? R : n_er;
        result = -1;
        do {
            for (int k = 0; k <= R; k++) {
                La[k][lane] = k == 0 ? 1 : 0;
            }
            for (int e = 0; e < n_er; e++) {
                const int bp = (int)erasures[(size_t)i * R + e] + pad; // block position
                const int X = ex[(((62 - bp) % 63) + 63) % 63];
                for (int j = e + 1; j > 0; j--) {
                    La[j][lane] ^= (uint8_t)gmul(La[j - 1][lane], X);
                }
            }
            for (int k = 0; k <= R; k++) {
                Bp[k][lane] = La[k][lane];
            }
            int el = n_er;
            for (int r = n_er + 1; r <= R; r++) {
                int d = 0;
                for (int k = 0; k < r; k++) {
                    d ^= gmul(La[k][lane], Sy[r - k - 1][lane]);
                }
                if (d == 0) {
                    for (int k = R; k > 0; k--) {
                        Bp[k][lane] = Bp[k - 1][lane];
                    }
                    Bp[0][lane] = 0;
                    continue;
                }
                Tp[0][lane] = La[0][lane];
                for (int k = 0; k < R; k++) {
                    Tp[k + 1][lane] = La[k + 1][lane] ^ (uint8_t)gmul(d, Bp[k][lane]);
                }
                if (2 * el <= r + n_er - 1) {
                    el = r + n_er - el;
                    for (int k = 0; k <= R; k++) {
                        Bp[k][lane] = (uint8_t)gdiv(La[k][lane], d);
                    }
                } else {
                    for (int k = R; k > 0; k--) {
                        Bp[k][lane] = Bp[k - 1][lane];
                    }
                    Bp[0][lane] = 0;
                }
                for (int k = 0; k <= R; k++) {
                    La[k][lane] = Tp[k][lane];
                }
            }
            int deg = 0;
            for (int k = 0; k <= R; k++) {
                deg = La[k][lane] ? k : deg;
            }
            int count = 0;
            for (int r = 1; r <= 63 && count < deg; r++) {
                int q = 1;
                for (int j = 1; j <= deg; j++) {
                    q ^= gpow(La[j][lane], r * j);
                }
                if (q == 0) {
                    Rt[count][lane] = (uint8_t)r;
                    count++;
                }
            }
            if (count != deg || deg == 0) {
                break;
            }
            for (int k = 0; k < deg; k++) {
                int v = 0;
                for (int j = 0; j <= k; j++) {
                    v ^= gmul(Sy[k - j][lane], La[j][lane]);
                }
                Om[k][lane] = (uint8_t)v;
            }
            const int top = (deg < R - 1 ? deg : R - 1) & ~1;
            bool ok = true;
            if (probe) {
                load_cw();
            }
            for (int j = count - 1; j >= 0 && ok; j--) {
                const int rt = Rt[j][lane], loc = rt - 1;
                int num = 0, den = 0;
                for (int k = 0; k < deg; k++) {
                    num ^= gpow(Om[k][lane], k * rt);
                }
                for (int k = top; k >= 0; k -= 2) {
                    den ^= gpow(La[k + 1][lane], k * rt);
                }
                if (den == 0 || (num != 0 && loc < pad)) {
                    ok = false;
                } else if (num != 0) {
                    Cw[loc][lane] ^= (uint8_t)gdiv(num, den);
                }
            }
            if (ok) {
                result = count;
            }
        } while (0);
        const bool rejected = ess_rule && att == 0 && result >= 15;
        if (result >= 0 && !rejected) {
            for (int k = 0; k < (probe ? n_data : 0); k++) {
                stage[(size_t)gid * n_data + k] = Cw[first + k][lane];
            }
            for (int k = 0; k < (probe ? 0 : n_data); k++) {
                const int v = Cw[first + k][lane];
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    pl[6 * k + b] = (uint8_t)((v >> (5 - b)) & 1);
                }
            }
            if (att > 0 && used_dynamic && !probe) {
                used_dynamic[i] = 1;
            }
            break;
        }
        result = -1;
        if (att >= att_last || n_fixed + att + 1 > (n_erasures ? (int)n_erasures[i] : 0)) {
            break;
        }
    }
    if (probe) {
        probe_res[gid] = result;
    } else {
        status[i] = result;
    }
}

// per failed section: the first attempt whose probe succeeded - its corrected data symbols become the payload, its count the status
__global__ __launch_bounds__(256) void
k_rs28_pick(int kind, uint8_t* __restrict__ payload_bits, int n, int32_t* __restrict__ status, uint8_t* __restrict__ used_dynamic, int probe_A,
            const int32_t* __restrict__ probe_res, const uint8_t* __restrict__ stage) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || status[i] >= 0) {
        return;
    }
    const int n_data = kind == 0 ? 16 : (kind == 1 ? 26 : 30);
    for (int k = 0; k < probe_A; k++) {
        const int r = probe_res[(long)i * probe_A + k];
        if (r == -2) {
            return; // the list ends here: every due retry failed, the payload stays as received
        }
        if (r >= 0) {
            const uint8_t* sy = stage + ((size_t)i * probe_A + k) * n_data;
            uint8_t* pl = payload_bits + (size_t)i * n_data * 6;
            for (int d = 0; d < n_data; d++) {
                const int v = sy[d];
#pragma unroll
                for (int b = 0; b < 6; b++) {
                    pl[6 * d + b] = (uint8_t)((v >> (5 - b)) & 1);
                }
            }
            status[i] = r;
            if (used_dynamic) {
                used_dynamic[i] = 1;
            }
            return;
        }
    }
}

// the retries of every failed section side by side (see k_rs28): syndromes on file, probe pass over (section, attempt) pairs, pick
static hipError_t
rs28_retries(int kind, uint8_t* payload_bits, const uint8_t* parity_bits, const int8_t* erasures28, const uint8_t* n_total, int n,
             int32_t* status, int n_fixed, uint8_t* used_dynamic, int max_add, hipStream_t st) {
    // Side by side costs more decodes in total (every due attempt of a failed section runs, not only those up to the first success)
    // and buys depth - a decode is ~0.5 ms deep for its thread however many run: measured on one MI355X, 4096 channels x 6 groups of
    // Phase 2 traffic (12 k - 30 k sections per call) and 65 536 ESS sections (lists up to 28 deep) gain, 65 536 FACCH bursts (655 k
    // pairs, lists 10 deep: the device is already full) lose - so very large batches of short lists keep the loop in the thread.
    const long pairs = (long)n * max_add;
    if (pairs > 600000 && max_add < 28) {
        hipLaunchKernelGGL(k_rs28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, kind, payload_bits, parity_bits, erasures28, n_total, n,
                           status, 1, n_fixed, used_dynamic, max_add, 0, 0, (int32_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr);
        return hipGetLastError();
    }
    const int n_data = kind == 0 ? 16 : (kind == 1 ? 26 : 30);
    int32_t* res = nullptr;
    hipError_t e = hipMallocAsync((void**)&res, (size_t)pairs * sizeof(int32_t) + (size_t)n * 28 + (size_t)pairs * n_data, st);
    if (e != hipSuccess) {
        return e;
    }
    uint8_t* syn = (uint8_t*)(res + pairs);
    uint8_t* stage = syn + (size_t)n * 28;
    hipLaunchKernelGGL(k_rs28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, kind, payload_bits, parity_bits, erasures28, n_total, n,
                       status, DDN_RS28_SYN, n_fixed, used_dynamic, 0, 0, max_add, res, syn, stage);
    hipLaunchKernelGGL(k_rs28, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, st, kind, payload_bits, parity_bits, erasures28, n_total, n,
                       status, DDN_RS28_PROBE, n_fixed, used_dynamic, 0, 0, max_add, res, syn, stage);
    hipLaunchKernelGGL(k_rs28_pick, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kind, payload_bits, n, status, used_dynamic, max_add, res,
                       stage);
    e = hipGetLastError();
    const hipError_t f = hipFreeAsync(res, st);
    return e != hipSuccess ? e : f;
}

// ---- P25 Phase 2 FACCH / SACCH burst gather + ranked erasure list -----------------------------------------------------------------
// p25p2_process_facchc() / process_SACCHs() (src/protocol/p25/phase2/p25p2_frame.c:473-495,652-671): where the RS(63,35) section's
// payload and parity bits sit in the 360 bits of a timeslot (around the DUID fields and, for the FACCH, the sync);
// p25p2_facch_soft_erasures() / p25p2_sacch_soft_erasures() (p25p2_soft.c:40-108,255-329): reliability of a 6-bit symbol = the least
// min(|LLR|, 255) of its bits; the symbols not erased yet ordered by (reliability, position); as many as fall below the threshold, at
// least 5 / 8, at most 10 / 16, appended to the fixed erasures (the unsent and punctured symbols).  One burst per thread.
__global__ void
k_p2_xcch_gather(int kind, const uint8_t* __restrict__ bits360, const int16_t* __restrict__ llr360, int n, int threshold,
                 uint8_t* __restrict__ payload_bits, uint8_t* __restrict__ parity_bits, int8_t* __restrict__ erasures28,
                 uint8_t* __restrict__ n_total, uint8_t* __restrict__ used_dynamic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const int n_pl = kind == 0 ? 156 : 180, n_pa = kind == 0 ? 114 : 132;
    const int first = kind == 0 ? 9 : 5, max_add = kind == 0 ? 10 : 16, min_add = kind == 0 ? 5 : 8;
    const uint8_t* b = bits360 + (size_t)i * 360;
    const int16_t* l = llr360 + (size_t)i * 360;
    auto pos_pl = [&](int k) { return kind == 0 ? (k < 72 ? k + 2 : (k < 134 ? k - 72 + 76 : k - 134 + 180)) : (k < 72 ? k + 2 : k - 72 + 76); };
    auto pos_pa = [&](int k) { return kind == 0 ? (k < 42 ? k + 202 : k - 42 + 246) : (k < 60 ? k + 184 : k - 60 + 246); };
    for (int k = 0; k < n_pl; k++) {
        payload_bits[(size_t)i * n_pl + k] = b[pos_pl(k)] & 1;
    }
    for (int k = 0; k < n_pa; k++) {
        parity_bits[(size_t)i * n_pa + k] = b[pos_pa(k)] & 1;
    }
    // fixed erasures: positions 0 .. first - 1 and 35 + n_pa / 6 .. 62; every transmitted symbol is a candidate
    int8_t* er = erasures28 + (size_t)i * 28;
    int ne = 0;
    for (int p = 0; p < first; p++) {
        er[ne++] = (int8_t)p;
    }
    for (int p = 35 + n_pa / 6; p < 63; p++) {
        er[ne++] = (int8_t)p;
    }
    __shared__ uint16_t keys[52][64]; // reliability << 8 | position: the order of sort_candidates(); a column per thread (64 per block)
#define key(j) keys[j][threadIdx.x]
    int nc = 0;
    for (int part = 0; part < 2; part++) {
        const int nh = (part == 0 ? n_pl : n_pa) / 6;
        for (int hb = 0; hb < nh; hb++) {
            int r = 255;
            for (int q = 0; q < 6; q++) {
                int v = l[part == 0 ? pos_pl(6 * hb + q) : pos_pa(6 * hb + q)];
                v = v < 0 ? -v : v;
                v = v > 255 ? 255 : v;
                r = v < r ? v : r;
            }
            key(nc++) = (uint16_t)((r << 8) | ((part == 0 ? first : 35) + hb));
        }
    }
    int add = 0;
    for (int k = 0; k < nc; k++) {
        add += (key(k) >> 8) < threshold ? 1 : 0;
    }
    add = add < min_add ? min_add : add;
    add = add > max_add ? max_add : add;
    add = add > nc ? nc : add;
    for (int k = 0; k < add; k++) { // the `add` smallest keys, in order
        int best = k;
        for (int j = k + 1; j < nc; j++) {
            best = key(j) < key(best) ? j : best;
        }
        const uint16_t t = key(k);
        key(k) = key(best);
        key(best) = t;
        er[ne++] = (int8_t)(key(k) & 0xFF);
    }
    for (int k = ne; k < 28; k++) {
        er[k] = 0;
    }
    n_total[i] = (uint8_t)ne;
    used_dynamic[i] = 0;
#undef key
}

extern "C" hipError_t
ddn_dev_rs28(int kind, uint8_t* payload_bits, const uint8_t* parity_bits, const int8_t* erasures, const uint8_t* n_erasures,
             int n, int32_t* status, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_rs28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, kind, payload_bits, parity_bits, erasures,
                       n_erasures, n, status, -1, 0, (uint8_t*)nullptr, -1, 0, 0, (int32_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr);
    return hipGetLastError();
}

// ---- P25 Phase 2 MAC PDU checksums ------------------------------------------------------------------------------------------------------
// p25p2_xcch_validate_facch_crc() / _sacch_crc() (src/protocol/p25/phase2/p25p2_xcch.c:444-497): CRC12 over the section's payload less
// its last 12 bits (crc12_xb_bridge(), src/protocol/p25/p25_crc.c:78-147: x^12 + x^11 + x^7 + x^4 + x^2 + x + 1, remainder inverted),
// and for a SACCH on a control channel (LCCH) CRC-CCITT16 over the first 164 bits (crc16_lb_bridge(), :17-75).  One burst per thread.
__global__ void
k_p2_mac_crc(int kind, const uint8_t* __restrict__ payload_bits, int n, uint8_t* __restrict__ crc12_ok, uint8_t* __restrict__ crc16_ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const int n_pl = kind == 0 ? 156 : 180, len = n_pl - 12;
    const uint8_t* b = payload_bits + (size_t)i * n_pl;
    uint32_t reg = 0; // the long division: 13-bit register, the polynomial's bits 1 1000 1001 0111
    for (int k = 0; k < len + 12; k++) {
        reg = (reg << 1) | (k < len ? (uint32_t)(b[k] & 1) : 0u);
        if (reg & 0x1000u) {
            reg ^= 0x1897u;
        }
    }
    uint32_t got = 0;
    for (int k = 0; k < 12; k++) {
        got = (got << 1) | (uint32_t)(b[len + k] & 1);
    }
    crc12_ok[i] = ((reg ^ 0xFFFu) & 0xFFFu) == got ? 1 : 0;
    if (crc16_ok) {
        uint32_t c = 0, g16 = 0;
        if (kind == 1) {
            for (int k = 0; k < 164; k++) {
                c = ((((c >> 15) & 1u) ^ (uint32_t)(b[k] & 1)) ? ((c << 1) ^ 0x1021u) : (c << 1)) & 0xFFFFu;
            }
            c ^= 0xFFFFu;
            for (int k = 0; k < 16; k++) {
                g16 = (g16 << 1) | (uint32_t)(b[164 + k] & 1);
            }
        }
        crc16_ok[i] = (kind == 1 && c == g16) ? 1 : 0;
    }
}

extern "C" hipError_t
ddn_dev_p25p2_mac_crc(int kind, const uint8_t* payload_bits, int n, uint8_t* crc12_ok, uint8_t* crc16_ok, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_mac_crc, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, kind, payload_bits, n, crc12_ok, crc16_ok);
    return hipGetLastError();
}

// ---- P25 Phase 2 ESS: ranked erasure list and the rule for the plain decode ---------------------------------------------------------
// p25p2_ess_soft_erasures_ranked() (p25p2_soft.c:331-383) and p25p2_ess_decode_with_soft_erasures() (p25p2_frame.c:1061-1091): the
// plain decode stands when it located fewer than 15 symbols; otherwise the section is retried from its received bits with the first
// 1, 2, ... erasures of the list (k_rs28: ess_rule on the plain attempt, then the retries side by side - rs28_retries()).
__global__ void
k_p2_ess_prepare(const uint8_t* __restrict__ payload_bits, const int16_t* __restrict__ payload_llr, const int16_t* __restrict__ parity_llr,
                 int n, int threshold, uint8_t* __restrict__ work, int8_t* __restrict__ erasures28, uint8_t* __restrict__ n_total,
                 uint8_t* __restrict__ used_dynamic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    for (int k = 0; k < 96; k++) {
        work[(size_t)i * 96 + k] = payload_bits[(size_t)i * 96 + k] & 1;
    }
    __shared__ uint16_t keys[44][64]; // a column per thread (64 per block)
#define key(j) keys[j][threadIdx.x]
    int below = 0;
    for (int hb = 0; hb < 44; hb++) {
        const int16_t* l = hb < 16 ? payload_llr + (size_t)i * 96 + 6 * hb : parity_llr + (size_t)i * 168 + 6 * (hb - 16);
        int r = 255;
        for (int b = 0; b < 6; b++) {
            int v = l[b];
            v = v < 0 ? -v : v;
            v = v > 255 ? 255 : v;
            r = v < r ? v : r;
        }
        key(hb) = (uint16_t)((r << 8) | hb);
        below += r < threshold ? 1 : 0;
    }
    int cnt = below < 14 ? 14 : below;
    cnt = cnt > 28 ? 28 : cnt;
    int8_t* er = erasures28 + (size_t)i * 28;
    for (int k = 0; k < cnt; k++) {
        int best = k;
        for (int j = k + 1; j < 44; j++) {
            best = key(j) < key(best) ? j : best;
        }
        const uint16_t t = key(k);
        key(k) = key(best);
        key(best) = t;
        er[k] = (int8_t)(key(k) & 0xFF);
    }
    for (int k = cnt; k < 28; k++) {
        er[k] = 0;
    }
    n_total[i] = (uint8_t)cnt;
    used_dynamic[i] = 0;
#undef key
}

extern "C" hipError_t
ddn_dev_p25p2_ess(const uint8_t* payload_bits, const int16_t* payload_llr, const uint8_t* parity_bits, const int16_t* parity_llr, int n,
                  int threshold, uint8_t* work, int8_t* erasures28, uint8_t* n_total, int32_t* status, uint8_t* used_dynamic, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const dim3 grid((unsigned)((n + 63) / 64)), blk(64);
    hipLaunchKernelGGL(k_p2_ess_prepare, grid, blk, 0, st, payload_bits, payload_llr, parity_llr, n, threshold, work, erasures28, n_total,
                       used_dynamic);
    // the plain decode for every section (not taken when it located 15 symbols or more), then one launch for the retries
    hipLaunchKernelGGL(k_rs28, grid, blk, 0, st, 0, work, parity_bits, erasures28, n_total, n, status, 0, 0, used_dynamic, -1, 1, 0,
                       (int32_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr);
    return rs28_retries(0, work, parity_bits, erasures28, n_total, n, status, 0, used_dynamic, 28, st);
}

// ---- P25 Phase 2 frame scrambler ----------------------------------------------------------------------------------------------------
// p25p2_generate_scramble_bits() (src/protocol/p25/phase2/p25p2_scramble.c:12-26): the 44-bit Fibonacci LFSR x^44 + x^34 + x^20 + x^15 +
// x^9 + x^4 + 1 seeded with WACN | SYSID | NAC; process_Frame_Scramble() (p25p2_frame.c:370-392): a superframe's 4320 sequence bits,
// the received bit i against sequence bit i + 20 + 360 * offset (the sequence doubled up = taken mod 4320), the soft metric's sign
// flipped where the sequence bit is set.  The sequence is a 4320-step recurrence: one thread per system, 64 steps per word.
__global__ void
k_p2_scramble_bits(const uint64_t* __restrict__ seed44, int n, int bit_count, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint64_t s = seed44[i];
    uint8_t* o = out + (size_t)i * bit_count;
    for (int k = 0; k < bit_count; k++) {
        o[k] = (uint8_t)((s >> 43) & 1u);
        const uint64_t b = ((s >> 33) ^ (s >> 19) ^ (s >> 14) ^ (s >> 8) ^ (s >> 3) ^ (s >> 43)) & 1u;
        s = (s << 1) | b;
    }
}

__global__ void
k_p2_descramble(const uint8_t* __restrict__ bits, const int16_t* __restrict__ llr, const uint8_t* __restrict__ lbits4320,
                const int32_t* __restrict__ offset, const int32_t* __restrict__ seq_of, int n_bits, int n_llr, uint8_t* __restrict__ xbits,
                int16_t* __restrict__ xllr) {
    const int item = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bits) {
        return;
    }
    const uint8_t* lb = lbits4320 + (size_t)(seq_of ? seq_of[item] : item) * 4320;
    const int q = (i + 20 + 360 * offset[item]) % 4320;
    const int l = lb[q < 0 ? q + 4320 : q];
    xbits[(size_t)item * n_bits + i] = (uint8_t)((bits[(size_t)item * n_bits + i] ^ l) & 1);
    if (i < n_llr) {
        const int16_t v = llr[(size_t)item * n_llr + i];
        xllr[(size_t)item * n_llr + i] = l ? (int16_t)-v : v;
    }
}

extern "C" hipError_t
ddn_dev_p25p2_scramble_bits(const uint64_t* seed44, int n, int bit_count, uint8_t* out, hipStream_t st) {
    if (n <= 0 || bit_count <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_scramble_bits, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, seed44, n, bit_count, out);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_p25p2_descramble(const uint8_t* bits, const int16_t* llr, const uint8_t* lbits4320, const int32_t* offset, const int32_t* seq_of,
                         int n, int n_bits, int n_llr, uint8_t* xbits, int16_t* xllr, hipStream_t st) {
    if (n <= 0 || n_bits <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_descramble, dim3((unsigned)((n_bits + 255) / 256), (unsigned)n), dim3(256), 0, st, bits, llr, lbits4320, offset,
                       seq_of, n_bits, n_llr, xbits, xllr);
    return hipGetLastError();
}

// ---- P25 Phase 2 timeslot fields: DUID and I-ISCH ----------------------------------------------------------------------------------
// p25p2_duid_collect_and_decode() (p25p2_frame.c:1462-1478): the eight DUID bits at offsets 0, 1, 74, 75, 244, 245, 318, 319 of the
// timeslot -> p25p2_duid_lookup_soft() (:208-248): the (8,4) code's table (a canonical word or a word one bit from exactly one of
// them; 0x80 is withheld - "triggers false 4V on bad signal"); a word the table rejects is given to the canonical words one or two
// bits away whose differing bits all have reliability below the threshold, the cheapest (sum of those reliabilities) if it is
// alone; 0x80 only goes to 0, and only when its first bit alone is weak.  p25p2_process_isch() (:708-745): bits 320..359 and their
// reliabilities -> isch_lookup_soft() (k_isch_lookup).  One timeslot per thread; the I-ISCH word and its reliability row are laid out
// for the lookup kernel.
__device__ __forceinline__ int
p2_duid_hard(int r) {
    if (r == 0x80) {
        return -1;
    }
    int found = -1, n = 0;
    for (int d = 0; d < 16; d++) {
        // canonical word d: data nibble d, then the parity nibble of the (8,4,4) code
        const int canon[16] = {0x00, 0x17, 0x2E, 0x39, 0x4B, 0x5C, 0x65, 0x72, 0x8D, 0x9A, 0xA3, 0xB4, 0xC6, 0xD1, 0xE8, 0xFF};
        if (__popc((unsigned)(r ^ canon[d])) <= 1) {
            found = d;
            n++;
        }
    }
    return n == 1 ? found : -1;
}

__global__ void
k_p2_burst_fields(const uint8_t* __restrict__ bits360, const int16_t* __restrict__ llr360, int n, int threshold,
                  int32_t* __restrict__ duid, uint64_t* __restrict__ isch_word, uint8_t* __restrict__ isch_rel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* b = bits360 + (size_t)i * 360;
    const int16_t* l = llr360 + (size_t)i * 360;
    auto rel_of = [&](int k) {
        int v = l[k];
        v = v < 0 ? -v : v;
        return v > 255 ? 255 : v;
    };
    const int off[8] = {0, 1, 74, 75, 244, 245, 318, 319};
    const int canon[16] = {0x00, 0x17, 0x2E, 0x39, 0x4B, 0x5C, 0x65, 0x72, 0x8D, 0x9A, 0xA3, 0xB4, 0xC6, 0xD1, 0xE8, 0xFF};
    int r = 0, rel[8];
    for (int k = 0; k < 8; k++) {
        r = (r << 1) | (b[off[k]] & 1);
        rel[k] = rel_of(off[k]);
    }
    const int hard = p2_duid_hard(r);
    int out = hard;
    const bool exact = hard >= 0 && r == canon[hard];
    bool allowed = true;
    if (r == 0x80) { // p25p2_duid_080_soft_allowed()
        allowed = rel[0] < threshold;
        for (int k = 1; k < 8; k++) {
            allowed = allowed && rel[k] >= threshold;
        }
    }
    if (!exact && allowed) {
        int best = hard, best_cost = 999999;
        bool tied = false;
        for (int d = 0; d < 16; d++) {
            const int dist = __popc((unsigned)(r ^ canon[d]));
            if (dist < 1 || dist > 2 || (r == 0x80 && d != 0)) {
                continue;
            }
            int cost = 0;
            for (int k = 0; k < 8; k++) {
                if (((r ^ canon[d]) >> (7 - k)) & 1) {
                    cost = (rel[k] >= threshold || cost >= 999999) ? 999999 : cost + rel[k];
                }
            }
            if (cost >= 999999) {
                continue;
            }
            if (cost < best_cost) {
                best_cost = cost;
                best = d;
                tied = false;
            } else if (cost == best_cost && d != best) {
                tied = true;
            }
        }
        out = tied ? hard : best;
    }
    duid[i] = out;
    uint64_t w = 0;
    for (int k = 0; k < 40; k++) {
        w = (w << 1) | (uint64_t)(b[320 + k] & 1);
        isch_rel[(size_t)i * 40 + k] = (uint8_t)rel_of(320 + k);
    }
    isch_word[i] = w;
}

extern "C" hipError_t
ddn_dev_p25p2_burst_fields(const uint8_t* bits360, const int16_t* llr360, int n, int threshold, int32_t* duid, uint64_t* isch_word,
                           uint8_t* isch_rel, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_burst_fields, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, bits360, llr360, n, threshold, duid, isch_word,
                       isch_rel);
    return hipGetLastError();
}

// the Phase 2 FACCH (kind 0) / SACCH (kind 1) burst stage: gather + ranked list, then the decode with the fixed erasures and the
// retries with one more erasure each, side by side (rs28_retries(): only the sections that failed do any work)
extern "C" hipError_t
ddn_dev_p25p2_xcch(int kind, const uint8_t* bits360, const int16_t* llr360, int n, int threshold, uint8_t* payload_bits,
                   uint8_t* parity_bits, int8_t* erasures28, uint8_t* n_total, int32_t* status, uint8_t* used_dynamic, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_xcch_gather, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, kind, bits360, llr360, n, threshold,
                       payload_bits, parity_bits, erasures28, n_total, used_dynamic);
    const int n_fixed = kind == 0 ? 18 : 11, max_add = kind == 0 ? 10 : 16;
    // the decode with the fixed erasures for every burst, then the failed bursts' retries (most bursts of real traffic never get
    // there: those threads leave at once)
    hipLaunchKernelGGL(k_rs28, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, kind + 1, payload_bits, parity_bits, erasures28, n_total, n,
                       status, 0, n_fixed, used_dynamic, -1, 0, 0, (int32_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr);
    return rs28_retries(kind + 1, payload_bits, parity_bits, erasures28, n_total, n, status, n_fixed, used_dynamic, max_add, st);
}

// ---- P25 Phase 2 I-ISCH lookup == isch_lookup / isch_lookup_soft (src/fec/ez.cpp:325-384) ---------------------------------
// One received 40-bit word per lane against the 128 codewords of the (40,9,16) code plus the S-ISCH word; the table index is
// wave-uniform, so the codewords arrive through scalar loads from constant memory and the lane work is xor + popcount.  Hard:
// exact match, else the nearest entry within 7 bits - with minimum distance 16 the only tie is a codeword against the S-ISCH
// word 14 bits from it, settled by the measured walk order of the reference's map (DDN_ISCH_S_FIRST_INIT).  Soft (a 40-byte
// reliability row per word): the entry within 7 bits with the least (sum of reliabilities of differing bits, differing bits,
// answer).  -2 = S-ISCH / nothing within reach.  The word is NOT masked to 40 bits: the reference is not (dsd_popcount64 of the
// whole xor, src/fec/ez.cpp:335, while isch_weighted_mismatch_cost :345-352 walks the 40 field bits only), so a stray high bit
// adds to the distance, never to the cost, and never indexes the reliability row (`b < 40` below; tests: words_high, bit 47 set).
__constant__ uint64_t c_isch[128] = DDN_ISCH_TABLE_INIT;
__constant__ uint8_t c_isch_s_first[128] = DDN_ISCH_S_FIRST_INIT;

__global__ __launch_bounds__(256) void
k_isch_lookup(const uint64_t* __restrict__ words, const uint8_t* __restrict__ reliab40, int n, int32_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint64_t w = words[i];
    const uint8_t* rel = reliab40 ? reliab40 + (size_t)i * 40 : nullptr;
    int exact = -1, hb = -2, hd = 40;            // hard: best index, its distance
    int sb = -2, sc = 0x7fffffff, sp = 40;       // soft: best answer, cost, distance
    for (int k = -1; k < 128; k++) {
        const uint64_t cw = k < 0 ? (uint64_t)DDN_ISCH_S_WORD : c_isch[k];
        uint64_t diff = w ^ cw;
        const int d = __popcll(diff);
        if (d == 0) {
            exact = k < 0 ? -2 : k;
        }
        if (d > 7) {
            continue;
        }
        if (rel) {
            int cost = 0;
            while (diff) {
                const int b = 63 - __clzll((long long)diff);
                diff &= ~(1ULL << b);
                if (b < 40) { // isch_weighted_mismatch_cost() walks the 40 field bits only; stray high bits count in d alone
                    cost += rel[39 - b];
                }
            }
            const int val = k < 0 ? -2 : k;
            if (cost < sc || (cost == sc && d < sp) || (cost == sc && d == sp && val < sb)) {
                sb = val;
                sc = cost;
                sp = d;
            }
        } else if (k < 0) {
            hd = d;                              // S-ISCH first; a codeword replaces it when strictly nearer, or on the
        } else if (d < hd || (d == hd && hb == -2 && !c_isch_s_first[k])) { // 7 / 7 tie when the map reaches it first
            hb = k;
            hd = d;
        }
    }
    out[i] = exact != -1 ? exact : (rel ? sb : hb);
}

hipError_t
ddn_dev_isch_lookup(const uint64_t* words, const uint8_t* reliab40, int n, int32_t* out, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_isch_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, reliab40, n, out);
    return hipGetLastError();
}
