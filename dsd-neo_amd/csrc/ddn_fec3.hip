// ddn_fec3.hip - DMR / NXDN block codes downstream of the receive loop: the Hamming family,
// Golay(20,8), Golay(24,12), QR(16,7,6), BPTC(196,96) and Reed-Solomon(12,9), batched.
//
// reference: src/fec/fec.c:133-838 (syndrome by parity-check matrix, correction positions by syndrome table built in the
// *_init() loops), src/fec/bptc.c:27-160 (de-interleave: Output[13 i mod 196] = Input[i], 13 x 15 matrix, rows Hamming(15,11), columns
// Hamming(13,9), two passes), src/fec/rs-12-9.c (GF(256) x^8+x^4+x^3+x^2+1, syndromes at alpha^1..3, Massey, Chien over
// r = 1..255, Forney).  Callers: src/protocol/dmr/dmr_dburst.c:500-530 and friends, src/protocol/nxdn/*.
//
// All integer work, one item per lane: a code word is packed into a 32-bit register (bit j = rxBits[j]), the syndrome is
// r parities popc(word & row mask), the correction a table lookup.  The tables are built once per device by the host with
// the reference's own assignment order (its init loops overwrite single slots of an entry, so for the distance-6 codes
// the survivor of colliding patterns depends on that order - reproduced, including the quirks noted at each decoder).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "ddn_device.h"
#include "ddn_fec3.h"
#include "ddn_tables_fec3.h"
#include "ddn_tables_bptc.h"

namespace {

using Tables = DdnFec3Tables;

int
syn_of(uint32_t w, const uint32_t* H, int r) {
    int s = 0;
    for (int i = 0; i < r; i++) {
        s |= (__builtin_popcount(w & H[i]) & 1) << (r - 1 - i);
    }
    return s;
}

void
build_hamming(uint8_t* tab, int size, const uint32_t* H, int n, int r) {
    // correctable positions = the columns of H (explicit assignments of Hamming_*_init(), fec.c:133-143,183-198,...)
    memset(tab, 0xFF, (size_t)size);
    for (int p = 0; p < n; p++) {
        tab[syn_of(1u << p, H, r)] = (uint8_t)p;
    }
}

// Golay_20_8_init / Golay_24_12_init (fec.c:437-512, 566-640): kd message bits, 12 parity bits, slot-wise assignments in
// the reference's loop order
void
build_golay(uint8_t (*tab)[3], const uint32_t* H, int kd) {
    memset(tab, 0xFF, 4096 * 3);
    auto S = [&](uint32_t w) { return syn_of(w, H, 12); };
    auto P = [&](int ip) { return 1 << (11 - ip); };
    for (int i1 = 0; i1 < kd; i1++) {
        for (int i2 = i1 + 1; i2 < kd; i2++) {
            for (int i3 = i2 + 1; i3 < kd; i3++) {
                const int s = S((1u << i1) | (1u << i2) | (1u << i3));
                tab[s][0] = (uint8_t)i1, tab[s][1] = (uint8_t)i2, tab[s][2] = (uint8_t)i3;
            }
            const int s = S((1u << i1) | (1u << i2));
            tab[s][0] = (uint8_t)i1, tab[s][1] = (uint8_t)i2;
            for (int ip = 0; ip < 12; ip++) {
                const int sp = s ^ P(ip);
                tab[sp][0] = (uint8_t)i1, tab[sp][1] = (uint8_t)i2, tab[sp][2] = (uint8_t)(kd + ip);
            }
        }
        const int s = S(1u << i1);
        tab[s][0] = (uint8_t)i1;
        for (int ip1 = 0; ip1 < 12; ip1++) {
            const int s1 = s ^ P(ip1);
            tab[s1][0] = (uint8_t)i1, tab[s1][1] = (uint8_t)(kd + ip1);
            for (int ip2 = ip1 + 1; ip2 < 12; ip2++) {
                const int s2 = s1 ^ P(ip2);
                tab[s2][0] = (uint8_t)i1, tab[s2][1] = (uint8_t)(kd + ip1), tab[s2][2] = (uint8_t)(kd + ip2);
            }
        }
    }
    for (int ip1 = 0; ip1 < 12; ip1++) {
        const int s1 = P(ip1);
        tab[s1][0] = (uint8_t)(kd + ip1);
        for (int ip2 = ip1 + 1; ip2 < 12; ip2++) {
            const int s2 = s1 ^ P(ip2);
            tab[s2][0] = (uint8_t)(kd + ip1), tab[s2][1] = (uint8_t)(kd + ip2);
            for (int ip3 = ip2 + 1; ip3 < 12; ip3++) {
                const int s3 = s2 ^ P(ip3);
                tab[s3][0] = (uint8_t)(kd + ip1), tab[s3][1] = (uint8_t)(kd + ip2), tab[s3][2] = (uint8_t)(kd + ip3);
            }
        }
    }
}

// QR_16_7_6_init (fec.c:744-780): 7 message bits, 9 parity bits, up to two positions
void
build_qr(uint8_t (*tab)[2], const uint32_t* H) {
    memset(tab, 0xFF, 512 * 2);
    auto S = [&](uint32_t w) { return syn_of(w, H, 9); };
    auto P = [&](int ip) { return 1 << (8 - ip); };
    for (int i1 = 0; i1 < 7; i1++) {
        for (int i2 = i1 + 1; i2 < 7; i2++) {
            const int s = S((1u << i1) | (1u << i2));
            tab[s][0] = (uint8_t)i1, tab[s][1] = (uint8_t)i2;
        }
        const int s = S(1u << i1);
        tab[s][0] = (uint8_t)i1;
        for (int ip = 0; ip < 9; ip++) {
            const int sp = s ^ P(ip);
            tab[sp][0] = (uint8_t)i1, tab[sp][1] = (uint8_t)(7 + ip);
        }
    }
    for (int ip1 = 0; ip1 < 9; ip1++) {
        const int s1 = P(ip1);
        tab[s1][0] = (uint8_t)(7 + ip1);
        for (int ip2 = ip1 + 1; ip2 < 9; ip2++) {
            const int s2 = s1 ^ P(ip2);
            tab[s2][0] = (uint8_t)(7 + ip1), tab[s2][1] = (uint8_t)(7 + ip2);
        }
    }
}

hipError_t
device_tables(const Tables** out, hipStream_t st) {
    static std::mutex mu;
    static Tables* dev_tabs[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        return e;
    }
    if (dev < 0 || dev >= 64) {
        return hipErrorInvalidDevice;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!dev_tabs[dev]) {
        std::vector<Tables> h(1);
        build_hamming(h[0].h74, 8, ddn_hamming_7_4_H, 7, 3);
        build_hamming(h[0].h128, 16, ddn_hamming_12_8_H, 12, 4);
        build_hamming(h[0].h139, 16, ddn_hamming_13_9_H, 13, 4);
        build_hamming(h[0].h1511, 16, ddn_hamming_15_11_H, 15, 4);
        build_hamming(h[0].h16114, 32, ddn_hamming_16_11_4_H, 16, 5);
        build_golay(h[0].g208, ddn_golay_20_8_H, 8);
        build_golay(h[0].g2412, ddn_golay_24_12_H, 12);
        build_qr(h[0].qr, ddn_qr_16_7_6_H);
        Tables* d = nullptr;
        e = hipMalloc(&d, sizeof(Tables));
        if (e != hipSuccess) {
            return e;
        }
        e = hipMemcpyAsync(d, h.data(), sizeof(Tables), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) {
            (void)hipFree(d);
            return e;
        }
        dev_tabs[dev] = d;
    }
    *out = dev_tabs[dev];
    return hipSuccess;
}

template <int R>
__device__ __forceinline__ int
syndrome(uint32_t w, const uint32_t (&H)[R]) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        s |= (__popc(w & H[i]) & 1) << (R - 1 - i);
    }
    return s;
}

__device__ __forceinline__ uint32_t
pack(const uint8_t* b, int n) {
    uint32_t w = 0;
    for (int j = 0; j < n; j++) {
        w |= (uint32_t)(b[j] & 1u) << j;
    }
    return w;
}

// ---- Hamming family -------------------------------------------------------------------------------------------------
// One item = one call of the reference with nb code words.  Reproduced quirks: the corrected position is applied to
// rxBits[pos] WITHOUT the code word's offset (fec.c:225, 283, 342, ...: rxBits[m_corr[s]] ^= 1), i.e. always inside the
// first code word of the call; (12,8) keeps going after an uncorrectable word, the others stop at it (no further copies).
template <int N, int K, int R, bool BREAKS>
__device__ __forceinline__ bool
hamming_item(uint8_t* rx, uint8_t* dec, int nb, const uint32_t (&H)[R], const uint8_t* corr) {
    bool ok = true;
    for (int ic = 0; ic < nb; ic++) {
        const int s = syndrome<R>(pack(rx + N * ic, N), H);
        if (s > 0) {
            const uint8_t p = corr[s];
            if (p == 0xFF) {
                ok = false;
                if (BREAKS) {
                    break;
                }
            } else {
                rx[p] ^= 1;
            }
        }
        if (dec) {
            for (int j = 0; j < K; j++) {
                dec[K * ic + j] = rx[N * ic + j];
            }
        }
    }
    return ok;
}

__global__ __launch_bounds__(256) void
k_block_code(int code, uint8_t* __restrict__ bits, size_t n_items, int nb, uint8_t* __restrict__ decoded,
             uint8_t* __restrict__ okv, const Tables* __restrict__ T) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) {
        return;
    }
    bool ok = true;
    switch (code) {
        case DDN_CODE_HAMMING_7_4: {
            uint8_t* rx = bits + i * 7;
            const int s = syndrome<3>(pack(rx, 7), ddn_hamming_7_4_H);
            if (s > 0) {
                const uint8_t p = T->h74[s];
                if (p == 0xFF) {
                    ok = false;
                } else {
                    rx[p] ^= 1;
                }
            }
            break;
        }
        case DDN_CODE_HAMMING_12_8:
            ok = hamming_item<12, 8, 4, false>(bits + i * 12 * nb, decoded ? decoded + i * 8 * nb : nullptr, nb, ddn_hamming_12_8_H, T->h128);
            break;
        case DDN_CODE_HAMMING_13_9:
            ok = hamming_item<13, 9, 4, true>(bits + i * 13 * nb, decoded ? decoded + i * 9 * nb : nullptr, nb, ddn_hamming_13_9_H, T->h139);
            break;
        case DDN_CODE_HAMMING_15_11:
            ok = hamming_item<15, 11, 4, true>(bits + i * 15 * nb, decoded ? decoded + i * 11 * nb : nullptr, nb, ddn_hamming_15_11_H, T->h1511);
            break;
        case DDN_CODE_HAMMING_16_11_4:
            ok = hamming_item<16, 11, 5, true>(bits + i * 16 * nb, decoded ? decoded + i * 11 * nb : nullptr, nb, ddn_hamming_16_11_4_H, T->h16114);
            break;
        case DDN_CODE_GOLAY_20_8: { // fec.c:514-561: flips applied even when the tally then exceeds two
            uint8_t* rx = bits + i * 20;
            const int s = syndrome<12>(pack(rx, 20), ddn_golay_20_8_H);
            if (s > 0) {
                int k = 0;
                for (; k < 3; k++) {
                    const uint8_t p = T->g208[s][k];
                    if (p == 0xFF) {
                        break;
                    }
                    rx[p] ^= 1;
                }
                ok = !(k == 0 || k > 2);
            }
            break;
        }
        case DDN_CODE_GOLAY_24_12: { // fec.c:656-690
            uint8_t* rx = bits + i * 24;
            const int s = syndrome<12>(pack(rx, 24), ddn_golay_24_12_H);
            if (s > 0) {
                int k = 0;
                for (; k < 3; k++) {
                    const uint8_t p = T->g2412[s][k];
                    if (p == 0xFF) {
                        break;
                    }
                    rx[p] ^= 1;
                }
                ok = k != 0;
            }
            break;
        }
        default: { // DDN_CODE_QR_16_7_6, fec.c:782-822
            uint8_t* rx = bits + i * 16;
            const int s = syndrome<9>(pack(rx, 16), ddn_qr_16_7_6_H);
            if (s > 0) {
                int k = 0;
                for (; k < 2; k++) {
                    const uint8_t p = T->qr[s][k];
                    if (p == 0xFF) {
                        break;
                    }
                    rx[p] ^= 1;
                }
                ok = k != 0;
            }
            break;
        }
    }
    if (okv) {
        okv[i] = ok ? 1 : 0;
    }
}

// ---- BPTC(196,96) ---------------------------------------------------------------------------------------------------
// rows as 15-bit masks (bit j = column j), columns handled through the same masks.  A failed Hamming(13,9) column leaves
// the reference's col_corrected[] untouched, so the column takes the PREVIOUS column's corrected bits (bptc.c:95-111; for
// column 0 the reference reads an uninitialised array - zeros here); Hamming(15,11) is perfect and never fails.
__global__ __launch_bounds__(128) void
k_bptc_196x96(const uint8_t* __restrict__ in, int deinterleave, size_t n, uint8_t* __restrict__ out96,
              uint8_t* __restrict__ r3, uint32_t* __restrict__ errs, const Tables* __restrict__ T) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* src = in + i * 196;
    uint32_t row[13];
    for (int r = 0; r < 13; r++) {
        uint32_t w = 0;
        for (int j = 0; j < 15; j++) {
            const int k = 1 + r * 15 + j; // fill_matrix starts at input[1]
            // BPTCDeInterleaveDMRData: Output[(a * 13) % 196] = Input[a]  <=> Output[k] = Input[(k * 181) % 196] (13 * 181 = 1 mod 196)
            const int a = deinterleave ? (k * 181) % 196 : k;
            w |= (uint32_t)(src[a] & 1u) << j;
        }
        row[r] = w;
    }
    uint32_t bad = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint32_t e = 0;
        for (int r = 0; r < 9; r++) { // rows 0..8 only (bptc.c:76)
            const int s = syndrome<4>(row[r], ddn_hamming_15_11_H);
            if (s > 0) {
                const uint8_t p = T->h1511[s];
                if (p == 0xFF) {
                    e++;
                } else if (p < 11) { // only the eleven information bits are written back (bptc.c:84-86)
                    row[r] ^= 1u << p;
                }
            }
        }
        uint32_t prev = 0; // the nine corrected bits of the previous column
        for (int c = 0; c < 15; c++) {
            uint32_t col = 0;
            for (int r = 0; r < 13; r++) {
                col |= ((row[r] >> c) & 1u) << r;
            }
            const int s = syndrome<4>(col, ddn_hamming_13_9_H);
            bool okc = true;
            if (s > 0) {
                const uint8_t p = T->h139[s];
                if (p == 0xFF) {
                    okc = false;
                    e++;
                } else {
                    col ^= 1u << p;
                }
            }
            const uint32_t nine = okc ? (col & 0x1FFu) : prev;
            prev = nine;
            for (int r = 0; r < 9; r++) {
                row[r] = (row[r] & ~(1u << c)) | (((nine >> r) & 1u) << c);
            }
        }
        if (pass == 1) {
            bad = e;
        }
    }
    uint8_t* o = out96 + i * 96;
    int k = 0;
    for (int j = 3; j < 11; j++) {
        o[k++] = (uint8_t)((row[0] >> j) & 1u);
    }
    for (int r = 1; r < 9; r++) {
        for (int j = 0; j < 11; j++) {
            o[k++] = (uint8_t)((row[r] >> j) & 1u);
        }
    }
    if (r3) {
        r3[i * 3 + 0] = (uint8_t)((row[0] >> 2) & 1u);
        r3[i * 3 + 1] = (uint8_t)((row[0] >> 1) & 1u);
        r3[i * 3 + 2] = (uint8_t)(row[0] & 1u);
    }
    if (errs) {
        errs[i] = bad;
    }
}

// ---- Reed-Solomon (12,9) over GF(256) ----------------------------------------------------------------------------------
struct Gf {
    uint8_t exp[256], log[256];
};
__device__ __forceinline__ uint8_t
gmul(const Gf& g, uint8_t a, uint8_t b) {
    if (a == 0 || b == 0) {
        return 0;
    }
    return g.exp[(g.log[a] + g.log[b]) % 255];
}

__global__ __launch_bounds__(128) void
k_rs_12_9(uint8_t* __restrict__ cw, size_t n, uint8_t* __restrict__ result, uint8_t* __restrict__ found,
          uint8_t* __restrict__ syn_out) {
    __shared__ Gf g;
    if (threadIdx.x == 0) { // exp[255] = 1, log[0] = 0 as in the reference's tables (rs-12-9.c:33-62)
        uint32_t x = 1;
        for (int i = 0; i < 255; i++) {
            g.exp[i] = (uint8_t)x;
            g.log[x] = (uint8_t)i;
            x <<= 1;
            if (x & 0x100) {
                x ^= 0x11D;
            }
        }
        g.exp[255] = 1;
        g.log[0] = 0;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint8_t* c = cw + i * 12;
    uint8_t S[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 3; j++) { // rs_12_9_calc_syndrome: Horner at alpha^(j+1)
        uint8_t a = 0;
        for (int k = 0; k < 12; k++) {
            a = c[k] ^ gmul(g, g.exp[j + 1], a);
        }
        S[j] = a;
    }
    if (syn_out) {
        syn_out[i * 3] = S[0], syn_out[i * 3 + 1] = S[1], syn_out[i * 3 + 2] = S[2];
    }
    if (!(S[0] | S[1] | S[2])) { // rs_12_9_check_syndrome == 0: callers do not run the corrector
        result[i] = 0;
        if (found) {
            found[i] = 0;
        }
        return;
    }
    // rs_12_9_calculate: Massey over the three syndromes (polynomials of 6 coefficients)
    uint8_t loc[6] = {1, 0, 0, 0, 0, 0}, D[6] = {0, 1, 0, 0, 0, 0}, psi[6];
    int L = 0, kk = -1;
    for (int nn = 0; nn < 3; nn++) {
        uint8_t d = 0;
        for (int q = 0; q <= L; q++) {
            d ^= gmul(g, loc[q], S[nn - q]);
        }
        if (d != 0) {
            for (int q = 0; q < 6; q++) {
                psi[q] = loc[q] ^ gmul(g, d, D[q]);
            }
            if (L < nn - kk) {
                const int L2 = nn - kk;
                kk = nn - L;
                const uint8_t inv = g.exp[255 - g.log[d]];
                for (int q = 0; q < 6; q++) {
                    D[q] = gmul(g, loc[q], inv);
                }
                L = L2;
            }
            for (int q = 0; q < 6; q++) {
                loc[q] = psi[q];
            }
        }
        for (int q = 5; q > 0; q--) {
            D[q] = D[q - 1];
        }
        D[0] = 0;
    }
    // error evaluator = (locator * syndrome) mod z^3
    uint8_t ev[6] = {0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; a++) {
        for (int b = 0; a + b < 3; b++) {
            ev[a + b] ^= gmul(g, loc[a], S[b]);
        }
    }
    // Chien search r = 1..255 (locations 255 - r, in that order)
    uint8_t locs[8];
    int nroots = 0;
    for (int r = 1; r < 256; r++) {
        uint8_t sum = 0;
        for (int k = 0; k < 4; k++) {
            sum ^= gmul(g, g.exp[(k * r) % 255], loc[k]);
        }
        if (sum == 0) {
            if (nroots < 8) {
                locs[nroots] = (uint8_t)(255 - r);
            }
            nroots++;
        }
    }
    if (found) {
        found[i] = (uint8_t)nroots;
    }
    if (nroots == 0) {
        result[i] = 0; // RS_12_9_CORRECT_ERRORS_RESULT_NO_ERRORS_FOUND
        return;
    }
    if (nroots > 3) {
        result[i] = 2;
        return;
    }
    for (int r = 0; r < nroots; r++) {
        if (locs[r] >= 12) {
            result[i] = 2; // RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CANT_BE_CORRECTED
            return;
        }
    }
    for (int r = 0; r < nroots; r++) {
        const int loc_i = locs[r];
        uint8_t num = 0;
        for (int j = 0; j < 6; j++) {
            num ^= gmul(g, ev[j], g.exp[((255 - loc_i) * j) % 255]);
        }
        uint8_t den = 0;
        for (int j = 1; j < 6; j += 2) {
            den ^= gmul(g, loc[j], g.exp[((255 - loc_i) * (j - 1)) % 255]);
        }
        c[12 - loc_i - 1] ^= gmul(g, num, g.exp[255 - g.log[den]]);
    }
    result[i] = 1; // RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CORRECTED
}
} // namespace

extern "C" hipError_t
ddn_dev_block_code(int code, uint8_t* bits, size_t n_items, int nb, uint8_t* decoded, uint8_t* ok, hipStream_t st) {
    if (n_items == 0) {
        return hipSuccess;
    }
    const Tables* T = nullptr;
    hipError_t e = device_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_block_code, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, code, bits, n_items, nb,
                       decoded, ok, T);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_bptc_196x96(const uint8_t* in, int deinterleave, size_t n, uint8_t* out96, uint8_t* r3, uint32_t* errs,
                    hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    const Tables* T = nullptr;
    hipError_t e = device_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_bptc_196x96, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, in, deinterleave, n, out96, r3,
                       errs, T);
    return hipGetLastError();
}

// BPTC_128x77_Extract_Data (src/fec/bptc.c:167-258) and BPTC_16x2_Extract_Data (:278-336), one matrix per lane, rows as 16-bit
// words.  128x77: rows 0..6 through Hamming(16,11,4); a row that cannot be corrected takes the eleven bits the previous row
// decoded to (the reference's line buffer is written on success only; zeros for row 0, where the reference's is uninitialised),
// then the 77-bit read-out and the column parities against row 7.  16x2: the measured de-interleave, row 0 through the same
// code (left as received when it cannot be corrected - the reference reads an uninitialised buffer there), parity row compared
// bit by bit in the odd or even sense.
__constant__ uint8_t c_bptc_rc_perm[32] = DDN_BPTC_RC_PERM_INIT;

__global__ __launch_bounds__(128) void
k_bptc_128x77(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out77, uint32_t* __restrict__ errs,
              const Tables* __restrict__ T) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* src = in + i * 128;
    uint32_t row[8], line = 0, bad = 0;
    for (int r = 0; r < 8; r++) {
        uint32_t w = 0;
        for (int j = 0; j < 16; j++) {
            w |= (uint32_t)(src[r * 16 + j] & 1u) << j;
        }
        row[r] = w;
    }
    for (int r = 0; r < 7; r++) {
        uint32_t w = row[r];
        const int s = syndrome<5>(w, ddn_hamming_16_11_4_H);
        bool ok = true;
        if (s > 0) {
            const uint8_t p = T->h16114[s];
            if (p == 0xFF) {
                ok = false;
            } else {
                w ^= 1u << p;
            }
        }
        if (ok) {
            line = w & 0x7FFu;
        } else {
            bad++;
        }
        row[r] = (row[r] & ~0x7FFu) | line;
    }
    uint8_t* o = out77 + i * 77;
    int k = 0;
    for (int r = 0; r < 2; r++) {
        for (int j = 0; j < 11; j++) {
            o[k++] = (uint8_t)((row[r] >> j) & 1u);
        }
    }
    for (int r = 2; r < 7; r++) {
        for (int j = 0; j < 10; j++) {
            o[k++] = (uint8_t)((row[r] >> j) & 1u);
        }
    }
    for (int r = 2; r < 7; r++) {
        o[k++] = (uint8_t)((row[r] >> 10) & 1u);
    }
    const uint32_t par = row[0] ^ row[1] ^ row[2] ^ row[3] ^ row[4] ^ row[5] ^ row[6];
    bad += (uint32_t)__popc((par ^ row[7]) & 0xFFFFu);
    if (errs) {
        errs[i] = bad;
    }
}

__global__ __launch_bounds__(128) void
k_bptc_16x2(const uint8_t* __restrict__ in, size_t n, int parity_odd, uint8_t* __restrict__ out32, uint32_t* __restrict__ errs,
            const Tables* __restrict__ T) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t m = 0;
    for (int j = 0; j < 32; j++) {
        m |= (uint32_t)(in[i * 32 + j] & 1u) << c_bptc_rc_perm[j];
    }
    uint32_t bad = 0;
    const int s = syndrome<5>(m & 0xFFFFu, ddn_hamming_16_11_4_H);
    if (s > 0) {
        const uint8_t p = T->h16114[s];
        if (p == 0xFF) {
            bad = 1;
        } else if (p < 11) { // only the eleven data bits are copied back (bptc.c:311-313)
            m ^= 1u << p;
        }
    }
    for (int j = 0; j < 32; j++) {
        out32[i * 32 + j] = (uint8_t)((m >> j) & 1u);
    }
    const uint32_t diff = (m ^ (m >> 16)) & 0xFFFFu; // row-0 bit != parity-row bit
    bad += (uint32_t)__popc(parity_odd ? (~diff & 0xFFFFu) : diff);
    if (errs) {
        errs[i] = bad;
    }
}

extern "C" hipError_t
ddn_dev_bptc_128x77(const uint8_t* in, size_t n, uint8_t* out77, uint32_t* errs, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    const Tables* T = nullptr;
    hipError_t e = device_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_bptc_128x77, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, in, n, out77, errs, T);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_bptc_16x2(const uint8_t* in, size_t n, int parity_odd, uint8_t* out32, uint32_t* errs, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    const Tables* T = nullptr;
    hipError_t e = device_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_bptc_16x2, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, in, n, parity_odd, out32, errs, T);
    return hipGetLastError();
}

// trellis_decode() (src/core/util/dsd_misc.c:24-71; include/dsd-neo/fec/trellis.h:22): the hard-decision retry of the NXDN
// field decoders (nxdn_deperm.c:197-205).  Not a Viterbi search: for every output bit the 16 four-bit continuations of the
// current 5-bit register are re-encoded (rate 1/2, generators 0x19 / 0x17 as parity masks) and compared with the next 8 input
// bits; the first bit of the closest one is kept (ties: the lowest candidate index, strict <, candidate 0 taken first).
// One codeword per lane; the 16 x 8 comparisons are bit-parallel on one 8-bit word per candidate.
__global__ void
k_trellis_greedy(const uint8_t* __restrict__ src, int src_stride, size_t n, int result_len, uint8_t* __restrict__ out,
                 int out_stride, const uint8_t* __restrict__ wanted) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* s = src + i * (size_t)src_stride;
    uint8_t* o = out + i * (size_t)out_stride;
    if (wanted && !wanted[i]) { // optional: rows that are not wanted read zeros
        for (int p = 0; p < result_len; p++) {
            o[p] = 0;
        }
        return;
    }
    unsigned reg = 0;
    for (int p = 0; p < result_len; p++) {
        unsigned want = 0; // source[2p .. 2p+7], first bit in bit 7
        for (int j = 0; j < 8; j++) {
            want = (want << 1) | (s[2 * p + j] & 1u);
        }
        int min_d = 9999, min_bt = 0;
        for (int c = 0; c < 16; c++) {
            unsigned r = reg, enc = 0;
            for (int b = 0; b < 4; b++) {
                r = ((r << 1) | ((c >> (3 - b)) & 1u)) & 0x1Fu;
                enc = (enc << 2) | ((__popc(r & 0x19u) & 1u) << 1) | (__popc(r & 0x17u) & 1u);
            }
            const int d = __popc(enc ^ want);
            if (c == 0 || d < min_d) {
                min_d = d;
                min_bt = (c >> 3) & 1;
            }
        }
        o[p] = (uint8_t)min_bt;
        reg = ((reg << 1) | (unsigned)min_bt) & 0x1Fu;
    }
}

extern "C" hipError_t
ddn_dev_trellis_greedy_wanted(const uint8_t* src, int src_stride, size_t n, int result_len, uint8_t* out, int out_stride,
                              const uint8_t* wanted, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_trellis_greedy, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, src, src_stride, n, result_len, out,
                       out_stride, wanted);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_trellis_greedy(const uint8_t* src, int src_stride, size_t n, int result_len, uint8_t* out, int out_stride, hipStream_t st) {
    return ddn_dev_trellis_greedy_wanted(src, src_stride, n, result_len, out, out_stride, nullptr, st);
}

extern "C" hipError_t
ddn_dev_rs_12_9(uint8_t* cw, size_t n, uint8_t* result, uint8_t* found, uint8_t* syn_out, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_rs_12_9, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, cw, n, result, found, syn_out);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_fec3_tables(const DdnFec3Tables** out, hipStream_t st) {
    return device_tables(out, st);
}
