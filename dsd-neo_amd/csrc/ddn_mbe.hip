// ddn_mbe.hip - vocoder stage: IMBE 7200x4400 / AMBE 3600x2450 frame FEC decode, parameter unpack,
// spectral amplitude enhancement and synthesis to 160 float samples per 20 ms frame.
//
// reference: the calls src/core/vocoder/dsd_mbe.c:152-190,540-598 makes into mbelib-neo 2.x (mbe_decodeImbe7200x4400Frame,
// mbe_decodeAmbe3600x2450Frame, mbe_processImbe4400Dataf, mbe_processAmbe2450Dataf).  The library's source is not in
// the reference tree; the arithmetic below is the published algorithm of the mbelib lineage (mbelib 1.3 ecc.c,
// imbe7200x4400.c, ambe3600x2450.c, mbelib.c) - see include/ddn_mbe.h for what is pinned and what is not.
//
// Three kernels, split by what is actually sequential:
//   k_mbe_frame_decode  frames are independent: one lane per frame, 64 frames staged coalesced through LDS, Golay(23,12)
//                       by a 2048-entry syndrome -> pattern table in LDS (the code is perfect), Hamming(15,11) by its
//                       parity-check column, the PN sequence as four 23-bit and three 15-bit masks.  HBM-bound:
//                       184 B in + 108 B out per IMBE frame.
//   k_mbe_params        a talk path's frames depend on each other only through the parameter history (log-magnitude
//                       prediction, phase track, repeat counter): one wavefront per talk path walks its frames in
//                       order, lane = harmonic index, and leaves one self-contained synthesis record per frame
//                       (both sides' w0 / amplitudes / phases / voicing).  Sums run in harmonic order on every lane
//                       (LDS broadcast reads), so results do not depend on a reduction tree.
//   k_mbe_synth         all 160 x <= 56 oscillator terms of a frame are independent given its record: one wavefront
//                       per frame, lane = output sample (3 per lane), harmonics looped with wave-uniform branches on
//                       the voicing pair; record fields are wave-uniform loads.  Compute-bound (polynomial cosines),
//                       640 B of PCM per frame written coalesced.

#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_mbe.h"
#include "ddn_mbe_dev.h"
#include "ddn_mbe_math.h"

namespace {

// ---- frame FEC ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t
golay_syndrome(uint32_t w) {
#pragma unroll
    for (int i = 22; i >= 11; i--) {
        if ((w >> i) & 1u) {
            w ^= 0xC75u << (i - 11);
        }
    }
    return w & 0x7FFu;
}

__device__ __forceinline__ void
golay_table_lds(uint32_t* tab, int tid, int nthreads) {
    for (int idx = tid; idx < 23 * 23 * 23; idx += nthreads) {
        const int a = idx / 529, b = (idx / 23) % 23, c = idx % 23;
        if (a <= b && b <= c) {
            const uint32_t e = (1u << a) | (1u << b) | (1u << c);
            tab[golay_syndrome(e)] = e;
        }
    }
    if (tid == 0) {
        tab[0] = 0;
    }
}

// The syndrome -> pattern table once per device in global memory (8 KB): a decode block copies it into LDS instead of walking the
// 23^3 patterns itself (that walk was most of a block's work: 190 iterations per lane against the 64 frames' few hundred
// instructions).
__device__ uint32_t g_golay_tab[2048];

__global__ __launch_bounds__(256) void
k_golay_tab_init() {
    __shared__ uint32_t tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) {
        tab[i] = 0;
    }
    __syncthreads();
    golay_table_lds(tab, threadIdx.x, 256);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) {
        g_golay_tab[i] = tab[i];
    }
}

// corrects the 12 data bits (22..11) of a 23-bit word, parity bits stay as received (mbe_golay2312)
__device__ __forceinline__ uint32_t
golay_fix(uint32_t w, const uint32_t* tab, int& errs) {
    const uint32_t e = tab[golay_syndrome(w)] & 0x7FF800u;
    errs = __popc(e);
    return w ^ e;
}

__device__ __forceinline__ uint32_t
hamming_fix(uint32_t w, int& errs) {
    const uint32_t m0 = 0x7f08, m1 = 0x78e4, m2 = 0x66d2, m3 = 0x55b1;
    const uint32_t syn = ((__popc(w & m0) & 1) << 3) | ((__popc(w & m1) & 1) << 2) | ((__popc(w & m2) & 1) << 1)
                         | (__popc(w & m3) & 1);
    errs = 0;
    if (syn) {
        errs = 1;
#pragma unroll
        for (int p = 0; p < 15; p++) {
            const uint32_t col = (((m0 >> p) & 1u) << 3) | (((m1 >> p) & 1u) << 2) | (((m2 >> p) & 1u) << 1) | ((m3 >> p) & 1u);
            if (col == syn) {
                w ^= 1u << p;
            }
        }
    }
    return w;
}

#define DDN_MBE_FRAMES_PER_WG 32
template <int CODEC>
__global__ __launch_bounds__(64) void
k_mbe_frame_decode(const uint8_t* __restrict__ frames, const uint8_t* __restrict__ soft, size_t n,
                   uint8_t* __restrict__ bits_out, int32_t* __restrict__ result) {
    constexpr int FB = CODEC == DDN_MBE_IMBE_7200X4400 ? 184 : 96; // bytes per frame
    constexpr int OB = CODEC == DDN_MBE_IMBE_7200X4400 ? 88 : 49;
    constexpr int ROWLEN = CODEC == DDN_MBE_IMBE_7200X4400 ? 23 : 24;
    // 32 frames per workgroup, the Golay table read in place (8 KB: it stays in the L1): 8.7 KB of LDS per workgroup instead of 25.6,
    // which fits beside the front-end kernel of the next call (153 KB of a CU's 160) - in the chain object this kernel is on the path
    // the next receive loop waits for, and with 25.6 KB it waited for a front-end workgroup to end
    constexpr int FPW = DDN_MBE_FRAMES_PER_WG;
    __shared__ uint8_t stage[FPW * FB];
    __shared__ uint8_t outb[FPW * OB];
    const uint32_t* tab = g_golay_tab;
    const int lane = threadIdx.x;
    const size_t f0 = (size_t)blockIdx.x * FPW;
    const size_t nf = (n - f0) < (size_t)FPW ? (n - f0) : (size_t)FPW;
    const uint8_t* src = frames + f0 * FB;
    for (size_t i = lane; i < nf * FB; i += 64) {
        stage[i] = src[i];
    }
    __syncthreads();
    if ((size_t)lane < nf) {
        const uint8_t* f = stage + lane * FB;
        uint32_t row[8];
        bool bad = false;
        constexpr int NROW = CODEC == DDN_MBE_IMBE_7200X4400 ? 8 : 4;
#pragma unroll
        for (int r = 0; r < NROW; r++) {
            uint32_t w = 0;
            for (int j = 0; j < ROWLEN; j++) {
                const uint32_t v = f[r * ROWLEN + j];
                bad |= v > 1u;
                w |= (v & 1u) << j;
            }
            row[r] = w;
        }
        uint8_t* o = outb + lane * OB;
        int c0 = 0, c4 = 0, total = 0, e = 0;
        unsigned flags = MBE_PROCESS_FLAG_C0_VALID | (soft ? MBE_PROCESS_FLAG_SOFT_INPUT : 0u);
        if (CODEC == DDN_MBE_IMBE_7200X4400) {
            flags |= MBE_PROCESS_FLAG_C4_VALID;
            row[0] = golay_fix(row[0], tab, c0);
            total = c0;
            uint32_t pr = (16u * (row[0] >> 11)) & 0xFFFFu;
#pragma unroll
            for (int r = 1; r < 7; r++) {
                const int len = r < 4 ? 23 : 15;
                uint32_t mask = 0;
                for (int j = len - 1; j >= 0; j--) {
                    pr = (173u * pr + 13849u) & 0xFFFFu;
                    mask |= (pr >> 15) << j;
                }
                row[r] ^= mask;
            }
#pragma unroll
            for (int r = 1; r < 4; r++) {
                row[r] = golay_fix(row[r], tab, e);
                total += e;
            }
#pragma unroll
            for (int r = 4; r < 7; r++) {
                row[r] = hamming_fix(row[r] & 0x7FFFu, e);
                if (r == 4) {
                    c4 = e;
                }
                total += e;
            }
            int k = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                for (int j = 22; j > 10; j--) {
                    o[k++] = (uint8_t)((row[r] >> j) & 1u);
                }
            }
#pragma unroll
            for (int r = 4; r < 7; r++) {
                for (int j = 14; j >= 4; j--) {
                    o[k++] = (uint8_t)((row[r] >> j) & 1u);
                }
            }
            for (int j = 6; j >= 0; j--) {
                o[k++] = (uint8_t)((row[7] >> j) & 1u);
            }
        } else {
            // c0 = ambe_fr[0][1..23]: Golay word in bits 1..23 of row 0
            uint32_t w0 = golay_fix(row[0] >> 1, tab, c0);
            row[0] = (w0 << 1) | (row[0] & 1u);
            uint32_t pr = (16u * (w0 >> 11)) & 0xFFFFu;
            uint32_t mask = 0;
            for (int j = 22; j >= 0; j--) {
                pr = (173u * pr + 13849u) & 0xFFFFu;
                mask |= (pr >> 15) << j;
            }
            row[1] = golay_fix((row[1] ^ mask) & 0x7FFFFFu, tab, e);
            total = c0 + e;
            int k = 0;
            for (int j = 23; j > 11; j--) {
                o[k++] = (uint8_t)((row[0] >> j) & 1u);
            }
            for (int j = 22; j > 10; j--) {
                o[k++] = (uint8_t)((row[1] >> j) & 1u);
            }
            for (int j = 10; j >= 0; j--) {
                o[k++] = (uint8_t)((row[2] >> j) & 1u);
            }
            for (int j = 13; j >= 0; j--) {
                o[k++] = (uint8_t)((row[3] >> j) & 1u);
            }
        }
        int32_t* r5 = result + (f0 + lane) * 5;
        if (bad) { // a byte other than 0 / 1: MBE_STATUS_INVALID_BITS for this frame
            for (int k = 0; k < OB; k++) {
                o[k] = 0;
            }
            r5[0] = (int32_t)DDN_MBE_RESULT_INVALID;
            r5[1] = r5[2] = r5[3] = r5[4] = 0;
        } else {
            r5[0] = (int32_t)flags;
            r5[1] = c0;
            r5[2] = c4;
            r5[3] = total;
            r5[4] = total - c0;
        }
    }
    __syncthreads();
    uint8_t* dst = bits_out + f0 * OB;
    for (size_t i = lane; i < nf * OB; i += 64) {
        dst[i] = outb[i];
    }
}

// ---- parameter kernel -----------------------------------------------------------------------------------------------
struct LaneParms { // one mbe_parms, lane l holding entry [l] of every array (l = 0..56; lanes 57..63 carry dummies)
    float w0, gamma;
    int L, K, repeat;
    int V;
    float Ml, log2Ml, PHI, PSI;
};

__device__ __forceinline__ LaneParms
lane_init(int l) {
    LaneParms p;
    p.w0 = 0.09378f;
    p.gamma = 0.0f;
    p.L = 30;
    p.K = 10;
    p.repeat = 0;
    p.V = 0;
    p.Ml = 0.0f;
    p.log2Ml = 0.0f;
    p.PHI = 0.0f;
    p.PSI = MBE_PI_F / 2.0f;
    (void)l;
    return p;
}

__device__ __forceinline__ LaneParms
lane_load(const mbe_parms* s, int l) {
    LaneParms p;
    const int i = l <= 56 ? l : 56;
    p.w0 = s->w0;
    p.gamma = s->gamma;
    p.L = s->L;
    p.K = s->K;
    p.repeat = s->repeat;
    p.V = s->Vl[i];
    p.Ml = s->Ml[i];
    p.log2Ml = s->log2Ml[i];
    p.PHI = s->PHIl[i];
    p.PSI = s->PSIl[i];
    return p;
}

__device__ __forceinline__ void
lane_store(mbe_parms* s, const LaneParms& p, int l, int un) {
    if (l == 0) {
        s->w0 = p.w0;
        s->gamma = p.gamma;
        s->L = p.L;
        s->K = p.K;
        s->repeat = p.repeat;
        s->un = un;
    }
    if (l <= 56) {
        s->Vl[l] = p.V;
        s->Ml[l] = p.Ml;
        s->log2Ml[l] = p.log2Ml;
        s->PHIl[l] = p.PHI;
        s->PSIl[l] = p.PSI;
    }
}

__device__ __forceinline__ float
dequant(uint32_t b, int bits, float step) {
    if (bits <= 0) {
        return 0.0f;
    }
    return step * (((float)b - (float)(1u << (bits - 1))) + 0.5f);
}

// c(j) = sum_k a(k) C[k] cos(pi (k-1)(j-1/2) / J), k ascending; C is a 1-based LDS array read wave-divergently
__device__ __forceinline__ float
idct_term(const float* C, int J, int j) {
    float acc = 0.0f;
    for (int k = 1; k <= J; k++) {
        const float a = (k == 1) ? 1.0f : 2.0f;
        const float ang = (MBE_PI_F * (float)(k - 1) * ((float)j - 0.5f)) / (float)J;
        acc = acc + (a * C[k]) * mbe_cosf(ang);
    }
    return acc;
}

template <int CODEC>
__global__ __launch_bounds__(64) void
k_mbe_params(const uint8_t* __restrict__ bits, const int32_t* __restrict__ res_in, int n_frames,
             const ddn_mbe_tables* __restrict__ T, const float* __restrict__ half_log2, DdnMbeStream* __restrict__ streams,
             int tail_rule, DdnMbeFrameRec* __restrict__ recs, int32_t* __restrict__ res_out) {
    constexpr int NB = CODEC == DDN_MBE_IMBE_7200X4400 ? 88 : 49;
    __shared__ uint32_t fld[64];
    __shared__ float sG[12], sR[12], sC[64], sPl[64], sA[64], sB[64];
    __shared__ float sCb[5][64]; // AMBE: per-block coefficient rows (1-based)
    const int s = blockIdx.x;
    const int l = threadIdx.x;
    DdnMbeStream* st = streams + s;
    LaneParms cur = lane_load(&st->cur, l), prev = lane_load(&st->prev, l), enh = lane_load(&st->enh, l);
    uint32_t frame_no = st->frame_no;
    const uint32_t seed = st->seed;

    for (int f = 0; f < n_frames; f++) {
        const size_t fi = (size_t)s * (size_t)n_frames + (size_t)f;
        const uint8_t* d = bits + fi * NB;
        int32_t r[5] = {0, 0, 0, 0, 0};
        if (res_in) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                r[k] = res_in[fi * 5 + k];
            }
        }
        unsigned flags = (unsigned)r[0];
        const int errs2 = r[3];
        DdnMbeFrameRec* rec = recs + fi;
        // the frame's bits as two ballots (bit p of m0 = d[p], bit p - 64 of m1 = d[p])
        const uint32_t v0 = d[l], v1 = (l + 64 < NB) ? d[l + 64] : 0u;
        const unsigned long long m0 = __ballot(v0 & 1u), m1 = __ballot(v1 & 1u);
        const bool invalid = __any((v0 > 1u) || (v1 > 1u)) || (flags & DDN_MBE_RESULT_INVALID);
        auto bit = [&](int p) -> uint32_t { return (uint32_t)(((p < 64 ? m0 >> p : m1 >> (p - 64))) & 1ull); };
        bool skip = invalid; // invalid bits: silence, history untouched (MBE_STATUS_INVALID_BITS)
        if (!skip && tail_rule && CODEC == DDN_MBE_IMBE_7200X4400 && errs2 >= 10) {
            // dsd_mbe.c:447-463 mbe_p25p1_is_tail_erasure
            uint32_t prefix = 0;
            for (int p = 0; p < 8; p++) {
                prefix = (prefix << 1) | bit(p);
            }
            const int set = __popcll(m0) + __popcll(m1 & 0xFFFFFFull);
            if (prefix == 0xFCu && set <= 24) {
                skip = true;
                flags = 0;
                r[1] = r[2] = r[3] = r[4] = 0;
            }
        }
        if (skip) {
            if (l == 0) {
                rec->flags = DDN_MBE_REC_SILENCE;
                rec->maxl = 0;
                if (res_out) {
                    res_out[fi * 5 + 0] = (int32_t)(invalid ? DDN_MBE_RESULT_INVALID : flags);
                    res_out[fi * 5 + 1] = r[1];
                    res_out[fi * 5 + 2] = r[2];
                    res_out[fi * 5 + 3] = r[3];
                    res_out[fi * 5 + 4] = r[4];
                }
            }
            continue;
        }

        // ---- parameter decode into `cur` ----
        cur.repeat = prev.repeat;
        int bad = 0;
        float Tl = 0.0f;        // this lane's T_l
        float rho = 0.65f, extra = 0.0f;
        int L = 0;
        float unvc = 1.0f;
        if (CODEC == DDN_MBE_IMBE_7200X4400) {
            uint32_t b0 = 0;
            for (int p = 0; p < 6; p++) {
                b0 = (b0 << 1) | bit(p);
            }
            b0 = (b0 << 2) | (bit(85) << 1) | bit(86);
            const int t = (2 * (int)b0 + 81) / 8;
            L = (9254 * t) / 10000;
            if (b0 > 207 || L > 56 || L < 9) {
                bad = 1;
            } else {
                cur.w0 = (4.0f * MBE_PI_F) / ((float)b0 + 39.5f);
                cur.L = L;
                const int K = (L < 37) ? (L + 2) / 3 : 12;
                cur.K = K;
                const uint8_t* order = &T->imbe_bit_order[L - 9][0][0];
                const uint8_t* nbits = T->imbe_bits[L - 9];
                fld[l] = 0;
                __syncthreads();
                if (v0 & 1u) {
                    atomicOr(&fld[order[2 * l]], 1u << order[2 * l + 1]);
                }
                if (l + 64 < NB && (v1 & 1u)) {
                    atomicOr(&fld[order[2 * (l + 64)]], 1u << order[2 * (l + 64) + 1]);
                }
                __syncthreads();
                if (l >= 1 && l <= L) {
                    int band = (l + 2) / 3;
                    band = band > K ? K : band;
                    cur.V = (int)((fld[1] >> (K - band)) & 1u);
                }
                if (l == 1) {
                    sG[1] = T->imbe_gain_b2[fld[2] & 63u];
                } else if (l >= 2 && l <= 6) {
                    const int B = nbits[l + 1];
                    sG[l] = dequant(fld[l + 1], B, T->imbe_gain_step[B] * T->imbe_gain_sigma[l - 2]);
                }
                // higher-order coefficients: lane = field index 8 .. L+1 -> (block i, k)
                if (l >= 8 && l <= L + 1) {
                    int field = 8, kk = 2;
                    for (int i = 1; i <= 6; i++) {
                        const int J = (L + i - 1) / 6;
                        if (l < field + (J - 1)) {
                            kk = 2 + (l - field);
                            break;
                        }
                        field += J - 1;
                    }
                    const int B = nbits[l];
                    const int ks = kk > 10 ? 10 : kk;
                    sC[l] = dequant(fld[l], B, T->imbe_hoc_step[B] * T->imbe_hoc_sigma[ks - 2]);
                }
                __syncthreads();
                if (l >= 1 && l <= 6) {
                    sR[l] = idct_term(sG, 6, l);
                }
                __syncthreads();
                if (l >= 1 && l <= L) {
                    // block i and position j of harmonic l; block coefficients: C[1] = R_i, C[k] = field (8 + off + k - 2)
                    int first = 1, field = 8, bi = 1, J = 1;
                    for (int i = 1; i <= 6; i++) {
                        J = (L + i - 1) / 6;
                        bi = i;
                        if (l < first + J) {
                            break;
                        }
                        first += J;
                        field += J - 1;
                    }
                    const int j = l - first + 1;
                    float acc = 0.0f;
                    for (int k = 1; k <= J; k++) {
                        const float a = (k == 1) ? 1.0f : 2.0f;
                        const float c = (k == 1) ? sR[bi] : sC[field + k - 2];
                        const float ang = (MBE_PI_F * (float)(k - 1) * ((float)j - 0.5f)) / (float)J;
                        acc = acc + (a * c) * mbe_cosf(ang);
                    }
                    Tl = acc;
                }
                rho = (L <= 15) ? 0.4f : ((L <= 24) ? ((0.03f * (float)L) - 0.05f) : 0.7f);
            }
        } else {
            const uint32_t b0 = (bit(0) << 6) | (bit(1) << 5) | (bit(2) << 4) | (bit(3) << 3) | (bit(37) << 2) | (bit(38) << 1) | bit(39);
            if (b0 >= 120 && b0 <= 123) {
                flags |= MBE_PROCESS_FLAG_ERASURE;
                bad = 2;
            } else if (b0 == 126 || b0 == 127) {
                flags |= MBE_PROCESS_FLAG_TONE;
                bad = 3;
            } else {
                const bool silence = (b0 == 124 || b0 == 125);
                float f0;
                if (silence) {
                    flags |= MBE_PROCESS_FLAG_SILENCE;
                    cur.w0 = MBE_TWO_PI_F / 32.0f;
                    f0 = 1.0f / 32.0f;
                    L = 14;
                } else {
                    f0 = T->ambe_f0[b0];
                    cur.w0 = f0 * MBE_TWO_PI_F;
                    L = T->ambe_L[b0];
                }
                cur.L = L;
                unvc = 0.2046f / sqrtf(cur.w0);
                const uint32_t b1 = (bit(4) << 4) | (bit(5) << 3) | (bit(6) << 2) | (bit(7) << 1) | bit(35);
                const uint32_t b2 = (bit(8) << 4) | (bit(9) << 3) | (bit(10) << 2) | (bit(11) << 1) | bit(36);
                uint32_t b3 = 0, b4 = 0, b5 = 0;
                for (int p = 12; p <= 19; p++) {
                    b3 = (b3 << 1) | bit(p);
                }
                b3 = (b3 << 1) | bit(40);
                for (int p = 20; p <= 23; p++) {
                    b4 = (b4 << 1) | bit(p);
                }
                b4 = (b4 << 3) | (bit(41) << 2) | (bit(42) << 1) | bit(43);
                for (int p = 24; p <= 27; p++) {
                    b5 = (b5 << 1) | bit(p);
                }
                b5 = (b5 << 1) | bit(44);
                const uint32_t b6 = (bit(28) << 3) | (bit(29) << 2) | (bit(30) << 1) | bit(45);
                const uint32_t b7 = (bit(31) << 3) | (bit(32) << 2) | (bit(33) << 1) | bit(46);
                const uint32_t b8 = (bit(34) << 2) | (bit(47) << 1) | bit(48);
                if (l >= 1 && l <= L) {
                    if (silence) {
                        cur.V = 0;
                    } else {
                        int jl = (int)((float)l * 16.0f * f0);
                        jl = jl > 7 ? 7 : jl;
                        cur.V = T->ambe_vuv[b1][jl];
                    }
                }
                cur.gamma = T->ambe_dg[b2] + (0.5f * prev.gamma);
                __syncthreads();
                if (l == 1) {
                    sG[1] = 0.0f;
                } else if (l >= 2 && l <= 4) {
                    sG[l] = T->ambe_prba24[b3][l - 2];
                } else if (l >= 5 && l <= 8) {
                    sG[l] = T->ambe_prba58[b4][l - 5];
                }
                __syncthreads();
                if (l >= 1 && l <= 8) {
                    sR[l] = idct_term(sG, 8, l);
                }
                __syncthreads();
                // block rows: lane (i, k) for i = 1..4, k = 1..6 -> lanes 0..23; the rest of each row is zero
                for (int i = 1; i <= 4; i++) {
                    sCb[i][l] = 0.0f;
                }
                __syncthreads();
                if (l < 24) {
                    const int i = l / 6 + 1, k = l % 6 + 1;
                    const int J = T->ambe_blocks[L][i - 1];
                    float c = 0.0f;
                    if (k == 1) {
                        c = 0.5f * (sR[2 * i - 1] + sR[2 * i]);
                    } else if (k == 2) {
                        c = 0.353553390593274f * (sR[2 * i - 1] - sR[2 * i]);
                    } else if (k <= J) {
                        const float* h = (i == 1) ? T->ambe_hoc5[b5] : ((i == 2) ? T->ambe_hoc6[b6] : ((i == 3) ? T->ambe_hoc7[b7] : T->ambe_hoc8[b8]));
                        c = h[k - 3];
                    }
                    sCb[i][k] = c;
                }
                __syncthreads();
                if (l >= 1 && l <= L) {
                    int first = 1, bi = 1, J = 1;
                    for (int i = 1; i <= 4; i++) {
                        J = T->ambe_blocks[L][i - 1];
                        bi = i;
                        if (l < first + J) {
                            break;
                        }
                        first += J;
                    }
                    Tl = idct_term(sCb[bi], J, l - first + 1);
                }
                // mean of T in harmonic order
                sA[l] = Tl;
                __syncthreads();
                float tsum = 0.0f;
                for (int k = 1; k <= L; k++) {
                    tsum = tsum + sA[k];
                }
                extra = (cur.gamma - half_log2[L]) - (tsum / (float)L);
                rho = 0.65f;
                __syncthreads();
            }
        }
        if (bad == 0) {
            // log-magnitude prediction from `prev`
            const int pL = prev.L;
            const float pl_edge = __shfl(prev.log2Ml, pL <= 56 ? pL : 56);
            const float pl_one = __shfl(prev.log2Ml, 1);
            sPl[l] = (l == 0) ? pl_one : ((l <= pL && l <= 56) ? prev.log2Ml : pl_edge);
            __syncthreads();
            const float ratio = (float)pL / (float)L;
            float sum = 0.0f;
            for (int k = 1; k <= L; k++) {
                const float kl = ratio * (float)k;
                const int ki = (int)kl;
                const float dl = kl - (float)ki;
                sum = sum + (((1.0f - dl) * sPl[ki]) + (dl * sPl[ki + 1]));
            }
            sum = sum * (rho / (float)L);
            if (l >= 1 && l <= L) {
                const float kl = ratio * (float)l;
                const int ki = (int)kl;
                const float dl = kl - (float)ki;
                cur.log2Ml = (((Tl + ((rho * (1.0f - dl)) * sPl[ki])) + ((rho * dl) * sPl[ki + 1])) - sum) + extra;
                const float m = mbe_expf(0.693f * cur.log2Ml);
                cur.Ml = (CODEC == DDN_MBE_AMBE_3600X2450 && !cur.V) ? unvc * m : m;
            }
            __syncthreads();
        }
        // ---- repeat / mute decision (mbe_processImbe4400Dataf / mbe_processAmbe2450Dataf) ----
        bool use_last;
        if (CODEC == DDN_MBE_IMBE_7200X4400) {
            use_last = (bad == 1) || (errs2 > 5);
        } else {
            use_last = (bad == 0) && (errs2 > 3);
        }
        if (use_last) {
            const int rp = cur.repeat;
            cur = prev; // mbe_useLastMbeParms
            cur.repeat = rp + 1;
            flags |= MBE_PROCESS_FLAG_REPEAT;
        } else {
            cur.repeat = 0;
        }
        // mbelib 1.3: mbe_processImbe4400Dataf synthesizes whenever repeat <= 3 (a frame with an invalid fundamental repeats the
        // last good one up to three times, the fourth mutes); only mbe_processAmbe2450Dataf also asks for bad == 0
        if ((CODEC == DDN_MBE_IMBE_7200X4400 || bad == 0) && cur.repeat <= 3) {
            prev = cur; // mbe_moveMbeParms (cur, prev)
            // ---- mbe_spectralAmpEnhance ----
            const int cL = cur.L;
            const float cw = mbe_cosf(cur.w0 * (float)l);
            const float m2 = cur.Ml * cur.Ml;
            sA[l] = m2;
            sB[l] = m2 * cw;
            __syncthreads();
            float Rm0 = 0.0f, Rm1 = 0.0f;
            for (int k = 1; k <= cL; k++) {
                Rm0 = Rm0 + sA[k];
                Rm1 = Rm1 + sB[k];
            }
            __syncthreads();
            const float R2m0 = Rm0 * Rm0, R2m1 = Rm1 * Rm1;
            if (l >= 1 && l <= cL && cur.Ml != 0.0f) {
                const float num = (0.96f * MBE_PI_F) * ((R2m0 + R2m1) - (((2.0f * Rm0) * Rm1) * cw));
                const float den = (cur.w0 * Rm0) * (R2m0 - R2m1);
                const float x = num / den;
                float W = 1.0f;
                if (!((8 * l) <= cL || !(x > 0.0f) || !(x < 3.0e38f))) {
                    const float tmp = sqrtf(cur.Ml) * sqrtf(sqrtf(x));
                    W = (tmp > 1.2f) ? 1.2f : ((tmp < 0.5f) ? 0.5f : tmp);
                }
                cur.Ml = cur.Ml * W;
            }
            sA[l] = cur.Ml * cur.Ml;
            __syncthreads();
            float sum2 = 0.0f;
            for (int k = 1; k <= cL; k++) {
                sum2 = sum2 + sA[k];
            }
            __syncthreads();
            const float gam = (sum2 == 0.0f) ? 1.0f : sqrtf(Rm0 / sum2);
            if (l >= 1 && l <= cL) {
                cur.Ml = gam * cur.Ml;
            }
            // ---- synthesis bookkeeping (mbe_synthesizeSpeechf up to the oscillator bank) against `enh` ----
            const int num_uv = __popcll(__ballot(l >= 1 && l <= cL && cur.V == 0));
            const float cw0 = cur.w0, pw0 = enh.w0;
            int maxl;
            if (cL > enh.L) {
                maxl = cL;
                if (l > enh.L && l <= maxl) {
                    enh.Ml = 0.0f;
                    enh.V = 1;
                }
            } else {
                maxl = enh.L;
                if (l > cL && l <= maxl) {
                    cur.Ml = 0.0f;
                    cur.V = 1;
                }
            }
            const uint32_t fbase = mbe_mix(mbe_mix(0x9E3779B9u, seed), frame_no);
            if (l >= 1 && l <= 56) {
                float psi = enh.PSI + ((pw0 + cw0) * ((float)(l * 160) / 2.0f));
                psi = __builtin_fmaf(-__builtin_rintf(psi * 0.159154943091895336f), MBE_TWO_PI_F, psi);
                cur.PSI = psi;
                if (l <= (cL / 4)) {
                    cur.PHI = psi;
                } else {
                    cur.PHI = psi + (((float)num_uv * mbe_rand_phase(mbe_mix(mbe_mix(fbase, (uint32_t)l), 0x100u))) / (float)cL);
                }
            }
            // the frame's record for k_mbe_synth
            const unsigned long long cvm = __ballot(cur.V != 0), pvm = __ballot(enh.V != 0);
            rec->cMl[l] = cur.Ml;
            rec->pMl[l] = enh.Ml;
            rec->cPHI[l] = cur.PHI;
            rec->pPHI[l] = enh.PHI;
            if (l == 0) {
                rec->cw0 = cw0;
                rec->pw0 = pw0;
                rec->maxl = maxl;
                rec->flags = 0;
                rec->cv_lo = (uint32_t)cvm;
                rec->cv_hi = (uint32_t)(cvm >> 32);
                rec->pv_lo = (uint32_t)pvm;
                rec->pv_hi = (uint32_t)(pvm >> 32);
                rec->fbase = fbase;
            }
            enh = cur; // mbe_moveMbeParms (cur, prev_mp_enhanced)
        } else {
            flags |= MBE_PROCESS_FLAG_MUTE;
            cur = prev = enh = lane_init(l);
            if (l == 0) {
                rec->flags = DDN_MBE_REC_SILENCE;
                rec->maxl = 0;
            }
        }
        frame_no++;
        if (l == 0 && res_out) {
            res_out[fi * 5 + 0] = (int32_t)flags;
            res_out[fi * 5 + 1] = r[1];
            res_out[fi * 5 + 2] = r[2];
            res_out[fi * 5 + 3] = r[3];
            res_out[fi * 5 + 4] = r[4];
        }
    }
    lane_store(&st->cur, cur, l, (int)frame_no);
    lane_store(&st->prev, prev, l, (int)frame_no);
    lane_store(&st->enh, enh, l, (int)frame_no);
    if (l == 0) {
        st->frame_no = frame_no;
    }
}

// ---- oscillator bank --------------------------------------------------------------------------------------------------
__device__ __forceinline__ float
unvoiced_mix(float w0, float w0l, int l, int n, uint32_t base, uint32_t tag) {
    float c3 = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float rph = mbe_rand_phase(mbe_mix(base, tag + (uint32_t)i));
        c3 = c3 + mbe_cosf(((w0 * (float)n) * (((float)l + ((float)i * MBE_UVSTEP)) - MBE_UVOFFSET)) + rph);
        if (w0l > MBE_UVTHRESHOLD) {
            c3 = c3 + (((w0l - MBE_UVTHRESHOLD) * MBE_UVRAND) * mbe_u01(mbe_mix(mbe_mix(base, tag + 0x40u + (uint32_t)i), (uint32_t)n)));
        }
    }
    return c3;
}

__global__ __launch_bounds__(64) void
k_mbe_synth(const DdnMbeFrameRec* __restrict__ recs, float* __restrict__ pcm) {
    const DdnMbeFrameRec* rec = recs + blockIdx.x;
    float* out = pcm + (size_t)blockIdx.x * 160;
    const int lane = threadIdx.x;
    const int maxl = rec->maxl;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    if (!(rec->flags & DDN_MBE_REC_SILENCE)) {
        const float cw0 = rec->cw0, pw0 = rec->pw0;
        const uint32_t fbase = rec->fbase;
        const unsigned long long cvm = ((unsigned long long)rec->cv_hi << 32) | rec->cv_lo;
        const unsigned long long pvm = ((unsigned long long)rec->pv_hi << 32) | rec->pv_lo;
        // The synthesis window is exactly zero on the previous frame's side from sample 105 on and on the current frame's side up
        // to sample 55, and a term weighted by that zero adds +-0 to a sum that is never -0: the three samples a lane owns are
        // taken one from each stretch (0..55: previous side only, 56..104: both, 105..159: current side only), so the two
        // one-sided stretches cost one oscillator per harmonic instead of two - same bits, a third fewer cosines.
        const int nA = lane, nB = 56 + lane, nC = 105 + lane;
        const bool inA = lane < 56, inB = lane < 49, inC = lane < 55;
        const float wpA = mbe_ws(nA + 160), wpB = mbe_ws(nB + 160), wcB = mbe_ws(nB), wcC = mbe_ws(nC);
        for (int l = 1; l <= maxl; l++) {
            const float cw0l = cw0 * (float)l, pw0l = pw0 * (float)l;
            const uint32_t base = mbe_mix(fbase, (uint32_t)l);
            const int cv = (int)((cvm >> l) & 1ull), pv = (int)((pvm >> l) & 1ull);
            const float cM = rec->cMl[l], pM = rec->pMl[l], cP = rec->cPHI[l], pP = rec->pPHI[l];
            // previous frame's side of sample n: a voiced oscillator or the unvoiced mix, weighted by Ws(n + 160)
            auto prev_side = [&](int n, float w) {
                if (pv == 1) {
                    return (w * pM) * mbe_cosf((pw0l * (float)n) + pP);
                }
                const float c3 = unvoiced_mix(pw0, pw0l, l, n, base, 0x300u);
                return (((c3 * MBE_UVSINE) * w) * pM) * MBE_QFACTOR;
            };
            auto cur_side = [&](int n, float w) {
                if (cv == 1) {
                    return (w * cM) * mbe_cosf((cw0l * (float)(n - 160)) + cP);
                }
                const float c3 = unvoiced_mix(cw0, cw0l, l, n, base, 0x200u);
                return (((c3 * MBE_UVSINE) * w) * cM) * MBE_QFACTOR;
            };
            // a side whose amplitude is exactly zero (the harmonics above that frame's L) likewise adds +-0: not evaluated
            const bool p_on = pM != 0.0f, c_on = cM != 0.0f;
            if (inA && p_on) {
                acc[0] = acc[0] + prev_side(nA, wpA);
            }
            if (inB && (p_on || c_on)) {
                // (the reference adds the two sides in an order that depends on the voicing pair; the sum is the same)
                acc[1] = acc[1] + ((p_on ? prev_side(nB, wpB) : 0.0f) + (c_on ? cur_side(nB, wcB) : 0.0f));
            }
            if (inC && c_on) {
                acc[2] = acc[2] + cur_side(nC, wcC);
            }
        }
    }
    if (lane < 56) {
        out[lane] = acc[0];
    }
    if (lane < 49) {
        out[56 + lane] = acc[1];
    }
    if (lane < 55) {
        out[105 + lane] = acc[2];
    }
}

__global__ void
k_mbe_result_skip(const uint8_t* __restrict__ skip, size_t n, int32_t* __restrict__ result) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && skip[i]) {
        result[i * 5] = (int32_t)((uint32_t)result[i * 5] | DDN_MBE_RESULT_INVALID);
    }
}

__global__ void
k_mbe_stream_init(DdnMbeStream* streams, int n, uint32_t seed0) {
    const int s = blockIdx.x;
    const int l = threadIdx.x;
    if (s >= n) {
        return;
    }
    const LaneParms p = lane_init(l);
    lane_store(&streams[s].cur, p, l, 0);
    lane_store(&streams[s].prev, p, l, 0);
    lane_store(&streams[s].enh, p, l, 0);
    if (l == 0) {
        streams[s].frame_no = 0;
        streams[s].seed = seed0 + (uint32_t)s;
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_mbe_frame_decode(int codec, const uint8_t* frames, const uint8_t* soft, size_t n, uint8_t* bits, int32_t* result,
                         hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    { // the Golay table of this device, built at the first decode
        static std::mutex mu;
        static bool up[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) {
            return hipErrorInvalidDevice;
        }
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64 || !up[dev]) {
            hipLaunchKernelGGL(k_golay_tab_init, dim3(1), dim3(256), 0, st);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) {
                e = hipStreamSynchronize(st); // other streams' decodes may follow at once
            }
            if (e != hipSuccess) {
                return e;
            }
            if (dev >= 0 && dev < 64) {
                up[dev] = true;
            }
        }
    }
    const dim3 grid((unsigned)((n + DDN_MBE_FRAMES_PER_WG - 1) / DDN_MBE_FRAMES_PER_WG)), blk(64);
    if (codec == DDN_MBE_IMBE_7200X4400) {
        hipLaunchKernelGGL(k_mbe_frame_decode<DDN_MBE_IMBE_7200X4400>, grid, blk, 0, st, frames, soft, n, bits, result);
    } else {
        hipLaunchKernelGGL(k_mbe_frame_decode<DDN_MBE_AMBE_3600X2450>, grid, blk, 0, st, frames, soft, n, bits, result);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_mbe_stream_init(DdnMbeStream* streams, int n_streams, uint32_t seed0, hipStream_t st) {
    if (n_streams <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_mbe_stream_init, dim3((unsigned)n_streams), dim3(64), 0, st, streams, n_streams, seed0);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_mbe_params(int codec, const uint8_t* bits, const int32_t* res_in, int n_streams, int n_frames,
                   const ddn_mbe_tables* d_tables, const float* d_half_log2, DdnMbeStream* streams, int tail_rule,
                   DdnMbeFrameRec* recs, int32_t* res_out, hipStream_t st) {
    if (n_streams <= 0 || n_frames <= 0) {
        return hipSuccess;
    }
    const dim3 grid((unsigned)n_streams), blk(64);
    if (codec == DDN_MBE_IMBE_7200X4400) {
        hipLaunchKernelGGL(k_mbe_params<DDN_MBE_IMBE_7200X4400>, grid, blk, 0, st, bits, res_in, n_frames, d_tables,
                           d_half_log2, streams, tail_rule, recs, res_out);
    } else {
        hipLaunchKernelGGL(k_mbe_params<DDN_MBE_AMBE_3600X2450>, grid, blk, 0, st, bits, res_in, n_frames, d_tables,
                           d_half_log2, streams, tail_rule, recs, res_out);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_mbe_synth(const DdnMbeFrameRec* recs, size_t n_frames_total, float* pcm, hipStream_t st) {
    if (n_frames_total == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_mbe_synth, dim3((unsigned)n_frames_total), dim3(64), 0, st, recs, pcm);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_mbe_result_skip(const uint8_t* skip, size_t n, int32_t* result, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_mbe_result_skip, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, skip, n, result);
    return hipGetLastError();
}
