// ddn_block.hip — batched block-code decoders of the P25 Phase 1 path (bit-exact integer arithmetic).
//
//   k_nid_decode        P25p1 NID: BCH(63,16,11) over GF(2^6) + DUID / parity validation + NAC-retry +
//                       Chase search (<= 3 flips among the <= 8 least reliable bits)
//                       reference: include/dsd-neo/fec/BCH_63_16.hpp:47-330,
//                                  src/protocol/p25/phase1/p25p1_check_nid.cpp:200-354
//   k_hamming_10_6_3    Hamming(10,6,3) single-error correct / double-error detect
//                       reference: src/fec/hamming_10_6_3.cpp:20-105
//
// One codeword per lane: these codes are tiny (63 / 10 bits) and the work per codeword is data dependent
// (0 .. 186 BCH trials on the soft path), so there is no useful intra-codeword parallelism; the batch supplies it.
// GF(64) exp/log tables live in LDS (per-lane random access).  The BCH decoder is Massey's polynomial-domain
// Berlekamp-Massey + Chien search: a bounded-distance decoder, hence the same (success, error count, data) as
// the reference's index-form implementation for every input (pinned on 0..16-error patterns and random words).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include "ddn_device.h"
#include "ddn_nid_dev.h"

using namespace ddn_nid;

__global__ __launch_bounds__(64) void
k_nid_decode(const uint8_t* __restrict__ bits63, const uint8_t* __restrict__ rel63, const int32_t* __restrict__ obs_nac,
             const uint8_t* __restrict__ parity, const uint8_t* __restrict__ parity_rel, int threshold, int n,
             int32_t* __restrict__ out4, int32_t* __restrict__ chase_list, int32_t* __restrict__ chase_count) {
    __shared__ uint8_t ex[128];
    __shared__ uint8_t lg[64];
    __shared__ uint8_t work[(23 + 24 + 24 + 24) * 64];
    if (threadIdx.x == 0) {
        gf_fill(ex, lg);
    }
    __syncthreads();
    const Gf gf = {ex, lg};
    const Work wk = {work + threadIdx.x, work + 23 * 64 + threadIdx.x, work + 47 * 64 + threadIdx.x,
                     work + 71 * 64 + threadIdx.x};
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) {
        return;
    }
    const uint8_t* bp = bits63 + (size_t)c * 63;
    uint64_t w = 0;
    for (int p = 0; p < 63; p++) {
        w |= (uint64_t)(bp[p] ? 1 : 0) << p;
    }
    const int par = parity ? (parity[c] ? 1 : 0) : 0;
    const int prel = parity_rel ? parity_rel[c] : 0;
    const int obs = obs_nac ? obs_nac[c] : 0;
    const bool obs_ok = obs > 0 && obs < 0xFFF;

    int failed = 0;
    NidRes hard = nid_codeword(gf, wk, w, par, &failed);
    if (hard.status == 0 && failed && obs_ok && rx_nac(w) != obs) {
        hard = nid_codeword(gf, wk, put_nac(w, obs), par, nullptr);
    }
    NidRes res = hard;
    if (hard.status <= 0 && rel63 && chase_list) {
        // the Chase search runs in k_nid_chase, one wavefront per NID with one candidate per lane
        chase_list[atomicAdd(chase_count, 1)] = c;
    } else if (hard.status <= 0 && rel63) {
        const uint8_t* rel = rel63 + (size_t)c * 63; // read in place (L1/L2-resident, 63 bytes per lane)
        // the 8 least reliable positions in (reliability, index) order; pool = first max(6, min(8, #below thr))
        uint64_t pool = 0; // 8 positions, one per byte
        uint64_t taken = 0;
        int below = 0;
        for (int i = 0; i < 63; i++) {
            below += rel[i] < threshold;
        }
        for (int k = 0; k < 8; k++) {
            int bi = -1, bv = 256;
            for (int i = 0; i < 63; i++) {
                if (!((taken >> i) & 1) && rel[i] < bv) {
                    bv = rel[i];
                    bi = i;
                }
            }
            pool |= (uint64_t)bi << (8 * k);
            taken |= 1ull << bi;
        }
        int np = below < 8 ? below : 8;
        if (np < 6) {
            np = 6;
        }
        ChaseBest best = {0, {0, 0, 0, 0}, 0, 0};
        chase_from(gf, wk, w, rel, pool, np, par, prel, threshold, &best);
        if (obs_ok && rx_nac(w) != obs) {
            chase_from(gf, wk, put_nac(w, obs), rel, pool, np, par, prel, threshold, &best);
        }
        if (best.found) {
            res = best.dec;
        }
    }
    int32_t* o = out4 + (size_t)c * 4;
    o[0] = res.status;
    o[1] = res.nac;
    o[2] = res.duid;
    o[3] = res.errs;
}

// bits10: [n][10] one bit per byte, 6 data bits then 4 parity bits.  data6 (in place: the first 6 bytes of each
// row) is rewritten only for single-bit corrections, exactly like hamming_10_6_3_decode(); errs[n] = 0/1/2.
__global__ void
k_hamming_10_6_3(uint8_t* __restrict__ bits10, int n, uint8_t* __restrict__ errs, DdnSel sel) {
    auto one = [&](long c) {
    uint8_t* b = bits10 + (size_t)c * 10;
    int word = 0;
    bool bad = false;
    for (int i = 0; i < 10; i++) {
        bad |= b[i] > 1;
        word = (word << 1) | (b[i] & 1);
    }
    if (bad) {
        errs[c] = 2;
        return;
    }
    const int masks[4] = {0x398, 0x354, 0x2E2, 0x1E1};
    int syn = 0;
    for (int k = 0; k < 4; k++) {
        syn = (syn << 1) | (__popc((unsigned)(word & masks[k])) & 1);
    }
    int e = 0;
    if (syn) {
        // syndrome -> bit index (0..3 parity bits, 4..9 data bits), -1 = uncorrectable
        const int8_t bad_bit[16] = {-2, 0, 1, 5, 2, -1, -1, 6, 3, -1, -1, 7, 4, 8, 9, -1};
        const int bb = bad_bit[syn];
        if (bb < 0) {
            e = 2;
        } else {
            e = 1;
            if (bb >= 4) {
                word ^= 1 << bb;
            }
            for (int i = 0; i < 6; i++) {
                b[i] = (uint8_t)((word >> (9 - i)) & 1);
            }
        }
    }
    errs[c] = (uint8_t)e;
    };
    ddn_sel_for_each(sel, (long)n, one);
}

// Chase search of one NID per wavefront: lane t evaluates candidate t of the reference's sequence (base word, then the
// base with the observed NAC written in; masks of <= 3 flips over the np least reliable bits in increasing order) and a
// wave-wide minimum over the packed key (score, status != 1, error count, flips, sequence index) picks what the
// reference's sequential "strictly better" scan would have kept (src/protocol/p25/phase1/p25p1_check_nid.cpp:204-322).
__global__ __launch_bounds__(64) void
k_nid_chase(const uint8_t* __restrict__ bits63, const uint8_t* __restrict__ rel63, const int32_t* __restrict__ obs_nac,
            const uint8_t* __restrict__ parity, const uint8_t* __restrict__ parity_rel, int threshold,
            const int32_t* __restrict__ chase_list, const int32_t* __restrict__ chase_count, int32_t* __restrict__ out4) {
    __shared__ uint8_t ex[128];
    __shared__ uint8_t lg[64];
    __shared__ uint8_t work[(23 + 24 + 24 + 24) * 64];
    __shared__ uint8_t masks[96];
    if ((int)blockIdx.x >= *chase_count) {
        return;
    }
    const int lane = threadIdx.x;
    if (lane == 0) {
        gf_fill(ex, lg);
        int k = 0;
        for (int m = 0; m < 256; m++) { // masks with at most three flips, increasing: 93 of them (42 below 64, 64 below 128)
            if (__popc((unsigned)m) <= 3) {
                masks[k++] = (uint8_t)m;
            }
        }
    }
    __syncthreads();
    const Gf gf = {ex, lg};
    const Work wk = {work + lane, work + 23 * 64 + lane, work + 47 * 64 + lane, work + 71 * 64 + lane};
    const int c = chase_list[blockIdx.x];
    const uint8_t* bp = bits63 + (size_t)c * 63;
    const uint8_t* rel = rel63 + (size_t)c * 63;
    uint64_t w = 0;
    for (int p = 0; p < 63; p++) {
        w |= (uint64_t)(bp[p] ? 1 : 0) << p;
    }
    const int par = parity ? (parity[c] ? 1 : 0) : 0;
    const int prel = parity_rel ? parity_rel[c] : 0;
    const int obs = obs_nac ? obs_nac[c] : 0;
    const bool two_bases = obs > 0 && obs < 0xFFF && rx_nac(w) != obs;
    uint64_t pool = 0, taken = 0;
    int below = 0;
    for (int i = 0; i < 63; i++) {
        below += rel[i] < threshold;
    }
    for (int k = 0; k < 8; k++) {
        int bi = -1, bv = 256;
        for (int i = 0; i < 63; i++) {
            if (!((taken >> i) & 1) && rel[i] < bv) {
                bv = rel[i];
                bi = i;
            }
        }
        pool |= (uint64_t)bi << (8 * k);
        taken |= 1ull << bi;
    }
    int np = below < 8 ? below : 8;
    np = np < 6 ? 6 : np;
    const int per_base = (np == 6) ? 42 : ((np == 7) ? 64 : 93);
    const int total = two_bases ? 2 * per_base : per_base;
    uint32_t best_key = 0xFFFFFFFFu;
    NidRes best_dec = {0, 0, 0, 0};
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        uint32_t key = 0xFFFFFFFFu;
        NidRes dec = {0, 0, 0, 0};
        bool run = t < total;
        int mask = 0, base_idx = 0, changed = 0, score = 0;
        uint64_t cand = w;
        if (run) {
            base_idx = t / per_base;
            mask = masks[t - base_idx * per_base];
            changed = __popc((unsigned)mask);
            cand = base_idx ? put_nac(w, obs) : w;
            for (int b = 0; b < np; b++) {
                if (mask & (1 << b)) {
                    const int pos = (int)((pool >> (8 * b)) & 0xFF);
                    cand ^= 1ull << pos;
                    score += rel[pos];
                }
            }
            run = !(changed && score > threshold * changed);
        }
        if (__any(run)) {
            if (run) {
                dec = nid_codeword(gf, wk, cand, par, nullptr);
            }
            if (run && dec.status > 0) {
                const int sc = score + (dec.status == 2 ? prel : 0);
                key = ((uint32_t)sc << 18) | ((uint32_t)(dec.status != 1) << 17) | ((uint32_t)(dec.errs & 63) << 11)
                      | ((uint32_t)changed << 9) | (uint32_t)(base_idx * 256 + mask);
            }
        }
        if (key < best_key) {
            best_key = key;
            best_dec = dec;
        }
    }
    // wave minimum of the (unique) keys, then the owning lane publishes its decode
    uint32_t m = best_key;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off);
        m = o < m ? o : m;
    }
    if (m != 0xFFFFFFFFu && best_key == m) {
        int32_t* o = out4 + (size_t)c * 4;
        o[0] = best_dec.status;
        o[1] = best_dec.nac;
        o[2] = best_dec.duid;
        o[3] = best_dec.errs;
    }
}

// One NID per wavefront through nid_decode_wave() - the decoder the receive loops' handler waves run (ddn_rx.hip, ddn_cqrx.hip):
// hard decode with the polynomials spread over the lanes, then the Chase candidates one per lane.  Small batches take this route:
// a word with bit errors costs one lane of k_nid_decode ~100 k cycles of dependent table look-ups, a wavefront ~10 k.
__global__ __launch_bounds__(64) void
k_nid_wave(const uint8_t* __restrict__ bits63, const uint8_t* __restrict__ rel63, const int32_t* __restrict__ obs_nac,
           const uint8_t* __restrict__ parity, const uint8_t* __restrict__ parity_rel, int threshold, int n,
           int32_t* __restrict__ out4) {
    __shared__ uint8_t ex[128];
    __shared__ uint8_t lg[64];
    __shared__ uint8_t work[(23 + 24 + 24 + 24) * 64];
    __shared__ uint8_t masks[96];
    __shared__ uint8_t relb[64];
    const int lane = threadIdx.x;
    if (lane == 0) {
        gf_fill(ex, lg);
        chase_masks_fill(masks);
    }
    const int c = blockIdx.x;
    const uint8_t* bp = bits63 + (size_t)c * 63;
    const uint64_t w = __ballot(lane < 63 && bp[lane < 63 ? lane : 0] != 0);
    relb[lane] = (rel63 && lane < 63) ? rel63[(size_t)c * 63 + lane] : 0;
    __syncthreads();
    const Gf gf = {ex, lg};
    const Work wk = {work + lane, work + 23 * 64 + lane, work + 47 * 64 + lane, work + 71 * 64 + lane};
    const int par = parity ? (parity[c] ? 1 : 0) : 0;
    const int prel = parity_rel ? parity_rel[c] : 0;
    const int obs = obs_nac ? obs_nac[c] : 0;
    const NidRes r = nid_decode_wave(gf, wk, w, rel63 ? relb : nullptr, par, prel, obs, threshold, masks, lane);
    if (lane == 0) {
        int32_t* o = out4 + (size_t)c * 4;
        o[0] = r.status;
        o[1] = r.nac;
        o[2] = r.duid;
        o[3] = r.errs;
    }
}

// batches up to this many NIDs take the wavefront-per-NID route (environment DDN_NID_WAVE_MAX overrides: 0 = never)
static int
nid_wave_max() {
    static int v = -1;
    if (v < 0) {
        const char* e = DDN_EXP_ENV("DDN_NID_WAVE_MAX");
        const long x = e ? strtol(e, nullptr, 10) : 16384;
        v = (x < 0 || x > (1 << 24)) ? 16384 : (int)x;
    }
    return v;
}

extern "C" hipError_t
ddn_dev_nid_decode(const uint8_t* bits63, const uint8_t* rel63, const int32_t* obs_nac, const uint8_t* parity,
                   const uint8_t* parity_rel, int threshold, int n, int32_t* out4, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    if (n <= nid_wave_max()) {
        hipLaunchKernelGGL(k_nid_wave, dim3((unsigned)n), dim3(64), 0, st, bits63, rel63, obs_nac, parity, parity_rel, threshold, n,
                           out4);
        return hipGetLastError();
    }
    // pass 1: hard decode (+ observed-NAC retry) for every NID, NIDs that need the Chase search are listed;
    // pass 2: one wavefront per listed NID.  The list and its counter are stream-ordered scratch (hipMallocAsync), so
    // concurrent calls on different streams or host threads never share them.
    int32_t* scratch = nullptr;
    int32_t *list = nullptr, *count = nullptr;
    if (rel63) {
        hipError_t e = hipMallocAsync((void**)&scratch, sizeof(int32_t) * ((size_t)n + 1), st);
        if (e != hipSuccess) {
            return e;
        }
        count = scratch;
        list = scratch + 1;
        e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
        if (e != hipSuccess) {
            (void)hipFreeAsync(scratch, st);
            return e;
        }
    }
    hipLaunchKernelGGL(k_nid_decode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, bits63, rel63, obs_nac, parity,
                       parity_rel, threshold, n, out4, list, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !rel63) {
        if (scratch) {
            (void)hipFreeAsync(scratch, st);
        }
        return e;
    }
    hipLaunchKernelGGL(k_nid_chase, dim3((unsigned)n), dim3(64), 0, st, bits63, rel63, obs_nac, parity, parity_rel,
                       threshold, (const int32_t*)list, (const int32_t*)count, out4);
    e = hipGetLastError();
    const hipError_t ef = hipFreeAsync(scratch, st);
    return e != hipSuccess ? e : ef;
}

extern "C" hipError_t
ddn_dev_hamming_10_6_3(uint8_t* bits10, int n, uint8_t* errs, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(24);
    hipLaunchKernelGGL(k_hamming_10_6_3, dim3(ddn_sel_grid(&sel, ((unsigned long)n + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, bits10, n, errs, sel);
    return hipGetLastError();
}

// ---- P25p1 low speed data: (16,8) cyclic code, hard + soft (src/protocol/p25/p25_lsd.c:31-160) -----------------------------
// g(x) = x^8 + x^5 + x^4 + x^3 + 1, parity = (data * x^8) mod g.  One thread per codeword; the soft search (every subset of
// the <= 6 least reliable weak bits, cheapest that decodes) is at most 63 syndrome evaluations on 16-bit words.
namespace {
__device__ __forceinline__ int
lsd_parity_of(int data) {
    int v = data << 8;
#pragma unroll
    for (int i = 15; i >= 8; i--) {
        if (v & (1 << i)) {
            v ^= 0x139 << (i - 8);
        }
    }
    return v & 0xFF;
}

// word = data byte << 8 | parity byte; returns the corrected word or -1
__device__ __forceinline__ int
lsd_hard(int word) {
    const int data = (word >> 8) & 0xFF, parity = word & 0xFF;
    const int synd = parity ^ lsd_parity_of(data);
    if (synd == 0) {
        return word;
    }
    if ((synd & (synd - 1)) == 0) {
        return word ^ synd; // single parity-bit error
    }
#pragma unroll
    for (int pos = 0; pos < 8; pos++) {
        if (lsd_parity_of(1 << (7 - pos)) == synd) {
            return word ^ (1 << (15 - pos));
        }
    }
    return -1;
}

__global__ void
k_p25_lsd(uint8_t* __restrict__ bits16, const int16_t* __restrict__ llr16, int n, uint8_t* __restrict__ ok, DdnSel sel) {
    auto one = [&](long i) {
    uint8_t* b = bits16 + (size_t)i * 16;
    int word = 0;
    for (int k = 0; k < 16; k++) {
        word = (word << 1) | (b[k] & 1);
    }
    int fixed = lsd_hard(word);
    if (fixed < 0 && llr16) {
        const int16_t* l = llr16 + (size_t)i * 16;
        // the <= 6 least reliable bits with |llr| < 64, ordered by (|llr|, position): selection by repeated minimum
        int cand[6], rel[6], nc = 0;
        unsigned taken = 0;
        for (int c = 0; c < 6; c++) {
            int bi = -1, br = 1 << 30;
            for (int k = 0; k < 16; k++) {
                const int r = l[k] < 0 ? -(int)l[k] : (int)l[k];
                if (r < 64 && !(taken & (1u << k)) && r < br) {
                    br = r;
                    bi = k;
                }
            }
            if (bi < 0) {
                break;
            }
            taken |= 1u << bi;
            cand[nc] = bi;
            rel[nc] = br;
            nc++;
        }
        int best = -1, best_pen = 999999;
        for (int mask = 1; mask < (1 << nc); mask++) {
            int w = word, pen = 0;
            for (int c = 0; c < nc; c++) {
                if (mask & (1 << c)) {
                    w ^= 1 << (15 - cand[c]);
                    pen += rel[c];
                }
            }
            if (pen >= best_pen) {
                continue;
            }
            const int f = lsd_hard(w);
            if (f >= 0) {
                best = f;
                best_pen = pen;
            }
        }
        fixed = best;
    }
    if (fixed >= 0) {
        for (int k = 0; k < 16; k++) {
            b[k] = (uint8_t)((fixed >> (15 - k)) & 1);
        }
    }
    ok[i] = fixed >= 0 ? 1 : 0;
    };
    ddn_sel_for_each(sel, (long)n, one);
}
} // namespace

extern "C" hipError_t
ddn_dev_p25_lsd(uint8_t* bits16, const int16_t* llr16, int n, uint8_t* ok, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(2);
    hipLaunchKernelGGL(k_p25_lsd, dim3(ddn_sel_grid(&sel, ((unsigned long)n + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, bits16, llr16, n, ok, sel);
    return hipGetLastError();
}

// ---- CRC-CCITT16 of decoded TSBK / LCCH blocks (src/protocol/p25/p25_crc.c:18-76) -------------------------------------------
// polynomial 0x1021, zero start, MSB first, inverted; good when it equals the two bytes after the payload.  One thread per
// block, a byte at a time through the 8-step bit recurrence (blocks are 12 bytes; there is nothing to tile).
namespace {
__global__ void
k_p25_crc16(const uint8_t* __restrict__ bytes, int item_bytes, int n, uint8_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* b = bytes + (size_t)i * item_bytes;
    unsigned crc = 0;
    for (int k = 0; k < item_bytes - 2; k++) {
        const unsigned v = b[k];
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            const unsigned bit = (v >> j) & 1u;
            crc = (((crc >> 15) & 1u) ^ bit) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
        }
    }
    crc ^= 0xFFFFu;
    ok[i] = crc == (((unsigned)b[item_bytes - 2] << 8) | b[item_bytes - 1]) ? 1 : 0;
}
} // namespace

extern "C" hipError_t
ddn_dev_p25_crc16(const uint8_t* bytes, int item_bytes, int n, uint8_t* ok, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p25_crc16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bytes, item_bytes, n, ok);
    return hipGetLastError();
}
