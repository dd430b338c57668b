// ddn_trellis.hip — batched trellis / Viterbi decoders for gfx950 (bit-exact integer arithmetic).
//
// One codeword per group of S lanes (S = number of trellis states: 4, 8 or 16), 64/S codewords per wavefront:
//   * add-compare-select: lane (codeword, next_state) pulls the predecessor metrics it needs from its
//     neighbours with wave shuffles (ds_bpermute within the S-lane group) — no LDS round trip for metrics;
//   * K=5 decoders: the 16 decision bits of a step are one __ballot() slice, stored once per step to LDS;
//   * 4/8-state decoders: back-pointers stay in registers (2/3 bits per step per lane), traceback fetches them
//     with shuffles;
//   * observations (LLRs / dibits / soft symbols) are staged once, coalesced, into LDS, de-interleaved on the fly.
//
// Reference behaviour reproduced (tie-breaks included):
//   k_p25_half_rate   src/protocol/p25/p25_12.c:204-283      4 states x 49 steps, LLR disagreement costs,
//                                                             start bias 256, strict '<' keeps the lowest predecessor
//   k_r34             src/protocol/dmr/dmr_34_viterbi.c:205-255,365-407   8 states x 49 steps, hard or weighted
//   k_k5_nxdn         src/protocol/nxdn/nxdn_convolution.c:57-99,124-160  16 states, uint16 metrics that wrap,
//                                                             decision = (m0 >= m1), chainback from state 0
//   k_k5_m17          src/core/util/dsd_misc.c:118-283        16 states, uint32 metrics, 0x1FFFE complement costs
// Tables: TIA-102.BAAA / ETSI TS 102 361-1 trellis constants (reference src/fec/trellis34.c, p25_12.c:19).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {

__constant__ uint8_t c_half_rate_nibble[16] = {2, 12, 1, 15, 14, 0, 13, 3, 9, 7, 10, 4, 5, 11, 6, 8};
__constant__ uint8_t c_r34_point_to_nibble[16] = {2, 10, 7, 15, 14, 6, 11, 3, 13, 5, 8, 0, 1, 9, 4, 12};
__constant__ uint8_t c_r34_nibble_to_point[16] = {11, 12, 0, 7, 14, 9, 5, 2, 10, 13, 1, 6, 15, 8, 4, 3};
__constant__ uint8_t c_r34_fsm[64] = {0, 8,  4, 12, 2, 10, 6, 14, 4, 12, 2, 10, 6, 14, 0, 8, 1, 9,  5, 13, 3, 11,
                                      7, 15, 5, 13, 3, 11, 7, 15, 1, 9,  3, 11, 7, 15, 1, 9, 5, 13, 7, 15, 1, 9,
                                      5, 13, 3, 11, 2, 10, 6, 14, 0, 8,  4, 12, 6, 14, 0, 8, 4, 12, 2, 10};

// position of received dibit i inside the de-interleaved block: 49 dibit pairs dealt to 4 lanes round-robin
__device__ __forceinline__ int
deinterleave98(int i) {
    // received order: lane 0 pairs (0,4,8,...,48) [13 pairs], lane 1 (1,5,...,45) [12], lane 2 [12], lane 3 [12]
    const int pair_rx = i >> 1;
    int lane, k;
    if (pair_rx < 13) {
        lane = 0;
        k = pair_rx;
    } else {
        lane = 1 + (pair_rx - 13) / 12;
        k = (pair_rx - 13) % 12;
    }
    return 2 * (lane + 4 * k) + (i & 1);
}

} // namespace

// ------------------------------------------------------------------------------------------------------
// P25 1/2-rate: 4 lanes per codeword
__global__ __launch_bounds__(256) void
k_p25_half_rate(const int16_t* __restrict__ llr, int n, uint8_t* __restrict__ out, int32_t* __restrict__ metric_out) {
    constexpr int CW = 64;               // codewords per block
    __shared__ int32_t d[CW][98 + 1];    // de-interleaved LLR pairs (hi 16 = second llr of the dibit)
    const int tid = threadIdx.x;
    const int cw0 = blockIdx.x * CW;
    for (int idx = tid; idx < CW * 98; idx += 256) {
        const int c = idx / 98, i = idx - c * 98;
        if (cw0 + c < n) {
            const int32_t pair = *(const int32_t*)(llr + ((size_t)(cw0 + c) * 196 + 2 * i));
            d[c][deinterleave98(i)] = pair;
        }
    }
    __syncthreads();
    const int c = tid >> 2, ns = tid & 3;
    const int lane = tid & 63, base = lane & ~3;
    const bool live = (cw0 + c) < n;
    uint32_t prev = (ns == 0) ? 0u : 256u;
    uint64_t bp0 = 0, bp1 = 0;
    uint8_t e[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        e[ps] = c_half_rate_nibble[(ps << 2) | ns];
    }
    for (int t = 0; t < 49; t++) {
        const int32_t p0 = live ? d[c][2 * t] : 0, p1 = live ? d[c][2 * t + 1] : 0;
        int l[4] = {(int16_t)(p0 & 0xFFFF), (int16_t)(p0 >> 16), (int16_t)(p1 & 0xFFFF), (int16_t)(p1 >> 16)};
        uint32_t c0[4], c1[4]; // cost if the expected bit is 0 / 1
#pragma unroll
        for (int b = 0; b < 4; b++) {
            c0[b] = l[b] > 0 ? (uint32_t)l[b] : 0u;
            c1[b] = l[b] < 0 ? (uint32_t)(-l[b]) : 0u;
        }
        uint32_t best = 0xFFFFFFFFu;
        uint32_t arg = 0;
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const uint32_t pm = __shfl(prev, base + ps);
            uint32_t cost = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                cost += ((e[ps] >> (3 - b)) & 1) ? c1[b] : c0[b];
            }
            const uint32_t m = pm + cost;
            if (m < best) {
                best = m;
                arg = ps;
            }
        }
        prev = best;
        if (t < 32) {
            bp0 |= (uint64_t)arg << (2 * t);
        } else {
            bp1 |= (uint64_t)arg << (2 * (t - 32));
        }
    }
    // best final state (lowest index wins ties), traceback on lane ns == 0
    uint32_t bestm = __shfl(prev, base);
    int st = 0;
#pragma unroll
    for (int j = 1; j < 4; j++) {
        const uint32_t m = __shfl(prev, base + j);
        if (m < bestm) {
            bestm = m;
            st = j;
        }
    }
    uint32_t w[3] = {0, 0, 0}; // 12 output bytes, MSB-first dibits
#pragma unroll
    for (int t = 48; t >= 0; t--) {
        if (t < 48) {
            const int byte = t >> 2;
            w[byte >> 2] |= (uint32_t)st << (8 * (byte & 3) + 6 - 2 * (t & 3));
        }
        const uint64_t word = (t < 32) ? bp0 : bp1;
        const int sh = (t < 32) ? 2 * t : 2 * (t - 32);
        const uint32_t lo = __shfl((uint32_t)(word & 0xFFFFFFFFu), base + st);
        const uint32_t hi = __shfl((uint32_t)(word >> 32), base + st);
        const uint64_t src = ((uint64_t)hi << 32) | lo;
        st = (int)((src >> sh) & 3);
    }
    if (live && ns == 0) {
        uint32_t* o = (uint32_t*)(out + (size_t)(cw0 + c) * 12);
        o[0] = w[0];
        o[1] = w[1];
        o[2] = w[2];
        if (metric_out) {
            metric_out[cw0 + c] = (int32_t)(bestm >> 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// 3/4-rate: 8 lanes per codeword
template <bool SOFT>
__global__ __launch_bounds__(256) void
k_r34(const uint8_t* __restrict__ dibits, const uint8_t* __restrict__ reliab, int n, uint8_t* __restrict__ out) {
    constexpr int CW = 32;
    constexpr int INF = 1000000000;
    __shared__ uint8_t dd[CW][100];
    __shared__ uint8_t rr[CW][100];
    const int tid = threadIdx.x;
    const int cw0 = blockIdx.x * CW;
    for (int idx = tid; idx < CW * 98; idx += 256) {
        const int c = idx / 98, i = idx - c * 98;
        if (cw0 + c < n) {
            const int p = deinterleave98(i);
            dd[c][p] = dibits[(size_t)(cw0 + c) * 98 + i] & 3u;
            if (SOFT) {
                rr[c][p] = reliab[(size_t)(cw0 + c) * 98 + i];
            }
        }
    }
    __syncthreads();
    const int c = tid >> 3, ns = tid & 7;
    const int lane = tid & 63, base = lane & ~7;
    const bool live = (cw0 + c) < n;
    int prev = (ns == 0) ? 0 : INF;
    uint64_t bp[3] = {0, 0, 0}; // 21 steps x 3 bits per word
    uint8_t en[8], ep[8];
#pragma unroll
    for (int ps = 0; ps < 8; ps++) {
        ep[ps] = c_r34_fsm[ps * 8 + ns];
        en[ps] = c_r34_point_to_nibble[ep[ps]];
    }
    for (int t = 0; t < 49; t++) {
        const int d0 = live ? dd[c][2 * t] : 0, d1 = live ? dd[c][2 * t + 1] : 0;
        const int nib = (d0 << 2) | d1;
        const int point = c_r34_nibble_to_point[nib];
        const int rhi = (SOFT && live) ? rr[c][2 * t] : 1, rlo = (SOFT && live) ? rr[c][2 * t + 1] : 1;
        int cur = INF;
        int arg = 0;
#pragma unroll
        for (int ps = 0; ps < 8; ps++) {
            const int pm = __shfl(prev, base + ps);
            int cost;
            if (SOFT) {
                const int x = en[ps] ^ nib;
                cost = ((x >> 3) & 1) * rhi + ((x >> 2) & 1) * rhi + ((x >> 1) & 1) * rlo + (x & 1) * rlo;
            } else {
                cost = __popc((unsigned)((ep[ps] ^ point) & 15));
            }
            const int m = pm + cost;
            if (pm < INF && m < cur) {
                cur = m;
                arg = ps;
            }
        }
        prev = cur;
        bp[t / 21] |= (uint64_t)arg << (3 * (t % 21));
    }
    int st = 0; // terminated trellis: traceback from state 0
    uint32_t grp = 0;
    uint8_t bytes[18];
#pragma unroll
    for (int t = 48; t >= 0; t--) {
        if (t < 48) {
            grp |= (uint32_t)(st & 7) << (3 * (7 - (t & 7)));
            if ((t & 7) == 0) {
                const int g = t >> 3;
                bytes[3 * g] = (uint8_t)(grp >> 16);
                bytes[3 * g + 1] = (uint8_t)(grp >> 8);
                bytes[3 * g + 2] = (uint8_t)grp;
                grp = 0;
            }
        }
        const uint64_t word = bp[t / 21];
        const uint32_t lo = __shfl((uint32_t)(word & 0xFFFFFFFFu), base + st);
        const uint32_t hi = __shfl((uint32_t)(word >> 32), base + st);
        const uint64_t src = ((uint64_t)hi << 32) | lo;
        st = (int)((src >> (3 * (t % 21))) & 7);
    }
    if (live && ns == 0) {
        uint16_t* o = (uint16_t*)(out + (size_t)(cw0 + c) * 18);
#pragma unroll
        for (int k = 0; k < 9; k++) {
            o[k] = (uint16_t)(bytes[2 * k] | (bytes[2 * k + 1] << 8));
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K=5, NXDN flavour: 16 lanes per codeword, uint16 metrics
template <bool SOFT>
__global__ __launch_bounds__(256) void
k_k5_nxdn(const uint8_t* __restrict__ sym, const uint8_t* __restrict__ rel, int n, int n_steps, int n_bits,
          uint16_t* __restrict__ metrics_io, uint8_t* __restrict__ out, int out_stride, const uint8_t* __restrict__ wanted,
          int wanted_div) {
    constexpr int CW = 16;
    extern __shared__ uint8_t smem[];
    const int row = 2 * n_steps + 2;                 // bytes per codeword of symbols
    uint8_t* s = smem;                               // [CW][row]
    uint8_t* r = s + CW * row;                       // [CW][row] (SOFT only)
    uint16_t* dec = (uint16_t*)(r + (SOFT ? CW * row : 0)); // [CW][n_steps]
    const int tid = threadIdx.x;
    const int cw0 = blockIdx.x * CW;
    if (wanted) { // optional (the chains' sparse slot arrays): a block of 16 code words none of which is wanted writes zeros and leaves
        const bool in = tid < CW && cw0 + tid < n;
        if (!__syncthreads_or(in && wanted[(cw0 + tid) / wanted_div] != 0)) {
            const int nby = (n_bits + 7) / 8;
            for (int idx = tid; idx < CW * nby; idx += 256) {
                const int c = idx / nby;
                if (cw0 + c < n) {
                    out[(size_t)(cw0 + c) * out_stride + (idx - c * nby)] = 0;
                }
            }
            return;
        }
    }
    for (int idx = tid; idx < CW * 2 * n_steps; idx += 256) {
        const int c = idx / (2 * n_steps), i = idx - c * 2 * n_steps;
        if (cw0 + c < n) {
            s[c * row + i] = sym[(size_t)(cw0 + c) * 2 * n_steps + i];
            if (SOFT) {
                r[c * row + i] = rel[(size_t)(cw0 + c) * 2 * n_steps + i];
            }
        }
    }
    __syncthreads();
    const int c = tid >> 4, j = tid & 15;
    const int lane = tid & 63, base = lane & ~15, grp = (lane >> 4);
    const bool live = (cw0 + c) < n;
    const int i = j >> 1, b = j & 1;
    const int t1 = (i >= 4) ? 2 : 0;
    const int t2 = (((i & 3) == 1) || ((i & 3) == 2)) ? 2 : 0;
    uint32_t m = (live && metrics_io) ? metrics_io[(size_t)(cw0 + c) * 16 + j] : 0u;
    for (int t = 0; t < n_steps; t++) {
        const int s0 = live ? s[c * row + 2 * t] : 0, s1 = live ? s[c * row + 2 * t + 1] : 0;
        const uint32_t lo = __shfl(m, base + i), hi = __shfl(m, base + i + 8);
        const int a0 = t1 - s0, a1 = t2 - s1;
        const uint32_t diff0 = (uint32_t)(a0 < 0 ? -a0 : a0), diff1 = (uint32_t)(a1 < 0 ? -a1 : a1);
        int dbit;
        uint32_t nm;
        if (SOFT) {
            const uint32_t r0 = live ? r[c * row + 2 * t] : 0, r1 = live ? r[c * row + 2 * t + 1] : 0;
            uint32_t metric = (diff0 * r0 + diff1 * r1) / 128u;
            if (metric > 8u) {
                metric = 8u;
            }
            const uint32_t x = b ? (8u - metric) : metric;
            const uint32_t m0 = lo + x, m1 = hi + (8u - x);
            dbit = m0 >= m1;
            nm = (dbit ? m1 : m0) & 0xFFFFu;
        } else {
            const uint32_t metric = diff0 + diff1;
            const uint32_t x = b ? (4u - metric) : metric;
            const uint32_t m0 = (lo + x) & 0xFFFFu, m1 = (hi + (4u - x)) & 0xFFFFu;
            dbit = m0 >= m1;
            nm = dbit ? m1 : m0;
        }
        const unsigned long long mask = __ballot(dbit);
        if (j == 0 && live) {
            dec[c * n_steps + t] = (uint16_t)((mask >> (16 * grp)) & 0xFFFFull);
        }
        m = nm;
    }
    if (live && metrics_io) {
        metrics_io[(size_t)(cw0 + c) * 16 + j] = (uint16_t)m;
    }
    if (live && j == 0) {
        uint8_t* o = out + (size_t)(cw0 + c) * out_stride;
        uint32_t state = 0;
        int t = n_steps;
        uint32_t cur = 0;
        for (int nb = n_bits; nb-- > 0;) {
            --t;
            const uint32_t bit = (dec[c * n_steps + t] >> (state >> 4)) & 1u;
            state = (bit << 7) | (state >> 1);
            cur |= bit << (7 - (nb & 7));
            if ((nb & 7) == 0) {
                o[nb >> 3] = (uint8_t)cur;
                cur = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K=5, libM17 flavour: 16 lanes per codeword, uint32 metrics, uint16 soft symbols, optional depuncture
__global__ __launch_bounds__(256) void
k_k5_m17(const uint16_t* __restrict__ in, int n, int in_len, int u_len, DdnPuncture pu, uint8_t* __restrict__ out,
         int out_stride, uint32_t* __restrict__ cost_out, const uint8_t* __restrict__ wanted) {
    constexpr int CW = 16;
    extern __shared__ uint8_t smem[];
    if (wanted) { // (optional: a workgroup none of whose sixteen code words is wanted leaves them undecoded)
        const int i = blockIdx.x * CW + (int)threadIdx.x;
        if (!__syncthreads_or(threadIdx.x < CW && i < n && wanted[i] != 0)) {
            return;
        }
    }
    uint16_t* um = (uint16_t*)smem;                 // [CW][u_len + 2]
    const int row = u_len + 2;
    const int n_steps = u_len >> 1;
    uint16_t* hist = um + CW * row;                 // [CW][n_steps]
    const int tid = threadIdx.x;
    const int cw0 = blockIdx.x * CW;
    for (int idx = tid; idx < CW * u_len; idx += 256) {
        const int c = idx / u_len, u = idx - c * u_len;
        if (cw0 + c < n) {
            uint16_t v;
            if (pu.p_len > 0) {
                const int full = u / pu.p_len, rem = u - full * pu.p_len;
                const int i = full * pu.ones_total + pu.ones_before[rem];
                v = (pu.keep[rem] && i < in_len) ? in[(size_t)(cw0 + c) * in_len + i] : (uint16_t)0x7FFF;
            } else {
                v = in[(size_t)(cw0 + c) * in_len + u];
            }
            um[c * row + u] = v;
        }
    }
    __syncthreads();
    const int c = tid >> 4, j = tid & 15;
    const int lane = tid & 63, base = lane & ~15, grp = (lane >> 4);
    const bool live = (cw0 + c) < n;
    const int i = j >> 1, b = j & 1;
    const uint32_t k0 = (i >= 4) ? 0xFFFFu : 0u;
    const uint32_t k1 = (((i & 3) == 1) || ((i & 3) == 2)) ? 0xFFFFu : 0u;
    uint32_t m = 0;
    for (int t = 0; t < n_steps; t++) {
        const uint32_t s0 = live ? um[c * row + 2 * t] : 0u, s1 = live ? um[c * row + 2 * t + 1] : 0u;
        const uint32_t lo = __shfl(m, base + i), hi = __shfl(m, base + i + 8);
        const uint32_t metric = (k0 > s0 ? k0 - s0 : s0 - k0) + (k1 > s1 ? k1 - s1 : s1 - k1);
        const uint32_t x = b ? (0x1FFFEu - metric) : metric;
        const uint32_t m0 = lo + x, m1 = hi + (0x1FFFEu - x);
        const int dbit = m0 >= m1;
        const unsigned long long mask = __ballot(dbit);
        if (j == 0 && live) {
            hist[c * n_steps + t] = (uint16_t)((mask >> (16 * grp)) & 0xFFFFull);
        }
        m = dbit ? m1 : m0;
    }
    // minimum final metric over the 16 states
    uint32_t best = m;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if (live && j == 0) {
        uint8_t* o = out + (size_t)(cw0 + c) * out_stride;
        for (int k = 0; k < out_stride; k++) {
            o[k] = 0;
        }
        uint32_t state = 0;
        int bitpos = n_steps + 4;
        for (int pos = n_steps; pos > 0;) {
            bitpos--;
            pos--;
            const uint32_t bit = hist[c * n_steps + pos] & (1u << (state >> 4));
            state >>= 1;
            if (bit) {
                state |= 0x80u;
                o[bitpos >> 3] |= (uint8_t)(1u << (7 - (bitpos & 7)));
            }
        }
        if (cost_out) {
            cost_out[cw0 + c] = best - (uint32_t)(u_len - in_len) * 0x7FFFu;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// P25 1/2-rate LIST decoder (src/protocol/p25/p25_12.c:31-202): 8 survivors per state, 4 lanes per codeword.
// A lane owns one next-state: its 8 survivor metrics stay sorted in registers; every step it pulls the 4 x 8 predecessor
// metrics with shuffles and inserts the 32 extensions with a branch-free compare/shift ladder.  "Insert before the first
// strictly larger metric", applied predecessor by predecessor and rank by rank, is exactly what the ladder does (an
// absent survivor is UINT32_MAX and never inserts).  Back-pointers ((prev_state << 3) | prev_rank, one byte per
// survivor) go to LDS; after the last step every lane traces its 8 paths and lane 0 of the group merges the 32
// candidates in (state, rank) order: duplicates by the 12 output bytes are dropped, the rest kept sorted by metric.
// CW code words per workgroup of NT threads (4 lanes each; NT = max(64, 4 CW)).  <32, 128> for dense batches; <4, 64> for the chains'
// sparse lists: 10 KB of LDS and one wavefront per workgroup, which finds room beside a kernel that fills the device (DDN_WG).
template <int CW, int NT>
__global__ __launch_bounds__(NT) void
k_p25_half_rate_list(const int16_t* __restrict__ llr, int n, int max_cand, uint32_t* __restrict__ cand_out,
                     int32_t* __restrict__ count_out, const uint8_t* __restrict__ wanted) {
    constexpr int K = 8;
    __shared__ int32_t d[CW][98 + 1];
    __shared__ uint2 back[CW][49][4];   // 8 back-pointer bytes per (step, state)
    __shared__ uint4 cand[CW][32];      // {bytes 0-3, 4-7, 8-11, metric}
    __shared__ uint8_t cvalid[CW][32];
    const int tid = threadIdx.x;
    // a workgroup walks groups of 32 code words: with a `wanted` list (the chains' sparse data-unit blocks) the launch is a few
    // workgroups that skip the groups nobody wants - each needs 80 KB of LDS to start at all, and 1792 of them queued behind the
    // matched filter held the decode stream (and with it the next receive loop) for 0.9 ms; without a list it is one group each
    const int n_groups = (n + CW - 1) / CW;
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int cw0 = g * CW;
    __syncthreads(); // (the LDS arrays of the previous group are free)
    if (wanted) { // a group of 32 code words none of which is wanted reports count 0 and is skipped
        const bool in = tid < CW && cw0 + tid < n;
        if (!__syncthreads_or(in && wanted[cw0 + tid] != 0)) {
            if (in) {
                count_out[cw0 + tid] = 0;
            }
            continue;
        }
    }
    for (int idx = tid; idx < CW * 98; idx += NT) {
        const int c = idx / 98, i = idx - c * 98;
        if (cw0 + c < n) {
            const int32_t pair = *(const int32_t*)(llr + ((size_t)(cw0 + c) * 196 + 2 * i));
            d[c][deinterleave98(i)] = pair;
        }
    }
    __syncthreads();
    const int c = tid >> 2, ns = tid & 3;
    const int lane = tid & 63, base = lane & ~3;
    const bool lane_on = c < CW;                 // (NT > 4 CW: the spare lanes of the wavefront carry nothing)
    const bool live = lane_on && (cw0 + c) < n;
    const uint32_t MAXM = 0xFFFFFFFFu;
    uint32_t pm[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        pm[r] = MAXM;
    }
    pm[0] = (ns == 0) ? 0u : 256u;
    uint8_t e[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        e[ps] = c_half_rate_nibble[(ps << 2) | ns];
    }
    for (int t = 0; t < 49; t++) {
        const int32_t p0 = live ? d[c][2 * t] : 0, p1 = live ? d[c][2 * t + 1] : 0;
        int l[4] = {(int16_t)(p0 & 0xFFFF), (int16_t)(p0 >> 16), (int16_t)(p1 & 0xFFFF), (int16_t)(p1 >> 16)};
        uint32_t c0[4], c1[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            c0[b] = l[b] > 0 ? (uint32_t)l[b] : 0u;
            c1[b] = l[b] < 0 ? (uint32_t)(-l[b]) : 0u;
        }
        uint32_t cm[K], cb[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
            cm[r] = MAXM;
            cb[r] = 0;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            uint32_t cost = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                cost += ((e[ps] >> (3 - b)) & 1) ? c1[b] : c0[b];
            }
#pragma unroll
            for (int r = 0; r < K; r++) {
                const uint32_t q = __shfl(pm[r], base + ps);
                const uint32_t m = (q == MAXM) ? MAXM : q + cost;
                const uint32_t bp = (uint32_t)((ps << 3) | r);
#pragma unroll
                for (int i = K - 1; i >= 1; i--) {
                    const bool lt_prev = m < cm[i - 1], lt_cur = m < cm[i];
                    cb[i] = lt_prev ? cb[i - 1] : (lt_cur ? bp : cb[i]);
                    cm[i] = lt_prev ? cm[i - 1] : (lt_cur ? m : cm[i]);
                }
                const bool lt0 = m < cm[0];
                cb[0] = lt0 ? bp : cb[0];
                cm[0] = lt0 ? m : cm[0];
            }
        }
#pragma unroll
        for (int r = 0; r < K; r++) {
            pm[r] = cm[r];
        }
        uint2 w;
        w.x = cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24);
        w.y = cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24);
        if (lane_on) {
            back[c][t][ns] = w;
        }
    }
    __syncthreads();
    // every lane traces its 8 survivors
#pragma unroll 1
    for (int rk = 0; rk < K; rk++) {
        uint32_t mfin = pm[0];
#pragma unroll
        for (int r = 1; r < K; r++) {
            mfin = (r == rk) ? pm[r] : mfin;
        }
        uint32_t w[3] = {0, 0, 0};
        int s = ns, r = rk;
        if (mfin != MAXM && lane_on) {
            for (int t = 48; t >= 0; t--) {
                if (t < 48) {
                    const int byte = t >> 2;
                    w[byte >> 2] |= (uint32_t)s << (8 * (byte & 3) + 6 - 2 * (t & 3));
                }
                const uint2 bw = back[c][t][s];
                const uint32_t word = (r < 4) ? bw.x : bw.y;
                const uint32_t p = (word >> (8 * (r & 3))) & 0xFFu;
                s = (int)((p >> 3) & 3u);
                r = (int)(p & 7u);
            }
        }
        if (lane_on) {
            cand[c][ns * K + rk] = make_uint4(w[0], w[1], w[2], mfin);
            cvalid[c][ns * K + rk] = (mfin != MAXM) ? 1 : 0;
        }
    }
    __syncthreads();
    if (ns == 0 && live) {
        const int mx = max_cand > K ? K : max_cand;
        uint4 outl[K];
        int count = 0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            outl[i] = make_uint4(0, 0, 0, 0);
        }
        for (int k = 0; k < 32; k++) {
            if (!cvalid[c][k]) {
                continue;
            }
            const uint4 cd = cand[c][k];
            bool dup = false;
            int at = count;
            bool found = false;
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (i < count) {
                    dup |= (outl[i].x == cd.x && outl[i].y == cd.y && outl[i].z == cd.z);
                    if (!found && cd.w < outl[i].w) {
                        at = i;
                        found = true;
                    }
                }
            }
            if (dup) {
                continue;
            }
            if (count < mx) {
                count++;
            } else if (at >= mx) {
                continue;
            }
#pragma unroll
            for (int i = K - 1; i >= 1; i--) {
                if (i < count && i > at) {
                    outl[i] = outl[i - 1];
                }
            }
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (i == at) {
                    outl[i] = cd;
                }
            }
        }
        uint32_t* o = cand_out + (size_t)(cw0 + c) * (K * 4);
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint4 v = (i < count) ? outl[i] : make_uint4(0, 0, 0, 0);
            // 12 output bytes are MSB-first inside each 32-bit group of 4: byte j of the group sits at bits 8*j
            o[4 * i + 0] = v.x;
            o[4 * i + 1] = v.y;
            o[4 * i + 2] = v.z;
            o[4 * i + 3] = v.w;
        }
        count_out[cw0 + c] = count;
    }
    } // groups
}

// ------------------------------------------------------------------------------------------------------
// 3/4-rate LIST decoder (src/protocol/dmr/dmr_34_viterbi.c:255-362,446-474): 32 survivors per state, 8 lanes per codeword.
// The reference inserts an extension before the first survivor whose metric is >= its own, predecessor by predecessor and
// rank by rank, i.e. it keeps the 32 smallest extensions under (metric ascending, arrival index DEscending).  Each
// extension is therefore one 32-bit key (metric << 8) | (255 - (prev_state * 32 + prev_rank)) - metrics stay below 2^24,
// all keys of a step are distinct - and a lane keeps its next-state's 32 smallest keys sorted in registers with a
// min/max insertion ladder: c[i] = max(c[i-1], min(c[i], key)) walked from the top.  The key's low byte IS the back
// pointer.  Back-pointers (32 bytes per state per step) go to a global scratch area; the candidates are state 0's
// survivors in list order, each traced by one of the group's lanes.
template <bool SOFT>
__global__ __launch_bounds__(64) void
k_r34_list(const uint8_t* __restrict__ dibits, const uint8_t* __restrict__ reliab, int n, int max_cand,
           uint8_t* __restrict__ backs, uint32_t* __restrict__ cand_out, int32_t* __restrict__ count_out,
           const uint8_t* __restrict__ wanted) {
    constexpr int CW = 8, K = 32;
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    __shared__ uint8_t dd[CW][100];
    __shared__ uint8_t rr[CW][100];
    const int tid = threadIdx.x;
    const int cw0 = blockIdx.x * CW;
    // wanted (optional): only the codewords with a non-zero byte are decoded, the others report count 0
    const bool in_range = (cw0 + (tid >> 3)) < n;
    const bool want_me = in_range && (!wanted || wanted[cw0 + (tid >> 3)] != 0);
    if (!__any(want_me)) {
        if ((tid & 7) == 0 && in_range) {
            count_out[cw0 + (tid >> 3)] = 0;
        }
        return;
    }
    for (int idx = tid; idx < CW * 98; idx += 64) {
        const int c = idx / 98, i = idx - c * 98;
        if (cw0 + c < n) {
            const int p = deinterleave98(i);
            dd[c][p] = dibits[(size_t)(cw0 + c) * 98 + i] & 3u;
            if (SOFT) {
                rr[c][p] = reliab[(size_t)(cw0 + c) * 98 + i];
            }
        }
    }
    __syncthreads();
    const int c = tid >> 3, ns = tid & 7;
    const int base = tid & ~7;
    const bool live = want_me;
    const size_t cw = (size_t)(cw0 + (live ? c : 0));
    uint32_t pk[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        pk[r] = EMPTY;
    }
    if (ns == 0) {
        pk[0] = 255u; // metric 0
    }
    uint8_t en[8];
#pragma unroll
    for (int ps = 0; ps < 8; ps++) {
        en[ps] = c_r34_point_to_nibble[c_r34_fsm[ps * 8 + ns]];
    }
    for (int t = 0; t < 49; t++) {
        const int d0 = live ? dd[c][2 * t] : 0, d1 = live ? dd[c][2 * t + 1] : 0;
        const int nib = (d0 << 2) | d1;
        const int rhi = (SOFT && live) ? rr[c][2 * t] : 1, rlo = (SOFT && live) ? rr[c][2 * t + 1] : 1;
        uint32_t ck[K];
#pragma unroll
        for (int r = 0; r < K; r++) {
            ck[r] = EMPTY;
        }
#pragma unroll
        for (int ps = 0; ps < 8; ps++) {
            const int x = en[ps] ^ nib;
            uint32_t cost;
            if (SOFT) {
                cost = (uint32_t)(((x >> 3) & 1) * rhi + ((x >> 2) & 1) * rhi + ((x >> 1) & 1) * rlo + (x & 1) * rlo);
            } else {
                cost = x ? (uint32_t)(256 + __popc((unsigned)x)) : 0u;
            }
#pragma unroll
            for (int pr = 0; pr < K; pr++) {
                const uint32_t q = __shfl(pk[pr], base + ps);
                const uint32_t key = (q == EMPTY) ? EMPTY : ((((q >> 8) + cost) << 8) | (uint32_t)(255 - (ps * K + pr)));
#pragma unroll
                for (int i = K - 1; i >= 1; i--) {
                    ck[i] = max(ck[i - 1], min(ck[i], key));
                }
                ck[0] = min(ck[0], key);
            }
        }
        // back-pointers of this step: 32 bytes per (codeword, step, state)
        if (live) {
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                w[j] = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t k = ck[4 * j + b];
                    const uint32_t bp = (k == EMPTY) ? 0u : (255u - (k & 255u));
                    w[j] |= bp << (8 * b);
                }
            }
            uint4* bo = (uint4*)(backs + ((cw * 49 + t) * 8 + ns) * 32);
            bo[0] = make_uint4(w[0], w[1], w[2], w[3]);
            bo[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
#pragma unroll
        for (int r = 0; r < K; r++) {
            pk[r] = ck[r];
        }
    }
    __threadfence_block();
    __syncthreads();
    // candidates = survivors of state 0 in list order; lane j of the group traces ranks j, j+8, j+16, j+24
    const int mx = max_cand > K ? K : max_cand;
    uint32_t mine[4];
    int count = 0;
#pragma unroll
    for (int r = 0; r < K; r++) {
        const uint32_t v = __shfl(pk[r], base);
        count += (v != EMPTY) ? 1 : 0;
        if ((r & 7) == ns) {
            mine[r >> 3] = v;
        }
    }
    count = count > mx ? mx : count; // survivors are packed at the front of the list
    if (live) {
#pragma unroll 1
        for (int j = 0; j < 4; j++) {
            const int r0 = ns + 8 * j;
            uint32_t* o = cand_out + ((size_t)cw * K + r0) * 6; // {metric, 18 bytes, 2 pad} = 24 bytes
            if (r0 < count) {
                uint32_t g24[6] = {0, 0, 0, 0, 0, 0};
                int s = 0, rk = r0;
                for (int t = 48; t >= 0; t--) {
                    if (t < 48) {
                        g24[t >> 3] |= (uint32_t)s << (21 - 3 * (t & 7));
                    }
                    const uint32_t idx = backs[((cw * 49 + t) * 8 + s) * 32 + rk];
                    s = (int)(idx >> 5);
                    rk = (int)(idx & 31u);
                }
                uint8_t by[20];
#pragma unroll
                for (int g = 0; g < 6; g++) {
                    by[3 * g] = (uint8_t)(g24[g] >> 16);
                    by[3 * g + 1] = (uint8_t)(g24[g] >> 8);
                    by[3 * g + 2] = (uint8_t)g24[g];
                }
                by[18] = 0;
                by[19] = 0;
                o[0] = mine[j] >> 8;
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    o[1 + q] = (uint32_t)by[4 * q] | ((uint32_t)by[4 * q + 1] << 8) | ((uint32_t)by[4 * q + 2] << 16)
                               | ((uint32_t)by[4 * q + 3] << 24);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    o[q] = 0;
                }
            }
        }
        if (ns == 0) {
            count_out[cw] = count;
        }
    } else if (ns == 0 && in_range) {
        count_out[cw0 + c] = 0;
    }
}

// ------------------------------------------------------------------------------------------------------
extern "C" hipError_t
ddn_dev_p25_half_rate(const int16_t* llr, int n, uint8_t* out, int32_t* metric, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p25_half_rate, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, llr, n, out, metric);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_r34(const uint8_t* dibits, const uint8_t* reliab, int n, uint8_t* out, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    dim3 grid((unsigned)((n + 31) / 32));
    if (reliab) {
        hipLaunchKernelGGL((k_r34<true>), grid, dim3(256), 0, st, dibits, reliab, n, out);
    } else {
        hipLaunchKernelGGL((k_r34<false>), grid, dim3(256), 0, st, dibits, reliab, n, out);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_k5_nxdn_wanted(const uint8_t* sym, const uint8_t* rel, int n, int n_steps, int n_bits, uint16_t* metrics_io,
                       uint8_t* out, int out_stride, const uint8_t* wanted, int wanted_div, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    dim3 grid((unsigned)((n + 15) / 16));
    const size_t row = 2 * (size_t)n_steps + 2;
    const size_t shm = 16 * row * (rel ? 2 : 1) + 16 * (size_t)n_steps * 2 + 16;
    if (rel) {
        hipLaunchKernelGGL((k_k5_nxdn<true>), grid, dim3(256), shm, st, sym, rel, n, n_steps, n_bits, metrics_io, out,
                           out_stride, wanted, wanted_div > 0 ? wanted_div : 1);
    } else {
        hipLaunchKernelGGL((k_k5_nxdn<false>), grid, dim3(256), shm, st, sym, rel, n, n_steps, n_bits, metrics_io, out,
                           out_stride, wanted, wanted_div > 0 ? wanted_div : 1);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_k5_nxdn(const uint8_t* sym, const uint8_t* rel, int n, int n_steps, int n_bits, uint16_t* metrics_io,
                uint8_t* out, int out_stride, hipStream_t st) {
    return ddn_dev_k5_nxdn_wanted(sym, rel, n, n_steps, n_bits, metrics_io, out, out_stride, nullptr, 1, st);
}

extern "C" hipError_t
ddn_dev_k5_m17_wanted(const uint16_t* in, int n, int in_len, int u_len, const DdnPuncture* pu, uint8_t* out, int out_stride,
                      uint32_t* cost, const uint8_t* wanted, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    dim3 grid((unsigned)((n + 15) / 16));
    const size_t shm = 16 * ((size_t)u_len + 2) * 2 + 16 * (size_t)(u_len / 2) * 2 + 16;
    hipLaunchKernelGGL(k_k5_m17, grid, dim3(256), shm, st, in, n, in_len, u_len, *pu, out, out_stride, cost, wanted);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_k5_m17(const uint16_t* in, int n, int in_len, int u_len, const DdnPuncture* pu, uint8_t* out, int out_stride,
               uint32_t* cost, hipStream_t st) {
    return ddn_dev_k5_m17_wanted(in, n, in_len, u_len, pu, out, out_stride, cost, nullptr, st);
}

extern "C" hipError_t
ddn_dev_p25_half_rate_list_wanted(const int16_t* llr, int n, int max_cand, const uint8_t* wanted, uint32_t* cand, int32_t* count,
                                  hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    if (wanted) {
        const unsigned groups = (unsigned)((n + 3) / 4);
        hipLaunchKernelGGL((k_p25_half_rate_list<4, 64>), dim3(groups > 1024 ? 1024u : groups), dim3(64), 0, st, llr, n, max_cand, cand, count,
                           wanted);
    } else {
        hipLaunchKernelGGL((k_p25_half_rate_list<32, 128>), dim3((unsigned)((n + 31) / 32)), dim3(128), 0, st, llr, n, max_cand, cand, count,
                           wanted);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_p25_half_rate_list(const int16_t* llr, int n, int max_cand, uint32_t* cand, int32_t* count, hipStream_t st) {
    return ddn_dev_p25_half_rate_list_wanted(llr, n, max_cand, nullptr, cand, count, st);
}

extern "C" hipError_t
ddn_dev_r34_list_wanted(const uint8_t* dibits, const uint8_t* reliab, int n, int max_cand, const uint8_t* wanted, uint8_t* backs,
                        uint32_t* cand, int32_t* count, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const dim3 grid((unsigned)((n + 7) / 8)), blk(64);
    if (reliab) {
        hipLaunchKernelGGL((k_r34_list<true>), grid, blk, 0, st, dibits, reliab, n, max_cand, backs, cand, count, wanted);
    } else {
        hipLaunchKernelGGL((k_r34_list<false>), grid, blk, 0, st, dibits, reliab, n, max_cand, backs, cand, count, wanted);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_r34_list(const uint8_t* dibits, const uint8_t* reliab, int n, int max_cand, uint8_t* backs, uint32_t* cand,
                 int32_t* count, hipStream_t st) {
    return ddn_dev_r34_list_wanted(dibits, reliab, n, max_cand, nullptr, backs, cand, count, st);
}


// ---- DMR rate 3/4 data bursts: the candidate pool of dmr_dburst_pick_trellis_payload() (src/protocol/dmr/dmr_dburst.c:397-536) ----------
// One thread per wanted burst: pool = {hard decode, soft decode, the list decoder's candidates} in that order, identical payloads
// once, each with dmr_r34_candidate_metric() (dmr_34_viterbi.c:410-444: the weighted cost of the payload's own path), its DBSN and
// whether its CRC9 checks (dmr_dburst_trellis_candidate_metrics()).  Two picks: the one the reference makes while
// state->data_conf_data = 0 (the cheaper of hard and soft - the list is not consulted then) and the one it makes for confirmed data
// before the DBSN expectation is applied (cheapest candidate with a good CRC9, else the cheapest); the pool itself is kept for the
// caller that tracks the DBSN sequence.  Pool entry = ddn_r34_candidate {metric, 18 bytes} + {crc9 ok, DBSN} in the two pad bytes.
__global__ void
k_dmr_r34_pick(const uint8_t* __restrict__ td98, const uint8_t* __restrict__ rel98, const uint8_t* __restrict__ wanted,
               const uint8_t* __restrict__ hard18, const uint8_t* __restrict__ soft18, const uint8_t* __restrict__ list24,
               const int32_t* __restrict__ list_n, int n, uint8_t* __restrict__ pool24, int32_t* __restrict__ pool_n,
               uint8_t* __restrict__ unconf18, uint8_t* __restrict__ conf18, uint8_t* __restrict__ conf_crc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    if (!wanted[i]) {
        pool_n[i] = 0;
        conf_crc[i] = 0;
        for (int b = 0; b < 18; b++) {
            unconf18[(size_t)i * 18 + b] = 0;
            conf18[(size_t)i * 18 + b] = 0;
        }
        return;
    }
    uint8_t nib[49], rhi[49], rlo[49];
    {
        uint8_t dei[98], rdei[98];
        for (int k = 0; k < 98; k++) {
            const int p = deinterleave98(k);
            dei[p] = td98[(size_t)i * 98 + k] & 3;
            rdei[p] = rel98[(size_t)i * 98 + k];
        }
        for (int t = 0; t < 49; t++) {
            nib[t] = (uint8_t)((dei[2 * t] << 2) | dei[2 * t + 1]);
            rhi[t] = rdei[2 * t];
            rlo[t] = rdei[2 * t + 1];
        }
    }
    uint8_t* pool = pool24 + (size_t)i * 34 * 24;
    int count = 0;
    const int nl = list_n[i] < 32 ? list_n[i] : 32;
    int n_hs = 0;
    for (int src = 0; src < 2 + nl; src++) {
        const uint8_t* b18 = src == 0 ? hard18 + (size_t)i * 18 : (src == 1 ? soft18 + (size_t)i * 18 : list24 + ((size_t)i * 32 + (src - 2)) * 24 + 4);
        bool dup = false;
        for (int c = 0; c < count && !dup; c++) {
            bool same = true;
            for (int b = 0; b < 18; b++) {
                same = same && pool[c * 24 + 4 + b] == b18[b];
            }
            dup = same;
        }
        if (!dup && count < 34) {
            int metric = 0, prev = 0;
            for (int t = 0; t < 49; t++) {
                int next = 0;
                if (t < 48) {
                    const int g = t >> 3, item = t & 7;
                    const uint32_t packed = ((uint32_t)b18[3 * g] << 16) | ((uint32_t)b18[3 * g + 1] << 8) | b18[3 * g + 2];
                    next = (int)((packed >> (21 - 3 * item)) & 7u);
                }
                const int x = c_r34_point_to_nibble[c_r34_fsm[prev * 8 + next]] ^ nib[t];
                metric += ((x >> 3) & 1) * rhi[t] + ((x >> 2) & 1) * rhi[t] + ((x >> 1) & 1) * rlo[t] + (x & 1) * rlo[t];
                prev = next;
            }
            // DBSN(7) | CRC9 ^ 0x1FF | 16 bytes: ComputeCrc9Bit over the 128 payload bits, then the DBSN
            uint32_t c9 = 0;
            for (int b = 16; b < 144; b++) {
                const int bit = (b18[b >> 3] >> (7 - (b & 7))) & 1;
                c9 = (((c9 >> 8) & 1) ^ (uint32_t)bit) ? ((c9 << 1) ^ 0x059u) : (c9 << 1);
            }
            for (int b = 0; b < 7; b++) {
                const int bit = (b18[0] >> (7 - b)) & 1;
                c9 = (((c9 >> 8) & 1) ^ (uint32_t)bit) ? ((c9 << 1) ^ 0x059u) : (c9 << 1);
            }
            const uint32_t ext = ((((uint32_t)b18[0] & 1u) << 8) | b18[1]) ^ 0x1FFu;
            uint8_t* o = pool + count * 24;
            o[0] = (uint8_t)metric;
            o[1] = (uint8_t)(metric >> 8);
            o[2] = (uint8_t)(metric >> 16);
            o[3] = (uint8_t)(metric >> 24);
            for (int b = 0; b < 18; b++) {
                o[4 + b] = b18[b];
            }
            o[22] = (((c9 & 0x1FFu) ^ 0x1FFu) == ext) ? 1 : 0;
            o[23] = (uint8_t)(b18[0] >> 1);
            count++;
        }
        if (src == 1) {
            n_hs = count;
        }
    }
    pool_n[i] = count;
    int best_u = -1, best_any = -1, best_crc = -1, mu = 0, ma = 0, mc = 0;
    for (int c = 0; c < count; c++) {
        const uint8_t* o = pool + c * 24;
        const int m = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24));
        if (c < n_hs && (best_u < 0 || m < mu)) {
            best_u = c;
            mu = m;
        }
        if (best_any < 0 || m < ma) {
            best_any = c;
            ma = m;
        }
        if (o[22] && (best_crc < 0 || m < mc)) {
            best_crc = c;
            mc = m;
        }
    }
    const int pick_c = best_crc >= 0 ? best_crc : best_any;
    for (int b = 0; b < 18; b++) {
        unconf18[(size_t)i * 18 + b] = best_u >= 0 ? pool[best_u * 24 + 4 + b] : 0;
        conf18[(size_t)i * 18 + b] = pick_c >= 0 ? pool[pick_c * 24 + 4 + b] : 0;
    }
    conf_crc[i] = best_crc >= 0 ? 1 : 0;
}

extern "C" hipError_t
ddn_dev_dmr_r34_pick(const uint8_t* td98, const uint8_t* rel98, const uint8_t* wanted, const uint8_t* hard18, const uint8_t* soft18,
                     const uint8_t* list24, const int32_t* list_n, int n, uint8_t* pool24, int32_t* pool_n, uint8_t* unconf18,
                     uint8_t* conf18, uint8_t* conf_crc, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_r34_pick, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, td98, rel98, wanted, hard18, soft18, list24, list_n, n,
                       pool24, pool_n, unconf18, conf18, conf_crc);
    return hipGetLastError();
}

// ---- P25 Phase 1 confirmed data: the rate 3/4 blocks' LLR list decoder (p25_mbf34_decode_soft_list, src/protocol/p25/phase1/
// p25p1_mbf34.c:129-213) ----------------------------------------------------------------------------------------------------------
// One wavefront per block, lane (p, r) = survivor r of state p (8 x 8 = P25_MBF34_MAX_CANDIDATES per state).  The reference inserts,
// per target state, every extension in front of the first strictly more expensive survivor - the 8 cheapest of up to 64 under the
// order (metric, arrival), arrival = previous state major, rank minor.  Here every lane finds its extension's place among the 64 by
// counting the extensions that come first (one LDS broadcast read of the 64 metrics per step, the 8 x 8 branch costs tabulated per
// step by the 64 lanes), and writes {metric, where it came from} into that place when it is below 8.  The 64 paths are then traced
// back lane by lane, packed from their first 48 states, and merged by one lane with the reference's rule: insert by metric, skip a
// path whose bytes are already in the list (:77-109).  Rare work (confirmed data units): no attempt to shorten the per-step chain.
namespace {
struct Mbf34Lds {
    int16_t d[196];
    uint32_t m[2][64];
    uint32_t ct[64];          // [ns][ps]
    uint8_t bp[49][64];       // survivor (ns, pos) of step t came from (ps << 3) | rank
    uint8_t path[64][18];
    uint32_t fin[64];
};

__global__ __launch_bounds__(64) void
k_p25_mbf34_list(const int16_t* __restrict__ llr, int n, int max_cand, uint8_t* __restrict__ cand24, int32_t* __restrict__ count,
                 const uint8_t* __restrict__ wanted) {
    __shared__ Mbf34Lds L;
    const int lane = threadIdx.x;
    const uint32_t INF = 0x3FFFFFFFu;
    for (long item = blockIdx.x; item < n; item += gridDim.x) {
        if (wanted && !wanted[item]) { // (block-uniform) an item nobody asked for: no candidates
            if (lane == 0) {
                count[item] = 0;
            }
            continue;
        }
        const int16_t* in = llr + (size_t)item * 196;
        // de-interleave: received dibit i carries de-interleaved dibit il[i] (13 + 12 + 12 + 12 pairs dealt with stride 4 pairs)
        for (int i = lane; i < 98; i += 64) {
            const int pair_rx = i >> 1;
            int ln4, k;
            if (pair_rx < 13) {
                ln4 = 0;
                k = pair_rx;
            } else {
                ln4 = 1 + (pair_rx - 13) / 12;
                k = (pair_rx - 13) % 12;
            }
            const int p = 2 * (ln4 + 4 * k) + (i & 1);
            L.d[2 * p] = in[2 * i];
            L.d[2 * p + 1] = in[2 * i + 1];
        }
        const int ps = lane >> 3, r = lane & 7;
        L.m[0][lane] = r == 0 ? (ps == 0 ? 0u : 1024u) : INF;
        __syncthreads();
        for (int t = 0; t < 49; t++) {
            const uint32_t* mb = L.m[t & 1];
            uint32_t* mn = L.m[(t + 1) & 1];
            { // branch cost of (q -> ns), q = lane >> 3, ns = lane & 7
                const int e = c_r34_point_to_nibble[c_r34_fsm[(lane >> 3) * 8 + (lane & 7)] & 15] & 15;
                uint32_t c = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int v = L.d[4 * t + b];
                    c += ((e >> (3 - b)) & 1) ? (v < 0 ? (uint32_t)(-v) : 0u) : (v > 0 ? (uint32_t)v : 0u);
                }
                L.ct[(lane & 7) * 8 + (lane >> 3)] = c;
            }
            mn[lane] = INF;
            __syncthreads();
            const uint32_t mine = mb[lane];
            if (mine != INF) {
                for (int ns = 0; ns < 8; ns++) {
                    const uint32_t val = mine + L.ct[ns * 8 + ps];
                    int pos = 0;
                    for (int q = 0; q < 8; q++) {
                        const uint32_t cq = L.ct[ns * 8 + q];
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const uint32_t o = mb[q * 8 + i];
                            const uint32_t ov = o + cq;
                            pos += (o != INF && (ov < val || (ov == val && (q * 8 + i) < lane))) ? 1 : 0;
                        }
                    }
                    if (pos < 8) {
                        mn[ns * 8 + pos] = val;
                        L.bp[t][ns * 8 + pos] = (uint8_t)lane;
                    }
                }
            }
            __syncthreads();
        }
        { // trace back this lane's path, pack its first 48 states
            const uint32_t fm = L.m[1][lane]; // 49 steps: the last write went to m[49 & 1]
            L.fin[lane] = fm;
            if (fm != INF) {
                uint8_t st[49];
                int cur = lane;
                for (int t = 48; t >= 0; t--) {
                    st[t] = (uint8_t)(cur >> 3);
                    cur = L.bp[t][cur];
                }
                for (int g = 0; g < 6; g++) {
                    uint32_t w = 0;
                    for (int k = 0; k < 8; k++) {
                        w = (w << 3) | st[8 * g + k];
                    }
                    L.path[lane][3 * g] = (uint8_t)(w >> 16);
                    L.path[lane][3 * g + 1] = (uint8_t)(w >> 8);
                    L.path[lane][3 * g + 2] = (uint8_t)w;
                }
            }
        }
        __syncthreads();
        if (lane == 0) { // p25_mbf34_collect_candidates(): arrival order = state major, rank minor
            uint8_t* out = cand24 + (size_t)item * 8 * 24;
            const int mx = max_cand > 8 ? 8 : max_cand;
            int cnt = 0;
            for (int a = 0; a < 64; a++) {
                const uint32_t me = L.fin[a];
                if (me == INF) {
                    continue;
                }
                bool dup = false;
                for (int i = 0; i < cnt && !dup; i++) {
                    bool same = true;
                    for (int b = 0; b < 18; b++) {
                        same = same && out[i * 24 + b] == L.path[a][b];
                    }
                    dup = same;
                }
                if (dup) {
                    continue;
                }
                int at = cnt;
                for (int i = 0; i < cnt; i++) {
                    if (me < *reinterpret_cast<const uint32_t*>(out + i * 24 + 20)) {
                        at = i;
                        break;
                    }
                }
                if (cnt < mx) {
                    cnt++;
                } else if (at >= mx) {
                    continue;
                }
                for (int i = cnt - 1; i > at; i--) {
                    for (int b = 0; b < 24; b++) {
                        out[i * 24 + b] = out[(i - 1) * 24 + b];
                    }
                }
                for (int b = 0; b < 18; b++) {
                    out[at * 24 + b] = L.path[a][b];
                }
                out[at * 24 + 18] = out[at * 24 + 19] = 0;
                *reinterpret_cast<uint32_t*>(out + at * 24 + 20) = me;
            }
            count[item] = cnt;
        }
        __syncthreads();
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_p25_mbf34_list(const int16_t* llr, int n, int max_cand, const uint8_t* wanted, uint8_t* cand24, int32_t* count, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    const int grid = n < 65536 ? n : 65536;
    hipLaunchKernelGGL(k_p25_mbf34_list, dim3((unsigned)grid), dim3(64), 0, st, llr, n, max_cand, cand24, count, wanted);
    return hipGetLastError();
}
