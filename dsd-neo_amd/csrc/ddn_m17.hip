// ddn_m17.hip - M17 link setup frames behind the receive loop's syncs (the consumer of SURVEY row a17, the libM17-style K = 5 decoder).
//
// processM17LSF() (src/protocol/m17/m17.c:1395-1408): the 184 payload symbols of an LSF frame as soft symbols ->
// soft_symbol_to_viterbi_cost() per bit (src/core/frames/dsd_dibit.c:1189-1242; llr_to_viterbi_cost :1150-1167) against the thresholds as
// they stand after the frame was read (static inside an M17 frame: the thresholds the sync left, ddn_fsk4_rx_set_sync_thresholds) ->
// de-randomised (the cost complemented where the randomiser bit is 1) -> de-interleaved (x = 45 i + 92 i^2 mod 368) :1187-1205 ->
// de-punctured with pattern P1, 0x7FFF where a bit was cut :1207-1217 -> viterbi_decode(488 costs) (src/core/util/dsd_misc.c:118-139:
// k_k5_m17, ddn_trellis.hip) -> bytes 1..30 = the LSF -> CRC16 (m17_crc16(), src/protocol/m17/m17_algorithms.c:19-35: poly 0x5935,
// init 0xFFFF) :1370-1393,1343-1368.
//   k_m17_lsf_cost    one wavefront per (channel, j): finds the channel's j-th LSF sync whose frame lies inside the records, computes
//                     its 368 costs (expf as the host's libm computes it: ddn_expf.h) and writes the 488 de-punctured ones
//   k_m17_lsf_finish  one lane per (channel, j): LSF bytes, CRC16, scattered to the sync's slot
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_expf.h"

namespace {

// M17 specification, "Randomizer": 46 bytes, most significant bit first (== m17_scramble[], src/protocol/m17/m17_tables.c:16-27)
__constant__ uint8_t k_m17_rand[46] = {0xD6, 0xB5, 0xE2, 0x30, 0x82, 0xFF, 0x84, 0x62, 0xBA, 0x4E, 0x96, 0x90, 0xD8, 0x98, 0xDD, 0x5D,
                                       0x0C, 0xC8, 0x52, 0x43, 0x91, 0x1D, 0xF8, 0x6E, 0x68, 0x2F, 0x35, 0xDA, 0x14, 0xEA, 0xCD, 0x76,
                                       0x19, 0x8D, 0xD5, 0x80, 0xD1, 0x33, 0x87, 0x13, 0x57, 0x18, 0x2D, 0x29, 0x78, 0xC3};

__device__ __forceinline__ float
min_sq2(float x, float a, float b) {
    const float da = x - a, db = x - b;
    const float d2a = da * da, d2b = db * db;
    return d2a < d2b ? d2a : d2b;
}

// soft_symbol_to_viterbi_cost(): thr = {center, umid, lmid, max, min}; bit 0 = the dibit's high bit
__device__ __forceinline__ uint32_t
m17_soft_cost(float symbol, const float* thr, int bit) {
    float center = thr[0], umid = thr[1], lmid = thr[2], max_val = thr[3], min_val = thr[4];
    if (!(min_val < lmid && lmid < center && center < umid && umid < max_val)) {
        float span = max_val - min_val;
        if (span < 1e-3f) {
            span = 2.0f;
        }
        const float half = span * 0.5f;
        min_val = center - half;
        max_val = center + half;
        lmid = center - (span / 6.0f);
        umid = center + (span / 6.0f);
    }
    const float n3 = 0.5f * (min_val + lmid), n1 = 0.5f * (lmid + center), p1 = 0.5f * (center + umid), p3 = 0.5f * (umid + max_val);
    float sigma = (max_val - min_val) / 6.0f;
    if (sigma < 1e-3f) {
        sigma = 1e-3f;
    }
    const float inv_2sigma2 = 0.5f / (sigma * sigma);
    float d0, d1;
    if ((bit & 1) == 0) {
        d0 = min_sq2(symbol, p1, p3);
        d1 = min_sq2(symbol, n1, n3);
    } else {
        d0 = min_sq2(symbol, n1, p1);
        d1 = min_sq2(symbol, n3, p3);
    }
    const float llr = (d1 - d0) * inv_2sigma2;
    if (llr >= 16.0f) {
        return 0u;
    }
    if (llr <= -16.0f) {
        return 65535u;
    }
    const float pr1 = 1.0f / (1.0f + ddn_expf(llr));
    long long q = __float2ll_rn(pr1 * 65535.0f); // lrintf
    q = q < 0 ? 0 : (q > 65535 ? 65535 : q);
    return (uint32_t)q;
}

__global__ __launch_bounds__(64) void
k_m17_lsf_cost(const uint8_t* __restrict__ rec, size_t stride, const int32_t* __restrict__ counts, const int32_t* __restrict__ sync_pos,
               const uint8_t* __restrict__ sync_pat, const int32_t* __restrict__ n_sync, const float* __restrict__ sync_thr, int max_syncs,
               int lmax, uint16_t* __restrict__ cost488, int32_t* __restrict__ slot_sync) {
    __shared__ uint16_t il[368]; // de-randomised costs in received order
    const int ch = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const size_t slot = (size_t)ch * lmax + j;
    // the channel's j-th LSF sync (pattern 4 / 5) with its 184 payload symbols inside this call's records
    int ns = n_sync[ch];
    ns = ns < max_syncs ? ns : max_syncs;
    const int cnt = counts[ch];
    int found = -1, seen = 0;
    for (int k0 = 0; k0 < ns && found < 0; k0 += 64) {
        const int k = k0 + lane;
        bool is = false;
        if (k < ns) {
            const int pat = sync_pat[(size_t)ch * max_syncs + k];
            is = (pat == 4 || pat == 5) && sync_pos[(size_t)ch * max_syncs + k] + 185 <= cnt;
        }
        const unsigned long long b = __ballot(is);
        const int nb = __popcll(b);
        if (seen + nb > j) {
            unsigned long long m = b;
            for (int q = 0; q < j - seen; q++) {
                m &= m - 1;
            }
            found = k0 + __ffsll((long long)m) - 1;
        }
        seen += nb;
    }
    if (lane == 0) {
        slot_sync[slot] = found;
    }
    uint16_t* out = cost488 + slot * 488;
    if (found < 0) {
        for (int i = lane; i < 488; i += 64) {
            out[i] = 0;
        }
        return;
    }
    const size_t so = (size_t)ch * max_syncs + found;
    const int pos = sync_pos[so];
    const float* thr = sync_thr + so * 5;
    const uint8_t* r0 = rec + ((size_t)ch * stride + (size_t)pos + 1) * 10;
    for (int i = lane; i < 368; i += 64) {
        const uint8_t* r = r0 + (size_t)(i >> 1) * 10;
        const uint32_t xb = (uint32_t)((const uint16_t*)r)[3] | ((uint32_t)((const uint16_t*)r)[4] << 16);
        const uint32_t c = m17_soft_cost(__uint_as_float(xb), thr, i & 1);
        const int rb = (k_m17_rand[i >> 3] >> (7 - (i & 7))) & 1;
        il[i] = (uint16_t)(rb ? (0xFFFFu - c) : c);
    }
    __syncthreads();
    // P1: 61 entries, every fourth of {1, 1, 0, 1} cut, except that the pattern ends 1, 1; kept bit number k reads the de-interleaved
    // stream: bits[k] = il[(45 k + 92 k^2) mod 368]
    for (int i = lane; i < 488; i += 64) {
        const int g = i / 61, jj = i - g * 61;
        const bool keep = (jj == 60) || ((jj & 3) != 2);
        // kept entries before position jj inside a group: jj - (cut ones among 0 .. jj - 1) = jj - floor((jj + 1) / 4); 46 kept per group
        const int k = g * 46 + jj - ((jj + 1) >> 2);
        uint16_t v = 0x7FFFu;
        if (keep) {
            const int x = (45 * k + 92 * k * k) % 368;
            v = il[x];
        }
        out[i] = v;
    }
}

__global__ void
k_m17_lsf_finish(const uint8_t* __restrict__ dec, int dec_stride, const uint32_t* __restrict__ cost, const int32_t* __restrict__ slot_sync,
                 int n_channels, int lmax, int max_syncs, uint8_t* __restrict__ lsf30, uint8_t* __restrict__ status,
                 uint32_t* __restrict__ path_cost) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_channels * lmax) {
        return;
    }
    const int k = slot_sync[slot];
    if (k < 0) {
        return;
    }
    const int ch = slot / lmax;
    const size_t so = (size_t)ch * max_syncs + k;
    const uint8_t* by = dec + (size_t)slot * dec_stride + 1; // viterbi_decode()'s bytes 1 .. 30
    uint32_t crc = 0xFFFFu;
    for (int i = 0; i < 30; i++) {
        const uint8_t b = by[i];
        lsf30[so * 30 + i] = b;
        if (i < 28) {
            crc ^= (uint32_t)b << 8;
            for (int q = 0; q < 8; q++) {
                crc <<= 1;
                if (crc & 0x10000u) {
                    crc = (crc ^ 0x5935u) & 0xFFFFu;
                }
            }
        }
    }
    const uint32_t ext = ((uint32_t)by[28] << 8) | by[29];
    status[so] = (crc & 0xFFFFu) == ext ? 2 : 1;
    if (path_cost) {
        path_cost[so] = cost[slot];
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_m17_lsf_cost(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos, const uint8_t* sync_pat,
                     const int32_t* n_sync, const float* sync_thr, int n_channels, int max_syncs, int lmax, uint16_t* cost488,
                     int32_t* slot_sync, hipStream_t st) {
    hipLaunchKernelGGL(k_m17_lsf_cost, dim3((unsigned)n_channels, (unsigned)lmax), dim3(64), 0, st, rec, stride, counts, sync_pos, sync_pat,
                       n_sync, sync_thr, max_syncs, lmax, cost488, slot_sync);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_m17_lsf_finish(const uint8_t* dec, int dec_stride, const uint32_t* cost, const int32_t* slot_sync, int n_channels, int lmax,
                       int max_syncs, uint8_t* lsf30, uint8_t* status, uint32_t* path_cost, hipStream_t st) {
    const int n = n_channels * lmax;
    hipLaunchKernelGGL(k_m17_lsf_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, dec, dec_stride, cost, slot_sync, n_channels,
                       lmax, max_syncs, lsf30, status, path_cost);
    return hipGetLastError();
}
