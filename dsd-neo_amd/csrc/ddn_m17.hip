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
//
// processM17STR() (:1122-1176): the 184 payload dibits of a stream frame as hard bits -> de-randomised -> de-interleaved -> 96 LICH
// bits = four Golay(24,12) words (m17_lich_decode_bits, m17_algorithms.c:598-612 over Golay_24_12_decode, src/fec/fec.c:656-690) ->
// 40 LSF bits + 3-bit chunk counter (m17_lich_parse_content :562-583); when all four words decode and the counter is < 6:
// M17prepareStream() :1039-1120 = the other 272 bits de-punctured with P2 (11 of 12 kept, the cut bit reads 0) -> symbol values
// bit << 1 -> CNXDNConvolution over 148 steps, 144 bits chained back (k_k5_nxdn, ddn_trellis.hip) -> frame number + 16 payload bytes.
//   k_m17_str_bits    one wavefront per (channel, j): the channel's j-th stream sync with a complete frame -> LICH + the 296 symbols
//   k_m17_str_finish  one lane per (channel, j): scatter to the sync's slot
//   k_m17_lich        one lane per channel: the LSF reassembled from six LICH chunks in the order of the syncs (dispatch_m17.c:39,
//                     m17.c:250, M17finalizeLICH: CRC16 over the reassembled 30 bytes), with the decoded LSF frames and EOT markers
//                     in between as the reference applies them
#include "ddn_tables_ambe.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_expf.h"
#include "ddn_fec3.h"
#include "ddn_tables_fec3.h"

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
// the channel's j-th sync of one of two patterns whose 184 payload symbols lie inside the call's records (whole wavefront; -1: none)
__device__ __forceinline__ int
m17_find_sync(const int32_t* sync_pos, const uint8_t* sync_pat, int ns, int cnt, int j, int pat_a, int pat_b, int lane) {
    int found = -1, seen = 0;
    for (int k0 = 0; k0 < ns && found < 0; k0 += 64) {
        const int k = k0 + lane;
        bool is = false;
        if (k < ns) {
            const int pat = sync_pat[k];
            is = (pat == pat_a || pat == pat_b) && sync_pos[k] + 185 <= cnt;
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
    return found;
}

__global__ __launch_bounds__(64) void
k_m17_str_bits(const uint8_t* __restrict__ rec, size_t stride, const int32_t* __restrict__ counts, const int32_t* __restrict__ sync_pos,
               const uint8_t* __restrict__ sync_pat, const int32_t* __restrict__ n_sync, int max_syncs, int lmax,
               const DdnFec3Tables* __restrict__ T, uint8_t* __restrict__ sym296, int32_t* __restrict__ slot_sync,
               uint8_t* __restrict__ slot_lich6, uint8_t* __restrict__ slot_cnt, uint8_t* __restrict__ slot_ok) {
    __shared__ uint8_t bits[368]; // de-randomised, de-interleaved
    __shared__ uint32_t word[4];
    __shared__ int werr[4];
    const int ch = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const size_t slot = (size_t)ch * lmax + j;
    int ns = n_sync[ch];
    ns = ns < max_syncs ? ns : max_syncs;
    const int found = m17_find_sync(sync_pos + (size_t)ch * max_syncs, sync_pat + (size_t)ch * max_syncs, ns, counts[ch], j, 8, 9, lane);
    if (lane == 0) {
        slot_sync[slot] = found;
        slot_ok[slot] = 0;
    }
    if (found < 0) {
        return;
    }
    const int pos = sync_pos[(size_t)ch * max_syncs + found];
    const uint8_t* r0 = rec + ((size_t)ch * stride + (size_t)pos + 1) * 10;
    for (int i = lane; i < 368; i += 64) {
        const int x = (45 * i + 92 * i * i) % 368; // bits[i] = rnd[x] ^ rand(x)
        const int d = r0[(size_t)(x >> 1) * 10] & 3;
        const int b = (x & 1) ? (d & 1) : (d >> 1);
        bits[i] = (uint8_t)((b ^ ((k_m17_rand[x >> 3] >> (7 - (x & 7))) & 1)) & 1);
    }
    __syncthreads();
    if (lane < 4) { // Golay_24_12_decode on rx[0 .. 23] (bit j of the word = rx[j]), fec.c:656-690
        uint32_t w = 0;
        for (int q = 0; q < 24; q++) {
            w |= (uint32_t)bits[24 * lane + q] << q;
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            s |= (__popc(w & ddn_golay_24_12_H[i]) & 1) << (11 - i);
        }
        bool ok = true;
        if (s > 0) {
            int k = 0;
            for (; k < 3; k++) {
                const uint8_t p = T->g2412[s][k];
                if (p == 0xFF) {
                    break;
                }
                w ^= 1u << p;
            }
            ok = k != 0;
        }
        word[lane] = w;
        werr[lane] = ok ? 0 : 1;
    }
    __syncthreads();
    // content bit 12 b + q = corrected word b, bit q (the first twelve of each word)
    const int c40 = (word[3] >> 4) & 1, c41 = (word[3] >> 5) & 1, c42 = (word[3] >> 6) & 1;
    const int cnt = (c40 << 2) | (c41 << 1) | c42;
    const bool err = (werr[0] | werr[1] | werr[2] | werr[3]) != 0 || cnt >= 6;
    if (lane < 6) {
        uint32_t by = 0;
        for (int q = 0; q < 8; q++) {
            const int i = 8 * lane + q;
            by |= ((word[i / 12] >> (i % 12)) & 1u) << (7 - q);
        }
        slot_lich6[slot * 6 + lane] = (uint8_t)by;
    }
    if (lane == 0) {
        slot_cnt[slot] = (uint8_t)cnt;
        slot_ok[slot] = err ? 1 : 2;
    }
    uint8_t* out = sym296 + slot * 296;
    for (int i = lane; i < 296; i += 64) { // P2: groups of 11 kept + 1 cut; 272 bits fill 24 groups and 8 of the 25th
        const int g = i / 12, q = i - g * 12;
        const int x = g * 11 + q;
        const int b = (q < 11 && x < 272 && !err) ? bits[96 + x] : 0;
        out[i] = (uint8_t)(b << 1);
    }
}

__global__ void
k_m17_str_finish(const uint8_t* __restrict__ dec, int dec_stride, const int32_t* __restrict__ slot_sync, const uint8_t* __restrict__ slot_lich6,
                 const uint8_t* __restrict__ slot_cnt, const uint8_t* __restrict__ slot_ok, int n_channels, int lmax, int max_syncs,
                 uint8_t* __restrict__ lich6, uint8_t* __restrict__ lich_cnt, uint8_t* __restrict__ fn_payload18, uint8_t* __restrict__ status) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_channels * lmax) {
        return;
    }
    const int k = slot_sync[slot];
    if (k < 0) {
        return;
    }
    const size_t so = (size_t)(slot / lmax) * max_syncs + k;
    const int ok = slot_ok[slot];
    for (int i = 0; i < 6; i++) {
        lich6[so * 6 + i] = slot_lich6[(size_t)slot * 6 + i];
    }
    lich_cnt[so] = slot_cnt[slot];
    for (int i = 0; i < 18; i++) {
        fn_payload18[so * 18 + i] = ok == 2 ? dec[(size_t)slot * dec_stride + i] : 0;
    }
    status[so] = (uint8_t)ok;
}

// One lane per channel: walks the channel's syncs in order.  asm30 [B][32] is the carried assembly buffer (30 bytes + 2 spare).
__global__ void
k_m17_lich(const uint8_t* __restrict__ sync_pat, const int32_t* __restrict__ n_sync, int n_channels, int max_syncs,
           const uint8_t* __restrict__ lsf30, const uint8_t* __restrict__ lsf_status, const uint8_t* __restrict__ lich6,
           const uint8_t* __restrict__ lich_cnt, const uint8_t* __restrict__ str_status, uint8_t* __restrict__ asm30,
           uint8_t* __restrict__ lich_lsf30, uint8_t* __restrict__ lich_status) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    uint8_t* a = asm30 + (size_t)ch * 32;
    int ns = n_sync[ch];
    ns = ns < max_syncs ? ns : max_syncs;
    for (int k = 0; k < ns; k++) {
        const size_t so = (size_t)ch * max_syncs + k;
        const int pat = sync_pat[so];
        lich_status[so] = 0;
        if (pat == 2 || pat == 3) { // EOT: state->m17_lsf cleared (dispatch_m17.c:39)
            for (int i = 0; i < 30; i++) {
                a[i] = 0;
            }
        } else if ((pat == 4 || pat == 5) && lsf_status && lsf_status[so] != 0) { // m17_decode_lsf_soft_bits: m17_lsf = the decoded LSF
            for (int i = 0; i < 30; i++) {
                a[i] = lsf30[so * 30 + i];
            }
        } else if ((pat == 8 || pat == 9) && str_status[so] == 2) {
            const int cnt = lich_cnt[so];
            for (int i = 0; i < 5; i++) { // 40 bits of chunk cnt
                a[5 * cnt + i] = lich6[so * 6 + i];
            }
            if (cnt == 5) { // M17finalizeLICH: CRC over the reassembled LSF, then the buffer is cleared (:250)
                uint32_t crc = 0xFFFFu;
                for (int i = 0; i < 30; i++) {
                    lich_lsf30[so * 30 + i] = a[i];
                    if (i < 28) {
                        crc ^= (uint32_t)a[i] << 8;
                        for (int q = 0; q < 8; q++) {
                            crc <<= 1;
                            if (crc & 0x10000u) {
                                crc = (crc ^ 0x5935u) & 0xFFFFu;
                            }
                        }
                    }
                }
                lich_status[so] = (crc & 0xFFFFu) == (((uint32_t)a[28] << 8) | a[29]) ? 2 : 1;
                for (int i = 0; i < 30; i++) {
                    a[i] = 0;
                }
            }
        }
    }
}
// ---- Yaesu System Fusion: the frame information channel (ysf_conv_fich(), src/protocol/ysf/ysf.c:357-424) - the K = 5 decoder's second
// consumer.  k_ysf_fich_cost: one wavefront per (channel, j-th sync with its 100 FICH dibits inside the records): dibit de-interleave
// (20 x 5) -> hard costs 0 / 65535 per bit (ysf_dibit_to_soft_costs, ysf_frame.c:74-80) -> 200 costs for viterbi_decode_punctured()
// with a pattern of all ones.  k_ysf_fich_finish: bits 8 .. 103 of the decoded bytes -> four Golay(24,12) words -> CRC16 (ysf_crc16,
// ysf.c:229-242) -> the 32 FICH bits packed into four bytes, status 1 good / 2 a Golay word failed / 3 CRC failed.
__global__ __launch_bounds__(64) void
k_ysf_fich_cost(const uint8_t* __restrict__ rec, size_t stride, const int32_t* __restrict__ counts, const int32_t* __restrict__ sync_pos,
                const int32_t* __restrict__ n_sync, int max_syncs, int lmax, uint16_t* __restrict__ cost200, int32_t* __restrict__ slot_sync) {
    const int ch = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const size_t slot = (size_t)ch * lmax + j;
    int ns = n_sync[ch];
    ns = ns < max_syncs ? ns : max_syncs;
    const int cnt = counts[ch];
    // the j-th sync whose FICH is complete
    int found = -1, seen = 0;
    for (int k0 = 0; k0 < ns && found < 0; k0 += 64) {
        const int k = k0 + lane;
        const bool is = k < ns && sync_pos[(size_t)ch * max_syncs + k] + 101 <= cnt;
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
    uint16_t* out = cost200 + slot * 200;
    if (found < 0) {
        for (int i = lane; i < 200; i += 64) {
            out[i] = 0;
        }
        return;
    }
    const int pos = sync_pos[(size_t)ch * max_syncs + found];
    const uint8_t* r0 = rec + ((size_t)ch * stride + (size_t)pos + 1) * 10;
    for (int q = lane; q < 100; q += 64) { // buf[jj + 5 i] = input[i + 20 jj]
        const int i = q / 5, jj = q - 5 * i;
        const int d = r0[(size_t)(i + 20 * jj) * 10] & 3;
        out[2 * q] = (d & 2) ? 0xFFFFu : 0u;
        out[2 * q + 1] = (d & 1) ? 0xFFFFu : 0u;
    }
}

__global__ void
k_ysf_fich_finish(const uint8_t* __restrict__ dec, int dec_stride, const uint32_t* __restrict__ cost, const int32_t* __restrict__ slot_sync,
                  int n_channels, int lmax, int max_syncs, const DdnFec3Tables* __restrict__ T, uint8_t* __restrict__ fich4,
                  uint8_t* __restrict__ status, uint32_t* __restrict__ v_error) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_channels * lmax) {
        return;
    }
    const int k = slot_sync[slot];
    if (k < 0) {
        return;
    }
    const size_t so = (size_t)(slot / lmax) * max_syncs + k;
    const uint8_t* by = dec + (size_t)slot * dec_stride;
    bool bad = false;
    uint64_t fich = 0; // 48 bits, first bit = bit 47
    for (int w4 = 0; w4 < 4; w4++) {
        uint32_t w = 0;
        for (int q = 0; q < 24; q++) { // trellis bit 24 w4 + q = decoded bit 8 + 24 w4 + q (MSB first in the bytes)
            const int b = 8 + 24 * w4 + q;
            w |= (uint32_t)((by[b >> 3] >> (7 - (b & 7))) & 1) << q;
        }
        int s = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            s |= (__popc(w & ddn_golay_24_12_H[i]) & 1) << (11 - i);
        }
        if (s > 0) { // Golay_24_12_decode, src/fec/fec.c:656-690
            int kk = 0;
            for (; kk < 3; kk++) {
                const uint8_t p = T->g2412[s][kk];
                if (p == 0xFF) {
                    break;
                }
                w ^= 1u << p;
            }
            bad = bad || kk == 0;
        }
        for (int q = 0; q < 12; q++) {
            fich = (fich << 1) | ((w >> q) & 1u);
        }
    }
    uint32_t crc = 0; // ysf_crc16 over the 48 bits: a good frame leaves 0
    for (int i = 0; i < 48; i++) {
        const uint32_t bit = (uint32_t)((fich >> (47 - i)) & 1u);
        crc = ((crc << 1) | bit) & 0x1ffffu;
        if (crc & 0x10000u) {
            crc = (crc & 0xffffu) ^ 0x1021u;
        }
    }
    crc ^= 0xffffu;
    for (int i = 0; i < 4; i++) {
        fich4[so * 4 + i] = (uint8_t)((fich >> (40 - 8 * i)) & 0xFF);
    }
    status[so] = (crc & 0xffffu) != 0 ? 3 : (bad ? 2 : 1); // (the reference's err: -2 wins over -1)
    if (v_error) {
        v_error[so] = cost[slot];
    }
}
} // namespace

// ---- Yaesu System Fusion: the payload behind the FICH (round 5) ---------------------------------------------------------------------
// ysf_dispatch_payload() (src/protocol/ysf/ysf.c:908-922) by the type the frame is read as: FI = 1 & DT = 2 -> V/D mode 2 (five
// sub-frames of 20 data + 52 voice dibits: ysf_handle_vd_type2 :724-774), FI = 1 & DT = 0 -> V/D mode 1 (5 x 36 data + 36 voice:
// :667-685), DT = 1 or FI = 0 / 2 -> full-rate data (10 x 36 dibits, two data-channel blocks in turn: :844-864).
//   k_ysf_plan           one thread per channel, the syncs of the decode list in order: a frame whose FICH failed is read as the last
//                        good frame's DT / FI (ysf_parse_fich's two statics, :553-555 - kept per channel in last2[][2] across calls);
//                        the frames whose 360 payload dibits lie inside the records get a slot of the dense decode list
//   k_ysf_payload_costs  one wavefront per slot: V/D2 voice = ysf_read_type2_vech_bits + ysf_build_type2_ambe (:687-722: 26 x 4 bit
//                        de-interleave, PN9 de-whitening, 27 majority votes + 22 plain bits = ambe_d[49], errs2 = bit 103); the data
//                        channel(s): dibit de-interleave 20 x 5 (DCH2) / 20 x 9 (DCH) -> hard costs for the K = 5 decoder
//   k_ysf_dch_finish     ysf_conv_dch2 / ysf_conv_dch (:245-355) behind the decoder: bits 8 .. of the decoded bytes, ysf_crc16 over
//                        all of them (0 = good), PN9 de-whitening of the 80 / 160 data bits (dsd_ysf_dewhiten_bits, ysf_frame.c:59-73)
__device__ inline int
ysf_pn9_next(unsigned& l) {
    const int bit = (int)(l & 1u);
    const unsigned fb = ((l >> 4) ^ l) & 1u;
    l = (l >> 1) | (fb << 8);
    return bit;
}

__global__ void
k_ysf_plan(const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, const int32_t* __restrict__ counts, int n_channels,
           int max_syncs, int lmax, const uint8_t* __restrict__ fich4, const uint8_t* __restrict__ fich_status, uint8_t* __restrict__ last2,
           uint8_t* __restrict__ info, int32_t* __restrict__ slot_sync) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    int ns = n_sync[ch];
    ns = ns < max_syncs ? ns : max_syncs;
    const int cnt = counts[ch];
    int dt = last2[2 * ch], fi = last2[2 * ch + 1], j = 0;
    for (int k = 0; k < ns; k++) {
        const size_t so = (size_t)ch * max_syncs + k;
        const int st = fich_status[so];
        if (st == 0) { // no FICH behind this sync inside the records: not a frame of this call
            info[2 * so] = 0;
            info[2 * so + 1] = 0;
            continue;
        }
        if (st == 1) {
            dt = fich4[4 * so + 2] & 3;
            fi = fich4[4 * so] >> 6;
        }
        int kind = 0;
        kind |= (fi == 1 && dt == 0) ? 1 : 0;
        kind |= (fi == 1 && dt == 2) ? 2 : 0;
        kind |= (fi == 1 && dt == 3) ? 4 : 0;
        kind |= (dt == 1 || fi == 0 || fi == 2) ? 8 : 0;
        const bool complete = sync_pos[so] + 461 <= cnt;
        int flags = fi | (dt << 2) | (st != 1 ? 16 : 0) | 32;
        // full-rate voice with FT = 1, FN = 0 carries a data-channel block and two voice slots (ysf_is_full_rate_csd3, ysf.c:776-779;
        // the fields of a failed FICH read 9: never CSD3)
        if (kind == 4 && st == 1 && (fich4[4 * so + 1] & 7) == 1 && ((fich4[4 * so + 1] >> 3) & 7) == 0) {
            flags |= 128;
        }
        if (complete && kind != 0) {
            if (j < lmax) {
                slot_sync[(size_t)ch * lmax + j] = k;
                j++;
            } else {
                flags |= 64; // (more frames than a row can hold 480 symbols apart: left undecoded, said so)
            }
        }
        info[2 * so] = (uint8_t)((complete && !(flags & 64)) ? kind : 0);
        info[2 * so + 1] = (uint8_t)flags;
    }
    last2[2 * ch] = (uint8_t)dt;
    last2[2 * ch + 1] = (uint8_t)fi;
}

__global__ __launch_bounds__(64) void
k_ysf_payload_costs(const uint8_t* __restrict__ rec, size_t stride, const int32_t* __restrict__ sync_pos, int max_syncs, int lmax,
                    const uint8_t* __restrict__ info, const int32_t* __restrict__ slot_sync, uint16_t* __restrict__ cost200,
                    uint16_t* __restrict__ cost360, uint8_t* __restrict__ ambe49, uint8_t* __restrict__ errs2,
                    uint8_t* __restrict__ want200, uint8_t* __restrict__ want360, uint8_t* __restrict__ frames, uint8_t* __restrict__ n_frames) {
    const int ch = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const size_t slot = (size_t)ch * lmax + j;
    const int k = slot_sync[slot];
    if (k < 0) {
        return; // (the cost arrays were cleared as a whole)
    }
    const size_t so = (size_t)ch * max_syncs + k;
    const int kind = info[2 * so];
    const uint8_t* r0 = rec + ((size_t)ch * stride + (size_t)sync_pos[so] + 101) * 10;
    auto dib = [&](int x) { return (int)(r0[(size_t)x * 10] & 3); };
    if (kind & 2) {
        __shared__ uint8_t pn[104], v[5][104];
        if (lane == 0) {
            unsigned l = 0x1C9;
            for (int i = 0; i < 104; i++) {
                pn[i] = (uint8_t)ysf_pn9_next(l);
            }
        }
        __syncthreads();
        for (int t = lane; t < 5 * 104; t += 64) {
            const int sf = t / 104, kb = t - 104 * sf;            // serial bit kb of sub-frame sf: dibit kb / 2, high bit first
            const int d = dib(72 * sf + 20 + (kb >> 1));
            const int dest = (kb & 3) * 26 + (kb >> 2);
            v[sf][dest] = (uint8_t)((((kb & 1) ? d : (d >> 1)) & 1) ^ pn[dest]);
        }
        __syncthreads();
        for (int t = lane; t < 5 * 49; t += 64) {
            const int sf = t / 49, b = t - 49 * sf;
            int o;
            if (b < 27) {
                o = (v[sf][3 * b] + v[sf][3 * b + 1] + v[sf][3 * b + 2]) >= 2 ? 1 : 0;
            } else {
                o = v[sf][81 + (b - 27)];
            }
            ambe49[(so * 5 + sf) * 49 + b] = (uint8_t)o;
        }
        if (lane < 5) {
            errs2[so * 5 + lane] = v[lane][103];
        }
        uint16_t* out = cost200 + slot * 200;
        if (lane == 0) {
            want200[slot] = 1;
        }
        for (int q = lane; q < 100; q += 64) { // buf[jj + 5 i] = input[i + 20 jj]; input[x] = data dibit x % 20 of sub-frame x / 20
            const int i = q / 5, jj = q - 5 * i, x = i + 20 * jj;
            const int d = dib(72 * (x / 20) + x % 20);
            out[2 * q] = (d & 2) ? 0xFFFFu : 0u;
            out[2 * q + 1] = (d & 1) ? 0xFFFFu : 0u;
        }
    } else if (kind & 1) {
        uint16_t* out = cost360 + slot * 2 * 360;
        if (lane == 0) {
            want360[slot * 2] = 1;
        }
        for (int q = lane; q < 180; q += 64) { // buf[jj + 9 i] = input[i + 20 jj]; input[x] = data dibit x % 36 of sub-frame x / 36
            const int i = q / 9, jj = q - 9 * i, x = i + 20 * jj;
            const int d = dib(72 * (x / 36) + x % 36);
            out[2 * q] = (d & 2) ? 0xFFFFu : 0u;
            out[2 * q + 1] = (d & 1) ? 0xFFFFu : 0u;
        }
    }
    // the voice frames of V/D mode 1 (ysf_ehr: four AMBE 3600x2450 frames through the 36-dibit schedule) and of full-rate frames
    // (dsd_ysf_unpack_full_rate_imbe: five IMBE 7200x4400 frames, two behind a CSD3 data block), as processMbeFrame() gets them
    if (kind == 1) {
        static const uint8_t __attribute__((address_space(4))) amap[36][4] = DDN_AMBE2450_MAP_INIT;
        for (int t = lane; t < 4 * 36; t += 64) {
            const int sf = t / 36, i = t - 36 * sf;
            const int d = dib(72 * sf + 36 + i);
            uint8_t* f = frames + (so * 5 + sf) * 184;
            f[amap[i][0] * 24 + amap[i][1]] = (uint8_t)(d >> 1);
            f[amap[i][2] * 24 + amap[i][3]] = (uint8_t)(d & 1);
        }
        if (lane == 0) {
            n_frames[so] = 4;
        }
    } else if (kind == 4) {
        const bool csd3 = (info[2 * so + 1] & 128) != 0;
        const int nf = csd3 ? 2 : 5, off = csd3 ? 216 : 0;
        for (int t = lane; t < nf * 144; t += 64) {
            const int i = t / 144, k = t - 144 * i;            // k-th bit of the de-interleaved stream of slot i
            const int r = k / 24, c = k - 24 * r;
            const int src = 12 * (c >> 1) + ((c & 1) ? ((r ^ 1) + 6) : r);
            const int d = dib(off + 72 * i + (src >> 1));
            const int bit = ((src & 1) ? d : (d >> 1)) & 1;
            int n, m;                                           // rows 0-3: 23 bits, 4-6: 15, 7: seven; highest column first
            if (k < 92) {
                n = k / 23, m = 22 - (k - 23 * n);
            } else if (k < 137) {
                n = 4 + (k - 92) / 15, m = 14 - ((k - 92) % 15);
            } else {
                n = 7, m = 6 - (k - 137);
            }
            frames[(so * 5 + i) * 184 + n * 23 + m] = (uint8_t)bit;
        }
        if (lane == 0) {
            n_frames[so] = (uint8_t)nf;
        }
        if (csd3) { // the 180 data dibits as they lie: buf[jj + 9 i] = input[i + 20 jj]
            uint16_t* out = cost360 + slot * 2 * 360;
            if (lane == 0) {
                want360[slot * 2] = 1;
            }
            for (int q = lane; q < 180; q += 64) {
                const int i = q / 9, jj = q - 9 * i;
                const int d = dib(i + 20 * jj);
                out[2 * q] = (d & 2) ? 0xFFFFu : 0u;
                out[2 * q + 1] = (d & 1) ? 0xFFFFu : 0u;
            }
        }
    }
    if (kind == 8) {
        for (int b = 0; b < 2; b++) {
            uint16_t* out = cost360 + (slot * 2 + b) * 360;
            if (lane == 0) {
                want360[slot * 2 + b] = 1;
            }
            for (int q = lane; q < 180; q += 64) { // input_b[x] = dibit x % 36 of chunk 2 (x / 36) + b
                const int i = q / 9, jj = q - 9 * i, x = i + 20 * jj;
                const int d = dib(36 * (2 * (x / 36) + b) + x % 36);
                out[2 * q] = (d & 2) ? 0xFFFFu : 0u;
                out[2 * q + 1] = (d & 1) ? 0xFFFFu : 0u;
            }
        }
    }
}

__global__ void
k_ysf_dch_finish(const uint8_t* __restrict__ decA, const uint32_t* __restrict__ pcA, const uint8_t* __restrict__ decB,
                 const uint32_t* __restrict__ pcB, const int32_t* __restrict__ slot_sync, const uint8_t* __restrict__ info, int n_channels,
                 int lmax, int max_syncs, uint8_t* __restrict__ dch40, uint8_t* __restrict__ dch_status, uint32_t* __restrict__ dch_cost) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_channels * lmax * 2) {
        return;
    }
    const int slot = t >> 1, blk = t & 1;
    const int k = slot_sync[slot];
    if (k < 0) {
        return;
    }
    const size_t so = (size_t)(slot / lmax) * max_syncs + k;
    const int kind = info[2 * so];
    const bool vd2 = (kind & 2) != 0;
    if ((blk == 1 && kind != 8) || kind == 0 || (kind == 4 && !(info[2 * so + 1] & 128))) {
        return; // (one data-channel block per V/D frame, none behind full-rate voice but for CSD3)
    }
    const uint8_t* by = vd2 ? decA + (size_t)slot * 16 : decB + (size_t)(slot * 2 + blk) * 32;
    const int nbits = vd2 ? 96 : 176;
    uint32_t crc = 0;
    for (int i = 0; i < nbits; i++) {
        const int b = 8 + i;
        const uint32_t bit = (by[b >> 3] >> (7 - (b & 7))) & 1u;
        crc = ((crc << 1) | bit) & 0x1ffff;
        if (crc & 0x10000) {
            crc = (crc & 0xffff) ^ 0x1021u;
        }
    }
    crc = (crc ^ 0xffff) & 0xffff;
    unsigned l = 0x1C9;
    uint8_t* out = dch40 + (so * 2 + blk) * 20;
    for (int i = 0; i < (nbits - 16) / 8; i++) {
        int o = 0;
        for (int q = 0; q < 8; q++) {
            const int b = 8 + 8 * i + q;
            o = (o << 1) | ((int)((by[b >> 3] >> (7 - (b & 7))) & 1u) ^ ysf_pn9_next(l));
        }
        out[i] = (uint8_t)o;
    }
    dch_status[so * 2 + blk] = crc == 0 ? 1 : 3;
    dch_cost[so * 2 + blk] = vd2 ? pcA[slot] : pcB[slot * 2 + blk];
}

// The voice of a call filed by talk path (= channel), the frames of the channel in stream order, five positions per frame:
//   mode 0, the AMBE 3600x2450 path: a V/D mode 2 frame -> its five sub-frames' ambe_d + result rows {0, 0, 0, errs2, errs2}
//     (ysf_handle_vd_type2, ysf.c:745-752); a V/D mode 1 frame -> the four frames ysf_ehr() decodes, as the frame FEC left them
//     (bits_fd / res_fd by slot x 5), the fifth position skipped;
//   mode 1, the IMBE 7200x4400 path: a full-rate voice frame -> its five (CSD3: two) frames from the frame FEC, the rest skipped.
// skip flags behind the last frame too; v_n[c] = frames filed, v_slot[c][j] = the sync slot frame j came from.  One wavefront per channel.
__global__ __launch_bounds__(64) void
k_ysf_voice_file(const int32_t* __restrict__ n_sync, int max_syncs, const uint8_t* __restrict__ info, const uint8_t* __restrict__ ambe49,
                 const uint8_t* __restrict__ errs2, const uint8_t* __restrict__ bits_fd, const int32_t* __restrict__ res_fd,
                 const uint8_t* __restrict__ n_frames, int mode, int vf, uint8_t* __restrict__ bits, int32_t* __restrict__ res,
                 uint8_t* __restrict__ skip, int32_t* __restrict__ v_n, int32_t* __restrict__ v_slot) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int nb = mode ? 88 : 49;
    int ns = n_sync[c];
    ns = ns < max_syncs ? ns : max_syncs;
    int j = 0;
    for (int k0 = 0; k0 < ns; k0 += 64) {
        const int k = k0 + lane;
        const int kd = k < ns ? info[2 * ((size_t)c * max_syncs + k)] : 0;
        const bool is = mode ? kd == 4 : (kd == 2 || kd == 1);
        unsigned long long b = __ballot(is);
        while (b && j < vf) {
            const int kk = k0 + __ffsll((long long)b) - 1;
            b &= b - 1;
            const size_t so = (size_t)c * max_syncs + kk, d0 = ((size_t)c * vf + j) * 5;
            const bool direct = !mode && info[2 * so] == 2;
            const int nf = direct ? 5 : n_frames[so];
            for (int t = lane; t < 5 * nb; t += 64) {
                const int f = t / nb;
                bits[d0 * nb + t] = f < nf ? (direct ? ambe49[so * 5 * 49 + t] : bits_fd[so * 5 * nb + t]) : 0;
            }
            if (lane < 5) {
                int32_t* r = res + (d0 + lane) * 5;
                if (lane >= nf) {
                    r[0] = 0, r[1] = 0, r[2] = 0, r[3] = 0, r[4] = 0;
                } else if (direct) {
                    const int e = errs2[so * 5 + lane];
                    r[0] = 0, r[1] = 0, r[2] = 0, r[3] = e, r[4] = e;
                } else {
                    const int32_t* q = res_fd + (so * 5 + lane) * 5;
                    r[0] = q[0], r[1] = q[1], r[2] = q[2], r[3] = q[3], r[4] = q[4];
                }
                skip[d0 + lane] = lane < nf ? 0 : 1;
            }
            if (lane == 0) {
                v_slot[(size_t)c * vf + j] = kk;
            }
            j++;
        }
    }
    if (lane == 0) {
        v_n[c] = j;
    }
    for (int t = 5 * j + lane; t < 5 * vf; t += 64) {
        skip[(size_t)c * vf * 5 + t] = 1;
        int32_t* r = res + ((size_t)c * vf * 5 + t) * 5;
        r[0] = 0, r[1] = 0, r[2] = 0, r[3] = 0, r[4] = 0;
    }
}

__global__ void
k_ysf_pack96(const uint8_t* __restrict__ frames184, size_t n, uint8_t* __restrict__ frames96) { // ambe_fr[4][24] out of the 184-byte slots
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 96) {
        frames96[i] = frames184[(i / 96) * 184 + i % 96];
    }
}

extern "C" hipError_t
ddn_dev_ysf_voice_file(const int32_t* n_sync, int n_channels, int max_syncs, const uint8_t* info, const uint8_t* ambe49, const uint8_t* errs2,
                       const uint8_t* bits_fd, const int32_t* res_fd, const uint8_t* n_frames, int mode, int vf, uint8_t* bits, int32_t* res,
                       uint8_t* skip, int32_t* v_n, int32_t* v_slot, hipStream_t st) {
    hipLaunchKernelGGL(k_ysf_voice_file, dim3((unsigned)n_channels), dim3(64), 0, st, n_sync, max_syncs, info, ambe49, errs2, bits_fd, res_fd,
                       n_frames, mode, vf, bits, res, skip, v_n, v_slot);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_pack96(const uint8_t* frames184, size_t n, uint8_t* frames96, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_ysf_pack96, dim3((unsigned)((n * 96 + 255) / 256)), dim3(256), 0, st, frames184, n, frames96);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_plan(const int32_t* sync_pos, const int32_t* n_sync, const int32_t* counts, int n_channels, int max_syncs, int lmax,
                 const uint8_t* fich4, const uint8_t* fich_status, uint8_t* last2, uint8_t* info, int32_t* slot_sync, hipStream_t st) {
    hipLaunchKernelGGL(k_ysf_plan, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, sync_pos, n_sync, counts, n_channels, max_syncs,
                       lmax, fich4, fich_status, last2, info, slot_sync);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_payload_costs(const uint8_t* rec, size_t stride, const int32_t* sync_pos, int n_channels, int max_syncs, int lmax,
                          const uint8_t* info, const int32_t* slot_sync, uint16_t* cost200, uint16_t* cost360, uint8_t* ambe49,
                          uint8_t* errs2, uint8_t* want200, uint8_t* want360, uint8_t* frames, uint8_t* n_frames, hipStream_t st) {
    hipLaunchKernelGGL(k_ysf_payload_costs, dim3((unsigned)n_channels, (unsigned)lmax), dim3(64), 0, st, rec, stride, sync_pos, max_syncs,
                       lmax, info, slot_sync, cost200, cost360, ambe49, errs2, want200, want360, frames, n_frames);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_dch_finish(const uint8_t* decA, const uint32_t* pcA, const uint8_t* decB, const uint32_t* pcB, const int32_t* slot_sync,
                       const uint8_t* info, int n_channels, int lmax, int max_syncs, uint8_t* dch40, uint8_t* dch_status,
                       uint32_t* dch_cost, hipStream_t st) {
    const int n = n_channels * lmax * 2;
    hipLaunchKernelGGL(k_ysf_dch_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, decA, pcA, decB, pcB, slot_sync, info,
                       n_channels, lmax, max_syncs, dch40, dch_status, dch_cost);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_fich_cost(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos, const int32_t* n_sync,
                      int n_channels, int max_syncs, int lmax, uint16_t* cost200, int32_t* slot_sync, hipStream_t st) {
    hipLaunchKernelGGL(k_ysf_fich_cost, dim3((unsigned)n_channels, (unsigned)lmax), dim3(64), 0, st, rec, stride, counts, sync_pos, n_sync,
                       max_syncs, lmax, cost200, slot_sync);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ysf_fich_finish(const uint8_t* dec, int dec_stride, const uint32_t* cost, const int32_t* slot_sync, int n_channels, int lmax,
                        int max_syncs, uint8_t* fich4, uint8_t* status, uint32_t* v_error, hipStream_t st) {
    const DdnFec3Tables* T = nullptr;
    const hipError_t e = ddn_dev_fec3_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    const int n = n_channels * lmax;
    hipLaunchKernelGGL(k_ysf_fich_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, dec, dec_stride, cost, slot_sync, n_channels,
                       lmax, max_syncs, T, fich4, status, v_error);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_m17_str_bits(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos, const uint8_t* sync_pat,
                     const int32_t* n_sync, int n_channels, int max_syncs, int lmax, uint8_t* sym296, int32_t* slot_sync, uint8_t* slot_lich6,
                     uint8_t* slot_cnt, uint8_t* slot_ok, hipStream_t st) {
    const DdnFec3Tables* T = nullptr;
    const hipError_t e = ddn_dev_fec3_tables(&T, st);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_m17_str_bits, dim3((unsigned)n_channels, (unsigned)lmax), dim3(64), 0, st, rec, stride, counts, sync_pos, sync_pat,
                       n_sync, max_syncs, lmax, T, sym296, slot_sync, slot_lich6, slot_cnt, slot_ok);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_m17_str_finish(const uint8_t* dec, int dec_stride, const int32_t* slot_sync, const uint8_t* slot_lich6, const uint8_t* slot_cnt,
                       const uint8_t* slot_ok, int n_channels, int lmax, int max_syncs, uint8_t* lich6, uint8_t* lich_cnt,
                       uint8_t* fn_payload18, uint8_t* status, hipStream_t st) {
    const int n = n_channels * lmax;
    hipLaunchKernelGGL(k_m17_str_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, dec, dec_stride, slot_sync, slot_lich6, slot_cnt,
                       slot_ok, n_channels, lmax, max_syncs, lich6, lich_cnt, fn_payload18, status);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_m17_lich(const uint8_t* sync_pat, const int32_t* n_sync, int n_channels, int max_syncs, const uint8_t* lsf30,
                 const uint8_t* lsf_status, const uint8_t* lich6, const uint8_t* lich_cnt, const uint8_t* str_status, uint8_t* asm30,
                 uint8_t* lich_lsf30, uint8_t* lich_status, hipStream_t st) {
    hipLaunchKernelGGL(k_m17_lich, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, sync_pat, n_sync, n_channels, max_syncs, lsf30,
                       lsf_status, lich6, lich_cnt, str_status, asm30, lich_lsf30, lich_status);
    return hipGetLastError();
}

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
