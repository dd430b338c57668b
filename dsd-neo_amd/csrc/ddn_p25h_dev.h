// ddn_p25h_dev.h - the P25 Phase 1 protocol handlers' hold on the receive loop, as the decision service the loop's handler
// wave runs (ddn_rx.hip, k_p25_rxw<CPW, true>).  In the reference a frame is read "in frame" for exactly as long as its handler
// keeps pulling dibits, and that depends on what the handler decodes:
//   NID 32 dibits + 1 status symbol, p25p1_nid_decode -> DUID        src/engine/dispatch/dispatch_p25p1.c:86-143,206-225
//   per DUID (:391-403): HDU 339 symbols, LDU1 / LDU2 807, TDU 15, TDULC 159, undefined / failed NID 0
//                                                                    p25p1_hdu.c:54-70,238-266,408; p25p1_ldu1.c:120-220;
//                                                                    p25p1_tdu.c:35-52; p25p1_tdulc.c:181-253
//   TSDU: 101 symbols per block, list-8 half-rate decode, first CRC16-clean candidate, until its last-block flag, <= 3 blocks
//                                                                    p25p1_tsbk.c:117-161,1051-1072
//   PDU: header block the same way, then blks + 1 blocks of 98 data dibits each (SAP 61 / 63 with blks > 10: 4)
//                                                                    p25p1_mdpu.c:177-198,270-307
// One decision at a time, the whole wavefront on it: the symbols of the phase are sliced (dibit + LLRs, ddn_slicer_dev.h) one
// per lane out of the loop's in-frame history ring; the NID goes through the full ladder (ddn_nid_dev.h: hard decode, NAC retry,
// Chase search one candidate per lane); a trellis block takes a short cut that is exact - when every LLR is non-zero, the hard
// dibits walk the trellis from state 0 and the bytes pass the CRC16, the list decoder's best candidate IS that code word
// (its metric is 0 and every other path's is positive) and the CRC scan stops at it - and otherwise the list-8 decoder proper,
// four lanes (one per state) as ddn_trellis.hip's k_p25_half_rate_list.
#ifndef DDN_P25H_DEV_H
#define DDN_P25H_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_nid_dev.h"
#include "ddn_slicer_dev.h"

namespace ddn_p25h {

constexpr int HN = 128; // in-frame history ring, symbols per channel (a phase is at most 101)
enum { PH_IDLE = 0, PH_NID = 1, PH_TSBK = 3, PH_MPDU = 4 };
enum { EV_NID = 1, EV_TSBK = 2, EV_MPDU = 3 };

struct Scratch { // one decision at a time
    uint8_t ex[128], lg[64];
    uint8_t work[(23 + 24 + 24 + 24) * 64];
    uint8_t masks[96];
    uint8_t nb[64], nr[64];  // NID bits / reliabilities (index 63 = the parity bit)
    int32_t d[98 + 2];       // de-interleaved LLR pairs (lo 16 = first bit of the dibit)
    uint8_t hd[100];         // de-interleaved hard dibits
    uint2 back[49][4];
    uint4 cand[32];
    uint8_t cvalid[32];
    uint32_t outl[8][4];     // the merged candidate list {bytes 0-3, 4-7, 8-11, metric}
    int n_out;
};

__device__ __forceinline__ int
deinterleave98(int i) { // as ddn_trellis.hip: 49 dibit pairs dealt to 4 lanes round-robin
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

__device__ __forceinline__ int
half_rate_nibble(int idx) { // trellis output for (state << 2) | next state, src/protocol/p25/p25_12.c:19
    // {2, 12, 1, 15, 14, 0, 13, 3, 9, 7, 10, 4, 5, 11, 6, 8} packed 4 bits each
    return (int)((0x86B5'4A79'3D0E'F1C2ull >> (4 * idx)) & 15u);
}

__device__ __forceinline__ int
crc16_ok(const uint32_t w[3]) { // p25_crc.c:18-36 over bytes 0..9 of the 12 (big-endian in the words' byte order below)
    unsigned crc = 0;
    for (int k = 0; k < 10; k++) {
        const unsigned v = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            const unsigned bit = (v >> j) & 1u;
            crc = (((crc >> 15) & 1u) ^ bit) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
        }
    }
    crc ^= 0xFFFFu;
    const unsigned b10 = (w[2] >> 16) & 0xFFu, b11 = (w[2] >> 24) & 0xFFu;
    return crc == ((b10 << 8) | b11);
}

// the list-8 decoder on lanes 0..3 of the wave (lane = next state), LLR pairs in sc.d: leaves the merged candidates in
// sc.outl[0 .. sc.n_out) sorted as the reference sorts them.  Byte k of a candidate = (outl[.][k >> 2] >> (8 * (k & 3))) & 0xFF.
__device__ inline void
half_rate_list_wave(Scratch& sc, int lane) {
    constexpr int K = 8;
    const uint32_t MAXM = 0xFFFFFFFFu;
    const int ns = lane & 3;
    const bool on = lane < 4;
    uint32_t pm[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        pm[r] = MAXM;
    }
    pm[0] = (ns == 0) ? 0u : 256u;
    int e[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        e[ps] = half_rate_nibble((ps << 2) | ns);
    }
    for (int t = 0; t < 49; t++) {
        const int32_t p0 = sc.d[2 * t], p1 = sc.d[2 * t + 1];
        const int l[4] = {(int16_t)(p0 & 0xFFFF), (int16_t)(p0 >> 16), (int16_t)(p1 & 0xFFFF), (int16_t)(p1 >> 16)};
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
                const uint32_t q = __shfl(pm[r], ps);
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
        if (on) {
            uint2 w;
            w.x = cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24);
            w.y = cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24);
            sc.back[t][ns] = w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 1
    for (int rk = 0; rk < K; rk++) {
        uint32_t mfin = pm[0];
#pragma unroll
        for (int r = 1; r < K; r++) {
            mfin = (r == rk) ? pm[r] : mfin;
        }
        uint32_t w[3] = {0, 0, 0};
        int s = ns, r = rk;
        if (on && mfin != MAXM) {
            for (int t = 48; t >= 0; t--) {
                if (t < 48) {
                    const int byte = t >> 2;
                    w[byte >> 2] |= (uint32_t)s << (8 * (byte & 3) + 6 - 2 * (t & 3));
                }
                const uint2 bw = sc.back[t][s];
                const uint32_t word = (r < 4) ? bw.x : bw.y;
                const uint32_t p = (word >> (8 * (r & 3))) & 0xFFu;
                s = (int)((p >> 3) & 3u);
                r = (int)(p & 7u);
            }
        }
        if (on) {
            sc.cand[ns * K + rk] = make_uint4(w[0], w[1], w[2], mfin);
            sc.cvalid[ns * K + rk] = (mfin != MAXM) ? 1 : 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) { // merge in (state, rank) order: duplicates by the 12 bytes dropped, the rest sorted by metric, strict less
        int count = 0;
        for (int k = 0; k < 32; k++) {
            if (!sc.cvalid[k]) {
                continue;
            }
            const uint4 cd = sc.cand[k];
            bool dup = false;
            int at = count;
            bool found = false;
            for (int i = 0; i < count; i++) {
                dup |= (sc.outl[i][0] == cd.x && sc.outl[i][1] == cd.y && sc.outl[i][2] == cd.z);
                if (!found && cd.w < sc.outl[i][3]) {
                    at = i;
                    found = true;
                }
            }
            if (dup) {
                continue;
            }
            if (count < K) {
                count++;
            } else if (at >= K) {
                continue;
            }
            for (int i = count - 1; i > at; i--) {
                for (int j = 0; j < 4; j++) {
                    sc.outl[i][j] = sc.outl[i - 1][j];
                }
            }
            sc.outl[at][0] = cd.x;
            sc.outl[at][1] = cd.y;
            sc.outl[at][2] = cd.z;
            sc.outl[at][3] = cd.w;
        }
        sc.n_out = count;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// skipdibit bookkeeping of tsbk_read_repetition_samples() / p25_mpdu_read_repetition(): symbol i of a block read with the
// counter at sk0 is a status symbol when the counter has reached 36
__device__ __forceinline__ void
block_scan(int sk0, int n_sym, int i_query, bool& is_status, int& data_index, int& sk_after, int& n_data) {
    int sk = sk0, k = 0;
    is_status = false;
    data_index = 0;
    for (int i = 0; i < n_sym; i++) {
        const bool st = (sk / 36) != 0;
        if (i == i_query) {
            is_status = st;
            data_index = k;
        }
        if (st) {
            sk = 0;
        } else {
            k++;
        }
        sk++;
    }
    sk_after = sk;
    n_data = k;
}

// symbols p25_mpdu_read_repetition() consumes for one repetition starting with the counter at sk0 (98 data dibits, <= 101 reads)
__device__ __forceinline__ int
mpdu_block_symbols(int sk0, int& sk_after) {
    int sk = sk0, k = 0, i = 0;
    for (; i < 101; i++) {
        if ((sk / 36) == 0) {
            k++;
        } else {
            sk = 0;
        }
        sk++;
        if (k == 98) {
            i++;
            break;
        }
    }
    sk_after = sk;
    return i;
}

struct Verdict {
    int ext;  // symbols the handler reads next (0 = it has returned)
    int more; // another decision falls due when those are read
};

} // namespace ddn_p25h
#endif
