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

constexpr int HN = 104; // in-frame history ring, symbols per channel (a phase is at most 101)
enum { PH_IDLE = 0, PH_NID = 1, PH_TSBK = 3, PH_MPDU = 4 };
enum { EV_NID = 1, EV_TSBK = 2, EV_MPDU = 3 };

struct Scratch { // one decision at a time
    uint8_t ex[128], lg[64];
    uint8_t masks[96];
    uint8_t nb[64], nr[64];  // NID bits / reliabilities (index 63 = the parity bit)
    int32_t d[98 + 2];       // de-interleaved LLR pairs (lo 16 = first bit of the dibit)
    uint16_t crc_cols[80];   // CRC16 register contribution of each message bit
    uint32_t byw[4];         // half_rate_best_wave: the decoded block, or-ed together lane by lane
    union {
        uint8_t work[(23 + 24 + 24 + 24) * 64]; // NID: the BCH decoder's per-lane arrays
        struct {                                // trellis block: the list decoder's
            uint2 back[49][4];       // as bytes [49][4][8]
            uint16_t ct[49][4][4];   // branch costs [step][state][predecessor]
            alignas(16) uint32_t mb[2][32]; // survivor metrics [step parity][state * 8 + rank]
            uint4 cand[32];
            uint8_t cvalid[32];
            uint32_t outl[8][4]; // the merged candidate list {bytes 0-3, 4-7, 8-11, metric}
            int n_out;
        };
    };
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

// The list-8 decoder (src/protocol/p25/p25_12.c:31-202), the whole wavefront on one block, LLR pairs in sc.d: leaves the merged
// candidates in sc.outl[0 .. sc.n_out) sorted as the reference sorts them (byte k of a candidate =
// (outl[.][k >> 2] >> (8 * (k & 3))) & 0xFF).
// The reference keeps, per state, the 8 smallest of the 32 extensions (4 predecessors x 8 ranks) by inserting them predecessor by
// predecessor, rank by rank, each before the first strictly larger metric - i.e. the 8 smallest under the order (metric, arrival
// index).  Here lane (p, r) of each half-wave owns survivor r of state p; every survivor list is sorted, so an extension's place
// among a state's 32 is its own rank plus, per other predecessor list, the number of entries below a threshold (its metric
// shifted by the difference of the two branch costs, +1 when that list arrives earlier) - 24 compares instead of an insertion
// ladder.  The two half-waves take two target states each.  Branch costs of all 49 steps are tabulated first; the per-step chain
// is one LDS round trip (the 32 metrics) + the counting.
__device__ inline void
half_rate_list_wave(Scratch& sc, int lane) {
    constexpr int K = 8;
    const uint32_t INF = 0x3FFFFFFFu; // an absent survivor (metrics stay below 2^17)
    const int L = lane & 31, p = L >> 3, r = L & 7, h = lane >> 5;
    // ---- branch-cost table ct[t][ns][ps] ------------------------------------------------------------------------------
    for (int idx = lane; idx < 49 * 16; idx += 64) {
        const int t = idx >> 4, ns = (idx >> 2) & 3, ps = idx & 3;
        const int32_t p0 = sc.d[2 * t], p1 = sc.d[2 * t + 1];
        const int l[4] = {(int16_t)(p0 & 0xFFFF), (int16_t)(p0 >> 16), (int16_t)(p1 & 0xFFFF), (int16_t)(p1 >> 16)};
        const int e = half_rate_nibble((ps << 2) | ns);
        uint32_t c = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            c += ((e >> (3 - b)) & 1) ? (l[b] < 0 ? (uint32_t)(-l[b]) : 0u) : (l[b] > 0 ? (uint32_t)l[b] : 0u);
        }
        sc.ct[t][ns][ps] = (uint16_t)c;
    }
    if (lane < 32) {
        sc.mb[0][lane] = (r == 0) ? ((p == 0) ? 0u : 256u) : INF;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint8_t* back = reinterpret_cast<uint8_t*>(&sc.back[0][0]); // [49][4][8]: (predecessor << 3) | rank of survivor (state, rank)
    for (int t = 0; t < 49; t++) {
        const uint32_t* mb = sc.mb[t & 1];
        uint32_t* mn = sc.mb[(t + 1) & 1];
        uint32_t m[32];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint4 v = *reinterpret_cast<const uint4*>(&mb[4 * q]);
            m[4 * q] = v.x;
            m[4 * q + 1] = v.y;
            m[4 * q + 2] = v.z;
            m[4 * q + 3] = v.w;
        }
        const uint32_t mine = mb[L];
        if (lane < 32) {
            mn[lane] = INF; // survivors nobody claims stay absent
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int ns = 2 * h + j;
            const uint2 cw = *reinterpret_cast<const uint2*>(&sc.ct[t][ns][0]);
            const uint32_t c[4] = {cw.x & 0xFFFFu, cw.x >> 16, cw.y & 0xFFFFu, cw.y >> 16};
            const uint32_t cp = p == 0 ? c[0] : (p == 1 ? c[1] : (p == 2 ? c[2] : c[3]));
            const uint32_t val = mine + cp;
            int rank = r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // entries of list q that come before (val, arrival): metric + c[q] < val, or == val when list q arrives earlier.
                // Signed compare: the threshold can be negative, an absent entry (INF) is above every threshold.
                const int32_t thr = (int32_t)(val - c[q] + (q < p ? 1u : 0u));
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    cnt += ((int32_t)m[8 * q + i] < thr) ? 1 : 0;
                }
                rank += (q != p) ? cnt : 0; // its own list contributes its rank
            }
            if (mine < INF && rank < K) {
                mn[8 * ns + rank] = val;
                back[(t * 4 + ns) * 8 + rank] = (uint8_t)L;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- trace the 32 final survivors back (lane = state * 8 + rank) ---------------------------------------------------------
    const uint32_t mfin = sc.mb[49 & 1][L];
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    {
        uint32_t w[3] = {0, 0, 0};
        int s = p, rk = r;
        if (lane < 32 && mfin < INF) {
            for (int t = 48; t >= 0; t--) {
                if (t < 48) {
                    const int byte = t >> 2;
                    w[byte >> 2] |= (uint32_t)s << (8 * (byte & 3) + 6 - 2 * (t & 3));
                }
                const int bp = back[(t * 4 + s) * 8 + rk];
                s = (bp >> 3) & 3;
                rk = bp & 7;
            }
        }
        w0 = w[0];
        w1 = w[1];
        w2 = w[2];
    }
    // ---- merge in (state, rank) order: a candidate whose 12 bytes are already listed is dropped, the rest kept sorted by metric
    // (inserted before the first strictly larger one), at most 8.  Slot i of the list lives in lane i; the candidates come round
    // as scalars.
    uint32_t o0 = 0, o1 = 0, o2 = 0, om = 0;
    int count = 0;
    for (int k = 0; k < 32; k++) {
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)w0, k), c1 = (uint32_t)__builtin_amdgcn_readlane((int)w1, k);
        const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)w2, k), cm = (uint32_t)__builtin_amdgcn_readlane((int)mfin, k);
        if (cm >= INF) {
            continue;
        }
        const bool mine_on = lane < count;
        if (__any(mine_on && o0 == c0 && o1 == c1 && o2 == c2)) {
            continue;
        }
        const int at = __popcll(__ballot(mine_on && !(cm < om))); // entries that stay in front: not strictly larger
        if (count < K) {
            count++;
        } else if (at >= K) {
            continue;
        }
        // slots at + 1 .. count - 1 take their left neighbour's entry, slot `at` the candidate
        const uint32_t s0 = __shfl_up(o0, 1), s1 = __shfl_up(o1, 1), s2 = __shfl_up(o2, 1), sm = __shfl_up(om, 1);
        const bool shift = lane > at && lane < count;
        o0 = lane == at ? c0 : (shift ? s0 : o0);
        o1 = lane == at ? c1 : (shift ? s1 : o1);
        o2 = lane == at ? c2 : (shift ? s2 : o2);
        om = lane == at ? cm : (shift ? sm : om);
    }
    if (lane < K) {
        sc.outl[lane][0] = o0;
        sc.outl[lane][1] = o1;
        sc.outl[lane][2] = o2;
        sc.outl[lane][3] = om;
    }
    if (lane == 0) {
        sc.n_out = count;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// skipdibit bookkeeping of tsbk_read_repetition_samples() / p25_mpdu_read_repetition() in closed form.  The counter starts a
// block at sk0 (1..36); a symbol read with the counter at 36 is a status symbol and restarts it, so status symbols sit at
// i0 = 36 - sk0 and every 36 after.
__device__ __forceinline__ bool
block_is_status(int sk0, int i, int& data_index) {
    const int i0 = 36 - sk0;
    const int before = (i <= i0) ? 0 : ((i - 1 - i0) / 36 + 1); // status symbols among 0 .. i - 1
    data_index = i - before;
    return i >= i0 && ((i - i0) % 36) == 0;
}
__device__ __forceinline__ int
block_counter_after(int sk0, int n_sym, int& n_data) { // the counter after n_sym symbols, and how many of them were data
    const int i0 = 36 - sk0;
    if (n_sym <= i0) {
        n_data = n_sym;
        return sk0 + n_sym;
    }
    const int n_status = (n_sym - 1 - i0) / 36 + 1;
    n_data = n_sym - n_status;
    return n_sym - (i0 + 36 * (n_status - 1));
}

// XOR over the wavefront: four DPP steps inside each row of 16 lanes, then the four rows through scalar reads
__device__ __forceinline__ uint32_t
wave_xor(uint32_t v) {
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); // row_half_mirror
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 16)
           ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 32) ^ (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// CRC16 (p25_crc.c:18-36: 0x1021, zero start, inverted) of bytes 0..9 against bytes 10..11, the wavefront on it: the register
// is linear in the message, so lane j adds in the contribution of message bits j and j + 64 (sc.crc_cols: the register after
// shifting a lone one through the remaining bits) and a wave-wide XOR finishes.  by[] is the same in every lane.
__device__ inline void
crc_cols_fill(uint16_t* cols, int lane) { // lanes 0..63 of one wave
    for (int i = lane; i < 80; i += 64) {
        unsigned crc = 0;
        for (int k = 0; k < 80; k++) {
            const unsigned bit = (k == i) ? 1u : 0u;
            crc = (((crc >> 15) & 1u) ^ bit) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
        }
        cols[i] = (uint16_t)crc;
    }
}
__device__ __forceinline__ int
crc16_ok_wave(const Scratch& sc, const uint32_t by[3], int lane) {
    auto msg_bit = [&](int i) { return (by[i >> 5] >> (8 * ((i >> 3) & 3) + 7 - (i & 7))) & 1u; };
    uint32_t v = msg_bit(lane) ? sc.crc_cols[lane] : 0u;
    if (lane < 16) {
        v ^= msg_bit(lane + 64) ? sc.crc_cols[lane + 64] : 0u;
    }
    const uint32_t crc = wave_xor(v) ^ 0xFFFFu;
    const uint32_t b10 = (by[2] >> 16) & 0xFFu, b11 = (by[2] >> 24) & 0xFFu;
    return crc == ((b10 << 8) | b11);
}

// Best path of the half-rate trellis over the LLR pairs in sc.d: the decoder p25_12_soft_llr() runs, and the list decoder's
// first candidate - its rank-0 survivors evolve exactly like this (the smallest of a state's 32 extensions is the smallest rank-0
// extension, first predecessor on a tie) and its merge keeps the smallest final metric, first state on a tie.
// Sixteen lanes, lane = state * 4 + predecessor (the rest of the wave repeats them).  Nothing but registers on the 49-step
// chain: the 98 LLR pairs are spread over the wave first (lane j keeps pairs j and j + 64, a step fetches its two by readlane),
// a branch's cost is packed 16-bit arithmetic on the pair (negate by the branch bit, clamp at 0, add), the four extensions of
// a state sit in one quad (two DPP minimum steps), the four new metrics go round as scalars (readlane), every lane keeps its
// state's decisions two bits per step, and the trace-back runs on scalars.  Every lane returns the 12 bytes (byte k =
// (by[k >> 2] >> (8 * (k & 3))) & 0xFF).
typedef short ddn_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t
map4_compose(uint32_t m, uint32_t b) { // (m o b)(x) = m(b(x)) for maps of {0..3} kept as four 2-bit fields
    uint32_t r = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t bx = (b >> (2 * x)) & 3u;
        r |= ((m >> (2 * bx)) & 3u) << (2 * x);
    }
    return r;
}
// Round 3 shape: (1) the branch costs of all 49 steps first - no chain between them, the LLR pairs are LDS broadcast reads - packed
// two per register; (2) the add-compare-select chain: two DPP quad minima, the four states' decisions or-ed across the row (two
// row rotations) and kept by lane t as step t's map state -> predecessor, the next step's predecessor metric by one ds_bpermute
// (was four readlanes + selects, whose scalar results stall the vector pipe; three DPP row rotations + selects measured slower); (3) the trace-back as a suffix composition of
// those maps over the lanes (six shuffle rounds) instead of 49 dependent scalar steps, the decoded dibits or-ed into three LDS
// words.  Same survivors, same ties (lowest predecessor, lowest final state) as before: 11.7 k -> 9.2 k cycles.  Round 5: the
// ds_bpermute went too (alternating lane layouts, below).
__device__ __forceinline__ void
half_rate_best_wave(Scratch& sc, int lane, uint32_t by[3]) {
    // Round 5: the (predecessor, state) pairs sit on the lanes in two layouts that take turns - even steps: predecessor = lane & 3,
    // state = quad (minimum over the quad, decisions or-ed across the quads); odd steps: state = lane & 3, predecessor = quad
    // (minimum across the quads, decisions or-ed inside the quad).  After an even step a quad holds its state's new metric, which is
    // the odd step's predecessor metric of that quad's lanes; after an odd step the lanes with the same lane & 3 hold it, the even
    // step's predecessor: no lane ever asks another for a metric (the ds_bpermute per step - an LDS round trip on the chain - is gone).
    const int lo = lane & 3, hi = (lane >> 2) & 3;
    const int eA = half_rate_nibble((lo << 2) | hi); // even steps: ps = lo, ns = hi
    const int eB = half_rate_nibble((hi << 2) | lo); // odd steps: ps = hi, ns = lo
    // branch bits as packed masks: bit set = the branch says 1 = the cost is the LLR's pull towards 0 = max(0, -llr)
    const ddn_s16x2 a01 = {(short)(((eA >> 3) & 1) ? -1 : 0), (short)(((eA >> 2) & 1) ? -1 : 0)};
    const ddn_s16x2 a23 = {(short)(((eA >> 1) & 1) ? -1 : 0), (short)((eA & 1) ? -1 : 0)};
    const ddn_s16x2 b01 = {(short)(((eB >> 3) & 1) ? -1 : 0), (short)(((eB >> 2) & 1) ? -1 : 0)};
    const ddn_s16x2 b23 = {(short)(((eB >> 1) & 1) ? -1 : 0), (short)((eB & 1) ? -1 : 0)};
    const ddn_s16x2 zero = {0, 0};
    uint32_t cst[25];
#pragma unroll
    for (int t = 0; t < 49; t++) {
        const ddn_s16x2 m01 = (t & 1) ? b01 : a01, m23 = (t & 1) ? b23 : a23;
        ddn_s16x2 a = __builtin_bit_cast(ddn_s16x2, sc.d[2 * t]), b = __builtin_bit_cast(ddn_s16x2, sc.d[2 * t + 1]);
        a = __builtin_elementwise_max((a ^ m01) - m01, zero);
        b = __builtin_elementwise_max((b ^ m23) - m23, zero);
        const ddn_s16x2 c = a + b; // <= 2 * 255 per half
        const uint32_t cw = __builtin_bit_cast(uint32_t, c);
        const uint32_t cost = (cw & 0xFFFFu) + (cw >> 16);
        if (t & 1) {
            cst[t >> 1] |= cost << 16;
        } else {
            cst[t >> 1] = cost;
        }
    }
    if (lane < 3) {
        sc.byw[lane] = 0;
    }
    uint32_t pm_ps = (lo == 0) ? 0u : 256u; // metric of this lane's predecessor state (step 0: predecessor = lane & 3)
    uint32_t mymap = 0xE4u, key = 0;        // lane t: step t's map state -> best predecessor (0xE4 = identity)
#pragma unroll
    for (int t = 0; t < 49; t++) {
        const uint32_t cost = (t & 1) ? (cst[t >> 1] >> 16) : (cst[t >> 1] & 0xFFFFu);
        uint32_t v;
        if (!(t & 1)) {
            key = ((pm_ps + cost) << 2) | (uint32_t)lo; // smallest metric, lowest predecessor on a tie
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xF, 0xF, true);
            key = o < key ? o : key;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xF, 0xF, true);
            key = o < key ? o : key;
            v = (key & 3u) << (2 * hi);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); // row_ror:4
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); // row_ror:8
        } else {
            key = ((pm_ps + cost) << 2) | (uint32_t)hi;
            uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, true);
            key = o < key ? o : key;
            o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, true);
            key = o < key ? o : key;
            v = (key & 3u) << (2 * lo);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
            v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
        }
        mymap = (lane == t) ? v : mymap;
        pm_ps = key >> 2;
    }
    const int ns = hi; // step 48 is an even one: every quad holds its state's metric
    // best final state, lowest on a tie: every quad holds its state's metric
    uint32_t f = ((key >> 2) << 2) | (uint32_t)ns;
    uint32_t g = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x124, 0xF, 0xF, true);
    f = g < f ? g : f;
    g = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x128, 0xF, 0xF, true);
    f = g < f ? g : f;
    const uint32_t s48 = f & 3u;
    // sigma_t (the state after step t = the decoded dibit t) = (D_{t+1} o ... o D_48)(s48): suffix composition over lanes 0..47
    uint32_t m = (uint32_t)__shfl_down((int)mymap, 1);
    m = lane < 48 ? m : 0xE4u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t bmap = (uint32_t)__shfl_down((int)m, d);
        bmap = (lane + d > 63) ? 0xE4u : bmap;
        m = map4_compose(m, bmap);
    }
    if (lane < 48) {
        const uint32_t sig = (m >> (2 * s48)) & 3u;
        atomicOr(&sc.byw[lane >> 4], sig << (8 * ((lane >> 2) & 3) + 6 - 2 * (lane & 3)));
    }
    by[0] = sc.byw[0];
    by[1] = sc.byw[1];
    by[2] = sc.byw[2];
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
