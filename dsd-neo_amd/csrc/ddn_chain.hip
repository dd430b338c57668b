// ddn_chain.hip - the small device helpers of the P25 Phase 1 chain object (ddn_api_chain.cpp): carrying the tail of a call's
// records into the next call so that frames crossing a call boundary decode whole, the two per-channel symbol counts the framer
// needs for that, and the candidate selection of a TSDU block.
//
// reference: tsbk_decode_repetition_bytes() / tsbk_select_crc_candidate(), src/protocol/p25/phase1/p25p1_tsbk.c:108-130 - the
// list decoder's first candidate whose CRC16 is clean, else its best one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {

// the last T records / flags of every channel's (carried + new) stretch -> the front of the other buffer set
__global__ __launch_bounds__(256) void
k_chain_carry(const uint8_t* __restrict__ rec_prev, const uint8_t* __restrict__ fl_prev, const int32_t* __restrict__ cnt_prev,
              int have_prev, uint8_t* __restrict__ rec_cur, uint8_t* __restrict__ fl_cur, size_t stride_sym, int T, int n_channels) {
    const int c = blockIdx.y;
    if (c >= n_channels) {
        return;
    }
    // a record is 10 bytes: move 16-bit words (5 per record)
    const int n_prev = have_prev ? cnt_prev[c] : 0; // new records of the previous call (they sit behind its T carried ones)
    const uint16_t* src = reinterpret_cast<const uint16_t*>(rec_prev + (size_t)c * stride_sym * 10);
    uint16_t* dst = reinterpret_cast<uint16_t*>(rec_cur + (size_t)c * stride_sym * 10);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T * 5; i += gridDim.x * 256) {
        const int k = i / 5;                   // record k of the new front = record n_prev + k of the previous stretch
        dst[i] = have_prev ? src[(size_t)(n_prev + k) * 5 + (i - 5 * k)] : (uint16_t)0;
    }
    for (int k = blockIdx.x * 256 + threadIdx.x; k < T; k += gridDim.x * 256) {
        fl_cur[(size_t)c * stride_sym + k] = have_prev ? fl_prev[(size_t)c * stride_sym + n_prev + k] : (uint8_t)0;
    }
}

// the host form of a record: {dibit | flags << 2, reliability} - what the reference's consumers of a dibit stream read
// (dibit, in-frame / sync / polarity marks, the reliability byte); the 16-bit soft values and the float symbol stay on the device
__global__ __launch_bounds__(256) void
k_chain_pack2(const uint8_t* __restrict__ rec, const uint8_t* __restrict__ fl, size_t n, uint16_t* __restrict__ out2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint16_t w = *reinterpret_cast<const uint16_t*>(rec + i * 10); // dibit | reliability << 8
        out2[i] = (uint16_t)((w & 0xFF03u) | ((uint16_t)(fl[i] & 0x3Fu) << 2));
    }
}

// The synthesized PCM frames of a call, dense: frame slots whose vocoder result does not carry the skip mark, in slot order, with
// their slot numbers - what a host reads instead of every slot's 640 bytes (most slots of a batch are empty: control channels,
// unused LDU slots).  Three small kernels: used slots per 1024-slot block, the blocks' offsets, the copy.
__global__ __launch_bounds__(256) void
k_pcm_count(const int32_t* __restrict__ result5, int n_slots, int32_t* __restrict__ block_cnt) {
    __shared__ int part[4];
    const int base = blockIdx.x * 1024;
    int c = 0;
    for (int k = 0; k < 4; k++) {
        const int i = base + k * 256 + (int)threadIdx.x;
        c += (i < n_slots && result5[(size_t)i * 5] >= 0) ? 1 : 0; // the skip mark is the word's top bit
    }
    for (int d = 32; d > 0; d >>= 1) {
        c += __shfl_down(c, d);
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
    }
}

__global__ __launch_bounds__(64) void
k_pcm_offsets(const int32_t* __restrict__ block_cnt, int n_blocks, int32_t* __restrict__ block_off, int32_t* __restrict__ total) {
    // one wavefront: running offsets over the blocks, 64 at a time
    int run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + (int)threadIdx.x;
        const int v = b < n_blocks ? block_cnt[b] : 0;
        int inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            inc += ((int)threadIdx.x >= d) ? t : 0;
        }
        if (b < n_blocks) {
            block_off[b] = run + inc - v;
        }
        run += __shfl(inc, 63);
    }
    if (threadIdx.x == 0) {
        *total = run;
    }
}

__global__ __launch_bounds__(256) void
k_pcm_compact(const int32_t* __restrict__ result5, const float* __restrict__ pcm, int n_slots, const int32_t* __restrict__ block_off,
              long capacity, float* __restrict__ dense, int32_t* __restrict__ slot_of) {
    __shared__ int woff[17];
    __shared__ int list[1024];
    const int base = blockIdx.x * 1024;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // slot order: wave w, round k covers slots base + (4 k + w) * 64 .. + 63
    int pos[4];
    unsigned long long m[4];
    int cnt[4];
    for (int k = 0; k < 4; k++) {
        const int i = base + (4 * k + wave) * 64 + lane;
        const bool used = i < n_slots && result5[(size_t)i * 5] >= 0;
        m[k] = __ballot(used);
        cnt[k] = __popcll(m[k]);
        pos[k] = used ? __popcll(m[k] & ((1ull << lane) - 1ull)) : -1;
    }
    if (lane == 0) {
        for (int k = 0; k < 4; k++) {
            woff[1 + 4 * k + wave] = cnt[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        woff[0] = 0;
        for (int j = 1; j <= 16; j++) {
            woff[j] += woff[j - 1];
        }
    }
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        if (pos[k] >= 0) {
            list[woff[4 * k + wave] + pos[k]] = base + (4 * k + wave) * 64 + lane;
        }
    }
    __syncthreads();
    const int n_used = woff[16];
    const long out0 = block_off[blockIdx.x];
    for (int j = wave; j < n_used; j += 4) { // one wavefront per frame: 160 floats
        const long o = out0 + j;
        if (o >= capacity) {
            break;
        }
        const int sl = list[j];
        const float* src = pcm + (size_t)sl * 160;
        float* dst = dense + (size_t)o * 160;
        for (int t = lane; t < 160; t += 64) {
            dst[t] = src[t];
        }
        if (lane == 0) {
            slot_of[o] = sl;
        }
    }
}

// scan limit (syncs accepted before it are decoded in this call: the T symbols behind it are there) and full length
__global__ void
k_chain_counts(const int32_t* __restrict__ cnt_new, int T, int n_channels, int flush, int32_t* __restrict__ cnt_scan,
               int32_t* __restrict__ cnt_full) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_channels) {
        cnt_full[c] = cnt_new[c] + T;
        cnt_scan[c] = flush ? cnt_new[c] + T : cnt_new[c];
    }
}

// The handler decisions of the records a row holds: the ones carried over from the previous call's list (position >= that call's
// new-record count, i.e. inside the carried tail; shifted into this row's indexing) followed by this call's (staged by the receive
// loop with positions relative to the new records).  {row index, kind, a, b} + the decoded payload, in position order.
__global__ void
k_chain_events(const int32_t* __restrict__ list_prev, const int32_t* __restrict__ data_prev, const int32_t* __restrict__ n_prev,
               const int32_t* __restrict__ new_prev, int have_prev, const int32_t* __restrict__ ev_new,
               const int32_t* __restrict__ evd_new, const int32_t* __restrict__ n_new, int E, int EL, int T, int n_channels,
               int32_t* __restrict__ list_cur, int32_t* __restrict__ data_cur, int32_t* __restrict__ n_cur) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) {
        return;
    }
    int4* lc = reinterpret_cast<int4*>(list_cur) + (size_t)c * EL;
    int4* dc = reinterpret_cast<int4*>(data_cur) + (size_t)c * EL;
    int k = 0;
    if (have_prev) {
        const int4* lp = reinterpret_cast<const int4*>(list_prev) + (size_t)c * EL;
        const int4* dp = reinterpret_cast<const int4*>(data_prev) + (size_t)c * EL;
        const int np = n_prev[c] < EL ? n_prev[c] : EL, shift = new_prev[c];
        for (int j = 0; j < np; j++) {
            int4 e = lp[j];
            if (e.x >= shift && k < EL) {
                e.x -= shift;
                lc[k] = e;
                dc[k] = dp[j];
                k++;
            }
        }
    }
    const int4* en = reinterpret_cast<const int4*>(ev_new) + (size_t)c * E;
    const int4* dn = reinterpret_cast<const int4*>(evd_new) + (size_t)c * E;
    const int nn = n_new[c] < E ? n_new[c] : E;
    for (int j = 0; j < nn && k < EL; j++) {
        int4 e = en[j];
        e.x += T;
        lc[k] = e;
        dc[k] = dn[j];
        k++;
    }
    n_cur[c] = k;
}

// Per frame slot (the k-th sync the framer indexed in this row): the NID the loop's handler decoded for it (kind 1, 33 symbols
// after the sync's last one) and its TSDU blocks (kind 2, at the block's last dibit) - p25p1_nid_decode / tsbk_decode_repetition_bytes
// ran once, inside the loop, exactly as the reference's handlers run them; this only files their results by frame.
__global__ void
k_chain_frames(const int32_t* __restrict__ list, const int32_t* __restrict__ data, const int32_t* __restrict__ n_list, int EL,
               const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_syncs, int n_channels, int F, int off0, int off1,
               int off2, int32_t* __restrict__ nid4, uint8_t* __restrict__ tsbk, uint8_t* __restrict__ tsbk_crc,
               uint8_t* __restrict__ cls, int32_t* __restrict__ lists, int32_t* __restrict__ list_n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) {
        return;
    }
    const size_t S = (size_t)n_channels * F;
    const int ns = n_syncs[c] < F ? n_syncs[c] : F;
    const int32_t* sp = sync_pos + (size_t)c * F;
    for (int k = 0; k < F; k++) {
        const size_t slot = (size_t)c * F + k;
        reinterpret_cast<int4*>(nid4)[slot] = make_int4(0, 0, 0, 0);
        cls[slot] = 0;
        for (int b = 0; b < 3; b++) {
            uint32_t* o = reinterpret_cast<uint32_t*>(tsbk + ((size_t)b * S + slot) * 12);
            o[0] = o[1] = o[2] = 0;
            tsbk_crc[(size_t)b * S + slot] = 0;
        }
    }
    const int4* l = reinterpret_cast<const int4*>(list) + (size_t)c * EL;
    const int4* d = reinterpret_cast<const int4*>(data) + (size_t)c * EL;
    const int n = n_list[c] < EL ? n_list[c] : EL;
    int k = 0;
    for (int j = 0; j < n; j++) {
        const int4 e = l[j];
        if (e.y != 1 && e.y != 2) {
            continue;
        }
        const int4 v = d[j];
        const int blk = (v.w >> 16) & 0xFF;
        const int a = e.y == 1 ? e.x - 33 : e.x - (blk == 0 ? off0 : (blk == 1 ? off1 : off2));
        while (k < ns && sp[k] < a) {
            k++;
        }
        if (k >= ns || sp[k] != a) {
            continue; // the frame's sync is not among this call's (it lies in the tail that is carried on)
        }
        const size_t slot = (size_t)c * F + k;
        if (e.y == 1) {
            reinterpret_cast<int4*>(nid4)[slot] = v;
            // frame type for the per-type decode launches that follow (DDN_CLS_*): only a decoded NID names one
            const int duid = v.z;
            const int ty = v.x > 0 ? (duid == 0x5 ? 0 : (duid == 0xA ? 1 : (duid == 0x0 ? 2 : (duid == 0xF ? 3 : -1)))) : -1;
            cls[slot] = ty >= 0 ? (uint8_t)(1 << ty) : 0;
            if (ty >= 0) { // the slot joins its frame type's work list (and the LDUs the low-speed-data list, 4): DDN_LIST_*
                lists[(size_t)ty * S + atomicAdd(&list_n[ty], 1)] = (int32_t)slot;
                if (ty < 2) {
                    lists[(size_t)4 * S + atomicAdd(&list_n[4], 1)] = (int32_t)slot;
                }
            }
        } else if (blk < 3) {
            uint32_t* o = reinterpret_cast<uint32_t*>(tsbk + ((size_t)blk * S + slot) * 12);
            o[0] = (uint32_t)v.x;
            o[1] = (uint32_t)v.y;
            o[2] = (uint32_t)v.z;
            tsbk_crc[(size_t)blk * S + slot] = (uint8_t)(v.w & 1);
        }
    }
}

// ---- P25 Phase 1 data units (DUID 0xC, processMPDU(), src/protocol/p25/phase1/p25p1_mdpu.c) -------------------------------------
// The header block is decoded inside the receive loop (its blocks-to-follow field decides how long the frame is read: event kind 3,
// p25_mpdu_update_header_from_first_block :309-326); the data blocks behind it are decoded here.  k_chain_pdu_index: one thread per
// channel files the call's data units - entry = channel * PF + rank in air order: the frame slot, the header as the loop decoded it
// {12 bytes, CRC16 good}, the number of blocks the reference reads (ctx->end, 3 when the header's CRC fails) - so that every later
// array has a fixed place.  k_chain_pdu_gather: block b >= 1 of an entry -> its 98 dibits' LLR pairs in received order (payload
// dibit n = 56 + 98 b + j of the frame sits at frame symbol n + n / 35: a status symbol follows every 35 dibits, :195-217).
// k_chain_pdu_finish: header fields, the data blocks' bytes, CRC32 over the data (crc32mbf :47-60, p25_mpdu_handle_rate12 :636-650),
// and - when the first header fails its CRC16 and the header says nothing about the length - the reference's first fallback: blocks
// 1 and 2 read as repetitions of the header, the first one with a good CRC16 taken (p25_mpdu_finalize_header :381-395).
__global__ void
k_chain_pdu_index(const int32_t* __restrict__ list, const int32_t* __restrict__ data, const int32_t* __restrict__ n_list, int EL,
                  const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_syncs, const int32_t* __restrict__ nid4,
                  int n_channels, int F, int off0, int PF, int32_t* __restrict__ pdu_slot, uint8_t* __restrict__ pdu_hdr,
                  int32_t* __restrict__ pdu_info, int32_t* __restrict__ n_pdu) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_channels) {
        return;
    }
    const int ns = n_syncs[c] < F ? n_syncs[c] : F;
    const int32_t* sp = sync_pos + (size_t)c * F;
    const int4* l = reinterpret_cast<const int4*>(list) + (size_t)c * EL;
    const int4* d = reinterpret_cast<const int4*>(data) + (size_t)c * EL;
    const int n = n_list[c] < EL ? n_list[c] : EL;
    int k = 0, r = 0, total = 0;
    for (int j = 0; j < n; j++) {
        const int4 e = l[j];
        if (e.y != 3) {
            continue;
        }
        const int a = e.x - off0;
        while (k < ns && sp[k] < a) {
            k++;
        }
        if (k >= ns || sp[k] != a) {
            continue; // the frame's sync is carried on to the next call
        }
        const size_t slot = (size_t)c * F + k;
        if (nid4[slot * 4] <= 0 || nid4[slot * 4 + 2] != 0xC) {
            continue;
        }
        total++;
        if (r >= PF) {
            continue;
        }
        const int4 v = d[j];
        const size_t o = (size_t)c * PF + r;
        pdu_slot[o] = (int32_t)slot;
        uint32_t* h = reinterpret_cast<uint32_t*>(pdu_hdr + o * 12);
        h[0] = (uint32_t)v.x;
        h[1] = (uint32_t)v.y;
        h[2] = (uint32_t)v.z;
        pdu_info[o * 4 + 0] = v.w & 1;          // header CRC16 good
        pdu_info[o * 4 + 1] = e.w & 0xFFFF;     // blocks the reference reads (header included)
        pdu_info[o * 4 + 2] = 0;
        pdu_info[o * 4 + 3] = 0;
        r++;
    }
    for (int q = r; q < PF; q++) {
        pdu_slot[(size_t)c * PF + q] = -1;
    }
    n_pdu[c] = total;
}

__global__ __launch_bounds__(128) void
k_chain_pdu_gather(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym, const int32_t* __restrict__ sync_pos,
                   const int32_t* __restrict__ pdu_slot, const int32_t* __restrict__ pdu_info, int F, int PF, int PB,
                   int16_t* __restrict__ llr, uint8_t* __restrict__ valid) {
    const int e = blockIdx.x, b = blockIdx.y + 1, j = threadIdx.x;
    const int slot = pdu_slot[e];
    const int ch = e / PF;
    const size_t o = (size_t)e * PB + (b - 1);
    const int end = slot >= 0 ? pdu_info[(size_t)e * 4 + 1] : 0;
    const int cnt = counts[ch] < (int)max_sym ? counts[ch] : (int)max_sym;
    // the block's last payload dibit decides whether the whole block lies inside this call's records
    const int n_last = 56 + 98 * b + 97;
    const bool ok = slot >= 0 && b < end && (slot >= 0 ? sync_pos[slot] - 23 + n_last + n_last / 35 < cnt : false);
    if (j == 0) {
        valid[o] = ok ? 1 : 0;
    }
    if (j >= 98) {
        return;
    }
    int l0 = 0, l1 = 0;
    if (ok) {
        const int nn = 56 + 98 * b + j;
        const uint8_t* q = rec + ((size_t)ch * max_sym + (size_t)(sync_pos[slot] - 23 + nn + nn / 35)) * 10;
        l0 = (int16_t)((uint16_t)q[2] | ((uint16_t)q[3] << 8));
        l1 = (int16_t)((uint16_t)q[4] | ((uint16_t)q[5] << 8));
    }
    llr[o * 196 + 2 * j] = (int16_t)l0;
    llr[o * 196 + 2 * j + 1] = (int16_t)l1;
}

__device__ __forceinline__ bool
pdu_crc16_ok(const uint8_t* b12) { // crc16_lb_bridge(bits, 80) == 0 (src/protocol/p25/p25_crc.c:18-36): CRC-CCITT over 80 bits, inverted
    uint32_t crc = 0;
    for (int i = 0; i < 80; i++) {
        const uint32_t bit = (b12[i >> 3] >> (7 - (i & 7))) & 1;
        crc = (((crc >> 15) & 1) ^ bit) ? ((crc << 1) ^ 0x1021) & 0xFFFF : (crc << 1) & 0xFFFF;
    }
    crc ^= 0xFFFF;
    return crc == (((uint32_t)b12[10] << 8) | b12[11]);
}

// The reference's second header fallback (p25_mpdu_try_combined_header, :336-360): when the first header block fails its CRC16 the three
// blocks read are three repetitions of the header; if neither of the other two passes on its own, the three blocks' LLRs are added
// position by position (saturating at int16, saturating_llr_add :36-45) and the sum goes through the half-rate list decoder.  One
// workgroup per entry: the header block's own LLRs come from the records (payload dibit 56 + j of the frame), the other two from the
// gather above.  wanted[e] says whether the entry takes this road at all.
__global__ __launch_bounds__(128) void
k_chain_pdu_combine(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym, const int32_t* __restrict__ sync_pos,
                    const int32_t* __restrict__ pdu_slot, const int32_t* __restrict__ pdu_info, const int16_t* __restrict__ llr,
                    const uint8_t* __restrict__ valid, const uint8_t* __restrict__ blocks12, int PF, int PB, int16_t* __restrict__ hllr,
                    uint8_t* __restrict__ wanted) {
    const int e = blockIdx.x, j = threadIdx.x;
    const int slot = pdu_slot[e];
    const int ch = e / PF;
    bool want = slot >= 0 && pdu_info[(size_t)e * 4 + 0] == 0 && pdu_info[(size_t)e * 4 + 1] >= 3 && PB >= 2 && valid[(size_t)e * PB] != 0
                && valid[(size_t)e * PB + 1] != 0;
    if (want) {
        const int cnt = counts[ch] < (int)max_sym ? counts[ch] : (int)max_sym;
        const int n_last = 56 + 97;
        want = sync_pos[slot] - 23 >= 0 && sync_pos[slot] - 23 + n_last + n_last / 35 < cnt
               && !pdu_crc16_ok(blocks12 + (size_t)e * PB * 12) && !pdu_crc16_ok(blocks12 + ((size_t)e * PB + 1) * 12);
    }
    if (j == 0) {
        wanted[e] = want ? 1 : 0;
    }
    if (j >= 98) {
        return;
    }
    int a0 = 0, a1 = 0;
    if (want) {
        const int nn = 56 + j;
        const uint8_t* q = rec + ((size_t)ch * max_sym + (size_t)(sync_pos[slot] - 23 + nn + nn / 35)) * 10;
        a0 = (int16_t)((uint16_t)q[2] | ((uint16_t)q[3] << 8));
        a1 = (int16_t)((uint16_t)q[4] | ((uint16_t)q[5] << 8));
        for (int rep = 0; rep < 2; rep++) {
            const int16_t* l = llr + ((size_t)e * PB + rep) * 196 + 2 * j;
            a0 += l[0];
            a0 = a0 > 32767 ? 32767 : (a0 < -32768 ? -32768 : a0);
            a1 += l[1];
            a1 = a1 > 32767 ? 32767 : (a1 < -32768 ? -32768 : a1);
        }
    }
    hllr[(size_t)e * 196 + 2 * j] = (int16_t)a0;
    hllr[(size_t)e * 196 + 2 * j + 1] = (int16_t)a1;
}

// data blocks (block_idx >= 1) take the list decoder's first candidate (p25_mpdu_decode_r12_block(), :236-239), which is not always
// p25_12_soft_llr()'s path: the two break ties at the unprotected tail differently
__global__ void
k_chain_pdu_take_first(const uint8_t* __restrict__ cand16, const int32_t* __restrict__ counts, int n_blocks, uint8_t* __restrict__ blocks12,
                       int32_t* __restrict__ metric) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) {
        return;
    }
    const uint8_t* c = cand16 + (size_t)i * 8 * 16;
    const bool have = counts[i] > 0;
    for (int b = 0; b < 12; b++) {
        blocks12[(size_t)i * 12 + b] = have ? c[b] : 0;
    }
    metric[i] = have ? (int32_t)((uint32_t)c[12] | ((uint32_t)c[13] << 8) | ((uint32_t)c[14] << 16) | ((uint32_t)c[15] << 24)) : -1;
}

// confirmed data (the first header has a good CRC16, A/N = 1, format 0x16: ctx->r34, :319): which block entries go through the rate 3/4
// list decoder
__global__ void
k_chain_pdu_r34_wanted(const int32_t* __restrict__ pdu_slot, const uint8_t* __restrict__ pdu_hdr, const int32_t* __restrict__ pdu_info,
                       const uint8_t* __restrict__ valid, int n_blocks, int PB, uint8_t* __restrict__ wanted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) {
        return;
    }
    const int e = i / PB;
    const uint8_t* h = pdu_hdr + (size_t)e * 12;
    const bool r34 = pdu_slot[e] >= 0 && pdu_info[(size_t)e * 4] != 0 && ((h[0] >> 6) & 1) && (h[0] & 0x1F) == 0x16;
    wanted[i] = (r34 && valid[i]) ? 1 : 0;
}

__device__ __forceinline__ uint32_t
pdu_crc9(const uint8_t* b18) { // p25_mpdu_candidate_crc9(): 7 DBSN bits + the 16 payload bytes, ComputeCrc9Bit (dmr_utils.c:410-435)
    uint32_t crc = 0;
    for (int i = 0; i < 135; i++) {
        const int bit = i < 7 ? (b18[0] >> (7 - i)) & 1 : (b18[2 + ((i - 7) >> 3)] >> (7 - ((i - 7) & 7))) & 1;
        crc = (((crc >> 8) & 1) ^ (uint32_t)bit) ? ((crc << 1) ^ 0x059u) : (crc << 1);
    }
    return (crc & 0x1FFu) ^ 0x1FFu;
}

// p25_mpdu_select_mbf34_candidate(:176-187): the first candidate whose CRC9 matches, else the cheapest
__global__ void
k_chain_pdu_r34_select(const uint8_t* __restrict__ cand24, const int32_t* __restrict__ counts, const uint8_t* __restrict__ wanted,
                       int n_blocks, uint8_t* __restrict__ blocks18, uint8_t* __restrict__ crc9_ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) {
        return;
    }
    uint8_t* o = blocks18 + (size_t)i * 18;
    int ok = 0, pick = 0;
    const int n = wanted[i] ? counts[i] : 0;
    for (int k = 0; k < n && !ok; k++) {
        const uint8_t* b = cand24 + ((size_t)i * 8 + k) * 24;
        if (pdu_crc9(b) == ((((uint32_t)b[0] & 1u) << 8) | b[1])) {
            ok = 1;
            pick = k;
        }
    }
    for (int b = 0; b < 18; b++) {
        o[b] = n > 0 ? cand24[((size_t)i * 8 + pick) * 24 + b] : 0;
    }
    crc9_ok[i] = (uint8_t)ok;
}

__global__ void
k_chain_pdu_finish(const int32_t* __restrict__ pdu_slot, const uint8_t* __restrict__ blocks12, const uint8_t* __restrict__ valid,
                   const uint8_t* __restrict__ blocks18, const uint8_t* __restrict__ hcand16, const int32_t* __restrict__ hcount,
                   const uint8_t* __restrict__ hwanted, int n_entries, int PB, uint8_t* __restrict__ pdu_hdr,
                   int32_t* __restrict__ pdu_info) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries || pdu_slot[e] < 0) {
        return;
    }
    uint8_t* h = pdu_hdr + (size_t)e * 12;
    int hdr_ok = pdu_info[(size_t)e * 4 + 0];
    const int end = pdu_info[(size_t)e * 4 + 1];
    const uint8_t* blk = blocks12 + (size_t)e * PB * 12;
    const uint8_t* vld = valid + (size_t)e * PB;
    const bool r34 = hdr_ok && ((h[0] >> 6) & 1) && (h[0] & 0x1F) == 0x16; // (from the FIRST header, as ctx->r34)
    int flags = 0; // 1 header taken from repetition 1, 2 from repetition 2, 4 confirmed data: the blocks are rate 3/4 (blocks18),
                   // 8 a block lies beyond the call's records or beyond PB, 16 header unusable (its CRC16 fails whatever was tried),
                   // 32 header from the three repetitions' summed LLRs, 64 header = bitwise majority of the three decoded repetitions
    if (!hdr_ok) { // the header said nothing: the reference has read three blocks and tries the other two as header repetitions
        for (int rep = 1; rep <= 2 && !hdr_ok; rep++) {
            if (rep < end && rep - 1 < PB && vld[rep - 1] && pdu_crc16_ok(blk + (size_t)(rep - 1) * 12)) {
                for (int i = 0; i < 12; i++) {
                    h[i] = blk[(size_t)(rep - 1) * 12 + i];
                }
                hdr_ok = 1;
                flags |= rep;
            }
        }
        if (!hdr_ok && hwanted[e]) {
            // p25_mpdu_try_combined_header (:336-360): the first candidate of the summed LLRs with a good CRC16
            const int nc = hcount[e] < 8 ? hcount[e] : 8;
            for (int k = 0; k < nc && !hdr_ok; k++) {
                const uint8_t* cb = hcand16 + ((size_t)e * 8 + k) * 16;
                if (pdu_crc16_ok(cb)) {
                    for (int i = 0; i < 12; i++) {
                        h[i] = cb[i];
                    }
                    hdr_ok = 1;
                    flags |= 32;
                }
            }
            if (!hdr_ok) { // p25_mpdu_rebuild_header_from_majority (:362-379): two of three, bit by bit; kept whether or not its CRC16 holds
                for (int i = 0; i < 12; i++) {
                    const uint32_t a = h[i], b = blk[i], cc = blk[12 + i];
                    h[i] = (uint8_t)((a & b) | (a & cc) | (b & cc));
                }
                flags |= 64;
                hdr_ok = pdu_crc16_ok(h) ? 1 : 0;
            }
        }
        if (!hdr_ok) {
            flags |= 16;
        }
    }
    const int an = (h[0] >> 6) & 1, fmt = h[0] & 0x1F, blks = h[6] & 0x7F;
    (void)an;
    (void)fmt;
    if (r34) {
        flags |= 4;
    }
    int crc32_ok = 0;
    const int nd = end - 1; // data blocks read
    bool all = nd <= PB;
    for (int b = 0; b < nd && b < PB; b++) {
        all = all && vld[b] != 0;
    }
    if (!all) {
        flags |= 8;
    }
    if (hdr_ok && !(flags & (1 | 2 | 4 | 32 | 64)) && all) { // p25_mpdu_handle_rate12(): CRC32 over the data blocks but the last four bytes
        if (blks == 0) {
            crc32_ok = 1;
        } else if (blks == nd) {
            const int len = 96 * blks - 32;
            uint64_t crc = 0;
            for (int i = 0; i < len; i++) {
                crc <<= 1;
                const int bit = (blk[i >> 3] >> (7 - (i & 7))) & 1;
                if (((crc >> 32) ^ (uint64_t)bit) & 1) {
                    crc ^= 0x04c11db7ull;
                }
            }
            const uint32_t got = (uint32_t)((crc & 0xffffffffull) ^ 0xffffffffull);
            const uint8_t* t = blk + (size_t)blks * 12 - 4;
            const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
            crc32_ok = got == want ? 1 : 0;
        }
    }
    if (r34 && all) { // p25_mpdu_compute_rate34_crc(:501-519): CRC32 over the blocks' 16 payload bytes but the last four
        if (blks > 0 && blks == nd) {
            const uint8_t* b18 = blocks18 + (size_t)e * PB * 18;
            const int len = 128 * blks - 32;
            uint64_t crc = 0;
            for (int i = 0; i < len; i++) {
                crc <<= 1;
                const int bit = (b18[(size_t)(i >> 7) * 18 + 2 + ((i & 127) >> 3)] >> (7 - (i & 7))) & 1;
                if (((crc >> 32) ^ (uint64_t)bit) & 1) {
                    crc ^= 0x04c11db7ull;
                }
            }
            const uint32_t got = (uint32_t)((crc & 0xffffffffull) ^ 0xffffffffull);
            const uint8_t* t = b18 + (size_t)(blks - 1) * 18 + 14;
            const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
            crc32_ok = got == want ? 1 : 0;
        } else if (blks == 0) {
            crc32_ok = 1; // (crc_extracted = crc_computed = 0, :515-519)
        }
    }
    pdu_info[(size_t)e * 4 + 0] = hdr_ok;
    pdu_info[(size_t)e * 4 + 2] = flags;
    pdu_info[(size_t)e * 4 + 3] = crc32_ok;
}

// NXDN voice stage bookkeeping (nxdn_voice(): the LICH's profile says which of a frame's four 36-dibit fields are voice): the
// first vf sync slots of every channel feed the voice gather; field v of slot (c, j) is skipped unless the LICH parity held, the
// frame is complete and the LICH value announces voice in that half.
__global__ void
k_nxdn_voice_select(const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, const uint8_t* __restrict__ lich,
                    const uint8_t* __restrict__ valid, int n_channels, int my, int vf, int32_t* __restrict__ v_pos,
                    int32_t* __restrict__ v_n, uint8_t* __restrict__ skip4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_channels * vf) {
        return;
    }
    const int c = i / vf, j = i - c * vf;
    v_pos[i] = sync_pos[(size_t)c * my + j];
    if (j == 0) {
        v_n[c] = n_sync[c] < vf ? n_sync[c] : vf;
    }
    const int l = lich[(size_t)c * my + j];
    const int l7 = l & 0x7F;
    const bool good = (l & 0x80) != 0 && valid[(size_t)c * my + j] != 0;
    const bool both = l7 == 0x36 || l7 == 0x37 || l7 == 0x56 || l7 == 0x57 || l7 == 0x46 || l7 == 0x76 || l7 == 0x77;
    const bool first = l7 == 0x34 || l7 == 0x35 || l7 == 0x54 || l7 == 0x55 || l7 == 0x75;
    const bool last = l7 == 0x32 || l7 == 0x33 || l7 == 0x52 || l7 == 0x53 || l7 == 0x72 || l7 == 0x73;
    for (int v = 0; v < 4; v++) {
        const bool voiced = good && (both || (first && v < 2) || (last && v >= 2));
        skip4[(size_t)i * 4 + v] = voiced ? 0 : 1;
    }
}

// DMR / NXDN48 chain: which accepted syncs are decoded in this call and which wait for the next one.  A row holds the T records
// carried from the previous call, then this call's new ones; a sync (position = row index of its last symbol, with the 90-dibit
// hand-over the loop made for it) is decoded when the T records behind it are in the row, i.e. when its position is below the
// number of new records - the rest go to the carry list, re-based to the next row.  One workgroup per channel.
__global__ __launch_bounds__(64) void
k_fsk4_chain_syncs(const int32_t* __restrict__ c_pos, const uint8_t* __restrict__ c_pat, const uint8_t* __restrict__ c_pre,
                   const uint8_t* __restrict__ c_prel, const int32_t* __restrict__ c_n, int myc, const int32_t* __restrict__ s_pos,
                   const uint8_t* __restrict__ s_pat, const uint8_t* __restrict__ s_pre, const uint8_t* __restrict__ s_prel,
                   const int32_t* __restrict__ s_n, int my, const int32_t* __restrict__ n_new, int T, int flush,
                   int32_t* __restrict__ d_pos, uint8_t* __restrict__ d_pat, uint8_t* __restrict__ d_pre, uint8_t* __restrict__ d_prel,
                   int32_t* __restrict__ d_n, int myd, int32_t* __restrict__ o_pos, uint8_t* __restrict__ o_pat,
                   uint8_t* __restrict__ o_pre, uint8_t* __restrict__ o_prel, int32_t* __restrict__ o_n,
                   int32_t* __restrict__ dropped, const float* __restrict__ c_thr, const float* __restrict__ s_thr,
                   float* __restrict__ d_thr, float* __restrict__ o_thr) {
    // (c_thr / s_thr / d_thr / o_thr: optional [..][5] thresholds every sync left, filed with it - M17)
    const int c = blockIdx.x, lane = threadIdx.x;
    const int nc = c_n[c] < myc ? c_n[c] : myc, nsn = s_n[c] < my ? s_n[c] : my;
    const int limit = flush ? 0x7FFFFFFF : n_new[c], shift = n_new[c];
    int kd = 0, ko = 0, lost = 0;
    for (int i = 0; i < nc + nsn; i++) {
        const bool carried = i < nc;
        const int j = carried ? i : i - nc;
        const int p = carried ? c_pos[(size_t)c * myc + j] : s_pos[(size_t)c * my + j] + T;
        const uint8_t pat = carried ? c_pat[(size_t)c * myc + j] : s_pat[(size_t)c * my + j];
        const uint8_t* pre = carried ? c_pre + ((size_t)c * myc + j) * 90 : s_pre + ((size_t)c * my + j) * 90;
        const uint8_t* prel = carried ? c_prel + ((size_t)c * myc + j) * 90 : s_prel + ((size_t)c * my + j) * 90;
        uint8_t *qp, *qr;
        const float* th = s_thr ? (carried ? c_thr + ((size_t)c * myc + j) * 5 : s_thr + ((size_t)c * my + j) * 5) : nullptr;
        if (p < limit) {
            if (kd >= myd) {
                lost++; // more accepted syncs than decode slots: counted, never silent
                continue;
            }
            if (lane == 0) {
                d_pos[(size_t)c * myd + kd] = p;
                d_pat[(size_t)c * myd + kd] = pat;
            }
            if (th && lane < 5) {
                d_thr[((size_t)c * myd + kd) * 5 + lane] = th[lane];
            }
            qp = d_pre + ((size_t)c * myd + kd) * 90;
            qr = d_prel + ((size_t)c * myd + kd) * 90;
            kd++;
        } else {
            if (ko >= myc) {
                lost++;
                continue;
            }
            if (lane == 0) {
                o_pos[(size_t)c * myc + ko] = p - shift;
                o_pat[(size_t)c * myc + ko] = pat;
            }
            if (th && lane < 5) {
                o_thr[((size_t)c * myc + ko) * 5 + lane] = th[lane];
            }
            qp = o_pre + ((size_t)c * myc + ko) * 90;
            qr = o_prel + ((size_t)c * myc + ko) * 90;
            ko++;
        }
        for (int k = lane; k < 90; k += 64) {
            qp[k] = pre[k];
            qr[k] = prel[k];
        }
    }
    if (lane == 0) {
        d_n[c] = kd;
        o_n[c] = ko;
        dropped[c] += lost;
    }
}

__global__ void
k_u8_shr1(const uint8_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = in[i] >> 1;
    }
}

__device__ __forceinline__ bool
crc16_clean(const uint8_t* b) { // p25_crc.c:18-36 over 10 bytes against the next two
    unsigned crc = 0;
    for (int k = 0; k < 10; k++) {
        const unsigned v = b[k];
#pragma unroll
        for (int j = 7; j >= 0; j--) {
            const unsigned bit = (v >> j) & 1u;
            crc = (((crc >> 15) & 1u) ^ bit) ? (((crc << 1) ^ 0x1021u) & 0xFFFFu) : ((crc << 1) & 0xFFFFu);
        }
    }
    crc ^= 0xFFFFu;
    return crc == (((unsigned)b[10] << 8) | b[11]);
}

__global__ void
k_tsbk_select(const uint8_t* __restrict__ cand, const int32_t* __restrict__ counts, size_t n, uint8_t* __restrict__ out12,
              uint8_t* __restrict__ crc_ok, uint8_t* __restrict__ sel_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* c = cand + i * 8 * 16; // [8] x {12 bytes, u32 metric}
    const int cnt = counts[i];
    int sel = 0, ok = 0;
    for (int k = 0; k < cnt && k < 8; k++) {
        if (crc16_clean(c + 16 * k)) {
            sel = k;
            ok = 1;
            break;
        }
    }
    for (int k = 0; k < 12; k++) {
        out12[i * 12 + k] = cnt > 0 ? c[16 * sel + k] : 0;
    }
    crc_ok[i] = (uint8_t)ok;
    if (sel_out) {
        sel_out[i] = (uint8_t)sel;
    }
}

} // namespace

static thread_local DdnSel g_sel = {nullptr, nullptr, 1};
extern "C" void
ddn_sel_set(const int32_t* list, const int32_t* count) {
    g_sel.list = list;
    g_sel.count = count;
}
extern "C" void
ddn_sel_clear(void) {
    g_sel.list = nullptr;
    g_sel.count = nullptr;
}
extern "C" DdnSel
ddn_sel_for(int per_slot) {
    DdnSel r = g_sel;
    r.per_slot = per_slot > 0 ? per_slot : 1;
    return r;
}

extern "C" hipError_t
ddn_dev_chain_carry(const uint8_t* rec_prev, const uint8_t* fl_prev, const int32_t* cnt_prev, int have_prev, uint8_t* rec_cur,
                    uint8_t* fl_cur, size_t stride_sym, int T, int n_channels, hipStream_t st) {
    if (n_channels <= 0 || T <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_carry, dim3(4, (unsigned)n_channels), dim3(256), 0, st, rec_prev, fl_prev, cnt_prev, have_prev, rec_cur,
                       fl_cur, stride_sym, T, n_channels);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pcm_compact(const int32_t* result5, const float* pcm, int n_slots, long capacity, int32_t* block_cnt, int32_t* block_off,
                          float* dense, int32_t* slot_of, int32_t* total, hipStream_t st) {
    if (n_slots <= 0) {
        return hipSuccess;
    }
    const int nb = (n_slots + 1023) / 1024;
    hipLaunchKernelGGL(k_pcm_count, dim3((unsigned)nb), dim3(256), 0, st, result5, n_slots, block_cnt);
    hipLaunchKernelGGL(k_pcm_offsets, dim3(1), dim3(64), 0, st, block_cnt, nb, block_off, total);
    hipLaunchKernelGGL(k_pcm_compact, dim3((unsigned)nb), dim3(256), 0, st, result5, pcm, n_slots, block_off, capacity, dense, slot_of);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pack2(const uint8_t* rec, const uint8_t* fl, size_t n, uint8_t* out2, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_chain_pack2, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, rec, fl, n,
                       reinterpret_cast<uint16_t*>(out2));
    return hipGetLastError();
}

// a few words to zero on a stream: a kernel of our own instead of hipMemsetAsync.  Measured (round 5, rocprofv3 kernel trace of the
// headline step): the runtime's fill kernel for the 32-byte list-count reset sat on the decode stream for 2.03 ms - the whole length
// of the front-end kernel running beside it - and held back every decode kernel behind it, so the receive loop (which waits for the
// previous call's decode) started 1.5 ms late in every step.
__global__ void
k_zero_words(int32_t* __restrict__ p, int n, int32_t value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        p[i] = value;
    }
}

extern "C" hipError_t
ddn_dev_zero_words(int32_t* p, int n, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, p, n, 0);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_fill_words(int32_t* p, int n, int32_t value, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, p, n, value);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_counts(const int32_t* cnt_new, int T, int n_channels, int flush, int32_t* cnt_scan, int32_t* cnt_full, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_counts, dim3((unsigned)((n_channels + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, cnt_new, T, n_channels, flush,
                       cnt_scan, cnt_full);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_events(const int32_t* list_prev, const int32_t* data_prev, const int32_t* n_prev, const int32_t* new_prev, int have_prev,
                     const int32_t* ev_new, const int32_t* evd_new, const int32_t* n_new, int E, int EL, int T, int n_channels,
                     int32_t* list_cur, int32_t* data_cur, int32_t* n_cur, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_events, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, list_prev, data_prev, n_prev, new_prev,
                       have_prev, ev_new, evd_new, n_new, E, EL, T, n_channels, list_cur, data_cur, n_cur);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_frames(const int32_t* list, const int32_t* data, const int32_t* n_list, int EL, const int32_t* sync_pos,
                     const int32_t* n_syncs, int n_channels, int F, int off0, int off1, int off2, int32_t* nid4, uint8_t* tsbk,
                     uint8_t* tsbk_crc, uint8_t* cls, int32_t* lists, int32_t* list_n, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_frames, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, list, data, n_list, EL, sync_pos,
                       n_syncs, n_channels, F, off0, off1, off2, nid4, tsbk, tsbk_crc, cls, lists, list_n);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_index(const int32_t* list, const int32_t* data, const int32_t* n_list, int EL, const int32_t* sync_pos,
                        const int32_t* n_syncs, const int32_t* nid4, int n_channels, int F, int off0, int PF, int32_t* pdu_slot,
                        uint8_t* pdu_hdr, int32_t* pdu_info, int32_t* n_pdu, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_index, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, list, data, n_list, EL, sync_pos,
                       n_syncs, nid4, n_channels, F, off0, PF, pdu_slot, pdu_hdr, pdu_info, n_pdu);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos, const int32_t* pdu_slot,
                         const int32_t* pdu_info, int n_channels, int F, int PF, int PB, int16_t* llr, uint8_t* valid, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_gather, dim3((unsigned)(n_channels * PF), (unsigned)PB), dim3(128), 0, st, rec, counts, max_sym, sync_pos,
                       pdu_slot, pdu_info, F, PF, PB, llr, valid);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_finish(const int32_t* pdu_slot, const uint8_t* blocks12, const uint8_t* valid, const uint8_t* blocks18,
                         const uint8_t* hcand16, const int32_t* hcount, const uint8_t* hwanted, int n_entries, int PB, uint8_t* pdu_hdr,
                         int32_t* pdu_info, hipStream_t st) {
    if (n_entries <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_finish, dim3((unsigned)((n_entries + 63) / 64)), dim3(64), 0, st, pdu_slot, blocks12, valid, blocks18,
                       hcand16, hcount, hwanted, n_entries, PB, pdu_hdr, pdu_info);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_combine(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos, const int32_t* pdu_slot,
                          const int32_t* pdu_info, const int16_t* llr, const uint8_t* valid, const uint8_t* blocks12, int n_entries, int PF,
                          int PB, int16_t* hllr, uint8_t* wanted, hipStream_t st) {
    if (n_entries <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_combine, dim3((unsigned)n_entries), dim3(128), 0, st, rec, counts, max_sym, sync_pos, pdu_slot, pdu_info,
                       llr, valid, blocks12, PF, PB, hllr, wanted);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_take_first(const uint8_t* cand16, const int32_t* counts, int n_blocks, uint8_t* blocks12, int32_t* metric, hipStream_t st) {
    if (n_blocks <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_take_first, dim3((unsigned)((n_blocks + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, cand16, counts, n_blocks, blocks12,
                       metric);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_r34_wanted(const int32_t* pdu_slot, const uint8_t* pdu_hdr, const int32_t* pdu_info, const uint8_t* valid, int n_blocks,
                             int PB, uint8_t* wanted, hipStream_t st) {
    if (n_blocks <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_r34_wanted, dim3((unsigned)((n_blocks + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, pdu_slot, pdu_hdr, pdu_info, valid,
                       n_blocks, PB, wanted);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_chain_pdu_r34_select(const uint8_t* cand24, const int32_t* counts, const uint8_t* wanted, int n_blocks, uint8_t* blocks18,
                             uint8_t* crc9_ok, hipStream_t st) {
    if (n_blocks <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_pdu_r34_select, dim3((unsigned)((n_blocks + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, cand24, counts, wanted, n_blocks,
                       blocks18, crc9_ok);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_nxdn_voice_select(const int32_t* sync_pos, const int32_t* n_sync, const uint8_t* lich, const uint8_t* valid, int n_channels,
                          int my, int vf, int32_t* v_pos, int32_t* v_n, uint8_t* skip4, hipStream_t st) {
    if (n_channels <= 0 || vf <= 0) {
        return hipSuccess;
    }
    const int n = n_channels * vf;
    hipLaunchKernelGGL(k_nxdn_voice_select, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sync_pos, n_sync, lich, valid, n_channels,
                       my, vf, v_pos, v_n, skip4);
    return hipGetLastError();
}

extern "C" hipError_t ddn_dev_fsk4_chain_syncs_thr(const int32_t* c_pos, const uint8_t* c_pat, const uint8_t* c_pre, const uint8_t* c_prel,
                                                   const int32_t* c_n, int myc, const int32_t* s_pos, const uint8_t* s_pat, const uint8_t* s_pre,
                                                   const uint8_t* s_prel, const int32_t* s_n, int my, const int32_t* n_new, int T, int flush,
                                                   int32_t* d_pos, uint8_t* d_pat, uint8_t* d_pre, uint8_t* d_prel, int32_t* d_n, int myd,
                                                   int32_t* o_pos, uint8_t* o_pat, uint8_t* o_pre, uint8_t* o_prel, int32_t* o_n,
                                                   int32_t* dropped, int n_channels, const float* c_thr, const float* s_thr, float* d_thr,
                                                   float* o_thr, hipStream_t st);
extern "C" hipError_t
ddn_dev_fsk4_chain_syncs(const int32_t* c_pos, const uint8_t* c_pat, const uint8_t* c_pre, const uint8_t* c_prel, const int32_t* c_n, int myc,
                         const int32_t* s_pos, const uint8_t* s_pat, const uint8_t* s_pre, const uint8_t* s_prel, const int32_t* s_n, int my,
                         const int32_t* n_new, int T, int flush, int32_t* d_pos, uint8_t* d_pat, uint8_t* d_pre, uint8_t* d_prel,
                         int32_t* d_n, int myd, int32_t* o_pos, uint8_t* o_pat, uint8_t* o_pre, uint8_t* o_prel, int32_t* o_n,
                         int32_t* dropped, int n_channels, hipStream_t st) {
    return ddn_dev_fsk4_chain_syncs_thr(c_pos, c_pat, c_pre, c_prel, c_n, myc, s_pos, s_pat, s_pre, s_prel, s_n, my, n_new, T, flush, d_pos, d_pat,
                                        d_pre, d_prel, d_n, myd, o_pos, o_pat, o_pre, o_prel, o_n, dropped, n_channels, nullptr, nullptr, nullptr,
                                        nullptr, st);
}

// the same with the thresholds every sync left filed beside it ([..][5] floats: carried in, loop's, decode list, carried out)
extern "C" hipError_t
ddn_dev_fsk4_chain_syncs_thr(const int32_t* c_pos, const uint8_t* c_pat, const uint8_t* c_pre, const uint8_t* c_prel, const int32_t* c_n,
                             int myc, const int32_t* s_pos, const uint8_t* s_pat, const uint8_t* s_pre, const uint8_t* s_prel,
                             const int32_t* s_n, int my, const int32_t* n_new, int T, int flush, int32_t* d_pos, uint8_t* d_pat,
                             uint8_t* d_pre, uint8_t* d_prel, int32_t* d_n, int myd, int32_t* o_pos, uint8_t* o_pat, uint8_t* o_pre,
                             uint8_t* o_prel, int32_t* o_n, int32_t* dropped, int n_channels, const float* c_thr, const float* s_thr,
                             float* d_thr, float* o_thr, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_fsk4_chain_syncs, dim3((unsigned)n_channels), dim3(64), 0, st, c_pos, c_pat, c_pre, c_prel, c_n, myc, s_pos, s_pat,
                       s_pre, s_prel, s_n, my, n_new, T, flush, d_pos, d_pat, d_pre, d_prel, d_n, myd, o_pos, o_pat, o_pre, o_prel, o_n, dropped,
                       c_thr, s_thr, d_thr, o_thr);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_u8_shr1(const uint8_t* in, size_t n, uint8_t* out, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_u8_shr1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_tsbk_select(const uint8_t* cand, const int32_t* counts, size_t n, uint8_t* out12, uint8_t* crc_ok, uint8_t* sel, hipStream_t st) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_tsbk_select, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cand, counts, n, out12, crc_ok, sel);
    return hipGetLastError();
}
