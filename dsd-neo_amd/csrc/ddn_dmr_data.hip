// ddn_dmr_data.hip - the DMR chain's data-burst and embedded-signalling stages (include/ddn_chain.h: ddn_fsk4_chain_results, d_dmr_data_* /
// d_dmr_r34_* / d_dmr_emb_*).
//
// Which bursts the reference hands to dmr_data_burst_handler() is decided inside the receive loop (ddn_fsk4h_dev.h): every burst
// dmr_data_dispatch_burst() (src/protocol/dmr/dmr_data.c:262-280) lets through - found by the sync search or read inside dmrBS() -
// leaves a kind-6 event with VC = 0 at the burst's last symbol, carrying the time slot.  This file turns those events into the
// handler's FEC-level results (src/protocol/dmr/dmr_dburst.c):
//   k_dmr_data_select    one wavefront per channel: the dispatched bursts of this call in air order
//   k_dmr_data_gather    one workgroup per burst: slot type, the 196 info bits, the 98 info dibits + reliabilities (rel98)
//   (Golay(20,8), BPTC(196,96), RS(12,9): ddn_fec3.hip)
//   k_dmr_data_prep      data type, the 12 BPTC bytes, the masked RS(12,9) codeword of a full link control (VLC / TLC)
//   k_dmr_data_finish    the CRC the profile names (dmr_dburst_profile_resolve(), :105-174): RS(12,9) for VLC / TLC
//                        (ComputeAndCorrectFullLinkControlCrc(), dmr_utils.c:291-351: the corrected bytes replace the received
//                        ones), CRC-CCITT with the type's mask, CRC9 of a confirmed rate 1/2 / rate 1 block; marks the rate 3/4 bursts
//   (rate 3/4: hard / soft / list-32 decoders of ddn_trellis.hip, then k_dmr_r34_pick there)
//   k_dmr_emb_collect    one thread per channel: the 48 sync-field bits of every burst read under VC 2..6 filed as embedded
//                        signalling (read_dmr_bs_sync_segment(), dmr_bs.c:161-180; the store streams from call to call), and at each
//                        voice burst with VC 6 the 8 x 16 matrix dmr_dburst_handle_emb() (:363-394) builds from bursts B..E
//   k_dmr_emb_finish     the 5-bit checksum of the 77 bits BPTC(128,77) returns (ComputeCrc5Bit(), dmr_utils.c:365-392)
// Not here: everything that needs the protocol layer's running state (state->data_conf_data, the DBSN sequence, the block
// assembler: src/protocol/dmr/dmr_block.c) - where the reference's choice depends on it, both answers are given.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {

// dmr_dburst_profile_resolve(): CRC mask and length per data type
__constant__ uint32_t c_crcmask[12] = {0x6969, 0x969696, 0x999999, 0xA5A5, 0xAAAA, 0x0, 0xCCCC, 0x0F0, 0x1FF, 0x0, 0x10F, 0x3333};
__constant__ uint8_t c_crclen[12] = {16, 24, 24, 16, 16, 0, 16, 9, 9, 0, 9, 16};

// dpre: a burst the sync search found takes its first 90 dibits (and their reliabilities) from the hand-over the loop made at the sync
// (dmr_data_sync() reads them from the payload / soft history, dmr_data.c:56-100) - the sync's slot in this call's decode list, or
// n_channels * max_syncs + its index in the list carried to the next call; -1: a burst read inside dmrBS(), all 144 dibits live
__global__ __launch_bounds__(64) void
k_dmr_data_select(const int32_t* __restrict__ events, const int32_t* __restrict__ n_events, int max_events, int carry, int n_channels,
                  int max_bursts, const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, int max_syncs,
                  const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_n, int max_out, const int32_t* __restrict__ n_new,
                  int32_t* __restrict__ dstart, uint8_t* __restrict__ dslot, int32_t* __restrict__ dpre, int32_t* __restrict__ dn) {
    // one wavefront per channel: lane = event of the current chunk of 64, its place in the list = the dispatched events ahead of it
    const int ch = blockIdx.x, lane = threadIdx.x;
    int n = 0;
    const int ne = n_events[ch] < max_events ? n_events[ch] : max_events;
    const int ns = n_sync[ch] < max_syncs ? n_sync[ch] : max_syncs, no = out_n[ch] < max_out ? out_n[ch] : max_out;
    for (int base = 0; base < ne; base += 64) {
        const int i = base + lane;
        int4 e = make_int4(0, 0, 0, 0);
        if (i < ne) {
            e = *reinterpret_cast<const int4*>(events + ((size_t)ch * max_events + i) * 4);
        }
        const bool hit = i < ne && e.y == 6 && (e.w & 0xFFFF) == 0;
        const unsigned long long m = __ballot(hit);
        const int k = n + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && k < max_bursts) {
            dstart[(size_t)ch * max_bursts + k] = carry + e.x - 143;
            dslot[(size_t)ch * max_bursts + k] = (uint8_t)((e.w >> 16) & 1);
            int pre = -1;
            const int want = carry + e.x - 54;
            for (int j = 0; j < ns; j++) {
                if (sync_pos[(size_t)ch * max_syncs + j] == want) {
                    pre = ch * max_syncs + j;
                }
            }
            for (int j = 0; j < no && pre < 0; j++) {
                if (out_pos[(size_t)ch * max_out + j] == want - n_new[ch]) {
                    pre = n_channels * max_syncs + ch * max_out + j;
                }
            }
            dpre[(size_t)ch * max_bursts + k] = pre;
        }
        n += __popcll(m);
    }
    if (lane == 0) {
        dn[ch] = n < max_bursts ? n : max_bursts;
    }
    for (int k = n + lane; k < max_bursts; k += 64) {
        dstart[(size_t)ch * max_bursts + k] = -1;
        dslot[(size_t)ch * max_bursts + k] = 0xFF;
        dpre[(size_t)ch * max_bursts + k] = -1;
    }
}

__global__ __launch_bounds__(64) void
k_dmr_data_gather(const uint8_t* __restrict__ rec, size_t max_sym, const int32_t* __restrict__ dstart, const int32_t* __restrict__ dpre,
                  const uint8_t* __restrict__ pre90, const uint8_t* __restrict__ prel90, const uint8_t* __restrict__ pre90_out,
                  const uint8_t* __restrict__ prel90_out, long split, int max_bursts, uint8_t* __restrict__ slot_type,
                  uint8_t* __restrict__ info, uint8_t* __restrict__ td98, uint8_t* __restrict__ rel98) {
    const int k = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
    const size_t so = (size_t)ch * max_bursts + k;
    const int start = dstart[so];
    const long ps = start >= 0 ? (long)dpre[so] : -1;
    for (int d = 12 + t; d < 144; d += 64) {
        int dibit = 0, rel = 0;
        if (start >= 0) {
            const uint8_t* rr = rec + ((size_t)ch * max_sym + (size_t)(start + d)) * 10;
            dibit = rr[0] & 3;
            rel = rr[1];
            if (ps >= 0 && d < 90) {
                const bool outl = ps >= split;
                const size_t q = (size_t)(outl ? ps - split : ps) * 90 + d;
                dibit = (outl ? pre90_out[q] : pre90[q]) & 3;
                rel = outl ? prel90_out[q] : prel90[q];
            }
        }
        const uint8_t hi = (uint8_t)((dibit >> 1) & 1), lo = (uint8_t)(dibit & 1);
        if (d < 61) {
            info[so * 196 + 2 * (d - 12)] = hi;
            info[so * 196 + 2 * (d - 12) + 1] = lo;
            td98[so * 98 + (d - 12)] = (uint8_t)dibit;
            rel98[so * 98 + (d - 12)] = (uint8_t)rel;
        } else if (d < 66) {
            slot_type[so * 20 + 2 * (d - 61)] = hi;
            slot_type[so * 20 + 2 * (d - 61) + 1] = lo;
        } else if (d < 90) {
            // the sync word
        } else if (d < 95) {
            slot_type[so * 20 + 10 + 2 * (d - 90)] = hi;
            slot_type[so * 20 + 10 + 2 * (d - 90) + 1] = lo;
        } else {
            info[so * 196 + 98 + 2 * (d - 95)] = hi;
            info[so * 196 + 98 + 2 * (d - 95) + 1] = lo;
            td98[so * 98 + 49 + (d - 95)] = (uint8_t)dibit;
            rel98[so * 98 + 49 + (d - 95)] = (uint8_t)rel;
        }
    }
}

__global__ void
k_dmr_data_prep(const int32_t* __restrict__ dstart, const uint8_t* __restrict__ slot_type, const uint8_t* __restrict__ st_ok,
                const uint8_t* __restrict__ pdu96, int n, uint8_t* __restrict__ type, uint8_t* __restrict__ bytes12,
                uint8_t* __restrict__ cw12) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* st = slot_type + (size_t)i * 20;
    const int ty = (dstart[i] >= 0 && st_ok[i]) ? ((st[4] << 3) | (st[5] << 2) | (st[6] << 1) | st[7]) : 0xFF;
    type[i] = (uint8_t)ty;
    const uint32_t mask = (ty == 1 || ty == 2) ? c_crcmask[ty] : 0;
    for (int b = 0; b < 12; b++) {
        int v = 0;
        for (int j = 0; j < 8; j++) {
            v = (v << 1) | (pdu96[(size_t)i * 96 + 8 * b + j] & 1);
        }
        bytes12[(size_t)i * 12 + b] = (uint8_t)v;
        if (b == 9) {
            v ^= (int)((mask >> 16) & 0xFF);
        } else if (b == 10) {
            v ^= (int)((mask >> 8) & 0xFF);
        } else if (b == 11) {
            v ^= (int)(mask & 0xFF);
        }
        cw12[(size_t)i * 12 + b] = (ty == 1 || ty == 2) ? (uint8_t)v : 0;
    }
}

__device__ __forceinline__ uint32_t
crc9_step(uint32_t crc, int bit) { // ComputeCrc9Bit(), dmr_utils.c:410-435
    return (((crc >> 8) & 1) ^ (uint32_t)(bit & 1)) ? ((crc << 1) ^ 0x059u) : (crc << 1);
}

__global__ void
k_dmr_data_finish(const uint8_t* __restrict__ type, const uint8_t* __restrict__ pdu96, const uint8_t* __restrict__ info,
                  const uint8_t* __restrict__ cw12, const uint8_t* __restrict__ rs_result, int n, uint8_t* __restrict__ bytes12,
                  uint8_t* __restrict__ crc, uint8_t* __restrict__ r34_wanted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const int ty = type[i];
    const uint8_t* bits = pdu96 + (size_t)i * 96;
    int flags = 0; // 1 crc_correct with state->data_conf_data = 0, 2 the confirmed block's CRC9 (types 7 and 10), 4 RS(12,9) corrected
    r34_wanted[i] = ty == 8 ? 1 : 0;
    if (ty <= 11) {
        if (ty == 1 || ty == 2) { // full link control: RS(12,9) over the masked bytes; a decodable word replaces the received bytes
            const int res = rs_result[i];
            if (res == 0 || res == 1) {
                const uint32_t mask = c_crcmask[ty];
                for (int b = 0; b < 12; b++) {
                    int v = cw12[(size_t)i * 12 + b];
                    if (b == 9) {
                        v ^= (int)((mask >> 16) & 0xFF);
                    } else if (b == 10) {
                        v ^= (int)((mask >> 8) & 0xFF);
                    } else if (b == 11) {
                        v ^= (int)(mask & 0xFF);
                    }
                    bytes12[(size_t)i * 12 + b] = (uint8_t)v;
                }
                flags |= 1 | (res == 1 ? 4 : 0);
            }
        } else if (ty == 7) { // rate 1/2: unconfirmed blocks carry no CRC; confirmed: DBSN(7) | CRC9 | 10 bytes
            flags |= 1;
            uint32_t ext = 0, c9 = 0;
            for (int b = 7; b < 16; b++) {
                ext = (ext << 1) | (bits[b] & 1);
            }
            ext ^= c_crcmask[7];
            for (int b = 16; b < 96; b++) {
                c9 = crc9_step(c9, bits[b]);
            }
            for (int b = 0; b < 7; b++) {
                c9 = crc9_step(c9, bits[b]);
            }
            if (((c9 & 0x1FF) ^ 0x1FF) == ext) {
                flags |= 2;
            }
        } else if (ty == 8) {
            flags |= 1; // (the confirmed answer comes from the candidate pool: k_dmr_r34_pick)
        } else if (ty == 10) { // rate 1: the 196 info bits as they are; confirmed: DBSN(7) | CRC9 | 22 bytes
            flags |= 1;
            const uint8_t* in = info + (size_t)i * 196;
            uint32_t ext = 0, c9 = 0;
            for (int b = 7; b < 16; b++) {
                ext = (ext << 1) | (in[b] & 1);
            }
            ext ^= c_crcmask[10];
            for (int b = 16; b < 96; b++) {
                c9 = crc9_step(c9, in[b]);
            }
            for (int b = 100; b < 196; b++) {
                c9 = crc9_step(c9, in[b]);
            }
            for (int b = 0; b < 7; b++) {
                c9 = crc9_step(c9, in[b]);
            }
            if (((c9 & 0x1FF) ^ 0x1FF) == ext) {
                flags |= 2;
            }
        } else if (ty != 9) { // PI, CSBK, MBC header / continuation, data header, USBD: CRC-CCITT over the first 80 bits
            const int len = c_crclen[ty];
            uint32_t ext = 0, c16 = 0;
            for (int b = 0; b < len; b++) {
                ext = (ext << 1) | (bits[96 - len + b] & 1);
            }
            ext ^= c_crcmask[ty];
            for (int b = 0; b < 80; b++) { // ComputeCrcCCITT(), dmr_utils.c:253-275
                c16 = ((((c16 >> 15) & 1) ^ (uint32_t)(bits[b] & 1)) ? ((c16 << 1) ^ 0x1021u) : (c16 << 1)) & 0xFFFFu;
            }
            c16 ^= 0xFFFFu;
            if (c16 == ext) {
                flags |= 1;
            }
        }
    }
    crc[i] = (uint8_t)flags;
}

// ---- embedded signalling -----------------------------------------------------------------------------------------------------------
__global__ void
k_dmr_emb_collect(const int32_t* __restrict__ events, const int32_t* __restrict__ n_events, int max_events, int carry,
                  const uint8_t* __restrict__ rec, size_t max_sym, int n_channels, int max_lc, uint8_t* __restrict__ sig /* [B][2][7][48] */,
                  uint8_t* __restrict__ in128, int32_t* __restrict__ lc_pos, int32_t* __restrict__ lc_n) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    int n[2] = {0, 0};
    const int ne = n_events[ch] < max_events ? n_events[ch] : max_events;
    uint8_t* my = sig + (size_t)ch * 2 * 7 * 48;
    for (int i = 0; i < ne; i++) {
        const int32_t* e = events + ((size_t)ch * max_events + i) * 4;
        if (e[1] == 7) { // a burst of dmrBS(): its sync segment was filed under the VC it was read with
            const int slot = e[2] & 1, vcr = (e[3] >> 24) & 0xFF;
            if (vcr > 1 && vcr < 7) {
                const uint8_t* rr = rec + ((size_t)ch * max_sym + (size_t)(carry + e[0] - 143 + 66)) * 10;
                uint8_t* s = my + ((size_t)slot * 7 + (size_t)(vcr - 1)) * 48;
                for (int j = 0; j < 24; j++) {
                    const int d = rr[(size_t)j * 10] & 3;
                    s[2 * j] = (uint8_t)((d >> 1) & 1);
                    s[2 * j + 1] = (uint8_t)(d & 1);
                }
            }
        } else if (e[1] == 6 && (e[3] & 0xFFFF) == 6) { // handle_dmr_bs_slot_vc6_pre_link(): dmr_data_burst_handler(.., 0xEB, ..)
            const int slot = (e[3] >> 16) & 1;
            const int k = n[slot]++;
            if (k < max_lc) {
                const size_t so = ((size_t)ch * 2 + slot) * max_lc + k;
                uint8_t* m = in128 + so * 128;
                int q = 0, burst = 1;
                for (int col = 0; col < 16; col++) {
                    for (int row = 0; row < 8; row++) {
                        m[row * 16 + col] = my[((size_t)slot * 7 + burst) * 48 + q + 8];
                        q++;
                        if (q >= 32) {
                            q = 0;
                            burst++;
                        }
                    }
                }
                lc_pos[so] = carry + e[0];
            }
        }
    }
    for (int slot = 0; slot < 2; slot++) {
        lc_n[ch * 2 + slot] = n[slot] < max_lc ? n[slot] : max_lc;
        for (int k = n[slot]; k < max_lc; k++) {
            const size_t so = ((size_t)ch * 2 + slot) * max_lc + k;
            lc_pos[so] = -1;
            for (int b = 0; b < 128; b++) {
                in128[so * 128 + b] = 0;
            }
        }
    }
}

__global__ void
k_dmr_emb_finish(const uint8_t* __restrict__ out77, const int32_t* __restrict__ lc_pos, int n, uint8_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* b = out77 + (size_t)i * 77;
    uint32_t sum = 0, ext = 0;
    for (int k = 0; k < 9; k++) {
        uint32_t v = 0;
        for (int j = 0; j < 8; j++) {
            v = (v << 1) | (b[8 * k + j] & 1);
        }
        sum += v;
    }
    for (int j = 72; j < 77; j++) {
        ext = (ext << 1) | (b[j] & 1);
    }
    ok[i] = (lc_pos[i] >= 0 && (sum % 31) == ext) ? 1 : 0;
}
} // namespace

extern "C" hipError_t
ddn_dev_dmr_data_select(const int32_t* events, const int32_t* n_events, int max_events, int carry, int n_channels, int max_bursts,
                        const int32_t* sync_pos, const int32_t* n_sync, int max_syncs, const int32_t* out_pos, const int32_t* out_n,
                        int max_out, const int32_t* n_new, int32_t* dstart, uint8_t* dslot, int32_t* dpre, int32_t* dn, hipStream_t st) {
    if (n_channels <= 0 || max_bursts <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_data_select, dim3((unsigned)n_channels), dim3(64), 0, st, events, n_events, max_events, carry,
                       n_channels, max_bursts, sync_pos, n_sync, max_syncs, out_pos, out_n, max_out, n_new, dstart, dslot, dpre, dn);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_data_gather(const uint8_t* rec, size_t max_sym, const int32_t* dstart, const int32_t* dpre, const uint8_t* pre90,
                        const uint8_t* prel90, const uint8_t* pre90_out, const uint8_t* prel90_out, long split, int max_bursts,
                        int n_channels, uint8_t* slot_type, uint8_t* info, uint8_t* td98, uint8_t* rel98, hipStream_t st) {
    if (n_channels <= 0 || max_bursts <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_data_gather, dim3((unsigned)max_bursts, (unsigned)n_channels), dim3(64), 0, st, rec, max_sym, dstart, dpre,
                       pre90, prel90, pre90_out, prel90_out, split, max_bursts, slot_type, info, td98, rel98);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_data_prep(const int32_t* dstart, const uint8_t* slot_type, const uint8_t* st_ok, const uint8_t* pdu96, int n, uint8_t* type,
                      uint8_t* bytes12, uint8_t* cw12, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_data_prep, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, dstart, slot_type, st_ok, pdu96, n, type,
                       bytes12, cw12);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_data_finish(const uint8_t* type, const uint8_t* pdu96, const uint8_t* info, const uint8_t* cw12, const uint8_t* rs_result,
                        int n, uint8_t* bytes12, uint8_t* crc, uint8_t* r34_wanted, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_data_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, type, pdu96, info, cw12, rs_result, n,
                       bytes12, crc, r34_wanted);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_emb_collect(const int32_t* events, const int32_t* n_events, int max_events, int carry, const uint8_t* rec, size_t max_sym,
                        int n_channels, int max_lc, uint8_t* sig, uint8_t* in128, int32_t* lc_pos, int32_t* lc_n, hipStream_t st) {
    if (n_channels <= 0 || max_lc <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_emb_collect, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, events, n_events, max_events, carry,
                       rec, max_sym, n_channels, max_lc, sig, in128, lc_pos, lc_n);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_emb_finish(const uint8_t* out77, const int32_t* lc_pos, int n, uint8_t* ok, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_emb_finish, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, out77, lc_pos, n, ok);
    return hipGetLastError();
}
