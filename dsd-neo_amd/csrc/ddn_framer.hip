// ddn_framer.hip — from the receive loop's capture records to the FEC kernels' input layouts, on the device.
//
// reference: what the P25 Phase 1 frame handlers do with getDibitSoft() after a sync - read the NID's 32 dibits with the
// status symbol dropped (src/protocol/p25/phase1/dispatch_p25p1.c:123-143), then walk the frame body dibit by dibit,
// stepping over one status symbol after every 35 dibits (p25p1_ldu.c:27-39, p25p1_tsbk / p25p1_hdu readers).  Every
// field of a frame therefore sits at a fixed dibit offset from the frame sync; the host builds those offset tables once
// (ddn_host_p25_layout.c) and the device does two things:
//   k_find_syncs     one wavefront per channel scans the flag row (bit 1 = "this symbol completed a frame sync"),
//                    ballot + prefix popcount, and writes the record index of each sync's last dibit in order
//   k_gather_fields  one thread per (frame slot, dibit of the field): record -> hard bits, per-bit reliability and/or
//                    int16 LLRs in the layout the block-code / trellis kernels read.  Slots past a channel's sync count
//                    and fields that run past the channel's records are zero-filled and marked invalid.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {
__global__ __launch_bounds__(64) void
k_find_syncs(const uint8_t* __restrict__ flags, const int32_t* __restrict__ counts, size_t max_sym, int max_frames,
             int32_t* __restrict__ sync_pos, int32_t* __restrict__ n_syncs, int32_t* __restrict__ dropped) {
    const int ch = blockIdx.x;
    const int lane = threadIdx.x;
    const int cnt = counts[ch] < (int)max_sym ? counts[ch] : (int)max_sym; // the rx loop counts symbols it could not store
    const uint8_t* row = flags + (size_t)ch * max_sym;
    int found = 0;
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        const bool hit = i < cnt && (row[i] & 2);
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int k = found + __popcll(m & ((1ull << lane) - 1ull));
            if (k < max_frames) {
                sync_pos[(size_t)ch * max_frames + k] = i;
            }
        }
        found += __popcll(m);
    }
    if (lane == 0) {
        n_syncs[ch] = found < max_frames ? found : max_frames;
        if (dropped && found > max_frames) { // syncs that found no frame slot in this call: counted, never silently lost
            dropped[ch] += found - max_frames;
        }
    }
}

__global__ __launch_bounds__(256) void
k_gather_fields(const uint8_t* __restrict__ rec, size_t max_sym, const int32_t* __restrict__ counts,
                const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_syncs, int n_channels,
                int max_frames, const int32_t* __restrict__ offsets, int n_off, int max_off, uint8_t* __restrict__ bits,
                uint8_t* __restrict__ rel, int16_t* __restrict__ llr, int stride, int split_last,
                uint8_t* __restrict__ last_bit, uint8_t* __restrict__ last_rel, uint8_t* __restrict__ valid,
                uint8_t* __restrict__ dibits, uint8_t* __restrict__ dibit_rel, DdnSel sel) {
    auto one = [&](long t) {
    const long slot = t / n_off;
    const int i = (int)(t % n_off);
    const int ch = (int)(slot / max_frames), k = (int)(slot % max_frames);
    const int cnt = counts[ch] < (int)max_sym ? counts[ch] : (int)max_sym;
    bool ok = k < n_syncs[ch];
    int start = 0;
    if (ok) {
        start = sync_pos[slot] - 23; // record index of the frame sync's first dibit
        ok = start + max_off < cnt;
    }
    int d = 0, r = 0, l0 = 0, l1 = 0;
    if (ok) {
        const uint8_t* q = rec + ((size_t)ch * max_sym + (size_t)(start + offsets[i])) * 10;
        d = q[0];
        r = q[1];
        l0 = (int16_t)((uint16_t)q[2] | ((uint16_t)q[3] << 8));
        l1 = (int16_t)((uint16_t)q[4] | ((uint16_t)q[5] << 8));
    }
    const int b0 = (d >> 1) & 1, b1 = d & 1;
    const bool tail = split_last && i == n_off - 1; // NID: bit 63 is the parity bit, kept apart from the BCH word
    const size_t o = (size_t)slot * stride + 2 * (size_t)i;
    if (bits) {
        bits[o] = (uint8_t)b0;
        if (!tail) {
            bits[o + 1] = (uint8_t)b1;
        }
    }
    // per-BIT reliabilities are min(|llr|, 255) of that bit's own LLR (p25p1_llr_reliability dispatch_p25p1.c:59-66,
    // p25p1_append_bch_bits :73-83, soft_abs_i16 p25p1_ldu.c:21-24,135-152, p25p1_hdu.c:128-139, p25p1_tdulc.c:82-90);
    // the record's per-dibit byte (their minimum) is only what the 3/4-rate trellis takes (dibit_rel below)
    const int a0 = l0 < 0 ? -l0 : l0, a1 = l1 < 0 ? -l1 : l1;
    const int r0 = a0 > 255 ? 255 : a0, r1 = a1 > 255 ? 255 : a1;
    if (rel) {
        rel[o] = (uint8_t)r0;
        if (!tail) {
            rel[o + 1] = (uint8_t)r1;
        }
    }
    if (llr) {
        llr[o] = (int16_t)l0;
        if (!tail) {
            llr[o + 1] = (int16_t)l1;
        }
    }
    if (tail) {
        if (last_bit) {
            last_bit[slot] = (uint8_t)b1;
        }
        if (last_rel) {
            last_rel[slot] = (uint8_t)r1; // dispatch_p25p1.c:142
        }
    }
    if (dibits) { // one byte per dibit (the 3/4-rate trellis decoder's input), reliability alongside
        dibits[(size_t)slot * n_off + i] = (uint8_t)(d & 3);
    }
    if (dibit_rel) {
        dibit_rel[(size_t)slot * n_off + i] = (uint8_t)r;
    }
    if (i == 0 && valid) {
        valid[slot] = ok ? 1 : 0;
    }
    };
    ddn_sel_for_each(sel, (long)n_channels * max_frames * n_off, one);
}
} // namespace

extern "C" hipError_t
ddn_dev_find_syncs(const uint8_t* flags, const int32_t* counts, int n_channels, size_t max_sym, int max_frames,
                   int32_t* sync_pos, int32_t* n_syncs, int32_t* dropped, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_find_syncs, dim3((unsigned)n_channels), dim3(64), 0, st, flags, counts, max_sym, max_frames,
                       sync_pos, n_syncs, dropped);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_gather_fields(const uint8_t* rec, size_t max_sym, const int32_t* counts, const int32_t* sync_pos,
                      const int32_t* n_syncs, int n_channels, int max_frames, const int32_t* offsets, int n_off,
                      int max_off, uint8_t* bits, uint8_t* rel, int16_t* llr, int stride, int split_last,
                      uint8_t* last_bit, uint8_t* last_rel, uint8_t* valid, uint8_t* dibits, uint8_t* dibit_rel,
                      hipStream_t st) {
    const long total = (long)n_channels * max_frames * n_off;
    if (total <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(n_off);
    hipLaunchKernelGGL(k_gather_fields, dim3(ddn_sel_grid(&sel, ((unsigned long)total + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, rec, max_sym, counts,
                       sync_pos, n_syncs, n_channels, max_frames, offsets, n_off, max_off, bits, rel, llr, stride,
                       split_last, last_bit, last_rel, valid, dibits, dibit_rel, sel);
    return hipGetLastError();
}

namespace {
// record index (into the flat [B][max_sym] record array) and status counter of the nine voice frames of every LDU slot,
// the inputs of k_imbe_deinterleave; slots without a sync get first = -1 (flagged 0xFF there)
__global__ void
k_imbe_index(const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_syncs, int n_channels, int max_frames,
             size_t max_sym, const int32_t* __restrict__ first9, const int32_t* __restrict__ status9,
             int64_t* __restrict__ first, int32_t* __restrict__ status) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long slot = t / 9;
    const int v = (int)(t % 9);
    if (slot >= (long)n_channels * max_frames) {
        return;
    }
    const int ch = (int)(slot / max_frames), k = (int)(slot % max_frames);
    const bool ok = k < n_syncs[ch];
    first[t] = ok ? (int64_t)((size_t)ch * max_sym) + (sync_pos[slot] - 23) + first9[v] : -1;
    status[t] = status9[v];
}
} // namespace

namespace {
// decoded words [slots][n_words][wstride] (first six entries of a word = its hex symbol's bits) -> the RS decoder's
// layout: data [slots][n_data][6] and parity [slots][n_words - n_data][6], word order unchanged
// (LDU1 p25p1_ldu1.c:233-245, LDU2 p25p1_ldu2.c:256-262, HDU p25p1_hdu.c:252-270)
__global__ void
k_rs_pack(const uint8_t* __restrict__ words, long n_slots, int n_words, int wstride, int n_data,
          uint8_t* __restrict__ data, uint8_t* __restrict__ parity, DdnSel sel) {
    const int per = n_words * 6;
    auto one = [&](long t) {
    const long slot = t / per;
    const int r = (int)(t % per), w = r / 6, b = r % 6;
    const uint8_t v = words[(slot * n_words + w) * (long)wstride + b];
    if (w < n_data) {
        data[slot * (long)(n_data * 6) + w * 6 + b] = v;
    } else {
        parity[slot * (long)((n_words - n_data) * 6) + (w - n_data) * 6 + b] = v;
    }
    };
    ddn_sel_for_each(sel, n_slots * per, one);
}
} // namespace

extern "C" hipError_t
ddn_dev_rs_pack(const uint8_t* words, long n_slots, int n_words, int wstride, int n_data, uint8_t* data, uint8_t* parity,
                hipStream_t st) {
    if (n_slots <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(n_words * 6);
    hipLaunchKernelGGL(k_rs_pack, dim3(ddn_sel_grid(&sel, ((unsigned long)n_slots * n_words * 6 + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, words, n_slots,
                       n_words, wstride, n_data, data, parity, sel);
    return hipGetLastError();
}

namespace {
// TDULC: Golay-corrected dodeca words [slots][12][12] (dodeca_data[0..5], dodeca_parity[0..5]) -> RS(24,12,13) input with
// the two hex halves of every dodeca word swapped (swap_hex_words, p25p1_tdulc.c:47-72,210-213): hex 2i = bits 6..11,
// hex 2i + 1 = bits 0..5 of word i
__global__ void
k_tdulc_rs_pack(const uint8_t* __restrict__ words, long n_slots, uint8_t* __restrict__ data, uint8_t* __restrict__ parity,
                DdnSel sel) {
    auto one = [&](long t) {
    const long slot = t / 144;
    const int r = (int)(t % 144), hexw = r / 6, b = r % 6; // hexw 0..11 data, 12..23 parity
    const int dodeca = hexw / 2, half = hexw & 1;
    const uint8_t v = words[(slot * 12 + dodeca) * 12 + (half ? b : 6 + b)];
    if (hexw < 12) {
        data[slot * 72 + hexw * 6 + b] = v;
    } else {
        parity[slot * 72 + (hexw - 12) * 6 + b] = v;
    }
    };
    ddn_sel_for_each(sel, n_slots * 144, one);
}
} // namespace

extern "C" hipError_t
ddn_dev_tdulc_rs_pack(const uint8_t* words, long n_slots, uint8_t* data, uint8_t* parity, hipStream_t st) {
    if (n_slots <= 0) {
        return hipSuccess;
    }
    const DdnSel sel = ddn_sel_for(144);
    hipLaunchKernelGGL(k_tdulc_rs_pack, dim3(ddn_sel_grid(&sel, ((unsigned long)n_slots * 144 + DDN_WG - 1) / DDN_WG)), dim3(DDN_WG), 0, st, words, n_slots,
                       data, parity, sel);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_imbe_index(const int32_t* sync_pos, const int32_t* n_syncs, int n_channels, int max_frames, size_t max_sym,
                   const int32_t* first9, const int32_t* status9, int64_t* first, int32_t* status, hipStream_t st) {
    const long total = (long)n_channels * max_frames * 9;
    if (total <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_imbe_index, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sync_pos, n_syncs,
                       n_channels, max_frames, max_sym, first9, status9, first, status);
    return hipGetLastError();
}

namespace {
// voice frames of every channel in air order, compacted: the slots whose NID decoded (status > 0: NID_OK or NID_PARITY_OVERRIDE,
// include/dsd-neo/protocol/p25/p25p1_check_nid.h:30-41) to LDU1 / LDU2 - the
// only callers of process_IMBE (processLDU1 / processLDU2, src/engine/dispatch/dispatch_p25p1.c) - in sync order, nine
// frames each; unused entries get first = -1 (k_imbe_deinterleave flags those 0xFF).  One thread per channel: a
// channel has a handful of frames per call.
__global__ void
k_voice_index(const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_syncs, const int32_t* __restrict__ nid4,
              const int32_t* __restrict__ counts, int n_channels, int max_frames, int max_ldu, size_t max_sym, const int32_t* __restrict__ first9,
              const int32_t* __restrict__ status9, int64_t* __restrict__ first, int32_t* __restrict__ status,
              int32_t* __restrict__ n_ldu) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    int k = 0;
    const int ns = n_syncs[ch];
    const int cnt = counts[ch] < (int)max_sym ? counts[ch] : (int)max_sym;
    for (int slot = 0; slot < ns && k < max_ldu; slot++) {
        const int32_t* nd = nid4 + ((size_t)ch * max_frames + slot) * 4;
        if (nd[0] > 0 && (nd[2] == 0x5 || nd[2] == 0xA)) { // NID_OK and NID_PARITY_OVERRIDE both dispatch (dispatch_p25p1.c:214-218)
            const int start = sync_pos[(size_t)ch * max_frames + slot] - 23;
            const int64_t base = (int64_t)((size_t)ch * max_sym) + start;
            for (int v = 0; v < 9; v++) {
                // a frame whose 72 dibits (+ the status symbols stepped over on the way) run past this channel's records
                // is left for the next call's records: -1 here (the flat record array continues with the next channel)
                const int sc0 = status9[v], t1 = 35 - sc0;
                const int last = start + first9[v] + 71 + ((sc0 <= 35 && 71 >= t1) ? 1 + (71 - t1) / 35 : 0);
                first[((size_t)ch * max_ldu + k) * 9 + v] = last < cnt ? base + first9[v] : -1;
                status[((size_t)ch * max_ldu + k) * 9 + v] = status9[v];
            }
            k++;
        }
    }
    if (n_ldu) {
        n_ldu[ch] = k;
    }
    for (; k < max_ldu; k++) {
        for (int v = 0; v < 9; v++) {
            first[((size_t)ch * max_ldu + k) * 9 + v] = -1;
            status[((size_t)ch * max_ldu + k) * 9 + v] = 0;
        }
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_voice_index(const int32_t* sync_pos, const int32_t* n_syncs, const int32_t* nid4, const int32_t* counts,
                    int n_channels, int max_frames,
                    int max_ldu, size_t max_sym, const int32_t* first9, const int32_t* status9, int64_t* first,
                    int32_t* status, int32_t* n_ldu, hipStream_t st) {
    if (n_channels <= 0 || max_ldu <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_voice_index, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, sync_pos, n_syncs, nid4,
                       counts, n_channels, max_frames, max_ldu, max_sym, first9, status9, first, status, n_ldu);
    return hipGetLastError();
}
