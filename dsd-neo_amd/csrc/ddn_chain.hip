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
ddn_dev_chain_counts(const int32_t* cnt_new, int T, int n_channels, int flush, int32_t* cnt_scan, int32_t* cnt_full, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_chain_counts, dim3((unsigned)((n_channels + 255) / 256)), dim3(256), 0, st, cnt_new, T, n_channels, flush,
                       cnt_scan, cnt_full);
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
