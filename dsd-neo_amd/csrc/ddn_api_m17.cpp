// ddn_api_m17.cpp - C-ABI of the M17 frame decoders behind the receive loop (include/ddn_fsk4.h, "M17"): link setup frames through
// the K = 5 decoder of SURVEY row a17 (ddn_fec_viterbi_k5_batch).  Device pointers, asynchronous on the stream; the scratch is
// stream-ordered (hipMallocAsync), so concurrent calls on different streams never share it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "ddn_device.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            ddn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                  \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu)             \
                       ? DDN_ENODEV                                                                                    \
                       : (e_ == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP);                                          \
        }                                                                                                              \
    } while (0)

extern "C" hipError_t ddn_dev_m17_lsf_cost(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos,
                                           const uint8_t* sync_pat, const int32_t* n_sync, const float* sync_thr, int n_channels,
                                           int max_syncs, int lmax, uint16_t* cost488, int32_t* slot_sync, hipStream_t st);
extern "C" hipError_t ddn_dev_m17_lsf_finish(const uint8_t* dec, int dec_stride, const uint32_t* cost, const int32_t* slot_sync,
                                             int n_channels, int lmax, int max_syncs, uint8_t* lsf30, uint8_t* status,
                                             uint32_t* path_cost, hipStream_t st);
extern "C" int ddn_fec_viterbi_k5_batch(const uint16_t* d_soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* d_out,
                                        int out_stride, uint32_t* d_cost, void* hip_stream);

extern "C" int
ddn_m17_lsf_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                         const uint8_t* d_sync_pat, const int32_t* d_n_sync, const float* d_sync_thr5, int n_channels, size_t max_syncs,
                         uint8_t* d_lsf30, uint8_t* d_status, uint32_t* d_path_cost, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_sync_pat || !d_n_sync || !d_sync_thr5 || !d_lsf30 || !d_status || n_channels <= 0
        || max_syncs == 0 || max_syncs > (1u << 24) || stride_symbols == 0) {
        ddn_set_error("ddn_m17_lsf_decode_batch: bad argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // an LSF frame follows a preamble and takes 192 symbols: at most stride / 192 + 1 of them per channel and call
    const int lmax = (int)(stride_symbols / 192 + 1);
    const size_t S = (size_t)n_channels * (size_t)lmax;
    const size_t b_cost = ((S * 488 * sizeof(uint16_t)) + 255) & ~(size_t)255, b_dec = ((S * 32) + 255) & ~(size_t)255;
    const size_t b_pc = ((S * sizeof(uint32_t)) + 255) & ~(size_t)255, b_slot = ((S * sizeof(int32_t)) + 255) & ~(size_t)255;
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, b_cost + b_dec + b_pc + b_slot, st));
    uint16_t* cost = (uint16_t*)scratch;
    uint8_t* dec = scratch + b_cost;
    uint32_t* pc = (uint32_t*)(dec + b_dec);
    int32_t* slot = (int32_t*)((uint8_t*)pc + b_pc);
    int rc = DDN_OK;
    hipError_t e = hipMemsetAsync(d_status, 0, (size_t)n_channels * max_syncs, st);
    if (e == hipSuccess) {
        e = ddn_dev_m17_lsf_cost(d_records10, stride_symbols, d_counts, d_sync_pos, d_sync_pat, d_n_sync, d_sync_thr5, n_channels, (int)max_syncs,
                                 lmax, cost, slot, st);
    }
    if (e == hipSuccess) {
        rc = ddn_fec_viterbi_k5_batch(cost, S, 488, nullptr, 0, dec, 32, pc, st);
    }
    if (e == hipSuccess && rc == DDN_OK) {
        e = ddn_dev_m17_lsf_finish(dec, 32, pc, slot, n_channels, lmax, (int)max_syncs, d_lsf30, d_status, d_path_cost, st);
    }
    const hipError_t ef = hipFreeAsync(scratch, st);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(e);
    HIP_TRY(ef);
    return DDN_OK;
}
