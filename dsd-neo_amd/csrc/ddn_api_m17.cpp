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

extern "C" hipError_t ddn_dev_m17_str_bits(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos,
                                           const uint8_t* sync_pat, const int32_t* n_sync, int n_channels, int max_syncs, int lmax,
                                           uint8_t* sym296, int32_t* slot_sync, uint8_t* slot_lich6, uint8_t* slot_cnt, uint8_t* slot_ok,
                                           hipStream_t st);
extern "C" hipError_t ddn_dev_m17_str_finish(const uint8_t* dec, int dec_stride, const int32_t* slot_sync, const uint8_t* slot_lich6,
                                             const uint8_t* slot_cnt, const uint8_t* slot_ok, int n_channels, int lmax, int max_syncs,
                                             uint8_t* lich6, uint8_t* lich_cnt, uint8_t* fn_payload18, uint8_t* status, hipStream_t st);
extern "C" hipError_t ddn_dev_m17_lich(const uint8_t* sync_pat, const int32_t* n_sync, int n_channels, int max_syncs, const uint8_t* lsf30,
                                       const uint8_t* lsf_status, const uint8_t* lich6, const uint8_t* lich_cnt, const uint8_t* str_status,
                                       uint8_t* asm30, uint8_t* lich_lsf30, uint8_t* lich_status, hipStream_t st);

extern "C" int
ddn_m17_str_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                         const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_lich6,
                         uint8_t* d_lich_cnt, uint8_t* d_fn_payload18, uint8_t* d_status, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_sync_pat || !d_n_sync || !d_lich6 || !d_lich_cnt || !d_fn_payload18 || !d_status
        || n_channels <= 0 || max_syncs == 0 || max_syncs > (1u << 24) || stride_symbols == 0) {
        ddn_set_error("ddn_m17_str_decode_batch: bad argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int lmax = (int)(stride_symbols / 192 + 1); // a stream frame takes 192 symbols
    const size_t S = (size_t)n_channels * (size_t)lmax;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_sym = up(S * 296), b_dec = up(S * 18), b_slot = up(S * sizeof(int32_t)), b_l6 = up(S * 6), b_1 = up(S);
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, b_sym + b_dec + b_slot + b_l6 + 2 * b_1, st));
    uint8_t* sym = scratch;
    uint8_t* dec = sym + b_sym;
    int32_t* slot = (int32_t*)(dec + b_dec);
    uint8_t* l6 = (uint8_t*)slot + b_slot;
    uint8_t* cnt = l6 + b_l6;
    uint8_t* ok = cnt + b_1;
    hipError_t e = hipMemsetAsync(d_status, 0, (size_t)n_channels * max_syncs, st);
    if (e == hipSuccess) {
        e = ddn_dev_m17_str_bits(d_records10, stride_symbols, d_counts, d_sync_pos, d_sync_pat, d_n_sync, n_channels, (int)max_syncs, lmax, sym,
                                 slot, l6, cnt, ok, st);
    }
    if (e == hipSuccess) { // (blocks of 16 frames none of which has a LICH that decoded write zeros and leave)
        e = ddn_dev_k5_nxdn_wanted(sym, nullptr, (int)S, 148, 144, nullptr, dec, 18, ok, 1, st);
    }
    if (e == hipSuccess) {
        e = ddn_dev_m17_str_finish(dec, 18, slot, l6, cnt, ok, n_channels, lmax, (int)max_syncs, d_lich6, d_lich_cnt, d_fn_payload18, d_status, st);
    }
    const hipError_t ef = hipFreeAsync(scratch, st);
    HIP_TRY(e);
    HIP_TRY(ef);
    return DDN_OK;
}

extern "C" int
ddn_m17_lich_assemble_batch(const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs, const uint8_t* d_lsf30,
                            const uint8_t* d_lsf_status, const uint8_t* d_lich6, const uint8_t* d_lich_cnt, const uint8_t* d_str_status,
                            uint8_t* d_assembly32, uint8_t* d_lich_lsf30, uint8_t* d_lich_status, void* hip_stream) {
    if (!d_sync_pat || !d_n_sync || !d_lich6 || !d_lich_cnt || !d_str_status || !d_assembly32 || !d_lich_lsf30 || !d_lich_status
        || n_channels <= 0 || max_syncs == 0 || max_syncs > (1u << 24) || ((d_lsf30 == nullptr) != (d_lsf_status == nullptr))) {
        ddn_set_error("ddn_m17_lich_assemble_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_m17_lich(d_sync_pat, d_n_sync, n_channels, (int)max_syncs, d_lsf30, d_lsf_status, d_lich6, d_lich_cnt, d_str_status,
                             d_assembly32, d_lich_lsf30, d_lich_status, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" hipError_t ddn_dev_ysf_fich_cost(const uint8_t* rec, size_t stride, const int32_t* counts, const int32_t* sync_pos,
                                            const int32_t* n_sync, int n_channels, int max_syncs, int lmax, uint16_t* cost200,
                                            int32_t* slot_sync, hipStream_t st);
extern "C" hipError_t ddn_dev_ysf_fich_finish(const uint8_t* dec, int dec_stride, const uint32_t* cost, const int32_t* slot_sync,
                                              int n_channels, int lmax, int max_syncs, uint8_t* fich4, uint8_t* status, uint32_t* v_error,
                                              hipStream_t st);

extern "C" int
ddn_ysf_fich_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                          const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_fich4, uint8_t* d_status, uint32_t* d_v_error,
                          void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_n_sync || !d_fich4 || !d_status || n_channels <= 0 || max_syncs == 0
        || max_syncs > (1u << 24) || stride_symbols == 0) {
        ddn_set_error("ddn_ysf_fich_decode_batch: bad argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // syncs are at least a window (20 symbols) + the FICH apart
    const int lmax = (int)(stride_symbols / 120 + 1);
    const size_t S = (size_t)n_channels * (size_t)lmax;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_cost = up(S * 200 * sizeof(uint16_t)), b_dec = up(S * 16), b_pc = up(S * sizeof(uint32_t)), b_slot = up(S * sizeof(int32_t));
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, b_cost + b_dec + b_pc + b_slot, st));
    uint16_t* cost = (uint16_t*)scratch;
    uint8_t* dec = scratch + b_cost;
    uint32_t* pc = (uint32_t*)(dec + b_dec);
    int32_t* slot = (int32_t*)((uint8_t*)pc + b_pc);
    int rc = DDN_OK;
    hipError_t e = hipMemsetAsync(d_status, 0, (size_t)n_channels * max_syncs, st);
    if (e == hipSuccess) {
        e = ddn_dev_ysf_fich_cost(d_records10, stride_symbols, d_counts, d_sync_pos, d_n_sync, n_channels, (int)max_syncs, lmax, cost, slot, st);
    }
    if (e == hipSuccess) {
        static const uint8_t none[4] = {1, 1, 1, 1}; // DSD_YSF_PUNCTURE_NONE
        rc = ddn_fec_viterbi_k5_batch(cost, S, 200, none, 4, dec, 16, pc, st);
    }
    if (e == hipSuccess && rc == DDN_OK) {
        e = ddn_dev_ysf_fich_finish(dec, 16, pc, slot, n_channels, lmax, (int)max_syncs, d_fich4, d_status, d_v_error, st);
    }
    const hipError_t ef = hipFreeAsync(scratch, st);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(e);
    HIP_TRY(ef);
    return DDN_OK;
}

extern "C" hipError_t ddn_dev_ysf_plan(const int32_t* sync_pos, const int32_t* n_sync, const int32_t* counts, int n_channels, int max_syncs,
                                       int lmax, const uint8_t* fich4, const uint8_t* fich_status, uint8_t* last2, uint8_t* info,
                                       int32_t* slot_sync, hipStream_t st);
extern "C" hipError_t ddn_dev_ysf_payload_costs(const uint8_t* rec, size_t stride, const int32_t* sync_pos, int n_channels, int max_syncs,
                                                int lmax, const uint8_t* info, const int32_t* slot_sync, uint16_t* cost200,
                                                uint16_t* cost360, uint8_t* ambe49, uint8_t* errs2, uint8_t* want200, uint8_t* want360,
                                                uint8_t* frames, uint8_t* n_frames, hipStream_t st);
extern "C" int ddn_fec_viterbi_k5_batch_wanted(const uint16_t* d_soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* d_out,
                                               int out_stride, uint32_t* d_cost, const uint8_t* d_wanted, void* hip_stream);
extern "C" hipError_t ddn_dev_ysf_dch_finish(const uint8_t* decA, const uint32_t* pcA, const uint8_t* decB, const uint32_t* pcB,
                                             const int32_t* slot_sync, const uint8_t* info, int n_channels, int lmax, int max_syncs,
                                             uint8_t* dch40, uint8_t* dch_status, uint32_t* dch_cost, hipStream_t st);

// include/ddn_fsk4.h
extern "C" int
ddn_ysf_payload_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                             const int32_t* d_n_sync, int n_channels, size_t max_syncs, const uint8_t* d_fich4, const uint8_t* d_fich_status,
                             uint8_t* d_last_dt_fi, uint8_t* d_info2, uint8_t* d_dch40, uint8_t* d_dch_status2, uint32_t* d_dch_cost2,
                             uint8_t* d_ambe49x5, uint8_t* d_errs2x5, uint8_t* d_frames184x5, uint8_t* d_n_frames, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_n_sync || !d_fich4 || !d_fich_status || !d_last_dt_fi || !d_info2 || !d_dch40
        || !d_dch_status2 || !d_dch_cost2 || !d_ambe49x5 || !d_errs2x5 || !d_frames184x5 || !d_n_frames || n_channels <= 0 || max_syncs == 0 || max_syncs > (1u << 24)
        || stride_symbols == 0) {
        ddn_set_error("ddn_ysf_payload_decode_batch: bad argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // frames are a sync window + the lock (20 + 460 symbols) apart
    const int lmax = (int)(stride_symbols / 480 + 2);
    const size_t S = (size_t)n_channels * (size_t)lmax, SO = (size_t)n_channels * max_syncs;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_cA = up(S * 200 * sizeof(uint16_t)), b_cB = up(S * 2 * 360 * sizeof(uint16_t) + S * 3), b_dA = up(S * 16), b_dB = up(S * 2 * 32),
                 b_pA = up(S * sizeof(uint32_t)), b_pB = up(S * 2 * sizeof(uint32_t)), b_slot = up(S * sizeof(int32_t));
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, b_cA + b_cB + b_dA + b_dB + b_pA + b_pB + b_slot, st));
    uint16_t* cA = (uint16_t*)scratch;
    uint16_t* cB = (uint16_t*)(scratch + b_cA);
    uint8_t* wA = scratch + b_cA + S * 2 * 360 * sizeof(uint16_t); // which code words of the two lists hold a block (cleared with the costs)
    uint8_t* wB = wA + S;
    uint8_t* dA = scratch + b_cA + b_cB;
    uint8_t* dB = dA + b_dA;
    uint32_t* pA = (uint32_t*)(dB + b_dB);
    uint32_t* pB = (uint32_t*)((uint8_t*)pA + b_pA);
    int32_t* slot = (int32_t*)((uint8_t*)pB + b_pB);
    int rc = DDN_OK;
    hipError_t e = hipMemsetAsync(cA, 0, b_cA + b_cB, st);
    if (e == hipSuccess) {
        e = hipMemsetAsync(slot, 0xFF, b_slot, st);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(d_dch_status2, 0, SO * 2, st);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(d_dch_cost2, 0, SO * 2 * sizeof(uint32_t), st);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(d_dch40, 0, SO * 40, st);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(d_frames184x5, 0, SO * 5 * 184, st);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(d_n_frames, 0, SO, st);
    }
    if (e == hipSuccess) {
        e = ddn_dev_ysf_plan(d_sync_pos, d_n_sync, d_counts, n_channels, (int)max_syncs, lmax, d_fich4, d_fich_status, d_last_dt_fi, d_info2,
                             slot, st);
    }
    if (e == hipSuccess) {
        e = ddn_dev_ysf_payload_costs(d_records10, stride_symbols, d_sync_pos, n_channels, (int)max_syncs, lmax, d_info2, slot, cA, cB,
                                      d_ambe49x5, d_errs2x5, wA, wB, d_frames184x5, d_n_frames, st);
    }
    static const uint8_t none[4] = {1, 1, 1, 1}; // DSD_YSF_PUNCTURE_NONE
    if (e == hipSuccess) {
        rc = ddn_fec_viterbi_k5_batch_wanted(cA, S, 200, none, 4, dA, 16, pA, wA, st);
    }
    if (e == hipSuccess && rc == DDN_OK) {
        rc = ddn_fec_viterbi_k5_batch_wanted(cB, S * 2, 360, none, 4, dB, 32, pB, wB, st);
    }
    if (e == hipSuccess && rc == DDN_OK) {
        e = ddn_dev_ysf_dch_finish(dA, pA, dB, pB, slot, d_info2, n_channels, lmax, (int)max_syncs, d_dch40, d_dch_status2, d_dch_cost2, st);
    }
    const hipError_t ef = hipFreeAsync(scratch, st);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(e);
    HIP_TRY(ef);
    return DDN_OK;
}
