// ddn_api_ted.cpp — C-ABI of the batched Gardner timing-recovery stage (include/ddn_hip.h, "timing recovery").
// Batched analogue of op25_gardner_cc(struct demod_state*) (reference include/dsd-neo/dsp/costas.h): the carried
// ted_state_t of every channel lives on the device inside the ddn_ted_batch object.

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

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

struct ddn_ted_batch {
    int n_channels, sps, symbol_rate_hz;
    float ted_gain;
    long block_len; // 0: one reference call per ddn_gardner_run; else the reference's block size in samples
    DdnTedState* d_state;
    float* d_dl;
    int* d_count;
};

extern "C" int
ddn_ted_batch_create(int n_channels, int sps, int symbol_rate_hz, float ted_gain, ddn_ted_batch** out) {
    if (!out || n_channels <= 0 || sps < 2 || sps > 49) {
        ddn_set_error("ddn_ted_batch_create: bad argument (sps must be 2..49 for the %d-entry delay line)", DDN_TED_DL);
        return DDN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_ted_batch* b = new (std::nothrow) ddn_ted_batch();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->n_channels = n_channels;
    b->sps = sps;
    b->symbol_rate_hz = symbol_rate_hz;
    b->ted_gain = ted_gain;
    const size_t B = (size_t)n_channels;
    if (hipMalloc(&b->d_state, sizeof(DdnTedState) * B) != hipSuccess
        || hipMalloc(&b->d_dl, sizeof(float) * DDN_TED_DL * 4 * B) != hipSuccess
        || hipMalloc(&b->d_count, sizeof(int) * B) != hipSuccess
        || hipMemset(b->d_state, 0, sizeof(DdnTedState) * B) != hipSuccess
        || hipMemset(b->d_dl, 0, sizeof(float) * DDN_TED_DL * 4 * B) != hipSuccess) {
        ddn_set_error("ddn_ted_batch_create: device allocation failed");
        (void)hipFree(b->d_state);
        (void)hipFree(b->d_dl);
        (void)hipFree(b->d_count);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_ted_batch_destroy(ddn_ted_batch* b) {
    if (!b) {
        return;
    }
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_dl);
    (void)hipFree(b->d_count);
    delete b;
}

extern "C" int
ddn_ted_batch_reset(ddn_ted_batch* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->n_channels;
    HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(DdnTedState) * B, (hipStream_t)hip_stream));
    HIP_TRY(hipMemsetAsync(b->d_dl, 0, sizeof(float) * DDN_TED_DL * 4 * B, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_ted_batch_set_block_len(ddn_ted_batch* b, size_t block_len) {
    if (!b || (block_len != 0 && block_len < 4)) {
        ddn_set_error("ddn_ted_batch_set_block_len: bad argument (block_len must be 0 or >= 4)");
        return DDN_EINVAL;
    }
    b->block_len = (long)block_len;
    return DDN_OK;
}

extern "C" int
ddn_gardner_run(ddn_ted_batch* b, const float* d_iq, size_t n, float* d_sym, size_t sym_stride, int* d_sym_count,
                void* hip_stream) {
    if (!b || !d_iq || !d_sym) {
        ddn_set_error("ddn_gardner_run: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_gardner(d_iq, (long)n, n, b->n_channels, b->sps, b->ted_gain, b->symbol_rate_hz, b->block_len,
                            b->d_state, b->d_dl, d_sym, sym_stride, d_sym_count ? d_sym_count : b->d_count,
                            (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_gardner_run_host(ddn_ted_batch* b, const float* iq, size_t n, float* sym, size_t sym_stride, int* sym_count) {
    if (!b || !iq || !sym || !sym_count) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->n_channels;
    float *d_in = nullptr, *d_out = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_in, B * n * 8 + 8) != hipSuccess || hipMalloc(&d_out, B * sym_stride * 8 + 8) != hipSuccess) {
        ddn_set_error("ddn_gardner_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, iq, B * n * 8, hipMemcpyHostToDevice) != hipSuccess) {
        rc = DDN_EHIP;
    } else {
        rc = ddn_gardner_run(b, d_in, n, d_out, sym_stride, nullptr, nullptr);
        if (rc == DDN_OK
            && (hipMemcpy(sym, d_out, B * sym_stride * 8, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(sym_count, b->d_count, B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)) {
            rc = DDN_EHIP;
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

extern "C" int
ddn_ted_batch_get_state(ddn_ted_batch* b, int channel, float out8[8]) {
    if (!b || !out8 || channel < 0 || channel >= b->n_channels) {
        return DDN_EINVAL;
    }
    DdnTedState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    out8[0] = s.mu;
    out8[1] = s.omega;
    out8[2] = s.last_r;
    out8[3] = s.last_j;
    out8[4] = s.lock_accum;
    out8[5] = (float)s.lock_count;
    out8[6] = (float)s.dl_index;
    out8[7] = (float)s.twice_sps;
    return DDN_OK;
}
