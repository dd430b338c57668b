// ddn_api_slicer.cpp — C-ABI of the batched P25p1 slicer / soft-decision stage and the P25 matched filter
// (include/ddn_hip.h).  The per-channel slicer words of dsd_state (center/umid/lmid/min/max, the 128-symbol window,
// the two 1024-deep extrema rings and their binary64 sums) live on the device inside the batch object.

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

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

struct ddn_slicer_batch {
    int n_channels, negative;
    DdnSlicerState* d_state;
    float *d_sbuf, *d_minring, *d_maxring, *d_fhist;
};

static int
slicer_fill(ddn_slicer_batch* b) {
    // reset exactly like symbol_reset_rtl_fsk_discriminator_slicer() (reference src/dsp/dsd_symbol.c:1306-1326)
    const size_t B = (size_t)b->n_channels;
    DdnSlicerState s;
    memset(&s, 0, sizeof(s));
    s.center = 0.0f;
    s.min = -30000.0f;
    s.max = 30000.0f;
    s.lmid = -20000.0f;
    s.umid = 20000.0f;
    std::vector<DdnSlicerState> hs(B, s);
    std::vector<float> mn(1024 * B, -30000.0f), mx(1024 * B, 30000.0f);
    if (hipMemcpy(b->d_state, hs.data(), sizeof(s) * B, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(b->d_minring, mn.data(), sizeof(float) * 1024 * B, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(b->d_maxring, mx.data(), sizeof(float) * 1024 * B, hipMemcpyHostToDevice) != hipSuccess
        || hipMemset(b->d_sbuf, 0, sizeof(float) * 128 * B) != hipSuccess
        || hipMemset(b->d_fhist, 0, sizeof(float) * 90 * B) != hipSuccess) {
        ddn_set_error("slicer state upload failed");
        return DDN_EHIP;
    }
    return DDN_OK;
}

extern "C" int
ddn_slicer_batch_create(int n_channels, int negative_polarity, ddn_slicer_batch** out) {
    if (!out || n_channels <= 0) {
        return DDN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_slicer_batch* b = new (std::nothrow) ddn_slicer_batch();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->n_channels = n_channels;
    b->negative = negative_polarity ? 1 : 0;
    const size_t B = (size_t)n_channels;
    if (hipMalloc(&b->d_state, sizeof(DdnSlicerState) * B) != hipSuccess
        || hipMalloc(&b->d_sbuf, sizeof(float) * 128 * B) != hipSuccess
        || hipMalloc(&b->d_minring, sizeof(float) * 1024 * B) != hipSuccess
        || hipMalloc(&b->d_maxring, sizeof(float) * 1024 * B) != hipSuccess
        || hipMalloc(&b->d_fhist, sizeof(float) * 90 * B) != hipSuccess || slicer_fill(b) != DDN_OK) {
        ddn_set_error("ddn_slicer_batch_create: device allocation failed");
        (void)hipFree(b->d_state);
        (void)hipFree(b->d_sbuf);
        (void)hipFree(b->d_minring);
        (void)hipFree(b->d_maxring);
        (void)hipFree(b->d_fhist);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_slicer_batch_destroy(ddn_slicer_batch* b) {
    if (!b) {
        return;
    }
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_sbuf);
    (void)hipFree(b->d_minring);
    (void)hipFree(b->d_maxring);
    (void)hipFree(b->d_fhist);
    delete b;
}

extern "C" int
ddn_slicer_batch_reset(ddn_slicer_batch* b) {
    if (!b) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    return slicer_fill(b);
}

extern "C" int
ddn_p25_slicer_run(ddn_slicer_batch* b, const float* d_symbols, size_t n, uint8_t* d_records10, void* hip_stream) {
    if (!b || !d_symbols || !d_records10) {
        ddn_set_error("ddn_p25_slicer_run: null argument");
        return DDN_EINVAL;
    }
    // The sequential kernel (one lane per channel walking every symbol) remains for short calls; from a few hundred
    // symbols on the parallel decomposition (ddn_slicer_par.hip) wins.  DDN_SLICER_SEQ forces the sequential one (A/B).
    static const bool force_seq = DDN_EXP_ENV("DDN_SLICER_SEQ") != nullptr;
    if (n >= 256 && !force_seq) {
        const size_t need = sizeof(float) * 4 * n * (size_t)b->n_channels;
        float* scratch = nullptr;
        HIP_TRY(hipMallocAsync((void**)&scratch, need, (hipStream_t)hip_stream));
        const hipError_t e = ddn_dev_p25_slicer_par(d_symbols, (long)n, n, b->n_channels, b->negative, b->d_state, b->d_sbuf,
                                                    b->d_minring, b->d_maxring, scratch, d_records10, n * 10,
                                                    (hipStream_t)hip_stream);
        HIP_TRY(hipFreeAsync(scratch, (hipStream_t)hip_stream));
        HIP_TRY(e);
        return DDN_OK;
    }
    HIP_TRY(ddn_dev_p25_slicer(d_symbols, (long)n, n, b->n_channels, b->negative, b->d_state, b->d_sbuf, b->d_minring,
                               b->d_maxring, d_records10, n * 10, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25_slicer_run_host(ddn_slicer_batch* b, const float* symbols, size_t n, uint8_t* records10) {
    if (!b || !symbols || !records10) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->n_channels;
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    int rc;
    if (hipMalloc(&d_in, B * n * 4 + 4) != hipSuccess || hipMalloc(&d_out, B * n * 10 + 4) != hipSuccess) {
        ddn_set_error("ddn_p25_slicer_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, symbols, B * n * 4, hipMemcpyHostToDevice) != hipSuccess) {
        rc = DDN_EHIP;
    } else {
        rc = ddn_p25_slicer_run(b, d_in, n, d_out, nullptr);
        if (rc == DDN_OK && hipMemcpy(records10, d_out, B * n * 10, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = DDN_EHIP;
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

extern "C" int
ddn_slicer_batch_get_thresholds(ddn_slicer_batch* b, int channel, float out5[5]) {
    if (!b || !out5 || channel < 0 || channel >= b->n_channels) {
        return DDN_EINVAL;
    }
    DdnSlicerState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    out5[0] = s.center;
    out5[1] = s.umid;
    out5[2] = s.lmid;
    out5[3] = s.max;
    out5[4] = s.min;
    return DDN_OK;
}

extern "C" int
ddn_p25_matched_filter_run(ddn_slicer_batch* b, const float* d_in, size_t n, float* d_out, void* hip_stream) {
    if (!b || !d_in || !d_out) {
        ddn_set_error("ddn_p25_matched_filter_run: null argument");
        return DDN_EINVAL;
    }
    if (d_in == d_out) {
        ddn_set_error("ddn_p25_matched_filter_run: in-place operation is not supported");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25_matched_filter(d_in, (long)n, n, b->n_channels, b->d_fhist, d_out, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25_matched_filter_run_host(ddn_slicer_batch* b, const float* in, size_t n, float* out) {
    if (!b || !in || !out) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->n_channels;
    float *d_in = nullptr, *d_out = nullptr;
    int rc;
    if (hipMalloc(&d_in, B * n * 4 + 4) != hipSuccess || hipMalloc(&d_out, B * n * 4 + 4) != hipSuccess) {
        ddn_set_error("ddn_p25_matched_filter_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, in, B * n * 4, hipMemcpyHostToDevice) != hipSuccess) {
        rc = DDN_EHIP;
    } else {
        rc = ddn_p25_matched_filter_run(b, d_in, n, d_out, nullptr);
        if (rc == DDN_OK && hipMemcpy(out, d_out, B * n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = DDN_EHIP;
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}
