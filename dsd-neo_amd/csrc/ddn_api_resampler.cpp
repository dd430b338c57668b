// ddn_api_resampler.cpp — C-ABI of the batched rational resampler (include/ddn_hip.h, "rational resampler").
// Batched analogue of dsd_resampler_design / dsd_resampler_process_block (reference include/dsd-neo/dsp/resampler.h:
// 66-89; src/dsp/resampler.cpp:241-356): every channel shares L, M and therefore the polyphase index, so the only
// per-channel state on the device is the 15-sample input history.

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_device.h"
#include "ddn_internal.h"

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

struct ddn_resampler {
    int n_channels, L, M, phase;
    float* d_taps; // [L][16]
    float* d_hist; // [B][15], oldest first
};

static long long
out_len_for(long long n, int L, int M, int phase) {
    const long long num = n * L + M - 1 - phase; // dsd_resampler_required_out_len, src/dsp/resampler.cpp:124-137
    return num <= 0 ? 0 : num / M;
}

extern "C" int
ddn_resampler_create(int n_channels, int L, int M, ddn_resampler** out) {
    if (!out || n_channels <= 0 || L < 1 || M < 1 || L > DDN_RESAMP_MAX_L || M > (1 << 22)) {
        ddn_set_error("ddn_resampler_create: bad argument (1 <= L <= %d, 1 <= M <= 2^22)", DDN_RESAMP_MAX_L);
        return DDN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_resampler* b = new (std::nothrow) ddn_resampler();
    float* taps = new (std::nothrow) float[(size_t)16 * L];
    if (!b || !taps) {
        delete b;
        delete[] taps;
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->n_channels = n_channels;
    b->L = L;
    b->M = M;
    ddn_design_resampler(L, M, taps);
    const size_t hb = sizeof(float) * 15 * (size_t)n_channels;
    const bool ok = hipMalloc(&b->d_taps, sizeof(float) * 16 * (size_t)L) == hipSuccess
                    && hipMalloc(&b->d_hist, hb) == hipSuccess && hipMemset(b->d_hist, 0, hb) == hipSuccess
                    && hipMemcpy(b->d_taps, taps, sizeof(float) * 16 * (size_t)L, hipMemcpyHostToDevice) == hipSuccess;
    delete[] taps;
    if (!ok) {
        ddn_set_error("ddn_resampler_create: device allocation failed");
        (void)hipFree(b->d_taps);
        (void)hipFree(b->d_hist);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_resampler_destroy(ddn_resampler* b) {
    if (!b) {
        return;
    }
    (void)hipFree(b->d_taps);
    (void)hipFree(b->d_hist);
    delete b;
}

extern "C" int
ddn_resampler_reset(ddn_resampler* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->phase = 0;
    HIP_TRY(hipMemsetAsync(b->d_hist, 0, sizeof(float) * 15 * (size_t)b->n_channels, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" size_t
ddn_resampler_out_len(const ddn_resampler* b, size_t n) {
    return b ? (size_t)out_len_for((long long)n, b->L, b->M, b->phase) : 0;
}

extern "C" int
ddn_resampler_get_taps(const ddn_resampler* b, float* taps, int cap) {
    if (!b || !taps || cap < 16 * b->L) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipMemcpy(taps, b->d_taps, sizeof(float) * 16 * (size_t)b->L, hipMemcpyDeviceToHost));
    return 16 * b->L;
}

extern "C" int
ddn_resampler_run(ddn_resampler* b, const float* d_in, size_t n, float* d_out, size_t out_stride, void* hip_stream) {
    if (!b || !d_in || !d_out) {
        ddn_set_error("ddn_resampler_run: null argument");
        return DDN_EINVAL;
    }
    const long long n_out = out_len_for((long long)n, b->L, b->M, b->phase);
    if ((size_t)n_out > out_stride) {
        ddn_set_error("ddn_resampler_run: out_stride %zu < %lld outputs (state unchanged)", out_stride, n_out);
        return DDN_ERANGE;
    }
    HIP_TRY(ddn_dev_resample(d_in, (long)n, n, b->n_channels, b->d_hist, b->d_taps, b->L, b->M, b->phase, (long)n_out,
                             d_out, out_stride, (hipStream_t)hip_stream));
    b->phase = (int)((long long)b->phase + n_out * b->M - (long long)n * b->L);
    return DDN_OK;
}

extern "C" int
ddn_resampler_run_host(ddn_resampler* b, const float* in, size_t n, float* out, size_t out_stride) {
    if (!b || !in || !out) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->n_channels;
    float *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, sizeof(float) * (B * n + 1)) != hipSuccess
        || hipMalloc(&d_out, sizeof(float) * (B * out_stride + 1)) != hipSuccess) {
        (void)hipFree(d_in);
        ddn_set_error("ddn_resampler_run_host: device allocation failed (no device?)");
        return DDN_ENODEV;
    }
    int rc = DDN_EHIP;
    if (hipMemcpy(d_in, in, sizeof(float) * B * n, hipMemcpyHostToDevice) == hipSuccess) {
        rc = ddn_resampler_run(b, d_in, n, d_out, out_stride, nullptr);
        if (rc == DDN_OK
            && (hipDeviceSynchronize() != hipSuccess
                || hipMemcpy(out, d_out, sizeof(float) * B * out_stride, hipMemcpyDeviceToHost) != hipSuccess)) {
            rc = DDN_EHIP;
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

// ---- drop-in: dsd_resampler_process_block() on a caller-owned reference state (one stream) ------------------------------
extern "C" int
dsd_resampler_process_block(ddn_dsd_resampler_state* state, const float* in, int in_len, float* out, int out_cap) {
    if (!state || !in || in_len < 0 || !out || out_cap < 0) {
        return -1;
    }
    if (!state->enabled || !state->taps || !state->hist) { // passthrough, src/dsp/resampler.cpp:192-202
        if (out_cap < in_len) {
            return -1;
        }
        if (out != in) {
            memcpy(out, in, sizeof(float) * (size_t)in_len);
        }
        return in_len;
    }
    const int L = state->L, M = state->M, K = state->taps_per_phase;
    if (K != 16 || L < 1 || M < 1 || state->phase < 0 || state->hist_head < 0 || state->hist_head >= K) {
        ddn_set_error("dsd_resampler_process_block: only the reference's 16-taps-per-phase design is supported");
        return -1;
    }
    const long long n_out = out_len_for(in_len, L, M, state->phase);
    if (n_out > out_cap) {
        return -1; // state unchanged, like the reference
    }
    if (in_len == 0) {
        return 0;
    }
    float *d_taps = nullptr, *d_hist = nullptr, *d_in = nullptr, *d_out = nullptr;
    bool ok = hipMalloc(&d_taps, sizeof(float) * 16 * (size_t)L) == hipSuccess
              && hipMalloc(&d_hist, sizeof(float) * 15) == hipSuccess
              && hipMalloc(&d_in, sizeof(float) * (size_t)in_len) == hipSuccess
              && hipMalloc(&d_out, sizeof(float) * (size_t)(n_out + 1)) == hipSuccess;
    // the window is hist[head .. head + 15], oldest first; the newest 15 are what the next outputs can still see
    ok = ok && hipMemcpy(d_taps, state->taps, sizeof(float) * 16 * (size_t)L, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(d_hist, state->hist + state->hist_head + 1, sizeof(float) * 15, hipMemcpyHostToDevice) == hipSuccess
         && hipMemcpy(d_in, in, sizeof(float) * (size_t)in_len, hipMemcpyHostToDevice) == hipSuccess
         && ddn_dev_resample(d_in, in_len, (size_t)in_len, 1, d_hist, d_taps, L, M, state->phase, (long)n_out, d_out,
                             (size_t)n_out + 1, nullptr) == hipSuccess
         && hipDeviceSynchronize() == hipSuccess
         && (n_out == 0 || hipMemcpy(out, d_out, sizeof(float) * (size_t)n_out, hipMemcpyDeviceToHost) == hipSuccess);
    (void)hipFree(d_taps);
    (void)hipFree(d_hist);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (!ok) {
        ddn_set_error("dsd_resampler_process_block: HIP path failed (no device?)");
        return -1;
    }
    // mirrored ring and bookkeeping exactly as in_len pushes leave them (resampler_push_sample, :205-214)
    const int first = in_len > K ? in_len - K : 0;
    for (int i = first; i < in_len; i++) {
        const int idx = (state->hist_head + i) % K;
        state->hist[idx] = in[i];
        state->hist[idx + K] = in[i];
    }
    state->hist_head = (state->hist_head + in_len) % K;
    state->phase = (int)((long long)state->phase + n_out * M - (long long)in_len * L);
    return (int)n_out;
}
