// ddn_api_audio.cpp - C-ABI of the voice-frame auto gain (include/ddn_hip.h, kernel ddn_audio.hip)
#include <hip/hip_runtime.h>

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

static float
effective_gain(float audio_gain, int algid_0x21) { // agf_effective_gain(), src/core/audio/gain.c:47-58
    float gain = 1.0f;
    if (algid_0x21) {
        gain = 1.75f;
    }
    if (audio_gain != 0) {
        gain = audio_gain / 25.0f;
    }
    return gain;
}

extern "C" int
ddn_audio_agf_batch(float* d_pcm, int n_streams, int n_frames, float audio_gain, int algid_0x21, float* d_aout_gain,
                    void* hip_stream) {
    if (!d_pcm || !d_aout_gain || n_streams <= 0 || n_frames < 0) {
        ddn_set_error("ddn_audio_agf_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_agf(d_pcm, n_streams, n_frames, effective_gain(audio_gain, algid_0x21), d_aout_gain, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_audio_agf_host(float* pcm, int n_streams, int n_frames, float audio_gain, int algid_0x21, float* aout_gain) {
    if (!pcm || !aout_gain || n_streams <= 0 || n_frames < 0) {
        return DDN_EINVAL;
    }
    const size_t np = (size_t)n_streams * (size_t)n_frames * 160;
    float *d_p = nullptr, *d_g = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_p, np * 4 + 4) != hipSuccess || hipMalloc(&d_g, (size_t)n_streams * 4) != hipSuccess) {
        ddn_set_error("ddn_audio_agf_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_p, pcm, np * 4, hipMemcpyHostToDevice) != hipSuccess
               || hipMemcpy(d_g, aout_gain, (size_t)n_streams * 4, hipMemcpyHostToDevice) != hipSuccess
               || ddn_dev_agf(d_p, n_streams, n_frames, effective_gain(audio_gain, algid_0x21), d_g, nullptr) != hipSuccess
               || hipMemcpy(pcm, d_p, np * 4, hipMemcpyDeviceToHost) != hipSuccess
               || hipMemcpy(aout_gain, d_g, (size_t)n_streams * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_p);
    (void)hipFree(d_g);
    return rc;
}

// drop-in shape of agf() for one talk path: the caller's aout_gain travels as a plain float (the reference keeps it in dsd_state)
extern "C" int
ddn_agf_frame(float samp[160], float audio_gain, int algid_0x21, float* aout_gain_io) {
    return ddn_audio_agf_host(samp, 1, 1, audio_gain, algid_0x21, aout_gain_io);
}

// ---- the short-integer voice path (processAudio -> hpf_dL -> agsm) ----------------------------------------------------------
static float
hpf_d_coef() { // HPFilter_Init(&state->HRCFilterL, 960.0f, 1.0f / 8000.0f): src/core/util/dsd_misc.c:345-358,436-452
    float RC = 0.0;
    RC = 1.0 / (2 * 3.141592653 * 960.0f);
    return RC / ((1.0f / 8000.0f) + RC);
}

extern "C" int
ddn_audio_s16_state_init(float* state32, int n_streams) {
    if (!state32 || n_streams < 0) {
        return DDN_EINVAL;
    }
    for (int s = 0; s < n_streams; s++) {
        for (int i = 0; i < DDN_S16_STATE_FLOATS; i++) {
            state32[(size_t)s * DDN_S16_STATE_FLOATS + i] = 0.0f;
        }
        state32[(size_t)s * DDN_S16_STATE_FLOATS] = 25.0f; // state->aout_gain, src/core/util/dsd_init.c:580
    }
    return DDN_OK;
}

extern "C" int
ddn_audio_s16_batch(const float* d_pcm, int n_streams, int n_frames, float audio_gain, int use_hpf_d, int use_agsm,
                    int16_t* d_out, float* d_state32, float* d_gain_a, void* hip_stream) {
    if (!d_pcm || !d_out || !d_state32 || (use_agsm && !d_gain_a) || n_streams <= 0 || n_frames < 0) {
        ddn_set_error("ddn_audio_s16_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_audio_s16(d_pcm, n_streams, n_frames, audio_gain, use_hpf_d, use_agsm, hpf_d_coef(), d_out, d_state32,
                              d_gain_a, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_audio_s16_host(const float* pcm, int n_streams, int n_frames, float audio_gain, int use_hpf_d, int use_agsm, int16_t* out,
                   float* state32, float* gain_a) {
    if (!pcm || !out || !state32 || (use_agsm && !gain_a) || n_streams <= 0 || n_frames < 0) {
        return DDN_EINVAL;
    }
    const size_t np = (size_t)n_streams * (size_t)n_frames * 160, ns = (size_t)n_streams * DDN_S16_STATE_FLOATS * 4;
    float *d_p = nullptr, *d_s = nullptr, *d_g = nullptr;
    int16_t* d_o = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_p, np * 4 + 4) != hipSuccess || hipMalloc(&d_o, np * 2 + 4) != hipSuccess
        || hipMalloc(&d_s, ns) != hipSuccess || hipMalloc(&d_g, (size_t)n_streams * 4) != hipSuccess) {
        ddn_set_error("ddn_audio_s16_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_p, pcm, np * 4, hipMemcpyHostToDevice) != hipSuccess
               || hipMemcpy(d_s, state32, ns, hipMemcpyHostToDevice) != hipSuccess
               || (gain_a && hipMemcpy(d_g, gain_a, (size_t)n_streams * 4, hipMemcpyHostToDevice) != hipSuccess)
               || ddn_dev_audio_s16(d_p, n_streams, n_frames, audio_gain, use_hpf_d, use_agsm, hpf_d_coef(), d_o, d_s, d_g,
                                    nullptr) != hipSuccess
               || hipMemcpy(out, d_o, np * 2, hipMemcpyDeviceToHost) != hipSuccess
               || hipMemcpy(state32, d_s, ns, hipMemcpyDeviceToHost) != hipSuccess
               || (gain_a && hipMemcpy(gain_a, d_g, (size_t)n_streams * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_p);
    (void)hipFree(d_o);
    (void)hipFree(d_s);
    (void)hipFree(d_g);
    return rc;
}
