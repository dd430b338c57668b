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
