// ddn_audio.hip - float-path audio post-processing of synthesized voice frames: the auto gain dsd-neo
// applies to every 160-sample frame before output.
//   agf()  reference src/core/audio/gain.c:23-35,47-139 (called from playSynthesizedVoiceFS / FM / FS3, src/core/audio/
//          dsd_audio2.c:1100,1111).  The gain state (aout_gain) walks +-0.5 per 20-sample block, so a talk path's frames
//          are sequential; talk paths are independent -> one lane per talk path, frames in order, in place.
//   Kept on purpose: the block average reads the first twenty samples of the frame whatever block is being processed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {
__global__ void
k_agf(float* __restrict__ pcm, int n_streams, int n_frames, float gain, float* __restrict__ aout_gain) {
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= n_streams) {
        return;
    }
    float aout = aout_gain[sidx];
    for (int f = 0; f < n_frames; f++) {
        float* s = pcm + ((size_t)sidx * n_frames + f) * 160;
        bool silent = true;
        for (int i = 0; i < 160; i++) {
            const float v = s[i];
            silent = silent && !(v > 1e-12f || v < -1e-12f);
        }
        if (silent) {
            continue;
        }
        float first[20]; // the frame's first twenty samples as they stand (what the block average reads)
        for (int j = 0; j < 8; j++) {
            const float df = 384.0f * (50.0f - aout);
            float aavg = 0.0f;
            for (int i = 0; i < 20; i++) {
                const int idx = j * 20 + i;
                float v = s[idx] / df;
                v = v > 0.90f ? 0.90f : (v < -0.90f ? -0.90f : v);
                const float seen = (j == 0) ? v : first[i]; // block 0 reads its own sample before the gain multiply
                aavg += fabsf(seen);
                v *= gain * 0.8f;
                s[idx] = v;
                if (j == 0) {
                    first[i] = v;
                }
            }
            aavg /= 20.0f;
            if (aavg < 0.075f && aout < 46.0f) {
                aout += 0.5f;
            }
            if (aavg >= 0.075f && aout > 1.0f) {
                aout -= 0.5f;
            }
        }
    }
    aout_gain[sidx] = aout;
}
} // namespace

extern "C" hipError_t
ddn_dev_agf(float* pcm, int n_streams, int n_frames, float gain, float* aout_gain, hipStream_t st) {
    if (n_streams <= 0 || n_frames <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_agf, dim3((unsigned)((n_streams + 63) / 64)), dim3(64), 0, st, pcm, n_streams, n_frames, gain,
                       aout_gain);
    return hipGetLastError();
}
