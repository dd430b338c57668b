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

// The short-integer voice path of one talk path per lane: processAudio() (src/core/audio/dsd_audio.c:427-571: block peak, 25-block
// peak history, gain drops at once / climbs 5 % per block, 160-sample ramp, clamp, truncate) -> hpf_dL()
// (src/core/util/dsd_misc.c:345-371,516-522; the filter state is float, the output is clamped and truncated per sample, so the
// recurrence runs in sample order) -> agsm() (src/core/audio/gain.c:143-184).  A wave takes SPW = 16 talk paths: each frame's
// 16 x 160 floats are read row by row (coalesced, all 48 loads in flight at once) into a 161-float-pitch LDS tile, lanes 0..15
// then walk one row each (pitch 161: the rows fall in different banks), leave the int16 values in place, and the tile is
// stored row by row again.  The walk is a latency chain (~20 dependent ops per sample), so the kernel wants many small waves
// rather than full ones: 4096 talk paths are 256 waves, one per CU.
__global__ __launch_bounds__(64) void
k_audio_s16(const float* __restrict__ pcm, int n_streams, int n_frames, float audio_gain, int use_hpf, int use_agsm, float coef,
            int16_t* __restrict__ out, float* __restrict__ state, float* __restrict__ gain_a) {
    constexpr int SPW = 16;
    __shared__ float tile[SPW * 161];
    __shared__ float hist[25][SPW];
    const int lane = threadIdx.x;
    const int s0 = blockIdx.x * SPW;
    const int rows = min(SPW, n_streams - s0);
    const bool live = lane < rows;
    float* stp = state + (size_t)(s0 + (live ? lane : 0)) * 32;
    float aout = 25.0f, vin0 = 0.0f, vout0 = 0.0f, ga = 0.0f;
    int idx = 0;
    if (live) {
        aout = stp[0];
        idx = (int)stp[1];
        vin0 = stp[2];
        vout0 = stp[3];
        for (int i = 0; i < 25; i++) {
            hist[i][lane] = stp[4 + i];
        }
    }
    float* row = tile + (live ? lane : 0) * 161;
    float v0[SPW], v1[SPW], v2[SPW];
    auto fetch = [&](int f) {
#pragma unroll
        for (int r = 0; r < SPW; r++) {
            const float* src = pcm + ((size_t)(s0 + min(r, rows - 1)) * n_frames + f) * 160;
            v0[r] = src[lane];
            v1[r] = src[64 + lane];
            v2[r] = src[128 + (lane & 31)];
        }
    };
    if (n_frames > 0) {
        fetch(0);
    }
    for (int f = 0; f < n_frames; f++) {
        __syncthreads();
        float mx = 0.0f; // lane r: the block peak of row r (max is exact, any order)
#pragma unroll
        for (int r = 0; r < SPW; r++) {
            tile[r * 161 + lane] = v0[r];
            tile[r * 161 + 64 + lane] = v1[r];
            if (lane < 32) {
                tile[r * 161 + 128 + lane] = v2[r];
            }
            float m = fmaxf(fmaxf(fabsf(v0[r]), fabsf(v1[r])), lane < 32 ? fabsf(v2[r]) : 0.0f);
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                m = fmaxf(m, __shfl_xor(m, sh));
            }
            mx = lane == r ? m : mx;
        }
        if (f + 1 < n_frames) {
            fetch(f + 1); // in flight while the rows are walked
        }
        __syncthreads();
        if (live) {
            float gd = 0.0f;
            if (audio_gain == 0.0f) {
                hist[idx][lane] = mx;
                idx = idx >= 24 ? 0 : idx + 1;
#pragma unroll
                for (int i = 0; i < 25; i++) {
                    mx = fmaxf(mx, hist[i][lane]);
                }
                float gf = mx > 0.0f ? __fdiv_rn(30000.0f, mx) : 50.0f;
                if (gf < aout) {
                    aout = gf;
                } else {
                    gf = fminf(gf, 50.0f);
                    gd = __fsub_rn(gf, aout);
                    const float cap = __fmul_rn(0.05f, aout);
                    if (gd > cap) {
                        gd = cap;
                    }
                }
                gd = __fdiv_rn(gd, 160.0f);
            }
            const bool mul = !(audio_gain < 0.0f);
            float ma = 0.0f;
            for (int n0 = 0; n0 < 160; n0 += 16) {
                float xb[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    xb[k] = row[n0 + k];
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    float v = xb[k];
                    if (mul) {
                        v = __fmul_rn(__fadd_rn(aout, __fmul_rn((float)(n0 + k), gd)), v);
                    }
                    v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v);
                    xb[k] = truncf(v);
                }
                if (use_hpf) {
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const float vin1 = vin0;
                        vin0 = xb[k];
                        vout0 = __fmul_rn(coef, __fadd_rn(__fsub_rn(vin0, vin1), vout0));
                        xb[k] = vout0 > 32767.0f ? 32767.0f : (vout0 < -32768.0f ? -32768.0f : truncf(vout0));
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    ma = fmaxf(ma, fabsf(xb[k]));
                    row[n0 + k] = xb[k];
                }
            }
            if (mul) {
                aout = __fadd_rn(aout, __fmul_rn(160.0f, gd));
            }
            if (use_agsm) {
                ma = ma < 1e-6f ? 1e-6f : ma;
                float c = fabsf(__fdiv_rn(4800.0f, ma));
                c = c > 3.0f ? 3.0f : c;
                for (int n0 = 0; n0 < 160; n0 += 16) {
                    float xb[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        xb[k] = row[n0 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        float sc = __fmul_rn(xb[k], c);
                        sc = sc > 32767.0f ? 32767.0f : (sc < -32768.0f ? -32768.0f : sc);
                        row[n0 + k] = truncf(sc);
                    }
                }
                ga = c;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < rows; r++) {
            int16_t* dst = out + ((size_t)(s0 + r) * n_frames + f) * 160;
            dst[lane] = (int16_t)tile[r * 161 + lane];
            dst[64 + lane] = (int16_t)tile[r * 161 + 64 + lane];
            if (lane < 32) {
                dst[128 + lane] = (int16_t)tile[r * 161 + 128 + lane];
            }
        }
    }
    if (live) {
        stp[0] = aout;
        stp[1] = (float)idx;
        stp[2] = vin0;
        stp[3] = vout0;
        for (int i = 0; i < 25; i++) {
            stp[4 + i] = hist[i][lane];
        }
        if (use_agsm && n_frames > 0) {
            gain_a[s0 + lane] = ga;
        }
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_audio_s16(const float* pcm, int n_streams, int n_frames, float audio_gain, int use_hpf, int use_agsm, float coef,
                  int16_t* out, float* state, float* gain_a, hipStream_t st) {
    if (n_streams <= 0 || n_frames <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_audio_s16, dim3((unsigned)((n_streams + 15) / 16)), dim3(64), 0, st, pcm, n_streams, n_frames,
                       audio_gain, use_hpf, use_agsm, coef, out, state, gain_a);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_agf(float* pcm, int n_streams, int n_frames, float gain, float* aout_gain, hipStream_t st) {
    if (n_streams <= 0 || n_frames <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_agf, dim3((unsigned)((n_streams + 63) / 64)), dim3(64), 0, st, pcm, n_streams, n_frames, gain,
                       aout_gain);
    return hipGetLastError();
}
