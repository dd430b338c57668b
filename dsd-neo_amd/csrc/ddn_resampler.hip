// ddn_resampler.hip — batched rational L/M polyphase resampler.
//
// reference: dsd_resampler_process_block / resamp_process_block, src/dsp/resampler.cpp:205-233,318-356,405-419
// (16 taps per phase, dot product as four interleaved partial sums :84-99), applied by the demodulator thread to the
// discriminator output when the demod rate is not the symbol loop's rate (src/io/radio/rtl_sdr_fm.cpp:3311-3313).
//
// The reference walks inputs and emits 0..ceil(L/M) outputs per input; there is no feedback, so output q of a call
// that starts at polyphase index p0 is a closed form: u = p0 + q*M, input index n = u / L, phase = u % L, window =
// the 16 inputs ending at n.  One thread per output; a workgroup owns OT consecutive outputs of one channel, stages
// the input span they touch (previous call's last 15 samples first) and the whole tap table in LDS with coalesced
// loads, and every thread runs the reference's 4 x 4 multiply / add order (no FMA: -ffp-contract=off).
// HBM traffic = 4 B per input + 4 B per output; the 15-sample history and the phase are the only carried state.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {
constexpr int K = 16;
constexpr int OT_MAX = 1024; // outputs per workgroup (halved until the input span fits the LDS budget)

__global__ __launch_bounds__(256) void
k_resample(const float* __restrict__ in, long n, size_t in_stride, const float* __restrict__ hist,
           const float* __restrict__ taps, int L, int M, int p0, long n_out, float* __restrict__ out,
           size_t out_stride, int span_cap, int OT) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* st = sm;                 // [L][K]
    float* sx = sm + (size_t)L * K; // [span_cap] inputs n_lo - 15 ...
    const int ch = blockIdx.y;
    const long q0 = (long)blockIdx.x * OT;
    const long q1 = (q0 + OT < n_out) ? q0 + OT : n_out;
    for (int i = threadIdx.x; i < L * K; i += 256) {
        st[i] = taps[i];
    }
    const long n_lo = (long)(((long long)p0 + (long long)q0 * M) / L);       // newest input of the first output
    const long n_hi = (long)(((long long)p0 + (long long)(q1 - 1) * M) / L); // newest input of the last output
    const int span = (int)(n_hi - n_lo) + K;
    const float* row = in + (size_t)ch * in_stride;
    for (int i = threadIdx.x; i < span; i += 256) {
        const long j = n_lo - (K - 1) + i; // input index, negative = carried history
        sx[i] = (j >= 0) ? row[j] : hist[(size_t)ch * (K - 1) + (K - 1) + j];
    }
    __syncthreads();
    // Output q0 + j sits at upsampled position u0 + j * M: input offset (r0 + j * M) / L past n_lo and phase
    // (r0 + j * M) % L, with r0 = u0 % L < L.  All of it fits 32 bits (j < OT <= 1024), and a thread's next output (j + 256)
    // follows by adding the quotient and remainder of 256 * M / L with one carry - no division in the loop.
    const unsigned r0 = (unsigned)(((long long)p0 + (long long)q0 * M) % L);
    const unsigned step_q = (256u * (unsigned)M) / (unsigned)L, step_r = (256u * (unsigned)M) % (unsigned)L;
    unsigned off = (r0 + threadIdx.x * (unsigned)M) / (unsigned)L;
    unsigned ph = (r0 + threadIdx.x * (unsigned)M) % (unsigned)L;
    for (long q = q0 + threadIdx.x; q < q1; q += 256) {
        const float* w = sx + off;
        const float4* t4 = (const float4*)(st + (size_t)ph * K); // a phase's 16 taps are 64-byte aligned: four 16-byte reads
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
            const float4 t = t4[k >> 2];
            a0 += w[k + 0] * t.x;
            a1 += w[k + 1] * t.y;
            a2 += w[k + 2] * t.z;
            a3 += w[k + 3] * t.w;
        }
        out[(size_t)ch * out_stride + q] = (a0 + a1) + (a2 + a3);
        off += step_q;
        ph += step_r;
        if (ph >= (unsigned)L) {
            ph -= (unsigned)L;
            off++;
        }
    }
}

// carried history <- the last 15 inputs seen (older ones come from the previous history when n < 15)
__global__ void
k_resample_hist(const float* __restrict__ in, long n, size_t in_stride, int n_channels, float* __restrict__ hist) {
    const int ch = blockIdx.x * 16 + threadIdx.x / 16;
    const int k = threadIdx.x % 16;
    float v = 0.0f;
    const bool on = ch < n_channels && k < K - 1;
    if (on) {
        const long j = n - (K - 1) + k;
        v = (j >= 0) ? in[(size_t)ch * in_stride + j] : hist[(size_t)ch * (K - 1) + (K - 1) + j];
    }
    __syncthreads(); // all reads of the old history precede the writes (a channel's 15 slots sit in one workgroup)
    if (on) {
        hist[(size_t)ch * (K - 1) + k] = v;
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_resample(const float* in, long n, size_t in_stride, int n_channels, float* hist, const float* taps, int L, int M,
                 int p0, long n_out, float* out, size_t out_stride, hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    if (n_out > 0) {
        // inputs one workgroup can touch: OT outputs advance (OT - 1) * M / L inputs, plus the 16-tap window.  Steep
        // decimations (M / L in the tens) shrink OT so that taps + span stay within 64 KB of LDS.
        int OT = OT_MAX;
        while (OT > 1 && sizeof(float) * ((size_t)L * K + (size_t)(((long long)(OT - 1) * M) / L) + K + 2) > 64 * 1024) {
            OT >>= 1;
        }
        const int span_cap = (int)(((long long)(OT - 1) * M) / L) + K + 2;
        const size_t shm = sizeof(float) * ((size_t)L * K + (size_t)span_cap);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resample),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) {
            return e;
        }
        hipLaunchKernelGGL(k_resample, dim3((unsigned)((n_out + OT - 1) / OT), (unsigned)n_channels), dim3(256), shm,
                           st, in, n, in_stride, hist, taps, L, M, p0, n_out, out, out_stride, span_cap, OT);
        e = hipGetLastError();
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(k_resample_hist, dim3((unsigned)((n_channels + 15) / 16)), dim3(256), 0, st, in, n, in_stride,
                       n_channels, hist);
    return hipGetLastError();
}
