// ddn_halfband.hip — batched half-band decimate-by-2 cascade in front of the channel LPF.
//
// reference: full_demod_apply_halfband_decimation, src/dsp/demod_pipeline.cpp:983-1001 (31-tap first stage, 15-tap
// afterwards, taps src/dsp/halfband.cpp:35-74) calling simd_hb_decim2_complex, src/dsp/simd_fir.cpp:139-222 (AVX2 unit
// simd_fir_avx2.cpp): output m of a block is the symmetric FIR centred on input 2m, the window reads the previous
// samples of the stream on the left (the carried hist_i/hist_q) and REPLICATES THE BLOCK'S LAST SAMPLE on the right;
// blocks shorter than taps_len complex samples take the scalar unit's (mul, add) order, longer ones the FMA order.
//
// Unlike the discriminator this stage has no recurrence: every output depends only on the input stream, the block
// partition and the carried history, so it is one thread per output, channel-major coalesced loads (the window of a
// wave's 64 outputs is 128 + taps_len consecutive complex samples, served from L1/L2), one launch per stage.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));

namespace {
#define Q15(x) ((x) / 32768.0f)
__constant__ float c_hb15[15] = {Q15(-108.0f), 0.0f, Q15(1800.0f), 0.0f, Q15(-500.0f), 0.0f, Q15(7000.0f), 0.5f,
                                 Q15(7000.0f), 0.0f, Q15(-500.0f), 0.0f, Q15(1800.0f), 0.0f, Q15(-108.0f)};
__constant__ float c_hb31[31] = {0.0f, 0.0f, Q15(13.0f), 0.0f, Q15(-73.0f), 0.0f, Q15(233.0f), 0.0f, Q15(-587.0f), 0.0f,
                                 Q15(1314.0f), 0.0f, Q15(-2953.0f), 0.0f, Q15(10244.0f), Q15(16386.0f), Q15(10244.0f),
                                 0.0f, Q15(-2953.0f), 0.0f, Q15(1314.0f), 0.0f, Q15(-587.0f), 0.0f, Q15(233.0f), 0.0f,
                                 Q15(-73.0f), 0.0f, Q15(13.0f), 0.0f, 0.0f};
#undef Q15

template <int FMT>
__device__ __forceinline__ f2
load_iq(const void* base, size_t idx) {
    if (FMT == DDN_IN_CU8) {
        const uchar2 v = ((const uchar2*)base)[idx];
        f2 r = {((float)v.x - 127.5f) * (1.0f / 127.5f), ((float)v.y - 127.5f) * (1.0f / 127.5f)};
        return r;
    }
    return ((const f2*)base)[idx];
}
} // namespace

// One workgroup = 256 outputs of one reference block of one channel (grid.x enumerates (block, tile-in-block), so the
// right-edge replication is one clamp per staged sample).  The 512 + NT - 1 inputs those outputs touch are widened and
// staged in LDS once (each input feeds ~8 outputs); a thread then runs the centre tap and the non-zero symmetric pairs as
// packed (I, Q) operations.
template <int FMT, int NT>
__global__ __launch_bounds__(256) void
k_hb_decim2(const void* __restrict__ in, long n_in, size_t in_stride, int block_in, int tiles_per_block,
            const f2* __restrict__ hist, f2* __restrict__ out, size_t out_stride) {
    constexpr int H = NT - 1, CEN = H / 2;
    __shared__ f2 win[512 + H];
    const float* taps = (NT == 31) ? c_hb31 : c_hb15;
    const int ch = blockIdx.y;
    const long b = blockIdx.x / tiles_per_block;
    const int tile = blockIdx.x % tiles_per_block;
    const long s = b * block_in;
    if (s >= n_in) {
        return;
    }
    const long L = (n_in - s) < block_in ? (n_in - s) : block_in;
    const long m0 = (long)tile * 256; // first output of this tile inside the block
    if (m0 >= (L >> 1)) {
        return;
    }
    const long last = s + L - 1;
    const bool fused = L >= NT;
    const long j0 = s + 2 * m0 - CEN; // input index of win[0]
    for (int i = threadIdx.x; i < 512 + H; i += 256) {
        long j = j0 + i;
        j = j > last ? last : j;
        win[i] = (j < 0) ? hist[(size_t)ch * H + (size_t)(H + j)] : load_iq<FMT>(in, (size_t)ch * in_stride + (size_t)j);
    }
    __syncthreads();
    const long ml = m0 + threadIdx.x;
    if (ml >= (L >> 1)) {
        return;
    }
    const int c = 2 * threadIdx.x + CEN; // window index of this output's centre sample
    const f2 z = {0.0f, 0.0f};
    f2 acc;
    {
        const f2 h = {taps[CEN], taps[CEN]};
        acc = fused ? __builtin_elementwise_fma(h, win[c], z) : (z + h * win[c]);
    }
#pragma unroll
    for (int k = 0; k < CEN; k += 2) {
        const float hk = taps[k];
        if (hk == 0.0f) {
            continue;
        }
        const int d = CEN - k;
        const f2 sm = win[c - d] + win[c + d];
        const f2 h = {hk, hk};
        acc = fused ? __builtin_elementwise_fma(h, sm, acc) : (acc + h * sm);
    }
    out[(size_t)ch * out_stride + (size_t)((s >> 1) + ml)] = acc;
}

// hist <- last H samples of (hist ++ in[0..n))
template <int FMT>
__global__ void
k_hb_hist(const void* __restrict__ in, long n, size_t in_stride, int H, f2* __restrict__ hist) {
    const int ch = blockIdx.x, i = threadIdx.x;
    f2 v = {0.0f, 0.0f};
    if (i < H) {
        const long j = n - H + i;
        v = (j >= 0) ? load_iq<FMT>(in, (size_t)ch * in_stride + (size_t)j) : hist[(size_t)ch * H + (size_t)(H + j)];
    }
    __syncthreads();
    if (i < H) {
        hist[(size_t)ch * H + i] = v;
    }
}

extern "C" hipError_t
ddn_dev_hb_decim2(const void* in, int in_fmt, long n_in, size_t in_stride, int block_in, int n_channels, int taps_len,
                  void* hist, void* out, size_t out_stride, hipStream_t st) {
    if (n_channels <= 0 || n_in < 2) {
        return hipSuccess;
    }
    const long n_blocks = (n_in + block_in - 1) / block_in;
    const int tiles_per_block = ((block_in >> 1) + 255) / 256;
    const dim3 grid((unsigned)(n_blocks * tiles_per_block), (unsigned)n_channels), blk(256);
    const f2* h = (const f2*)hist;
    f2* o = (f2*)out;
    if (in_fmt == DDN_IN_CU8) {
        if (taps_len == 31) {
            hipLaunchKernelGGL((k_hb_decim2<DDN_IN_CU8, 31>), grid, blk, 0, st, in, n_in, in_stride, block_in, tiles_per_block, h, o, out_stride);
        } else {
            hipLaunchKernelGGL((k_hb_decim2<DDN_IN_CU8, 15>), grid, blk, 0, st, in, n_in, in_stride, block_in, tiles_per_block, h, o, out_stride);
        }
    } else {
        if (taps_len == 31) {
            hipLaunchKernelGGL((k_hb_decim2<DDN_IN_CF32, 31>), grid, blk, 0, st, in, n_in, in_stride, block_in, tiles_per_block, h, o, out_stride);
        } else {
            hipLaunchKernelGGL((k_hb_decim2<DDN_IN_CF32, 15>), grid, blk, 0, st, in, n_in, in_stride, block_in, tiles_per_block, h, o, out_stride);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        return e;
    }
    if (in_fmt == DDN_IN_CU8) {
        hipLaunchKernelGGL((k_hb_hist<DDN_IN_CU8>), dim3((unsigned)n_channels), dim3(32), 0, st, in, n_in, in_stride,
                           taps_len - 1, (f2*)hist);
    } else {
        hipLaunchKernelGGL((k_hb_hist<DDN_IN_CF32>), dim3((unsigned)n_channels), dim3(32), 0, st, in, n_in, in_stride,
                           taps_len - 1, (f2*)hist);
    }
    return hipGetLastError();
}
