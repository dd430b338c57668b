// ddn_front_end.hip — gfx950 kernels for the dsd-neo FSK front end:
//   cu8/cf32 widen -> zero-latency symmetric complex channel LPF -> phase delta      (k_fir_phase)
//   dc centring + peak AGC + clip (serial recurrences) + squelch gate                 (k_fsk_serial)
//
// Reference behaviour reproduced (paths relative to the dsd-neo tree):
//   widen                     src/dsp/simd_widen.cpp:139-149
//   channel LPF               src/dsp/simd_fir_avx2.cpp:119-143 (per-output FMA chain: centre tap, then
//                             k = 0..centre-1 of fma(h[k], x[n-d] + x[n+d], acc)), block-edge sample
//                             replication src/dsp/simd_fir.cpp:66-85
//   power / squelch           src/dsp/demod_pipeline.cpp:926-945,1003-1020,1173-1190
//   discriminator             src/dsp/fsk_modem.c:23-35,89-164
//
// Arithmetic contract: every float op below is an IEEE-754 binary32 op in the reference's order; the file is
// compiled with -ffp-contract=off and fused multiply-adds appear only where the reference's AVX2 unit has
// _mm256_fmadd_ps.  No MFMA: the symmetric pre-add makes the contraction a VALU (v_pk_add/v_pk_fma) job.
//
// Data layout in HBM: input [B][n] interleaved I/Q (2 B or 8 B per complex sample), output [B][n] f32,
// both channel-major so one channel's stream is contiguous (that is what the stream-read hook hands out).
// Per-channel carried state: CENTER widened samples of FIR history, 5 words of modem state.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------------
// helpers

__device__ __forceinline__ f2
ddn_load_iq(const void* base, int fmt, size_t ch_stride_samples, int ch, long p) {
    if (fmt == DDN_IN_CU8) {
        const uint16_t* s = (const uint16_t*)base + (size_t)ch * ch_stride_samples + p;
        const uint16_t v = *s;
        const float inv = 1.0f / 127.5f;
        f2 r;
        r.x = ((float)(v & 0xFF) - 127.5f) * inv;
        r.y = ((float)(v >> 8) - 127.5f) * inv;
        return r;
    }
    const f2* s = (const f2*)base + (size_t)ch * ch_stride_samples + p;
    return *s;
}

// src/dsp/fsk_modem.c:23-35 + :89-94
__device__ __forceinline__ float
ddn_phase_delta(f2 cur, f2 prev) {
    const float re = cur.x * prev.x + cur.y * prev.y;
    const float im = cur.y * prev.x - cur.x * prev.y;
    if (re > 1.0e-7f && fabsf(im) <= (0.35f * re)) {
        const float x = im / re;
        const float x2 = x * x;
        return x * (1.0f + x2 * (-0.3333333333333333f + x2 * 0.2f));
    }
    // large-angle branch: libm atan2f in the reference.  Evaluate in binary64 and round once; this equals
    // the correctly-rounded binary32 result except in astronomically rare double-rounding cases, and is
    // within 1 ulp of any faithful libm (tolerance stated in tests/test_front_end_gpu.py).
    return (float)atan2((double)im, (double)re);
}

// ------------------------------------------------------------------------------------------------------
// K1: widen + channel LPF + phase delta.  grid = (tiles, B), 256 threads, R outputs per thread.
//
// LDS window: logical element i (0 <= i < T + 2*CENTER) = input sample (tile_start - CENTER + i), already
// edge-replicated, stored at physical slot i + i/R: thread t's run of R consecutive elements then starts at
// t*(R+1), an odd multiple of 8 bytes, so a wave's ds_read_b64 at a common logical offset touches all 64
// banks exactly once (conflict-free) while every address is (thread base + compile-time constant).

template <int CENTER>
struct DdnTaps {
    float centre;
    float side[CENTER]; // side[k] multiplies x[n - (CENTER-k)] + x[n + (CENTER-k)]
};

template <int CENTER, int R, bool SKIPZ>
__global__ __launch_bounds__(256) void
k_fir_phase(DdnFirArgs a, DdnTaps<CENTER> taps) {
    constexpr int T = 256 * R;
    constexpr int W = T + 2 * CENTER;
    constexpr int WP = W + W / R + 1;
    __shared__ f2 win[WP];
    __shared__ f2 ylast[256];

    const int tid = threadIdx.x;
    const int ch = blockIdx.y;
    const int blk = blockIdx.x / a.tiles_per_block;
    const int jt = blockIdx.x - blk * a.tiles_per_block;
    const long blk_start = (long)blk * a.block_len;
    long blk_end = blk_start + a.block_len;
    if (blk_end > a.n) {
        blk_end = a.n;
    }
    const long start = blk_start + (long)jt * T;
    if (start >= blk_end) {
        return; // short last block: surplus tiles
    }
    const int valid = (int)((blk_end - start) < T ? (blk_end - start) : T);

    // ---- stage the window -------------------------------------------------------------------------
    const f2* carry = a.carry + (size_t)ch * DDN_CARRY_LEN;
    for (int i = tid; i < W; i += 256) {
        long p = start - CENTER + i;
        f2 v;
        if (p < 0) {
            v = carry[DDN_CARRY_LEN + p]; // last samples of the previous call (zeros on a fresh stream)
        } else {
            if (p > blk_end - 1) {
                p = blk_end - 1; // the reference replicates the block's last sample
            }
            v = ddn_load_iq(a.in, a.in_fmt, a.ch_stride, ch, p);
        }
        win[i + i / R] = v;
    }
    __syncthreads();

    // ---- symmetric FIR, R outputs per thread, sliding register windows ------------------------------
#define PH(off) ((off) + (off) / R)
    const f2* w = win + tid * (R + 1);
    f2 acc[R], xm[R], xp[R];
    const f2 zero = {0.0f, 0.0f};
    const f2 hc = {taps.centre, taps.centre};
#pragma unroll
    for (int j = 0; j < R; j++) {
        acc[j] = __builtin_elementwise_fma(hc, w[PH(CENTER + j)], zero);
        xm[j] = w[PH(j)];
        xp[j] = w[PH(2 * CENTER + j)];
    }
#pragma unroll
    for (int k = 0; k < CENTER; k++) {
        const float h = taps.side[k];
        if (!SKIPZ || h != 0.0f) {
            const f2 hh = {h, h};
#pragma unroll
            for (int j = 0; j < R; j++) {
                acc[j] = __builtin_elementwise_fma(hh, xm[j] + xp[j], acc[j]);
            }
        }
        if (k + 1 < CENTER) {
#pragma unroll
            for (int j = 0; j < R - 1; j++) {
                xm[j] = xm[j + 1];
            }
            xm[R - 1] = w[PH(k + 1 + R - 1)];
#pragma unroll
            for (int j = R - 1; j > 0; j--) {
                xp[j] = xp[j - 1];
            }
            xp[0] = w[PH(2 * CENTER - (k + 1))];
        }
    }
#undef PH

    // ---- phase delta against the previous output ----------------------------------------------------
    ylast[tid] = acc[R - 1];
    __syncthreads();
    f2 prev = (tid > 0) ? ylast[tid - 1] : zero;
    float fq[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        fq[j] = ddn_phase_delta(acc[j], prev);
        prev = acc[j];
    }
    // tile edges: first output (the serial kernel forms its phase delta against the previous tile's last
    // output) and last valid output
    f2* edge = a.tile_edge + ((size_t)ch * a.n_tiles + blockIdx.x) * 2;
    if (tid == 0) {
        edge[0] = acc[0];
        fq[0] = 0.0f;
    }
    const int lastv = valid - 1;
    if (tid == lastv / R) {
        f2 yl = acc[0];
#pragma unroll
        for (int j = 1; j < R; j++) {
            if (j == lastv % R) {
                yl = acc[j];
            }
        }
        edge[1] = yl;
    }

    // ---- block power for the squelch gate: first <=512 floats of the block's LPF output, sequential
    //      binary64 accumulation exactly like mean_power() -------------------------------------------
    if (a.squelch_on && jt == 0) {
        __syncthreads();
        f2* ybuf = win; // window no longer needed
        if (tid * R < 256) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (tid * R + j < 256) {
                    ybuf[tid * R + j] = acc[j];
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            const int len = (valid * 2 > 512) ? 512 : valid * 2;
            const float* s = (const float*)ybuf;
            double p = 0.0, t = 0.0;
            for (int i = 0; i < len; i++) {
                const double v = (double)s[i];
                t += v;
                p += v * v;
            }
            const double dc = (t * t) / (double)len;
            double e = p - dc;
            if (e < 0.0) {
                e = 0.0;
            }
            a.blk_pwr[(size_t)ch * a.n_blocks + blk] = (float)(e / (double)len);
        }
    }

    // ---- store raw phase deltas ----------------------------------------------------------------------
    float* o = a.out + (size_t)ch * a.out_stride + start + (long)tid * R;
    if ((tid + 1) * R <= valid) {
        if constexpr ((R % 4) == 0) {
            if ((((size_t)o) & 15) == 0) {
#pragma unroll
                for (int j = 0; j < R; j += 4) {
                    float4 v4 = make_float4(fq[j], fq[j + 1], fq[j + 2], fq[j + 3]);
                    *(float4*)(o + j) = v4;
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; j++) {
                    o[j] = fq[j];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) {
                o[j] = fq[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (tid * R + j < valid) {
                o[j] = fq[j];
            }
        }
    }
}

// Generic-centre variant (any odd taps_len <= 143): one output per thread per step, taps in LDS.  Used for
// sample rates whose tap count has no unrolled instance.  Same arithmetic order.
__global__ __launch_bounds__(256) void
k_fir_phase_generic(DdnFirArgs a, const float* __restrict__ taps_dev, int center) {
    constexpr int T = 1024;
    __shared__ f2 win[T + 2 * DDN_MAX_CENTER];
    __shared__ f2 ybuf[T];
    __shared__ float stap[DDN_MAX_CENTER + 1];
    const int tid = threadIdx.x;
    const int ch = blockIdx.y;
    const int blk = blockIdx.x / a.tiles_per_block;
    const int jt = blockIdx.x - blk * a.tiles_per_block;
    const long blk_start = (long)blk * a.block_len;
    long blk_end = blk_start + a.block_len;
    if (blk_end > a.n) {
        blk_end = a.n;
    }
    const long start = blk_start + (long)jt * T;
    if (start >= blk_end) {
        return;
    }
    const int valid = (int)((blk_end - start) < T ? (blk_end - start) : T);
    const int W = T + 2 * center;
    const f2* carry = a.carry + (size_t)ch * DDN_CARRY_LEN;
    for (int i = tid; i < W; i += 256) {
        long p = start - center + i;
        f2 v;
        if (p < 0) {
            v = carry[DDN_CARRY_LEN + p];
        } else {
            if (p > blk_end - 1) {
                p = blk_end - 1;
            }
            v = ddn_load_iq(a.in, a.in_fmt, a.ch_stride, ch, p);
        }
        win[i] = v;
    }
    if (tid <= center) {
        stap[tid] = taps_dev[tid];
    }
    __syncthreads();
    const f2 zero = {0.0f, 0.0f};
    for (int o = tid; o < T; o += 256) {
        const f2 hc = {stap[center], stap[center]};
        f2 acc = __builtin_elementwise_fma(hc, win[o + center], zero);
        for (int k = 0; k < center; k++) {
            const float h = stap[k];
            if (h == 0.0f) {
                continue;
            }
            const int d = center - k;
            const f2 hh = {h, h};
            acc = __builtin_elementwise_fma(hh, win[o + center - d] + win[o + center + d], acc);
        }
        ybuf[o] = acc;
    }
    __syncthreads();
    f2* edge = a.tile_edge + ((size_t)ch * a.n_tiles + blockIdx.x) * 2;
    if (tid == 0) {
        edge[0] = ybuf[0];
        edge[1] = ybuf[valid - 1];
    }
    float* out = a.out + (size_t)ch * a.out_stride + start;
    for (int o = tid; o < valid; o += 256) {
        out[o] = (o == 0) ? 0.0f : ddn_phase_delta(ybuf[o], ybuf[o - 1]);
    }
    if (a.squelch_on && jt == 0 && tid == 0) {
        const int len = (valid * 2 > 512) ? 512 : valid * 2;
        const float* s = (const float*)ybuf;
        double p = 0.0, t = 0.0;
        for (int i = 0; i < len; i++) {
            const double v = (double)s[i];
            t += v;
            p += v * v;
        }
        const double dc = (t * t) / (double)len;
        double e = p - dc;
        if (e < 0.0) {
            e = 0.0;
        }
        a.blk_pwr[(size_t)ch * a.n_blocks + blk] = (float)(e / (double)len);
    }
}

// ------------------------------------------------------------------------------------------------------
// K2: the per-channel serial recurrences.  One workgroup owns G channels and walks time in tiles of TT
// samples: all 256 threads move a [G][TT] tile of raw phase deltas HBM -> LDS (coalesced along time),
// lanes 0..G-1 of wave 0 run the dc / peak recurrences sample by sample (exact reference order), then all
// threads move the finished tile LDS -> HBM.

template <int G, int TT>
__global__ __launch_bounds__(256) void
k_fsk_serial(DdnSerialArgs a) {
    __shared__ float f[G][TT + 1];
    const int tid = threadIdx.x;
    const int ch0 = blockIdx.x * G;
    const int nch = (a.n_channels - ch0) < G ? (a.n_channels - ch0) : G;

    // per-lane carried state
    float prev_i = 0.f, prev_q = 0.f, dc = 0.f, peak = 0.f;
    int have_prev = 0;
    const int my = (tid < nch) ? (ch0 + tid) : -1;
    if (my >= 0) {
        const DdnFskState s = a.state[my];
        prev_i = s.prev_i;
        prev_q = s.prev_q;
        have_prev = s.have_prev;
        dc = s.dc_est;
        peak = s.peak_est;
    }

    for (long t0 = 0; t0 < a.n; t0 += TT) {
        const int tn = (int)((a.n - t0) < TT ? (a.n - t0) : TT);
        // ---- load tile; patch K1-tile-first samples with the cross-tile phase delta ------------------
        for (int idx = tid; idx < G * TT; idx += 256) {
            const int g = idx / TT, t = idx - g * TT;
            if (g < nch && t < tn) {
                const long p = t0 + t;
                float v = a.buf[(size_t)(ch0 + g) * a.stride + p];
                const long blk = p / a.block_len;
                const long rel = p - blk * a.block_len;
                if ((rel % a.fir_tile) == 0) {
                    const long tile = blk * a.tiles_per_block + rel / a.fir_tile;
                    const f2* e = a.tile_edge + ((size_t)(ch0 + g) * a.n_tiles + tile) * 2;
                    const f2 cur = e[0];
                    if (p > 0) {
                        long ptile = tile - 1;
                        if (rel == 0) { // previous block may have fewer tiles in use: its last tile
                            const long pblk = blk - 1;
                            ptile = pblk * a.tiles_per_block + (a.block_len - 1) / a.fir_tile;
                        }
                        const f2 pv = (a.tile_edge + ((size_t)(ch0 + g) * a.n_tiles + ptile) * 2)[1];
                        v = ddn_phase_delta(cur, pv);
                    } else {
                        const DdnFskState s = a.state[ch0 + g];
                        const f2 pv = {s.prev_i, s.prev_q};
                        v = ddn_phase_delta(cur, pv); // only used when have_prev
                    }
                }
                f[g][t] = v;
            }
        }
        __syncthreads();
        // ---- serial part -----------------------------------------------------------------------------
        if (my >= 0) {
            for (int t = 0; t < tn; t++) {
                const long p = t0 + t;
                const long blk = p / a.block_len;
                const long rel = p - blk * a.block_len;
                if (a.squelch_on) {
                    const float pw = a.blk_pwr[(size_t)my * a.n_blocks + blk];
                    if (pw < a.squelch_level) {
                        // squelched block: zeros out, modem reset (src/dsp/demod_pipeline.cpp:1179-1184)
                        have_prev = 0;
                        dc = 0.f;
                        peak = 0.f;
                        prev_i = 0.f;
                        prev_q = 0.f;
                        f[tid][t] = 0.0f;
                        continue;
                    }
                }
                (void)rel;
                if (!have_prev) {
                    have_prev = 1;
                    f[tid][t] = 0.0f;
                    continue;
                }
                const float fr = f[tid][t];
                dc += 0.00025f * (fr - dc);
                const float c = fr - dc;
                const float mag = fabsf(c);
                if (mag > 1.0e-7f) {
                    if (peak <= 1.0e-7f) {
                        peak = mag;
                    } else if (mag > peak) {
                        peak += 0.125f * (mag - peak);
                    } else {
                        peak += 0.00005f * (mag - peak);
                    }
                }
                float pk = peak;
                if (pk <= 1.0e-7f) {
                    pk = 1.0f;
                }
                float y = c * (30000.0f / pk);
                if (y > 32767.0f) {
                    y = 32767.0f;
                } else if (y < -32768.0f) {
                    y = -32768.0f;
                }
                f[tid][t] = y;
            }
        }
        __syncthreads();
        // ---- store ----------------------------------------------------------------------------------
        for (int idx = tid; idx < G * TT; idx += 256) {
            const int g = idx / TT, t = idx - g * TT;
            if (g < nch && t < tn) {
                a.buf[(size_t)(ch0 + g) * a.stride + t0 + t] = f[g][t];
            }
        }
        __syncthreads();
    }
    if (my >= 0 && a.n > 0) {
        // prev sample = last LPF output of this call (only meaningful when have_prev)
        const long lastp = a.n - 1;
        const long blk = lastp / a.block_len;
        const long rel = lastp - blk * a.block_len;
        const long tile = blk * a.tiles_per_block + rel / a.fir_tile;
        const f2 yl = (a.tile_edge + ((size_t)my * a.n_tiles + tile) * 2)[1];
        DdnFskState s;
        s.prev_i = have_prev ? yl.x : 0.f;
        s.prev_q = have_prev ? yl.y : 0.f;
        s.have_prev = have_prev;
        s.dc_est = dc;
        s.peak_est = peak;
        a.state[my] = s;
    }
}

// FIR history carry: the last DDN_CARRY_LEN widened input samples of each channel, for the next call.
__global__ void
k_carry_update(const void* in, int in_fmt, size_t ch_stride, long n, f2* carry) {
    const int ch = blockIdx.x;
    const int i = threadIdx.x; // 0..DDN_CARRY_LEN-1
    f2* c = carry + (size_t)ch * DDN_CARRY_LEN;
    const long p = n - DDN_CARRY_LEN + i;
    f2 v;
    if (p >= 0) {
        v = ddn_load_iq(in, in_fmt, ch_stride, ch, p);
    } else {
        v = c[i + n]; // shift older history down
    }
    __syncthreads();
    c[i] = v;
}

__global__ void
k_zero_u32(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        p[i] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers (called from ddn_api.cpp)

template <int CENTER, int R>
static hipError_t
launch_fir_t(const DdnFirArgs& a, const float* taps, bool has_zero, dim3 grid, hipStream_t st) {
    DdnTaps<CENTER> tp;
    tp.centre = taps[CENTER];
    for (int k = 0; k < CENTER; k++) {
        tp.side[k] = taps[k];
    }
    if (has_zero) {
        hipLaunchKernelGGL((k_fir_phase<CENTER, R, true>), grid, dim3(256), 0, st, a, tp);
    } else {
        hipLaunchKernelGGL((k_fir_phase<CENTER, R, false>), grid, dim3(256), 0, st, a, tp);
    }
    return hipGetLastError();
}

extern "C" int
ddn_dev_fir_tile(int center) {
    switch (center) {
        case 67: return 256 * DDN_FIR_R;
        case 33: return 256 * DDN_FIR_R;
        default: return 1024;
    }
}

extern "C" hipError_t
ddn_dev_launch_fir(const DdnFirArgs* a, const float* taps_host, const float* taps_dev, int center, int n_channels,
                   hipStream_t st) {
    bool has_zero = false;
    for (int k = 0; k < center; k++) {
        if (taps_host[k] == 0.0f) {
            has_zero = true;
        }
    }
    dim3 grid((unsigned)a->n_tiles, (unsigned)n_channels);
    switch (center) {
        case 67: return launch_fir_t<67, DDN_FIR_R>(*a, taps_host, has_zero, grid, st);
        case 33: return launch_fir_t<33, DDN_FIR_R>(*a, taps_host, has_zero, grid, st);
        default:
            hipLaunchKernelGGL(k_fir_phase_generic, grid, dim3(256), 0, st, *a, taps_dev, center);
            return hipGetLastError();
    }
}

extern "C" hipError_t
ddn_dev_launch_serial(const DdnSerialArgs* a, hipStream_t st) {
    constexpr int G = 16;
    dim3 grid((unsigned)((a->n_channels + G - 1) / G));
    hipLaunchKernelGGL((k_fsk_serial<G, 256>), grid, dim3(256), 0, st, *a);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_launch_carry(const void* in, int in_fmt, size_t ch_stride, long n, void* carry, int n_channels,
                     hipStream_t st) {
    hipLaunchKernelGGL(k_carry_update, dim3((unsigned)n_channels), dim3(DDN_CARRY_LEN), 0, st, in, in_fmt, ch_stride,
                       n, (f2*)carry);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_zero(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n);
    return hipGetLastError();
}
