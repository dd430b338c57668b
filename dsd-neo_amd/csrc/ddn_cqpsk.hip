// ddn_cqpsk.hip — batched P25 CQPSK/LSM chain around the Gardner kernel (ddn_ted.hip):
//
//   k_channel_lpf_c2c   channel LPF, complex in -> complex out (channel_lpf_apply, src/dsp/demod_pipeline.cpp:526-555 ->
//                       simd_fir_complex_apply, src/dsp/simd_fir.cpp:55-133: zero latency, block-edge replication, FMA
//                       order of the AVX2 unit for blocks >= taps_len samples, (mul, add) order for shorter ones)
//   k_cqpsk_agc_fll     RMS AGC (src/dsp/demod_pipeline.cpp:796-842) + FLL band-edge (src/dsp/costas.cpp:636-781,1176-1207,
//                       NCO polynomial :80-133), sample rate, one channel per lane
//   k_cqpsk_symbols     differential phasor (costas.cpp:871-901) + Costas loop (:536-603,934-962, detector :179-259) +
//                       phase extractor theta*4/pi (demod_pipeline.cpp:63-98,742-764), symbol rate, one channel per lane
// Chain order: demod_pipeline.cpp:1100-1118,1250-1257.  Everything except the LPF is a feedback recurrence, parallel
// across channels only; staging follows ddn_ted.hip (loader wave, [slot][lane] LDS state).  The FLL kernel rotates three
// LDS tiles: while the recurrence wave works on tile t in place, the second wave stores the finished tile t-1 to HBM
// with coalesced row writes and prefetches tile t+1.

#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#include "ddn_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));

namespace {
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kPi = 3.14159265358979323846f;

// FLL band-edge taps are a per-batch device buffer `fll` = [4][DDN_FLL_MAX_TAPS]: lower_r, lower_i, upper_r, upper_i
// (reversed, as the reference), uploaded once when the batch is created - no module-global state, so batches with
// different sps can run on different streams / threads at once

__device__ __forceinline__ float
clipf(float x, float lim) {
    return x > lim ? lim : (x < -lim ? -lim : x);
}
__device__ __forceinline__ float
clampr(float v, float lo, float hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}
__device__ __forceinline__ bool
finitef(float x) {
    return fabsf(x) <= 3.4028234663852886e38f; // false for NaN and +-inf
}

__device__ __forceinline__ void
sincos_half_pi(float phase, float* s, float* c) {
    const float x2 = phase * phase;
    *s = phase
         * (1.0f
            + x2
                  * (-0.16666666666666666667f
                     + x2
                           * (0.00833333333333333333f
                              + x2 * (-0.00019841269841269841f + x2 * (0.00000275573192239859f + x2 * -0.00000002505210838544f)))));
    *c = 1.0f
         + x2
               * (-0.5f
                  + x2
                        * (0.04166666666666666667f
                           + x2 * (-0.00138888888888888889f + x2 * (0.00002480158730158730f + x2 * -0.00000027557319223986f))));
}

// phase is kept inside [-2pi, 2pi] by the loop, which is the only range the reference's polynomial path covers
// (outside it the reference calls libm; a non-finite phase cannot be reproduced and is passed through the same
// polynomial so that the result is non-finite as well).
__device__ __forceinline__ void
sincos_two_pi(float phase, float* s, float* c) {
    // select-only form of the reference's three-way branch (src/dsp/costas.cpp:102-133): one polynomial evaluation on
    // the reflected argument, cosine negated outside [-pi/2, pi/2] - the same operations on the same values, but lanes
    // in different quadrants no longer serialise three copies of the polynomial
    const float p = phase > kPi ? phase - kTwoPi : (phase < -kPi ? phase + kTwoPi : phase);
    const bool hi = p > (kPi / 2.0f), lo = p < (-kPi / 2.0f);
    const float arg = hi ? (kPi - p) : (lo ? (-kPi - p) : p);
    float sv, cv;
    sincos_half_pi(arg, &sv, &cv);
    *s = sv;
    *c = (hi || lo) ? -cv : cv;
}

__device__ __forceinline__ float
smoothstep(float e0, float e1, float x) {
    if (x <= e0) {
        return 0.0f;
    }
    if (x >= e1) {
        return 1.0f;
    }
    const float t = (x - e0) / (e1 - e0);
    return t * t * (3.0f - 2.0f * t);
}

__device__ __forceinline__ float
atan_unit(float x) {
    const float ax = fabsf(x);
    return x * (0.78539816339744830962f - (ax - 1.0f) * (0.2447f + 0.0663f * ax));
}

__device__ __forceinline__ float
atan2_qpsk(float y, float x) {
    if (x == 0.0f && y == 0.0f) {
        return 0.0f;
    }
    const float ax = fabsf(x), ay = fabsf(y);
    if (ax >= ay) {
        float a = atan_unit(y / x);
        if (x < 0.0f) {
            a += (y < 0.0f) ? -3.14159265358979323846f : 3.14159265358979323846f;
        }
        return a;
    }
    const float a = atan_unit(x / y);
    return (y > 0.0f) ? (1.57079632679489661923f - a) : (-1.57079632679489661923f - a);
}

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

// ---- channel LPF, complex -> complex -------------------------------------------------------------------------------
// One output per thread; the 256 outputs of a workgroup share a (256 + taps_len - 1)-sample window staged in LDS with
// the block-edge rule already applied (left: previous samples of the stream / carried history; right: the block's last
// sample replicated).  A workgroup never straddles a block: grid.x enumerates (block, tile-in-block).
template <int FMT>
__global__ __launch_bounds__(256) void
k_channel_lpf_c2c(const void* __restrict__ in, long n, size_t in_stride, int block_len, int tiles_per_block,
                  const float* __restrict__ taps_g, int taps_len, const f2* __restrict__ hist, f2* __restrict__ out,
                  size_t out_stride) {
    __shared__ f2 win[256 + DDN_MAX_TAPS];
    __shared__ float taps[DDN_MAX_TAPS + 1];
    const int ch = blockIdx.y;
    const long b = blockIdx.x / tiles_per_block;
    const int tile = blockIdx.x % tiles_per_block;
    const long s = b * block_len;
    if (s >= n) {
        return;
    }
    const long L = (n - s) < block_len ? (n - s) : block_len;
    const long t0 = s + (long)tile * 256;
    if (t0 >= s + L) {
        return;
    }
    const int H = taps_len - 1, CEN = H / 2;
    const long last = s + L - 1;
    const bool fused = L >= taps_len;
    for (int i = threadIdx.x; i < taps_len; i += 256) {
        taps[i] = taps_g[i];
    }
    for (int i = threadIdx.x; i < 256 + H; i += 256) {
        long j = t0 - CEN + i;
        j = j > last ? last : j;
        f2 v;
        if (j < 0) {
            v = hist[(size_t)ch * H + (size_t)(H + j)];
        } else {
            v = load_iq<FMT>(in, (size_t)ch * in_stride + (size_t)j);
        }
        win[i] = v;
    }
    __syncthreads();
    const long m = t0 + threadIdx.x;
    if (m > last) {
        return;
    }
    const int c = threadIdx.x + CEN;
    const f2 z = {0.0f, 0.0f};
    f2 acc;
    {
        const f2 h = {taps[CEN], taps[CEN]};
        acc = fused ? __builtin_elementwise_fma(h, win[c], z) : (z + h * win[c]);
    }
    for (int k = 0; k < CEN; k++) {
        const float hk = taps[k];
        if (hk == 0.0f) {
            continue;
        }
        const int d = CEN - k;
        const f2 sm = win[c - d] + win[c + d];
        const f2 h = {hk, hk};
        acc = fused ? __builtin_elementwise_fma(h, sm, acc) : (acc + h * sm);
    }
    out[(size_t)ch * out_stride + (size_t)m] = acc;
}

// Same filter for the tap counts the channel-LPF design produces at the usual rates (135 taps at 48 kHz, 67 at 24 kHz):
// the tap loop is fully unrolled, a thread owns two adjacent outputs and slides both symmetric window pairs through
// registers (one new low-side and one new high-side sample per tap instead of four reads), taps come in as scalar operands.
// Per output the operation order is the generic kernel's: centre tap, then (x[c-d] + x[c+d]) * h in ascending tap order.
template <int FMT, int CEN_T, bool SKIPZ>
__global__ __launch_bounds__(128) void
k_channel_lpf_c2c_u(const void* __restrict__ in, long n, size_t in_stride, int block_len, int tiles_per_block,
                    const float* __restrict__ taps_g, const f2* __restrict__ hist, f2* __restrict__ out,
                    size_t out_stride) {
    constexpr int H = 2 * CEN_T;
    __shared__ f2 win[256 + H + 2];
    const int ch = blockIdx.y;
    const long b = blockIdx.x / tiles_per_block;
    const int tile = blockIdx.x % tiles_per_block;
    const long s = b * block_len;
    if (s >= n) {
        return;
    }
    const long L = (n - s) < block_len ? (n - s) : block_len;
    const long t0 = s + (long)tile * 256;
    if (t0 >= s + L) {
        return;
    }
    const long last = s + L - 1;
    const bool fused = L >= H + 1;
    for (int i = threadIdx.x; i < 256 + H; i += 128) {
        long j = t0 - CEN_T + i;
        j = j > last ? last : j;
        win[i] = (j < 0) ? hist[(size_t)ch * H + (size_t)(H + j)] : load_iq<FMT>(in, (size_t)ch * in_stride + (size_t)j);
    }
    __syncthreads();
    const long m = t0 + 2 * threadIdx.x;
    if (m > last) {
        return;
    }
    const f2* w = &win[2 * threadIdx.x]; // w[CEN_T] is output m's centre sample
    const f2 z = {0.0f, 0.0f};
    f2 acc0, acc1;
    // blocks of at least taps_len samples take the AVX2 unit's FMA order, shorter ones the scalar unit's multiply-then-add;
    // the choice is uniform over the workgroup, so each order gets its own straight-line tap loop
    auto run = [&](auto fused_c) {
        constexpr bool FUSED = decltype(fused_c)::value;
        const float hc = taps_g[CEN_T];
        const f2 hcc = {hc, hc};
        acc0 = FUSED ? __builtin_elementwise_fma(hcc, w[CEN_T], z) : (z + hcc * w[CEN_T]);
        acc1 = FUSED ? __builtin_elementwise_fma(hcc, w[CEN_T + 1], z) : (z + hcc * w[CEN_T + 1]);
        f2 a0 = w[0], a1 = w[1];                     // low side: x[c - d], x[c - d + 1] for d = CEN_T
        f2 b0 = w[2 * CEN_T], b1 = w[2 * CEN_T + 1]; // high side: x[c + d], x[c + d + 1]
#pragma unroll
        for (int k = 0; k < CEN_T; k++) {
            const float hk = taps_g[k];
            if (!SKIPZ || hk != 0.0f) {
                const f2 h = {hk, hk};
                acc0 = FUSED ? __builtin_elementwise_fma(h, a0 + b0, acc0) : (acc0 + h * (a0 + b0));
                acc1 = FUSED ? __builtin_elementwise_fma(h, a1 + b1, acc1) : (acc1 + h * (a1 + b1));
            }
            if (k + 1 < CEN_T) {
                a0 = a1;
                a1 = w[k + 2];
                b1 = b0;
                b0 = w[2 * CEN_T - (k + 1)];
            }
        }
    };
    if (fused) {
        run(std::true_type{});
    } else {
        run(std::false_type{});
    }
    out[(size_t)ch * out_stride + (size_t)m] = acc0;
    if (m + 1 <= last) {
        out[(size_t)ch * out_stride + (size_t)m + 1] = acc1;
    }
}

template <int FMT>
__global__ void
k_lpf_hist(const void* __restrict__ in, long n, size_t in_stride, int H, f2* __restrict__ hist) {
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

// ---- RMS AGC + FLL band-edge (sample rate) -------------------------------------------------------------------------
// Four lanes per channel: lane q of a quad owns one of the four band-edge accumulators (lower_r, lower_i, upper_r,
// upper_i).  Each accumulator is the ordered sum of t_k = d_r[k] * A_q[k] + d_i[k] * B_q[k] with (A, B) = (l_r, -l_i),
// (l_i, l_r), (u_r, -u_i), (u_i, u_r): x - y and x + (-y) round identically, so this is the reference's
// `acc += dr * a - di * b` / `acc += dr * b + di * a` (src/dsp/costas.cpp:677-690) with the per-sample instruction stream
// cut from 16 to 4 VALU per tap; the quad then swaps the four sums with wave shuffles and every lane advances the same
// (replicated) loop state.  16 channels per wavefront, so 4096 channels spread over 256 wavefronts.
// NT_T > 0: tap count known at compile time -> the lane's 2 * NT taps live in registers and the 2 * NT delay-line reads
// of a sample are issued back to back before the (order-preserving) accumulation.
template <int NT_T>
__global__ __launch_bounds__(128) void
k_cqpsk_agc_fll(const f2* __restrict__ in, long n, size_t stride, int n_channels, int nt_rt, float alpha, float beta,
                const float* __restrict__ fll, DdnCqpskState* __restrict__ state, float* __restrict__ delay_store,
                f2* __restrict__ out) {
    constexpr int TS = 32, CPW = 16;
    const int nt = NT_T > 0 ? NT_T : nt_rt;
    extern __shared__ float smem[];
    f2* tiles = (f2*)smem;                             // [3][CPW][TS + 1]
    float* dlr = (float*)(tiles + 3 * CPW * (TS + 1)); // [2 nt][CPW]
    float* dli = dlr + 2 * nt * CPW;                   // [2 nt][CPW]
    const int lane = threadIdx.x & 63;
    const bool helper = threadIdx.x >= 64;
    const int ch0 = blockIdx.x * CPW;
    const int cl = lane >> 2, q = lane & 3; // channel slot in the workgroup, accumulator owned
    const int ch = ch0 + cl;
    const bool live = !helper && ch < n_channels;
    DdnCqpskState s = {};
    if (live) {
        s = state[ch];
        if (q == 0) {
            for (int k = 0; k < 2 * nt; k++) {
                dlr[k * CPW + cl] = delay_store[((size_t)k * 2) * n_channels + ch];
                dli[k * CPW + cl] = delay_store[((size_t)k * 2 + 1) * n_channels + ch];
            }
        }
    }
    // this lane's tap pair: A multiplies the delayed real part, B the delayed imaginary part
    const int ia = (q == 0) ? 0 : ((q == 1) ? 1 : ((q == 2) ? 2 : 3));
    const int ib = (q == 0) ? 1 : ((q == 1) ? 0 : ((q == 2) ? 3 : 2));
    const float sb = (q == 0 || q == 2) ? -1.0f : 1.0f;
    float ta[NT_T > 0 ? NT_T : 1], tb[NT_T > 0 ? NT_T : 1];
    if (NT_T > 0) {
#pragma unroll
        for (int k = 0; k < NT_T; k++) {
            ta[k] = fll[ia * DDN_FLL_MAX_TAPS + k];
            tb[k] = sb * fll[ib * DDN_FLL_MAX_TAPS + k];
        }
    }
    float avg = s.agc_avg;
    if (avg <= 0.0f) {
        avg = 1.0f;
    }
    float phase = s.fll_phase, freq = s.fll_freq;
    int idx = s.fll_idx;
    auto stage = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        const int half = lane >> 5, col = lane & 31; // two rows per pass: 32 lanes x 8 B = one 256-B row segment
#pragma unroll
        for (int r = 0; r < CPW; r += 2) {
            const int cc = r + half;
            f2 v = {0.0f, 0.0f};
            if (ch0 + cc < n_channels && col < tn) {
                v = in[(size_t)(ch0 + cc) * stride + (size_t)t0 + col];
            }
            tiles[(buf * CPW + cc) * (TS + 1) + col] = v;
        }
    };
    auto drain = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        const int half = lane >> 5, col = lane & 31;
#pragma unroll
        for (int r = 0; r < CPW; r += 2) {
            const int cc = r + half;
            if (ch0 + cc < n_channels && col < tn) {
                out[(size_t)(ch0 + cc) * stride + (size_t)t0 + col] = tiles[(buf * CPW + cc) * (TS + 1) + col];
            }
        }
    };
    __syncthreads();
    if (helper && n > 0) {
        stage(0, 0);
    }
    __syncthreads();
    long t0 = 0;
    int it = 0;
    const int qb = lane & ~3;
    for (; t0 < n; t0 += TS, it++) {
        const int buf = it % 3;
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        if (helper) {
            if (it > 0) {
                drain(t0 - TS, (it + 2) % 3);
            }
            if (t0 + TS < n) {
                stage(t0 + TS, (it + 1) % 3);
            }
        } else if (live) {
            f2* row = tiles + (buf * CPW + cl) * (TS + 1);
            for (int s_i = 0; s_i < tn; s_i++) {
                f2 x = row[s_i];
                // RMS AGC
                const float m2 = x.x * x.x + x.y * x.y;
                avg = 0.55f * avg + 0.45f * m2;
                if (avg > 0.0f) {
                    const float sc = 0.85f / sqrtf(avg);
                    x.x = x.x * sc;
                    x.y = x.y * sc;
                }
                // FLL: NCO rotation, band-edge filters on the rotated stream, loop update
                float ns, nc;
                sincos_two_pi(phase, &ns, &nc);
                const float orr = x.x * nc - x.y * ns;
                const float oi = x.x * ns + x.y * nc;
                if (q == 0) {
                    dlr[idx * CPW + cl] = orr;
                    dli[idx * CPW + cl] = oi;
                    dlr[(idx + nt) * CPW + cl] = orr;
                    dli[(idx + nt) * CPW + cl] = oi;
                }
                __builtin_amdgcn_wave_barrier();
                float acc = 0.0f;
                const int base = idx + nt;
                if (NT_T > 0) {
                    float vr[NT_T > 0 ? NT_T : 1], vi[NT_T > 0 ? NT_T : 1];
#pragma unroll
                    for (int k = 0; k < NT_T; k++) {
                        vr[k] = dlr[(base - k) * CPW + cl];
                        vi[k] = dli[(base - k) * CPW + cl];
                    }
#pragma unroll
                    for (int k = 0; k < NT_T; k++) {
                        acc += vr[k] * ta[k] + vi[k] * tb[k];
                    }
                } else {
                    for (int k = 0; k < nt; k++) {
                        const float dr = dlr[(base - k) * CPW + cl], di = dli[(base - k) * CPW + cl];
                        acc += dr * fll[ia * DDN_FLL_MAX_TAPS + k] + di * (sb * fll[ib * DDN_FLL_MAX_TAPS + k]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const float lr = __shfl(acc, qb + 0), li = __shfl(acc, qb + 1);
                const float ur = __shfl(acc, qb + 2), ui = __shfl(acc, qb + 3);
                idx = (idx + 1 == nt) ? 0 : idx + 1;
                const float lm = lr * lr + li * li, um = ur * ur + ui * ui;
                const float err = clipf(um - lm, 1.0f);
                freq += beta * err;
                freq = clampr(freq, -1.0f, 1.0f);
                phase += freq + alpha * err;
                // the reference's while loops run at most once: |freq| <= 1 and alpha * |err| < 0.1 per sample
                phase = phase > kTwoPi ? phase - kTwoPi : phase;
                phase = phase < -kTwoPi ? phase + kTwoPi : phase;
                if (q == 0) {
                    const f2 y = {orr, oi};
                    row[s_i] = y;
                }
            }
        }
        __syncthreads();
    }
    if (helper && it > 0) {
        drain(t0 - TS, (it + 2) % 3);
    }
    if (live && q == 0) {
        s.agc_avg = avg;
        s.fll_phase = phase;
        s.fll_freq = freq;
        s.fll_idx = idx;
        state[ch] = s;
        for (int k = 0; k < 2 * nt; k++) {
            delay_store[((size_t)k * 2) * n_channels + ch] = dlr[k * CPW + cl];
            delay_store[((size_t)k * 2 + 1) * n_channels + ch] = dli[k * CPW + cl];
        }
    }
}

// Register-resident variant for the tap counts in use (2 sps + 1 = 9, 11, 21).  The per-sample feedback loop issues its
// instructions in order, so what limits it is the LDS round trips inside the loop (delay-line write -> read, shuffles),
// not the FLOPs: here the delay line is a circular buffer in REGISTERS - the sample loop is unrolled by NT so the write
// slot and every tap's read slot are compile-time register names - the four accumulators are exchanged with DPP
// quad-permutes (VALU, no LDS), and a tile holds a whole number of NT-sample chunks (TS = 3 NT or 2 NT) so the circular
// alignment survives tile boundaries; only the last tile of a call can end mid-chunk, and the carried state stores the
// delay line oldest-first so the next call starts aligned again.
// (round 6) The loop wave is bound by its own instruction stream - one wavefront issues a vector instruction every ~4.7 cycles and
// there is one such wave per sixteen channels, whatever the batch - so the work per sample is cut where the arithmetic allows:
//   * the RMS AGC does not depend on the frequency loop: the helper wave (which stages and drains the tiles) runs it, lane = channel,
//     on the tile it has just staged, a tile ahead of the loop wave (the `avg` recurrence is its own; sqrt and division off the loop
//     wave's stream);
//   * delay line and taps as (re, im) / (a, b) pairs: one packed multiply per tap gives both products, two adds follow
//     (acc += z.re a + z.im b with every product and sum rounded as before) - three instructions a tap instead of four and the moves
//     that paired the registers.
template <int NT>
__global__ __launch_bounds__(128) void
k_cqpsk_agc_fll_reg(const f2* __restrict__ in, long n, size_t stride, int n_channels, float alpha, float beta,
                    const float* __restrict__ fll, DdnCqpskState* __restrict__ state, float* __restrict__ delay_store,
                    f2* __restrict__ out) {
    constexpr int TS = 3 * NT, CPW = 16; // (a tile is staged lane = sample: 3 NT <= 63 for the tap counts in use)
    static_assert(TS <= 64, "a tile is staged one sample per lane");
    __shared__ f2 tiles[3][CPW][TS + 1];
    const int lane = threadIdx.x & 63;
    const bool helper = threadIdx.x >= 64;
    const int ch0 = blockIdx.x * CPW;
    const int cl = lane >> 2, q = lane & 3;
    const int ch = ch0 + cl;
    const bool live = !helper && ch < n_channels;
    __shared__ float avg_end[CPW];
    __shared__ float agc_t[CPW][TS + 1]; // 0.45 |x|^2, then the running mean square, of the tile being gain-controlled
    DdnCqpskState s = {};
    f2 z[NT]; // z[i] = rotated sample written at chunk position i; before a chunk z[i] = x_(i - NT)
#pragma unroll
    for (int k = 0; k < NT; k++) {
        z[k].x = 0.0f;
        z[k].y = 0.0f;
    }
    if (live) {
        s = state[ch];
#pragma unroll
        for (int k = 0; k < NT; k++) { // stored oldest first
            z[k].x = delay_store[((size_t)k * 2) * n_channels + ch];
            z[k].y = delay_store[((size_t)k * 2 + 1) * n_channels + ch];
        }
    }
    const int ia = q, ib = q ^ 1;
    const float sb = (q == 0 || q == 2) ? -1.0f : 1.0f;
    f2 tt[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        tt[k].x = fll[ia * DDN_FLL_MAX_TAPS + k];
        tt[k].y = sb * fll[ib * DDN_FLL_MAX_TAPS + k];
    }
    // helper wave, lane = channel: the AGC's running mean square
    const bool agc_lane = helper && lane < CPW && ch0 + lane < n_channels;
    float avg = 1.0f;
    if (agc_lane) {
        avg = state[ch0 + lane].agc_avg;
        if (avg <= 0.0f) {
            avg = 1.0f;
        }
    }
    float phase = s.fll_phase, freq = s.fll_freq;
    // ps[k], k = 1 .. NT - 1: the NEXT sample's product sums of the taps that look back (z.re a + z.im b of tap k on the sample k
    // back) - all of them known as soon as the current sample is rotated, so they are formed between the dependent adds of the current
    // sample's chain instead of each in front of the add that needs it (a whole chunk at a time: see the sample loop)
    float ps[NT];
#pragma unroll
    for (int k = 1; k < NT; k++) { // the call's first sample sits at chunk position 0: tap k looks at slot NT - k
        const f2 p = z[NT - k] * tt[k];
        ps[k] = p.x + p.y;
    }
    ps[0] = 0.0f;
    auto stage = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
#pragma unroll
        for (int r = 0; r < CPW; r++) {
            f2 v = {0.0f, 0.0f};
            if (ch0 + r < n_channels && lane < tn) {
                v = in[(size_t)(ch0 + r) * stride + (size_t)t0 + lane];
            }
            if (lane < TS) {
                tiles[buf][r][lane] = v;
            }
        }
    };
    auto drain = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
#pragma unroll
        for (int r = 0; r < CPW; r++) {
            if (ch0 + r < n_channels && lane < tn) {
                out[(size_t)(ch0 + r) * stride + (size_t)t0 + lane] = tiles[buf][r][lane];
            }
        }
    };
    // RMS AGC of a staged tile, in place (the wave's own LDS writes of stage() are in the queue ahead of these reads).  Only the
    // running mean square is serial in time - two dependent operations a sample, lane = channel; its input 0.45 |x|^2 before it and the
    // gain 0.85 / sqrt(avg) after it are formed lane = sample (a serial sqrt + division per sample made this wave the slower one).
    auto agc_tile = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        __builtin_amdgcn_wave_barrier();
        if (lane < tn) {
#pragma unroll
            for (int r = 0; r < CPW; r++) {
                const f2 x = tiles[buf][r][lane];
                const float m2 = x.x * x.x + x.y * x.y;
                agc_t[r][lane] = 0.45f * m2;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (agc_lane) {
            float* row = &agc_t[lane][0];
#pragma unroll 8
            for (int i = 0; i < tn; i++) {
                avg = 0.55f * avg + row[i];
                row[i] = avg;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < tn) {
#pragma unroll
            for (int r = 0; r < CPW; r++) {
                const float a = agc_t[r][lane];
                if (ch0 + r < n_channels && a > 0.0f) {
                    const float sc = 0.85f / sqrtf(a);
                    f2 x = tiles[buf][r][lane];
                    x.x = x.x * sc;
                    x.y = x.y * sc;
                    tiles[buf][r][lane] = x;
                }
            }
        }
    };
    auto bcast = [&](float v, int src) -> float { // value of lane `src` of this quad, via DPP quad_perm
        int iv = __float_as_int(v), r;
        switch (src) {
            case 0: r = __builtin_amdgcn_update_dpp(0, iv, 0x00, 0xF, 0xF, false); break;
            case 1: r = __builtin_amdgcn_update_dpp(0, iv, 0x55, 0xF, 0xF, false); break;
            case 2: r = __builtin_amdgcn_update_dpp(0, iv, 0xAA, 0xF, 0xF, false); break;
            default: r = __builtin_amdgcn_update_dpp(0, iv, 0xFF, 0xF, 0xF, false); break;
        }
        return __int_as_float(r);
    };
    if (helper && n > 0) {
        stage(0, 0);
        agc_tile(0, 0);
    }
    __syncthreads();
    long t0 = 0;
    int it = 0, tail = 0; // tail = samples of a final, partial chunk (0 = the call ended on a chunk boundary)
    for (; t0 < n; t0 += TS, it++) {
        const int buf = it % 3;
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        if (helper) {
            if (it > 0) {
                drain(t0 - TS, (it + 2) % 3);
            }
            if (t0 + TS < n) {
                stage(t0 + TS, (it + 1) % 3);
                agc_tile(t0 + TS, (it + 1) % 3);
            }
        } else if (live) {
            f2* row = &tiles[buf][cl][0];
            int c0 = 0;
            for (; c0 + NT <= tn; c0 += NT) { // whole chunks: straight-line code, no test per sample
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const f2 x = row[c0 + j]; // (gain-controlled by the helper wave)
                    float ns, nc;
                    sincos_two_pi(phase, &ns, &nc);
                    const float orr = x.x * nc - x.y * ns;
                    const float oi = x.x * ns + x.y * nc;
                    z[j].x = orr;
                    z[j].y = oi;
                    const f2 p0 = z[j] * tt[0];
                    float acc = 0.0f;
                    acc += p0.x + p0.y;
#pragma unroll
                    for (int k = 1; k < NT; k++) {
                        acc += ps[k];
                        const f2 pn = z[(j + 1 - k + NT) % NT] * tt[k]; // the next sample's tap k
                        ps[k] = pn.x + pn.y;
                        // (an empty statement that "touches" both: the product sum is formed before it, the chain's next add after
                        // it - left to itself the compiler runs the 21 dependent adds back to back and forms the sums afterwards)
                        asm volatile("" : "+v"(acc), "+v"(ps[k]));
                    }
                    const float lr = bcast(acc, 0), li = bcast(acc, 1), ur = bcast(acc, 2), ui = bcast(acc, 3);
                    const float lm = lr * lr + li * li, um = ur * ur + ui * ui;
                    const float err = clipf(um - lm, 1.0f);
                    freq += beta * err;
                    freq = clampr(freq, -1.0f, 1.0f);
                    phase += freq + alpha * err;
                    phase = phase > kTwoPi ? phase - kTwoPi : phase;
                    phase = phase < -kTwoPi ? phase + kTwoPi : phase;
                    if (q == 0) {
                        const f2 y = {orr, oi};
                        row[c0 + j] = y;
                    }
                }
            }
            for (; c0 < tn; c0 += NT) { // the call's last, partial chunk (ps[] is not kept up: nothing follows)
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    if (c0 + j < tn) {
                        const f2 x = row[c0 + j];
                        float ns, nc;
                        sincos_two_pi(phase, &ns, &nc);
                        const float orr = x.x * nc - x.y * ns;
                        const float oi = x.x * ns + x.y * nc;
                        z[j].x = orr;
                        z[j].y = oi;
                        float acc = 0.0f;
#pragma unroll
                        for (int k = 0; k < NT; k++) {
                            const int slot = (j - k + NT) % NT;
                            const f2 p = z[slot] * tt[k];
                            float ps; // (as one instruction: left alone, the vectoriser pairs two taps' sums into a packed add and
                                      // spends three moves on lining their halves up)
                            asm("v_add_f32 %0, %1, %2" : "=v"(ps) : "v"(p.x), "v"(p.y));
                            acc += ps;
                        }
                        const float lr = bcast(acc, 0), li = bcast(acc, 1), ur = bcast(acc, 2), ui = bcast(acc, 3);
                        const float lm = lr * lr + li * li, um = ur * ur + ui * ui;
                        const float err = clipf(um - lm, 1.0f);
                        freq += beta * err;
                        freq = clampr(freq, -1.0f, 1.0f);
                        phase += freq + alpha * err;
                        phase = phase > kTwoPi ? phase - kTwoPi : phase;
                        phase = phase < -kTwoPi ? phase + kTwoPi : phase;
                        if (q == 0) {
                            const f2 y = {orr, oi};
                            row[c0 + j] = y;
                        }
                        tail = (j + 1) % NT;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (helper && it > 0) {
        drain(t0 - TS, (it + 2) % 3);
    }
    if (agc_lane) {
        avg_end[lane] = avg;
    }
    __syncthreads();
    if (live && q == 0) {
        s.agc_avg = avg_end[cl];
        s.fll_phase = phase;
        s.fll_freq = freq;
        s.fll_idx = 0;
        state[ch] = s;
        // oldest first: after a partial chunk of `tail` samples the newest sits in slot tail - 1, the oldest in slot tail
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int dst = (k - tail + NT) % NT;
            delay_store[((size_t)dst * 2) * n_channels + ch] = z[k].x;
            delay_store[((size_t)dst * 2 + 1) * n_channels + ch] = z[k].y;
        }
    }
}

// ---- differential phasor + Costas + phase extractor (symbol rate) ---------------------------------------------------
__global__ __launch_bounds__(64) void
k_cqpsk_symbols(const f2* __restrict__ sym, size_t stride, const int* __restrict__ counts, int n_channels,
                DdnCqpskState* __restrict__ state, float* __restrict__ out, size_t out_stride) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    DdnCqpskState s = state[ch];
    if (!s.cos_init) {
        const float loop_bw = 0.008f, damping = 0.70710678118654752440f;
        const float denom = 1.0f + 2.0f * damping * loop_bw + loop_bw * loop_bw;
        s.cos_alpha = (4.0f * damping * loop_bw) / denom;
        s.cos_beta = (4.0f * loop_bw * loop_bw) / denom;
        s.cos_init = 1;
    }
    int cnt = counts[ch];
    cnt = cnt > (int)stride ? (int)stride : cnt;
    cnt = cnt > (int)out_stride ? (int)out_stride : cnt;
    if (cnt > 0) {
        const float max_phase = kPi / 2.0f, min_phase = -max_phase;
        float pr = s.diff_r, pj = s.diff_j;
        float phase = finitef(s.cos_phase) ? clampr(s.cos_phase, min_phase, max_phase) : 0.0f;
        float freq = s.cos_freq;
        float es = finitef(s.cos_es) ? s.cos_es : 0.0f;
        float last_error = 0.0f;
        const f2* ip = sym + (size_t)ch * stride;
        float* op = out + (size_t)ch * out_stride;
        const float k4pi = 4.0f / 3.14159265358979323846f;
        // The loop is a per-symbol recurrence and each lane reads its own row: eight symbols are fetched ahead of the
        // dependent chain (independent loads, one wait) instead of one load-and-wait per symbol.
        for (int n0 = 0; n0 < cnt; n0 += 8) {
          f2 pre[8];
#pragma unroll
          for (int k = 0; k < 8; k++) {
              const f2 zz = {0.0f, 0.0f};
              pre[k] = (n0 + k < cnt) ? ip[n0 + k] : zz;
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int n = n0 + k;
            if (n >= cnt) {
                break;
            }
            const f2 cur = pre[k];
            // y = x * conj(prev)
            const float ir = cur.x * pr + cur.y * pj;
            const float ij = cur.y * pr - cur.x * pj;
            pr = cur.x;
            pj = cur.y;
            float nj, nr;
            sincos_half_pi(-phase, &nj, &nr);
            const float rr = ir * nr - ij * nj;
            const float rj = ir * nj + ij * nr;
            // detector normalisation + confidence
            float dr, dj, conf;
            {
                const float mag2 = rr * rr + rj * rj;
                if (!finitef(mag2)) {
                    dr = 0.0f;
                    dj = 0.0f;
                    conf = 0.0f;
                } else if (mag2 <= 0.10f * 0.10f) {
                    dr = rr;
                    dj = rj;
                    conf = 0.0f;
                } else {
                    const float mag = sqrtf(mag2);
                    conf = (mag2 >= 0.35f * 0.35f) ? 1.0f : (finitef(mag) ? smoothstep(0.10f, 0.35f, mag) : 0.0f);
                    const float scale = (0.85f * 0.85f) / mag;
                    if (!finitef(scale)) {
                        dr = 0.0f;
                        dj = 0.0f;
                        conf = 0.0f;
                    } else {
                        dr = rr * scale;
                        dj = rj * scale;
                    }
                }
            }
            float error = 0.0f;
            if (conf <= 0.0f || !finitef(conf)) {
                es = 0.0f;
            } else {
                const float pd = ((dr > 0.0f ? 1.0f : -1.0f) * dj - (dj > 0.0f ? 1.0f : -1.0f) * dr);
                const float raw = clipf(pd * conf, 1.0f);
                float a;
                if (!finitef(raw) || !finitef(es) || fabsf(es) <= 1.0e-6f) {
                    a = 0.25f;
                } else {
                    const float kick = smoothstep(0.02f, 0.18f, fabsf(raw - es));
                    a = 0.25f + (0.10f - 0.25f) * kick;
                }
                es += a * (raw - es);
                error = clipf(es, 1.0f);
            }
            last_error = error;
            freq += s.cos_beta * error;
            phase += freq + s.cos_alpha * error;
            phase = clampr(phase, min_phase, max_phase);
            freq = clampr(freq, -1.0f, 1.0f);
            op[n] = atan2_qpsk(dj, dr) * k4pi;
          }
        }
        s.diff_r = pr;
        s.diff_j = pj;
        s.cos_phase = phase;
        s.cos_freq = freq;
        s.cos_err = last_error;
        s.cos_es = es;
    }
    state[ch] = s;
}

// ---- launchers -------------------------------------------------------------------------------------------------------
extern "C" hipError_t
ddn_dev_channel_lpf_c2c(const void* in, int in_fmt, long n, size_t in_stride, int block_len, int n_channels,
                        const float* taps_dev, int taps_len, int has_zero_tap, void* hist, void* out, size_t out_stride,
                        hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    const int tiles_per_block = (block_len + 255) / 256;
    const long n_blocks = (n + block_len - 1) / block_len;
    const dim3 grid((unsigned)(n_blocks * tiles_per_block), (unsigned)n_channels), blk(256);
#define DDN_C2C_U(FMTV, CENV)                                                                                          \
    do {                                                                                                               \
        if (has_zero_tap) {                                                                                            \
            hipLaunchKernelGGL((k_channel_lpf_c2c_u<FMTV, CENV, true>), grid, dim3(128), 0, st, in, n, in_stride,      \
                               block_len, tiles_per_block, taps_dev, (const f2*)hist, (f2*)out, out_stride);           \
        } else {                                                                                                       \
            hipLaunchKernelGGL((k_channel_lpf_c2c_u<FMTV, CENV, false>), grid, dim3(128), 0, st, in, n, in_stride,     \
                               block_len, tiles_per_block, taps_dev, (const f2*)hist, (f2*)out, out_stride);           \
        }                                                                                                              \
    } while (0)
    if (taps_len == 135 && in_fmt == DDN_IN_CU8) {
        DDN_C2C_U(DDN_IN_CU8, 67);
    } else if (taps_len == 135) {
        DDN_C2C_U(DDN_IN_CF32, 67);
    } else if (taps_len == 67 && in_fmt == DDN_IN_CU8) {
        DDN_C2C_U(DDN_IN_CU8, 33);
    } else if (taps_len == 67) {
        DDN_C2C_U(DDN_IN_CF32, 33);
    } else if (in_fmt == DDN_IN_CU8) {
        hipLaunchKernelGGL((k_channel_lpf_c2c<DDN_IN_CU8>), grid, blk, 0, st, in, n, in_stride, block_len, tiles_per_block,
                           taps_dev, taps_len, (const f2*)hist, (f2*)out, out_stride);
    } else {
        hipLaunchKernelGGL((k_channel_lpf_c2c<DDN_IN_CF32>), grid, blk, 0, st, in, n, in_stride, block_len, tiles_per_block,
                           taps_dev, taps_len, (const f2*)hist, (f2*)out, out_stride);
    }
#undef DDN_C2C_U
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        return e;
    }
    if (in_fmt == DDN_IN_CU8) {
        hipLaunchKernelGGL((k_lpf_hist<DDN_IN_CU8>), dim3((unsigned)n_channels), dim3(192), 0, st, in, n, in_stride,
                           taps_len - 1, (f2*)hist);
    } else {
        hipLaunchKernelGGL((k_lpf_hist<DDN_IN_CF32>), dim3((unsigned)n_channels), dim3(192), 0, st, in, n, in_stride,
                           taps_len - 1, (f2*)hist);
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_cqpsk_agc_fll(const void* in, long n, size_t stride, int n_channels, int nt, float alpha, float beta,
                      const float* d_fll_taps, DdnCqpskState* state, float* delay_store, void* out, hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    const size_t shm = sizeof(f2) * 3 * 16 * 33 + sizeof(float) * 2 * (size_t)(2 * nt) * 16;
    const dim3 grid((unsigned)((n_channels + 15) / 16)), blk(128);
#define DDN_LAUNCH_FLL(NTT)                                                                                            \
    do {                                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cqpsk_agc_fll<NTT>),                       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                      \
        if (e != hipSuccess) {                                                                                         \
            return e;                                                                                                  \
        }                                                                                                              \
        hipLaunchKernelGGL(k_cqpsk_agc_fll<NTT>, grid, blk, shm, st, (const f2*)in, n, stride, n_channels, nt, alpha,  \
                           beta, d_fll_taps, state, delay_store, (f2*)out);                                                        \
    } while (0)
    const dim3 rgrid((unsigned)((n_channels + 15) / 16));
    if (nt == 11) {
        hipLaunchKernelGGL(k_cqpsk_agc_fll_reg<11>, rgrid, blk, 0, st, (const f2*)in, n, stride, n_channels, alpha, beta,
                           d_fll_taps, state, delay_store, (f2*)out); // sps 5
    } else if (nt == 21) {
        hipLaunchKernelGGL(k_cqpsk_agc_fll_reg<21>, rgrid, blk, 0, st, (const f2*)in, n, stride, n_channels, alpha, beta,
                           d_fll_taps, state, delay_store, (f2*)out); // sps 10
    } else if (nt == 17) { // (eight samples per symbol: the Phase 2 chain at 48 kHz)
        hipLaunchKernelGGL(k_cqpsk_agc_fll_reg<17>, rgrid, blk, 0, st, (const f2*)in, n, stride, n_channels, alpha, beta,
                           d_fll_taps, state, delay_store, (f2*)out);
    } else if (nt == 9) {
        hipLaunchKernelGGL(k_cqpsk_agc_fll_reg<9>, rgrid, blk, 0, st, (const f2*)in, n, stride, n_channels, alpha, beta,
                           d_fll_taps, state, delay_store, (f2*)out); // sps 4 (P25p2 6000 sym/s at 24 ksps)
    } else {
        DDN_LAUNCH_FLL(0);
    }
#undef DDN_LAUNCH_FLL
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_cqpsk_symbols(const void* sym, size_t stride, const int* counts, int n_channels, DdnCqpskState* state, float* out,
                      size_t out_stride, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_cqpsk_symbols, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, (const f2*)sym, stride,
                       counts, n_channels, state, out, out_stride);
    return hipGetLastError();
}
