// ddn_slicer_par.hip — the stand-alone P25 slicer (ddn_p25_slicer_run) restructured for the machine: when the symbol
// stream is given (CQPSK symbols, recorded symbol captures) the thresholds never feed back into the symbols, so only one
// step of use_symbol() is a true recurrence.
//
// reference per symbol (src/core/frames/dsd_dibit.c:194-299): window[sidx] = symbol; {lo, hi} = mean of the two smallest /
// two largest of the 128-symbol window; 1024-deep moving averages of lo / hi kept as binary64 running sums
// (include/dsd-neo/core/state.h:1430-1455); centre / mid thresholds; slice + soft decision.
//   k_slicer_extrema   sliding-window extrema are a function of the last 128 symbols only: one thread per symbol scans
//                      its window in LDS (carried window in front of the new symbols).  Fully parallel.
//   k_slicer_sums      sum += (double)lo_k - (double)old_k in symbol order is the one sequential chain (binary64 adds do
//                      not re-associate): lane = channel, one add per symbol and ring, everything it reads and writes laid
//                      out [symbol][channel] so a wavefront's accesses are contiguous.  Emits min_k / max_k per symbol.
//   k_slicer_slice     thresholds from min_k / max_k, slice + soft decision, 10-byte records: one thread per symbol;
//                      [symbol][channel] tiles are transposed through LDS so loads and record stores stay coalesced.
//   k_slicer_window    carried 128-symbol window and indices for the next call.
// Same statistics as the sequential kernel for finite symbols (multisets: the two smallest of a window do not depend on
// scan order).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_slicer_dev.h"

using ddn_sl::two_max_insert;
using ddn_sl::two_min_insert;

namespace {
constexpr int SS = 128, MS = 1024;
constexpr int XC = 16, XK = 64; // extrema tile: 16 channels x 64 symbols per workgroup (1024 threads)

__global__ __launch_bounds__(XC* XK) void
k_slicer_extrema(const float* __restrict__ sym, long n, size_t stride, int n_channels,
                 const DdnSlicerState* __restrict__ state, const float* __restrict__ sbuf, float* __restrict__ lo,
                 float* __restrict__ hi) {
    // symbols k0 - 127 .. k0 + 63 of each channel (index j + 127); row stride 197 = 5 mod 64 keeps the 16 channels x 4
    // symbols of a wavefront on distinct banks (a stride of 192 would put all 16 channels on one)
    __shared__ float S[XC][197];
    const int c = threadIdx.x % XC, kk = threadIdx.x / XC;
    const int ch0 = blockIdx.y * XC;
    const long k0 = (long)blockIdx.x * XK;
    // stage: rows of the symbol array are contiguous in k, so let consecutive threads walk k
    for (int i = threadIdx.x; i < XC * (XK + SS - 1); i += XC * XK) {
        const int cc = i / (XK + SS - 1), j = i % (XK + SS - 1);
        const long k = k0 - (SS - 1) + j; // symbol index, negative = carried window
        const int ch = ch0 + cc;
        float v = 0.0f;
        if (ch < n_channels) {
            if (k >= 0) {
                v = k < n ? sym[(size_t)ch * stride + k] : 0.0f;
            } else {
                const int sidx = state[ch].sidx;
                v = sbuf[(size_t)((sidx + SS + (int)k) & (SS - 1)) * n_channels + ch]; // age order: newest = sidx - 1
            }
        }
        S[cc][j] = v;
    }
    __syncthreads();
    const long k = k0 + kk;
    const int ch = ch0 + c;
    if (ch >= n_channels || k >= n) {
        return;
    }
    const float* w = &S[c][kk]; // window = w[0..127], newest last
    float a1 = w[0], a2 = w[1];
    float b1 = a1, b2 = a2;
    if (a2 < a1) {
        const float t = a1;
        a1 = a2;
        a2 = t;
    }
    if (b2 > b1) {
        const float t = b1;
        b1 = b2;
        b2 = t;
    }
#pragma unroll 8
    for (int i = 2; i < SS; i++) {
        const float v = w[i];
        two_min_insert(v, a1, a2);
        two_max_insert(v, b1, b2);
    }
    lo[(size_t)k * n_channels + ch] = (a1 + a2) * 0.5f;
    hi[(size_t)k * n_channels + ch] = (b1 + b2) * 0.5f;
}

__global__ __launch_bounds__(64) void
k_slicer_sums(long n, int n_channels, DdnSlicerState* __restrict__ state, float* __restrict__ minring,
              float* __restrict__ maxring, const float* __restrict__ lo, const float* __restrict__ hi,
              float* __restrict__ minv, float* __restrict__ maxv) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    DdnSlicerState s = state[ch];
    if (!s.sums_valid) {
        // first push after a reset: sums are rebuilt from the rings (include/dsd-neo/core/state.h:1407-1426)
        double a = 0.0, b = 0.0;
        for (int k = 0; k < MS; k++) {
            a += (double)minring[(size_t)k * n_channels + ch];
            b += (double)maxring[(size_t)k * n_channels + ch];
        }
        s.min_sum = a;
        s.max_sum = b;
        s.sums_valid = 1;
        if (s.midx < 0 || s.midx >= MS) {
            s.midx = 0;
        }
    }
    double smin = s.min_sum, smax = s.max_sum;
    int midx = s.midx;
    float mn = s.min, mx = s.max;
    // 32 symbols per trip: all 128 loads are issued before the first dependent add, so HBM / L2 latency is paid once per
    // trip instead of once per symbol (the adds themselves stay strictly in symbol order)
    constexpr int U = 32;
    for (long k0 = 0; k0 < n; k0 += U) {
        float l[U], h[U], ol[U], oh[U];
        const bool from_ring = k0 < MS; // MS is a multiple of U: a trip never straddles the switch
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long k = k0 + u;
            const bool ok = k < n;
            const size_t o = (size_t)(ok ? k : k0) * n_channels + ch;
            int mi = midx + u;
            mi = mi >= MS ? mi - MS : mi;
            const size_t ro = (size_t)mi * n_channels + ch;
            l[u] = lo[o];
            h[u] = hi[o];
            // the value leaving the ring: its carried content for the first 1024 pushes, then this call's own extrema
            ol[u] = from_ring ? minring[ro] : lo[o - (size_t)MS * n_channels];
            oh[u] = from_ring ? maxring[ro] : hi[o - (size_t)MS * n_channels];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long k = k0 + u;
            if (k < n) {
                const size_t o = (size_t)k * n_channels + ch;
                smin += (double)l[u] - (double)ol[u];
                smax += (double)h[u] - (double)oh[u];
                if (k >= n - MS) {
                    const size_t ro = (size_t)midx * n_channels + ch;
                    minring[ro] = l[u]; // only the last 1024 pushes survive in the ring
                    maxring[ro] = h[u];
                }
                midx = (midx + 1 >= MS) ? 0 : midx + 1;
                mn = (float)(smin / (double)MS);
                mx = (float)(smax / (double)MS);
                minv[o] = mn;
                maxv[o] = mx;
            }
        }
    }
    if (n > 0) {
        s.min_sum = smin;
        s.max_sum = smax;
        s.midx = midx;
        s.min = mn;
        s.max = mx;
        s.center = (mx + mn) / 2.0f;
        s.umid = ((mx - s.center) * 5.0f / 8.0f) + s.center;
        s.lmid = ((mn - s.center) * 5.0f / 8.0f) + s.center;
    }
    state[ch] = s; // sidx is advanced by k_slicer_window, which still needs the old value
}

__global__ __launch_bounds__(256) void
k_slicer_slice(const float* __restrict__ sym, long n, size_t stride, int n_channels, int negative,
               const float* __restrict__ minv, const float* __restrict__ maxv, uint8_t* __restrict__ rec,
               size_t rec_stride) {
    __shared__ float tmin[64][65], tmax[64][65]; // [k][ch]
    const int ch0 = blockIdx.y * 64;
    const long k0 = (long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int kk = i / 64, cc = i % 64;
        const long k = k0 + kk;
        const int ch = ch0 + cc;
        const bool ok = k < n && ch < n_channels;
        tmin[kk][cc] = ok ? minv[(size_t)k * n_channels + ch] : 0.0f;
        tmax[kk][cc] = ok ? maxv[(size_t)k * n_channels + ch] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i / 64, kk = i % 64; // consecutive threads = consecutive symbols of one channel
        const long k = k0 + kk;
        const int ch = ch0 + cc;
        if (k >= n || ch >= n_channels) {
            continue;
        }
        const float x = sym[(size_t)ch * stride + k];
        const float mn = tmin[kk][cc], mx = tmax[kk][cc];
        const float center = (mx + mn) / 2.0f;
        const ddn_sl::Thr th = {center, ((mx - center) * 5.0f / 8.0f) + center, ((mn - center) * 5.0f / 8.0f) + center,
                                mx, mn};
        int dibit, relb, l0, l1;
        ddn_sl::slice_soft(x, th, negative, dibit, relb, l0, l1);
        uint8_t* r = rec + (size_t)ch * rec_stride + (size_t)k * 10;
        const uint32_t xb = __float_as_uint(x);
        ((uint16_t*)r)[0] = (uint16_t)((dibit & 3) | (relb << 8));
        ((uint16_t*)r)[1] = (uint16_t)(int16_t)l0;
        ((uint16_t*)r)[2] = (uint16_t)(int16_t)l1;
        ((uint16_t*)r)[3] = (uint16_t)(xb & 0xFFFFu);
        ((uint16_t*)r)[4] = (uint16_t)(xb >> 16);
    }
}

__global__ __launch_bounds__(SS) void
k_slicer_window(const float* __restrict__ sym, long n, size_t stride, int n_channels, DdnSlicerState* __restrict__ state,
                float* __restrict__ sbuf) {
    const int ch = blockIdx.x;
    const int j = threadIdx.x; // window slot
    const int sidx = state[ch].sidx;
    // the last symbol written to slot j: largest k < n with (sidx + k) % 128 == j
    const long first = (long)((j - sidx + SS) & (SS - 1));
    if (first < n) {
        const long k = first + ((n - 1 - first) / SS) * SS;
        sbuf[(size_t)j * n_channels + ch] = sym[(size_t)ch * stride + k];
    }
    __syncthreads();
    if (j == 0) {
        state[ch].sidx = (int)((sidx + n) & (SS - 1));
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_p25_slicer_par(const float* sym, long n, size_t sym_stride, int n_channels, int negative, DdnSlicerState* state,
                       float* sbuf_store, float* minring, float* maxring, float* scratch4, uint8_t* rec, size_t rec_stride,
                       hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    const size_t plane = (size_t)n * (size_t)n_channels;
    float *lo = scratch4, *hi = scratch4 + plane, *minv = scratch4 + 2 * plane, *maxv = scratch4 + 3 * plane;
    hipLaunchKernelGGL(k_slicer_extrema, dim3((unsigned)((n + XK - 1) / XK), (unsigned)((n_channels + XC - 1) / XC)),
                       dim3(XC * XK), 0, st, sym, n, sym_stride, n_channels, state, sbuf_store, lo, hi);
    hipLaunchKernelGGL(k_slicer_sums, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, n, n_channels, state,
                       minring, maxring, lo, hi, minv, maxv);
    hipLaunchKernelGGL(k_slicer_slice, dim3((unsigned)((n + 63) / 64), (unsigned)((n_channels + 63) / 64)), dim3(256), 0,
                       st, sym, n, sym_stride, n_channels, negative, minv, maxv, rec, rec_stride);
    hipLaunchKernelGGL(k_slicer_window, dim3((unsigned)n_channels), dim3(SS), 0, st, sym, n, sym_stride, n_channels, state,
                       sbuf_store);
    return hipGetLastError();
}
