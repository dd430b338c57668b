// ddn_dropin.hip — single-stream drop-in symbols with the reference's names (include/ddn_hip.h, family 2).
//
// Each call stages host buffers to the device, runs one small kernel and copies back: correct and
// order-exact, deliberately not fast ("used for parity tests, not throughput").
//   simd_fir_complex_apply / simd_hb_decim2_complex / simd_hb_decim2_real / simd_fir_get_impl_name
//       reference: include/dsd-neo/dsp/simd_fir.h:41-77, src/dsp/simd_fir.cpp:55-283,350-392
//   widen_u8_to_f32_bias127            reference: src/dsp/simd_widen.cpp:139-149
//   ddn_fsk_modem_discriminator_process reference: src/dsp/fsk_modem.c:135-164
// Like the reference's dispatcher (src/dsp/simd_fir.cpp:303-306,350-356) blocks shorter than 2*taps_len floats
// use the non-fused (mul, add) order, longer ones the FMA order of the AVX2 unit.

#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ddn_atan2f.h"
#include "ddn_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));

template <typename V>
__device__ __forceinline__ V
splat(float h);
template <>
__device__ __forceinline__ float
splat<float>(float h) {
    return h;
}
template <>
__device__ __forceinline__ f2
splat<f2>(float h) {
    f2 r = {h, h};
    return r;
}

// scratch = [hist (taps_len-1)] [block (n_in)] [pad: last sample repeated `center` times]
template <typename V>
__global__ void
k_sym_fir_single(const V* __restrict__ scratch, const float* __restrict__ taps, int taps_len, int n_out, int stride,
                 int tap_step, int fused, V* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) {
        return;
    }
    const int hist_len = taps_len - 1;
    const int center = hist_len >> 1;
    const int ci = hist_len + n * stride;
    V acc;
    if (fused) {
        acc = __builtin_elementwise_fma(splat<V>(taps[center]), scratch[ci], splat<V>(0.0f));
    } else {
        acc = splat<V>(0.0f) + splat<V>(taps[center]) * scratch[ci];
    }
    for (int k = 0; k < center; k += tap_step) {
        const float h = taps[k];
        if (h == 0.0f) {
            continue;
        }
        const int d = center - k;
        const V s = scratch[ci - d] + scratch[ci + d];
        if (fused) {
            acc = __builtin_elementwise_fma(splat<V>(h), s, acc);
        } else {
            acc = acc + splat<V>(h) * s;
        }
    }
    out[n] = acc;
}

__global__ void
k_widen_u8(const unsigned char* __restrict__ src, float* __restrict__ dst, uint32_t len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) {
        dst[i] = ((float)src[i] - 127.5f) * (1.0f / 127.5f);
    }
}

// rotate-by-j^n + widen (+ raw byte moments): reference src/dsp/simd_widen.cpp:167-199 (scalar), :24-43 (rotation table).
// One thread per I/Q pair (rot) or per byte (plain).  The moments are integer sums, so any reduction order is exact:
// wave-level shuffles, then one 64-bit atomic per wave.  mom = {sum, sum_sq, clipped, min, max} as five u64 words.
__device__ inline void
moments_reduce(unsigned long long sum, unsigned long long sq, unsigned int clip, unsigned int mn, unsigned int mx,
               unsigned long long* mom) {
    for (int o = 32; o > 0; o >>= 1) {
        sum += __shfl_xor(sum, o);
        sq += __shfl_xor(sq, o);
        clip += __shfl_xor(clip, o);
        mn = min(mn, (unsigned int)__shfl_xor(mn, o));
        mx = max(mx, (unsigned int)__shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&mom[0], sum);
        atomicAdd(&mom[1], sq);
        atomicAdd(&mom[2], (unsigned long long)clip);
        atomicMin(&mom[3], (unsigned long long)mn);
        atomicMax(&mom[4], (unsigned long long)mx);
    }
}

__global__ void
k_widen_u8_moments(const unsigned char* __restrict__ src, float* __restrict__ dst, uint32_t len,
                   unsigned long long* mom) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int b = 0, mn = 255, mx = 0, clip = 0;
    unsigned long long sum = 0, sq = 0;
    if (i < len) {
        b = src[i];
        dst[i] = ((float)b - 127.5f) * (1.0f / 127.5f);
        sum = b;
        sq = (unsigned long long)b * b;
        clip = (b <= 1u || b >= 254u) ? 1u : 0u;
        mn = mx = b;
    }
    moments_reduce(sum, sq, clip, mn, mx, mom);
}

__global__ void
k_widen_rot_u8(const unsigned char* __restrict__ src, float* __restrict__ dst, uint32_t pairs, uint32_t phase,
               unsigned long long* mom) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int mn = 255, mx = 0, clip = 0;
    unsigned long long sum = 0, sq = 0;
    if (n < pairs) {
        const unsigned int bi = src[2 * n], bq = src[2 * n + 1];
        const float i_raw = ((float)bi - 127.5f) * (1.0f / 127.5f);
        const float q_raw = ((float)bq - 127.5f) * (1.0f / 127.5f);
        float oi, oq;
        switch ((phase + n) & 3u) {
            case 0: oi = i_raw; oq = q_raw; break;
            case 1: oi = -q_raw; oq = i_raw; break;
            case 2: oi = -i_raw; oq = -q_raw; break;
            default: oi = q_raw; oq = -i_raw; break;
        }
        dst[2 * n] = oi;
        dst[2 * n + 1] = oq;
        sum = bi + bq;
        sq = (unsigned long long)bi * bi + (unsigned long long)bq * bq;
        clip = ((bi <= 1u || bi >= 254u) ? 1u : 0u) + ((bq <= 1u || bq >= 254u) ? 1u : 0u);
        mn = min(bi, bq);
        mx = max(bi, bq);
    }
    if (mom) {
        moments_reduce(sum, sq, clip, mn, mx, mom);
    }
}

__global__ void
k_fsk_single(ddn_fsk_modem_state* st, const f2* __restrict__ iq, int pairs, float* __restrict__ out, int max_out,
             int* out_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) {
        return;
    }
    float prev_i = st->prev_i, prev_q = st->prev_q, dc = st->dc_est, peak = st->discriminator_peak_est;
    int have_prev = st->have_prev;
    int w = 0;
    for (int n = 0; n < pairs && w < max_out; n++) {
        const f2 cur = iq[n];
        if (!have_prev) {
            prev_i = cur.x;
            prev_q = cur.y;
            have_prev = 1;
            out[w++] = 0.0f;
            continue;
        }
        const float re = cur.x * prev_i + cur.y * prev_q;
        const float im = cur.y * prev_i - cur.x * prev_q;
        float fr;
        if (re > 1.0e-7f && fabsf(im) <= (0.35f * re)) {
            const float x = im / re;
            const float x2 = x * x;
            fr = x * (1.0f + x2 * (-0.3333333333333333f + x2 * 0.2f));
        } else {
            fr = ddn_atan2f(im, re);
        }
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
        out[w++] = y;
        prev_i = cur.x;
        prev_q = cur.y;
    }
    st->prev_i = prev_i;
    st->prev_q = prev_q;
    st->have_prev = have_prev;
    st->dc_est = dc;
    st->discriminator_peak_est = peak;
    *out_count = w;
}

namespace {

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) {
            p = nullptr;
        }
    }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

void
shift_hist(float* hist, const float* in, int n, int hist_len, int in_stride, int in_off) {
    if (n >= hist_len) {
        for (int k = 0; k < hist_len; k++) {
            hist[k] = in[(size_t)(n - hist_len + k) * in_stride + in_off];
        }
    } else {
        const int keep = hist_len - n;
        memmove(hist, hist + n, (size_t)keep * sizeof(float));
        for (int k = 0; k < n; k++) {
            hist[keep + k] = in[(size_t)k * in_stride + in_off];
        }
    }
}

// returns complex outputs written, or <0 on device failure
int
run_complex(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps, int taps_len,
            int stride) {
    const int n = in_len >> 1;
    const int hist_len = taps_len - 1;
    const int center = hist_len >> 1;
    const int n_out = (stride == 2) ? (n >> 1) : n;
    const int fused = (in_len >= taps_len * 2) ? 1 : 0;
    std::vector<float> scratch((size_t)(hist_len + n + center + 2) * 2);
    for (int k = 0; k < hist_len; k++) {
        scratch[2 * (size_t)k] = hist_i[k];
        scratch[2 * (size_t)k + 1] = hist_q[k];
    }
    memcpy(scratch.data() + 2 * (size_t)hist_len, in, (size_t)n * 2 * sizeof(float));
    for (int k = 0; k < center + 2; k++) {
        scratch[2 * (size_t)(hist_len + n + k)] = in[2 * (n - 1)];
        scratch[2 * (size_t)(hist_len + n + k) + 1] = in[2 * (n - 1) + 1];
    }
    if (n_out > 0) {
        DevBuf ds(scratch.size() * sizeof(float)), dt((size_t)taps_len * sizeof(float)),
            dout((size_t)n_out * 2 * sizeof(float));
        if (!ds.p || !dt.p || !dout.p) {
            ddn_set_error("drop-in FIR: hipMalloc failed (no device?)");
            return DDN_ENODEV;
        }
        if (hipMemcpy(ds.p, scratch.data(), scratch.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess
            || hipMemcpy(dt.p, taps, (size_t)taps_len * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            ddn_set_error("drop-in FIR: H2D failed");
            return DDN_EHIP;
        }
        hipLaunchKernelGGL((k_sym_fir_single<f2>), dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, 0,
                           (const f2*)ds.p, (const float*)dt.p, taps_len, n_out, stride, stride, fused, (f2*)dout.p);
        if (hipGetLastError() != hipSuccess
            || hipMemcpy(out, dout.p, (size_t)n_out * 2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            ddn_set_error("drop-in FIR: kernel/D2H failed");
            return DDN_EHIP;
        }
    }
    shift_hist(hist_i, in, n, hist_len, 2, 0);
    shift_hist(hist_q, in, n, hist_len, 2, 1);
    return n_out;
}

} // namespace

extern "C" void
simd_fir_complex_apply(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                       int taps_len) {
    if (taps_len < 3 || (taps_len & 1) == 0 || in_len < 2 || !in || !out || !hist_i || !hist_q || !taps) {
        return; // the reference is a silent no-op on bad arguments (src/dsp/simd_fir.cpp:58-60)
    }
    (void)run_complex(in, in_len, out, hist_i, hist_q, taps, taps_len, 1);
}

extern "C" int
simd_hb_decim2_complex(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                       int taps_len) {
    if (taps_len < 3 || (taps_len & 1) == 0 || (in_len >> 1) <= 0 || !in || !out || !hist_i || !hist_q || !taps) {
        return 0;
    }
    const int n_out = run_complex(in, in_len, out, hist_i, hist_q, taps, taps_len, 2);
    return n_out < 0 ? 0 : (n_out << 1);
}

extern "C" int
simd_hb_decim2_real(const float* in, int in_len, float* out, float* hist, const float* taps, int taps_len) {
    if (taps_len < 3 || (taps_len & 1) == 0 || in_len <= 0 || !in || !out || !hist || !taps) {
        return 0;
    }
    const int hist_len = taps_len - 1;
    const int center = hist_len >> 1;
    const int n_out = in_len >> 1;
    const int fused = (in_len >= taps_len * 2) ? 1 : 0;
    std::vector<float> scratch((size_t)hist_len + in_len + center + 2);
    memcpy(scratch.data(), hist, (size_t)hist_len * sizeof(float));
    memcpy(scratch.data() + hist_len, in, (size_t)in_len * sizeof(float));
    for (int k = 0; k < center + 2; k++) {
        scratch[(size_t)hist_len + in_len + k] = in[in_len - 1];
    }
    if (n_out > 0) {
        DevBuf ds(scratch.size() * sizeof(float)), dt((size_t)taps_len * sizeof(float)),
            dout((size_t)n_out * sizeof(float));
        if (!ds.p || !dt.p || !dout.p) {
            ddn_set_error("drop-in HB: hipMalloc failed (no device?)");
            return 0;
        }
        (void)hipMemcpy(ds.p, scratch.data(), scratch.size() * sizeof(float), hipMemcpyHostToDevice);
        (void)hipMemcpy(dt.p, taps, (size_t)taps_len * sizeof(float), hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k_sym_fir_single<float>), dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, 0,
                           (const float*)ds.p, (const float*)dt.p, taps_len, n_out, 2, 2, fused, (float*)dout.p);
        if (hipGetLastError() != hipSuccess
            || hipMemcpy(out, dout.p, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            ddn_set_error("drop-in HB: kernel/D2H failed");
            return 0;
        }
    }
    shift_hist(hist, in, in_len, hist_len, 1, 0);
    return n_out;
}

extern "C" const char*
simd_fir_get_impl_name(void) {
    return "hip-gfx950";
}

extern "C" void
widen_u8_to_f32_bias127(const unsigned char* src, float* dst, uint32_t len) {
    if (!src || !dst || len == 0U) {
        return;
    }
    DevBuf ds(len), dd((size_t)len * sizeof(float));
    if (!ds.p || !dd.p) {
        ddn_set_error("drop-in widen: hipMalloc failed (no device?)");
        return;
    }
    (void)hipMemcpy(ds.p, src, len, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_widen_u8, dim3((len + 255) / 256), dim3(256), 0, 0, (const unsigned char*)ds.p, (float*)dd.p,
                       len);
    (void)hipMemcpy(dst, dd.p, (size_t)len * sizeof(float), hipMemcpyDeviceToHost);
}

// dsd_input_level_cu8_moments_merge (reference src/runtime/input_level.c:231-272): validity of both sides, overflow guards,
// "empty accumulator takes the addend".
static int
moments_valid(const dsd_input_level_cu8_moments* m) {
    if (!m || m->count == 0U || m->clipped > m->count || m->min_sample > m->max_sample) {
        return 0;
    }
    if ((m->count <= UINT64_MAX / 255U && m->sum > m->count * 255U)
        || (m->count <= UINT64_MAX / 65025U && m->sum_sq > m->count * 65025U)) {
        return 0;
    }
    return 1;
}

static void
moments_merge(dsd_input_level_cu8_moments* m, const dsd_input_level_cu8_moments* add) {
    if (!m || !moments_valid(add)) {
        return;
    }
    dsd_input_level_cu8_moments next = *m;
    if (next.count == 0U) {
        next = *add;
    } else {
        if (!moments_valid(&next) || UINT64_MAX - next.count < add->count || UINT64_MAX - next.sum < add->sum
            || UINT64_MAX - next.sum_sq < add->sum_sq || UINT64_MAX - next.clipped < add->clipped) {
            return;
        }
        next.count += add->count;
        next.sum += add->sum;
        next.sum_sq += add->sum_sq;
        next.clipped += add->clipped;
        next.min_sample = add->min_sample < next.min_sample ? add->min_sample : next.min_sample;
        next.max_sample = add->max_sample > next.max_sample ? add->max_sample : next.max_sample;
    }
    *m = next;
}

// shared body of the three moment / rotate variants.  rot: pairs are rotated by j^(phase+n); count = bytes consumed.
static int
widen_variant(const unsigned char* src, float* dst, uint32_t len, int rot, uint32_t phase,
              dsd_input_level_cu8_moments* moments) {
    const uint32_t used = rot ? (len & ~1u) : len;
    DevBuf ds(used), dd((size_t)used * sizeof(float)), dm(5 * sizeof(unsigned long long));
    if (!ds.p || !dd.p || !dm.p) {
        ddn_set_error("drop-in widen: hipMalloc failed (no device?)");
        return -1;
    }
    const unsigned long long init[5] = {0, 0, 0, 255, 0};
    (void)hipMemcpy(ds.p, src, used, hipMemcpyHostToDevice);
    (void)hipMemcpy(dm.p, init, sizeof(init), hipMemcpyHostToDevice);
    if (rot) {
        const uint32_t pairs = used >> 1;
        hipLaunchKernelGGL(k_widen_rot_u8, dim3((pairs + 255) / 256), dim3(256), 0, 0, (const unsigned char*)ds.p,
                           (float*)dd.p, pairs, phase, moments ? (unsigned long long*)dm.p : nullptr);
    } else {
        hipLaunchKernelGGL(k_widen_u8_moments, dim3((used + 255) / 256), dim3(256), 0, 0, (const unsigned char*)ds.p,
                           (float*)dd.p, used, (unsigned long long*)dm.p);
    }
    unsigned long long got[5];
    if (hipGetLastError() != hipSuccess
        || hipMemcpy(dst, dd.p, (size_t)used * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(got, dm.p, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess) {
        ddn_set_error("drop-in widen: kernel/D2H failed");
        return -1;
    }
    if (moments) {
        dsd_input_level_cu8_moments local;
        local.count = used;
        local.sum = got[0];
        local.sum_sq = got[1];
        local.clipped = got[2];
        local.min_sample = (uint8_t)got[3];
        local.max_sample = (uint8_t)got[4];
        moments_merge(moments, &local);
    }
    return 0;
}

extern "C" void
widen_u8_to_f32_bias127_moments(const unsigned char* src, float* dst, uint32_t len,
                                dsd_input_level_cu8_moments* moments) {
    if (!src || !dst || !moments || len == 0U) {
        return;
    }
    (void)widen_variant(src, dst, len, 0, 0, moments);
}

extern "C" uint32_t
widen_rotate90_u8_to_f32_bias127_phase_moments(const unsigned char* src, float* dst, uint32_t len, uint32_t phase,
                                               dsd_input_level_cu8_moments* moments) {
    const uint32_t cur = phase & 3U;
    if (!src || !dst || len < 2U) {
        return cur;
    }
    if (widen_variant(src, dst, len, 1, cur, moments) != 0) {
        return cur;
    }
    return (cur + (len >> 1)) & 3U;
}

extern "C" uint32_t
widen_rotate90_u8_to_f32_bias127_phase(const unsigned char* src, float* dst, uint32_t len, uint32_t phase) {
    return widen_rotate90_u8_to_f32_bias127_phase_moments(src, dst, len, phase, nullptr);
}

extern "C" int
ddn_fsk_modem_discriminator_process(ddn_fsk_modem_state* st, const float* iq_interleaved, int len_interleaved,
                                    float* out_samples, int max_samples) {
    if (!st || !out_samples || max_samples <= 0 || !iq_interleaved || len_interleaved < 2) {
        return 0;
    }
    const int pairs = len_interleaved >> 1;
    const int cap = pairs < max_samples ? pairs : max_samples;
    DevBuf dst(sizeof(*st)), diq((size_t)pairs * 2 * sizeof(float)), dout((size_t)cap * sizeof(float)),
        dcnt(sizeof(int));
    if (!dst.p || !diq.p || !dout.p || !dcnt.p) {
        ddn_set_error("drop-in discriminator: hipMalloc failed (no device?)");
        return 0;
    }
    (void)hipMemcpy(dst.p, st, sizeof(*st), hipMemcpyHostToDevice);
    (void)hipMemcpy(diq.p, iq_interleaved, (size_t)pairs * 2 * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fsk_single, dim3(1), dim3(64), 0, 0, (ddn_fsk_modem_state*)dst.p, (const f2*)diq.p, pairs,
                       (float*)dout.p, max_samples, (int*)dcnt.p);
    int cnt = 0;
    if (hipGetLastError() != hipSuccess || hipMemcpy(&cnt, dcnt.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
        ddn_set_error("drop-in discriminator: kernel failed");
        return 0;
    }
    (void)hipMemcpy(out_samples, dout.p, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipMemcpy(st, dst.p, sizeof(*st), hipMemcpyDeviceToHost);
    return cnt;
}

// the reference's own name for the same entry point (include/dsd-neo/dsp/fsk_modem.h:42; dsd_fsk_modem_state has the
// layout of ddn_fsk_modem_state), so tests/dsp/test_fsk_modem.c links against this library unchanged
extern "C" int
dsd_fsk_modem_discriminator_process(ddn_fsk_modem_state* st, const float* iq_interleaved, int len_interleaved,
                                    float* out_samples, int max_samples) {
    return ddn_fsk_modem_discriminator_process(st, iq_interleaved, len_interleaved, out_samples, max_samples);
}
