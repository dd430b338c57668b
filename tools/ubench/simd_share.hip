// simd_share.hip - does a SIMD interleave independent dependent chains for free?  One workgroup per CU (grid 256) of 4 / 8 / 16 waves
// = 1 / 2 / 4 chain waves per SIMD, every wave running the same dependent sequence with 2 active lanes; per-wave time from s_memtime.
// Chains: (a) v_add_f32 only, (b) a lean-trip-like mix (med3 / add / fma / min / max / f64 add / cvt, ~40 dependent-ish ops with two
// LDS reads and two LDS writes per step).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define N_ITER 4000
template <int KIND>
__global__ void k(float* out, const float* in, long long* t_out) {
    __shared__ float lds[16 * 64 * 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* my = lds + w * 256;
    my[lane] = in[lane];
    my[lane + 64] = in[lane + 64];
    my[lane + 128] = in[lane + 1];
    my[lane + 192] = in[lane + 2];
    __syncthreads();
    if (lane >= 2) {
        return;
    }
    float a = in[lane], b = in[lane + 1], mn = -1.0f, mx = 3.0f, p1 = 0.5f, p2 = 0.7f;
    double sum = 0.0;
    const long long t0 = clock64();
    for (int i = 0; i < N_ITER; i++) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 40; u++) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
            }
        } else {
            float x0 = my[lane], x1 = my[lane + 64], x2 = my[lane + 128], x3 = my[lane + 192];
            float acc = 0.0f;
            acc += __builtin_amdgcn_fmed3f(x0, mn, mx);
            acc += __builtin_amdgcn_fmed3f(x1, mn, mx);
            acc += __builtin_amdgcn_fmed3f(x2, mn, mx);
            acc += __builtin_amdgcn_fmed3f(x3, mn, mx);
            acc += __builtin_amdgcn_fmed3f(a, mn, mx);
            const float q = acc * 0.2f;
            const float r = __builtin_fmaf(-5.0f, q, acc);
            const float sym = __builtin_fmaf(r, 0.2f, q);
            const float n2 = __builtin_amdgcn_fmed3f(p1, p2, sym);
            p1 = fminf(p1, sym);
            p2 = n2;
            const float lo = (p1 + p2) * 0.5f;
            sum += (double)lo - 0.25;
            mn = (float)(sum * 0.0009765625) - 1.0f;
            mx = mn + 4.0f;
            my[lane] = sym + b;
            my[lane + 64] = lo;
            a = sym;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + mn + (float)sum;
    if (lane == 0) {
        t_out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
    }
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float *d_in, *d_out;
    long long* d_t;
    hipMalloc(&d_in, 4096);
    hipMalloc(&d_out, 1 << 22);
    hipMalloc(&d_t, 1 << 20);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = 0.3f + 0.001f * i;
    hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int w = 0; w < 30; w++) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, d_out, d_in, d_t);
    hipDeviceSynchronize();
    for (int kind = 0; kind < 2; kind++) {
        for (int waves : {4, 8, 12, 16}) {
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, d_out, d_in, d_t);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, d_out, d_in, d_t);
                hipDeviceSynchronize();
            }
            std::vector<long long> t(256 * waves);
            hipMemcpy(t.data(), d_t, 8 * t.size(), hipMemcpyDeviceToHost);
            std::sort(t.begin(), t.end());
            const double per = kind == 0 ? 40.0 : 1.0;
            printf("%s, %2d waves per CU (%d per SIMD): cycles per %s min %.1f med %.1f max %.1f\n", kind == 0 ? "v_add_f32 chain" : "lean-like step",
                   waves, waves / 4, kind == 0 ? "op" : "step", t[0] / (N_ITER * per), t[t.size() / 2] / (N_ITER * per), t.back() / (N_ITER * per));
        }
    }
    return 0;
}
