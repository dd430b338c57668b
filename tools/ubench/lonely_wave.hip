// lonely_wave.hip - why does a dependent chain run 2-3 x slower when its wave is alone on a SIMD?  One workgroup per CU (grid 256):
// 4 chain waves (the lean-like step of simd_share.hip, 2 active lanes) + 4 companion waves that (0) do not exist, (1) s_sleep in a loop,
// (2) spin on s_nop, (3) spin on a VALU op, (4) run the same chain (= 8 chain waves).  Per chain wave: cycles per step (s_memtime) and ns
// per step (s_memrealtime), min / median / max over the 1024 chain waves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define N_ITER 4000
__global__ void k(float* out, const float* in, long long* t_out, int companion, volatile int* stop) {
    __shared__ float lds[16 * 64 * 4];
    __shared__ int done;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* my = lds + w * 256;
    my[lane] = in[lane];
    my[lane + 64] = in[lane + 64];
    my[lane + 128] = in[lane + 1];
    my[lane + 192] = in[lane + 2];
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (w >= 4 && companion != 4) { // companion waves
        float v = in[lane];
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
            if (companion == 1) {
                __builtin_amdgcn_s_sleep(8);
            } else if (companion == 2) {
                for (int i = 0; i < 64; i++) asm volatile("s_nop 7");
            } else {
                for (int i = 0; i < 64; i++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = v;
        return;
    }
    float a = in[lane], b = in[lane + 1], mn = -1.0f, mx = 3.0f, p1 = 0.5f, p2 = 0.7f;
    double sum = 0.0;
    const long long t0 = clock64(), r0 = wall_clock64();
    if (lane < 2) {
    for (int i = 0; i < N_ITER; i++) {
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
    const long long t1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + mn + (float)sum;
    if (lane == 0) {
        const int nc = companion == 4 ? 8 : 4;
        t_out[3 * (blockIdx.x * nc + w)] = t1 - t0;
        t_out[3 * (blockIdx.x * nc + w) + 1] = r1 - r0;
        t_out[3 * (blockIdx.x * nc + w) + 2] = hwid;
        __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    float *d_in, *d_out;
    long long* d_t;
    int* d_stop;
    hipMalloc(&d_in, 4096);
    hipMalloc(&d_out, 1 << 22);
    hipMalloc(&d_t, 1 << 20);
    hipMalloc(&d_stop, 4);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = 0.3f + 0.001f * i;
    hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[] = {"no companion (4 waves / CU)", "s_sleep companions", "s_nop spinning companions", "VALU spinning companions", "8 chain waves"};
    for (int pass = 0; pass < 2; pass++) {
        for (int comp = 0; comp < 5; comp++) {
            const int waves = comp == 0 ? 4 : 8;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d_out, d_in, d_t, comp, d_stop);
                hipDeviceSynchronize();
            }
            const int nc = comp == 4 ? 8 : 4;
            std::vector<long long> t(3 * 256 * nc);
            hipMemcpy(t.data(), d_t, 8 * t.size(), hipMemcpyDeviceToHost);
            std::vector<double> cyc, ns;
            int simd_cnt[4] = {0, 0, 0, 0};
            for (int i = 0; i < 256 * nc; i++) {
                cyc.push_back(t[3 * i] / (double)N_ITER);
                ns.push_back(t[3 * i + 1] * 10.0 / N_ITER);
                simd_cnt[(t[3 * i + 2] >> 4) & 3]++;
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(ns.begin(), ns.end());
            if (pass == 1)
                printf("%-30s cycles/step min %6.1f med %6.1f max %6.1f | ns/step min %6.1f med %6.1f max %6.1f | chain waves per SIMD id %d %d %d %d\n", names[comp],
                       cyc[0], cyc[cyc.size() / 2], cyc.back(), ns[0], ns[ns.size() / 2], ns.back(), simd_cnt[0], simd_cnt[1], simd_cnt[2], simd_cnt[3]);
        }
    }
    return 0;
}
