#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// wave 8 (last) runs the dc chain over LDS tiles; waves 0..7 run a configurable background load
// bg: 0 idle, 1 pk_fma VALU only, 2 LDS ds_read_b64 only, 3 both
__global__ __launch_bounds__(576) void k(float* out, long long* cyc, int tiles, int bg, int prio, int serial_first) {
    __shared__ __attribute__((aligned(16))) float F[16][260];
    __shared__ f2 W[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * 260; i += 576) ((float*)F)[i] = 0.001f * (i % 97);
    for (int i = tid; i < 4096; i += 576) { f2 v = {0.001f * i, 0.002f * i}; W[i] = v; }
    __syncthreads();
    const bool serial = serial_first ? (tid < 64) : (tid >= 512);
    if (serial && prio) __builtin_amdgcn_s_setprio(3);
    float dc = 0.f; long long acc = 0;
    f2 a0 = {1.f, 2.f}, a1 = {1.f, 2.f}, a2 = {3.f, 1.f}, a3 = {0.5f, 2.f}, a4 = a0, a5 = a1, a6 = a2, a7 = a3;
    const f2 h = {0.999f, 1.001f};
    for (int it = 0; it < tiles; it++) {
        if (serial) {
            if ((tid & 63) < 16) {
                long long t0 = __builtin_readcyclecounter();
                float* Fr = &F[tid & 15][0];
                for (int t = 0; t + 8 <= 256; t += 8) {
                    f4 fa = *(const f4*)&Fr[t], fb = *(const f4*)&Fr[t + 4], ca, cb;
#pragma unroll
                    for (int k = 0; k < 4; k++) { dc += 0.00025f * (fa[k] - dc); ca[k] = fa[k] - dc; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { dc += 0.00025f * (fb[k] - dc); cb[k] = fb[k] - dc; }
                    *(f4*)&Fr[t] = ca; *(f4*)&Fr[t + 4] = cb;
                }
                acc += __builtin_readcyclecounter() - t0;
            }
        } else if (bg) {
            const int base = (tid * 9) & 2047;
            for (int k = 0; k < 540; k++) {
                if (bg & 2) {
                    f2 x = W[base + (k & 255)], y = W[base + 300 + (k & 255)];
                    a0 = __builtin_elementwise_fma(h, x + y, a0);
                }
                if (bg & 1) {
                    a1 = __builtin_elementwise_fma(h, a1 + a5, a1); a2 = __builtin_elementwise_fma(h, a2 + a6, a2);
                    a3 = __builtin_elementwise_fma(h, a3 + a7, a3); a4 = __builtin_elementwise_fma(h, a4 + a0, a4);
                    a5 = __builtin_elementwise_fma(h, a5 + a1, a5); a6 = __builtin_elementwise_fma(h, a6 + a2, a6);
                    a7 = __builtin_elementwise_fma(h, a7 + a3, a7);
                }
            }
        }
        __syncthreads();
    }
    out[blockIdx.x * 576 + tid] = dc + a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y;
    if (serial && (tid & 63) == 0) cyc[blockIdx.x] = acc;
}
int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8 * 1024);
    int tiles = 100;
    for (int sf = 0; sf < 2; sf++) for (int prio = 0; prio < 2; prio++) for (int bg = 0; bg < 4; bg++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(576), 0, 0, out, cyc, tiles, bg, prio, sf);
        (void)hipDeviceSynchronize();
        long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("serial_first=%d prio=%d bg=%d: %.1f cycles/sample\n", sf, prio, bg, (double)h[0] / tiles / 256.0);
    }
    return 0;
}
