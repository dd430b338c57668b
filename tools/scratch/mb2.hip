#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
// one wave runs the dc chain over LDS tiles while the other waves of the block sit at the barrier
template <int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void k_s1(float* out, long long* cyc, int tiles, int lanes, int mode) {
    __shared__ __attribute__((aligned(16))) float F[16][260];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * 260; i += NTHREADS) ((float*)F)[i] = 0.001f * (i % 97);
    __syncthreads();
    float dc = 0.f; long long acc = 0;
    for (int it = 0; it < tiles; it++) {
        if (tid >= NTHREADS - 64 && (tid & 63) < lanes) {
            long long t0 = __builtin_readcyclecounter();
            float* Fr = &F[tid & 15][0];
            if (mode == 0) {
                for (int t = 0; t + 8 <= 256; t += 8) {
                    f4 fa = *(const f4*)&Fr[t], fb = *(const f4*)&Fr[t + 4], ca, cb;
#pragma unroll
                    for (int k = 0; k < 4; k++) { dc += 0.00025f * (fa[k] - dc); ca[k] = fa[k] - dc; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { dc += 0.00025f * (fb[k] - dc); cb[k] = fb[k] - dc; }
                    *(f4*)&Fr[t] = ca; *(f4*)&Fr[t + 4] = cb;
                }
            } else {
                for (int t = 0; t < 256; t++) { float fr = Fr[t]; dc += 0.00025f * (fr - dc); Fr[t] = fr - dc; }
            }
            acc += __builtin_readcyclecounter() - t0;
        }
        __syncthreads();
    }
    out[blockIdx.x * NTHREADS + tid] = dc;
    if (tid == NTHREADS - 64) cyc[blockIdx.x] = acc;
}
int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8 * 1024);
    int tiles = 200;
    for (int mode = 0; mode < 2; mode++) for (int lanes : {64, 16}) {
        for (int nt : {64, 640}) {
            if (nt == 64) hipLaunchKernelGGL(k_s1<64>, dim3(256), dim3(64), 0, 0, out, cyc, tiles, lanes, mode);
            else hipLaunchKernelGGL(k_s1<640>, dim3(256), dim3(640), 0, 0, out, cyc, tiles, lanes, mode);
            (void)hipDeviceSynchronize();
            long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            printf("mode=%d lanes=%d threads=%d: %.1f cycles/sample\n", mode, lanes, nt, (double)h[0] / tiles / 256.0);
        }
    }
    return 0;
}
