#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(float* out, long long* cyc, float a, float b, int iters, int mode) {
    float x = threadIdx.x * 1e-3f, y = x + 1.f, z = x + 2.f, w = x + 3.f;
    long long t0 = __builtin_readcyclecounter();
    if (mode == 0) { // 1 dependent chain: sub, mul, add
        for (int i = 0; i < iters; i++) { float t = b - x; t = t * a; x = x + t; }
    } else if (mode == 1) { // 2 independent chains
        for (int i = 0; i < iters; i++) { float t = b - x; float u = b - y; t = t * a; u = u * a; x = x + t; y = y + u; }
    } else if (mode == 2) { // 4 independent chains
        for (int i = 0; i < iters; i++) { float t = b - x; float u = b - y; float v = b - z; float q = b - w; t = t * a; u = u * a; v *= a; q *= a; x = x + t; y = y + u; z += v; w += q; }
    } else if (mode == 3) { // pure dependent adds, unrolled: 24 dependent ops per iteration
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) { x = x + a; x = x * b; x = x + a; }
        }
    } else if (mode == 4) { // 2 independent chains, 24+24 ops
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) { x = x + a; y = y + a; x = x * b; y = y * b; x = x + a; y = y + a; }
        }
    } else if (mode == 5) { // 3 independent chains
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) { x = x + a; y = y + a; z = z + a; x = x * b; y = y * b; z = z * b; x = x + a; y = y + a; z = z + a; }
        }
    } else if (mode == 6) { // peak-like chain: sub, 2 mul, max, add
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) { float d = y - x; float u = 0.125f * d; float v = 0.00005f * d; x = x + fmaxf(u, v); y = y + a; }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 1024);
    for (int lanes : {64}) for (int mode = 3; mode < 7; mode++) {
        int iters = 20000;
        hipLaunchKernelGGL(k_chain, dim3(256), dim3(lanes), 0, 0, out, cyc, 1e-4f, 0.5f, iters, mode);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain, dim3(256), dim3(lanes), 0, 0, out, cyc, 1e-4f, 0.5f, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        int ops = (mode == 3 ? 24 : mode == 4 ? 48 : mode == 5 ? 72 : 48);
        printf("lanes=%d mode=%d: %.3f ms -> %.2f ns/iter, %.2f ns per VALU op; counter ticks/iter %.2f\n", lanes, mode, ms, ms * 1e6 / iters, ms * 1e6 / iters / ops, (double)h[0] / iters);
    }
    return 0;
}
