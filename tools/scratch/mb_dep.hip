// dependent vs independent VALU issue latency of a single wavefront (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long long* cyc, int n) {
    float a = out[threadIdx.x], b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a = a * 1.0001f;       // dependent mul
            a = a + 0.5f;          // dependent add
        }
    }
    long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a = a * 1.0001f; b = b * 1.0001f; c = c * 1.0001f; d = d * 1.0001f;   // 4 independent chains
            a = a + 0.5f; b = b + 0.5f; c = c + 0.5f; d = d + 0.5f;
        }
    }
    long long t2 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    float* o; long long* c; hipMalloc(&o, 256); hipMalloc(&c, 16); hipMemset(o, 0, 256);
    for (int lanes : {64, 16}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(lanes), 0, 0, o, c, 1000);
        long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        printf("lanes %d: dependent %.2f cyc/instr, 4-way independent %.2f cyc/instr (counter ticks)\n", lanes, h[0] / 32000.0, h[1] / 32000.0);
    }
    return 0;
}
