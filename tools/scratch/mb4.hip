#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
// throughput of the FIR step's instruction mix: mode 0: pk_fma only; 1: pk_add + pk_fma pairs (tap in VGPR);
// 2: scalar v_add + v_fma pairs (2x count); 3: pk_add + pk_fma with tap from SGPR (kernel arg)
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float hs) {
    f2 a[8], x[8], y[8];
    for (int j = 0; j < 8; j++) { a[j] = (f2){0.f, 0.f}; x[j] = (f2){threadIdx.x * 1e-3f + j, 1.f}; y[j] = (f2){2.f, j * 0.5f}; }
    float hv = out[threadIdx.x & 7];  // VGPR tap
    const f2 h = (MODE == 3) ? (f2){hs, hs} : (f2){hv, hv};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) { a[j] = __builtin_elementwise_fma(h, x[j], a[j]); }
            else if (MODE == 2) { float t0s = x[j].x + y[j].x, t1s = x[j].y + y[j].y; a[j].x = __builtin_fmaf(h.x, t0s, a[j].x); a[j].y = __builtin_fmaf(h.y, t1s, a[j].y); }
            else { f2 t = x[j] + y[j]; a[j] = __builtin_elementwise_fma(h, t, a[j]); }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { x[j] = y[j]; y[j] = a[(j + 1) & 7]; } // keep operands changing (renamed, no movs)
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int j = 0; j < 8; j++) s += a[j].x + a[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 8 * 4096); (void)hipMemset(out, 0, 1 << 24);
    int iters = 4000;
    for (int mode = 0; mode < 4; mode++) for (int wpb : {4, 8, 16, 32}) {  // waves per block == per CU (1 block per CU)
        dim3 g(256), b(wpb * 64);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, iters, 0.5f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, iters, 0.5f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, iters, 0.5f);
            if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, iters, 0.5f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[1]; (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        double lane_fma = (double)iters * 8 * 2 * 64 * wpb * 256;  // fp32 FMA-lanes (pk counts 2)
        double ops_per_wave = (double)iters * 8 * (mode == 0 ? 1 : (mode == 2 ? 4 : 2));
        printf("mode=%d waves/CU=%2d: %.3f ms  %.1f TFLOP/s(fma only)  %.2f cycles/instr/wave  (%.2f SIMD-cycles per instr)\n", mode, wpb, ms,
               lane_fma * 2 / (ms * 1e-3) / 1e12, (double)h[0] / ops_per_wave, (double)h[0] / ops_per_wave / (wpb / 4.0));
    }
    return 0;
}
