// single-wave issue-rate probe: cycles per instruction for dependent / independent fp32 adds, v_cndmask after v_cmp, fp64 adds,
// LDS read round trip, s_memtime cost
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(long long* out, float a, double da, int n) {
    __shared__ float sh[256];
    sh[threadIdx.x] = a + threadIdx.x;
    __syncthreads();
    float x = a + threadIdx.x, y = a * 2, z = a * 3, w = a * 5;
    double d = da + threadIdx.x;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = x + 1.25f; // dependent chain
    }
    long long t1 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { x = x + 1.25f; y = y + 1.5f; z = z + 1.75f; w = w + 2.25f; } // 4 independent chains
    }
    long long t2 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) d = d + 1.25; // dependent fp64
    }
    long long t3 = clock64();
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = (x > y) ? x - 1.0f : x + 2.0f; // cmp + select chain (3 instr)
    }
    long long t4 = clock64();
    int idx = threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) idx = (int)sh[idx & 255] & 255; // dependent LDS round trips
    }
    long long t5 = clock64();
    long long t6 = clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5;
    }
    if (x + y + z + w + (float)d + idx == 12345.678f) out[7] = 1;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    long long h[8];
    for (int waves = 1; waves <= 2; waves++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, 1.0, 100);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("waves/block %d: dep f32 add %.2f cyc/instr | 4 indep chains %.2f | dep f64 add %.2f | cmp+sel+add per step %.2f | LDS round trip %.1f | clock64 %lld\n",
               waves, h[0] / 6400.0, h[1] / 6400.0, h[2] / 6400.0, h[3] / 3200.0, h[4] / 1600.0, h[5]);
    }
    return 0;
}
