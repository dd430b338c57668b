// d2h_probe.hip - which engine moves a device -> pinned-host copy, and can it run beside a kernel that holds every CU?
// A "hog" kernel (256 VGPRs per wave via launch bounds + 64 KB LDS per workgroup, 2 workgroups per CU, spinning ~20 ms) runs on one
// stream; 128 MB device -> host copies run on another: hipMemcpyAsync into hipHostMalloc memory, and - the alternative - a copy KERNEL of
// our own storing straight into mapped host memory (zero-copy) launched BEFORE the hog.  Reports when each copy finished relative to the
// hog.  Run under different runtime environments (GPU_FORCE_BLIT_COPY_SIZE, HSA_ENABLE_SDMA_COPY_SIZE_OVERRIDE ...) to see what moves the
// copy off the shader.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256, 1) void hog(float* out, long long ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    float a = lds[(threadIdx.x + 1) & 255];
    while (wall_clock64() - t0 < ticks) {
        for (int i = 0; i < 64; i++) a = a * 1.0001f + 0.5f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ void copy_k(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const size_t nb = 128u << 20;
    void *d_src, *h_dst, *h_map;
    float* d_out;
    CK(hipMalloc(&d_src, nb));
    CK(hipMemset(d_src, 1, nb));
    CK(hipHostMalloc(&h_dst, nb, hipHostMallocDefault));
    CK(hipHostMalloc(&h_map, nb, hipHostMallocMapped));
    void* d_map = nullptr;
    CK(hipHostGetDevicePointer(&d_map, h_map, 0));
    CK(hipMalloc(&d_out, 512 * 256 * 4));
    CK(hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e_h0, e_h1, e_c0, e_c1;
    CK(hipEventCreate(&e_h0)); CK(hipEventCreate(&e_h1)); CK(hipEventCreate(&e_c0)); CK(hipEventCreate(&e_c1));
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            float hog_ms = 0, c0 = 0, c1 = 0;
            if (mode == 3) { // zero-copy store kernel queued first (it is resident before the hog arrives), low occupancy: 64 workgroups
                CK(hipEventRecord(e_c0, s2));
                hipLaunchKernelGGL(copy_k, dim3(64), dim3(256), 0, s2, (const uint4*)d_src, (uint4*)d_map, nb / 16);
                CK(hipEventRecord(e_c1, s2));
            }
            CK(hipEventRecord(e_h0, s1));
            if (mode != 0) hipLaunchKernelGGL(hog, dim3(512), dim3(256), 72 * 1024, s1, d_out, 2000000LL); // 20 ms at 100 MHz
            CK(hipEventRecord(e_h1, s1));
            if (mode == 0 || mode == 1) {
                CK(hipEventRecord(e_c0, s2));
                CK(hipMemcpyAsync(h_dst, d_src, nb, hipMemcpyDeviceToHost, s2));
                CK(hipEventRecord(e_c1, s2));
            } else if (mode == 2) {
                CK(hipEventRecord(e_c0, s2));
                hipLaunchKernelGGL(copy_k, dim3(64), dim3(256), 0, s2, (const uint4*)d_src, (uint4*)d_map, nb / 16);
                CK(hipEventRecord(e_c1, s2));
            }
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&hog_ms, e_h0, e_h1));
            CK(hipEventElapsedTime(&c0, e_h0, e_c0));
            CK(hipEventElapsedTime(&c1, e_h0, e_c1));
            if (rep == 1) {
                const char* nm[] = {"hipMemcpyAsync D2H alone", "hipMemcpyAsync D2H queued behind a resident hog", "zero-copy store kernel queued behind the hog",
                                    "zero-copy store kernel queued before the hog"};
                printf("%-52s hog %6.2f ms | copy ran %7.2f .. %7.2f ms after the hog's start (%.1f GB/s while running)\n", nm[mode], hog_ms, c0, c1,
                       nb / ((c1 - c0) * 1e-3) / 1e9);
            }
        }
    }
    return 0;
}
