// Where the dispatcher puts the waves of a 256-thread workgroup (HW_REG_HW_ID: SIMD, wave slot, CU, SE; HW_REG_XCC_ID) at the receive
// loop's launch shape (512 workgroups, 78 KB LDS each) - the measurement behind the "roles by SIMD" step of DESIGN 5g.
// hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    lds[threadIdx.x] = 0;
}
int main() {
    const int nb = 512;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 78000);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 78000, 0, d, 2000000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; b += (b < 8 ? 1 : 37)) {
        printf("wg %3d:", b);
        for (int w = 0; w < 4; w++) {
            unsigned id = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  [xcc %u se %u sh %u cu %2u simd %u slot %u]", x & 15, (id >> 13) & 7, (id >> 12) & 1, (id >> 8) & 15, (id >> 4) & 3, id & 15);
        }
        printf("\n");
    }
    // how many (xcc,se,sh,cu,simd) pairs host two wave-0/1 (recurrence) waves
    int hist[8][8][2][16][4] = {};
    for (int b = 0; b < nb; b++) for (int w = 0; w < 2; w++) {
        unsigned id = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
        hist[x & 7][(id >> 13) & 7][(id >> 12) & 1][(id >> 8) & 15][(id >> 4) & 3]++;
    }
    int c[4] = {};
    for (auto& a : hist) for (auto& b2 : a) for (auto& c2 : b2) for (auto& d2 : c2) for (int v : d2) c[v < 3 ? v : 3]++;
    printf("SIMDs with 0/1/2/3+ recurrence waves: %d %d %d %d\n", c[0], c[1], c[2], c[3]);
}
