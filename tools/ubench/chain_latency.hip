// chain_latency.hip - cycles per dependent instruction for a single wavefront per SIMD on gfx950 (what bounds the receive loop's
// recurrence wave): f32 add / med3 / fma chains, f64 add / ldexp / cvt chains, v_rcp, LDS read-to-use, DPP moves, a not-taken and a
// taken scalar branch, and the same chains with a second busy wave on the SIMD.  One workgroup of 64 threads per SIMD (grid 1024),
// 4 active lanes like the loop.  Prints cycles per op (s_memtime ticks are 100 MHz on gfx9; clock64() = s_memtime -> we use
// wall time of the kernel and the known op count instead, plus s_memrealtime deltas converted with the measured ratio).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define N_ITER 20000

template <int KIND>
__global__ __launch_bounds__(64) void k_chain(float* out, const float* in, int n_iter, int active) {
    __shared__ float lds[1024];
    const int lane = threadIdx.x;
    lds[lane] = in[lane];
    lds[lane + 64] = in[lane + 64];
    __syncthreads();
    if (lane >= active) {
        return;
    }
    float a = in[lane], b = in[lane + 1], c = in[lane + 2];
    double d = (double)in[lane + 3];
    int idx = lane;
    const long long t_core0 = clock64(), t_wall0 = wall_clock64();
    for (int i = 0; i < n_iter; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
            } else if (KIND == 1) {
                asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            } else if (KIND == 2) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            } else if (KIND == 3) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(d));
            } else if (KIND == 4) {
                asm volatile("v_ldexp_f64 %0, %0, -1" : "+v"(d));
            } else if (KIND == 5) { // f32 -> f64 -> f32 round trip
                asm volatile("v_cvt_f64_f32 %0, %1\n\tv_cvt_f32_f64 %1, %0" : "+v"(d), "+v"(a));
            } else if (KIND == 6) {
                asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
            } else if (KIND == 7) { // LDS read -> use as next address
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0xfc, %0" : "+v"(a), "+v"(idx));
            } else if (KIND == 8) {
                asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a));
            } else if (KIND == 9) { // independent pair: two chains interleaved
                asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));
            } else if (KIND == 10) { // v_cmp + exec-masked region (what an `if` on a lane predicate costs)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_saveexec_b64 s[10:11], vcc\n\tv_add_f32 %0, %0, %1\n\ts_or_b64 exec, exec, s[10:11]"
                             : "+v"(a) : "v"(b) : "vcc", "s10", "s11");
            } else if (KIND == 11) { // v_cmp + s_cbranch_vccz not taken
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_cbranch_vccz 1f\n\tv_add_f32 %0, %0, %1\n1:" : "+v"(a) : "v"(b) : "vcc");
            } else if (KIND == 12) { // global store in the chain (fire and forget)
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                out[4096 + lane + ((i * 16 + u) & 1023) * 64] = a;
            } else if (KIND == 13) { // ds_write in the chain
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                lds[lane + ((u & 7) << 6)] = a;
            } else if (KIND == 14) { // v_min3
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            } else if (KIND == 15) { // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d));
            }
        }
    }
    const long long t_core1 = clock64(), t_wall1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = a + c + (float)d + (float)idx;
    if (lane == 0) {
        ((long long*)(out + 3 * 1024 * 1024))[2 * blockIdx.x] = t_core1 - t_core0;
        ((long long*)(out + 3 * 1024 * 1024))[2 * blockIdx.x + 1] = t_wall1 - t_wall0;
    }
}

// a second wave per SIMD doing VALU work beside the chain (KIND 0 chain in wave 0, throughput loop in wave 1)
__global__ __launch_bounds__(128) void k_chain_shared(float* out, const float* in, int n_iter, int prio) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a = in[lane], b = in[lane + 1];
    if (w == 0) {
        if (prio) {
            __builtin_amdgcn_s_setprio(3);
        }
        if (lane >= 4) {
            return;
        }
        for (int i = 0; i < n_iter; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
            }
        }
    } else {
        float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
        for (int i = 0; i < n_iter; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
            }
        }
        a = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 128 + threadIdx.x] = a;
}

int main() {
    float *d_in, *d_out;
    hipMalloc(&d_in, 4096);
    hipMalloc(&d_out, 16u << 20);
    std::vector<float> h(1024, 1.0f);
    for (int i = 0; i < 1024; i++) {
        h[i] = 1.0f + 0.001f * i;
    }
    hipMemcpy(d_in, h.data(), 4096, hipMemcpyHostToDevice);
    setvbuf(stdout, NULL, _IONBF, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk_khz);
    const char* names[] = {"v_add_f32", "v_med3_f32", "v_fma_f32", "v_add_f64", "v_ldexp_f64", "cvt f32->f64->f32 (2 ops)", "v_rcp_f32",
                           "ds_read_b32 -> waitcnt -> v_and (3 ops)", "v_mov_b32_dpp row_ror:8", "2 independent v_add_f32 (2 ops)",
                           "v_cmp + saveexec + v_add + s_or (4 ops)", "v_cmp + s_cbranch_vccz(not taken) + v_add (3 ops)",
                           "v_add + global_store", "v_add + ds_write", "v_min3_f32", "v_pk_add_f32"};
    for (int w = 0; w < 40; w++) hipLaunchKernelGGL(k_chain<2>, dim3(4096), dim3(64), 0, 0, d_out, d_in, N_ITER, 64);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {1024, 2048}) {
        printf("grid %d workgroups of one wave (%d per SIMD), 4 active lanes\n", grid, grid / 1024);
        for (int kind = 0; kind < 16; kind++) {
            if (kind == 10) continue;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k_chain<K>, dim3(grid), dim3(64), 0, 0, d_out, d_in, N_ITER, 4); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<long long> t(2 * grid);
            hipMemcpy(t.data(), d_out + 3 * 1024 * 1024, 16 * grid, hipMemcpyDeviceToHost);
            std::vector<double> core(grid), ns(grid);
            for (int g = 0; g < grid; g++) {
                core[g] = (double)t[2 * g] / ((double)N_ITER * 16);
                ns[g] = (double)t[2 * g + 1] * 10.0 / ((double)N_ITER * 16);
            }
            std::sort(core.begin(), core.end());
            std::sort(ns.begin(), ns.end());
            printf("  %-52s s_memtime ticks/step min %6.2f med %6.2f max %6.2f | ns/step min %6.2f med %6.2f max %6.2f | kernel %.3f ms\n", names[kind],
                   core[0], core[grid / 2], core[grid - 1], ns[0], ns[grid / 2], ns[grid - 1], ms);
        }
    }
    for (int prio = 0; prio < 2; prio++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain_shared, dim3(1024), dim3(128), 0, 0, d_out + 4096 + 65536, d_in, N_ITER, prio);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("chain of v_add_f32 beside a wave of 16 independent fmas per step, setprio %d: kernel %.3f ms (chain alone would be add-rate)\n", prio, ms);
    }
    return 0;
}
