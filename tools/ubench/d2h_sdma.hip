// d2h_sdma.hip - can a device -> pinned-host copy leave on an SDMA engine (not a shader blit) while a kernel holds every CU, and do H2D + D2H
// then run duplex?  hipMemcpyAsync D2H into hipHostMalloc memory is a __amd_rocclr_copyBuffer shader kernel on this ROCm (DESIGN 6,
// d2h_probe.hip); here the copy is issued below HIP: hsa_amd_memory_async_copy_on_engine(host <- device) on an engine
// hsa_amd_memory_copy_engine_status reports for that direction, completion by HSA signal.
//   build: hipcc --offload-arch=gfx950 -O2 d2h_sdma.hip -o d2h_sdma -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); printf("%s: %s\n", #x, m); exit(1); } } while (0)

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

static std::vector<hsa_agent_t> gpus, cpus;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    (t == HSA_DEVICE_TYPE_GPU ? gpus : cpus).push_back(a);
    return HSA_STATUS_SUCCESS;
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const size_t nb = 256u << 20;
    void *d_src, *d_dst, *h_dst, *h_src;
    float* d_out;
    CK(hipMalloc(&d_src, nb));
    CK(hipMalloc(&d_dst, nb));
    CK(hipMemset(d_src, 1, nb));
    CK(hipHostMalloc(&h_dst, nb, hipHostMallocDefault));
    CK(hipHostMalloc(&h_src, nb, hipHostMallocDefault));
    CK(hipMalloc(&d_out, 512 * 256 * 4));
    CK(hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    HK(hsa_init());
    HK(hsa_iterate_agents(on_agent, nullptr));
    printf("agents: %zu gpu, %zu cpu\n", gpus.size(), cpus.size());
    if (gpus.empty() || cpus.empty()) return 1;
    hsa_agent_t gpu = gpus[0], cpu = cpus[0];
    // which CPU agent owns the pinned buffer does not matter for the engine choice; take the pointer's own agent if reported
    hsa_amd_pointer_info_t pi;
    pi.size = sizeof(pi);
    if (hsa_amd_pointer_info(h_dst, &pi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS) {
        printf("pinned host pointer: type %d, owner handle %llx\n", (int)pi.type, (unsigned long long)pi.agentOwner.handle);
        for (auto c : cpus) if (c.handle == pi.agentOwner.handle) cpu = c;
    }
    uint32_t mask_d2h = 0, mask_h2d = 0, pref_d2h = 0, pref_h2d = 0;
    hsa_status_t s = hsa_amd_memory_copy_engine_status(cpu, gpu, &mask_d2h);
    printf("engine status D2H (dst cpu, src gpu): status %d mask 0x%x\n", (int)s, mask_d2h);
    s = hsa_amd_memory_copy_engine_status(gpu, cpu, &mask_h2d);
    printf("engine status H2D (dst gpu, src cpu): status %d mask 0x%x\n", (int)s, mask_h2d);
    s = hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &pref_d2h);
    printf("preferred D2H: status %d mask 0x%x\n", (int)s, pref_d2h);
    s = hsa_amd_memory_get_preferred_copy_engine(gpu, cpu, &pref_h2d);
    printf("preferred H2D: status %d mask 0x%x\n", (int)s, pref_h2d);
    hsa_signal_t sig, sig2;
    HK(hsa_signal_create(1, 0, nullptr, &sig));
    HK(hsa_signal_create(1, 0, nullptr, &sig2));

    auto wait_sig = [](hsa_signal_t g) {
        while (hsa_signal_wait_scacquire(g, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
    };
    // pick engines: lowest set bit of the mask for D2H, a different one for H2D if there is one
    auto lowbit = [](uint32_t m) { return m & (~m + 1); };
    uint32_t e_d2h = lowbit(pref_d2h ? pref_d2h : mask_d2h);
    uint32_t rest = (pref_h2d ? pref_h2d : mask_h2d) & ~e_d2h;
    uint32_t e_h2d = lowbit(rest ? rest : mask_h2d);
    printf("using engine 0x%x for D2H, 0x%x for H2D\n", e_d2h, e_h2d);

    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const bool with_hog = mode == 1 || mode == 3 || mode == 5;
            const double t0 = now_ms();
            if (with_hog) hipLaunchKernelGGL(hog, dim3(512), dim3(256), 72 * 1024, s1, d_out, 2000000LL); // 20 ms at 100 MHz
            double t_d2h = -1, t_h2d = -1;
            if (mode == 0 || mode == 1) { // D2H alone on the engine
                hsa_signal_store_relaxed(sig, 1);
                hsa_status_t r = e_d2h ? hsa_amd_memory_async_copy_on_engine(h_dst, cpu, d_src, gpu, nb, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e_d2h, false)
                                       : hsa_amd_memory_async_copy(h_dst, cpu, d_src, gpu, nb, 0, nullptr, sig);
                if (r != HSA_STATUS_SUCCESS) { printf("copy_on_engine D2H: status %d\n", (int)r); return 1; }
                wait_sig(sig);
                t_d2h = now_ms() - t0;
            } else if (mode == 2 || mode == 3) { // duplex: H2D through hipMemcpyAsync (SDMA), D2H on the engine
                hsa_signal_store_relaxed(sig, 1);
                CK(hipMemcpyAsync(d_dst, h_src, nb, hipMemcpyHostToDevice, s2));
                hsa_status_t r = e_d2h ? hsa_amd_memory_async_copy_on_engine(h_dst, cpu, d_src, gpu, nb, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e_d2h, false)
                                       : hsa_amd_memory_async_copy(h_dst, cpu, d_src, gpu, nb, 0, nullptr, sig);
                if (r != HSA_STATUS_SUCCESS) { printf("copy_on_engine D2H: status %d\n", (int)r); return 1; }
                wait_sig(sig);
                t_d2h = now_ms() - t0;
                CK(hipStreamSynchronize(s2));
                t_h2d = now_ms() - t0;
            } else { // both through HSA engines
                hsa_signal_store_relaxed(sig, 1);
                hsa_signal_store_relaxed(sig2, 1);
                hsa_status_t r = hsa_amd_memory_async_copy_on_engine(d_dst, gpu, h_src, cpu, nb, 0, nullptr, sig2, (hsa_amd_sdma_engine_id_t)e_h2d, false);
                if (r != HSA_STATUS_SUCCESS) { printf("copy_on_engine H2D: status %d\n", (int)r); return 1; }
                r = hsa_amd_memory_async_copy_on_engine(h_dst, cpu, d_src, gpu, nb, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e_d2h, false);
                if (r != HSA_STATUS_SUCCESS) { printf("copy_on_engine D2H: status %d\n", (int)r); return 1; }
                wait_sig(sig);
                t_d2h = now_ms() - t0;
                wait_sig(sig2);
                t_h2d = now_ms() - t0;
            }
            CK(hipDeviceSynchronize());
            const double t_all = now_ms() - t0;
            if (rep == 1) {
                const char* nm[] = {"HSA engine D2H alone", "HSA engine D2H beside a hog holding every CU", "hipMemcpyAsync H2D + HSA engine D2H", "the same beside the hog",
                                    "HSA engine H2D + HSA engine D2H", "the same beside the hog"};
                printf("%-46s D2H done at %6.2f ms (%5.1f GB/s)", nm[mode], t_d2h, nb / (t_d2h * 1e-3) / 1e9);
                if (t_h2d >= 0) printf("  H2D done at %6.2f ms (%5.1f GB/s)  both %5.1f GB/s", t_h2d, nb / (t_h2d * 1e-3) / 1e9, 2.0 * nb / ((t_d2h > t_h2d ? t_d2h : t_h2d) * 1e-3) / 1e9);
                printf("  everything idle at %6.2f ms\n", t_all);
            }
        }
    }
    unsigned char* hb = (unsigned char*)h_dst;
    printf("data check: h_dst[0] = %d, h_dst[last] = %d (want 1)\n", hb[0], hb[nb - 1]);
    return 0;
}
