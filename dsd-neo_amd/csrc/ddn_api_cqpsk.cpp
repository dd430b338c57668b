// ddn_api_cqpsk.cpp — C-ABI of the batched P25 CQPSK/LSM front end (include/ddn_hip.h): channel LPF -> RMS AGC -> FLL
// band-edge -> Gardner -> differential phasor -> Costas -> phase extractor, i.e. full_demod() with cqpsk_enable
// (reference src/dsp/demod_pipeline.cpp:1100-1118,1330-1350).  Per-channel loop state lives on the device in the batch.

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "ddn_device.h"

static int
taps_have_zero(const float* taps, int n) {
    for (int i = 0; i < n; i++) {
        if (taps[i] == 0.0f) {
            return 1;
        }
    }
    return 0;
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            ddn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                  \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu)             \
                       ? DDN_ENODEV                                                                                    \
                       : (e_ == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP);                                          \
        }                                                                                                              \
    } while (0)

struct ddn_cqpsk_batch {
    ddn_cqpsk_config cfg;
    int sps, taps_len, fll_nt;
    float taps[DDN_MAX_TAPS + 1];
    float fll_taps[4 * DDN_FLL_MAX_TAPS];
    float fll_alpha, fll_beta;
    float* d_taps;
    float* d_fll_taps;      // [4][DDN_FLL_MAX_TAPS], this batch's band-edge taps (uploaded once at create)
    void* d_lpf_hist;       // [B][taps_len-1] complex
    DdnCqpskState* d_state; // [B]
    float* d_delay;         // [2*nt][2][B]
    ddn_ted_batch* ted;
    void *d_a, *d_b;        // [B][n] complex work streams (LPF out; AGC/FLL out)
    size_t work_cap;
    void* d_sym;            // [B][sym_cap] complex Gardner output
    size_t sym_cap;
    int* d_cnt;
};

static void
cq_free(ddn_cqpsk_batch* b) {
    (void)hipFree(b->d_taps);
    (void)hipFree(b->d_fll_taps);
    (void)hipFree(b->d_lpf_hist);
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_delay);
    (void)hipFree(b->d_a);
    (void)hipFree(b->d_b);
    (void)hipFree(b->d_sym);
    (void)hipFree(b->d_cnt);
    if (b->ted) {
        ddn_ted_batch_destroy(b->ted);
    }
}

static int
cq_fill(ddn_cqpsk_batch* b, hipStream_t st) {
    const size_t B = (size_t)b->cfg.n_channels;
    HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(DdnCqpskState) * B, st));
    HIP_TRY(hipMemsetAsync(b->d_delay, 0, sizeof(float) * 4 * (size_t)b->fll_nt * B, st));
    if (b->taps_len >= 3) {
        HIP_TRY(hipMemsetAsync(b->d_lpf_hist, 0, sizeof(float) * 2 * (size_t)(b->taps_len - 1) * B, st));
    }
    return ddn_ted_batch_reset(b->ted, st);
}

extern "C" int
ddn_cqpsk_batch_create(const ddn_cqpsk_config* cfg, ddn_cqpsk_batch** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->sample_rate_hz <= 0 || cfg->symbol_rate_hz <= 0
        || cfg->block_len < 4 || (cfg->input_format != DDN_IN_CU8 && cfg->input_format != DDN_IN_CF32)) {
        ddn_set_error("ddn_cqpsk_batch_create: bad configuration");
        return DDN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_cqpsk_batch* b = new (std::nothrow) ddn_cqpsk_batch();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->cfg = *cfg;
    b->sps = cfg->sample_rate_hz / cfg->symbol_rate_hz;
    if (b->sps < 2) {
        delete b;
        ddn_set_error("ddn_cqpsk_batch_create: fewer than 2 samples per symbol");
        return DDN_ERANGE;
    }
    b->taps_len = cfg->lpf_enable ? ddn_design_channel_lpf(cfg->sample_rate_hz, cfg->lpf_profile, b->taps, DDN_MAX_TAPS) : 0;
    if (cfg->lpf_enable && b->taps_len < 3) {
        delete b;
        ddn_set_error("channel LPF design failed for rate %d profile %d", cfg->sample_rate_hz, cfg->lpf_profile);
        return DDN_ERANGE;
    }
    b->fll_nt = ddn_design_fll_band_edge(b->sps, b->fll_taps, &b->fll_alpha, &b->fll_beta);
    const size_t B = (size_t)cfg->n_channels;
    if (hipMalloc(&b->d_taps, sizeof(float) * (DDN_MAX_TAPS + 1)) != hipSuccess
        || hipMalloc(&b->d_fll_taps, sizeof(b->fll_taps)) != hipSuccess
        || hipMemcpy(b->d_fll_taps, b->fll_taps, sizeof(b->fll_taps), hipMemcpyHostToDevice) != hipSuccess
        || hipMalloc(&b->d_lpf_hist, sizeof(float) * 2 * DDN_MAX_TAPS * B) != hipSuccess
        || hipMalloc(&b->d_state, sizeof(DdnCqpskState) * B) != hipSuccess
        || hipMalloc(&b->d_delay, sizeof(float) * 4 * (size_t)b->fll_nt * B) != hipSuccess
        || hipMalloc(&b->d_cnt, sizeof(int) * B) != hipSuccess
        || (b->taps_len >= 3
            && hipMemcpy(b->d_taps, b->taps, sizeof(float) * (size_t)b->taps_len, hipMemcpyHostToDevice) != hipSuccess)
        || ddn_ted_batch_create(cfg->n_channels, b->sps, cfg->symbol_rate_hz, cfg->ted_gain, &b->ted) != DDN_OK
        || ddn_ted_batch_set_block_len(b->ted, (size_t)cfg->block_len) != DDN_OK
        || cq_fill(b, nullptr) != DDN_OK || hipDeviceSynchronize() != hipSuccess) {
        ddn_set_error("ddn_cqpsk_batch_create: device allocation failed");
        cq_free(b);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_cqpsk_batch_destroy(ddn_cqpsk_batch* b) {
    if (!b) {
        return;
    }
    cq_free(b);
    delete b;
}

extern "C" int
ddn_cqpsk_batch_reset(ddn_cqpsk_batch* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    return cq_fill(b, (hipStream_t)hip_stream);
}

extern "C" size_t
ddn_cqpsk_max_symbols(const ddn_cqpsk_batch* b, size_t n) {
    return b ? n / (size_t)b->sps + n / (size_t)(b->sps * 100) + 8 : 0;
}

extern "C" int
ddn_cqpsk_run(ddn_cqpsk_batch* b, const void* d_iq, size_t n, float* d_symbols, size_t sym_stride, int32_t* d_counts,
              void* hip_stream) {
    if (!b || !d_iq || !d_symbols || !d_counts) {
        ddn_set_error("ddn_cqpsk_run: null argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int B = b->cfg.n_channels;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)B, st));
        return DDN_OK;
    }
    const size_t tail = n % (size_t)b->cfg.block_len;
    if ((tail != 0 && tail < 4) || n < 4) {
        // op25_gardner_cc returns early on blocks shorter than 4 samples and the later stages then see sample-rate
        // data (reference src/dsp/costas.cpp:808-812); not reproduced
        ddn_set_error("ddn_cqpsk_run: a block of fewer than 4 samples is not supported (n %zu, block_len %d)", n,
                      b->cfg.block_len);
        return DDN_ERANGE;
    }
    if (sym_stride < ddn_cqpsk_max_symbols(b, n)) {
        ddn_set_error("ddn_cqpsk_run: sym_stride %zu < ddn_cqpsk_max_symbols() = %zu", sym_stride,
                      ddn_cqpsk_max_symbols(b, n));
        return DDN_ERANGE;
    }
    const size_t need = sizeof(float) * 2 * (size_t)B * n;
    if (b->work_cap < need) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(b->d_a);
        (void)hipFree(b->d_b);
        b->d_a = b->d_b = nullptr;
        b->work_cap = 0;
        HIP_TRY(hipMalloc(&b->d_a, need));
        HIP_TRY(hipMalloc(&b->d_b, need));
        b->work_cap = need;
    }
    const size_t scap = ddn_cqpsk_max_symbols(b, n);
    if (b->sym_cap < scap) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(b->d_sym);
        b->d_sym = nullptr;
        b->sym_cap = 0;
        HIP_TRY(hipMalloc(&b->d_sym, sizeof(float) * 2 * (size_t)B * scap));
        b->sym_cap = scap;
    }
    const void* cur = d_iq;
    int fmt = b->cfg.input_format;
    if (b->taps_len >= 3) {
        HIP_TRY(ddn_dev_channel_lpf_c2c(cur, fmt, (long)n, n, b->cfg.block_len, B, b->d_taps, b->taps_len, taps_have_zero(b->taps, b->taps_len), b->d_lpf_hist,
                                        b->d_a, n, st));
        cur = b->d_a;
        fmt = DDN_IN_CF32;
    } else if (fmt != DDN_IN_CF32) {
        ddn_set_error("ddn_cqpsk_run: cu8 input needs the channel LPF stage (it does the widening)");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_cqpsk_agc_fll(cur, (long)n, n, B, b->fll_nt, b->fll_alpha, b->fll_beta, b->d_fll_taps, b->d_state, b->d_delay, b->d_b,
                                  st));
    int rc = ddn_gardner_run(b->ted, (const float*)b->d_b, n, (float*)b->d_sym, b->sym_cap, b->d_cnt, st);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(ddn_dev_cqpsk_symbols(b->d_sym, b->sym_cap, b->d_cnt, B, b->d_state, d_symbols, sym_stride, st));
    HIP_TRY(hipMemcpyAsync(d_counts, b->d_cnt, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, st));
    return DDN_OK;
}

extern "C" int
ddn_cqpsk_run_host(ddn_cqpsk_batch* b, const void* iq, size_t n, float* symbols, size_t sym_stride, int32_t* counts) {
    if (!b || !iq || !symbols || !counts) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    const size_t in_bytes = B * n * (b->cfg.input_format == DDN_IN_CU8 ? 2 : 8);
    const size_t scap = ddn_cqpsk_max_symbols(b, n);
    if (sym_stride < scap) {
        ddn_set_error("ddn_cqpsk_run_host: sym_stride %zu < %zu", sym_stride, scap);
        return DDN_ERANGE;
    }
    void* d_in = nullptr;
    float* d_out = nullptr;
    int32_t* d_cnt = nullptr;
    int rc;
    if (hipMalloc(&d_in, in_bytes + 8) != hipSuccess || hipMalloc(&d_out, B * scap * 4 + 8) != hipSuccess
        || hipMalloc(&d_cnt, B * 4) != hipSuccess) {
        ddn_set_error("ddn_cqpsk_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, iq, in_bytes, hipMemcpyHostToDevice) != hipSuccess
               || hipMemset(d_out, 0, B * scap * 4) != hipSuccess) {
        rc = DDN_EHIP;
    } else {
        rc = ddn_cqpsk_run(b, d_in, n, d_out, scap, d_cnt, nullptr);
        if (rc == DDN_OK) {
            if (hipDeviceSynchronize() != hipSuccess
                || hipMemcpy2D(symbols, sym_stride * 4, d_out, scap * 4, scap * 4, B, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(counts, d_cnt, B * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                ddn_set_error("ddn_cqpsk_run_host: %s", hipGetErrorString(hipGetLastError()));
                rc = DDN_EHIP;
            }
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_cnt);
    return rc;
}

extern "C" int
ddn_cqpsk_get_state(ddn_cqpsk_batch* b, int channel, float out8[8]) {
    if (!b || !out8 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    DdnCqpskState s;
    float t8[8];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    int rc = ddn_ted_batch_get_state(b->ted, channel, t8);
    if (rc != DDN_OK) {
        return rc;
    }
    out8[0] = s.agc_avg;
    out8[1] = s.fll_freq;
    out8[2] = s.fll_phase;
    out8[3] = s.cos_phase;
    out8[4] = s.cos_freq;
    out8[5] = s.cos_es;
    out8[6] = t8[0];
    out8[7] = t8[1];
    return DDN_OK;
}
