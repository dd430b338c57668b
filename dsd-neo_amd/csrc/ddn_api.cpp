// ddn_api.cpp — C-ABI of libdsdneo_hip.so (batched front end).  See include/ddn_hip.h.
//
// Host-side counterpart of the reference's per-stream demod state (struct demod_state,
// include/dsd-neo/dsp/demod_state.h:67-262) re-laid-out for B channels on one GPU: only the carried words
// live on the device (FIR look-back, modem dc/peak/prev); the 10 MB of per-stream scratch buffers of the
// reference are replaced by per-call tile side-buffers sized for the call.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ddn_device.h"

static thread_local char g_err[512] = "";

extern "C" void
ddn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char*
ddn_last_error(void) {
    return g_err;
}

extern "C" const char*
ddn_version(void) {
    return "dsdneo-hip 0.1 (gfx950)";
}

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
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu              \
                    || e_ == hipErrorInsufficientDriver)                                                               \
                       ? DDN_ENODEV                                                                                    \
                       : (e_ == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP);                                          \
        }                                                                                                              \
    } while (0)

struct ddn_batch {
    ddn_front_end_config cfg;
    float taps[DDN_MAX_TAPS + 1];
    int taps_len;
    int center;
    int group; // channels per workgroup of the fused kernel (8 or 16; 0 = by batch size)
    float* d_taps;
    // segments (ddn_batch_set_segments): tap sets of the second and third run of the channel index, where those runs start
    int n_seg, seg_first[3];
    float taps_s[2][DDN_MAX_TAPS + 1];
    float* d_taps_s[2];
    ddn_f2* d_carry;      // [B][DDN_CARRY_LEN]
    DdnFskState* d_state; // [B]
    // host-call staging
    void* d_in;
    size_t in_cap;
    float* d_out;
    size_t out_cap;
    int timing;
    hipEvent_t ev[3];
    int ev_valid;
    // half-band decimation cascade in front of the channel LPF (0 = none)
    int passes;
    void* d_hbhist[DDN_MAX_HB_PASSES]; // [B][taps_len-1] complex per stage
    void* d_dec[2];                    // ping-pong decimated streams
    size_t dec_cap[2];
    // optional IQ conditioning (row a5): unfused route channel LPF -> k_iq_cond_disc
    DdnIqCondConfig iqc;
    DdnIqCondState* d_iqstate; // [B]
    void* d_lpf_hist;          // [B][taps_len - 1] complex
    void* d_lpf;               // [B][n] complex channel-LPF output
    size_t lpf_cap;
};

static int
ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        ddn_set_error("no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return DDN_ENODEV;
    }
    return DDN_OK;
}

extern "C" int
ddn_batch_create(const ddn_front_end_config* cfg, ddn_batch** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->sample_rate_hz <= 0) {
        ddn_set_error("ddn_batch_create: bad config");
        return DDN_EINVAL;
    }
    if (cfg->input_format != DDN_IN_CU8 && cfg->input_format != DDN_IN_CF32) {
        ddn_set_error("ddn_batch_create: unknown input_format %d", cfg->input_format);
        return DDN_EINVAL;
    }
    int rc = ensure_device();
    if (rc != DDN_OK) {
        return rc;
    }
    ddn_batch* b = new (std::nothrow) ddn_batch();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->cfg = *cfg;
    b->taps_len = ddn_design_channel_lpf(cfg->sample_rate_hz, cfg->lpf_profile, b->taps, DDN_MAX_TAPS);
    if (b->taps_len < 3) {
        ddn_set_error("channel LPF design failed for rate %d profile %d (taps=%d); the reference falls back to a "
                      "fixed 63-tap table here, which this build does not carry",
                      cfg->sample_rate_hz, cfg->lpf_profile, b->taps_len);
        delete b;
        return DDN_ERANGE;
    }
    b->center = (b->taps_len - 1) / 2;
    {
        const char* gsel = DDN_EXP_ENV("DDN_GROUP");
        b->group = (gsel && atoi(gsel) == 16) ? 16 : ((gsel && atoi(gsel) == 8) ? 8 : 0);
    }
    // the reference routes blocks shorter than 2*taps_len floats to its non-FMA scalar unit
    // (src/dsp/simd_fir.cpp:303-306); this build implements the FMA (AVX2-unit) order only.
    if (cfg->block_len < b->taps_len) {
        ddn_set_error("block_len %d < taps_len %d is not supported", cfg->block_len, b->taps_len);
        delete b;
        return DDN_ERANGE;
    }
    const size_t B = (size_t)cfg->n_channels;
    auto fail = [&](int code) {
        ddn_batch_destroy(b);
        return code;
    };
    if (hipMalloc(&b->d_taps, sizeof(float) * (DDN_MAX_TAPS + 1)) != hipSuccess
        || hipMalloc(&b->d_carry, sizeof(ddn_f2) * B * DDN_CARRY_LEN) != hipSuccess
        || hipMalloc(&b->d_state, sizeof(DdnFskState) * B) != hipSuccess) {
        ddn_set_error("hipMalloc failed for batch state");
        return fail(DDN_ENOMEM);
    }
    if (hipMemcpy(b->d_taps, b->taps, sizeof(float) * (size_t)b->taps_len, hipMemcpyHostToDevice) != hipSuccess) {
        ddn_set_error("tap upload failed");
        return fail(DDN_EHIP);
    }
    for (int i = 0; i < 3; i++) {
        if (hipEventCreate(&b->ev[i]) != hipSuccess) {
            ddn_set_error("hipEventCreate failed");
            return fail(DDN_EHIP);
        }
    }
    rc = ddn_batch_reset(b, nullptr);
    if (rc != DDN_OK) {
        return fail(rc);
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        return fail(DDN_EHIP);
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_batch_destroy(ddn_batch* b) {
    if (!b) {
        return;
    }
    (void)hipFree(b->d_taps);
    (void)hipFree(b->d_taps_s[0]);
    (void)hipFree(b->d_taps_s[1]);
    (void)hipFree(b->d_carry);
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_in);
    (void)hipFree(b->d_out);
    (void)hipFree(b->d_dec[0]);
    (void)hipFree(b->d_dec[1]);
    (void)hipFree(b->d_iqstate);
    (void)hipFree(b->d_lpf_hist);
    (void)hipFree(b->d_lpf);
    for (int i = 0; i < DDN_MAX_HB_PASSES; i++) {
        (void)hipFree(b->d_hbhist[i]);
    }
    for (int i = 0; i < 3; i++) {
        if (b->ev[i]) {
            (void)hipEventDestroy(b->ev[i]);
        }
    }
    delete b;
}

extern "C" int
ddn_batch_set_decimation(ddn_batch* b, int passes) {
    if (!b || passes < 0 || passes > DDN_MAX_HB_PASSES) {
        ddn_set_error("ddn_batch_set_decimation: passes must be 0..%d", DDN_MAX_HB_PASSES);
        return DDN_EINVAL;
    }
    if ((b->cfg.block_len & ((1 << passes) - 1)) != 0 || (b->cfg.block_len >> passes) < b->taps_len) {
        ddn_set_error("ddn_batch_set_decimation: block_len %d must be a multiple of %d and leave >= %d samples",
                      b->cfg.block_len, 1 << passes, b->taps_len);
        return DDN_ERANGE;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    for (int i = 0; i < passes; i++) {
        if (!b->d_hbhist[i] && hipMalloc(&b->d_hbhist[i], sizeof(ddn_f2) * B * 30) != hipSuccess) {
            ddn_set_error("ddn_batch_set_decimation: hipMalloc failed");
            return DDN_ENOMEM;
        }
    }
    b->passes = passes;
    return ddn_batch_reset(b, nullptr);
}

extern "C" int
ddn_batch_reset(ddn_batch* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    for (int i = 0; i < b->passes; i++) {
        HIP_TRY(hipMemsetAsync(b->d_hbhist[i], 0, sizeof(ddn_f2) * (size_t)b->cfg.n_channels * 30,
                               (hipStream_t)hip_stream));
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t B = (size_t)b->cfg.n_channels;
    HIP_TRY(ddn_dev_zero(b->d_carry, sizeof(ddn_f2) * B * DDN_CARRY_LEN, st));
    HIP_TRY(ddn_dev_zero(b->d_state, sizeof(DdnFskState) * B, st));
    if (b->d_iqstate) {
        HIP_TRY(hipMemsetAsync(b->d_iqstate, 0, sizeof(DdnIqCondState) * B, st));
        HIP_TRY(hipMemsetAsync(b->d_lpf_hist, 0, sizeof(ddn_f2) * B * (size_t)(b->taps_len - 1), st));
    }
    return DDN_OK;
}

extern "C" int
ddn_batch_set_iq_conditioning(ddn_batch* b, int dc_block_enable, int dc_shift, int iqbal_enable, float iqbal_thr,
                              float iqbal_ema_alpha) {
    if (!b) {
        return DDN_EINVAL;
    }
    if ((dc_block_enable || iqbal_enable) && !b->d_iqstate) {
        const size_t B = (size_t)b->cfg.n_channels;
        if (hipMalloc(&b->d_iqstate, sizeof(DdnIqCondState) * B) != hipSuccess
            || hipMalloc(&b->d_lpf_hist, sizeof(ddn_f2) * B * (size_t)(b->taps_len - 1)) != hipSuccess) {
            (void)hipFree(b->d_iqstate);
            b->d_iqstate = nullptr;
            ddn_set_error("ddn_batch_set_iq_conditioning: device allocation failed");
            return DDN_ENOMEM;
        }
    }
    b->iqc.dc_enable = dc_block_enable ? 1 : 0;
    b->iqc.dc_shift = dc_shift;
    b->iqc.bal_enable = iqbal_enable ? 1 : 0;
    b->iqc.bal_thr = iqbal_thr;
    b->iqc.bal_ema_a = iqbal_ema_alpha;
    return ddn_batch_reset(b, nullptr);
}

extern "C" int
ddn_batch_get_taps(const ddn_batch* b, float* taps_out, int cap) {
    if (!b || !taps_out) {
        return DDN_EINVAL;
    }
    for (int i = 0; i < b->taps_len && i < cap; i++) {
        taps_out[i] = b->taps[i];
    }
    return b->taps_len;
}

static int
grow(void** p, size_t* cap, size_t need) {
    if (*cap >= need) {
        return DDN_OK;
    }
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    if (hipMalloc(p, need) != hipSuccess) {
        ddn_set_error("hipMalloc(%zu) failed", need);
        return DDN_ENOMEM;
    }
    *cap = need;
    return DDN_OK;
}

extern "C" int
ddn_batch_set_channels_per_workgroup(ddn_batch* b, int channels) {
    if (!b || (channels != 0 && channels != 8 && channels != 16)) {
        return DDN_EINVAL;
    }
    b->group = channels;
    return DDN_OK;
}

// Segments: the batch's channel index cut into up to three runs, each with its own channel low-pass profile and (at run time) its own
// input and output arrays - the protocol groups of a mixed batch behind ONE front-end launch of ceil(n_channels / 16) workgroups.
extern "C" int
ddn_batch_set_segments(ddn_batch* b, int n_seg, const int32_t* seg_channels, const int32_t* lpf_profiles) {
    if (!b || n_seg < 1 || n_seg > 3 || !seg_channels || !lpf_profiles) {
        return DDN_EINVAL;
    }
    long total = 0;
    for (int k = 0; k < n_seg; k++) {
        if (seg_channels[k] <= 0) {
            ddn_set_error("ddn_batch_set_segments: empty segment %d", k);
            return DDN_EINVAL;
        }
        total += seg_channels[k];
    }
    if (total != b->cfg.n_channels || lpf_profiles[0] != b->cfg.lpf_profile || b->passes > 0 || b->iqc.dc_enable || b->iqc.bal_enable
        || b->cfg.squelch_level > 0.0f) {
        ddn_set_error("ddn_batch_set_segments: the segments must add up to the batch, segment 0 keeps the batch's profile, and the "
                      "half-band cascade / IQ conditioning / squelch routes have no segmented form");
        return DDN_EINVAL;
    }
    for (int k = 1; k < n_seg; k++) {
        const int len = ddn_design_channel_lpf(b->cfg.sample_rate_hz, lpf_profiles[k], b->taps_s[k - 1], DDN_MAX_TAPS);
        if (len != b->taps_len) { // one kernel instance walks every channel: the tap count is the launch's
            ddn_set_error("ddn_batch_set_segments: profile %d designs %d taps, the batch's %d", lpf_profiles[k], len, b->taps_len);
            return DDN_ERANGE;
        }
        if (!b->d_taps_s[k - 1]) {
            HIP_TRY(hipMalloc(&b->d_taps_s[k - 1], sizeof(float) * (DDN_MAX_TAPS + 1)));
        }
        HIP_TRY(hipMemcpy(b->d_taps_s[k - 1], b->taps_s[k - 1], sizeof(float) * (size_t)len, hipMemcpyHostToDevice));
    }
    b->n_seg = n_seg;
    b->seg_first[0] = 0;
    b->seg_first[1] = n_seg > 1 ? seg_channels[0] : b->cfg.n_channels;
    b->seg_first[2] = n_seg > 2 ? seg_channels[0] + seg_channels[1] : b->cfg.n_channels;
    return DDN_OK;
}

extern "C" int
ddn_front_end_run_segments(ddn_batch* b, const void* const* d_iq, size_t n, float* const* d_disc, void* hip_stream) {
    if (!b || !d_iq || !d_disc || b->n_seg < 1) {
        ddn_set_error("ddn_front_end_run_segments: null argument or no segments set");
        return DDN_EINVAL;
    }
    for (int k = 0; k < b->n_seg; k++) {
        if (!d_iq[k] || !d_disc[k]) {
            ddn_set_error("ddn_front_end_run_segments: segment %d without arrays", k);
            return DDN_EINVAL;
        }
    }
    if (n == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int block_len = b->cfg.block_len;
    const long n_blocks = (long)((n + (size_t)block_len - 1) / (size_t)block_len);
    const int tiles_per_block = (block_len + DDN_TILE - 1) / DDN_TILE;
    DdnFusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.in = d_iq[0];
    fa.out = d_disc[0];
    fa.carry = b->d_carry;
    // (the kernel refreshes the carried look-back itself; a call shorter than the look-back has no segmented carry kernel)
    if (n < (size_t)DDN_CARRY_LEN) {
        ddn_set_error("ddn_front_end_run_segments: n = %zu < %d samples", n, DDN_CARRY_LEN);
        return DDN_ERANGE;
    }
    fa.carry_out = (ddn_f2*)b->d_carry;
    fa.state = b->d_state;
    fa.taps_dev = b->d_taps;
    fa.ch_stride = n;
    fa.out_stride = n;
    fa.n = (long)n;
    fa.n_tiles = n_blocks * tiles_per_block;
    fa.n_channels = b->cfg.n_channels;
    fa.in_fmt = b->cfg.input_format;
    fa.block_len = block_len;
    fa.tiles_per_block = tiles_per_block;
    fa.center = b->center;
    fa.n_seg = b->n_seg;
    fa.seg_first1 = b->seg_first[1];
    fa.seg_first2 = b->seg_first[2];
    fa.seg_in0 = d_iq[0];
    fa.seg_out0 = d_disc[0];
    fa.seg_in1 = b->n_seg > 1 ? d_iq[1] : nullptr;
    fa.seg_out1 = b->n_seg > 1 ? d_disc[1] : nullptr;
    fa.seg_in2 = b->n_seg > 2 ? d_iq[2] : nullptr;
    fa.seg_out2 = b->n_seg > 2 ? d_disc[2] : nullptr;
    fa.seg_taps1 = b->d_taps_s[0];
    fa.seg_taps2 = b->d_taps_s[1];
    bool has_zero = taps_have_zero(b->taps, b->taps_len);
    for (int k = 1; k < b->n_seg; k++) {
        has_zero = has_zero || taps_have_zero(b->taps_s[k - 1], b->taps_len);
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[0], st));
    }
    HIP_TRY(ddn_dev_launch_fused_ex(&fa, has_zero, b->group ? b->group : 16, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[1], st));
        HIP_TRY(hipEventRecord(b->ev[2], st));
        b->ev_valid = 1;
    }
    return DDN_OK;
}

extern "C" int
ddn_front_end_run(ddn_batch* b, const void* d_iq, size_t n, float* d_disc, void* hip_stream) {
    if (!b || !d_iq || !d_disc) {
        ddn_set_error("ddn_front_end_run: null argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int B = b->cfg.n_channels;
    int block_len = b->cfg.block_len;
    int in_fmt = b->cfg.input_format;
    if (b->passes > 0) {
        // half-band cascade: 31-tap first stage, 15-tap afterwards, each stage on the block partition it inherits
        if ((n & (((size_t)1 << b->passes) - 1)) != 0) {
            ddn_set_error("ddn_front_end_run: n = %zu must be a multiple of %d with %d decimation passes", n,
                          1 << b->passes, b->passes);
            return DDN_ERANGE;
        }
        for (int i = 0; i < b->passes; i++) {
            const size_t n_in = n >> i, n_o = n_in >> 1;
            const int w = i & 1;
            const size_t need = sizeof(ddn_f2) * (size_t)B * n_o;
            if (b->dec_cap[w] < need) {
                HIP_TRY(hipStreamSynchronize(st));
                (void)hipFree(b->d_dec[w]);
                b->d_dec[w] = nullptr;
                b->dec_cap[w] = 0;
                HIP_TRY(hipMalloc(&b->d_dec[w], need));
                b->dec_cap[w] = need;
            }
            HIP_TRY(ddn_dev_hb_decim2(d_iq, in_fmt, (long)n_in, n_in, block_len, B, i == 0 ? 31 : 15, b->d_hbhist[i],
                                      b->d_dec[w], n_o, st));
            d_iq = b->d_dec[w];
            in_fmt = DDN_IN_CF32;
            block_len >>= 1;
        }
        n >>= b->passes;
    }
    if (b->iqc.dc_enable || b->iqc.bal_enable) {
        // IQ conditioning on: the stages sit between the channel LPF and the discriminator, so the LPF result goes
        // through HBM once (8 B/sample each way) and the recurrences run one lane per channel (ddn_iqcond.hip)
        const size_t need = sizeof(ddn_f2) * (size_t)B * n;
        if (b->lpf_cap < need) {
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(b->d_lpf);
            b->d_lpf = nullptr;
            b->lpf_cap = 0;
            HIP_TRY(hipMalloc(&b->d_lpf, need));
            b->lpf_cap = need;
        }
        DdnIqCondConfig c = b->iqc;
        c.squelch_on = b->cfg.squelch_level > 0.0f ? 1 : 0;
        c.squelch_level = b->cfg.squelch_level;
        if (b->timing) {
            HIP_TRY(hipEventRecord(b->ev[0], st));
        }
        HIP_TRY(ddn_dev_channel_lpf_c2c(d_iq, in_fmt, (long)n, n, block_len, B, b->d_taps, b->taps_len, taps_have_zero(b->taps, b->taps_len), b->d_lpf_hist,
                                        b->d_lpf, n, st));
        if (b->timing) {
            HIP_TRY(hipEventRecord(b->ev[1], st));
        }
        HIP_TRY(ddn_dev_iq_cond_disc(b->d_lpf, (long)n, n, block_len, B, &c, b->d_state, b->d_iqstate, d_disc, n, st));
        if (b->timing) {
            HIP_TRY(hipEventRecord(b->ev[2], st));
            b->ev_valid = 1;
        }
        return DDN_OK;
    }
    const long n_blocks = (long)((n + (size_t)block_len - 1) / (size_t)block_len);
    const int tiles_per_block = (block_len + DDN_TILE - 1) / DDN_TILE;

    DdnFusedArgs fa;
    fa.in = d_iq;
    fa.out = d_disc;
    fa.carry = b->d_carry;
    fa.carry_out = (n >= (size_t)DDN_CARRY_LEN && !DDN_EXP_ENV("DDN_CARRY_KERNEL")) ? (ddn_f2*)b->d_carry : nullptr;
    fa.state = b->d_state;
    fa.taps_dev = b->d_taps;
    fa.ch_stride = n;
    fa.out_stride = n;
    fa.n = (long)n;
    fa.n_tiles = n_blocks * tiles_per_block;
    fa.n_channels = B;
    fa.in_fmt = in_fmt;
    fa.block_len = block_len;
    fa.tiles_per_block = tiles_per_block;
    fa.center = b->center;
    fa.squelch_on = b->cfg.squelch_level > 0.0f ? 1 : 0;
    fa.squelch_level = b->cfg.squelch_level;
    fa.n_seg = 0;
    fa.seg_first1 = fa.seg_first2 = 0;
    fa.seg_in0 = fa.seg_in1 = fa.seg_in2 = nullptr;
    fa.seg_out0 = fa.seg_out1 = fa.seg_out2 = nullptr;
    fa.seg_taps1 = fa.seg_taps2 = nullptr;
    {
        const char* d = DDN_EXP_ENV("DDN_DBG");
        fa.dbg = d ? atoi(d) : 0;
        fa.dbg_out = nullptr;
        if (fa.dbg & 64) {
            static long long* dbg_buf = nullptr;
            if (!dbg_buf) {
                (void)hipMalloc(&dbg_buf, 64 * 4 * sizeof(long long));
                (void)hipMemset(dbg_buf, 0, 64 * 4 * sizeof(long long)); // (rows 8, 9 stay zero without the extra filter waves)
            }
            fa.dbg_out = dbg_buf;
        }
    }

    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[0], st));
    }
    HIP_TRY(ddn_dev_launch_fused(&fa, b->taps, b->group, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[1], st));
    }
    if (!fa.carry_out) {
        HIP_TRY(ddn_dev_launch_carry(d_iq, in_fmt, n, (long)n, b->d_carry, B, st));
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[2], st));
        b->ev_valid = 1;
    }
    if (fa.dbg_out && DDN_EXP_ENV("DDN_DBG_PRINT")) {
        long long h[64 * 4];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, fa.dbg_out, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 12; w++) { // 0-9 filter waves (8, 9: the two that only take work items), 10 = dc wave S1, 11 = peak wave S2
            fprintf(stderr, "wave %d: simd %lld  phaseA %lld  phaseB %lld  barrier-wait %lld cycles/tile\n", w, h[w * 4],
                    h[w * 4 + 1] / (long long)fa.n_tiles, h[w * 4 + 2] / (long long)fa.n_tiles,
                    h[w * 4 + 3] / (long long)fa.n_tiles);
        }
    }
    return DDN_OK;
}

extern "C" int
ddn_front_end_run_host(ddn_batch* b, const void* h_iq, size_t n, float* h_disc) {
    if (!b || !h_iq || !h_disc) {
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    const size_t in_bytes = B * n * (b->cfg.input_format == DDN_IN_CU8 ? 2 : 8);
    const size_t out_bytes = B * (n >> b->passes) * sizeof(float);
    int rc = grow(&b->d_in, &b->in_cap, in_bytes);
    if (rc != DDN_OK) {
        return rc;
    }
    rc = grow((void**)&b->d_out, &b->out_cap, out_bytes);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(hipMemcpy(b->d_in, h_iq, in_bytes, hipMemcpyHostToDevice));
    rc = ddn_front_end_run(b, b->d_in, n, b->d_out, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(hipMemcpy(h_disc, b->d_out, out_bytes, hipMemcpyDeviceToHost));
    return DDN_OK;
}

extern "C" int
ddn_batch_get_fsk_state(ddn_batch* b, int channel, float out5[5]) {
    if (!b || !out5 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    DdnFskState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    out5[0] = s.prev_i;
    out5[1] = s.prev_q;
    out5[2] = (float)s.have_prev;
    out5[3] = s.dc_est;
    out5[4] = s.peak_est;
    return DDN_OK;
}

extern "C" int
ddn_batch_set_timing(ddn_batch* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->timing = enable ? 1 : 0;
    b->ev_valid = 0;
    return DDN_OK;
}

extern "C" int
ddn_batch_get_timing(ddn_batch* b, float out3[3]) {
    if (!b || !out3 || !b->ev_valid) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipEventSynchronize(b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&out3[0], b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&out3[1], b->ev[1], b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&out3[2], b->ev[0], b->ev[2]));
    return DDN_OK;
}
