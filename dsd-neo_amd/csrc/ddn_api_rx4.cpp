// ddn_api_rx4.cpp - C-ABI of the batched DMR / NXDN48 receive loop (include/ddn_fsk4.h, kernels ddn_rx4.hip).
// The profile a batch runs (sync words, matched filter, window / slip rules, what an accepted sync does) is built here from
// {protocol, rf_mod, inverted}; per-channel decoder words live on the device inside the batch object and persist across calls.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ddn_device.h"
#include "ddn_fsk4.h"
#include "ddn_tables_fsk4.h"

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

namespace {
// sync words as sign strings ('1' = +3, '3' = -3): ETSI TS 102 361-1 table 9.2 (BS / MS / direct-mode sourced data and voice)
// and the NXDN frame sync word with the four single-symbol variants the reference also accepts
// (include/dsd-neo/core/sync_patterns.h:58-80, src/dsp/dsd_frame_sync.c:1508-1512)
struct Pat {
    const char* s;
    int type, cls;
};
// type ids only need to be non-zero and distinct per family; these mirror synctype_ids.h + 1
enum { T_BS_DATA = 11, T_BS_VOICE_NEG = 12, T_BS_VOICE = 13, T_BS_DATA_NEG = 14, T_MS_VOICE = 33, T_MS_DATA = 34, T_NX_POS = 29, T_NX_NEG = 30 };
const Pat kDmr[8] = {{"313333111331131131331131", T_BS_DATA, 0}, {"131111333113313313113313", T_BS_VOICE, 1},
                     {"311131133313133331131113", T_MS_DATA, 0}, {"133313311131311113313331", T_MS_VOICE, 1},
                     {"331333313111313133311111", T_MS_DATA, 0}, {"311311111333113333133311", T_MS_DATA, 0},
                     {"113111131333131311133333", T_MS_VOICE, 1}, {"133133333111331111311133", T_MS_VOICE, 1}};
const char* const kNxPos[5] = {"3131331131", "3331331131", "3131331111", "3331331111", "3131311131"};
const char* const kNxNeg[5] = {"1313113313", "1113113313", "1313113333", "1113113333", "1313133313"};

uint32_t
sign_bits(const char* s) {
    uint32_t v = 0;
    for (; *s; s++) {
        v = (v << 1) | (*s == '1' ? 1u : 0u);
    }
    return v;
}
} // namespace

struct ddn_fsk4_rx {
    ddn_fsk4_rx_config cfg;
    DdnFsk4Config dc;
    DdnFsk4State* d_state;
    float *d_lbuf, *d_shist, *d_fhist, *d_fstale, *d_filt, *d_taps;
    DdnFsk4Config* d_cfg;
    uint8_t *d_phist, *d_rhist;
    int32_t* d_lock;
    size_t filt_cap;
    int channels_per_wave;
    int dbg_flags; // ddn_fsk4_rx_set_debug_flags
    // handler mode (ddn_fsk4_rx_set_handlers): the handlers' words [B][32] i32, the burst's dibits [B][144], event buffers
    int32_t* d_hwords;
    uint8_t* d_hpay;
    int32_t *d_events, *d_n_events;
    size_t max_events;
    int32_t *d_ev_own, *d_nev_own; // event buffers of the host convenience calls (ddn_fsk4_rx_events_host_*)
    bool timing;
    hipEvent_t ev[3];
};

static void
rx4_free(ddn_fsk4_rx* b) {
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_lbuf);
    (void)hipFree(b->d_shist);
    (void)hipFree(b->d_phist);
    (void)hipFree(b->d_rhist);
    (void)hipFree(b->d_fhist);
    (void)hipFree(b->d_fstale);
    (void)hipFree(b->d_filt);
    (void)hipFree(b->d_taps);
    (void)hipFree(b->d_cfg);
    (void)hipFree(b->d_lock);
    (void)hipFree(b->d_hwords);
    (void)hipFree(b->d_hpay);
    (void)hipFree(b->d_ev_own);
    (void)hipFree(b->d_nev_own);
    for (int i = 0; i < 3; i++) {
        if (b->ev[i]) {
            (void)hipEventDestroy(b->ev[i]);
        }
    }
}

static int
rx4_fill(ddn_fsk4_rx* b) {
    const size_t B = (size_t)b->cfg.n_channels;
    DdnFsk4State s0;
    memset(&s0, 0, sizeof(s0));
    s0.jitter = -1;
    // symbol_reset_rtl_fsk_timing_if_needed() + symbol_reset_rtl_fsk_discriminator_slicer() (dsd_symbol.c:1306-1341)
    s0.center = 0.0f;
    s0.min = -30000.0f;
    s0.max = 30000.0f;
    s0.lmid = -20000.0f;
    s0.umid = 20000.0f;
    s0.minref = -24000.0f;
    s0.maxref = 24000.0f;
    s0.lmin = s0.min;
    s0.lmax = s0.max;
    std::vector<DdnFsk4State> st(B, s0);
    HIP_TRY(hipMemcpy(b->d_state, st.data(), sizeof(DdnFsk4State) * B, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b->d_lbuf, 0, sizeof(float) * 24 * B));
    HIP_TRY(hipMemset(b->d_shist, 0, sizeof(float) * DDN_FSK4_HIST * B));
    HIP_TRY(hipMemset(b->d_phist, 0, DDN_FSK4_HIST * B));
    HIP_TRY(hipMemset(b->d_rhist, 0, DDN_FSK4_HIST * B));
    HIP_TRY(hipMemset(b->d_fhist, 0, sizeof(float) * (DDN_FSK4_MAX_TAPS - 1) * B));
    HIP_TRY(hipMemset(b->d_fstale, 0, sizeof(float) * (DDN_FSK4_MAX_TAPS - 1) * B));
    {   // dmr_confidence_reset() + state->dmr_color_code = 16 (dsd_init.c:1099): field order of ddn_fsk4h_dev.h
        std::vector<int32_t> hw(32 * B, 0);
        for (size_t c = 0; c < B; c++) {
            hw[32 * c + 1] = 16;  // F_CONF_CC
            hw[32 * c + 2] = 16;  // F_CAND_CC
            hw[32 * c + 11] = 16; // F_COLOR
        }
        HIP_TRY(hipMemcpy(b->d_hwords, hw.data(), sizeof(int32_t) * 32 * B, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(b->d_hpay, 0, 144 * B));
    }
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_create(const ddn_fsk4_rx_config* cfg, ddn_fsk4_rx** out) {
    if (!cfg || !out) {
        return DDN_EINVAL;
    }
    *out = nullptr;
    if (cfg->n_channels <= 0 || cfg->out_rate_hz <= 0
        || (cfg->protocol != DDN_FSK4_DMR && cfg->protocol != DDN_FSK4_NXDN48 && cfg->protocol != DDN_FSK4_NXDN96 && cfg->protocol != DDN_FSK4_M17
            && cfg->protocol != DDN_FSK4_YSF)
        || (cfg->rf_mod != 0 && cfg->rf_mod != 2) || (cfg->inverted && cfg->protocol != DDN_FSK4_DMR)) {
        ddn_set_error("ddn_fsk4_rx_create: bad configuration (protocol DMR | NXDN48 | NXDN96 | M17 | YSF, rf_mod 0 | 2, inverted only for DMR)");
        return DDN_EINVAL;
    }
    {
        const int sym_rate = cfg->protocol == DDN_FSK4_NXDN48 ? 2400 : 4800;
        if (cfg->out_rate_hz / sym_rate < 8 || cfg->out_rate_hz / sym_rate > 21) {
            // the kernel's per-round hand-off queue and its whole-symbol pass are sized for 8..21 samples per symbol
            ddn_set_error("ddn_fsk4_rx_create: out_rate_hz / symbol rate must be within 8..21 (48 ksps: DMR 10, NXDN48 20)");
            return DDN_ERANGE;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("ddn_fsk4_rx_create: no HIP device");
        return DDN_ENODEV;
    }
    ddn_fsk4_rx* b = new (std::nothrow) ddn_fsk4_rx();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->cfg = *cfg;
    DdnFsk4Config& d = b->dc;
    d.out_rate = cfg->out_rate_hz;
    d.rf_mod = cfg->rf_mod;
    d.use_filter = cfg->use_matched_filter ? 1 : 0;
    const unsigned int* tap_bits;
    int lock_default[4] = {0, 0, 0, 0};
    if (cfg->protocol == DDN_FSK4_M17) {
        // -fz: C4FM lock at 4800 symbols/s, NO matched filter (decode_mode_apply_m17(), src/runtime/decode_mode.c:486-510), the
        // eight-symbol words matched by frame_sync_try_m17()'s rules inside the kernel (m17_hit, ddn_rx4.hip); the table names the
        // twelve outcomes: type ids = synctype_ids.h:52-63 + 1, odd rows negative polarity, class 1 = the preamble (skipDibit(8)),
        // class 0 = a frame or the EOT marker (184 dibits) - dispatch_m17.c:25-68
        static const uint8_t kM17Types[12] = {99, 100, 101, 102, 17, 18, 77, 78, 9, 10, 87, 88};
        d.sym_rate = 4800;
        d.win_len = 8;
        d.t_max = 24;
        d.warm_len = 8;
        d.n_pat = 12;
        d.use_filter = 0;
        for (int k = 0; k < 12; k++) {
            d.pat_bits[k] = 0xFFu;
            d.pat_type[k] = kM17Types[k];
            d.pat_neg[k] = (uint8_t)(k & 1);
            d.pat_class[k] = (uint8_t)(k < 2 ? 1 : 0);
        }
        d.nt = DDN_DMR_FILTER_TAPS; // (unused)
        tap_bits = ddn_dmr_filter_bits;
        lock_default[0] = 184;
        lock_default[1] = 8;
    } else if (cfg->protocol == DDN_FSK4_YSF) {
        // -fy: the 20-symbol FUSION_SYNC exact in both polarities (include/dsd-neo/core/sync_patterns.h:30-31; types = synctype_ids.h:
        // 109-110 + 1), the DMR matched filter once a YSF sync is the last type, 100 FICH + 360 payload dibits behind a sync
        static const char kYsf[] = "31111311313113131131";
        char inv[21];
        for (int k = 0; k < 20; k++) {
            inv[k] = kYsf[k] == '3' ? '1' : '3';
        }
        inv[20] = 0;
        d.sym_rate = 4800;
        d.win_len = 20;
        d.t_max = 24;
        d.warm_len = 20;
        d.n_pat = 2;
        d.pat_bits[0] = sign_bits(kYsf);
        d.pat_type[0] = 31;
        d.pat_bits[1] = sign_bits(inv);
        d.pat_type[1] = 32;
        d.pat_neg[1] = 1;
        d.nt = DDN_DMR_FILTER_TAPS;
        tap_bits = ddn_dmr_filter_bits;
        lock_default[0] = 460;
    } else if (cfg->protocol == DDN_FSK4_DMR) {
        d.sym_rate = 4800;
        d.win_len = d.t_max = d.warm_len = 24;
        d.dmr_window = 1;
        d.redigitize = 1;
        d.n_pat = 8;
        for (int k = 0; k < 8; k++) {
            d.pat_bits[k] = sign_bits(kDmr[k].s);
            int type = kDmr[k].type, cls = kDmr[k].cls, neg = 0;
            if (cfg->inverted) { // frame_sync_try_dmr_*: the data word marks an inverted voice burst and vice versa
                cls ^= 1;
                if (type == T_BS_DATA) {
                    type = T_BS_VOICE_NEG;
                    neg = 1;
                } else if (type == T_BS_VOICE) {
                    type = T_BS_DATA_NEG;
                    neg = 1;
                } else {
                    type = (type == T_MS_DATA) ? T_MS_VOICE : T_MS_DATA; // MS types carry no digitize() polarity
                }
            }
            d.pat_type[k] = (uint8_t)type;
            d.pat_class[k] = (uint8_t)cls;
            d.pat_neg[k] = (uint8_t)neg;
        }
        d.nt = DDN_DMR_FILTER_TAPS;
        tap_bits = ddn_dmr_filter_bits;
        lock_default[0] = 120;
        lock_default[1] = 54 + 6 * 288;
    } else {
        const bool n96 = cfg->protocol == DDN_FSK4_NXDN96;
        d.sym_rate = n96 ? 4800 : 2400;
        d.win_len = 10;
        d.t_max = n96 ? 24 : 12;
        d.warm_len = 10;
        d.confirm = 1;
        d.n_pat = 10;
        for (int k = 0; k < 5; k++) {
            d.pat_bits[k] = sign_bits(kNxPos[k]);
            d.pat_type[k] = T_NX_POS;
            d.pat_bits[5 + k] = sign_bits(kNxNeg[k]);
            d.pat_type[5 + k] = T_NX_NEG;
            d.pat_neg[5 + k] = 1;
        }
        d.nt = n96 ? DDN_DMR_FILTER_TAPS : DDN_NXDN48_FILTER_TAPS;
        tap_bits = n96 ? ddn_dmr_filter_bits : ddn_nxdn48_filter_bits;
        lock_default[0] = 182;
    }
    bool all_zero = true;
    for (int k = 0; k < 4; k++) {
        all_zero = all_zero && cfg->lock_symbols[k] == 0;
    }
    for (int k = 0; k < 4; k++) {
        b->cfg.lock_symbols[k] = all_zero ? lock_default[k] : cfg->lock_symbols[k];
    }
    const size_t B = (size_t)cfg->n_channels;
    // fewer channels per wavefront = fewer unsynchronised recurrences sharing one instruction stream, and a trip is only a
    // lean trip when every lane of the wave is inside a frame (measured at 4096 DMR channels: 11.8 ms with 4 lanes per wave,
    // 13.6 with 8, 16.7 with 16); two-wave workgroups, so 4096 channels at 4 per wave are eight wavefronts per CU
    // (round 3, with the bulk hunting pass: at 1365 channels 9.5 / 7.6 / 6.2 ms DMR and 6.5 / 4.0 / 2.6 ms NXDN48 with 4 / 2 / 1
    // lanes per wave - as long as all the workgroups are resident at once: 2048 one-channel workgroups run in two rounds)
    b->channels_per_wave = 32;
    for (int cpw = 1; cpw <= 32; cpw *= 2) {
        if ((B + cpw - 1) / cpw <= 1536) {
            b->channels_per_wave = cpw;
            break;
        }
    }
    if (const char* e = DDN_EXP_ENV("DDN_RX4_CPW")) { // (experiments; the same values ddn_fsk4_rx_set_channels_per_wave accepts)
        const int v = atoi(e);
        if (v >= 1 && v <= 32 && (v & (v - 1)) == 0) {
            b->channels_per_wave = v;
        }
    }
    std::vector<int32_t> lock(B * 4);
    for (size_t c = 0; c < B; c++) {
        for (int k = 0; k < 4; k++) {
            lock[c * 4 + k] = b->cfg.lock_symbols[k];
        }
    }
    float taps[DDN_FSK4_MAX_TAPS] = {0};
    memcpy(taps, tap_bits, sizeof(float) * (size_t)d.nt);
    if (hipMalloc(&b->d_state, sizeof(DdnFsk4State) * B) != hipSuccess || hipMalloc(&b->d_lbuf, sizeof(float) * 24 * B) != hipSuccess
        || hipMalloc(&b->d_shist, sizeof(float) * DDN_FSK4_HIST * B) != hipSuccess
        || hipMalloc(&b->d_phist, DDN_FSK4_HIST * B) != hipSuccess || hipMalloc(&b->d_rhist, DDN_FSK4_HIST * B) != hipSuccess
        || hipMalloc(&b->d_fhist, sizeof(float) * (DDN_FSK4_MAX_TAPS - 1) * B) != hipSuccess
        || hipMalloc(&b->d_fstale, sizeof(float) * (DDN_FSK4_MAX_TAPS - 1) * B) != hipSuccess
        || hipMalloc(&b->d_taps, sizeof(taps)) != hipSuccess || hipMalloc(&b->d_cfg, sizeof(DdnFsk4Config)) != hipSuccess
        || hipMemcpy(b->d_cfg, &b->dc, sizeof(DdnFsk4Config), hipMemcpyHostToDevice) != hipSuccess || hipMalloc(&b->d_lock, sizeof(int32_t) * 4 * B) != hipSuccess
        || hipMalloc(&b->d_hwords, sizeof(int32_t) * 32 * B) != hipSuccess || hipMalloc(&b->d_hpay, 144 * B) != hipSuccess
        || hipMemcpy(b->d_taps, taps, sizeof(taps), hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(b->d_lock, lock.data(), sizeof(int32_t) * 4 * B, hipMemcpyHostToDevice) != hipSuccess) {
        ddn_set_error("ddn_fsk4_rx_create: device allocation failed");
        rx4_free(b);
        delete b;
        return DDN_ENOMEM;
    }
    const int rc = rx4_fill(b);
    if (rc != DDN_OK) {
        rx4_free(b);
        delete b;
        return rc;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_fsk4_rx_destroy(ddn_fsk4_rx* b) {
    if (b) {
        rx4_free(b);
        delete b;
    }
}

extern "C" int
ddn_fsk4_rx_set_handlers(ddn_fsk4_rx* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    if (enable && b->cfg.inverted) {
        ddn_set_error("ddn_fsk4_rx_set_handlers: the handlers are the reference's plain -fs ones (inverted = 0)");
        return DDN_EINVAL;
    }
    b->dc.handlers = enable ? 1 : 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b->d_cfg, &b->dc, sizeof(DdnFsk4Config), hipMemcpyHostToDevice));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_set_events(ddn_fsk4_rx* b, int32_t* d_events, int32_t* d_n_events, size_t max_events) {
    if (!b || ((d_events == nullptr) != (d_n_events == nullptr)) || (d_events && max_events == 0) || max_events > 0x7FFFFFFF) {
        return DDN_EINVAL;
    }
    b->d_events = d_events;
    b->d_n_events = d_n_events;
    b->max_events = d_events ? max_events : 0;
    b->dc.max_events = (int)b->max_events;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b->d_cfg, &b->dc, sizeof(DdnFsk4Config), hipMemcpyHostToDevice));
    return DDN_OK;
}

// host convenience: event buffers owned by the batch object, read back after a run
extern "C" int
ddn_fsk4_rx_events_host_arm(ddn_fsk4_rx* b, size_t max_events) {
    if (!b || max_events == 0 || max_events > 0x7FFFFFFF) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(b->d_ev_own);
    (void)hipFree(b->d_nev_own);
    b->d_ev_own = b->d_nev_own = nullptr;
    HIP_TRY(hipMalloc(&b->d_ev_own, B * max_events * 16));
    HIP_TRY(hipMalloc(&b->d_nev_own, B * 4));
    HIP_TRY(hipMemset(b->d_ev_own, 0, B * max_events * 16));
    HIP_TRY(hipMemset(b->d_nev_own, 0, B * 4));
    return ddn_fsk4_rx_set_events(b, b->d_ev_own, b->d_nev_own, max_events);
}

extern "C" int
ddn_fsk4_rx_events_host_read(ddn_fsk4_rx* b, int32_t* events, int32_t* n_events) {
    if (!b || !events || !n_events || !b->d_ev_own || b->d_events != b->d_ev_own) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(events, b->d_ev_own, B * b->max_events * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(n_events, b->d_nev_own, B * 4, hipMemcpyDeviceToHost));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_reset(ddn_fsk4_rx* b) {
    if (!b) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    return rx4_fill(b);
}

extern "C" size_t
ddn_fsk4_rx_max_symbols(const ddn_fsk4_rx* b, size_t n) {
    if (!b) {
        return 0;
    }
    int whole = b->dc.out_rate / b->dc.sym_rate;
    whole = whole < 2 ? 2 : (whole > 64 ? 64 : whole);
    return n / (size_t)(whole - 1) + 2;
}

extern "C" size_t
ddn_fsk4_rx_max_syncs(const ddn_fsk4_rx* b, size_t n) {
    if (!b) {
        return 0;
    }
    // two accepted syncs are at least one sync window apart (the hunt restarts with an empty window)
    return ddn_fsk4_rx_max_symbols(b, n) / (size_t)b->dc.win_len + 2;
}

extern "C" int
ddn_fsk4_rx_set_lock_symbols(ddn_fsk4_rx* b, const int32_t* lock4) {
    if (!b || !lock4) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b->d_lock, lock4, sizeof(int32_t) * 4 * (size_t)b->cfg.n_channels, hipMemcpyHostToDevice));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_run(ddn_fsk4_rx* b, const float* d_disc, size_t n, uint8_t* d_records10, uint8_t* d_flags, uint8_t* d_payload2,
                int32_t* d_counts, size_t max_symbols, int32_t* d_sync_pos, uint8_t* d_sync_pat, uint8_t* d_pre,
                uint8_t* d_pre_rel, int32_t* d_n_sync, size_t max_syncs, void* hip_stream) {
    if (!b || !d_disc || !d_records10 || !d_flags || !d_payload2 || !d_counts || !d_sync_pos || !d_sync_pat || !d_pre || !d_pre_rel
        || !d_n_sync) {
        ddn_set_error("ddn_fsk4_rx_run: null argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int B = b->cfg.n_channels;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)B, st));
        HIP_TRY(hipMemsetAsync(d_n_sync, 0, sizeof(int32_t) * (size_t)B, st));
        return DDN_OK;
    }
    if (max_symbols < ddn_fsk4_rx_max_symbols(b, n) || max_syncs < ddn_fsk4_rx_max_syncs(b, n)) {
        ddn_set_error("ddn_fsk4_rx_run: max_symbols %zu / max_syncs %zu below ddn_fsk4_rx_max_symbols / _max_syncs (%zu / %zu)",
                      max_symbols, max_syncs, ddn_fsk4_rx_max_symbols(b, n), ddn_fsk4_rx_max_syncs(b, n));
        return DDN_ERANGE;
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[0], st));
    }
    if (b->dc.use_filter) {
        if (b->filt_cap < n) {
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(b->d_filt);
            b->d_filt = nullptr;
            b->filt_cap = 0;
            HIP_TRY(hipMalloc(&b->d_filt, sizeof(float) * (size_t)B * n));
            b->filt_cap = n;
        }
        HIP_TRY(ddn_dev_fsk4_matched_filter(b->dc.nt, d_disc, (long)n, n, B, b->d_fhist, b->d_filt, st));
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[1], st));
    }
    {
        int want = b->dbg_flags;
        if (const char* e = DDN_EXP_ENV("DDN_RX4_DBG")) {
            want = atoi(e);
        }
        if (b->dc.dbg != want) {
            b->dc.dbg = want;
            HIP_TRY(hipMemcpy(b->d_cfg, &b->dc, sizeof(DdnFsk4Config), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(ddn_dev_fsk4_rx(d_disc, b->d_filt, b->d_fhist, b->d_fstale, b->d_taps, (long)n, n, B, b->d_cfg, b->d_state, b->d_lbuf,
                            b->d_shist, b->d_phist, b->d_rhist, d_records10, d_flags, d_payload2, d_counts, max_symbols, b->d_lock,
                            d_sync_pos, d_sync_pat, d_pre, d_pre_rel, d_n_sync, (int)max_syncs, b->channels_per_wave,
                            b->dc.out_rate / b->dc.sym_rate + (b->dc.out_rate % b->dc.sym_rate ? 1 : 0), b->cfg.protocol,
                            b->dc.handlers, b->d_hwords, b->d_hpay, b->d_events, b->d_n_events, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[2], st));
    }
    HIP_TRY(ddn_dev_fsk4_filter_hist_update(b->dc.nt, d_disc, (long)n, n, B, b->d_fhist, st));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_set_channels_per_wave(ddn_fsk4_rx* b, int channels_per_wave) {
    if (!b || channels_per_wave < 1 || channels_per_wave > 32 || (channels_per_wave & (channels_per_wave - 1))) {
        return DDN_EINVAL;
    }
    b->channels_per_wave = channels_per_wave;
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_set_debug_flags(ddn_fsk4_rx* b, int flags) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->dbg_flags = flags;
    return DDN_OK;
}

// per-sync thresholds {center, umid, lmid, max, min} [B][max_syncs][5] (device memory, NULL = off): written by the loop's helper wave
// beside sync_pos / sync_pat; max_syncs = the value the run calls are given
extern "C" int
ddn_fsk4_rx_set_sync_thresholds(ddn_fsk4_rx* b, float* d_thr5) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->dc.sync_thr = d_thr5;
    HIP_TRY(hipMemcpy(b->d_cfg, &b->dc, sizeof(DdnFsk4Config), hipMemcpyHostToDevice));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_set_timing(ddn_fsk4_rx* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    if (enable && !b->ev[0]) {
        for (int i = 0; i < 3; i++) {
            HIP_TRY(hipEventCreate(&b->ev[i]));
        }
    }
    b->timing = enable != 0;
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_get_timing(ddn_fsk4_rx* b, float* ms2) {
    if (!b || !ms2 || !b->ev[0]) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipEventSynchronize(b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms2[0], b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms2[1], b->ev[1], b->ev[2]));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_get_thresholds(ddn_fsk4_rx* b, int channel, float out7[7]) {
    if (!b || !out7 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    DdnFsk4State s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    out7[0] = s.center;
    out7[1] = s.umid;
    out7[2] = s.lmid;
    out7[3] = s.max;
    out7[4] = s.min;
    out7[5] = s.maxref;
    out7[6] = s.minref;
    return DDN_OK;
}

extern "C" int
ddn_fsk4_rx_run_host(ddn_fsk4_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags, uint8_t* payload2,
                     int32_t* counts, size_t max_symbols, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre, uint8_t* pre_rel,
                     int32_t* n_sync, size_t max_syncs) {
    if (!b || !disc || !records10 || !flags || !payload2 || !counts || !sync_pos || !sync_pat || !pre || !pre_rel || !n_sync) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels, S = B * max_symbols, Y = B * max_syncs;
    float* d_in = nullptr;
    uint8_t *d_rec = nullptr, *d_fl = nullptr, *d_pay = nullptr, *d_spat = nullptr, *d_pre = nullptr, *d_prel = nullptr;
    int32_t *d_cnt = nullptr, *d_spos = nullptr, *d_ns = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_in, B * n * 4 + 4) != hipSuccess || hipMalloc(&d_rec, S * 10 + 4) != hipSuccess
        || hipMalloc(&d_fl, S + 4) != hipSuccess || hipMalloc(&d_pay, S * 2 + 4) != hipSuccess
        || hipMalloc(&d_cnt, B * 4) != hipSuccess || hipMalloc(&d_spos, Y * 4 + 4) != hipSuccess
        || hipMalloc(&d_spat, Y + 4) != hipSuccess || hipMalloc(&d_pre, Y * DDN_FSK4_PRE + 4) != hipSuccess
        || hipMalloc(&d_prel, Y * DDN_FSK4_PRE + 4) != hipSuccess || hipMalloc(&d_ns, B * 4) != hipSuccess) {
        ddn_set_error("ddn_fsk4_rx_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, disc, B * n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(d_rec, 0, S * 10) != hipSuccess
               || hipMemset(d_fl, 0, S) != hipSuccess || hipMemset(d_pay, 0, S * 2) != hipSuccess
               || hipMemset(d_spos, 0, Y * 4) != hipSuccess || hipMemset(d_spat, 0, Y) != hipSuccess
               || hipMemset(d_pre, 0, Y * DDN_FSK4_PRE) != hipSuccess || hipMemset(d_prel, 0, Y * DDN_FSK4_PRE) != hipSuccess) {
        rc = DDN_EHIP;
    } else {
        rc = ddn_fsk4_rx_run(b, d_in, n, d_rec, d_fl, d_pay, d_cnt, max_symbols, d_spos, d_spat, d_pre, d_prel, d_ns, max_syncs,
                             nullptr);
        if (rc == DDN_OK
            && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(records10, d_rec, S * 10, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(flags, d_fl, S, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(payload2, d_pay, S * 2, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(counts, d_cnt, B * 4, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(sync_pos, d_spos, Y * 4, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(sync_pat, d_spat, Y, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(pre, d_pre, Y * DDN_FSK4_PRE, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(pre_rel, d_prel, Y * DDN_FSK4_PRE, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(n_sync, d_ns, B * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
            ddn_set_error("ddn_fsk4_rx_run_host: %s", hipGetErrorString(hipGetLastError()));
            rc = DDN_EHIP;
        }
    }
    (void)hipFree(d_in);
    (void)hipFree(d_rec);
    (void)hipFree(d_fl);
    (void)hipFree(d_pay);
    (void)hipFree(d_cnt);
    (void)hipFree(d_spos);
    (void)hipFree(d_spat);
    (void)hipFree(d_pre);
    (void)hipFree(d_prel);
    (void)hipFree(d_ns);
    return rc;
}

extern "C" int
ddn_dmr_burst_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                     const uint8_t* d_pre, const int32_t* d_n_sync, int n_channels, size_t max_syncs, int inverted,
                     uint8_t* d_slot_type, uint8_t* d_info, uint8_t* d_cach, uint8_t* d_valid, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_pre || !d_n_sync || !d_slot_type || !d_info || !d_cach || !d_valid
        || n_channels <= 0) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_dmr_burst_gather(d_records10, d_counts, max_symbols, d_sync_pos, d_pre, d_n_sync, n_channels, (int)max_syncs,
                                     inverted, d_slot_type, d_info, d_cach, d_valid, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_nxdn_frame_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                      const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_lich, uint8_t* d_sacch_sym,
                      uint8_t* d_sacch_rel, uint8_t* d_facch_sym, uint8_t* d_facch_rel, uint8_t* d_valid, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_n_sync || !d_lich || !d_sacch_sym || !d_sacch_rel || !d_facch_sym || !d_facch_rel
        || !d_valid || n_channels <= 0) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_nxdn_frame_gather(d_records10, d_counts, max_symbols, d_sync_pos, d_n_sync, n_channels, (int)max_syncs, d_lich,
                                      d_sacch_sym, d_sacch_rel, d_facch_sym, d_facch_rel, d_valid, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_ambe2450_deinterleave_batch(const uint8_t* d_dibits36, const uint8_t* d_reliab36, size_t n, uint8_t* d_ambe_fr,
                                uint8_t* d_ambe_rel, void* hip_stream) {
    if (!d_dibits36 || !d_ambe_fr || (d_ambe_rel && !d_reliab36)) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_ambe2450_deinterleave(d_dibits36, d_reliab36, (int)n, d_ambe_fr, d_ambe_rel, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_nxdn_voice_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                      const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_ambe_fr, uint8_t* d_ambe_rel,
                      uint8_t* d_valid, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_sync_pos || !d_n_sync || !d_ambe_fr || n_channels <= 0) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_nxdn_voice_gather(d_records10, d_counts, max_symbols, d_sync_pos, d_n_sync, (int)max_syncs, n_channels,
                                      d_ambe_fr, d_ambe_rel, d_valid, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_dmr_voice_burst_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                           const int32_t* d_burst_start, int n_channels, size_t max_bursts, int inverted, uint8_t* d_ambe_fr,
                           uint8_t* d_ambe_rel, uint8_t* d_sync48, uint8_t* d_cach24, uint8_t* d_valid, void* hip_stream) {
    if (!d_records10 || !d_counts || !d_burst_start || !d_ambe_fr || n_channels <= 0) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_dmr_voice_gather(d_records10, d_counts, max_symbols, d_burst_start, (int)max_bursts, n_channels, inverted,
                                     d_ambe_fr, d_ambe_rel, d_sync48, d_cach24, d_valid, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_nxdn_crc_check_batch(const uint8_t* d_bytes, int stride, size_t n, int kind, uint8_t* d_ok, void* hip_stream) {
    if (!d_bytes || !d_ok || stride <= 0 || kind < 0 || kind > 3 || stride * ((kind & 2) ? 1 : 8) < ((kind & 1) == 0 ? 32 : 92)) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_nxdn_crc(d_bytes, stride, (int)n, kind, d_ok, (hipStream_t)hip_stream));
    return DDN_OK;
}
