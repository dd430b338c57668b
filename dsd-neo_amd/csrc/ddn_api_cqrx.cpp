// ddn_api_cqrx.cpp - C-ABI of the symbol-rate receive loop behind the CQPSK demodulator (ddn_cqrx.hip): batch object, carried
// per-channel state, the sync words under the rotation maps.  Host-only code.
//
// What it stands in for in a dsd-neo host: the consumer thread between the demodulator's symbol output and the capture records -
// getSymbol()'s symbol-rate fast path (src/dsp/dsd_symbol.c:1581-1624), getFrameSync() on a QPSK profile
// (src/dsp/dsd_frame_sync.c:3098-3148) and get_dibit_and_analog_signal() (src/core/frames/dsd_dibit.c:1043-1075) - B channels wide.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_device.h"
#include "ddn_hip.h"
#include "ddn_internal.h"

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

struct ddn_cq_rx {
    ddn_cq_rx_config cfg;
    DdnCqConfig dc;
    DdnCqState* d_state;
    int32_t *d_events, *d_n_events, *d_event_data;
};

static const uint8_t k_map[5][4] = {{0, 1, 2, 3}, {2, 3, 0, 1}, {3, 2, 1, 0}, {1, 3, 0, 2}, {2, 0, 3, 1}};

extern "C" int
ddn_cq_rx_create(const ddn_cq_rx_config* cfg, ddn_cq_rx** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || (cfg->protocol != DDN_CQ_P25P1 && cfg->protocol != DDN_CQ_P25P2)) {
        ddn_set_error("ddn_cq_rx_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    ddn_cq_rx* b = new (std::nothrow) ddn_cq_rx();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->cfg = *cfg;
    DdnCqConfig& dc = b->dc;
    const bool p2 = cfg->protocol == DDN_CQ_P25P2;
    dc.protocol = cfg->protocol;
    dc.sync_len = p2 ? 20 : 24;
    dc.t_max = p2 ? 19 : 24;
    dc.lock_symbols = p2 ? (cfg->lock_symbols > 0 ? cfg->lock_symbols : 700) : (cfg->lock_symbols == 0 ? -1 : cfg->lock_symbols);
    dc.nid_threshold = cfg->nid_erasure_threshold > 0 ? cfg->nid_erasure_threshold : 64;
    // apply_cqpsk_snr_weight(), src/core/frames/dsd_dibit.c:408-430
    const double snr = (double)cfg->snr_cqpsk_db;
    if (cfg->snr_cqpsk_db == 0.0f || snr <= -50.0) { // (0 = not given)
        dc.snr_scale = -1;
    } else {
        int w256 = 0;
        if (snr >= 25.0) {
            w256 = 255;
        } else if (snr > 0.0) {
            w256 = (int)((snr / 25.0) * 255.0 + 0.5);
        }
        dc.snr_scale = 204 + (w256 >> 2);
    }
    // include/dsd-neo/core/sync_patterns.h:33-37; a raw window matches under map m when it equals m's inverse image of the pattern
    static const char* pat[2][2] = {{"111113113311333313133333", "333331331133111131311111"}, {"11131131111333133333", "33313313333111311111"}};
    const int order[4] = {0, 2, 3, 4};
    for (int pol = 0; pol < 2; pol++) {
        for (int m = 0; m < 4; m++) {
            uint64_t w = 0;
            for (int i = 0; i < dc.sync_len; i++) {
                const int want = pat[p2 ? 1 : 0][pol][i] - '0';
                int raw = want;
                for (int q = 0; q < 4; q++) {
                    if (k_map[order[m]][q] == want) {
                        raw = q;
                    }
                }
                w = (w << 2) | (uint64_t)raw;
            }
            dc.target[pol][m] = w;
        }
    }
    if (hipMalloc(&b->d_state, sizeof(DdnCqState) * (size_t)cfg->n_channels) != hipSuccess
        || ddn_dev_cq_rx_init(b->d_state, cfg->n_channels, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        ddn_set_error("ddn_cq_rx_create: device allocation failed");
        (void)hipFree(b->d_state);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_cq_rx_destroy(ddn_cq_rx* b) {
    if (!b) {
        return;
    }
    (void)hipFree(b->d_state);
    delete b;
}

extern "C" int
ddn_cq_rx_reset(ddn_cq_rx* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_cq_rx_init(b->d_state, b->cfg.n_channels, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_cq_rx_set_events(ddn_cq_rx* b, int32_t* d_events, int32_t* d_n_events, int32_t* d_event_data, size_t max_events) {
    if (!b || ((d_events == nullptr) != (d_n_events == nullptr)) || (d_events && max_events == 0) || max_events > 0x7FFFFFFF
        || (d_event_data && !d_events)) {
        return DDN_EINVAL;
    }
    b->d_events = d_events;
    b->d_n_events = d_n_events;
    b->d_event_data = d_events ? d_event_data : nullptr;
    b->dc.max_events = d_events ? (int)max_events : 0;
    return DDN_OK;
}

extern "C" int
ddn_cq_rx_run(ddn_cq_rx* b, const float* d_symbols, const int32_t* d_counts_in, size_t n, size_t sym_stride, uint8_t* d_records10,
              uint8_t* d_flags, int32_t* d_counts, size_t max_symbols, void* hip_stream) {
    if (!b || !d_symbols || !d_records10 || !d_flags || !d_counts || n > 0x7FFFFFFF || sym_stride < n) {
        ddn_set_error("ddn_cq_rx_run: bad argument");
        return DDN_EINVAL;
    }
    if (max_symbols < n) {
        ddn_set_error("ddn_cq_rx_run: max_symbols %zu < n %zu", max_symbols, n);
        return DDN_ERANGE;
    }
    HIP_TRY(ddn_dev_cq_rx(d_symbols, d_counts_in, sym_stride, (int)n, b->cfg.n_channels, &b->dc, b->d_state, d_records10, d_flags, d_counts,
                          max_symbols, b->d_events, b->d_n_events, b->d_event_data, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_cq_rx_get_state(ddn_cq_rx* b, int channel, float out8[8]) {
    if (!b || !out8 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    static DdnCqState h; // (10 KB: not on the stack)
    HIP_TRY(hipMemcpy(&h, b->d_state + channel, sizeof(h), hipMemcpyDeviceToHost));
    out8[0] = (h.max + h.min) / 2.0f;
    out8[1] = h.max;
    out8[2] = h.min;
    out8[3] = (float)h.map_idx;
    out8[4] = (float)h.lastsync;
    out8[5] = (float)h.have_sync;
    out8[6] = (float)h.hunt_pos;
    out8[7] = (float)h.midx;
    return DDN_OK;
}
