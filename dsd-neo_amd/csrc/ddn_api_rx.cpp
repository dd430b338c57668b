// ddn_api_rx.cpp — C-ABI of the batched fixed-protocol P25p1 receive loop (include/ddn_hip.h, kernel ddn_rx.hip).
// Per-channel decoder words (timing, hunting window, thresholds, the 128-symbol window, the two 1024-deep extrema
// rings, the matched filter's 90-sample memory) live on the device inside the batch object and persist across calls.

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ddn_device.h"

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

struct ddn_p25_rx {
    ddn_p25_rx_config cfg;
    DdnRxState* d_state;
    float *d_sbuf, *d_lbuf, *d_shist, *d_minring, *d_maxring, *d_fhist;
    float* d_fstale; // [B][90] the matched filter's memory as the last carrier loss left it (zeros on a fresh stream)
    float* d_filt; // always-on matched-filter output of the current call, [B][filt_cap]
    size_t filt_cap;
    int channels_per_wave;
    int filter_in_loop; // ddn_p25_rx_set_filter_in_loop
    int dbg_flags;      // ddn_p25_rx_set_debug_flags
    int32_t* d_lock; // [B] in-frame symbols after a sync, per channel (cfg.lock_symbols unless overridden)
    // handler mode (ddn_p25_rx_set_handlers): per-channel handler words, the in-frame history ring [B][104][3] f32, the
    // caller's event buffers (or a one-event dummy of our own)
    int handlers, nid_threshold;
    DdnP25HState* d_hstate;
    float* d_hh;
    int32_t *d_events, *d_n_events, *d_event_data;
    size_t max_events;
    int32_t *d_ev_dummy, *d_nev_dummy;
    bool timing;
    hipEvent_t ev[3];
    float last_ms[2]; // matched filter, receive-loop kernel
    // timing history: event triples of the last DDN_RX_TRING launches, so a caller can average launch durations over a run
    // without synchronising between launches (ddn_p25_rx_get_timing_avg)
    hipEvent_t ring[64][3];
    int ring_n, ring_head;
    hipEvent_t gate_event; // the next run's loop kernel waits for it (ddn_p25_rx_gate_loop)
    hipEvent_t loop_event; // recorded between the matched filter and the loop kernel of the next run (ddn_p25_rx_mark_loop_start)
};

static void
rx_free(ddn_p25_rx* b) {
    (void)hipFree(b->d_state);
    (void)hipFree(b->d_sbuf);
    (void)hipFree(b->d_lbuf);
    (void)hipFree(b->d_shist);
    (void)hipFree(b->d_minring);
    (void)hipFree(b->d_maxring);
    (void)hipFree(b->d_fhist);
    (void)hipFree(b->d_fstale);
    (void)hipFree(b->d_filt);
    (void)hipFree(b->d_lock);
    (void)hipFree(b->d_hstate);
    (void)hipFree(b->d_hh);
    (void)hipFree(b->d_ev_dummy);
    (void)hipFree(b->d_nev_dummy);
    for (int i = 0; i < 3; i++) {
        if (b->ev[i]) {
            (void)hipEventDestroy(b->ev[i]);
        }
    }
    for (int k = 0; k < 64; k++) {
        for (int i = 0; i < 3; i++) {
            if (b->ring[k][i]) {
                (void)hipEventDestroy(b->ring[k][i]);
            }
        }
    }
}

static int
rx_fill(ddn_p25_rx* b) {
    // symbol_reset_rtl_fsk_timing_if_needed() + symbol_reset_rtl_fsk_discriminator_slicer()
    // (reference src/dsp/dsd_symbol.c:1306-1341): jitter -1, accumulator 0, slicer words at their RTL-FSK reset values
    const size_t B = (size_t)b->cfg.n_channels;
    DdnRxState s;
    memset(&s, 0, sizeof(s));
    s.center = 0.0f;
    s.min = -30000.0f;
    s.max = 30000.0f;
    s.lmid = -20000.0f;
    s.umid = 20000.0f;
    s.minref = -24000.0f;
    s.maxref = 24000.0f;
    s.fill_min = s.min;
    s.fill_max = s.max;
    s.since_fill = 0;
    s.min_sum = (double)s.min * 1024.0;
    s.max_sum = (double)s.max * 1024.0;
    s.jitter = -1;
    s.lmin = s.min;
    s.lmax = s.max;
    std::vector<DdnRxState> hs(B, s);
    if (hipMemcpy(b->d_state, hs.data(), sizeof(s) * B, hipMemcpyHostToDevice) != hipSuccess
        || hipMemset(b->d_sbuf, 0, sizeof(float) * 128 * B) != hipSuccess
        || hipMemset(b->d_lbuf, 0, sizeof(float) * 24 * B) != hipSuccess
        || hipMemset(b->d_shist, 0, sizeof(float) * 24 * B) != hipSuccess
        || hipMemset(b->d_minring, 0, sizeof(float) * 1024 * B) != hipSuccess
        || hipMemset(b->d_maxring, 0, sizeof(float) * 1024 * B) != hipSuccess
        || hipMemset(b->d_fhist, 0, sizeof(float) * 90 * B) != hipSuccess
        || hipMemset(b->d_fstale, 0, sizeof(float) * 90 * B) != hipSuccess
        || hipMemset(b->d_hstate, 0, sizeof(DdnP25HState) * B) != hipSuccess
        || hipMemset(b->d_hh, 0, sizeof(float) * 3 * 104 * B) != hipSuccess) {
        ddn_set_error("p25 rx state upload failed");
        return DDN_EHIP;
    }
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_create(const ddn_p25_rx_config* cfg, ddn_p25_rx** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->out_rate_hz <= 0 || cfg->sym_rate_hz <= 0
        || cfg->lock_symbols < 0) {
        ddn_set_error("ddn_p25_rx_create: bad configuration");
        return DDN_EINVAL;
    }
    if (cfg->use_matched_filter && (cfg->out_rate_hz != 10 * cfg->sym_rate_hz)) {
        // the P25 matched filter's coefficient set is per samples/symbol (reference src/dsp/dsd_filters.c:368);
        // this library carries the 10 samples/symbol set
        ddn_set_error("ddn_p25_rx_create: matched filter needs out_rate == 10 * sym_rate");
        return DDN_ERANGE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_p25_rx* b = new (std::nothrow) ddn_p25_rx();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->cfg = *cfg;
    const size_t B = (size_t)cfg->n_channels;
    if (hipMalloc(&b->d_state, sizeof(DdnRxState) * B) != hipSuccess
        || hipMalloc(&b->d_sbuf, sizeof(float) * 128 * B) != hipSuccess
        || hipMalloc(&b->d_lbuf, sizeof(float) * 24 * B) != hipSuccess
        || hipMalloc(&b->d_shist, sizeof(float) * 24 * B) != hipSuccess
        || hipMalloc(&b->d_minring, sizeof(float) * 1024 * B) != hipSuccess
        || hipMalloc(&b->d_maxring, sizeof(float) * 1024 * B) != hipSuccess
        || hipMalloc(&b->d_fhist, sizeof(float) * 90 * B) != hipSuccess
        || hipMalloc(&b->d_fstale, sizeof(float) * 90 * B) != hipSuccess
        || hipMalloc(&b->d_hstate, sizeof(DdnP25HState) * B) != hipSuccess
        || hipMalloc(&b->d_hh, sizeof(float) * 3 * 104 * B) != hipSuccess
        || hipMalloc(&b->d_ev_dummy, sizeof(int32_t) * 4 * B) != hipSuccess
        || hipMalloc(&b->d_nev_dummy, sizeof(int32_t) * B) != hipSuccess
        || hipMalloc(&b->d_lock, sizeof(int32_t) * B) != hipSuccess || rx_fill(b) != DDN_OK
        || ddn_p25_rx_set_lock_symbols(b, nullptr) != DDN_OK) {
        ddn_set_error("ddn_p25_rx_create: device allocation failed");
        rx_free(b);
        delete b;
        return DDN_ENOMEM;
    }
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_p25_rx_destroy(ddn_p25_rx* b) {
    if (!b) {
        return;
    }
    rx_free(b);
    delete b;
}

extern "C" int
ddn_p25_rx_reset(ddn_p25_rx* b) {
    if (!b) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    return rx_fill(b);
}

extern "C" int
ddn_p25_rx_set_lock_symbols(ddn_p25_rx* b, const int32_t* per_channel) {
    if (!b) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    std::vector<int32_t> v(B, b->cfg.lock_symbols);
    if (per_channel) {
        for (size_t c = 0; c < B; c++) {
            if (per_channel[c] < 0) {
                ddn_set_error("ddn_p25_rx_set_lock_symbols: channel %zu has a negative value", c);
                return DDN_EINVAL;
            }
            v[c] = per_channel[c];
        }
    }
    HIP_TRY(hipMemcpy(b->d_lock, v.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_handlers(ddn_p25_rx* b, int enable, int nid_erasure_threshold) {
    if (!b || nid_erasure_threshold > 255) {
        return DDN_EINVAL;
    }
    if (enable && b->cfg.out_rate_hz / b->cfg.sym_rate_hz < 9) {
        ddn_set_error("ddn_p25_rx_set_handlers: needs at least 9 samples per symbol");
        return DDN_ERANGE;
    }
    b->handlers = enable ? 1 : 0;
    b->nid_threshold = nid_erasure_threshold > 0 ? nid_erasure_threshold : 64;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_events(ddn_p25_rx* b, int32_t* d_events, int32_t* d_n_events, size_t max_events) {
    if (!b || ((d_events == nullptr) != (d_n_events == nullptr)) || (d_events && max_events == 0) || max_events > 0x7FFFFFFF) {
        return DDN_EINVAL;
    }
    b->d_events = d_events;
    b->d_n_events = d_n_events;
    b->max_events = d_events ? max_events : 0;
    if (!d_events) {
        b->d_event_data = nullptr;
    }
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_event_data(ddn_p25_rx* b, int32_t* d_event_data) {
    if (!b || (d_event_data && !b->d_events)) {
        return DDN_EINVAL; // the payload rows follow the event list's indexing: set the list first
    }
    b->d_event_data = d_event_data;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_channels_per_wave(ddn_p25_rx* b, int channels_per_wave) {
    if (!b || (channels_per_wave != 0 && channels_per_wave != 4 && channels_per_wave != 8 && channels_per_wave != 16 && channels_per_wave != 32
               && channels_per_wave != 64)) {
        return DDN_EINVAL;
    }
    b->channels_per_wave = channels_per_wave;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_debug_flags(ddn_p25_rx* b, int flags) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->dbg_flags = flags;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_filter_in_loop(ddn_p25_rx* b, int on) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->filter_in_loop = on != 0;
    return DDN_OK;
}

extern "C" size_t
ddn_p25_rx_max_symbols(const ddn_p25_rx* b, size_t n) {
    if (!b) {
        return 0;
    }
    int whole = b->cfg.out_rate_hz / b->cfg.sym_rate_hz;
    whole = whole < 2 ? 2 : (whole > 64 ? 64 : whole);
    // every symbol consumes at least whole - 1 samples (one-sample slip while hunting); + the one carried in
    return n / (size_t)(whole - 1) + 2;
}

extern "C" int
ddn_p25_rx_run(ddn_p25_rx* b, const float* d_disc, size_t n, uint8_t* d_records10, uint8_t* d_flags, int32_t* d_counts,
               size_t max_symbols, void* hip_stream) {
    if (!b || !d_disc || !d_records10 || !d_flags || !d_counts) {
        ddn_set_error("ddn_p25_rx_run: null argument");
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int B = b->cfg.n_channels;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)B, st));
        return DDN_OK;
    }
    if (max_symbols < ddn_p25_rx_max_symbols(b, n)) {
        // the loop counts every symbol but stores only max_symbols of them: downstream readers (framer) trust counts
        ddn_set_error("ddn_p25_rx_run: max_symbols %zu < ddn_p25_rx_max_symbols(n) = %zu", max_symbols,
                      ddn_p25_rx_max_symbols(b, n));
        return DDN_ERANGE;
    }
    hipEvent_t* rv = nullptr;
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[0], st));
        rv = b->ring[b->ring_head];
        for (int i = 0; i < 3; i++) {
            if (!rv[i]) {
                HIP_TRY(hipEventCreate(&rv[i]));
            }
        }
        HIP_TRY(hipEventRecord(rv[0], st));
    }
    DdnRxConfig dc = {b->cfg.out_rate_hz, b->cfg.sym_rate_hz, b->cfg.lock_symbols, b->cfg.use_matched_filter ? 1 : 0, 0,
                      b->handlers, b->nid_threshold, b->d_events ? (int)b->max_events : 1,
                      b->d_events ? b->d_event_data : nullptr};
    dc.dbg = b->dbg_flags;
    if (const char* e = DDN_EXP_ENV("DDN_RX_DBG")) {
        dc.dbg = (int)strtoll(e, nullptr, 0);
    }
    // handler mode, on request (ddn_p25_rx_set_filter_in_loop): the loop kernel filters each staged tile itself (ddn_rx.hip "the
    // matched filter inside the loop") - no filter kernel, no second f32 row in HBM
    const bool fused = b->filter_in_loop && ddn_dev_p25_rx_fuses_filter(&dc, b->channels_per_wave, B) != 0;
    if (b->cfg.use_matched_filter && !fused) {
        if (b->filt_cap < n) {
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(b->d_filt);
            b->d_filt = nullptr;
            b->filt_cap = 0;
            HIP_TRY(hipMalloc(&b->d_filt, sizeof(float) * (size_t)B * n));
            b->filt_cap = n;
        }
        HIP_TRY(ddn_dev_p25_matched_filter_only(d_disc, (long)n, n, B, b->d_fhist, b->d_filt, st));
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[1], st));
        HIP_TRY(hipEventRecord(rv[1], st));
    }
    if (b->gate_event) { // one shot: the loop kernel (not the matched filter above) waits for this event
        HIP_TRY(hipStreamWaitEvent(st, b->gate_event, 0));
        b->gate_event = nullptr;
    }
    if (b->loop_event) { // one shot: the chain object releases its result copies when the loop kernel is next on the stream
        HIP_TRY(hipEventRecord(b->loop_event, st));
        b->loop_event = nullptr;
    }
    HIP_TRY(ddn_dev_p25_rx(d_disc, fused ? nullptr : b->d_filt, b->d_fhist, b->d_fstale, (long)n, n, B, &dc, b->d_state, b->d_sbuf, b->d_lbuf,
                           b->d_shist, b->d_minring, b->d_maxring, d_records10, d_flags, d_counts, max_symbols,
                           b->channels_per_wave, b->d_lock, b->d_hstate, b->d_hh, b->d_events ? b->d_events : b->d_ev_dummy,
                           b->d_n_events ? b->d_n_events : b->d_nev_dummy, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[2], st));
        HIP_TRY(hipEventRecord(rv[2], st));
        b->ring_head = (b->ring_head + 1) % 64;
        b->ring_n = b->ring_n < 64 ? b->ring_n + 1 : 64;
    }
    // the filter memory (last 90 raw samples) moves on only after the loop has read the previous tail
    HIP_TRY(ddn_dev_p25_filter_hist_update(d_disc, (long)n, n, B, b->d_fhist, st));
    return DDN_OK;
}

// internal (ddn_internal.h): the next ddn_p25_rx_run records `hip_event` on its stream after the matched filter, right before the
// loop kernel
// internal: the next ddn_p25_rx_run makes its stream wait for `hip_event` between the matched filter and the loop kernel (the loop
// fills the device and must not start while the previous call's LDS-hungry decode kernels hold CUs; the matched filter may)
extern "C" int
ddn_p25_rx_gate_loop(ddn_p25_rx* b, void* hip_event) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->gate_event = (hipEvent_t)hip_event;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_mark_loop_start(ddn_p25_rx* b, void* hip_event) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->loop_event = (hipEvent_t)hip_event;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_set_timing(ddn_p25_rx* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    if (enable && !b->ev[0]) {
        for (int i = 0; i < 3; i++) {
            HIP_TRY(hipEventCreate(&b->ev[i]));
        }
    }
    b->timing = enable != 0;
    b->ring_n = 0;
    b->ring_head = 0;
    return DDN_OK;
}

// average {matched filter, receive-loop kernel} launch durations over the launches recorded since timing was switched on
// (the last 64 at most); *n_launches = how many went into it.  Synchronises on the last one.
extern "C" int
ddn_p25_rx_get_timing_avg(ddn_p25_rx* b, float* ms2, int* n_launches) {
    if (!b || !ms2 || !n_launches) {
        return DDN_EINVAL;
    }
    ms2[0] = ms2[1] = 0.0f;
    *n_launches = b->ring_n;
    if (b->ring_n == 0) {
        return DDN_OK;
    }
    const int last = (b->ring_head + 63) % 64;
    HIP_TRY(hipEventSynchronize(b->ring[last][2]));
    for (int k = 0; k < b->ring_n; k++) {
        const int idx = (b->ring_head + 64 - 1 - k) % 64;
        float a = 0.0f, c = 0.0f;
        HIP_TRY(hipEventElapsedTime(&a, b->ring[idx][0], b->ring[idx][1]));
        HIP_TRY(hipEventElapsedTime(&c, b->ring[idx][1], b->ring[idx][2]));
        ms2[0] += a;
        ms2[1] += c;
    }
    ms2[0] /= (float)b->ring_n;
    ms2[1] /= (float)b->ring_n;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_get_timing(ddn_p25_rx* b, float* ms2) {
    if (!b || !ms2 || !b->ev[0]) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipEventSynchronize(b->ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms2[0], b->ev[0], b->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms2[1], b->ev[1], b->ev[2]));
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_run_host_ev(ddn_p25_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags, int32_t* counts,
                       size_t max_symbols, int32_t* events, int32_t* n_events, size_t max_events, int32_t* event_data) {
    if (!b || !disc || !records10 || !flags || !counts || ((events == nullptr) != (n_events == nullptr))
        || (events && max_events == 0) || (event_data && !events)) {
        return DDN_EINVAL;
    }
    const size_t B = (size_t)b->cfg.n_channels;
    float* d_in = nullptr;
    uint8_t *d_rec = nullptr, *d_fl = nullptr;
    int32_t *d_cnt = nullptr, *d_ev = nullptr, *d_nev = nullptr, *d_evd = nullptr;
    int32_t* const keep_evd = b->d_event_data;
    int32_t* const keep_ev = b->d_events;
    int32_t* const keep_nev = b->d_n_events;
    const size_t keep_max = b->max_events;
    int rc;
    if (hipMalloc(&d_in, B * n * 4 + 4) != hipSuccess || hipMalloc(&d_rec, B * max_symbols * 10 + 4) != hipSuccess
        || hipMalloc(&d_fl, B * max_symbols + 4) != hipSuccess || hipMalloc(&d_cnt, B * 4) != hipSuccess
        || (events && (hipMalloc(&d_ev, B * max_events * 16) != hipSuccess || hipMalloc(&d_nev, B * 4) != hipSuccess))
        || (event_data && hipMalloc(&d_evd, B * max_events * 16) != hipSuccess)) {
        ddn_set_error("ddn_p25_rx_run_host: device allocation failed (no device?)");
        rc = DDN_ENODEV;
    } else if (hipMemcpy(d_in, disc, B * n * 4, hipMemcpyHostToDevice) != hipSuccess
               || hipMemset(d_rec, 0, B * max_symbols * 10) != hipSuccess
               || hipMemset(d_fl, 0, B * max_symbols) != hipSuccess
               || (events && (hipMemset(d_ev, 0, B * max_events * 16) != hipSuccess || hipMemset(d_nev, 0, B * 4) != hipSuccess))
               || (event_data && hipMemset(d_evd, 0, B * max_events * 16) != hipSuccess)) {
        rc = DDN_EHIP;
    } else {
        if (events) {
            b->d_events = d_ev;
            b->d_n_events = d_nev;
            b->max_events = max_events;
            b->d_event_data = d_evd;
        }
        rc = ddn_p25_rx_run(b, d_in, n, d_rec, d_fl, d_cnt, max_symbols, nullptr);
        if (rc == DDN_OK
            && (hipDeviceSynchronize() != hipSuccess
                || hipMemcpy(records10, d_rec, B * max_symbols * 10, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(flags, d_fl, B * max_symbols, hipMemcpyDeviceToHost) != hipSuccess
                || hipMemcpy(counts, d_cnt, B * 4, hipMemcpyDeviceToHost) != hipSuccess
                || (events
                    && (hipMemcpy(events, d_ev, B * max_events * 16, hipMemcpyDeviceToHost) != hipSuccess
                        || hipMemcpy(n_events, d_nev, B * 4, hipMemcpyDeviceToHost) != hipSuccess))
                || (event_data && hipMemcpy(event_data, d_evd, B * max_events * 16, hipMemcpyDeviceToHost) != hipSuccess))) {
            ddn_set_error("ddn_p25_rx_run_host: %s", hipGetErrorString(hipGetLastError()));
            rc = DDN_EHIP;
        }
    }
    b->d_events = keep_ev;
    b->d_n_events = keep_nev;
    b->max_events = keep_max;
    b->d_event_data = keep_evd;
    (void)hipDeviceSynchronize();
    (void)hipFree(d_evd);
    (void)hipFree(d_in);
    (void)hipFree(d_rec);
    (void)hipFree(d_fl);
    (void)hipFree(d_cnt);
    (void)hipFree(d_ev);
    (void)hipFree(d_nev);
    return rc;
}

extern "C" int
ddn_p25_rx_run_host(ddn_p25_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags, int32_t* counts,
                    size_t max_symbols) {
    return ddn_p25_rx_run_host_ev(b, disc, n, records10, flags, counts, max_symbols, nullptr, nullptr, 0, nullptr);
}

// timing experiments (DDN_RX_DBG bit 65536): handler requests of a channel and the cycles its lane spent waiting for the answers
extern "C" int
ddn_p25_rx_debug_counters(ddn_p25_rx* b, int channel, long long out2[2]) {
    if (!b || !out2 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    DdnRxState s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_state + channel, sizeof(s), hipMemcpyDeviceToHost));
    out2[0] = s.dbg_nreq;
    out2[1] = s.dbg_wait;
    return DDN_OK;
}

extern "C" int
ddn_p25_rx_get_thresholds(ddn_p25_rx* b, int channel, float out7[7]) {
    if (!b || !out7 || channel < 0 || channel >= b->cfg.n_channels) {
        return DDN_EINVAL;
    }
    DdnRxState s;
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
