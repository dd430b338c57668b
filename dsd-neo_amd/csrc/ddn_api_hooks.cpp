// ddn_api_hooks.cpp — the drop-in seam on the consumer side: per-channel sample queues that
// serve the reference's stream-read hook from the results of batched runs.
//
// reference contract:
//   dsd_rtl_stream_io_hooks { int (*read)(void* rtl_ctx, float* out, size_t count, int* out_got);
//                             double (*return_pwr)(const void* rtl_ctx); }   include/dsd-neo/runtime/rtl_stream_io_hooks.h:25-32
//   read: called only from the decoder thread of one stream (rtl_symbol_cache_refill, src/dsp/dsd_symbol.c:889-920, and
//   symbol_read_sample_rtl :1412-1435) with count 512 or 1; blocks until at least one sample is available, writes up to
//   `count` floats, *out_got = number written, returns < 0 on end of stream / shutdown.
//   dsd_rtl_stream_metrics_hooks: output_rate_hz / output_kind / symbol_profile / stream_generation are argument-less
//   getters there (one stream per process); here they take the channel context, the host's hook shims pass it.
//
// Host-only code (no kernel launches): one single-producer / single-consumer ring per channel.  The producer is whoever
// drives the batched front end (ddn_front_end_run_host / ddn_cqpsk_run results, rows of one batch interval); each
// consumer is one dsd-neo decoder instance whose state->rtl_ctx is ddn_stream_set_ctx(set, channel).

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ddn_internal.h"

struct ddn_stream_chan {
    ddn_stream_set* set;
    std::vector<float> ring;
    size_t head = 0, fill = 0; // head = index of the oldest sample
    double pwr = 0.0;
    std::mutex mu;
    std::condition_variable can_read, can_write;
};

struct ddn_stream_set {
    int n_channels;
    size_t capacity;
    unsigned output_rate_hz;
    int output_kind, symbol_rate_hz, levels, channel_profile;
    std::atomic<uint32_t> generation;
    std::atomic<bool> closed;
    ddn_stream_chan* ch;
};

extern "C" int
ddn_stream_set_create(int n_channels, size_t capacity_samples, unsigned output_rate_hz, int output_kind,
                      int symbol_rate_hz, int levels, int channel_profile, ddn_stream_set** out) {
    if (!out || n_channels <= 0 || capacity_samples == 0 || (output_kind != 0 && output_kind != 1)) {
        ddn_set_error("ddn_stream_set_create: bad argument");
        return DDN_EINVAL;
    }
    ddn_stream_set* s = new (std::nothrow) ddn_stream_set();
    if (!s) {
        return DDN_ENOMEM;
    }
    s->n_channels = n_channels;
    s->capacity = capacity_samples;
    s->output_rate_hz = output_rate_hz;
    s->output_kind = output_kind;
    s->symbol_rate_hz = symbol_rate_hz;
    s->levels = levels;
    s->channel_profile = channel_profile;
    s->generation = 1;
    s->closed = false;
    s->ch = new (std::nothrow) ddn_stream_chan[(size_t)n_channels];
    if (!s->ch) {
        delete s;
        return DDN_ENOMEM;
    }
    try {
        for (int c = 0; c < n_channels; c++) {
            s->ch[c].set = s;
            s->ch[c].ring.assign(capacity_samples, 0.0f);
        }
    } catch (...) {
        delete[] s->ch;
        delete s;
        return DDN_ENOMEM;
    }
    *out = s;
    return DDN_OK;
}

extern "C" void
ddn_stream_set_close(ddn_stream_set* s) {
    if (!s) {
        return;
    }
    for (int c = 0; c < s->n_channels; c++) {
        std::lock_guard<std::mutex> lk(s->ch[c].mu);
        s->closed = true;
        s->ch[c].can_read.notify_all();
        s->ch[c].can_write.notify_all();
    }
}

extern "C" void
ddn_stream_set_destroy(ddn_stream_set* s) {
    if (!s) {
        return;
    }
    ddn_stream_set_close(s);
    delete[] s->ch;
    delete s;
}

extern "C" void*
ddn_stream_set_ctx(ddn_stream_set* s, int channel) {
    return (s && channel >= 0 && channel < s->n_channels) ? (void*)&s->ch[channel] : nullptr;
}

extern "C" int
ddn_stream_set_push(ddn_stream_set* s, const float* rows, size_t n, size_t row_stride, const int32_t* counts) {
    if (!s || !rows || row_stride < n) {
        ddn_set_error("ddn_stream_set_push: bad argument");
        return DDN_EINVAL;
    }
    for (int c = 0; c < s->n_channels; c++) {
        ddn_stream_chan& q = s->ch[c];
        size_t m = counts ? (size_t)(counts[c] < 0 ? 0 : counts[c]) : n;
        if (m > n) {
            m = n;
        }
        const float* src = rows + (size_t)c * row_stride;
        size_t done = 0;
        while (done < m) {
            std::unique_lock<std::mutex> lk(q.mu);
            q.can_write.wait(lk, [&] { return s->closed || q.fill < s->capacity; });
            if (s->closed) {
                ddn_set_error("ddn_stream_set_push: stream set is closed");
                return DDN_EINVAL;
            }
            size_t k = s->capacity - q.fill;
            if (k > m - done) {
                k = m - done;
            }
            for (size_t i = 0; i < k; i++) {
                q.ring[(q.head + q.fill + i) % s->capacity] = src[done + i];
            }
            q.fill += k;
            done += k;
            q.can_read.notify_one();
        }
    }
    return DDN_OK;
}

extern "C" int
ddn_stream_set_set_power(ddn_stream_set* s, int channel, double mean_power) {
    if (!s || channel < 0 || channel >= s->n_channels) {
        return DDN_EINVAL;
    }
    std::lock_guard<std::mutex> lk(s->ch[channel].mu);
    s->ch[channel].pwr = mean_power;
    return DDN_OK;
}

extern "C" int
ddn_stream_set_bump_generation(ddn_stream_set* s) {
    if (!s) {
        return DDN_EINVAL;
    }
    for (int c = 0; c < s->n_channels; c++) { // a retune / restart: queued samples of the old stream are dropped
        std::lock_guard<std::mutex> lk(s->ch[c].mu);
        s->ch[c].head = s->ch[c].fill = 0;
        s->ch[c].can_write.notify_all();
    }
    s->generation++;
    return DDN_OK;
}

// ---- the hook functions themselves ------------------------------------------------------------------------------------
extern "C" int
ddn_hooks_read(void* rtl_ctx, float* out, size_t count, int* out_got) {
    if (out_got) {
        *out_got = 0;
    }
    ddn_stream_chan* q = (ddn_stream_chan*)rtl_ctx;
    if (!q || !out || !out_got || count == 0) {
        return -1;
    }
    ddn_stream_set* s = q->set;
    std::unique_lock<std::mutex> lk(q->mu);
    q->can_read.wait(lk, [&] { return q->fill > 0 || s->closed; });
    if (q->fill == 0) {
        return -1; // closed and drained
    }
    size_t k = q->fill < count ? q->fill : count;
    for (size_t i = 0; i < k; i++) {
        out[i] = q->ring[(q->head + i) % s->capacity];
    }
    q->head = (q->head + k) % s->capacity;
    q->fill -= k;
    *out_got = (int)k;
    q->can_write.notify_one();
    return 0;
}

extern "C" double
ddn_hooks_return_pwr(const void* rtl_ctx) {
    ddn_stream_chan* q = (ddn_stream_chan*)rtl_ctx;
    if (!q) {
        return 0.0;
    }
    std::lock_guard<std::mutex> lk(q->mu);
    return q->pwr;
}

extern "C" unsigned int
ddn_hooks_output_rate_hz(const void* rtl_ctx) {
    const ddn_stream_chan* q = (const ddn_stream_chan*)rtl_ctx;
    return q ? q->set->output_rate_hz : 0u;
}

extern "C" int
ddn_hooks_output_kind(const void* rtl_ctx) {
    const ddn_stream_chan* q = (const ddn_stream_chan*)rtl_ctx;
    return q ? q->set->output_kind : 0;
}

extern "C" int
ddn_hooks_symbol_profile(const void* rtl_ctx, int* out_symbol_rate_hz, int* out_levels, int* out_channel_profile) {
    const ddn_stream_chan* q = (const ddn_stream_chan*)rtl_ctx;
    if (!q) {
        return -1;
    }
    if (out_symbol_rate_hz) {
        *out_symbol_rate_hz = q->set->symbol_rate_hz;
    }
    if (out_levels) {
        *out_levels = q->set->levels;
    }
    if (out_channel_profile) {
        *out_channel_profile = q->set->channel_profile;
    }
    return 0;
}

extern "C" uint32_t
ddn_hooks_stream_generation(const void* rtl_ctx) {
    const ddn_stream_chan* q = (const ddn_stream_chan*)rtl_ctx;
    return q ? q->set->generation.load() : 0u;
}
