// ddn_api_adapter.cpp - single-stream full_demod() / op25_gardner_cc() under the reference's names (include/ddn_demod_adapter.h):
// each stream gets one-channel batch objects created on first use and keyed by the configuration members; a call is one block.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "ddn_demod_adapter.h"
#include "ddn_device.h"

namespace {
constexpr int kBlock = 1 << 16; // a call is one full_demod() block: any lp_len / 2 up to this is a single (short) block

struct Adapter {
    ddn_batch* fe = nullptr;
    ddn_cqpsk_batch* cq = nullptr;
    ddn_ted_batch* ted = nullptr;
    int fe_key[5] = {0, 0, 0, 0, 0};
    float fe_sq = 0.0f;
    int cq_key[4] = {0, 0, 0, 0};
    float cq_gain = 0.0f;
    int ted_key[2] = {0, 0};
    float ted_gain = 0.0f;
    std::vector<float> cq_row; // CQPSK symbols of one call at the loop's own row capacity
};

Adapter*
adapter_of(struct demod_state* s) {
    if (!s->ddn_adapter) {
        s->ddn_adapter = new (std::nothrow) Adapter();
    }
    return static_cast<Adapter*>(s->ddn_adapter);
}
} // namespace

extern "C" void
ddn_demod_state_release(struct demod_state* s) {
    if (!s || !s->ddn_adapter) {
        return;
    }
    Adapter* a = static_cast<Adapter*>(s->ddn_adapter);
    if (a->fe) {
        ddn_batch_destroy(a->fe);
    }
    if (a->cq) {
        ddn_cqpsk_batch_destroy(a->cq);
    }
    if (a->ted) {
        ddn_ted_batch_destroy(a->ted);
    }
    delete a;
    s->ddn_adapter = nullptr;
}

extern "C" void
full_demod(struct demod_state* s) {
    if (!s) {
        return;
    }
    s->result_len = 0;
    if (!s->lowpassed || s->lp_len < 2) {
        return;
    }
    const int n = s->lp_len >> 1;
    if (n > kBlock) {
        ddn_set_error("full_demod adapter: %d complex samples per call exceed the adapter's block (%d)", n, kBlock);
        return;
    }
    Adapter* a = adapter_of(s);
    if (!a) {
        return;
    }
    const int rate = s->rate_out > 0 ? s->rate_out : s->rate_in;
    if (s->output_kind == DSD_DEMOD_OUTPUT_FSK_DISCRIMINATOR && !s->cqpsk_enable) {
        if (!s->channel_lpf_enable) {
            ddn_set_error("full_demod adapter: the FSK-discriminator batch always runs the channel LPF (channel_lpf_enable = 0 unsupported)");
            return;
        }
        const int key[5] = {rate, s->symbol_rate_hz, s->symbol_levels, s->channel_lpf_profile, 1};
        if (!a->fe || memcmp(key, a->fe_key, sizeof(key)) != 0 || a->fe_sq != s->channel_squelch_level) {
            if (a->fe) {
                ddn_batch_destroy(a->fe);
                a->fe = nullptr;
            }
            ddn_front_end_config c = {1, rate, s->symbol_rate_hz, s->symbol_levels, s->channel_lpf_profile, DDN_IN_CF32, kBlock,
                                      s->channel_squelch_level};
            if (ddn_batch_create(&c, &a->fe) != DDN_OK) {
                return;
            }
            memcpy(a->fe_key, key, sizeof(key));
            a->fe_sq = s->channel_squelch_level;
        }
        if (ddn_front_end_run_host(a->fe, s->lowpassed, (size_t)n, s->result) == DDN_OK) {
            s->result_len = n;
        }
        return;
    }
    if (s->output_kind == DSD_DEMOD_OUTPUT_SYMBOL_CQPSK || s->cqpsk_enable) {
        const int sym_rate = s->symbol_rate_hz > 0 ? s->symbol_rate_hz : (s->ted_sps > 0 ? rate / s->ted_sps : 4800);
        const int key[4] = {rate, sym_rate, s->channel_lpf_profile, s->channel_lpf_enable};
        if (!a->cq || memcmp(key, a->cq_key, sizeof(key)) != 0 || a->cq_gain != s->ted_gain) {
            if (a->cq) {
                ddn_cqpsk_batch_destroy(a->cq);
                a->cq = nullptr;
            }
            ddn_cqpsk_config c = {1, rate, sym_rate, s->channel_lpf_profile, s->channel_lpf_enable, DDN_IN_CF32, kBlock, s->ted_gain};
            if (ddn_cqpsk_batch_create(&c, &a->cq) != DDN_OK) {
                return;
            }
            memcpy(a->cq_key, key, sizeof(key));
            a->cq_gain = s->ted_gain;
        }
        // the loop's row capacity (n / sps + n / (100 sps) + 8) can exceed the caller's result buffer (>= lp_len / 2 floats) on
        // very short blocks: run into a scratch row of that capacity and hand back the symbols that were produced
        int32_t cnt = 0;
        const size_t cap = ddn_cqpsk_max_symbols(a->cq, (size_t)n);
        if (a->cq_row.size() < cap) {
            a->cq_row.resize(cap);
        }
        if (ddn_cqpsk_run_host(a->cq, s->lowpassed, (size_t)n, a->cq_row.data(), cap, &cnt) == DDN_OK) {
            const int keep = cnt < n ? cnt : n;
            memcpy(s->result, a->cq_row.data(), sizeof(float) * (size_t)(keep > 0 ? keep : 0));
            s->result_len = keep;
        }
        return;
    }
    ddn_set_error("full_demod adapter: output_kind %d (audio monitor) is outside this library's path", s->output_kind);
}

extern "C" void
op25_gardner_cc(struct demod_state* s) {
    if (!s || !s->lowpassed || s->lp_len < 2 || !s->cqpsk_enable) {
        return;
    }
    const int n = s->lp_len >> 1;
    if (n < 4 || n > kBlock) {
        return; // the reference returns early below four samples as well (src/dsp/costas.cpp:811-813)
    }
    Adapter* a = adapter_of(s);
    if (!a) {
        return;
    }
    const int sps = s->ted_sps > 0 ? s->ted_sps : 5;
    const int sym_rate = s->symbol_rate_hz > 0 ? s->symbol_rate_hz : 4800;
    const int key[2] = {sps, sym_rate};
    if (!a->ted || memcmp(key, a->ted_key, sizeof(key)) != 0 || a->ted_gain != s->ted_gain) {
        if (a->ted) {
            ddn_ted_batch_destroy(a->ted);
            a->ted = nullptr;
        }
        if (ddn_ted_batch_create(1, sps, sym_rate, s->ted_gain, &a->ted) != DDN_OK) {
            return;
        }
        memcpy(a->ted_key, key, sizeof(key));
        a->ted_gain = s->ted_gain;
    }
    int cnt = 0;
    // symbols go back into lowpassed (never more than the samples that came in), lp_len becomes 2 x symbols
    float* tmp = new (std::nothrow) float[(size_t)n * 2 + 8];
    if (!tmp) {
        return;
    }
    if (ddn_gardner_run_host(a->ted, s->lowpassed, (size_t)n, tmp, (size_t)n + 4, &cnt) == DDN_OK) {
        memcpy(s->lowpassed, tmp, sizeof(float) * 2 * (size_t)cnt);
        s->lp_len = 2 * cnt;
    }
    delete[] tmp;
}
