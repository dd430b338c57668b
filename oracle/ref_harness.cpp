// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY.
//
// Our own thin driver, compiled INTO oracle/_ref/libdsdneo_ref.so next to the reference's unmodified
// translation units (oracle/Makefile).  It owns a `struct demod_state` and feeds the reference's public
// full_demod() (include/dsd-neo/dsp/demod_pipeline.h:106) with FIXED-size blocks so that its block-edge
// sample replication (src/dsp/simd_fir.cpp:66-85) is deterministic — the threaded CLI reads "whatever is in
// the ring" (src/io/radio/rtl_sdr_fm.cpp:2290-2305) and cannot be a parity oracle.
//
// Nothing here restates reference arithmetic; it only sequences reference calls the way the demod thread
// does (src/io/radio/rtl_sdr_fm.cpp:3458-3516: read block -> full_demod -> write result) after the replay
// device's cu8 widening (src/io/radio/rtl_device.cpp:1777 -> widen_u8_to_f32_bias127_moments).

#include <dsd-neo/dsp/demod_pipeline.h>
#include <dsd-neo/dsp/demod_state.h>
#include <dsd-neo/dsp/fsk_modem.h>
#include <dsd-neo/dsp/simd_fir.h>
#include <dsd-neo/dsp/simd_widen.h>
#include <dsd-neo/runtime/mem.h>

#include <cstdint>
#include <cstring>
#include <new>

namespace {
struct RefFrontEnd {
    demod_state* d;
};
} // namespace

extern "C" {

// Field set follows rtl_demod_init_for_mode()/demod_init_common_defaults
// (src/io/radio/rtl_demod_config.cpp:258-338,535) for an FSK-discriminator digital mode.
void*
refh_fe_create(int rate_hz, int symbol_rate_hz, int levels, int lpf_profile, int lpf_enable, float squelch_level,
               int downsample_passes) {
    void* mem = dsd_neo_aligned_malloc(sizeof(demod_state));
    if (!mem) {
        return nullptr;
    }
    demod_state* d = new (mem) demod_state();
    d->rate_in = rate_hz << (downsample_passes > 0 ? downsample_passes : 0);
    d->rate_out = rate_hz;
    d->downsample_passes = downsample_passes > 0 ? downsample_passes : 0;
    d->mode_demod = &dsd_fm_demod;
    d->output_kind = DSD_DEMOD_OUTPUT_FSK_DISCRIMINATOR;
    d->symbol_rate_hz = symbol_rate_hz;
    d->symbol_levels = levels;
    d->ted_sps = (symbol_rate_hz > 0) ? rate_hz / symbol_rate_hz : 10;
    d->sps_is_integer = (symbol_rate_hz > 0 && (rate_hz % symbol_rate_hz) == 0) ? 1 : 0;
    d->channel_lpf_enable = lpf_enable;
    d->channel_lpf_profile = lpf_profile;
    d->channel_squelch_level = squelch_level;
    d->squelch_env = 1.0f;
    d->squelch_gate_open = 1;
    dsd_fsk_modem_config cfg = {rate_hz, symbol_rate_hz, levels, lpf_profile};
    dsd_fsk_modem_init(&d->fsk_modem_state, &cfg);
    RefFrontEnd* fe = new RefFrontEnd{d};
    return fe;
}

void
refh_fe_destroy(void* h) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    if (!fe) {
        return;
    }
    fe->d->~demod_state();
    dsd_neo_aligned_free(fe->d);
    delete fe;
}

// One full_demod() call on one block of already-widened interleaved floats.  Returns result_len.
int
refh_fe_block_f32(void* h, const float* iq, int n_complex, float* out, int out_cap) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    demod_state* d = fe->d;
    if (n_complex * 2 > MAXIMUM_BUF_LENGTH) {
        return -1;
    }
    std::memcpy(d->input_cb_buf, iq, (size_t)n_complex * 2 * sizeof(float));
    d->lowpassed = d->input_cb_buf;
    d->lp_len = n_complex * 2;
    full_demod(d);
    int n = d->result_len < out_cap ? d->result_len : out_cap;
    std::memcpy(out, d->result, (size_t)n * sizeof(float));
    return n;
}

// Whole capture: cu8 -> widen -> full_demod per `block_len` complex samples (last block may be short).
// Returns total discriminator samples written to out (== n_complex >> downsample_passes for FSK).
long
refh_fe_run_cu8(void* h, const uint8_t* iq_u8, long n_complex, int block_len, float* out) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    demod_state* d = fe->d;
    long done = 0, written = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        widen_u8_to_f32_bias127(iq_u8 + 2 * done, d->input_cb_buf, (uint32_t)(2 * n));
        d->lowpassed = d->input_cb_buf;
        d->lp_len = (int)(2 * n);
        full_demod(d);
        std::memcpy(out + written, d->result, (size_t)d->result_len * sizeof(float));
        written += d->result_len;
        done += n;
    }
    return written;
}

long
refh_fe_run_f32(void* h, const float* iq, long n_complex, int block_len, float* out) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    long done = 0, written = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        int got = refh_fe_block_f32(h, iq + 2 * done, (int)n, out + written, (int)n);
        if (got < 0) {
            return -1;
        }
        written += got;
        done += n;
    }
    (void)fe;
    return written;
}

int
refh_fe_get_taps(void* h, float* taps, int cap) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    int n = fe->d->channel_lpf_plan_taps_len;
    for (int i = 0; i < n && i < cap; i++) {
        taps[i] = fe->d->channel_lpf_plan_taps[i];
    }
    return n;
}

// {prev_i, prev_q, have_prev, dc_est, peak_est, channel_pwr, channel_squelched}
void
refh_fe_get_state(void* h, float* out7) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    const dsd_fsk_modem_state* s = &fe->d->fsk_modem_state;
    out7[0] = s->prev_i;
    out7[1] = s->prev_q;
    out7[2] = (float)s->have_prev;
    out7[3] = s->dc_est;
    out7[4] = s->discriminator_peak_est;
    out7[5] = fe->d->channel_pwr;
    out7[6] = (float)fe->d->channel_squelched;
}

} // extern "C"

// ---- block codes: thin value-returning wrappers around the reference's C++ entry points -----------------
#include <dsd-neo/fec/BCH_63_16.hpp>
#include <dsd-neo/protocol/p25/p25p1_check_nid.h>

extern "C" {

// returns success (0/1); *err_count = corrected bits
int
refh_bch_63_16_decode(const char* in63, char* out16, int* err_count) {
    static BCH_63_16_11 bch;
    BCH_63_16_Result r = bch.decode_with_result(in63, out16);
    *err_count = r.error_count;
    return r.success ? 1 : 0;
}

// out4 = {status, nac, duid, error_count}
void
refh_nid_decode(const char* bch_code63, const uint8_t* reliab63, int observed_nac, int parity, int parity_reliab,
                int out4[4]) {
    struct p25p1_nid_result r =
        p25p1_nid_decode(bch_code63, reliab63, observed_nac, (unsigned char)parity, (uint8_t)parity_reliab);
    out4[0] = (int)r.status;
    out4[1] = r.nac;
    out4[2] = r.duid;
    out4[3] = r.error_count;
}

} // extern "C"

// ---- Gardner timing recovery: drive the reference's op25_gardner_cc() block by block ----------------------
#include <dsd-neo/dsp/costas.h>
#include "mmse_interp.h" /* src/dsp/mmse_interp.h (private header, on the -I path) */

extern "C" {

void*
refh_ted_create(int sps, float ted_gain, int symbol_rate_hz) {
    void* mem = dsd_neo_aligned_malloc(sizeof(demod_state));
    if (!mem) {
        return nullptr;
    }
    demod_state* d = new (mem) demod_state();
    d->cqpsk_enable = 1;
    d->ted_sps = sps;
    d->ted_gain = ted_gain;
    d->symbol_rate_hz = symbol_rate_hz;
    d->rate_out = sps * symbol_rate_hz;
    ted_init_state(&d->ted_state);
    return d;
}

void
refh_ted_destroy(void* h) {
    demod_state* d = static_cast<demod_state*>(h);
    if (d) {
        d->~demod_state();
        dsd_neo_aligned_free(d);
    }
}

// returns floats written to out (symbol-rate interleaved I/Q)
int
refh_ted_block(void* h, const float* iq, int n_complex, float* out) {
    demod_state* d = static_cast<demod_state*>(h);
    std::memcpy(d->input_cb_buf, iq, (size_t)n_complex * 2 * sizeof(float));
    d->lowpassed = d->input_cb_buf;
    d->lp_len = n_complex * 2;
    op25_gardner_cc(d);
    if (d->lowpassed == d->input_cb_buf) {
        return 0; // gardner returned early without producing symbols
    }
    std::memcpy(out, d->lowpassed, (size_t)d->lp_len * sizeof(float));
    return d->lp_len;
}

// {mu, omega, last_r, last_j, lock_accum, lock_count, dl_index, twice_sps}
void
refh_ted_state(void* h, float* out8) {
    const ted_state_t* t = &static_cast<demod_state*>(h)->ted_state;
    out8[0] = t->mu;
    out8[1] = t->omega;
    out8[2] = t->last_r;
    out8[3] = t->last_j;
    out8[4] = t->lock_accum;
    out8[5] = (float)t->lock_count;
    out8[6] = (float)t->dl_index;
    out8[7] = (float)t->twice_sps;
}

void
refh_mmse(const float* samples, float mu, float* re, float* im) {
    dsd_mmse_interp_complex_8tap(samples, mu, re, im);
}

} // extern "C"
