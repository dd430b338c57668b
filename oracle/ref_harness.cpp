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

// optional IQ conditioning switches of struct demod_state (iq_dc_block / iq balance, off by default)
void
refh_fe_set_iq_options(void* h, int dc_enable, int dc_shift, int bal_enable, float bal_thr, float bal_ema_a) {
    demod_state* d = static_cast<RefFrontEnd*>(h)->d;
    d->iq_dc_block_enable = dc_enable;
    d->iq_dc_shift = dc_shift;
    d->iqbal_enable = bal_enable;
    d->iqbal_thr = bal_thr;
    d->iqbal_alpha_ema_a = bal_ema_a;
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

// ---- slicer / soft-decision path: the reference's own digitize()/use_symbol()/soft metrics (dsd_dibit.c) ------
// src/dsp/dsd_symbol.c cannot be built here (it includes <sndfile.h>), so the sample source getSymbol() is supplied
// by this harness the same way the reference's own unit tests substitute it (tests/CMakeLists.txt:1009-1015 wrap
// get_dibit_and_analog_signal / processMbeFrame): it hands out a caller-provided symbol vector, nothing else.
#include <cstdlib>
#include <dsd-neo/core/dibit.h>
#include <dsd-neo/core/opts.h>
#include <dsd-neo/core/state.h>
#include <dsd-neo/core/sync_patterns.h>
#include <dsd-neo/core/synctype_ids.h>
#include <dsd-neo/dsp/sps_filters.h>
#include <dsd-neo/dsp/symbol.h>

namespace {
const float* g_sym_src = nullptr;
long g_sym_n = 0, g_sym_pos = 0;
int g_cq_fast_path = 0; // the symbol-rate fast path's threshold constants ahead of every symbol (refh_cq_slicer_run)
} // namespace

extern "C" {

float
getSymbol(dsd_opts* opts, dsd_state* state, int have_sync) {
    (void)opts;
    (void)have_sync;
    if (!g_sym_src || g_sym_pos >= g_sym_n) {
        return 0.0f;
    }
    if (g_cq_fast_path) { // apply_rtl_symbol_thresholds(state, 4), src/dsp/dsd_symbol.c:744-765
        state->center = 0.0f;
        state->min = -3.0f;
        state->max = 3.0f;
        state->lmid = -2.0f;
        state->umid = 2.0f;
        state->minref = -2.4f;
        state->maxref = 2.4f;
    }
    const float v = g_sym_src[g_sym_pos++];
    state->lastsample = v;
    state->symbolcnt++;
    return v;
}

struct RefSlicer {
    dsd_opts* opts;
    dsd_state* state;
};

// P25 Phase 1 C4FM receive context after a positive-polarity sync: rf_mod 0, synctype/lastsynctype P25P1_POS,
// slicer reset exactly as symbol_reset_rtl_fsk_discriminator_slicer() does (src/dsp/dsd_symbol.c:1306-1326),
// window sizes from initOpts (src/core/util/dsd_init.c:169-170: ssize 128, msize 1024).
void*
refh_slicer_create(int synctype) {
    RefSlicer* s = new RefSlicer();
    s->opts = static_cast<dsd_opts*>(calloc(1, sizeof(dsd_opts)));
    s->state = static_cast<dsd_state*>(calloc(1, sizeof(dsd_state)));
    dsd_opts* o = s->opts;
    dsd_state* st = s->state;
    o->ssize = 128;
    o->msize = 1024;
    o->audio_in_type = AUDIO_IN_PULSE; // anything but RTL / symbol-bin: no hook traffic, no replay overrides
    st->rf_mod = 0;
    st->synctype = synctype;
    st->lastsynctype = synctype;
    st->center = 0.0f;
    st->min = -30000.0f;
    st->max = 30000.0f;
    st->lmid = -20000.0f;
    st->umid = 20000.0f;
    st->minref = -24000.0f;
    st->maxref = 24000.0f;
    const int cap = (int)(sizeof(st->minbuf) / sizeof(st->minbuf[0]));
    for (int i = 0; i < cap; i++) {
        st->minbuf[i] = st->min;
        st->maxbuf[i] = st->max;
    }
    st->midx = 0;
    dsd_state_invalidate_minmax_sums(st);
    st->dibit_buf = static_cast<int*>(calloc(1000000, sizeof(int)));
    st->dibit_buf_p = st->dibit_buf + 200;
    st->dmr_payload_buf = static_cast<int*>(calloc(1000000, sizeof(int)));
    st->dmr_payload_p = st->dmr_payload_buf + 200;
    st->dmr_soft_buf = static_cast<dsd_dibit_soft_t*>(calloc(1000000, sizeof(dsd_dibit_soft_t)));
    st->dmr_soft_p = st->dmr_soft_buf + 200;
    return s;
}

void
refh_slicer_destroy(void* h) {
    RefSlicer* s = static_cast<RefSlicer*>(h);
    free(s->state->dibit_buf);
    free(s->state->dmr_payload_buf);
    free(s->state->dmr_soft_buf);
    free(s->state);
    free(s->opts);
    delete s;
}

// Feed n symbols through getDibitSoft(): out records = {dibit, reliability, llr0, llr1} as int32 x4 per symbol,
// thresholds = {center, umid, lmid, max, min} after each symbol.
void
refh_slicer_run(void* h, const float* symbols, long n, int* out4, float* thr5) {
    RefSlicer* s = static_cast<RefSlicer*>(h);
    g_sym_src = symbols;
    g_sym_n = n;
    g_sym_pos = 0;
    for (long i = 0; i < n; i++) {
        dsd_dibit_soft_t soft;
        const int d = getDibitSoft(s->opts, s->state, &soft);
        out4[4 * i] = d;
        out4[4 * i + 1] = soft.reliability;
        out4[4 * i + 2] = soft.llr[0];
        out4[4 * i + 3] = soft.llr[1];
        if (thr5) {
            thr5[5 * i] = s->state->center;
            thr5[5 * i + 1] = s->state->umid;
            thr5[5 * i + 2] = s->state->lmid;
            thr5[5 * i + 3] = s->state->max;
            thr5[5 * i + 4] = s->state->min;
        }
    }
    g_sym_src = nullptr;
}

// ---- the same digitize() path behind the CQPSK demodulator (symbol-rate output) ---------------------------------------------------
// rf_mod 1, RTL input, the metrics hooks reporting "CQPSK + timing active" and a caller-given CQPSK SNR
// (dsd_rtl_stream_metrics_hooks_set, include/dsd-neo/runtime/rtl_stream_metrics_hooks.h): select_four_level_dibit() then takes the
// fixed CQPSK slice, the rotation map and the CQPSK ideals (src/core/frames/dsd_dibit.c:951-1000,658-721).  The harness getSymbol()
// above hands out the caller's symbols; with g_cq_fast_path set it first does what dsd_symbol.c's symbol-rate fast path does before
// it returns a symbol - apply_rtl_symbol_thresholds(state, 4), constants only (src/dsp/dsd_symbol.c:744-765,1594).
#include <dsd-neo/runtime/rtl_stream_metrics_hooks.h>

namespace {
double g_cq_snr = -100.0;
int
cq_status_hook(int* out_cqpsk, int* out_timing) {
    if (out_cqpsk) {
        *out_cqpsk = 1;
    }
    if (out_timing) {
        *out_timing = 1;
    }
    return 0;
}
double
cq_snr_hook(void) {
    return g_cq_snr;
}
} // namespace

void*
refh_cq_slicer_create(int synctype, int map_idx, double snr_db) {
    RefSlicer* s = static_cast<RefSlicer*>(refh_slicer_create(synctype));
    dsd_opts* o = s->opts;
    dsd_state* st = s->state;
    o->audio_in_type = AUDIO_IN_RTL;
    o->frame_p25p1 = 1;
    o->frame_p25p2 = 1;
    st->rf_mod = 1;
    st->p25_cqpsk_dibit_map_idx = (uint8_t)map_idx;
    // initState() values (src/core/util/dsd_init.c:519-539), not the FSK-discriminator reset
    st->center = 0.0f;
    st->min = -15000.0f;
    st->max = 15000.0f;
    st->lmid = 0.0f;
    st->umid = 0.0f;
    const int cap = (int)(sizeof(st->minbuf) / sizeof(st->minbuf[0]));
    for (int i = 0; i < cap; i++) {
        st->minbuf[i] = -15000.0f;
        st->maxbuf[i] = 15000.0f;
    }
    st->midx = 0;
    dsd_state_invalidate_minmax_sums(st);
    g_cq_snr = snr_db;
    dsd_rtl_stream_metrics_hooks hooks;
    memset(&hooks, 0, sizeof(hooks));
    hooks.cqpsk_status = cq_status_hook;
    hooks.snr_cqpsk_db = cq_snr_hook;
    dsd_rtl_stream_metrics_hooks_set(&hooks);
    return s;
}

void
refh_cq_slicer_destroy(void* h) {
    dsd_rtl_stream_metrics_hooks none;
    memset(&none, 0, sizeof(none));
    dsd_rtl_stream_metrics_hooks_set(&none);
    refh_slicer_destroy(h);
}

// n symbols through getDibitSoft() in that context: out4 / thr5 as refh_slicer_run
void
refh_cq_slicer_run(void* h, const float* symbols, long n, int* out4, float* thr5) {
    g_cq_fast_path = 1;
    refh_slicer_run(h, symbols, n, out4, thr5);
    g_cq_fast_path = 0;
}

// P25 matched (de-emphasis) filter exactly as the symbolizer applies it per sample: p25_filter(sample, sps)
// (src/dsp/dsd_filters.c:368).  reset != 0 clears the process-global filter memory first.
void
refh_p25_filter_run(const float* in, long n, int sps, int reset, float* out) {
    if (reset) {
        init_rrc_filter_memory();
    }
    for (long i = 0; i < n; i++) {
        out[i] = p25_filter(in[i], sps);
    }
}

// the DMR (RRC alpha 0.7) and NXDN48 matched filters on the same terms (src/dsd_filters.c:348-356); which: 1 DMR, 2 NXDN
void
refh_fsk4_filter_run(int which, const float* in, long n, int sps, int reset, float* out) {
    if (reset) {
        init_rrc_filter_memory();
    }
    for (long i = 0; i < n; i++) {
        out[i] = which == 1 ? dmr_filter(in[i], sps) : nxdn_filter(in[i], sps);
    }
}

// dmr_compute_reliability on the slicer harness state is not needed: it is c4fm_reliability_from_thresholds + the SNR
// weight, both already exercised through getDibitSoft (refh_slicer_*).

extern "C" void agf(const dsd_opts* opts, dsd_state* state, float samp[160], int slot); // include/dsd-neo/core/audio.h:87
extern "C" void agsm(dsd_opts* opts, dsd_state* state, short* input, int len);            // include/dsd-neo/core/audio.h:89
extern "C" void init_audio_filters(dsd_state* state, int sample_rate_hz);                 // include/dsd-neo/core/audio_filters.h:22
extern "C" void hpf_dL(dsd_state* state, short* input, int len);                          // include/dsd-neo/core/audio_filters.h:35
// agf(): the float-path auto gain applied to every synthesized 160-sample voice frame (src/core/audio/gain.c:119-139), frames
// of one talk path in order; aout_gain is carried in and out like state->aout_gain (slot 0).
void
refh_agf_run(float* samp, int n_frames, float audio_gain, int algid_0x21, float* aout_gain_io) {
    dsd_opts* o = static_cast<dsd_opts*>(calloc(1, sizeof(dsd_opts)));
    dsd_state* st = static_cast<dsd_state*>(calloc(1, sizeof(dsd_state)));
    o->audio_gain = audio_gain;
    st->payload_algid = algid_0x21 ? 0x21 : 0;
    st->aout_gain = *aout_gain_io;
    for (int f = 0; f < n_frames; f++) {
        agf(o, st, samp + (size_t)f * 160, 0);
    }
    *aout_gain_io = st->aout_gain;
    free(o);
    free(st);
}

// hpf_dL() (src/core/util/dsd_misc.c:516-522) after init_audio_filters(state, 48000) and agsm() (src/core/audio/gain.c:143-184)
// on consecutive 160-sample frames of one talk path, each stage switchable; returns the last agsm coefficient.
float
refh_s16_post_run(short* pcm, int n_frames, int use_hpf_d, int use_agsm) {
    dsd_opts* o = static_cast<dsd_opts*>(calloc(1, sizeof(dsd_opts)));
    dsd_state* st = static_cast<dsd_state*>(calloc(1, sizeof(dsd_state)));
    init_audio_filters(st, 48000);
    for (int f = 0; f < n_frames; f++) {
        if (use_hpf_d) {
            hpf_dL(st, pcm + (size_t)f * 160, 160);
        }
        if (use_agsm) {
            agsm(o, st, pcm + (size_t)f * 160, 160);
        }
    }
    const float g = st->aout_gainA;
    free(o);
    free(st);
    return g;
}

int
refh_sync_p25p1_pos(void) {
    return DSD_SYNC_P25P1_POS;
}
int
refh_sync_p25p1_neg(void) {
    return DSD_SYNC_P25P1_NEG;
}

} // extern "C"

// ---- sync-time threshold calibration (src/dsp/sync_calibration.c) and the hunting level window
// ---- (src/dsp/frame_sync_level.c), driven on the same dsd_state the slicer harness owns -------------------------
#include <dsd-neo/dsp/sync_calibration.h>
#include "frame_sync_level.h" /* src/dsp/frame_sync_level.h (private header, on the -I path) */
#include <dsd-neo/core/vocoder.h>
#include <dsd-neo/io/iq_replay.h>
#include <dsd-neo/protocol/p25/p25p1_const.h>

extern "C" {
// Push `n` symbols (oldest first) through dsd_symbol_history_push(), then run the reference's
// dsd_sync_warm_start_thresholds_outer_only(opts, state, sync_len).  out7 = center, umid, lmid, max, min, maxref,
// minref after the call; returns the reference's result code.
int
refh_slicer_warm_start(void* h, const float* symbols_oldest_first, int n, int sync_len, float out7[7]) {
    RefSlicer* s = static_cast<RefSlicer*>(h);
    dsd_state* st = s->state;
    if (st->symbol_history == nullptr) {
        st->symbol_history_size = 2048;
        st->symbol_history = static_cast<float*>(calloc((size_t)st->symbol_history_size, sizeof(float)));
        st->symbol_history_head = 0;
        st->symbol_history_count = 0;
    }
    for (int i = 0; i < n; i++) {
        dsd_symbol_history_push(st, symbols_oldest_first[i]);
    }
    const int rc = (int)dsd_sync_warm_start_thresholds_outer_only(s->opts, st, sync_len);
    out7[0] = st->center;
    out7[1] = st->umid;
    out7[2] = st->lmid;
    out7[3] = st->max;
    out7[4] = st->min;
    out7[5] = st->maxref;
    out7[6] = st->minref;
    return rc;
}

// frame_sync_set_basic_lock()'s max/min averaging is two plain lines in a file that cannot be built here; the level
// estimate feeding it can:
void
refh_level_estimate(const float* sorted, int count, float* lo, float* hi) {
    dsd_frame_sync_estimate_sorted_window_levels(sorted, count, lo, hi);
}
} // extern "C"

// ---- CQPSK: full_demod() with cqpsk_enable (channel LPF -> RMS AGC -> FLL -> Gardner -> diff phasor -> Costas ->
// ---- phase extractor), field set after rtl_demod_init_for_mode()'s CQPSK branch ---------------------------------
extern "C" {
void*
refh_cqpsk_create(int rate_hz, int symbol_rate_hz, int lpf_profile, int lpf_enable) {
    void* mem = dsd_neo_aligned_malloc(sizeof(demod_state));
    if (!mem) {
        return nullptr;
    }
    demod_state* d = new (mem) demod_state();
    d->rate_in = rate_hz;
    d->rate_out = rate_hz;
    d->mode_demod = &qpsk_differential_demod;
    d->output_kind = DSD_DEMOD_OUTPUT_SYMBOL_CQPSK;
    d->cqpsk_enable = 1;
    d->symbol_rate_hz = symbol_rate_hz;
    d->symbol_levels = 4;
    d->ted_sps = (symbol_rate_hz > 0) ? rate_hz / symbol_rate_hz : 5;
    d->sps_is_integer = 1;
    d->channel_lpf_enable = lpf_enable;
    d->channel_lpf_profile = lpf_profile;
    d->squelch_env = 1.0f;
    d->squelch_gate_open = 1;
    return new RefFrontEnd{d};
}

// Whole capture of already-widened complex floats, `block_len` complex samples per full_demod() call.
// Returns the number of symbols written.
long
refh_cqpsk_run_f32(void* h, const float* iq, long n_complex, int block_len, float* out, long out_cap) {
    RefFrontEnd* fe = static_cast<RefFrontEnd*>(h);
    demod_state* d = fe->d;
    long done = 0, written = 0;
    while (done < n_complex) {
        long n = n_complex - done;
        if (n > block_len) {
            n = block_len;
        }
        std::memcpy(d->input_cb_buf, iq + 2 * done, (size_t)n * 2 * sizeof(float));
        d->lowpassed = d->input_cb_buf;
        d->lp_len = (int)(2 * n);
        full_demod(d);
        for (int k = 0; k < d->result_len && written < out_cap; k++) {
            out[written++] = d->result[k];
        }
        done += n;
    }
    return written;
}

// out8 = {agc_avg, fll.freq, fll.phase, costas.phase, costas.freq, costas.error_smooth, ted.mu, ted.omega}
void
refh_cqpsk_get_state(void* h, float out8[8]) {
    const demod_state* d = static_cast<RefFrontEnd*>(h)->d;
    out8[0] = d->cqpsk_agc_avg;
    out8[1] = d->fll_band_edge_state.freq;
    out8[2] = d->fll_band_edge_state.phase;
    out8[3] = d->costas_state.phase;
    out8[4] = d->costas_state.freq;
    out8[5] = d->costas_state.error_smooth;
    out8[6] = d->ted_state.mu;
    out8[7] = d->ted_state.omega;
}

// process_IMBE()'s store loop (src/protocol/p25/phase1/p25p1_ldu.c:89-112) on caller-supplied dibits: the tables and the
// soft-bit conversion are the reference's own (p25p1_const.h, vocoder.h); getDibitSoft() is replaced by array reads.
int
refh_imbe_deinterleave(const uint8_t* dibits, const int16_t* llr0, const int16_t* llr1, int status_count,
                       char fr[8][23], uint8_t soft[8][23][2], int* status_count_out) {
    std::memset(fr, 0, 8 * 23);
    std::memset(soft, 0, 8 * 23 * 2);
    int pos = 0;
    for (int j = 0; j < 72; j++) {
        if (status_count == 35) {
            pos++;
            status_count = 1;
        } else {
            status_count++;
        }
        const int dibit = dibits[pos];
        const int w = p25p1_imbe_interleave_w[j], x = p25p1_imbe_interleave_x[j];
        const int y = p25p1_imbe_interleave_y[j], z = p25p1_imbe_interleave_z[j];
        fr[w][x] = (char)(1 & (dibit >> 1));
        fr[y][z] = (char)(1 & dibit);
        const dsd_vocoder_soft_bit a = dsd_vocoder_soft_bit_from_hard_llr(fr[w][x], llr0[pos]);
        const dsd_vocoder_soft_bit b = dsd_vocoder_soft_bit_from_hard_llr(fr[y][z], llr1[pos]);
        soft[w][x][0] = a.bit;
        soft[w][x][1] = a.reliability;
        soft[y][z][0] = b.bit;
        soft[y][z][1] = b.reliability;
        pos++;
    }
    *status_count_out = status_count;
    return pos;
}

// ---- I/Q capture metadata / replay reader (src/io/iq/iq_replay.c), flattened for ctypes ------------------------------
// out[] = version, format, sample_rate, base_decimation, post_downsample, demod_rate, retune_count, event_count,
//         center, capture_center, data_bytes, drops, drop_blocks, ring_drops, ppm, gain, bw, offset_tuning, fs4,
//         historical_cu8_two_pass, muted_excluded, contains_retunes, size_limit_reached
int
refh_iq_meta(const char* path, int for_replay, int64_t out[23], char* stage64, char* data_path2048, char* err256) {
    dsd_iq_replay_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    err256[0] = 0;
    int rc = for_replay ? dsd_iq_replay_open(path, &cfg, nullptr, err256, 256)
                        : dsd_iq_replay_read_metadata(path, &cfg, err256, 256);
    if (rc != 0) {
        return rc;
    }
    const int64_t v[23] = {cfg.metadata_version, (int64_t)cfg.format, cfg.sample_rate_hz, cfg.base_decimation,
                           cfg.post_downsample, cfg.demod_rate_hz, cfg.capture_retune_count, cfg.event_count,
                           (int64_t)cfg.center_frequency_hz, (int64_t)cfg.capture_center_frequency_hz,
                           (int64_t)cfg.data_bytes, (int64_t)cfg.capture_drops, (int64_t)cfg.capture_drop_blocks,
                           (int64_t)cfg.input_ring_drops, cfg.ppm, cfg.tuner_gain_tenth_db, cfg.rtl_dsp_bw_khz,
                           cfg.offset_tuning_enabled, cfg.fs4_shift_enabled, cfg.historical_cu8_two_pass,
                           cfg.muted_bytes_excluded, cfg.contains_retunes, cfg.size_limit_reached};
    std::memcpy(out, v, sizeof(v));
    std::strncpy(stage64, cfg.capture_stage, 63);
    stage64[63] = 0;
    std::strncpy(data_path2048, cfg.data_path, 2047);
    data_path2048[2047] = 0;
    dsd_iq_replay_config_clear(&cfg);
    return 0;
}

// event i of a capture: out[] = kind, byte_offset, duration_bytes, center, capture_center, sample_rate
int
refh_iq_event(const char* path, unsigned index, int64_t out[6]) {
    dsd_iq_replay_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    char err[256];
    int rc = dsd_iq_replay_read_metadata(path, &cfg, err, sizeof(err));
    if (rc != 0) {
        return rc;
    }
    if (index >= cfg.event_count) {
        dsd_iq_replay_config_clear(&cfg);
        return -100;
    }
    const dsd_iq_event& e = cfg.events[index];
    out[0] = (int64_t)e.kind;
    out[1] = (int64_t)e.byte_offset;
    out[2] = (int64_t)e.duration_bytes;
    out[3] = (int64_t)e.center_frequency_hz;
    out[4] = (int64_t)e.capture_center_frequency_hz;
    out[5] = e.sample_rate_hz;
    dsd_iq_replay_config_clear(&cfg);
    return 0;
}

// read a whole capture through dsd_iq_replay_open / _read in `chunk`-byte requests; returns bytes delivered or < 0
long
refh_iq_read_all(const char* path, uint8_t* buf, long cap, int chunk) {
    dsd_iq_replay_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    dsd_iq_replay_source* src = nullptr;
    char err[256];
    int rc = dsd_iq_replay_open(path, &cfg, &src, err, sizeof(err));
    if (rc != 0) {
        return rc;
    }
    long total = 0;
    while (total < cap) {
        size_t got = 0;
        size_t want = (size_t)(cap - total < chunk ? cap - total : chunk);
        rc = dsd_iq_replay_read(src, buf + total, want, &got);
        if (rc != 0 || got == 0) {
            break;
        }
        total += (long)got;
    }
    dsd_iq_replay_close(src);
    dsd_iq_replay_config_clear(&cfg);
    return total;
}
} // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The AMBE 3600x2450 dibit-to-frame interleave schedule shared by the DMR / NXDN / YSF voice paths
// (include/dsd-neo/core/ambe_interleave.h:25-38), exposed entry by entry so that tools/gen_tables_ambe.py can measure it.
#include <dsd-neo/core/ambe_interleave.h>
extern "C" int
refh_ambe2450_map(int i, int out4[4]) {
    if (i < 0 || i >= DSD_AMBE_2450_DIBITS) {
        return -1;
    }
    out4[0] = dsd_ambe_2450_dibit_map[i].high_row;
    out4[1] = dsd_ambe_2450_dibit_map[i].high_col;
    out4[2] = dsd_ambe_2450_dibit_map[i].low_row;
    out4[3] = dsd_ambe_2450_dibit_map[i].low_col;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// P25 Phase 2 FACCH / SACCH burst decode: the gather loops and the ranked-erasure retry of p25p2_process_facchc() /
// process_SACCHs() -> p25p2_decode_facch_ranked() / _sacch_ranked() (src/protocol/p25/phase2/p25p2_frame.c:395-470,473-495,
// 652-700; static there, so their few lines are repeated here) around the reference's own ez_rs28_facch / ez_rs28_sacch
// (src/fec/ez.cpp) and p25p2_facch_soft_erasures / p25p2_sacch_soft_erasures (src/protocol/p25/phase2/p25p2_soft.c, compiled in
// place).  p2llr / p2xllr are the frame layer's soft-metric buffers p25p2_soft.c reads; the reference's own unit test
// (tests/protocol/p25/test_p25p2_soft_erasure.c:24-26) defines them the same way.
#include <dsd-neo/fec/ez.h>
#include <dsd-neo/protocol/p25/p25p2_soft.h>
int16_t p2llr[1400] = {0};
int16_t p2xllr[1400] = {0};
extern "C" int
refh_p25p2_xcch(int kind /* 0 FACCH, 1 SACCH */, const uint8_t* bits360, const int16_t* llr360, int ts_counter, uint8_t* payload_out,
                int* used_dynamic) {
    static const int facch_fixed[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 54, 55, 56, 57, 58, 59, 60, 61, 62};
    static const int sacch_fixed[11] = {0, 1, 2, 3, 4, 57, 58, 59, 60, 61, 62};
    if (ts_counter < 0 || ts_counter > 3) {
        return -3;
    }
    for (int i = 0; i < 1400; i++) {
        p2llr[i] = 0;
    }
    for (int i = 0; i < 360 && i + 360 * ts_counter < 1400; i++) {
        p2llr[i + 360 * ts_counter] = llr360[i];
    }
    int payload[180], parity[132], orig_payload[180], orig_parity[132];
    int n_pl, n_pa, n_fixed, max_add;
    const int* fixed;
    if (kind == 0) {
        for (int i = 0; i < 72; i++) {
            payload[i] = bits360[i + 2];
        }
        for (int i = 0; i < 62; i++) {
            payload[i + 72] = bits360[i + 76];
        }
        for (int i = 0; i < 22; i++) {
            payload[i + 134] = bits360[i + 180];
        }
        for (int i = 0; i < 42; i++) {
            parity[i] = bits360[i + 202];
        }
        for (int i = 0; i < 72; i++) {
            parity[i + 42] = bits360[i + 246];
        }
        n_pl = 156, n_pa = 114, n_fixed = 18, max_add = 10, fixed = facch_fixed;
    } else {
        for (int i = 0; i < 72; i++) { // process_SACCHs(): 72 + 108 payload bits, 60 + 72 parity bits
            payload[i] = bits360[i + 2];
        }
        for (int i = 0; i < 108; i++) {
            payload[i + 72] = bits360[i + 76];
        }
        for (int i = 0; i < 60; i++) {
            parity[i] = bits360[i + 184];
        }
        for (int i = 0; i < 72; i++) {
            parity[i + 60] = bits360[i + 246];
        }
        n_pl = 180, n_pa = 132, n_fixed = 11, max_add = 16, fixed = sacch_fixed;
    }
    for (int i = 0; i < n_pl; i++) {
        orig_payload[i] = payload[i];
    }
    for (int i = 0; i < n_pa; i++) {
        orig_parity[i] = parity[i];
    }
    *used_dynamic = 0;
    int ec = kind == 0 ? ez_rs28_facch(payload, parity, fixed, n_fixed) : ez_rs28_sacch(payload, parity, fixed, n_fixed);
    if (ec < 0) {
        int erasures[28] = {0};
        for (int i = 0; i < n_fixed; i++) {
            erasures[i] = fixed[i];
        }
        const int n_er = kind == 0 ? p25p2_facch_soft_erasures(ts_counter, 0, erasures, n_fixed, max_add)
                                   : p25p2_sacch_soft_erasures(ts_counter, 0, erasures, n_fixed, max_add);
        for (int n = n_fixed + 1; n <= n_er; n++) {
            for (int i = 0; i < n_pl; i++) {
                payload[i] = orig_payload[i];
            }
            for (int i = 0; i < n_pa; i++) {
                parity[i] = orig_parity[i];
            }
            ec = kind == 0 ? ez_rs28_facch(payload, parity, erasures, n) : ez_rs28_sacch(payload, parity, erasures, n);
            if (ec >= 0) {
                *used_dynamic = 1;
                break;
            }
        }
        if (ec < 0) {
            for (int i = 0; i < n_pl; i++) {
                payload[i] = orig_payload[i];
            }
        }
    }
    for (int i = 0; i < n_pl; i++) {
        payload_out[i] = (uint8_t)payload[i];
    }
    return ec;
}

// P25 Phase 2 ESS: p25p2_ess_decode_with_soft_erasures() (p25p2_frame.c:1061-1091; static there) around the reference's ez_rs28_ess and
// p25p2_ess_soft_erasures_ranked (p25p2_soft.c, compiled in place).  Returns accepted; *ec as the reference leaves it.
extern "C" int
refh_p25p2_ess(const uint8_t* payload_bits96, const int16_t* payload_llr96, const uint8_t* parity_bits168, const int16_t* parity_llr168,
               uint8_t* payload_out96, int* ec) {
    int payload[96], parity[168], op[96], oq[168];
    for (int i = 0; i < 96; i++) {
        payload[i] = op[i] = payload_bits96[i] & 1;
    }
    for (int i = 0; i < 168; i++) {
        parity[i] = oq[i] = parity_bits168[i] & 1;
    }
    int accepted = 0;
    *ec = ez_rs28_ess(payload, parity, NULL, 0);
    if (*ec >= 0 && *ec < 15) {
        accepted = 1;
    } else {
        for (int i = 0; i < 96; i++) {
            payload[i] = op[i];
        }
        int erasures[44];
        const int n_er = p25p2_ess_soft_erasures_ranked(payload_llr96, parity_llr168, erasures, 28);
        for (int n = 1; n <= n_er && !accepted; n++) {
            for (int i = 0; i < 96; i++) {
                payload[i] = op[i];
            }
            for (int i = 0; i < 168; i++) {
                parity[i] = oq[i];
            }
            *ec = ez_rs28_ess(payload, parity, erasures, n);
            if (*ec >= 0) {
                accepted = 1;
            }
        }
        if (!accepted) {
            for (int i = 0; i < 96; i++) {
                payload[i] = op[i];
            }
        }
    }
    for (int i = 0; i < 96; i++) {
        payload_out96[i] = (uint8_t)payload[i];
    }
    return accepted;
}
