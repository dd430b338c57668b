/* ddn_device.h — kernel argument structs shared by the .hip kernels and the C-ABI host code. */
#ifndef DDN_DEVICE_H
#define DDN_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ddn_internal.h"

/* Experiment knobs.  The product build reads nothing from the environment: DDN_EXP_ENV(name) is NULL there, so every "if the variable
 * is set" branch folds away and the library's schedule depends on its arguments alone.  The timing tools build a second library
 * with -DDDN_EXPERIMENTS (tools/build_variant.sh exp), in which the knobs DESIGN / profiles/README.md name are live. */
#ifdef DDN_EXPERIMENTS
#include <stdlib.h>
#define DDN_EXP_ENV(name) getenv(name)
#else
#define DDN_EXP_ENV(name) ((const char*)0)
#endif

#define DDN_TILE 256        /* time tile (samples) of the fused front-end kernel */
#define DDN_DEFAULT_GROUP 8 /* channels per workgroup */
#define DDN_CARRY_LEN 72 /* >= DDN_MAX_CENTER widened samples of FIR look-back per channel */

typedef struct DdnFskState { /* == the fields of dsd_fsk_modem_state the path carries */
    float prev_i, prev_q;
    int have_prev;
    float dc_est, peak_est;
} DdnFskState;

typedef float ddn_f2 __attribute__((ext_vector_type(2)));

typedef struct DdnIqCondConfig { /* optional IQ conditioning switches of struct demod_state (row a5) + squelch gate */
    int dc_enable, dc_shift, bal_enable, squelch_on;
    float bal_thr, bal_ema_a, squelch_level;
} DdnIqCondConfig;
typedef struct DdnIqCondState { /* iq_dc_avg_r/i, iqbal_alpha_ema_r/i */
    float dc_r, dc_i, er, ei;
} DdnIqCondState;

typedef struct DdnFusedArgs {
    const void* in;       /* [B][ch_stride] complex samples (cu8 pairs or float pairs) */
    float* out;           /* [B][out_stride] discriminator samples */
    const ddn_f2* carry;  /* [B][DDN_CARRY_LEN] FIR look-back from the previous call */
    ddn_f2* carry_out;    /* same array when the kernel itself refreshes it at the end (n >= DDN_CARRY_LEN), else NULL */
    DdnFskState* state;   /* [B] */
    const float* taps_dev; /* [taps_len] channel LPF taps (first half + centre are read) */
    size_t ch_stride;
    size_t out_stride;
    long n;               /* complex samples per channel in this call */
    long n_tiles;         /* n_blocks * tiles_per_block */
    int n_channels;
    int in_fmt;
    int block_len;
    int tiles_per_block;
    int center;
    int squelch_on;
    float squelch_level;
    long long* dbg_out; /* optional: per-wave phase timings of workgroup 0 (DDN_DBG bit 6) */
    int dbg; /* timing experiments only (DDN_DBG env): 1 skip recurrences, 2 skip filter, 4 skip finish, 8 skip staging */
    /* (round 6) segments: the batch's channel index is cut into up to three runs [seg_first1, seg_first2) ... with an input array, an
     * output array and a tap set of their own (same tap count) - three protocol groups of a mixed batch in ONE launch, so the grid is
     * ceil(total / G) workgroups however the groups' sizes fall (a workgroup may hold channels of two groups).  n_seg <= 1: in / out /
     * taps_dev above, as before.  carry / state are the batch's arrays, indexed by the batch's channel index either way. */
    int n_seg, seg_first1, seg_first2;
    const void *seg_in0, *seg_in1, *seg_in2;
    float *seg_out0, *seg_out1, *seg_out2;
    const float *seg_taps1, *seg_taps2; /* (segment 0's are taps_dev) */
} DdnFusedArgs;

#define DDN_TED_DL 100 /* == TED_DL_SIZE (include/dsd-neo/dsp/ted.h:19) */
typedef struct DdnTedState { /* == the carried fields of ted_state_t */
    float mu, omega, omega_mid, omega_min, omega_max, omega_rel;
    float last_r, last_j, lock_accum;
    int lock_count, dl_index, twice_sps, sps;
} DdnTedState;

typedef struct DdnSlicerState { /* per-channel slicer words of dsd_state the P25p1 path carries */
    float center, umid, lmid, max, min;
    int sidx, midx, sums_valid;
    double min_sum, max_sum;
} DdnSlicerState;

typedef struct DdnRxConfig { /* fixed-protocol P25p1 receive loop (ddn_rx.hip) */
    int out_rate, sym_rate, lock_symbols, use_filter;
    int dbg; /* profiling only (env DDN_RX_DBG): 1 = skip symbol commit, 2 = skip record stores, 4 = skip sample loop body */
    int handlers;      /* 1 = the reference's per-DUID handlers decide the in-frame length (lock_symbols ignored) */
    int nid_threshold; /* p25p1_get_erasure_threshold() */
    int max_events;    /* capacity of the per-channel event list */
    int32_t* event_data; /* [B][max_events][4] what each decision decoded (NULL = not wanted): NID {status, nac, duid, errors}, TSBK /
                            PDU header block {12 bytes as three words, crc good | candidate << 8 | block << 16} */
} DdnRxConfig;

typedef struct DdnP25HState { /* per-channel words of the P25p1 handlers (ddn_p25h_dev.h) carried across calls */
    int phase, block, end, skipdibit, nac, p2_cc, r0, r1;
} DdnP25HState;

typedef struct DdnRxState { /* per-channel words of dsd_state / frame_sync_runtime_ctx the P25p1 loop carries */
    double min_sum, max_sum;
    long long filt_start, n_abs; /* absolute sample index of the filter's first sample / of the next input sample */
    float center, umid, lmid, max, min, maxref, minref;
    float fill_min, fill_max; /* value the extrema rings were last refilled with */
    float sum, lastsample, lmin, lmax;
    int sidx, midx, since_fill; /* since_fill: ring pushes since the refill, saturating at 1024 */
    int sps_accum, jitter, in_symbol, span, centre, i, count;
    int filter_on, have_sync, lock_left, lastsync; /* lastsync: 0 none, 1 +P25p1, 2 -P25p1 */
    int lidx, level_count, hist_count, shead, scount;
    uint32_t hist_bits;
    int hunt_pos;   /* rt.synctest_pos: symbols hunted since the hunt (re)started */
    int need_reset; /* noCarrier() ran: the next symbol start re-initialises timing and slicer */
    /* handler mode: hphase != 0 = a handler decision falls due when lock_left runs out (1 the NID, 2 a trellis block);
       hw = symbols written to the in-frame history ring; hn = symbols of the current phase; hnc = noCarrier() ran since the
       last NID (the handlers' NAC memory is cleared with it, engine.c:1889) */
    int hphase, hw, hn, hnc;
    int dbg_nreq;           /* timing experiments (cfg.dbg bit 65536): handler requests / cycles spent waiting for the answers */
    long long dbg_wait;
} DdnRxState;

/* ---- profile-driven 4-level FSK receive loop (DMR / NXDN48), ddn_rx4.hip ---- */
#define DDN_FSK4_MAX_PAT  20
#define DDN_FSK4_MAX_TAPS 135
#define DDN_FSK4_HIST     128 /* symbol / payload history kept per channel: 90 reachable + what the helper wave lags */
#define DDN_FSK4_PRE      90
typedef struct DdnFsk4Config {
    int out_rate, sym_rate, rf_mod, win_len, t_max, warm_len, n_pat;
    uint32_t pat_bits[DDN_FSK4_MAX_PAT];
    uint8_t pat_type[DDN_FSK4_MAX_PAT], pat_neg[DDN_FSK4_MAX_PAT], pat_class[DDN_FSK4_MAX_PAT];
    int confirm, dmr_window, redigitize, slow_type, use_filter, nt;
    int dbg; /* profiling only (env DDN_RX4_DBG) */
    int handlers;   /* 1 = the reference's handlers decide the in-frame length (ddn_fsk4h_dev.h) */
    int max_events; /* capacity of the per-channel event list */
    float* sync_thr; /* optional [B][max_sync][5]: {center, umid, lmid, max, min} as every accepted sync leaves them (after the warm
                        start) - what a frame decoder that works on soft symbols needs (M17 LSF: thresholds are static inside a frame) */
} DdnFsk4Config;
typedef struct DdnFsk4State {
    long long filt_start, n_abs;
    float center, umid, lmid, max, min, maxref, minref;
    float sum, lastsample, lmin, lmax;
    int sps_accum, jitter, in_symbol, span, centre, i, count;
    int filter_on, have_sync, lock_left, lastsync, cur_pat;
    int lidx, level_count, hist_count, shead, scount;
    uint32_t hist_bits;
    int hunt_pos, need_reset;
    int hmode, hidx; /* handler mode: phase (ddn_fsk4h_dev.h) and the index of the next dibit inside the burst / frame */
    int hlich;       /* NXDN: the LICH's high bits so far */
} DdnFsk4State;

typedef struct DdnCqpskState { /* per-channel words of demod_state the CQPSK chain carries besides ted_state_t */
    float agc_avg;                          /* cqpsk_agc_avg */
    float fll_phase, fll_freq;              /* fll_band_edge_state */
    int fll_idx;
    float diff_r, diff_j;                   /* cqpsk_diff_prev_r/j */
    float cos_phase, cos_freq, cos_err, cos_es, cos_alpha, cos_beta; /* costas_state */
    int cos_init;
} DdnCqpskState;

typedef struct DdnPuncture { /* puncture pattern of the K=5 decoder, expanded on the host */
    int p_len;               /* 0 = not punctured */
    int ones_total;
    uint8_t keep[64];
    uint8_t ones_before[64]; /* kept positions before pattern index r */
} DdnPuncture;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t ddn_dev_gardner(const void* in, long n, size_t in_stride, int n_channels, int sps, float ted_gain,
                           int symbol_rate_hz, long block_len, DdnTedState* state, float* dl_store, void* out,
                           size_t out_stride, int* out_count, hipStream_t st);
hipError_t ddn_dev_p25_slicer(const float* sym, long n, size_t sym_stride, int n_channels, int negative,
                              DdnSlicerState* state, float* sbuf_store, float* minring, float* maxring, uint8_t* rec,
                              size_t rec_stride, hipStream_t st);
hipError_t ddn_dev_p25_slicer_par(const float* sym, long n, size_t sym_stride, int n_channels, int negative,
                                  DdnSlicerState* state, float* sbuf_store, float* minring, float* maxring,
                                  float* scratch4, uint8_t* rec, size_t rec_stride, hipStream_t st);
hipError_t ddn_dev_p25_matched_filter(const float* in, long n, size_t stride, int n_channels, float* hist, float* out,
                                      hipStream_t st);
#define DDN_MAX_HB_PASSES 4
hipError_t ddn_dev_hb_decim2(const void* in, int in_fmt, long n_in, size_t in_stride, int block_in, int n_channels,
                             int taps_len, void* hist, void* out, size_t out_stride, hipStream_t st);
hipError_t ddn_dev_p25_matched_filter_only(const float* in, long n, size_t stride, int n_channels, const float* hist,
                                           float* out, hipStream_t st);
hipError_t ddn_dev_p25_filter_hist_update(const float* in, long n, size_t stride, int n_channels, float* hist,
                                          hipStream_t st);
int ddn_dev_p25_rx_fuses_filter(const DdnRxConfig* cfg, int channels_per_wave, int n_channels);
hipError_t ddn_dev_p25_rx(const float* raw, const float* filt, const float* prev_tail, float* fstale, long n, size_t stride,
                          int n_channels, const DdnRxConfig* cfg, DdnRxState* state, float* sbuf_store,
                          float* lbuf_store, float* shist_store, float* minring, float* maxring, uint8_t* rec,
                          uint8_t* flags, int32_t* counts, size_t max_sym, int channels_per_wave,
                          const int32_t* lock_cfg, DdnP25HState* hstate, float* hh_store, int32_t* events,
                          int32_t* n_events, hipStream_t st);
hipError_t ddn_dev_fsk4_matched_filter(int nt, const float* in, long n, size_t stride, int n_channels, const float* hist,
                                       float* out, hipStream_t st);
hipError_t ddn_dev_fsk4_filter_hist_update(int nt, const float* in, long n, size_t stride, int n_channels, float* hist,
                                           hipStream_t st);
hipError_t ddn_dev_fsk4_rx(const float* raw, const float* filt, const float* prev_tail, float* fstale, const float* taps,
                           long n, size_t stride, int n_channels, const DdnFsk4Config* cfg, DdnFsk4State* state,
                           float* lbuf_store, float* shist_store, uint8_t* phist_store, uint8_t* rhist_store, uint8_t* rec,
                           uint8_t* flags, uint8_t* pay, int32_t* counts, size_t max_sym, const int32_t* lock4,
                           int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre, uint8_t* pre_rel, int32_t* n_sync,
                           int max_sync, int channels_per_wave, int samples_per_symbol, int protocol, int handlers,
                           int32_t* hwords, uint8_t* hpay, int32_t* events, int32_t* n_events, hipStream_t st);
hipError_t ddn_dev_dmr_burst_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                                    const uint8_t* pre, const int32_t* n_sync, int n_channels, int max_sync, int inverted,
                                    uint8_t* slot_type, uint8_t* info, uint8_t* cach, uint8_t* valid, hipStream_t st);
hipError_t ddn_dev_ambe2450_deinterleave(const uint8_t* dibits, const uint8_t* reliab, int n, uint8_t* fr, uint8_t* rl,
                                         hipStream_t st);
hipError_t ddn_dev_nxdn_voice_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                                     const int32_t* n_sync, int max_sync, int n_channels, uint8_t* fr, uint8_t* rl,
                                     uint8_t* valid, hipStream_t st);
hipError_t ddn_dev_dmr_voice_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* burst_start,
                                    int max_bursts, int n_channels, int inverted, uint8_t* fr, uint8_t* rl, uint8_t* sync48,
                                    uint8_t* cach24, uint8_t* valid, hipStream_t st);
hipError_t ddn_dev_dmr_voice_select(const int32_t* events, const int32_t* n_events, int max_events, int carry, const int32_t* sync_pos,
                                    const int32_t* n_sync, int max_syncs, int n_channels, int max_bursts, int32_t* vstart,
                                    int32_t* vpre, int32_t* vn, const int32_t* out_pos, const int32_t* out_n, int max_out,
                                    const int32_t* n_new, hipStream_t st);
hipError_t ddn_dev_dmr_voice_gather_paths(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* vstart,
                                          const int32_t* vpre, const uint8_t* pre90, int max_bursts, int n_channels, int inverted,
                                          uint8_t* fr, uint8_t* skip3, const uint8_t* pre90_out, long split, hipStream_t st);
hipError_t ddn_dev_nxdn_frame_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                                     const int32_t* n_sync, int n_channels, int max_sync, uint8_t* lich, uint8_t* sacch_sym,
                                     uint8_t* sacch_rel, uint8_t* facch_sym, uint8_t* facch_rel, uint8_t* valid, hipStream_t st);
hipError_t ddn_dev_nxdn_crc(const uint8_t* bytes, int stride, int n, int kind, uint8_t* ok, hipStream_t st);
hipError_t ddn_dev_audio_s16(const float* pcm, int n_streams, int n_frames, float audio_gain, int use_hpf, int use_agsm, float coef,
                             int16_t* out, float* state, float* gain_a, hipStream_t st);
hipError_t ddn_dev_agf(float* pcm, int n_streams, int n_frames, float gain, float* aout_gain, hipStream_t st);
hipError_t ddn_dev_channel_lpf_c2c(const void* in, int in_fmt, long n, size_t in_stride, int block_len, int n_channels,
                                   const float* taps_dev, int taps_len, int has_zero_tap, void* hist, void* out,
                                   size_t out_stride, hipStream_t st);
hipError_t ddn_dev_cqpsk_agc_fll(const void* in, long n, size_t stride, int n_channels, int nt, float alpha, float beta,
                                 const float* d_fll_taps,
                                 DdnCqpskState* state, float* delay_store, void* out, hipStream_t st);
hipError_t ddn_dev_cqpsk_symbols(const void* sym, size_t stride, const int* counts, int n_channels, DdnCqpskState* state,
                                 float* out, size_t out_stride, hipStream_t st);
hipError_t ddn_dev_nid_decode(const uint8_t* bits63, const uint8_t* rel63, const int32_t* obs_nac, const uint8_t* parity,
                              const uint8_t* parity_rel, int threshold, int n, int32_t* out4, hipStream_t st);
hipError_t ddn_dev_golay24(uint8_t* data, const uint8_t* parity, int len, int n, uint8_t* status, int32_t* fixed,
                           hipStream_t st);
hipError_t ddn_dev_golay24_soft(uint8_t* data, const uint8_t* parity, const int32_t* reliab, int len, int n,
                                uint8_t* status, int32_t* fixed, hipStream_t st);
hipError_t ddn_dev_hamming_10_6_3_soft(const uint8_t* bits, const int32_t* reliab, int n, uint8_t* out, uint8_t* status,
                                       hipStream_t st);
hipError_t ddn_dev_isch_lookup(const uint64_t* words, const uint8_t* reliab40, int n, int32_t* out, hipStream_t st);
hipError_t ddn_dev_p25p2_mac_crc(int kind, const uint8_t* payload_bits, int n, uint8_t* crc12_ok, uint8_t* crc16_ok, hipStream_t st);
hipError_t ddn_dev_p25p2_ess(const uint8_t* payload_bits, const int16_t* payload_llr, const uint8_t* parity_bits, const int16_t* parity_llr,
                             int n, int threshold, uint8_t* work, int8_t* erasures28, uint8_t* n_total, int32_t* status,
                             uint8_t* used_dynamic, hipStream_t st);
hipError_t ddn_dev_p25p2_voice_unpack(const uint8_t* xbits360, const int16_t* xllr360, int n, int frame_count, uint8_t* fr, uint8_t* rl,
                                      hipStream_t st);
hipError_t ddn_dev_p25p2_scramble_bits(const uint64_t* seed44, int n, int bit_count, uint8_t* out, hipStream_t st);
hipError_t ddn_dev_p25p2_descramble(const uint8_t* bits, const int16_t* llr, const uint8_t* lbits4320, const int32_t* offset,
                                    const int32_t* seq_of, int n, int n_bits, int n_llr, uint8_t* xbits, int16_t* xllr, hipStream_t st);
hipError_t ddn_dev_p25p2_burst_fields(const uint8_t* bits360, const int16_t* llr360, int n, int threshold, int32_t* duid,
                                      uint64_t* isch_word, uint8_t* isch_rel, hipStream_t st);
hipError_t ddn_dev_p25p2_xcch(int kind, const uint8_t* bits360, const int16_t* llr360, int n, int threshold, uint8_t* payload_bits,
                              uint8_t* parity_bits, int8_t* erasures28, uint8_t* n_total, int32_t* status, uint8_t* used_dynamic,
                              hipStream_t st);
hipError_t ddn_dev_rs28(int kind, uint8_t* payload_bits, const uint8_t* parity_bits, const int8_t* erasures,
                        const uint8_t* n_erasures, int n, int32_t* status, hipStream_t st);
hipError_t ddn_dev_rs63_soft(uint8_t* data6, const uint8_t* parity6, const uint8_t* data_rel, const uint8_t* parity_rel,
                             int n_par, int n_data, int t, int threshold, int n, uint8_t* status, hipStream_t st);
hipError_t ddn_dev_rs63(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t, int n, uint8_t* status,
                        hipStream_t st);
hipError_t ddn_dev_hamming_10_6_3(uint8_t* bits10, int n, uint8_t* errs, hipStream_t st);
hipError_t ddn_dev_p25_crc16(const uint8_t* bytes, int item_bytes, int n, uint8_t* ok, hipStream_t st);
hipError_t ddn_dev_p25_lsd(uint8_t* bits16, const int16_t* llr16, int n, uint8_t* ok, hipStream_t st);
hipError_t ddn_dev_iq_cond_disc(const void* in, long n, size_t stride, int block_len, int n_channels,
                                const DdnIqCondConfig* cfg, DdnFskState* fsk, DdnIqCondState* cond, float* out,
                                size_t out_stride, hipStream_t st);
hipError_t ddn_dev_find_syncs(const uint8_t* flags, const int32_t* counts, int n_channels, size_t max_sym, int max_frames,
                              int32_t* sync_pos, int32_t* n_syncs, int32_t* dropped, hipStream_t st);
hipError_t ddn_dev_gather_fields(const uint8_t* rec, size_t max_sym, const int32_t* counts, const int32_t* sync_pos,
                                 const int32_t* n_syncs, int n_channels, int max_frames, const int32_t* offsets, int n_off,
                                 int max_off, uint8_t* bits, uint8_t* rel, int16_t* llr, int stride, int split_last,
                                 uint8_t* last_bit, uint8_t* last_rel, uint8_t* valid, uint8_t* dibits,
                                 uint8_t* dibit_rel, hipStream_t st);
hipError_t ddn_dev_rs_pack(const uint8_t* words, long n_slots, int n_words, int wstride, int n_data, uint8_t* data,
                           uint8_t* parity, hipStream_t st);
hipError_t ddn_dev_tdulc_rs_pack(const uint8_t* words, long n_slots, uint8_t* data, uint8_t* parity, hipStream_t st);
hipError_t ddn_dev_voice_index(const int32_t* sync_pos, const int32_t* n_syncs, const int32_t* nid4, const int32_t* counts, int n_channels,
                               int max_frames, int max_ldu, size_t max_sym, const int32_t* first9, const int32_t* status9,
                               int64_t* first, int32_t* status, int32_t* n_ldu, hipStream_t st);
hipError_t ddn_dev_imbe_index(const int32_t* sync_pos, const int32_t* n_syncs, int n_channels, int max_frames,
                              size_t max_sym, const int32_t* first9, const int32_t* status9, int64_t* first,
                              int32_t* status, hipStream_t st);
hipError_t ddn_dev_resample(const float* in, long n, size_t in_stride, int n_channels, float* hist, const float* taps,
                            int L, int M, int p0, long n_out, float* out, size_t out_stride, hipStream_t st);
hipError_t ddn_dev_imbe_deinterleave(const uint8_t* rec, long n_records, const int64_t* first, const int32_t* status_count,
                                     int n_frames, uint8_t* fr, uint8_t* soft, uint8_t* flags, int32_t* status_out,
                                     hipStream_t st);
hipError_t ddn_dev_r34_list(const uint8_t* dibits, const uint8_t* reliab, int n, int max_cand, uint8_t* backs,
                            uint32_t* cand, int32_t* count, hipStream_t st);
hipError_t ddn_dev_p25_mbf34_list(const int16_t* llr, int n, int max_cand, const uint8_t* wanted, uint8_t* cand24, int32_t* count,
                                  hipStream_t st);
hipError_t ddn_dev_p25_half_rate_list(const int16_t* llr, int n, int max_cand, uint32_t* cand, int32_t* count,
                                      hipStream_t st);
hipError_t ddn_dev_p25_half_rate(const int16_t* llr, int n, uint8_t* out, int32_t* metric, hipStream_t st);
hipError_t ddn_dev_r34(const uint8_t* dibits, const uint8_t* reliab, int n, uint8_t* out, hipStream_t st);
hipError_t ddn_dev_p25_half_rate_list_wanted(const int16_t* llr, int n, int max_cand, const uint8_t* wanted, uint32_t* cand, int32_t* count,
                                             hipStream_t st);
hipError_t ddn_dev_k5_nxdn_wanted(const uint8_t* sym, const uint8_t* rel, int n, int n_steps, int n_bits, uint16_t* metrics_io,
                                  uint8_t* out, int out_stride, const uint8_t* wanted, int wanted_div, hipStream_t st);
hipError_t ddn_dev_trellis_greedy_wanted(const uint8_t* src, int src_stride, size_t n, int result_len, uint8_t* out, int out_stride,
                                         const uint8_t* wanted, hipStream_t st);
hipError_t ddn_dev_r34_list_wanted(const uint8_t* dibits, const uint8_t* reliab, int n, int max_cand, const uint8_t* wanted, uint8_t* backs,
                                   uint32_t* cand, int32_t* count, hipStream_t st);
// the DMR chain's data-burst / embedded-signalling stages (ddn_dmr_data.hip, k_dmr_r34_pick in ddn_trellis.hip)
hipError_t ddn_dev_dmr_data_select(const int32_t* events, const int32_t* n_events, int max_events, int carry, int n_channels,
                                   int max_bursts, const int32_t* sync_pos, const int32_t* n_sync, int max_syncs, const int32_t* out_pos,
                                   const int32_t* out_n, int max_out, const int32_t* n_new, int32_t* dstart, uint8_t* dslot,
                                   int32_t* dpre, int32_t* dn, hipStream_t st);
hipError_t ddn_dev_dmr_data_gather(const uint8_t* rec, size_t max_sym, const int32_t* dstart, const int32_t* dpre, const uint8_t* pre90,
                                   const uint8_t* prel90, const uint8_t* pre90_out, const uint8_t* prel90_out, long split, int max_bursts,
                                   int n_channels, uint8_t* slot_type, uint8_t* info, uint8_t* td98, uint8_t* rel98, hipStream_t st);
hipError_t ddn_dev_dmr_data_prep(const int32_t* dstart, const uint8_t* slot_type, const uint8_t* st_ok, const uint8_t* pdu96, int n,
                                 uint8_t* type, uint8_t* bytes12, uint8_t* cw12, hipStream_t st);
hipError_t ddn_dev_dmr_data_finish(const uint8_t* type, const uint8_t* pdu96, const uint8_t* info, const uint8_t* cw12,
                                   const uint8_t* rs_result, int n, uint8_t* bytes12, uint8_t* crc, uint8_t* r34_wanted, hipStream_t st);
hipError_t ddn_dev_dmr_r34_pick(const uint8_t* td98, const uint8_t* rel98, const uint8_t* wanted, const uint8_t* hard18,
                                const uint8_t* soft18, const uint8_t* list24, const int32_t* list_n, int n, uint8_t* pool24,
                                int32_t* pool_n, uint8_t* unconf18, uint8_t* conf18, uint8_t* conf_crc, hipStream_t st);
hipError_t ddn_dev_dmr_emb_collect(const int32_t* events, const int32_t* n_events, int max_events, int carry, const uint8_t* rec,
                                   size_t max_sym, int n_channels, int max_lc, uint8_t* sig, uint8_t* in128, int32_t* lc_pos,
                                   int32_t* lc_n, hipStream_t st);
hipError_t ddn_dev_dmr_emb_finish(const uint8_t* out77, const int32_t* lc_pos, int n, uint8_t* ok, hipStream_t st);
hipError_t ddn_dev_k5_nxdn(const uint8_t* sym, const uint8_t* rel, int n, int n_steps, int n_bits, uint16_t* metrics_io,
                           uint8_t* out, int out_stride, hipStream_t st);
hipError_t ddn_dev_k5_m17(const uint16_t* in, int n, int in_len, int u_len, const DdnPuncture* pu, uint8_t* out,
                          int out_stride, uint32_t* cost, hipStream_t st);
hipError_t ddn_dev_k5_m17_wanted(const uint16_t* in, int n, int in_len, int u_len, const DdnPuncture* pu, uint8_t* out, int out_stride,
                                 uint32_t* cost, const uint8_t* wanted, hipStream_t st);
hipError_t ddn_dev_launch_fused(const DdnFusedArgs* a, const float* taps_host, int group, hipStream_t st);
hipError_t ddn_dev_launch_fused_ex(const DdnFusedArgs* a, bool has_zero, int group, hipStream_t st);
hipError_t ddn_dev_launch_carry(const void* in, int in_fmt, size_t ch_stride, long n, void* carry, int n_channels,
                                hipStream_t st);
hipError_t ddn_dev_zero(void* p, size_t bytes, hipStream_t st);
/* Frame-slot selection for the per-slot framer / FEC launches of a chain (ddn_api_chain.cpp): a launch made while a selection is
 * set walks `*count` frame slots taken from `list` (per_slot items each) instead of every slot - the work follows the frames of
 * that type, not the slot capacity; the other slots' outputs are left as they are.  Thread-local, set around the launches it is
 * meant for; list == NULL = every item. */
/* Workgroup size of the chains' small decode kernels (gathers, packs, block codes, selections): ONE wavefront.  In the pipelined chain
 * they run beside the next call's front-end kernel, whose ten waves per CU take every register of two of a CU's four SIMDs (3 x 168):
 * a workgroup of several waves needs room on every SIMD at once and waited for a front-end workgroup to end (~2 ms, measured);
 * single-wave workgroups go to the SIMDs that have room. */
#define DDN_WG 64
typedef struct DdnSel {
    const int32_t* list;  /* frame slots of the type, any order */
    const int32_t* count; /* device word: entries in list */
    int per_slot;         /* items of this launch per frame slot */
} DdnSel;
void ddn_sel_set(const int32_t* list, const int32_t* count);
void ddn_sel_clear(void);
DdnSel ddn_sel_for(int per_slot);
/* blocks for a launch of n_blocks_full blocks when a selection is active (the list decides the work, the grid only has to fill the
 * device) */
static inline unsigned
ddn_sel_grid(const DdnSel* sel, unsigned long n_blocks_full) {
    const unsigned long cap = 4096; /* x DDN_WG (64 threads) */
    return (unsigned)((sel->list && n_blocks_full > cap) ? cap : (n_blocks_full ? n_blocks_full : 1));
}
hipError_t ddn_dev_chain_pcm_compact(const int32_t* result5, const float* pcm, int n_slots, long capacity, int32_t* block_cnt,
                                     int32_t* block_off, float* dense, int32_t* slot_of, int32_t* total, hipStream_t st);
hipError_t ddn_dev_chain_pack2(const uint8_t* rec, const uint8_t* fl, size_t n, uint8_t* out2, hipStream_t st);
hipError_t ddn_dev_chain_carry(const uint8_t* rec_prev, const uint8_t* fl_prev, const int32_t* cnt_prev, int have_prev,
                               uint8_t* rec_cur, uint8_t* fl_cur, size_t stride_sym, int T, int n_channels, hipStream_t st);
hipError_t ddn_dev_chain_counts(const int32_t* cnt_new, int T, int n_channels, int flush, int32_t* cnt_scan, int32_t* cnt_full,
                                hipStream_t st);
/* ---- ddn_cqrx.hip: the symbol-rate receive loop behind the CQPSK demodulator ---- */
typedef struct DdnCqConfig {
    int protocol;      /* 0 P25 Phase 1, 1 P25 Phase 2 */
    int sync_len;      /* 24 / 20 dibits */
    int t_max;         /* level ring: 24 / 19 */
    int lock_symbols;  /* < 0: the P25p1 handlers decide; Phase 2: 700 */
    int snr_scale;     /* the CQPSK SNR weight as the numerator over 256 (204 + (w256 >> 2)), -1 = no weight (SNR not available) */
    int nid_threshold; /* p25p1_get_erasure_threshold() */
    int max_events;
    uint64_t target[2][4]; /* [polarity][identity, X2400, N1200, P1200]: the sync word as raw dibits, oldest in the high bits */
} DdnCqConfig;
typedef struct DdnCqState { /* one channel, carried from call to call */
    float minbuf[1024], maxbuf[1024]; /* the extrema average's window (dsd_state: minbuf / maxbuf, msize 1024) */
    float sbuf[128];                  /* the slicer window (ssize 128) */
    float lbuf[24], shist[24];        /* level ring of the frame search, symbol history */
    int32_t d[100];                   /* a trellis block being read: de-interleaved LLR pairs */
    uint8_t nb[64], nr[64];           /* a NID being read: bits / reliabilities */
    double min_sum, max_sum;
    uint64_t hist;
    float max, min, lmin, lmax;
    int32_t sidx, midx, sums_valid;
    int32_t have_sync, lock_left, lastsync, map_idx, lidx, level_count, hist_count, shead, scount, hunt_pos;
    int32_t h_phase, h_idx, h_left, h_block, h_end, h_skip, h_k, h_nac, h_p2cc;
    int32_t pad_;
} DdnCqState;
hipError_t ddn_dev_cq_rx_init(DdnCqState* states, int n_channels, hipStream_t st);
hipError_t ddn_dev_cq_rx(const float* symbols, const int32_t* counts_in, size_t sym_stride, int n_fixed, int n_channels, const DdnCqConfig* cfg,
                         DdnCqState* states, uint8_t* rec, uint8_t* flags, int32_t* counts_out, size_t max_sym, int32_t* events,
                         int32_t* n_events, int32_t* event_data, hipStream_t st);
hipError_t ddn_dev_zero_words(int32_t* p, int n, hipStream_t st);
hipError_t ddn_dev_fill_words(int32_t* p, int n, int32_t value, hipStream_t st);
hipError_t ddn_dev_chain_events(const int32_t* list_prev, const int32_t* data_prev, const int32_t* n_prev, const int32_t* new_prev,
                                int have_prev, const int32_t* ev_new, const int32_t* evd_new, const int32_t* n_new, int E, int EL, int T,
                                int n_channels, int32_t* list_cur, int32_t* data_cur, int32_t* n_cur, hipStream_t st);
hipError_t ddn_dev_chain_pdu_index(const int32_t* list, const int32_t* data, const int32_t* n_list, int EL, const int32_t* sync_pos,
                                   const int32_t* n_syncs, const int32_t* nid4, int n_channels, int F, int off0, int PF,
                                   int32_t* pdu_slot, uint8_t* pdu_hdr, int32_t* pdu_info, int32_t* n_pdu, hipStream_t st);
hipError_t ddn_dev_chain_pdu_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                                    const int32_t* pdu_slot, const int32_t* pdu_info, int n_channels, int F, int PF, int PB,
                                    int16_t* llr, uint8_t* valid, hipStream_t st);
hipError_t ddn_dev_chain_pdu_finish(const int32_t* pdu_slot, const uint8_t* blocks12, const uint8_t* valid, const uint8_t* blocks18,
                                    const uint8_t* hcand16, const int32_t* hcount, const uint8_t* hwanted, int n_entries, int PB,
                                    uint8_t* pdu_hdr, int32_t* pdu_info, hipStream_t st);
hipError_t ddn_dev_chain_pdu_combine(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                                     const int32_t* pdu_slot, const int32_t* pdu_info, const int16_t* llr, const uint8_t* valid,
                                     const uint8_t* blocks12, int n_entries, int PF, int PB, int16_t* hllr, uint8_t* wanted, hipStream_t st);
hipError_t ddn_dev_chain_pdu_take_first(const uint8_t* cand16, const int32_t* counts, int n_blocks, uint8_t* blocks12, int32_t* metric,
                                        hipStream_t st);
hipError_t ddn_dev_chain_pdu_r34_wanted(const int32_t* pdu_slot, const uint8_t* pdu_hdr, const int32_t* pdu_info, const uint8_t* valid,
                                        int n_blocks, int PB, uint8_t* wanted, hipStream_t st);
hipError_t ddn_dev_chain_pdu_r34_select(const uint8_t* cand24, const int32_t* counts, const uint8_t* wanted, int n_blocks,
                                        uint8_t* blocks18, uint8_t* crc9_ok, hipStream_t st);
hipError_t ddn_dev_chain_frames(const int32_t* list, const int32_t* data, const int32_t* n_list, int EL, const int32_t* sync_pos,
                                const int32_t* n_syncs, int n_channels, int F, int off0, int off1, int off2, int32_t* nid4,
                                uint8_t* tsbk, uint8_t* tsbk_crc, uint8_t* cls, int32_t* lists, int32_t* list_n, hipStream_t st);
enum { DDN_CLS_LDU1 = 1, DDN_CLS_LDU2 = 2, DDN_CLS_HDU = 4, DDN_CLS_TDULC = 8 }; /* frame-type bits of a slot's class byte */
enum { DDN_LIST_LDU1 = 0, DDN_LIST_LDU2 = 1, DDN_LIST_HDU = 2, DDN_LIST_TDULC = 3, DDN_LIST_LSD = 4, DDN_LIST_COUNT = 5 }; /* work lists [k][S] */
hipError_t ddn_dev_nxdn_voice_select(const int32_t* sync_pos, const int32_t* n_sync, const uint8_t* lich, const uint8_t* valid,
                                     int n_channels, int my, int vf, int32_t* v_pos, int32_t* v_n, uint8_t* skip4, hipStream_t st);
hipError_t ddn_dev_fsk4_chain_syncs(const int32_t* c_pos, const uint8_t* c_pat, const uint8_t* c_pre, const uint8_t* c_prel,
                                    const int32_t* c_n, int myc, const int32_t* s_pos, const uint8_t* s_pat, const uint8_t* s_pre,
                                    const uint8_t* s_prel, const int32_t* s_n, int my, const int32_t* n_new, int T, int flush,
                                    int32_t* d_pos, uint8_t* d_pat, uint8_t* d_pre, uint8_t* d_prel, int32_t* d_n, int myd,
                                    int32_t* o_pos, uint8_t* o_pat, uint8_t* o_pre, uint8_t* o_prel, int32_t* o_n, int32_t* dropped,
                                    int n_channels, hipStream_t st);
hipError_t ddn_dev_fsk4_chain_syncs_thr(const int32_t* c_pos, const uint8_t* c_pat, const uint8_t* c_pre, const uint8_t* c_prel,
                                        const int32_t* c_n, int myc, const int32_t* s_pos, const uint8_t* s_pat, const uint8_t* s_pre,
                                        const uint8_t* s_prel, const int32_t* s_n, int my, const int32_t* n_new, int T, int flush,
                                        int32_t* d_pos, uint8_t* d_pat, uint8_t* d_pre, uint8_t* d_prel, int32_t* d_n, int myd,
                                        int32_t* o_pos, uint8_t* o_pat, uint8_t* o_pre, uint8_t* o_prel, int32_t* o_n, int32_t* dropped,
                                        int n_channels, const float* c_thr, const float* s_thr, float* d_thr, float* o_thr, hipStream_t st);
hipError_t ddn_dev_u8_shr1(const uint8_t* in, size_t n, uint8_t* out, hipStream_t st);
hipError_t ddn_dev_tsbk_select(const uint8_t* cand, const int32_t* counts, size_t n, uint8_t* out12, uint8_t* crc_ok, uint8_t* sel,
                               hipStream_t st);
#ifdef __cplusplus
}
#endif
#if defined(__HIPCC__)
/* f(item) for every item of the launch: all n_items, or the selected slots' items, grid-strided */
template <typename F>
__device__ __forceinline__ void
ddn_sel_for_each(const DdnSel& sel, long n_items, F&& f) {
    const long stride = (long)gridDim.x * (long)blockDim.x;
    long t = (long)blockIdx.x * (long)blockDim.x + (long)threadIdx.x;
    if (!sel.list) {
        for (; t < n_items; t += stride) {
            f(t);
        }
        return;
    }
    const long total = (long)(*sel.count) * (long)sel.per_slot;
    for (; t < total; t += stride) {
        const long e = t / sel.per_slot;
        f((long)sel.list[e] * (long)sel.per_slot + (t - e * (long)sel.per_slot));
    }
}
#endif

#endif
