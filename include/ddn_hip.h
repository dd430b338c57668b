/*
 * include/ddn_hip.h — C-ABI of libdsdneo_hip.so: the MI355X (gfx950) implementation of dsd-neo's
 * sample-streaming hot path.  Plain C, plain pointers and sizes; no torch/HIP types.
 *
 * Two families of entry points:
 *
 *  (1) Batched `ddn_*` calls — B independent channels (one I/Q capture each) per call.  These are what a
 *      dsd-neo build would bind behind its demod thread / stream-read hook when it hosts many channels:
 *        ddn_front_end_run       replaces the per-stream loop "widen -> full_demod()" of the demod thread
 *                                (reference src/io/radio/rtl_device.cpp:1777 + src/io/radio/rtl_sdr_fm.cpp:3458-3516,
 *                                 full_demod: include/dsd-neo/dsp/demod_pipeline.h:106)
 *        ddn_hooks_read / ddn_stream_set_*   serve include/dsd-neo/runtime/rtl_stream_io_hooks.h:25-28 `read`
 *      Pointers named d_* are DEVICE pointers (hipMalloc / torch .data_ptr()), h_* are host pointers.
 *
 *  (2) Drop-in single-stream symbols with the reference's own names and signatures (host pointers), so the
 *      reference's unit tests for this path can link against this library unchanged:
 *        simd_fir_complex_apply, simd_hb_decim2_complex, simd_hb_decim2_real, simd_fir_get_impl_name
 *                                (include/dsd-neo/dsp/simd_fir.h:41-77)
 *        widen_u8_to_f32_bias127 (include/dsd-neo/dsp/simd_widen.h:51)
 *        ddn_fsk_modem_discriminator_process (== dsd_fsk_modem_discriminator_process,
 *                                include/dsd-neo/dsp/fsk_modem.h:42; state struct layout identical)
 *      Per-call granularity is far too fine for a GPU; these exist for parity, not throughput.
 *
 * All functions return 0 (or a non-negative count) on success and a negative DDN_E* code on failure; they
 * never fall back to a CPU path: if no gfx950 device / kernel image is available they fail with DDN_ENODEV.
 *
 * Threading: batch objects (ddn_batch, ddn_p25_rx, ddn_cqpsk_batch, ddn_ted_batch, ddn_slicer_batch, ddn_resampler,
 * ddn_p25p1_framer) carry per-channel state and are not internally synchronised - one caller at a time per object,
 * any number of objects in parallel.  The stateless ddn_fec_* / ddn_p25p1_* batch entry points take their scratch
 * from the stream (hipMallocAsync) and may be called concurrently from several host threads / streams.
 * ddn_last_error() is per thread.
 */
#ifndef DDN_HIP_H
#define DDN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDN_OK       0
#define DDN_EINVAL  -1 /* bad argument (mirrors the reference's silent no-op on bad args, but reported) */
#define DDN_ENODEV  -2 /* no HIP device / kernel image for this device */
#define DDN_ENOMEM  -3
#define DDN_EHIP    -4 /* HIP runtime error; ddn_last_error() has the string */
#define DDN_ERANGE  -5 /* unsupported size (e.g. taps_len > 143, block_len too small) */

const char* ddn_last_error(void);
const char* ddn_version(void);

/* Channel LPF profile ids == DSD_CH_LPF_PROFILE_* (include/dsd-neo/dsp/demod_state.h:36-43) */
enum {
    DDN_LPF_WIDE = 0,
    DDN_LPF_6K25 = 1,
    DDN_LPF_12K5 = 2,
    DDN_LPF_PROVOICE = 3,
    DDN_LPF_P25_C4FM = 4,
    DDN_LPF_P25_CQPSK = 5,
};

enum { DDN_IN_CU8 = 0, DDN_IN_CF32 = 1 };

/* Per-batch front-end configuration: the subset of struct demod_state (include/dsd-neo/dsp/demod_state.h:67-262)
 * that the FSK-discriminator path reads, fixed for all channels of the batch. */
typedef struct ddn_front_end_config {
    int n_channels;      /* B */
    int sample_rate_hz;  /* demod rate (rate_out) */
    int symbol_rate_hz;  /* informational (fsk modem cfg) */
    int levels;          /* 2 or 4 */
    int lpf_profile;     /* DDN_LPF_* */
    int input_format;    /* DDN_IN_CU8 / DDN_IN_CF32 */
    int block_len;       /* complex samples per reference full_demod() block (edge-replication granule) */
    float squelch_level; /* channel_squelch_level; 0 = disabled (reference default) */
} ddn_front_end_config;

typedef struct ddn_batch ddn_batch;

int ddn_batch_create(const ddn_front_end_config* cfg, ddn_batch** out);
void ddn_batch_destroy(ddn_batch* b);
/* forget all carried per-channel state (FIR history, modem dc/peak/prev): a fresh stream */
int ddn_batch_reset(ddn_batch* b, void* hip_stream);
/* taps actually in use (host-designed, same rule as channel_lpf_ensure_plan); returns taps_len */
int ddn_batch_get_taps(const ddn_batch* b, float* taps_out, int cap);

/* Half-band decimate-by-2 cascade in front of the channel LPF == demod_state.downsample_passes
 * (full_demod_apply_halfband_decimation, src/dsp/demod_pipeline.cpp:983-1001: 31-tap first stage, 15-tap afterwards,
 * src/dsp/halfband.cpp:35-74; each stage is simd_hb_decim2_complex on the block it inherits).  With passes = p the
 * input runs at sample_rate_hz << p, cfg.block_len (input samples) must be a multiple of 2^p with
 * block_len >> p >= taps_len, n must be a multiple of 2^p and d_disc holds [B][n >> p].  Resets the batch. */
int ddn_batch_set_decimation(ddn_batch* b, int passes);

/* Optional IQ conditioning between the channel LPF and the discriminator (all off by default like
 * the reference, src/io/radio/rtl_demod_config.cpp:479-489):
 *   dc_block_enable / dc_shift   == demod_state.iq_dc_block_enable / iq_dc_shift: iq_dc_block(), a per-sample leaky
 *                                integrator with alpha = 2^-k, k clamped to 6..15 (src/dsp/demod_pipeline.cpp:948-978)
 *   iqbal_enable / thr / ema     == iqbal_enable / iqbal_thr (0 -> 0.02) / iqbal_alpha_ema_a (0 -> 0.2):
 *                                full_demod_apply_iq_balance(), per-block image estimate in binary64, EMA across blocks,
 *                                correction applied once |alpha| >= thr (:1131-1171)
 * With either switch on the batch runs the unfused route (channel LPF result through HBM, then one lane per channel);
 * blocks closed by the squelch gate skip both stages like the reference.  Resets the batch. */
int ddn_batch_set_iq_conditioning(ddn_batch* b, int dc_block_enable, int dc_shift, int iqbal_enable, float iqbal_thr,
                                  float iqbal_ema_alpha);

/* One pass of widen -> channel LPF -> (squelch) -> FSK discriminator over n complex samples per channel.
 *   d_iq   : [B][n] interleaved I/Q, u8 pairs (CU8) or float pairs (CF32), channel-major
 *   d_disc : [B][n] float discriminator samples (AGC'd to +-30000, clipped to int16 range)
 * The call is equivalent to ceil(n / block_len) consecutive full_demod() calls per channel; carried state
 * persists in `b` across calls.  Asynchronous on `hip_stream` (NULL = default stream). */
int ddn_front_end_run(ddn_batch* b, const void* d_iq, size_t n, float* d_disc, void* hip_stream);
/* same, host buffers (H2D, run, D2H, synchronous) */
int ddn_front_end_run_host(ddn_batch* b, const void* h_iq, size_t n, float* h_disc);
/* channels per workgroup of the fused kernel: 0 = by batch size (8 up to 2048 channels - twice the workgroups for a launch that has
 * the device to itself - else 16), 8, 16.  A workgroup's time does not depend on how many of its slots are filled, so a host that runs
 * several front ends side by side asks for 16.  Results do not depend on it. */
int ddn_batch_set_channels_per_workgroup(ddn_batch* b, int channels);
/* Segments (round 6): the batch's channel index as up to three runs with a channel low-pass profile each (DDN_LPF_*; segment 0 keeps
 * the batch's own; all must design the same tap count, DDN_ERANGE otherwise) - the protocol groups of a mixed batch, each a separate
 * dsd-neo demodulator configuration (src/dsp/demod_pipeline.cpp:443-524), behind ONE launch of ceil(n_channels / 16) workgroups
 * whatever the groups' sizes (a workgroup may hold channels of two groups).  seg_channels add up to n_channels.  Not with the
 * half-band cascade, IQ conditioning or squelch. */
int ddn_batch_set_segments(ddn_batch* b, int n_seg, const int32_t* seg_channels, const int32_t* lpf_profiles);
/* ddn_front_end_run over the segments: d_iq[k] / d_disc[k] = segment k's [seg_channels[k]][n] arrays; n >= 72 */
int ddn_front_end_run_segments(ddn_batch* b, const void* const* d_iq, size_t n, float* const* d_disc, void* hip_stream);

/* per-channel modem state after the last run: {prev_i, prev_q, have_prev, dc_est, peak_est} (synchronous) */
int ddn_batch_get_fsk_state(ddn_batch* b, int channel, float out5[5]);

/* duration in ms of the dominant kernel(s) of the most recent ddn_front_end_run, measured with HIP events
 * on the stream the kernels were launched on; valid after the stream has been synchronised.
 * out3 = {fir_ms, serial_ms, total_ms}.  Only recorded when ddn_batch_set_timing(b,1) was called. */
int ddn_batch_set_timing(ddn_batch* b, int enable);
int ddn_batch_get_timing(ddn_batch* b, float out3[3]);

/* ---- P25 Phase 1 C4FM slicer + soft decisions, and the per-sample P25 matched filter, batched --------------
 * ddn_p25_slicer_run == one getDibitSoft() per symbol per channel (include/dsd-neo/core/dibit.h:43-52) for a stream
 * whose last sync type is P25p1 (thresholds track continuously): symbols [B][n] f32 -> records [B][n][10], the
 * reference's symbol-capture record {u8 dibit, u8 reliability, i16 llr0, i16 llr1, f32 symbol} little-endian
 * (src/core/frames/dsd_dibit.c:794-818).  Slicer state starts from the reference's reset values and is carried.
 * ddn_p25_matched_filter_run == p25_filter(sample, 10) over every sample of every channel (include/dsd-neo/dsp/
 * sps_filters.h), 90-sample history carried in the batch; d_in != d_out. */
typedef struct ddn_slicer_batch ddn_slicer_batch;
int ddn_slicer_batch_create(int n_channels, int negative_polarity, ddn_slicer_batch** out);
void ddn_slicer_batch_destroy(ddn_slicer_batch* b);
int ddn_slicer_batch_reset(ddn_slicer_batch* b);
int ddn_p25_slicer_run(ddn_slicer_batch* b, const float* d_symbols, size_t n, uint8_t* d_records10, void* hip_stream);
int ddn_p25_slicer_run_host(ddn_slicer_batch* b, const float* symbols, size_t n, uint8_t* records10);
int ddn_slicer_batch_get_thresholds(ddn_slicer_batch* b, int channel, float out5[5]);
int ddn_p25_matched_filter_run(ddn_slicer_batch* b, const float* d_in, size_t n, float* d_out, void* hip_stream);
int ddn_p25_matched_filter_run_host(ddn_slicer_batch* b, const float* in, size_t n, float* out);

/* ---- fixed-protocol P25 Phase 1 C4FM receive loop, batched ------------------------------------------------
 * Discriminator samples -> symbol capture records, i.e. what one dsd-neo decoder thread does per stream between
 * rtl_stream_read() and the protocol handler:
 *   getSymbol(opts, state, have_sync)   include/dsd-neo/dsp/symbol.h; src/dsp/dsd_symbol.c:1854-1880 (jitter timing,
 *                                       C4FM 5-sample window mean, matched filter once P25p1 has been seen, in-sync clip)
 *   getFrameSync(opts, state)           include/dsd-neo/dsp/frame_sync.h; src/dsp/dsd_frame_sync.c:3098-3148 restricted
 *                                       to the P25p1 pattern with the modulation locked to C4FM (-mc), incl. the
 *                                       hunting level window and dsd_sync_warm_start_thresholds_outer_only()
 *   getDibitSoft(opts, state, &soft)    include/dsd-neo/core/dibit.h:43-52 for every in-frame symbol
 * How long a frame is read in frame: with ddn_p25_rx_set_handlers(b, 1, ..) the reference's own handlers decide, inside the
 * loop (processFrame -> dsd_dispatch_handle_p25p1, src/engine/dispatch/dispatch_p25p1.c:86-143,206-225,391-403): the NID's 33
 * symbols, p25p1_nid_decode (hard decode, NAC retry, Chase search), then per DUID - HDU 339, LDU1 / LDU2 807, TDU 15, TDULC
 * 159, TSDU 101 per block until the decoded block's last-block flag (p25p1_tsbk.c:1051-1072), PDU by its header block
 * (p25p1_mdpu.c:270-307), nothing for an undefined DUID or a failed NID.  Without it the loop stays in frame for a
 * caller-given count (cfg.lock_symbols / ddn_p25_rx_set_lock_symbols) - a deviation kept for experiments.
 * Parity: the slicer, warm start and level window are pinned to the compiled reference; the sample / hunting loops are
 * restated from source (dsd_symbol.c / dsd_frame_sync.c do not build here) and anchored on the reference's symbolizer KATs
 * and its full-chain known answers.
 *   d_disc      : [B][n] f32 discriminator samples (ddn_front_end_run output), channel-major
 *   d_records10 : [B][max_symbols][10] capture records {u8 dibit, u8 reliability, i16 llr0, i16 llr1, f32 symbol}
 *                 (write_symbol_capture_record layout, src/core/frames/dsd_dibit.c:794-818); while hunting the dibit
 *                 is the sign decision ('1'/'3') and the soft fields are 0
 *   d_flags     : [B][max_symbols] bit0 = symbol read in frame (have_sync = 1), bit1 = frame sync accepted on this
 *                 symbol, bit2 = negative polarity
 *   d_counts    : [B] symbols produced by this call (<= ddn_p25_rx_max_symbols(n)) */
typedef struct ddn_p25_rx_config {
    int n_channels;
    int out_rate_hz;        /* discriminator sample rate (48000) */
    int sym_rate_hz;        /* 4800 */
    int lock_symbols;       /* symbols read with have_sync = 1 after each accepted sync */
    int use_matched_filter; /* opts->use_cosine_filter */
} ddn_p25_rx_config;
typedef struct ddn_p25_rx ddn_p25_rx;
int ddn_p25_rx_create(const ddn_p25_rx_config* cfg, ddn_p25_rx** out);
void ddn_p25_rx_destroy(ddn_p25_rx* b);
int ddn_p25_rx_reset(ddn_p25_rx* b);
/* in-frame symbol count after a sync, per channel (host array [n_channels]; NULL = cfg.lock_symbols everywhere): lets one
 * batch mix traffic classes, e.g. 840 for voice channels (LDUs) and 156 / 336 for one- / three-block TSDU control channels */
int ddn_p25_rx_set_lock_symbols(ddn_p25_rx* b, const int32_t* per_channel);
/* enable != 0: the reference's per-DUID handlers decide the in-frame length (see above); nid_erasure_threshold =
 * p25p1_get_erasure_threshold() (64 unless the reference's config overrides it; <= 0 selects 64).  Every handler decision is
 * reported: ddn_p25_rx_set_events() gives the device buffers of the next runs - d_events i32 [B][max_events][4] =
 * {output index of the deciding symbol, kind, a, b}, d_n_events i32 [B] (decisions of that call; may exceed max_events, only
 * the first max_events are stored):
 *   kind 1 NID        a = p25p1_nid_decode status (1 ok, 2 parity override, <= 0 failed), b = NAC | DUID << 16 (DUID 0xFF: none)
 *   kind 2 TSDU block a = block index, b = CRC16 good | byte 1 of the block << 8 | (last-block flag << 8 | selected list candidate) << 16
 *   kind 3 PDU header a = header CRC16 good, b = blocks to read | byte 0 << 16
 * Without buffers the decisions are still taken, just not reported. */
int ddn_p25_rx_set_handlers(ddn_p25_rx* b, int enable, int nid_erasure_threshold);
int ddn_p25_rx_set_events(ddn_p25_rx* b, int32_t* d_events, int32_t* d_n_events, size_t max_events);
/* what each decision decoded, row for row beside d_events: d_event_data i32 [B][max_events][4] (NULL = not wanted; set the event
 * list first) - kind 1: p25p1_nid_decode's {status, NAC, DUID, corrected bits}; kind 2 / 3: the block's 12 bytes as three
 * little-endian words + {CRC16 good | selected list candidate << 8 | block index << 16}.  With these the frames' NIDs and TSDU blocks
 * need no second decode after the loop (the chain object, include/ddn_chain.h, takes them from here). */
int ddn_p25_rx_set_event_data(ddn_p25_rx* b, int32_t* d_event_data);
/* kernel times of the last ddn_p25_rx_run(), HIP events on the launch stream: ms2 = {matched filter, receive-loop kernel} */
int ddn_p25_rx_set_timing(ddn_p25_rx* b, int enable);
int ddn_p25_rx_get_timing(ddn_p25_rx* b, float* ms2);
/* the same averaged over the launches since timing was switched on (at most the last 64), without synchronising between them */
int ddn_p25_rx_get_timing_avg(ddn_p25_rx* b, float* ms2, int* n_launches);
/* lanes (= channels) per recurrence wavefront: 0 = automatic (8 up to 4096 channels, 16 up to 8192, 32 up to 16384, else 64),
 * else 8 / 16 / 32 / 64.  Results do not depend on it; fewer lanes per wave = fewer trips shared with a hunting lane. */
int ddn_p25_rx_set_channels_per_wave(ddn_p25_rx* b, int channels_per_wave);
/* 1 = handler mode computes the P25 matched filter (src/dsp/dsd_filters.c:173-200,299-324) INSIDE the loop kernel, from the raw tile
 * where it is staged in LDS (the handler wave, between decisions), instead of in a kernel of its own that writes a second f32 row to
 * HBM: no filter kernel, 1.57 GB less HBM traffic per 4096 x 48000 call, 786 MB less device memory - and a slower step (round 6,
 * measured: 8.2 against 7.9 ms; the loop's helper waves have no spare issue slots for 91 taps per sample).  Default 0.
 * Same records, flags and events bit for bit either way.  No effect outside handler mode or at sixteen channels per workgroup. */
int ddn_p25_rx_set_filter_in_loop(ddn_p25_rx* b, int on);
/* A/B selectors of the loop kernel's schedule (which of two equivalent code paths a launch takes; the results never depend on them):
 * 4096 = the bulk hunting passes one owner lane at a time, 2097152 = lean runs of one trip, 16777216 = four recurrence waves of two
 * lanes ... (ddn_rx.hip documents each bit where it is read).  For the parity tests and timing tools; 0 = the product's choice. */
int ddn_p25_rx_set_debug_flags(ddn_p25_rx* b, int flags);
size_t ddn_p25_rx_max_symbols(const ddn_p25_rx* b, size_t n);
int ddn_p25_rx_run(ddn_p25_rx* b, const float* d_disc, size_t n, uint8_t* d_records10, uint8_t* d_flags,
                   int32_t* d_counts, size_t max_symbols, void* hip_stream);
int ddn_p25_rx_run_host(ddn_p25_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags,
                        int32_t* counts, size_t max_symbols);
/* the same with the handlers' event list brought back to host arrays (events [B][max_events][4], n_events [B]; both NULL = none;
 * event_data [B][max_events][4] or NULL: the decoded payload of each event, ddn_p25_rx_set_event_data) */
int ddn_p25_rx_run_host_ev(ddn_p25_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags, int32_t* counts,
                           size_t max_symbols, int32_t* events, int32_t* n_events, size_t max_events, int32_t* event_data);
int ddn_p25_rx_get_thresholds(ddn_p25_rx* b, int channel, float out7[7]);
/* timing experiments (ddn_p25_rx_set_debug_flags bit 65536): handler requests of a channel so far, cycles its lane waited for them */
int ddn_p25_rx_debug_counters(ddn_p25_rx* b, int channel, long long out2[2]);

/* ---- Gardner symbol-timing recovery (CQPSK branch), batched ----------------------------------------------
 * == op25_gardner_cc(struct demod_state*) (include/dsd-neo/dsp/costas.h; src/dsp/costas.cpp:804-858) applied to B
 * channels at once; each channel's ted_state_t is carried inside the batch object.
 *   d_iq  : [B][n] complex f32 at sample rate (post channel-LPF / AGC / FLL), channel-major
 *   d_sym : [B][sym_stride] complex f32 at symbol rate; d_sym_count[B] = symbols produced this call (may exceed
 *           sym_stride, in which case the surplus was dropped: size sym_stride >= n/sps*1.01 + 2)
 * out8 of get_state = {mu, omega, last_r, last_j, lock_accum, lock_count, dl_index, twice_sps}. */
typedef struct ddn_ted_batch ddn_ted_batch;
int ddn_ted_batch_create(int n_channels, int sps, int symbol_rate_hz, float ted_gain, ddn_ted_batch** out);
void ddn_ted_batch_destroy(ddn_ted_batch* b);
int ddn_ted_batch_reset(ddn_ted_batch* b, void* hip_stream);
/* One ddn_gardner_run() call == one op25_gardner_cc() call by default.  With block_len > 0 the call is equivalent to
 * ceil(n / block_len) consecutive op25_gardner_cc() calls of block_len samples (the last one shorter): the loop gain is
 * re-selected (src/dsp/costas.cpp:143-168, matters for symbol rates >= 5500) and a symbol pending at a block's last
 * sample is deferred to the next block, exactly as when full_demod() drives it block by block. */
int ddn_ted_batch_set_block_len(ddn_ted_batch* b, size_t block_len);
int ddn_gardner_run(ddn_ted_batch* b, const float* d_iq, size_t n, float* d_sym, size_t sym_stride, int* d_sym_count,
                    void* hip_stream);
int ddn_gardner_run_host(ddn_ted_batch* b, const float* iq, size_t n, float* sym, size_t sym_stride, int* sym_count);
int ddn_ted_batch_get_state(ddn_ted_batch* b, int channel, float out8[8]);

/* ---- dsd-neo I/Q captures: the feeder side ------------------------------------------------------------
 * Reader for the "dsd-neo-iq" capture format that --iq-capture writes and --iq-replay ingests (JSON sidecar v1 / v2 + raw
 * cu8 / cf32 / cs16 samples; docs/iq-capture-replay.md:37-76, src/io/iq/iq_replay.c): same path resolution, required
 * fields, validation and replayable-byte rule as dsd_iq_replay_read_metadata / _open / _read
 * (include/dsd-neo/io/iq_replay.h:71-100); return values are the reference's dsd_iq_error codes
 * (include/dsd-neo/io/iq_types.h:24-36).  Host-only C.  base_decimation = 2^p maps to ddn_batch_set_decimation(b, p),
 * demod_rate_hz to the batch's sample_rate_hz. */
enum {
    DDN_IQ_OK = 0, DDN_IQ_ERR_IO = -1, DDN_IQ_ERR_INVALID_META = -2, DDN_IQ_ERR_UNSUPPORTED_VER = -3,
    DDN_IQ_ERR_UNSUPPORTED_FMT = -4, DDN_IQ_ERR_ALIGNMENT = -5, DDN_IQ_ERR_RATE_CHAIN = -6, DDN_IQ_ERR_RETUNE_REJECT = -7,
    DDN_IQ_ERR_ALLOC = -8, DDN_IQ_ERR_INVALID_ARG = -10
};
enum { DDN_IQ_FORMAT_CU8 = 1, DDN_IQ_FORMAT_CF32 = 2, DDN_IQ_FORMAT_CS16 = 3 };
enum { DDN_IQ_EVENT_RETUNE = 1, DDN_IQ_EVENT_MUTE = 2, DDN_IQ_EVENT_RESET = 3 };
typedef struct ddn_iq_event { /* == dsd_iq_event (iq_types.h:50-58) */
    int kind;
    uint64_t byte_offset, duration_bytes, center_frequency_hz, capture_center_frequency_hz;
    uint32_t sample_rate_hz;
    char reason[64];
} ddn_iq_event;
typedef struct ddn_iq_capture_info { /* the fields of dsd_iq_replay_config a feeder needs (iq_replay.h:26-66) */
    uint32_t metadata_version;
    int sample_format;
    uint32_t sample_rate_hz, base_decimation, post_downsample, demod_rate_hz, capture_retune_count, event_count;
    uint64_t center_frequency_hz, capture_center_frequency_hz, data_bytes, capture_drops, capture_drop_blocks,
        input_ring_drops;
    uint64_t actual_file_bytes, effective_bytes; /* on-disk size; replayable bytes (whole samples) */
    int ppm, tuner_gain_tenth_db, rtl_dsp_bw_khz;
    int offset_tuning_enabled, fs4_shift_enabled, combine_rotate_enabled, muted_bytes_excluded, contains_retunes,
        size_limit_reached, size_mismatch;
    char capture_stage[64];
    char data_path[2048], metadata_path[2048];
} ddn_iq_capture_info;
typedef struct ddn_iq_capture ddn_iq_capture;
int ddn_iq_capture_read_info(const char* path, ddn_iq_capture_info* out_info); /* == dsd_iq_replay_read_metadata */
int ddn_iq_capture_open(const char* path, ddn_iq_capture** out);               /* == dsd_iq_replay_open */
void ddn_iq_capture_close(ddn_iq_capture* c);
const ddn_iq_capture_info* ddn_iq_capture_get_info(const ddn_iq_capture* c);
const ddn_iq_event* ddn_iq_capture_get_events(const ddn_iq_capture* c, uint32_t* out_count);
int ddn_iq_capture_read(ddn_iq_capture* c, void* out, size_t max_bytes, size_t* out_bytes); /* 0 bytes = end */
int ddn_iq_capture_rewind(ddn_iq_capture* c);
int ddn_iq_effective_bytes(uint64_t data_bytes, uint64_t actual_file_size, int sample_format, uint64_t* out_effective,
                           int* out_size_mismatch); /* == dsd_iq_replay_compute_effective_bytes */
/* B captures of one format / rate chain -> one malloc'ed channel-major buffer [B][n] (n = shortest capture), the input
 * shape of ddn_front_end_run_host / ddn_cqpsk_run_host; release with ddn_iq_free */
int ddn_iq_load_batch(const char* const* paths, int n_captures, void** out_buf, size_t* out_n_samples,
                      ddn_iq_capture_info* out_info0);
void ddn_iq_free(void* p);

/* ---- downstream of the vocoder / of the receive loop: what dsd-neo does with the path's results -------------
 * ddn_audio_agf_*  == agf() (include/dsd-neo/core/audio.h:87, src/core/audio/gain.c:119-139): the float-path auto gain applied
 *                  to every synthesized 160-sample frame (playSynthesizedVoiceFS / FM, src/core/audio/dsd_audio2.c:1100,1111).
 *                  pcm f32 [n_streams][n_frames][160] in place (the vocoder's output, int16-scale floats), aout_gain f32
 *                  [n_streams] = state->aout_gain carried across calls (dsd-neo starts it at 25); audio_gain = opts->audio_gain
 *                  (0 = automatic), algid_0x21 = the reference's x1.75 for that ALGID.
 * ddn_symbol_capture_write  dsd-neo's -c symbol-capture file from a receive loop's records + flags (header + records,
 *                  src/core/file/dsd_file.c:876-890, src/core/frames/dsd_dibit.c:794-818); append != 0 continues a file.
 * ddn_wav_write_s16  PCM16 RIFF / WAVE file from float PCM (full_scale = the float value that maps to 32767). */
int ddn_audio_agf_batch(float* d_pcm, int n_streams, int n_frames, float audio_gain, int algid_0x21, float* d_aout_gain,
                        void* hip_stream);
int ddn_audio_agf_host(float* pcm, int n_streams, int n_frames, float audio_gain, int algid_0x21, float* aout_gain);
int ddn_agf_frame(float samp[160], float audio_gain, int algid_0x21, float* aout_gain_io);

/* The short-integer voice path, one talk path per row of d_pcm [n_streams][n_frames][160] (synthesized float frames at int16
 * scale) -> d_out [n_streams][n_frames][160] int16, three stages in the reference's order:
 *   1  processAudio()  src/core/audio/dsd_audio.c:427-571 - automatic level when audio_gain == 0 (block peak, 25-block peak
 *      history, gain = 30000 / peak falling at once and rising at most 5 % per frame up to 50, ramped across the frame), the
 *      caller's aout_gain applied unchanged when audio_gain > 0, no multiply when audio_gain < 0; clamp and truncate to int16
 *   2  hpf_dL()        src/core/util/dsd_misc.c:345-371,516-522 (use_hpf_d; opts->use_hpf_d), 960 Hz one-pole at 8 kHz
 *   3  agsm()          src/core/audio/gain.c:143-184 (use_agsm), per 160-sample frame; d_gain_a[stream] = state->aout_gainA
 * d_state32: DDN_S16_STATE_FLOATS floats per talk path carried between calls ([0] aout_gain, [1] history index, [2] [3] the
 * filter's v_in[0] / v_out[0], [4..28] the peak history); ddn_audio_s16_state_init() writes the reference's power-on values
 * (aout_gain 25, src/core/util/dsd_init.c:580-585) into a HOST array.  Six-fold sample repetition for 48 kHz sinks
 * (upsample(), src/core/audio/dsd_upsample.c:19-47) is a copy the sink does; it is not a kernel here. */
#define DDN_S16_STATE_FLOATS 32
int ddn_audio_s16_state_init(float* state32, int n_streams);
int ddn_audio_s16_batch(const float* d_pcm, int n_streams, int n_frames, float audio_gain, int use_hpf_d, int use_agsm,
                        int16_t* d_out, float* d_state32, float* d_gain_a, void* hip_stream);
int ddn_audio_s16_host(const float* pcm, int n_streams, int n_frames, float audio_gain, int use_hpf_d, int use_agsm, int16_t* out,
                       float* state32, float* gain_a);
int ddn_symbol_capture_write(const char* path, const uint8_t* records10, const uint8_t* flags, size_t count, int append);
int ddn_wav_write_s16(const char* path, int sample_rate_hz, int channels, const float* pcm, size_t frames, float full_scale);

/* ---- the demod thread's mode matrix -----------------------------------------------------------------
 * == rtl_demod_init_for_mode() + demod_apply_channel_lpf_defaults() (src/io/radio/rtl_demod_config.cpp:63-258,491-553): from
 * the enabled protocol flags (dsd_opts frame_* fields, 1 = enabled), the CQPSK modulation choice and the demod rate to what a
 * batch must be created with.  Host logic, no device needed. */
typedef struct ddn_mode_flags {
    int p25p1, p25p2, provoice, dmr, nxdn48, nxdn96, x2tdma, ysf, dstar, dpmr, m17;
    int mod_qpsk;    /* opts->mod_qpsk: CQPSK / LSM reception requested */
    int analog_only; /* opts->analog_only or the M17 encoder: audio monitor, no symbol output */
} ddn_mode_flags;
enum { DDN_OUTPUT_AUDIO_MONITOR = 0, DDN_OUTPUT_FSK_DISCRIMINATOR = 1, DDN_OUTPUT_SYMBOL_CQPSK = 2 };
typedef struct ddn_mode_result {
    int output_kind;        /* DDN_OUTPUT_*: which of ddn_front_end_run / ddn_cqpsk_run serves the batch */
    int symbol_rate_hz, symbol_levels;
    int lpf_profile;        /* DDN_LPF_* */
    int channel_lpf_enable; /* on from 20 kHz demod rate */
    int cqpsk_enable, ted_enabled;
    int samples_per_symbol; /* demod rate / symbol rate, rounded (the Gardner loop's sps on the CQPSK path) */
} ddn_mode_result;
int ddn_mode_config(const ddn_mode_flags* flags, int demod_rate_hz, ddn_mode_result* out);

/* ---- the consumer-side seam: serving dsd-neo's stream-read hook from batched results ---------------
 * A dsd-neo decoder thread pulls samples through dsd_rtl_stream_io_hooks.read(rtl_ctx, out, count, &got)
 * (include/dsd-neo/runtime/rtl_stream_io_hooks.h:25-32; callers src/dsp/dsd_symbol.c:889-920,1412-1435: count is 512 or
 * 1, blocks until >= 1 sample, < 0 on end of stream).  A ddn_stream_set holds one single-producer / single-consumer
 * float queue per channel: the producer pushes the rows of each batch interval (discriminator samples from
 * ddn_front_end_run_host, or CQPSK symbols with their per-channel counts), every decoder instance gets
 * state->rtl_ctx = ddn_stream_set_ctx(set, channel) and the host installs
 *     dsd_rtl_stream_io_hooks h = { ddn_hooks_read, ddn_hooks_return_pwr };  dsd_rtl_stream_io_hooks_set(h);
 * The metrics getters the symbol loop polls (rtl_stream_metrics_hooks.h:28-48: output_rate_hz, output_kind 0 = FSK
 * discriminator / 1 = CQPSK symbols, symbol_profile, stream_generation) are argument-less in the reference (one stream
 * per process); here they take the channel context.  Host-only, thread-safe, no device work. */
typedef struct ddn_stream_set ddn_stream_set;
int ddn_stream_set_create(int n_channels, size_t capacity_samples, unsigned output_rate_hz, int output_kind,
                          int symbol_rate_hz, int levels, int channel_profile, ddn_stream_set** out);
void ddn_stream_set_destroy(ddn_stream_set* s); /* close first and let every reader return from read() before this */
void* ddn_stream_set_ctx(ddn_stream_set* s, int channel);
/* rows [n_channels][row_stride], n samples each (or counts[c] <= n when counts != NULL); blocks while a queue is full */
int ddn_stream_set_push(ddn_stream_set* s, const float* rows, size_t n, size_t row_stride, const int32_t* counts);
int ddn_stream_set_set_power(ddn_stream_set* s, int channel, double mean_power); /* value return_pwr reports */
int ddn_stream_set_bump_generation(ddn_stream_set* s); /* retune / restart: drop queued samples, generation + 1 */
void ddn_stream_set_close(ddn_stream_set* s);          /* readers drain what is queued, then read() returns -1 */
int ddn_hooks_read(void* rtl_ctx, float* out, size_t count, int* out_got);
double ddn_hooks_return_pwr(const void* rtl_ctx);
unsigned int ddn_hooks_output_rate_hz(const void* rtl_ctx);
int ddn_hooks_output_kind(const void* rtl_ctx);
int ddn_hooks_symbol_profile(const void* rtl_ctx, int* out_symbol_rate_hz, int* out_levels, int* out_channel_profile);
uint32_t ddn_hooks_stream_generation(const void* rtl_ctx);

/* ---- P25 Phase 1 framer: receive-loop records -> FEC kernel inputs, on the device -------------------------------------
 * What the reference's P25p1 handlers do with getDibitSoft() between the sync and the decoders, as gathers: every field of
 * a frame sits at a fixed dibit offset from the frame sync (24 sync dibits; NID = 32 dibits + the status symbol at frame
 * index 35, src/protocol/p25/phase1/dispatch_p25p1.c:123-143; one status symbol at every frame index = 35 mod 36,
 * p25p1_ldu.c:27-39).  ddn_p25p1_layout_* are the host-side offset tables (0 = first sync dibit; pure functions, no
 * device): NID 32 dibits; trellis block b = 0..2 of a TSDU / PDU (98 dibits each); LDU1 / LDU2 Hamming words as
 * [24][5] dibits in the order hex_data[0..11], hex_parity[0..11] (LDU1) or hex_data[0..15], hex_parity[0..7] (LDU2)
 * (p25p1_ldu1.c, p25p1_ldu2.c:211-236); the nine voice frames' first dibit + status counter (process_IMBE input).
 * Frame slots: slot = channel * max_frames_per_channel + k for the k-th sync of the channel (ascending); slots beyond
 * the channel's sync count, and fields that run past the channel's records, are zero-filled with valid = 0.
 *   gather_nid            -> bits63 / reliab63 [slots][63], parity / parity_reliab [slots]: ddn_p25p1_nid_decode_batch input
 *   gather_trellis_block  -> llr [slots][196] int16: ddn_fec_p25_12_soft_batch input (and/or hard bits [slots][196])
 *   gather_ldu_words      -> bits [slots][24][10] (+ per-bit reliability): ddn_fec_hamming_10_6_3_* input
 *   imbe_index            -> first record / status counter [slots][9]: ddn_p25p1_imbe_deinterleave_batch input */
int ddn_p25p1_layout_nid(int32_t out32[32]);
int ddn_p25p1_layout_trellis_block(int block, int32_t out98[98]);
int ddn_p25p1_layout_ldu_words(int ldu, int32_t out120[120]);
int ddn_p25p1_layout_ldu_imbe(int32_t first9[9], int32_t status9[9]);
typedef struct ddn_p25p1_framer ddn_p25p1_framer;
int ddn_p25p1_framer_create(int n_channels, int max_frames_per_channel, ddn_p25p1_framer** out);
void ddn_p25p1_framer_destroy(ddn_p25p1_framer* f);
/* scan the receive loop's flags (bit 1 = sync accepted at this symbol) of one ddn_p25_rx_run() call */
int ddn_p25p1_framer_index(ddn_p25p1_framer* f, const uint8_t* d_flags, const int32_t* d_counts, size_t max_symbols,
                           void* hip_stream);
int ddn_p25p1_framer_get_syncs(ddn_p25p1_framer* f, int32_t* n_syncs, int32_t* sync_pos); /* host copies, synchronous */
/* the same arrays as device pointers: n_syncs [B], sync_pos [B][max_frames] (valid until the next _index on this object) */
int ddn_p25p1_framer_device_syncs(ddn_p25p1_framer* f, const int32_t** d_n_syncs, const int32_t** d_sync_pos);
/* d_dropped [B]: running count of accepted syncs that found no frame slot in their call (n_syncs is clamped to
 * max_frames_per_channel; what the clamp cut off is counted here instead of vanishing).  0 unless the slots are too few. */
int ddn_p25p1_framer_device_dropped(ddn_p25p1_framer* f, const int32_t** d_dropped);
/* tsbk_decode_repetition_bytes() after the list decoder (src/protocol/p25/phase1/p25p1_tsbk.c:108-130): of each item's
 * candidates [n][8] (ddn_fec_p25_12_soft_list_batch output) the first whose CRC16 is clean, else the first; out12 [n][12],
 * crc_ok [n], sel [n] (may be NULL) */
struct ddn_p25_12_candidate;
int ddn_fec_p25_tsbk_select_batch(const struct ddn_p25_12_candidate* d_candidates8, const int32_t* d_counts, size_t n, uint8_t* d_out12,
                                  uint8_t* d_crc_ok, uint8_t* d_sel, void* hip_stream);
int ddn_p25p1_framer_gather_nid(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts,
                                size_t max_symbols, uint8_t* d_bits63, uint8_t* d_reliab63, uint8_t* d_parity,
                                uint8_t* d_parity_reliab, uint8_t* d_valid, void* hip_stream);
int ddn_p25p1_framer_gather_trellis_block(ddn_p25p1_framer* f, int block, const uint8_t* d_records10,
                                          const int32_t* d_counts, size_t max_symbols, int16_t* d_llr196,
                                          uint8_t* d_dibit_bits196, uint8_t* d_valid, void* hip_stream);
/* same 98 dibits as one byte per dibit + its reliability byte: ddn_fec_r34_batch / dmr_r34_viterbi_decode_soft input
 * (confirmed-data and MBT blocks coded at rate 3/4, src/protocol/p25/phase1/p25p1_mbf34.c) */
int ddn_p25p1_framer_gather_r34_block(ddn_p25p1_framer* f, int block, const uint8_t* d_records10, const int32_t* d_counts,
                                      size_t max_symbols, uint8_t* d_dibits98, uint8_t* d_reliab98, uint8_t* d_valid,
                                      void* hip_stream);
int ddn_p25p1_framer_gather_ldu_words(ddn_p25p1_framer* f, int ldu, const uint8_t* d_records10, const int32_t* d_counts,
                                      size_t max_symbols, uint8_t* d_bits240, uint8_t* d_reliab240, uint8_t* d_valid,
                                      void* hip_stream);
/* Hamming-corrected LDU words [slots][24][10] -> ddn_fec_p25_rs_batch input: data [slots][12|16][6], parity
 * [slots][12|8][6] (LDU1: RS(24,12,13), p25p1_ldu1.c:233-245; LDU2: RS(24,16,9), p25p1_ldu2.c:256-262) */
int ddn_p25p1_framer_pack_ldu_rs(ddn_p25p1_framer* f, int ldu, const uint8_t* d_words240, uint8_t* d_data_bits,
                                 uint8_t* d_parity_bits, void* hip_stream);
/* HDU (p25p1_hdu.c:191-200,252-270): 36 Golay(24,6) words.  gather_hdu -> hex bits [slots][36][6] and parity bits
 * [slots][36][12] (ddn_fec_golay24_batch / _soft_batch input with data_len 6, n = slots * 36; optional int16 LLRs per bit
 * for the soft variant), word order hex_data[0..19], hex_parity[0..15]; pack_hdu_rs -> RS(36,20,17) input
 * (ddn_fec_p25_rs_batch with DDN_RS_36_20_17): data [slots][20][6], parity [slots][16][6]. */
int ddn_p25p1_layout_hdu(int32_t hex3[108], int32_t par6[216]);
int ddn_p25p1_framer_gather_hdu(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts,
                                size_t max_symbols, uint8_t* d_hex_bits216, uint8_t* d_parity_bits432,
                                int16_t* d_hex_llr216, int16_t* d_parity_llr432, uint8_t* d_valid, void* hip_stream);
int ddn_p25p1_framer_pack_hdu_rs(ddn_p25p1_framer* f, const uint8_t* d_hex_bits216, uint8_t* d_data_bits,
                                 uint8_t* d_parity_bits, void* hip_stream);
/* TDULC (p25p1_tdulc.c:199-224,297): twelve Golay(24,12) words.  gather_tdulc -> data bits [slots][12][12] and parity
 * bits [slots][12][12] (ddn_fec_golay24_batch with data_len 12, n = slots * 12), word order dodeca_data[0..5],
 * dodeca_parity[0..5]; pack_tdulc_rs swaps the hex halves of every word like swap_hex_words() and yields the
 * RS(24,12,13) input data [slots][12][6], parity [slots][12][6]. */
int ddn_p25p1_layout_tdulc(int32_t data6[72], int32_t par6[72]);
int ddn_p25p1_framer_gather_tdulc(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts,
                                  size_t max_symbols, uint8_t* d_data_bits144, uint8_t* d_parity_bits144,
                                  int16_t* d_data_llr144, int16_t* d_parity_llr144, uint8_t* d_valid, void* hip_stream);
int ddn_p25p1_framer_pack_tdulc_rs(ddn_p25p1_framer* f, const uint8_t* d_data_bits144, uint8_t* d_rs_data_bits,
                                   uint8_t* d_rs_parity_bits, void* hip_stream);
/* LDU low speed data (p25p1_ldu1.c:145-183,318-323): 16 dibits = two (16,8) cyclic codewords -> bits [slots][2][16]
 * (+ int16 LLR per bit), the input of ddn_fec_p25_lsd_batch with n = slots * 2 */
int ddn_p25p1_layout_ldu_lsd(int32_t out16[16]);
int ddn_p25p1_framer_gather_lsd(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts,
                                size_t max_symbols, uint8_t* d_bits32, int16_t* d_llr32, uint8_t* d_valid,
                                void* hip_stream);
int ddn_p25p1_framer_imbe_index(ddn_p25p1_framer* f, size_t max_symbols, int64_t* d_first_record,
                                int32_t* d_status_count, void* hip_stream);
/* The same index compacted to voice traffic: for every channel the slots whose decoded NID (d_nid4 [slots][4] from
 * ddn_p25p1_nid_decode_batch: status > 0 (NID_OK or NID_PARITY_OVERRIDE), DUID 0x5 / 0xA - processLDU1 / processLDU2 are the only callers of process_IMBE,
 * src/engine/dispatch/dispatch_p25p1.c) in sync order, nine voice frames each: d_first_record / d_status_count
 * [n_channels][max_ldu_per_channel][9] (unused entries and frames that run past the channel's d_counts records: -1 -> ddn_p25p1_imbe_deinterleave_batch
 * flags them 0xFF),
 * d_n_ldu [n_channels] (optional).  The row of a channel is its talk path for ddn_mbe_synth_batch. */
int ddn_p25p1_framer_voice_index(ddn_p25p1_framer* f, const int32_t* d_nid4, const int32_t* d_counts,
                                 int max_ldu_per_channel, size_t max_symbols,
                                 int64_t* d_first_record, int32_t* d_status_count, int32_t* d_n_ldu, void* hip_stream);

/* ---- rational resampler, batched -------------------------------------------------------------------
 * == dsd_resampler_design + dsd_resampler_process_block (include/dsd-neo/dsp/resampler.h:66-89;
 * src/dsp/resampler.cpp:166-190,241-356), which the demodulator thread applies to the discriminator output when its
 * rate differs from the symbol loop's (src/io/radio/rtl_sdr_fm.cpp:3311-3313): L/M polyphase, 16 taps per phase,
 * Hamming-windowed sinc at 0.45 / max(L, M).  B channels share L, M and the polyphase index; a call is equivalent to one
 * dsd_resampler_process_block() per channel.  d_in [B][n] f32, d_out [B][out_stride] f32 with
 * ddn_resampler_out_len(b, n) <= out_stride outputs per channel (DDN_ERANGE and untouched state otherwise). */
typedef struct ddn_resampler ddn_resampler;
int ddn_resampler_create(int n_channels, int L, int M, ddn_resampler** out); /* 1 <= L <= 512, 1 <= M <= 2^22 */
void ddn_resampler_destroy(ddn_resampler* b);
int ddn_resampler_reset(ddn_resampler* b, void* hip_stream);                 /* == dsd_resampler_clear_history */
size_t ddn_resampler_out_len(const ddn_resampler* b, size_t n);              /* outputs the next run of n inputs gives */
int ddn_resampler_get_taps(const ddn_resampler* b, float* taps, int cap);    /* [L][16], oldest tap first; returns 16 L */
int ddn_resampler_run(ddn_resampler* b, const float* d_in, size_t n, float* d_out, size_t out_stride, void* hip_stream);
int ddn_resampler_run_host(ddn_resampler* b, const float* in, size_t n, float* out, size_t out_stride);
/* drop-in with the reference's name and state layout (include/dsd-neo/dsp/resampler.h:30-42, :89): one stream per call
 * through the same kernel, caller-owned taps / mirrored history updated in place; returns outputs written or -1. */
typedef struct ddn_dsd_resampler_state {
    int enabled, target_hz, L, M, phase, taps_len, taps_per_phase, hist_head;
    float* taps;
    float* hist;
    uint64_t internal_cookie;
} ddn_dsd_resampler_state;
int dsd_resampler_process_block(ddn_dsd_resampler_state* state, const float* in, int in_len, float* out, int out_cap);

/* ---- P25 CQPSK / LSM front end, batched -------------------------------------------------------------------
 * == full_demod(struct demod_state*) with cqpsk_enable (include/dsd-neo/dsp/demod_pipeline.h:106;
 * src/dsp/demod_pipeline.cpp:1100-1118,1330-1350): channel LPF (profile DDN_LPF_P25_CQPSK) -> cqpsk_rms_agc ->
 * op25_fll_band_edge_cc -> op25_gardner_cc -> op25_diff_phasor_cc -> op25_costas_loop_cc -> qpsk_differential_demod
 * (include/dsd-neo/dsp/costas.h) for B channels; all loop state is carried inside the batch object.
 *   d_iq      : [B][n] interleaved I/Q (cu8 or cf32), channel-major; block_len = the reference's block size (LPF edge
 *               rule); every block must hold >= 4 samples
 *   d_symbols : [B][sym_stride] f32 symbols (theta * 4/pi, nominal levels +-1 / +-3), sym_stride >=
 *               ddn_cqpsk_max_symbols(n); d_counts[B] = symbols produced by this call
 * get_state out8 = {agc_avg, fll.freq, fll.phase, costas.phase, costas.freq, costas.error_smooth, ted.mu, ted.omega} */
typedef struct ddn_cqpsk_config {
    int n_channels;
    int sample_rate_hz; /* rate_out: 24000 (sps 5) or 48000 (sps 10) */
    int symbol_rate_hz; /* 4800 (6000 for P25p2) */
    int lpf_profile;    /* DDN_LPF_P25_CQPSK */
    int lpf_enable;
    int input_format;   /* DDN_IN_CU8 / DDN_IN_CF32 */
    int block_len;
    float ted_gain;     /* 0 = the reference default */
} ddn_cqpsk_config;
typedef struct ddn_cqpsk_batch ddn_cqpsk_batch;
int ddn_cqpsk_batch_create(const ddn_cqpsk_config* cfg, ddn_cqpsk_batch** out);
void ddn_cqpsk_batch_destroy(ddn_cqpsk_batch* b);
int ddn_cqpsk_batch_reset(ddn_cqpsk_batch* b, void* hip_stream);
size_t ddn_cqpsk_max_symbols(const ddn_cqpsk_batch* b, size_t n);
int ddn_cqpsk_run(ddn_cqpsk_batch* b, const void* d_iq, size_t n, float* d_symbols, size_t sym_stride,
                  int32_t* d_counts, void* hip_stream);
int ddn_cqpsk_run_host(ddn_cqpsk_batch* b, const void* iq, size_t n, float* symbols, size_t sym_stride,
                       int32_t* counts);
int ddn_cqpsk_get_state(ddn_cqpsk_batch* b, int channel, float out8[8]);

/* ---- the symbol-rate receive loop behind the CQPSK demodulator ---------------------------------------------------------------
 * ddn_cqpsk_run() ends at one float per symbol (levels +-1 / +-3).  What dsd-neo's consumer thread does with them until they are
 * capture records - getSymbol()'s symbol-rate fast path (src/dsp/dsd_symbol.c:1581-1624: the slicer thresholds reset to fixed values
 * at every symbol), getFrameSync() on a QPSK profile (src/dsp/dsd_frame_sync.c:3098-3148: 4-level slice :2061-2075, the level window
 * feeding the 1024-deep extrema average :2319-2336, P25 Phase 1 (24) / Phase 2 (20) sync compared exactly :698-716 / :801-816 and
 * under the rotated constellations X2400 / N1200 / P1200 with the raw-level fit :417-548, :668-696) and the in-frame symbol
 * (src/core/frames/dsd_dibit.c:243-275 use_symbol, :951-1000 the fixed CQPSK slice around the running centre + rotation map +
 * polarity, :376-430, :609-721 soft metrics) - for B channels at once, one wavefront per channel.
 *   protocol DDN_CQ_P25P1: the reference's per-DUID handlers run inside the loop and decide how long a frame is read in frame (NID
 *     BCH + Chase, TSDU blocks list-8 + CRC16 until the last-block flag, data-unit header; lock_symbols = 0) - their decisions and
 *     payloads go to the event list exactly as ddn_p25_rx_set_events / _set_event_data describe; lock_symbols > 0 reads that many
 *     symbols per sync instead.
 *   protocol DDN_CQ_P25P2: 700 dibits per sync (p2_dibit_buffer(), src/protocol/p25/phase2/p25p2_frame.c:352-370).
 * Output = the C4FM loop's: d_records10 [B][max_symbols][10] {dibit, reliability, llr0 i16, llr1 i16, symbol f32} (hunting symbols:
 * the raw 4-level dibit, zeros), d_flags [B][max_symbols] bit 0 in frame, bit 1 a sync completed on this symbol, bit 2 inverted
 * polarity, bits 4-6 (with bit 1) the rotation map the sync was found under (DSD_P25_CQPSK_DIBIT_MAP_*); d_counts [B].  A call
 * consumes d_counts_in[c] symbols of row c (NULL: n for every channel) and writes as many records; state carries across calls.
 * snr_cqpsk_db: what dsd_rtl_stream_metrics_hook_snr_cqpsk_db() would report - in the reference an asynchronous estimate of the radio
 * thread; 0 or <= -50 = not available (no reliability weight, dsd_dibit.c:411-413).  Modulation is locked (as -mq): no voting. */
enum { DDN_CQ_P25P1 = 0, DDN_CQ_P25P2 = 1 };
typedef struct ddn_cq_rx_config {
    int n_channels;
    int protocol;              /* DDN_CQ_* */
    int lock_symbols;          /* P25p1: 0 = handlers; Phase 2: 0 = 700 */
    int nid_erasure_threshold; /* 0 = 64 */
    float snr_cqpsk_db;
} ddn_cq_rx_config;
typedef struct ddn_cq_rx ddn_cq_rx;
int ddn_cq_rx_create(const ddn_cq_rx_config* cfg, ddn_cq_rx** out);
void ddn_cq_rx_destroy(ddn_cq_rx* b);
int ddn_cq_rx_reset(ddn_cq_rx* b, void* hip_stream);
/* d_events i32 [B][max_events][4] {record index in this call, kind, a, b}, d_n_events i32 [B] (per call), d_event_data i32
 * [B][max_events][4] or NULL: kinds and words as ddn_p25_rx_set_events / ddn_p25_rx_set_event_data */
int ddn_cq_rx_set_events(ddn_cq_rx* b, int32_t* d_events, int32_t* d_n_events, int32_t* d_event_data, size_t max_events);
int ddn_cq_rx_run(ddn_cq_rx* b, const float* d_symbols, const int32_t* d_counts_in, size_t n, size_t sym_stride, uint8_t* d_records10,
                  uint8_t* d_flags, int32_t* d_counts, size_t max_symbols, void* hip_stream);
/* {centre, max, min, map index, last sync (0 none, 1 +, 2 -), in frame, hunted symbols, extrema window index} of one channel */
int ddn_cq_rx_get_state(ddn_cq_rx* b, int channel, float out8[8]);

/* ---- batched trellis / Viterbi decoders (bit-exact integer) ------------------------------------------
 * d_* = device pointers, asynchronous on hip_stream; *_host = host pointers, synchronous.
 *   ddn_fec_p25_12_soft_*   P25 1/2-rate 4-state trellis on bit LLRs: [n][196] int16 -> [n][12] bytes (+ metric>>8)
 *                           == p25_12_soft_llr (include/dsd-neo/protocol/p25/p25_12.h:21)
 *   ddn_fec_r34_*           3/4-rate 8-state trellis: [n][98] dibits (+ optional [n][98] reliabilities) -> [n][18]
 *                           == dmr_r34_viterbi_decode / _decode_soft (include/dsd-neo/protocol/dmr/r34_viterbi.h:19-26)
 *   ddn_fec_nxdn_conv_*     K=5 R=1/2, uint16 wrapping metrics: [n][n_steps][2] symbols 0..2 (+ optional reliabilities)
 *                           -> [n][out_stride] bytes holding n_bits chained-back bits, MSB first; metrics_io ([n][16],
 *                           optional) carries the path metrics in and out like the reference's static state
 *                           == CNXDNConvolution_start/decode[_soft]/chainback (.../nxdn/nxdn_convolution.h:24-28)
 *   ddn_fec_viterbi_k5_*    K=5 R=1/2, uint32 metrics on uint16 soft bits, optional puncture pattern (host array):
 *                           [n][in_len] -> [n][out_stride] bytes (bit position = step + 4), cost per codeword
 *                           == viterbi_decode / viterbi_decode_punctured (include/dsd-neo/fec/viterbi.h:23-25)   */
int ddn_fec_p25_12_soft_batch(const int16_t* d_llr196, size_t n, uint8_t* d_out12, int32_t* d_metric, void* hip_stream);
/* list variant == p25_12_soft_llr_list (include/dsd-neo/protocol/p25/p25_12.h:19-32; src/protocol/p25/p25_12.c:144-202):
 * 8 survivors per state, candidates de-duplicated by their 12 bytes and sorted by metric.  candidates8 is [n][8] of
 * the reference's p25_12_candidate_t layout, entries >= counts[i] zeroed; max_candidates is clamped to 8. */
typedef struct ddn_p25_12_candidate {
    uint8_t bytes[12];
    uint32_t metric;
} ddn_p25_12_candidate;
int ddn_fec_p25_12_soft_list_batch(const int16_t* d_llr196, size_t n, int max_candidates,
                                   ddn_p25_12_candidate* d_candidates8, int32_t* d_counts, void* hip_stream);
int ddn_fec_p25_12_soft_list_host(const int16_t* llr196, size_t n, int max_candidates, ddn_p25_12_candidate* candidates8,
                                  int32_t* counts);
int ddn_fec_p25_12_soft_host(const int16_t* llr196, size_t n, uint8_t* out12, int32_t* metric);
int ddn_fec_r34_batch(const uint8_t* d_dibits98, const uint8_t* d_reliab98, size_t n, uint8_t* d_out18,
                      void* hip_stream);
int ddn_fec_r34_host(const uint8_t* dibits98, const uint8_t* reliab98, size_t n, uint8_t* out18);
/* list variant == dmr_r34_viterbi_decode_list (include/dsd-neo/protocol/dmr/r34_viterbi.h:51-70;
 * src/protocol/dmr/dmr_34_viterbi.c:255-362,446-474): 32 survivors per state, candidates = the survivors ending in
 * state 0 in increasing metric, no de-duplication.  candidates32 is [n][32] of the reference's dmr_r34_candidate
 * layout (entries >= counts[i] zeroed); reliab98 NULL = unweighted costs; max_candidates clamped to 32. */
typedef struct ddn_r34_candidate {
    int32_t metric;
    uint8_t bytes18[18];
} ddn_r34_candidate;
/* P25 Phase 1 confirmed data: the rate 3/4 blocks' LLR list decoder == p25_mbf34_decode_soft_list (include/dsd-neo/protocol/p25/
 * p25p1_mbf34.h:19-29, src/protocol/p25/phase1/p25p1_mbf34.c:129-213): 98 LLR pairs in received order -> up to 8 candidates
 * {18 bytes: DBSN(7) | CRC9 | 16 payload bytes, metric}, cheapest first, ties by arrival, identical payloads once; d_counts [n].
 * d_wanted u8 [n] (optional): only items with a non-zero byte are decoded, the others get count 0. */
typedef struct ddn_p25_mbf34_candidate {
    uint8_t bytes[18];
    uint32_t metric;
} ddn_p25_mbf34_candidate; /* == p25_mbf34_candidate_t */
int ddn_fec_p25_mbf34_list_batch(const int16_t* d_llr196, size_t n, int max_candidates, const uint8_t* d_wanted,
                                 ddn_p25_mbf34_candidate* d_candidates8, int32_t* d_counts, void* hip_stream);
int ddn_fec_p25_mbf34_list_host(const int16_t* llr196, size_t n, int max_candidates, ddn_p25_mbf34_candidate* candidates8,
                                int32_t* counts);
int p25_mbf34_decode_soft_list(const uint8_t dibits[98], const int16_t bit_llr[196], ddn_p25_mbf34_candidate* candidates,
                               int max_candidates);
int ddn_fec_r34_list_batch(const uint8_t* d_dibits98, const uint8_t* d_reliab98, size_t n, int max_candidates,
                           ddn_r34_candidate* d_candidates32, int32_t* d_counts, void* hip_stream);
int ddn_fec_r34_list_host(const uint8_t* dibits98, const uint8_t* reliab98, size_t n, int max_candidates,
                          ddn_r34_candidate* candidates32, int32_t* counts);
int ddn_fec_nxdn_conv_batch(const uint8_t* d_sym, const uint8_t* d_rel, size_t n, int n_steps, int n_bits,
                            uint16_t* d_metrics_io, uint8_t* d_out, int out_stride, void* hip_stream);
int ddn_fec_nxdn_conv_host(const uint8_t* sym, const uint8_t* rel, size_t n, int n_steps, int n_bits,
                           uint16_t* metrics_io, uint8_t* out, int out_stride);
int ddn_fec_viterbi_k5_batch(const uint16_t* d_soft, size_t n, int in_len, const uint8_t* punct, int p_len,
                             uint8_t* d_out, int out_stride, uint32_t* d_cost, void* hip_stream);
int ddn_fec_viterbi_k5_host(const uint16_t* soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* out,
                            int out_stride, uint32_t* cost);

/* ---- P25 Phase 1 block codes ----------------------------------------------------------------------------
 *   ddn_p25p1_nid_decode_*   NID: BCH(63,16,11) + DUID/parity validation + observed-NAC retry + Chase search over
 *                            the least reliable bits.  bits63 [n][63] one bit per byte (data first), rel63 [n][63]
 *                            reliabilities (NULL = hard decision only), observed_nac [n] (NULL/0 = none),
 *                            parity [n], parity_rel [n]; out4 [n][4] = {status, nac, duid, error_count} with
 *                            status 0 fail / 1 ok / 2 parity override.  erasure_threshold: the reference default is 64
 *                            == p25p1_nid_decode (include/dsd-neo/protocol/p25/p25p1_check_nid.h:38-39)
 *   ddn_fec_hamming_10_6_3_* Hamming(10,6,3): bits10 [n][10] (6 data + 4 parity, bit per byte), data corrected in
 *                            place on single errors, errs [n] = 0/1/2 == hamming_10_6_3_decode
 *   ddn_fec_golay24_*        Golay(24,12,8) (data_len 12) and the P25 shortened (18,6,8) (data_len 6): data bits
 *                            [n][data_len] (char-array order of the reference, one bit per byte) corrected in place,
 *                            parity [n][12] (11 check bits + overall parity), status [n] 0 ok / 1 irrecoverable,
 *                            fixed [n] (optional) the reference's *fixed_errors
 *                            == check_and_fix_golay_24_6 / _24_12 (include/dsd-neo/protocol/p25/p25p1_check_hdu.h)
 *   ddn_fec_p25_rs_*         RS over GF(64): data bits [n][n_data][6] MSB first, corrected in place, parity bits
 *                            [n][n_par][6], status [n] 0 ok / 1 irrecoverable (data then unchanged)
 *                            == check_and_fix_reedsolomon_24_12_13 / _24_16_9 (p25p1_check_ldu.h),
 *                               check_and_fix_redsolomon_36_20_17 (p25p1_check_hdu.h)   (hard decision; the erasure
 *                               variants are ddn_fec_p25_rs_soft_* further down) */
enum { DDN_RS_24_12_13 = 0, DDN_RS_24_16_9 = 1, DDN_RS_36_20_17 = 2 };
int ddn_fec_golay24_batch(int data_len, uint8_t* d_data_bits, const uint8_t* d_parity12, size_t n, uint8_t* d_status,
                          int32_t* d_fixed, void* hip_stream);
int ddn_fec_golay24_host(int data_len, uint8_t* data_bits, const uint8_t* parity12, size_t n, uint8_t* status,
                         int32_t* fixed);
/* soft (Chase) variants == check_and_fix_golay_24_6_soft / _24_12_soft, hamming_10_6_3_soft
 * (include/dsd-neo/protocol/p25/p25p1_soft.h; src/protocol/p25/phase1/p25p1_soft.cpp:175-593): reliab = int32 per bit
 * (data bits then parity bits; clamped to 0..255 like the reference), hard decode as seed, then every subset of <= 4 of
 * the 8 least reliable bits (Golay) / <= 2 of the 5 least reliable bits (Hamming) through the hard decoder, least
 * reliability penalty wins, hard-decision override margin 8.  Hamming: out10 [n][10], status 0 / 1 / 2. */
int ddn_fec_golay24_soft_batch(int data_len, uint8_t* d_data_bits, const uint8_t* d_parity12, const int32_t* d_reliab,
                               size_t n, uint8_t* d_status, int32_t* d_fixed, void* hip_stream);
int ddn_fec_golay24_soft_host(int data_len, uint8_t* data_bits, const uint8_t* parity12, const int32_t* reliab, size_t n,
                              uint8_t* status, int32_t* fixed);
int ddn_fec_hamming_10_6_3_soft_batch(const uint8_t* d_bits10, const int32_t* d_reliab10, size_t n, uint8_t* d_out10,
                                      uint8_t* d_status, void* hip_stream);
int ddn_fec_hamming_10_6_3_soft_host(const uint8_t* bits10, const int32_t* reliab10, size_t n, uint8_t* out10,
                                     uint8_t* status);
/* == the de-interleave of process_IMBE() (src/protocol/p25/phase1/p25p1_ldu.c:89-120) for n_frames voice frames at
 * once, reading the receive loop's 10-byte capture records in place:
 *   d_records10      flat record array (any number of channels back to back), n_records records long
 *   d_first_record   [n_frames] index of the record holding the frame's first dibit
 *   d_status_count   [n_frames] the reference's status_count on entry (dibits since the last status symbol, 0..35);
 *                    a record is stepped over as status symbol whenever the counter shows 35 (:27-39)
 *   d_imbe_fr        [n_frames][8][23] hard bits (char imbe_fr[8][23]; cells the schedule never writes stay 0)
 *   d_imbe_soft      [n_frames][8][23][2] = dsd_vocoder_soft_bit {bit, reliability = min(|llr|, 255)}
 *                    (include/dsd-neo/core/vocoder.h:25-38)
 *   d_flags          [n_frames] 1 = c0 is the non-standard word the reference skips (:55-66), 0 = normal,
 *                    0xFF = the frame runs past n_records (outputs for the missing dibits are 0)
 *   d_status_count_out [n_frames] counter after the 72 dibits (what the next process_IMBE() call starts from) */
int ddn_p25p1_imbe_deinterleave_batch(const uint8_t* d_records10, size_t n_records, const int64_t* d_first_record,
                                      const int32_t* d_status_count, size_t n_frames, uint8_t* d_imbe_fr,
                                      uint8_t* d_imbe_soft, uint8_t* d_flags, int32_t* d_status_count_out,
                                      void* hip_stream);
int ddn_p25p1_imbe_deinterleave_host(const uint8_t* records10, size_t n_records, const int64_t* first_record,
                                     const int32_t* status_count, size_t n_frames, uint8_t* imbe_fr, uint8_t* imbe_soft,
                                     uint8_t* flags, int32_t* status_count_out);
/* == crc16_lb_bridge (include/dsd-neo/protocol/p25/p25_crc.h; src/protocol/p25/p25_crc.c:18-76): CRC-CCITT16 (0x1021, zero
 * start, inverted) of decoded TSBK / LCCH blocks.  d_bytes [n][item_bytes] = payload then the two CRC bytes (12 for a
 * TSBK out of ddn_fec_p25_12_soft_batch); ok [n] = 1 when the CRC matches.  The drop-in takes the reference's one-bit-per-
 * int payload (byte-aligned lengths up to 240 bits) and returns 0 good / 65535 bad ((uint16_t)-1, as the reference). */
int ddn_fec_p25_crc16_batch(const uint8_t* d_bytes, int item_bytes, size_t n, uint8_t* d_ok, void* hip_stream);
int ddn_fec_p25_crc16_host(const uint8_t* bytes, int item_bytes, size_t n, uint8_t* ok);
int crc16_lb_bridge(const int* payload, int len);
/* == p25_lsd_fec_16x8 / p25_lsd_fec_16x8_soft (include/dsd-neo/protocol/p25/p25_lsd.h; src/protocol/p25/p25_lsd.c:31-160):
 * the (16,8) cyclic code of P25p1 low speed data, g(x) = x^8 + x^5 + x^4 + x^3 + 1.  bits16 [n][16] = 8 data bits then 8
 * parity bits, MSB first, corrected in place; ok [n] = 1 valid / corrected, 0 uncorrectable.  With d_llr16 != NULL a
 * failed hard decode is retried over every subset of the <= 6 least reliable bits whose |llr| < 64 (cheapest wins). */
int ddn_fec_p25_lsd_batch(uint8_t* d_bits16, const int16_t* d_llr16, size_t n, uint8_t* d_ok, void* hip_stream);
int ddn_fec_p25_lsd_host(uint8_t* bits16, const int16_t* llr16, size_t n, uint8_t* ok);
int p25_lsd_fec_16x8(uint8_t* bits16);
int p25_lsd_fec_16x8_soft(uint8_t* bits16, const int16_t llr16[16]);
/* P25 Phase 2 RS(63,35) sections with caller-given erasures == ez_rs28_ess / _facch / _sacch (src/fec/ez.cpp:104-281 over
 * the vendored ezpwd RS<63,35>, src/third_party/ezpwd/rs_base:1380-1720): ESS 16 payload + 28 parity symbols, FACCH 26 + 19
 * (9 parity symbols punctured), SACCH 30 + 22 (6 punctured).  Bits one per byte, most significant bit of a 6-bit symbol
 * first; payload corrected in place.  erasures28 [n][28] (int8) / n_erasures [n]: erased symbol positions in the
 * reference's convention (ESS: 0..43 counted from the first payload symbol; FACCH / SACCH: positions 0..62 of the 63-symbol
 * block, payload at 9.. / 5.., parity at 35.., so the punctured parity is 54..62 / 57..62); positions must be valid and
 * distinct (the reference raises otherwise); n_erasures may be NULL (none).  status [n] = the reference's return value:
 * number of symbols located (errors + erasures), 0 for a clean word, -1 when the word is beyond the code (payload untouched). */
enum { DDN_RS28_ESS = 0, DDN_RS28_FACCH = 1, DDN_RS28_SACCH = 2 };
int ddn_fec_rs28_batch(int kind, uint8_t* d_payload_bits, const uint8_t* d_parity_bits, const int8_t* d_erasures28,
                       const uint8_t* d_n_erasures, size_t n, int32_t* d_status, void* hip_stream);
int ddn_fec_rs28_host(int kind, uint8_t* payload_bits, const uint8_t* parity_bits, const int8_t* erasures28,
                      const uint8_t* n_erasures, size_t n, int32_t* status);
/* P25 Phase 2 FACCH (kind 0) / SACCH (kind 1) burst stage: what p25p2_process_facchc() / process_FACCHs() / process_SACCHc() /
 * process_SACCHs() do between the timeslot's bits and the MAC PDU (src/protocol/p25/phase2/p25p2_frame.c:473-495,534-560,652-671):
 * the RS(63,35) section's payload and parity bits are taken from their places in the 360 bits of the timeslot (p2bit / p2xbit from
 * ts_counter * 360 on: the caller passes the plain or the de-scrambled row), decoded with the fixed erasures, and - when that fails -
 * retried with 1, 2, ... more erasures from the list p25p2_facch_soft_erasures() / p25p2_sacch_soft_erasures() rank by the bits'
 * soft metrics (p25p2_decode_facch_ranked() / _sacch_ranked(), :408-470; src/protocol/p25/phase2/p25p2_soft.c:40-108,255-329;
 * threshold = p25p2_soft_erasure_threshold(), 64 unless configured).  d_bits360 u8 [n][360] one bit per byte, d_llr360 i16 [n][360]
 * (p2llr / p2xllr), d_payload_bits u8 [n][156 | 180] (corrected, or as received when ec < 0), d_ec i32 [n] = the reference's ec,
 * d_used_dynamic u8 [n] = its used_dynamic_erasure. */
/* the decoded section's MAC PDU checksums == p25p2_xcch_validate_facch_crc() / _sacch_crc() (src/protocol/p25/phase2/p25p2_xcch.c:444-497):
 * d_crc12_ok u8 [n] = crc12_xb_bridge(payload, 156 - 12 | 180 - 12) == 0; d_crc16_ok (optional; SACCH on a control channel, LCCH)
 * = crc16_lb_bridge(payload, 164) == 0 (src/protocol/p25/p25_crc.c:17-147); d_payload_bits as ddn_p25p2_xcch_batch writes them */
int ddn_p25p2_mac_crc_batch(int kind, const uint8_t* d_payload_bits, size_t n, uint8_t* d_crc12_ok, uint8_t* d_crc16_ok, void* hip_stream);
int ddn_p25p2_mac_crc_host(int kind, const uint8_t* payload_bits, size_t n, uint8_t* crc12_ok, uint8_t* crc16_ok);
/* P25 Phase 2 ESS == p25p2_ess_decode_with_soft_erasures() (src/protocol/p25/phase2/p25p2_frame.c:1061-1091): payload = the four ESS-B
 * fragments (96 bits, 24 from bit 148 of each 4V burst: p25p2_collect_ess_b_fragment(), :902-915), parity = ESS-A (96 bits from bit
 * 148 + 72 from bit 246 of the 2V burst: p25p2_collect_ess_a(), :1399-1411), with their soft metrics (p2xllr).  The plain RS(44,16)
 * decode stands when it located fewer than 15 symbols, else retries with 1, 2, ... erasures of p25p2_ess_soft_erasures_ranked()'s
 * list (p25p2_soft.c:331-383).  d_payload_out96 = corrected (d_ec >= 0) or as received; d_used_dynamic = a retry decoded it. */
int ddn_p25p2_ess_batch(const uint8_t* d_payload_bits96, const int16_t* d_payload_llr96, const uint8_t* d_parity_bits168,
                        const int16_t* d_parity_llr168, size_t n, int threshold, uint8_t* d_payload_out96, int32_t* d_ec,
                        uint8_t* d_used_dynamic, void* hip_stream);
int ddn_p25p2_ess_host(const uint8_t* payload_bits96, const int16_t* payload_llr96, const uint8_t* parity_bits168,
                       const int16_t* parity_llr168, size_t n, int threshold, uint8_t* payload_out96, int32_t* ec, uint8_t* used_dynamic);
/* P25 Phase 2 4V / 2V bursts == p25p2_unpack_voice_frames() (p25p2_frame.c:250-262,849-900): frame_count (4 / 2) AMBE 3600x2450 frames
 * of 72 bits from bit 2 / 76 / 172 / 246 of the de-scrambled timeslot -> d_ambe_fr u8 [n][frame_count][4][24] + d_ambe_rel (the soft
 * bits' reliabilities, min(|LLR|, 255)): the input of ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, ..). */
int ddn_p25p2_voice_frames_batch(const uint8_t* d_xbits360, const int16_t* d_xllr360, size_t n, int frame_count, uint8_t* d_ambe_fr,
                                 uint8_t* d_ambe_rel, void* hip_stream);
/* P25 Phase 2 frame scrambler == p25p2_generate_scramble_bits() (src/protocol/p25/phase2/p25p2_scramble.c:12-26: 44-bit LFSR seeded with
 * wacn << 24 | sysid << 12 | nac; d_seed44 u64 [n] holds that value, d_out_bits u8 [n][bit_count]) and the de-scrambling of
 * process_Frame_Scramble() (p25p2_frame.c:370-392): xbit[i] = bit[i] ^ sequence[(i + 20 + 360 * offset) mod 4320], the soft metric's
 * sign flipped with it (d_llr / d_xllr i16 [n][n_llr], n_llr <= n_bits: the reference keeps metrics for the first 1400 bits of its
 * 4300); d_scramble4320 u8 [..][4320]; d_sequence_of i32 [n] (optional) picks each item's sequence row (channels of one system share
 * theirs), NULL = row i; d_offset i32 [n] = state->p2_scramble_offset. */
int ddn_p25p2_scramble_bits_batch(const uint64_t* d_seed44, size_t n, size_t bit_count, uint8_t* d_out_bits, void* hip_stream);
int ddn_p25p2_descramble_batch(const uint8_t* d_bits, const int16_t* d_llr, const uint8_t* d_scramble4320, const int32_t* d_offset,
                               const int32_t* d_sequence_of, size_t n, int n_bits, int n_llr, uint8_t* d_xbits, int16_t* d_xllr,
                               void* hip_stream);
void p25p2_generate_scramble_bits(uint64_t wacn, uint64_t sysid, uint64_t nac, uint8_t* out_bits, size_t bit_count);
/* the timeslot's DUID and I-ISCH: d_duid i32 [n] = p25p2_duid_lookup_soft() over the eight DUID bits (p25p2_frame.c:208-248,1462-1478;
 * 0..15, -1 rejected), d_isch i32 [n] = isch_lookup_soft() over bits 320..359 (p25p2_process_isch(), :708-745; the 7-bit value, -2 for
 * the S-ISCH word or nothing within reach) - reliabilities min(|LLR|, 255) as p25p2_reliability_for_abs_bit() */
int ddn_p25p2_burst_fields_batch(const uint8_t* d_bits360, const int16_t* d_llr360, size_t n, int threshold, int32_t* d_duid,
                                 int32_t* d_isch, void* hip_stream);
int ddn_p25p2_burst_fields_host(const uint8_t* bits360, const int16_t* llr360, size_t n, int threshold, int32_t* duid, int32_t* isch);
int ddn_p25p2_xcch_batch(int kind, const uint8_t* d_bits360, const int16_t* d_llr360, size_t n, int threshold, uint8_t* d_payload_bits,
                         int32_t* d_ec, uint8_t* d_used_dynamic, void* hip_stream);
int ddn_p25p2_xcch_host(int kind, const uint8_t* bits360, const int16_t* llr360, size_t n, int threshold, uint8_t* payload_bits,
                        int32_t* ec, uint8_t* used_dynamic);
/* P25 Phase 2 above the bursts == processP2() on the 700 dibits behind a sync (src/protocol/p25/phase2/p25p2_frame.c:1760-1798),
 * batched over channels x groups.  A group = 1400 bits + metrics as p2_dibit_buffer() leaves them in p2bit / p2llr (:354-370; the
 * fourth timeslot's ISCH is not captured: zeros, reliability 0).  Per group: the four ISCH words move the channel's
 * p2_scramble_offset when one is a channel-1 I-ISCH (p25p2_process_isch(), :708-745), the first timeslot's logical channel is
 * offset % 2, the bits are de-scrambled at offset (process_Frame_Scramble(), :372-392), and each timeslot's DUID is dispatched
 * (p25p2_process_duid(), :1742-1760; p25p2_duid_dispatch(), :1580-1640): 4V / 2V voice (frames unpacked, ESS-B fragment filed under
 * the slot's fourv_counter, ESS decoded at the 2V burst, :902-925,1377-1452), SACCH / FACCH / LCCH clear or scrambled (RS(63,35) with
 * the ranked retries, MAC CRC-12 / LCCH CRC-16), bursts that need a valid site skipped without one (:1455-1459), an unknown DUID
 * counted - the second one ends the group and zeroes both 4V counters (:1642-1655).
 * Carried per channel across calls: ddn_p25p2_seq_state (all zeros = a fresh channel, p25_p2_frame_reset()).
 * d_bits1400 u8 / d_llr1400 i16 [n_channels][n_groups][1400]; d_groups_of i32 [n_channels] (optional, NULL = n_groups each) = how many
 * of a channel's n_groups places hold a group (the rest are reported "not reached"); d_seed44 u64 [n_channels] = wacn << 24 | sysid <<
 * 12 | cc.
 * Results per timeslot row r = (channel * n_groups + group) * 4 + ts:
 *   d_info i32 [rows][8] = { duid (-1 rejected, -3 not reached), isch (7-bit value, -2 none), scramble offset of the group, logical
 *                            channel 0 / 1 (-1 not reached), DDN_P2_* action, ec (RS return value of the burst / of the ESS),
 *                            fourv_counter the burst met, flags: 1 used_dynamic_erasure | 2 CRC-12 good | 4 CRC-16 good | 8 ESS accepted }
 *   d_payload u8 [rows][180]        the MAC PDU bits of a FACCH (156) / SACCH / LCCH (180) burst, corrected or as received
 *   d_ambe_fr / d_ambe_rel u8 [rows][4][4][24]   the 4 (4V) or 2 (2V) AMBE 3600x2450 frames = ddn_mbe_frame_decode_batch() input
 *   d_ess u8 [rows][96]             the ESS payload a 2V burst decoded (corrected when accepted)
 * Rows a decoder did not write keep what the caller put there (clear them once).  The call waits for the sequencing pass (one
 * hipStreamSynchronize on hip_stream: the decoders are launched over exact counts); the decoders then run on three streams of the
 * calling thread's own, and hip_stream continues behind them (results are in hip_stream's order as for any other call). */
enum { DDN_P2_NONE = 0, DDN_P2_4V, DDN_P2_2V, DDN_P2_SACCH_S, DDN_P2_SACCH_C, DDN_P2_FACCH_C, DDN_P2_FACCH_S, DDN_P2_LCCH_C, DDN_P2_LCCH_S,
       DDN_P2_ERR, DDN_P2_NOSITE };
typedef struct ddn_p25p2_seq_state {
    int32_t offset;        /* state->p2_scramble_offset */
    int32_t fourv[2];      /* state->fourv_counter */
    int32_t reserved;
    uint8_t ess_b[2][96];  /* state->ess_b / ess_b_llr */
    int16_t ess_b_llr[2][96];
} ddn_p25p2_seq_state;
int ddn_p25p2_groups_batch(const uint8_t* d_bits1400, const int16_t* d_llr1400, int n_channels, int n_groups, const int32_t* d_groups_of,
                           const uint64_t* d_seed44, ddn_p25p2_seq_state* d_state, int threshold, int32_t* d_info, uint8_t* d_payload,
                           uint8_t* d_ambe_fr, uint8_t* d_ambe_rel, uint8_t* d_ess, void* hip_stream);
/* The step below the groups, on a channel's dibit stream: the frame search's Phase 2 test is an exact match of the last 20 dibits with
 * P25P2_SYNC "11131131111333133333" or its inverse (frame_sync_try_p25p2(), src/dsp/dsd_frame_sync.c:800-816; include/dsd-neo/core/
 * sync_patterns.h:36-37), a window that must have filled since the search (re)started (frame_sync_match_window_ready(), :352-355);
 * processP2() then takes the next 700 dibits (p2_dibit_buffer()) and the search starts again behind them.  Inverted sync: the dibits
 * are taken through invert_dibit() (src/core/frames/dsd_dibit.c:301-311: dibit ^ 2; the first bit's metric changes sign with it).
 * This is the dibit-level rule only - the symbol-rate receive loop in front of it (timing, thresholds, the CQPSK rotations the radio
 * build tries) is the CQPSK chain's (a7) and not part of this call.
 * d_dibits u8 [n_channels][stride] (0..3), d_llr2 i16 [n_channels][stride][2], n dibits valid per channel; d_cursor_in i32 [n_channels]
 * (optional, NULL = 0) = where each channel's search starts in this buffer.  A group is cut only when its 700 dibits are inside the
 * buffer; d_cursor_out = where the next call's search must start (the unfinished sync's first dibit, or n - 19 when none is open: the
 * caller re-presents the stream from there).  d_group_pos i32 [n_channels][max_groups] = first dibit of each group, d_n_groups i32
 * [n_channels] (syncs beyond max_groups stay for the next call), d_bits1400 / d_llr1400 [n_channels][max_groups][1400] = the input
 * of ddn_p25p2_groups_batch (d_groups_of = d_n_groups). */
int ddn_p25p2_sync_cut_batch(const uint8_t* d_dibits, const int16_t* d_llr2, int n_channels, int n, size_t stride, const int32_t* d_cursor_in,
                             int max_groups, int32_t* d_n_groups, int32_t* d_group_pos, int32_t* d_cursor_out, uint8_t* d_bits1400,
                             int16_t* d_llr1400, void* hip_stream);
/* the two calls with host pointers (staged through the device, synchronous; state = the host's copy of the carried structs, in / out) */
int ddn_p25p2_groups_host(const uint8_t* bits1400, const int16_t* llr1400, int n_channels, int n_groups, const int32_t* groups_of,
                          const uint64_t* seed44, ddn_p25p2_seq_state* state, int threshold, int32_t* info, uint8_t* payload, uint8_t* ambe_fr,
                          uint8_t* ambe_rel, uint8_t* ess);
int ddn_p25p2_sync_cut_host(const uint8_t* dibits, const int16_t* llr2, int n_channels, int n, size_t stride, const int32_t* cursor_in,
                            int max_groups, int32_t* n_groups, int32_t* group_pos, int32_t* cursor_out, uint8_t* bits1400, int16_t* llr1400);
int ez_rs28_ess(int payload[96], int parity[168], const int* erasures, int n_erasures);
int ez_rs28_facch(int payload[156], int parity[114], const int* erasures, int n_erasures);
int ez_rs28_sacch(int payload[180], int parity[132], const int* erasures, int n_erasures);

/* P25 Phase 2 I-ISCH (40,9,16) lookup == isch_lookup / isch_lookup_soft (src/fec/ez.cpp:283-384; declared in
 * include/dsd-neo/fec/ez.h).  words[i] holds the 40 received bits (first bit = bit 39); the answer is the 7-bit ISCH value,
 * or -2 for the S-ISCH sync word and for anything further than 7 bits from every entry.  reliab40 (n rows of 40 bytes, row
 * byte b = reliability of bit 39-b) switches the whole batch to the soft rule (least summed reliability of the differing
 * bits, then fewest bits, then lowest value); NULL = the hard rule (nearest; the one possible 7 / 7 tie, codeword against
 * S-ISCH, goes the way the reference's map walk takes it).  Exact matches are authoritative in both.  _batch: device
 * pointers, asynchronous on stream; _host and the two reference-named single-word calls stage through the device. */
int ddn_fec_isch_lookup_batch(const uint64_t* d_words, const uint8_t* d_reliab40, size_t n, int32_t* d_out, void* stream);
int ddn_fec_isch_lookup_host(const uint64_t* words, const uint8_t* reliab40, size_t n, int32_t* out);
int isch_lookup(uint64_t isch);
int isch_lookup_soft(uint64_t isch, const uint8_t reliab40[40]);
int ddn_fec_p25_rs_batch(int code, uint8_t* d_data_bits, const uint8_t* d_parity_bits, size_t n, uint8_t* d_status,
                         void* hip_stream);
int ddn_fec_p25_rs_host(int code, uint8_t* data_bits, const uint8_t* parity_bits, size_t n, uint8_t* status);
/* == p25p1_rs_24_12_13_soft_reliability / _24_16_9_ / _36_20_17_ (include/dsd-neo/protocol/p25/p25p1_soft.h:90-110):
 * hard decode, then errors-and-erasures decodes with the n = 1, 2, ... least reliable symbols erased (ranked by
 * (reliability, position); n up to max(#symbols below the erasure threshold 64, t), capped at 2t).  data_reliab
 * [n][n_data], parity_reliab [n][n_par] one byte per symbol.  status 0 ok / 1 irrecoverable (data unchanged). */
int ddn_fec_p25_rs_soft_batch(int code, uint8_t* d_data_bits, const uint8_t* d_parity_bits, const uint8_t* d_data_reliab,
                              const uint8_t* d_parity_reliab, size_t n, uint8_t* d_status, void* hip_stream);
int ddn_fec_p25_rs_soft_host(int code, uint8_t* data_bits, const uint8_t* parity_bits, const uint8_t* data_reliab,
                             const uint8_t* parity_reliab, size_t n, uint8_t* status);
int ddn_p25p1_nid_decode_batch(const uint8_t* d_bits63, const uint8_t* d_rel63, const int32_t* d_observed_nac,
                               const uint8_t* d_parity, const uint8_t* d_parity_rel, int erasure_threshold, size_t n,
                               int32_t* d_out4, void* hip_stream);
int ddn_p25p1_nid_decode_host(const uint8_t* bits63, const uint8_t* rel63, const int32_t* observed_nac,
                              const uint8_t* parity, const uint8_t* parity_rel, int erasure_threshold, size_t n,
                              int32_t* out4);
int ddn_p25p1_nid_decode(const char bch_code[63], const uint8_t* reliab63, int observed_nac, unsigned char parity,
                         uint8_t parity_reliab, int erasure_threshold, int out4[4]);
int ddn_fec_hamming_10_6_3_batch(uint8_t* d_bits10, size_t n, uint8_t* d_errs, void* hip_stream);
int ddn_fec_hamming_10_6_3_host(uint8_t* bits10, size_t n, uint8_t* errs);
int hamming_10_6_3_decode(char* data, const char* parity);
/* include/dsd-neo/protocol/p25/p25p1_check_hdu.h, p25p1_check_ldu.h (one codeword per call, reference signatures) */
int p25p1_rs_24_12_13_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                       const uint8_t* parity_reliab);
int p25p1_rs_24_16_9_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                      const uint8_t* parity_reliab);
int p25p1_rs_36_20_17_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                       const uint8_t* parity_reliab);
int check_and_fix_golay_24_6_soft(char* data, const char* parity, const int* reliab, int* fixed);
int check_and_fix_golay_24_12_soft(char* data, const char* parity, const int* reliab, int* fixed);
int hamming_10_6_3_soft(const char* bits, const int* reliab, char* out_bits);
int check_and_fix_golay_24_6(char* hex, const char* parity, int* fixed_errors);
int check_and_fix_golay_24_12(char* dodeca, const char* parity, int* fixed_errors);
int check_and_fix_reedsolomon_24_12_13(char* data, const char* parity);
int check_and_fix_reedsolomon_24_16_9(char* data, const char* parity);
int check_and_fix_redsolomon_36_20_17(char* data, const char* parity);

/* single-codeword drop-ins with the reference's names */
int p25_12_soft_llr_list(const uint8_t* input, const int16_t* bit_llr196, ddn_p25_12_candidate* candidates,
                         int max_candidates);
int p25_12_soft_llr(const uint8_t* input, const int16_t* bit_llr196, uint8_t treturn[12]);
int dmr_r34_viterbi_decode_list(const uint8_t* dibits98, const uint8_t* reliab98, ddn_r34_candidate* out_candidates,
                                int max_candidates, int* out_count);
int dmr_r34_viterbi_decode(const uint8_t* dibits98, uint8_t out_bytes18[18]);
int dmr_r34_viterbi_decode_soft(const uint8_t* dibits98, const uint8_t* reliab98, uint8_t out_bytes18[18]);
uint32_t viterbi_decode(uint8_t* out, const uint16_t* in, const uint16_t len);
uint32_t viterbi_decode_punctured(uint8_t* out, const uint16_t* in, const uint8_t* punct, const uint16_t in_len,
                                  const uint16_t p_len);
/* step-wise form (include/dsd-neo/fec/viterbi.h:26-28; src/core/util/dsd_misc.c:188-283): per-thread symbol history,
 * decoded on the device at chainback */
void viterbi_decode_bit(uint16_t s0, uint16_t s1, const size_t pos);
uint32_t viterbi_chainback(uint8_t* out, size_t pos, uint16_t len);
void viterbi_reset(void);
void CNXDNConvolution_init(void);
void CNXDNConvolution_start(void);
void CNXDNConvolution_decode(uint8_t s0, uint8_t s1);
void CNXDNConvolution_decode_soft(uint8_t s0, uint8_t s1, uint8_t r0, uint8_t r1);
void CNXDNConvolution_chainback(unsigned char* out, unsigned int nBits);

/* ---- drop-in single-stream symbols (reference names; host pointers) ------------------------------- */
void simd_fir_complex_apply(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                            int taps_len);
int simd_hb_decim2_complex(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                           int taps_len);
int simd_hb_decim2_real(const float* in, int in_len, float* out, float* hist, const float* taps, int taps_len);
const char* simd_fir_get_impl_name(void);
void widen_u8_to_f32_bias127(const unsigned char* src, float* dst, uint32_t len);
/* == include/dsd-neo/core/input_level.h:61-68 (raw CU8 byte moments; every I and Q byte is one sample) */
typedef struct dsd_input_level_cu8_moments {
    uint64_t count, sum, sum_sq, clipped;
    uint8_t min_sample, max_sample;
} dsd_input_level_cu8_moments;
/* == include/dsd-neo/dsp/simd_widen.h:61-62, 72, 89-90 (src/dsp/simd_widen.cpp:151-204): widen + merge the block's raw
 * moments into *moments by dsd_input_level_cu8_moments_merge's rules; the rotate variants multiply pair n by j^(phase+n)
 * and return the phase after the block (an odd trailing byte is not consumed). */
void widen_u8_to_f32_bias127_moments(const unsigned char* src, float* dst, uint32_t len,
                                     dsd_input_level_cu8_moments* moments);
uint32_t widen_rotate90_u8_to_f32_bias127_phase(const unsigned char* src, float* dst, uint32_t len, uint32_t phase);
uint32_t widen_rotate90_u8_to_f32_bias127_phase_moments(const unsigned char* src, float* dst, uint32_t len,
                                                        uint32_t phase, dsd_input_level_cu8_moments* moments);

/* ---- block codes downstream of the receive loop: DMR / NXDN ------------------------------------------
 * == include/dsd-neo/fec/block_codes.h:19-43 (src/fec/fec.c:133-838), bptc.h:20-26 (src/fec/bptc.c), rs_12_9.h:38-42
 * (src/fec/rs-12-9.c).  One bit per byte like the reference; results are the reference's, quirks included (see
 * dsd-neo_amd/csrc/ddn_fec3.hip).
 *   ddn_fec_block_code_batch   d_bits [n_items][nb_codewords][n] corrected in place, d_ok [n_items] = the bool the reference
 *                              returns; the multi-code-word Hamming forms also write d_decoded [n_items][nb_codewords][k]
 *                              (nb_codewords is 1 for the other codes)
 *   ddn_fec_bptc_196x96_batch  d_in196 [n][196] (deinterleave != 0: still in air order, BPTCDeInterleaveDMRData is applied
 *                              on the fly) -> d_out96 [n][96], d_r3 [n][3], d_errs [n] (irrecoverable Hamming checks of the
 *                              second pass, BPTC_196x96_Extract_Data's return value)
 *   ddn_fec_rs_12_9_batch      d_codewords12 [n][12] corrected in place; d_result [n] = RS_12_9_CORRECT_ERRORS_RESULT_* (0
 *                              also for a zero syndrome), d_errors_found [n], d_syndrome3 [n][3] (optional) */
enum {
    DDN_CODE_HAMMING_7_4 = 0,
    DDN_CODE_HAMMING_12_8 = 1,
    DDN_CODE_HAMMING_13_9 = 2,
    DDN_CODE_HAMMING_15_11 = 3,
    DDN_CODE_HAMMING_16_11_4 = 4,
    DDN_CODE_GOLAY_20_8 = 5,
    DDN_CODE_GOLAY_24_12 = 6,
    DDN_CODE_QR_16_7_6 = 7,
};
int ddn_fec_block_code_batch(int code, uint8_t* d_bits, size_t n_items, int nb_codewords, uint8_t* d_decoded, uint8_t* d_ok,
                             void* hip_stream);
int ddn_fec_block_code_host(int code, uint8_t* bits, size_t n_items, int nb_codewords, uint8_t* decoded, uint8_t* ok);
int ddn_fec_bptc_196x96_batch(const uint8_t* d_in196, int deinterleave, size_t n, uint8_t* d_out96, uint8_t* d_r3,
                              uint32_t* d_errs, void* hip_stream);
int ddn_fec_bptc_196x96_host(const uint8_t* in196, int deinterleave, size_t n, uint8_t* out96, uint8_t* r3, uint32_t* errs);
int ddn_fec_rs_12_9_batch(uint8_t* d_codewords12, size_t n, uint8_t* d_result, uint8_t* d_errors_found, uint8_t* d_syndrome3,
                          void* hip_stream);
int ddn_fec_rs_12_9_host(uint8_t* codewords12, size_t n, uint8_t* result, uint8_t* errors_found, uint8_t* syndrome3);
/* == trellis_decode (include/dsd-neo/fec/trellis.h:22, src/core/util/dsd_misc.c:24-71): the greedy four-bit-lookahead decoder of
 * the K = 5 rate-1/2 code that the NXDN field decoders retry with when the soft Viterbi result fails its CRC
 * (src/protocol/nxdn/nxdn_deperm.c:197-205).  One bit per byte; row i reads source bits [0, 2 * result_len + 6) and writes
 * result_len bits. */
int ddn_fec_trellis_decode_batch(const uint8_t* d_source_bits, int source_stride, size_t n, int result_len,
                                 uint8_t* d_result_bits, int result_stride, void* hip_stream);
int ddn_fec_trellis_decode_host(const uint8_t* source_bits, int source_stride, size_t n, int result_len, uint8_t* result_bits,
                                int result_stride);
void trellis_decode(uint8_t result[], const uint8_t source[], int result_len);
/* drop-ins with the reference's names (single item, host pointers) */
#ifndef __cplusplus
#include <stdbool.h>
#endif
void Hamming_7_4_init(void);
bool Hamming_7_4_decode(unsigned char* rxBits);
void Hamming_12_8_init(void);
bool Hamming_12_8_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords);
void Hamming_13_9_init(void);
bool Hamming_13_9_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords);
void Hamming_15_11_init(void);
bool Hamming_15_11_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords);
void Hamming_16_11_4_init(void);
bool Hamming_16_11_4_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords);
void Golay_20_8_init(void);
bool Golay_20_8_decode(unsigned char* rxBits);
void Golay_24_12_init(void);
bool Golay_24_12_decode(unsigned char* rxBits);
void QR_16_7_6_init(void);
bool QR_16_7_6_decode(unsigned char* rxBits);
void InitAllFecFunction(void);
void BPTCDeInterleaveDMRData(const uint8_t* Input, uint8_t* Output);
uint32_t BPTC_196x96_Extract_Data(uint8_t InputDeInteleavedData[196], uint8_t DMRDataExtracted[96], uint8_t R[3]);
/* DMR embedded signalling BPTC(128,77) and the reverse-channel single-burst BPTC (16 x 2) == BPTC_128x77_Extract_Data /
 * BPTC_16x2_Extract_Data (src/fec/bptc.c:167-258, :278-336; include/dsd-neo/fec/bptc.h).  One byte per bit.  128x77: the 8 x 16
 * matrix row-major in, 77 bits out (72 payload + the 5 CRC bits); errs = rows Hamming(16,11,4) could not correct + failed
 * column parities; an uncorrectable row takes the previous row's decoded bits as in the reference (row 0: zeros - the
 * reference reads an uninitialised buffer there).  16x2: 32 interleaved bits in, 32 out (11 corrected data bits, 5 Hamming
 * bits, 16 parity-row bits); errs = uncorrectable row + parity mismatches in the odd (reverse channel) or even sense; an
 * uncorrectable row is left as received (the reference's output is undefined there). */
int ddn_fec_bptc_128x77_batch(const uint8_t* d_in128, size_t n, uint8_t* d_out77, uint32_t* d_errs, void* stream);
int ddn_fec_bptc_128x77_host(const uint8_t* in128, size_t n, uint8_t* out77, uint32_t* errs);
int ddn_fec_bptc_16x2_batch(const uint8_t* d_in32, size_t n, int parity_odd, uint8_t* d_out32, uint32_t* d_errs, void* stream);
int ddn_fec_bptc_16x2_host(const uint8_t* in32, size_t n, int parity_odd, uint8_t* out32, uint32_t* errs);
uint32_t BPTC_128x77_Extract_Data(uint8_t InputDataMatrix[8][16], uint8_t DMRDataExtracted[77]);
uint32_t BPTC_16x2_Extract_Data(uint8_t InputInterleavedData[32], uint8_t DMRDataExtracted[32], uint32_t ParityCheckTypeOdd);
typedef struct {
    uint8_t data[12];
} rs_12_9_codeword_t;
typedef struct {
    uint8_t data[6];
} rs_12_9_poly_t;
#define RS_12_9_CORRECT_ERRORS_RESULT_NO_ERRORS_FOUND          0
#define RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CORRECTED         1
#define RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CANT_BE_CORRECTED 2
typedef uint8_t rs_12_9_correct_errors_result_t;
void rs_12_9_calc_syndrome(const rs_12_9_codeword_t* codeword, rs_12_9_poly_t* syndrome);
uint8_t rs_12_9_check_syndrome(const rs_12_9_poly_t* syndrome);
rs_12_9_correct_errors_result_t rs_12_9_correct_errors(rs_12_9_codeword_t* codeword, const rs_12_9_poly_t* syndrome,
                                                       uint8_t* errors_found);

typedef struct ddn_fsk_modem_state { /* layout == dsd_fsk_modem_state, include/dsd-neo/dsp/fsk_modem.h:22-36 */
    int cfg_sample_rate_hz, cfg_symbol_rate_hz, cfg_levels, cfg_channel_profile;
    float prev_i, prev_q;
    int have_prev;
    float dc_est;
    float discriminator_peak_est;
} ddn_fsk_modem_state;
int ddn_fsk_modem_discriminator_process(ddn_fsk_modem_state* st, const float* iq_interleaved, int len_interleaved,
                                        float* out_samples, int max_samples);
/* the same under the reference's name (include/dsd-neo/dsp/fsk_modem.h:42; `dsd_fsk_modem_state` == ddn_fsk_modem_state) */
int dsd_fsk_modem_discriminator_process(ddn_fsk_modem_state* st, const float* iq_interleaved, int len_interleaved,
                                        float* out_samples, int max_samples);

#ifdef __cplusplus
}
#endif
#endif /* DDN_HIP_H */
