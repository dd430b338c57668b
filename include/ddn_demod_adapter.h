/* ddn_demod_adapter.h - the single-stream pipeline ABI under the reference's own names, for unit tests and
 * A/B runs: one stream, one block per call, host pointers - per-call granularity is far too fine for a GPU, the batched
 * entry points (ddn_front_end_run, ddn_cqpsk_run, ddn_gardner_run) are the product.
 *
 *   void full_demod(struct demod_state*)        include/dsd-neo/dsp/demod_pipeline.h:106 (src/dsp/demod_pipeline.cpp:1300-1350)
 *   void op25_gardner_cc(struct demod_state*)   include/dsd-neo/dsp/costas.h (src/dsp/costas.cpp:804-858)
 *   dsd_fsk_modem_discriminator_process         is in ddn_hip.h
 *
 * `struct demod_state` here is NOT the reference's private 10 MB object: it holds the members these two entry points read
 * and write on this path, under the reference's member names (include/dsd-neo/dsp/demod_state.h:67-262), so a test written
 * against the reference's struct compiles against this header as long as it touches only these.  The per-stream decoder words
 * the reference keeps inside its struct (FIR history, modem state, loop state) live on the device behind `ddn_adapter`.
 */
#ifndef DDN_DEMOD_ADAPTER_H
#define DDN_DEMOD_ADAPTER_H
#ifdef __cplusplus
extern "C" {
#endif

enum { DSD_DEMOD_OUTPUT_AUDIO_MONITOR = 0, DSD_DEMOD_OUTPUT_FSK_DISCRIMINATOR = 1, DSD_DEMOD_OUTPUT_SYMBOL_CQPSK = 2 };

/* the reference's own sizing (include/dsd-neo/dsp/demod_state.h:28-30) */
#ifndef MAXIMUM_BUF_LENGTH
#define DEFAULT_BUF_LENGTH 16384
#define MAXIMUM_OVERSAMPLE 16
#define MAXIMUM_BUF_LENGTH (MAXIMUM_OVERSAMPLE * DEFAULT_BUF_LENGTH)
#endif

struct demod_state {
    /* out: discriminator samples / CQPSK symbols.  An in-struct array as in the reference (demod_state.h:78), so a test that reads
     * s->result[i] after full_demod(s) needs no allocation of its own (1 MB: allocate the struct on the heap or statically, as the
     * reference's callers do).  The member ORDER and the struct's size are still not the reference's: source-compatible for the
     * members below, not ABI-compatible. */
    float result[MAXIMUM_BUF_LENGTH];
    float* lowpassed;  /* in: interleaved I/Q floats of one block (op25_gardner_cc: also out, the symbols) */
    int lp_len;        /* floats in lowpassed (2 per complex sample) */
    int result_len;
    int rate_in, rate_out;                     /* demod sample rate (rate_out is what the modem / loops are configured with) */
    int output_kind;                           /* DSD_DEMOD_OUTPUT_* */
    int symbol_rate_hz, symbol_levels;
    int channel_lpf_enable, channel_lpf_profile; /* profile = DDN_LPF_* (same numbering as DSD_CH_LPF_PROFILE_*) */
    float channel_squelch_level;               /* 0 = off */
    int cqpsk_enable, ted_enabled, ted_sps;
    float ted_gain;                            /* 0 = the reference default */
    void* ddn_adapter;                         /* owned by the adapter; release with ddn_demod_state_release() */
};

void full_demod(struct demod_state* s);
void op25_gardner_cc(struct demod_state* s);
/* frees what the two calls above created for this stream (and with it the stream's carried state) */
void ddn_demod_state_release(struct demod_state* s);
#ifdef __cplusplus
}
#endif
#endif
