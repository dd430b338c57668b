/* ddn_chain.h - the whole P25 Phase 1 path as one C object: what a dsd-neo host would put in place of its demodulator thread
 * loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516: widen -> full_demod() per block) and of processFrame()'s P25p1 branch
 * (src/engine/protocol_dispatch.c:30-44 -> src/engine/dispatch/dispatch_p25p1.c) for B channels at once:
 *
 *   cu8 / cf32 I/Q  -> front end (ddn_front_end_run)                          widen, channel LPF, FSK discriminator
 *                   -> receive loop with the reference's handlers inside it   getSymbol / getFrameSync / getDibitSoft, per-DUID
 *                      (ddn_p25_rx_run, ddn_p25_rx_set_handlers)              in-frame lengths, TSDU last-block flag
 *                   -> framer (ddn_p25p1_framer_*)                            field gathers at fixed offsets from each sync
 *                   -> NID BCH(63,16,11) + Chase                              p25p1_nid_decode
 *                   -> TSDU blocks 0..2: list-8 half-rate decode, first CRC16-clean candidate   tsbk_decode_repetition_bytes
 *                   -> LDU1 / LDU2: 24 x Hamming(10,6,3) + RS(24,12,13) / RS(24,16,9), low speed data (16,8)
 *                   -> HDU: 36 x Golay(24,6) + RS(36,20,17); TDULC: 12 x Golay(24,12) + RS(24,12,13)
 *                   -> nine IMBE frames per LDU: de-interleave, frame FEC, parameters, synthesis -> f32 PCM
 *
 * Everything stays on the device; the object owns the buffers, the stage order, the double buffering and the HIP streams /
 * events of the two pipelined forms.  Frames that cross a call boundary decode whole: the last carry_symbols records of a call
 * are kept back and decoded with the next call (one call of extra latency for those frames, none lost).
 * Host language: C.  Device pointers in, device pointers out (ddn_p25_chain_results); _run_host adds pinned-host staging with
 * the copies overlapped on a third stream. */
#ifndef DDN_CHAIN_H
#define DDN_CHAIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddn_p25_chain_config {
    int n_channels;
    int samples_per_call; /* complex samples per channel and call (fixed per object) */
    int block_len;        /* front end: samples per reference full_demod() block (8192) */
    int input_format;     /* DDN_IN_CU8 / DDN_IN_CF32 */
    int vocoder;          /* 1 = IMBE synthesis to PCM, 0 = stop after the voice frames' FEC */
    int max_frames;       /* frame slots per channel and call; 0 = samples_per_call / 1800 + 6 */
    int max_ldu;          /* voice LDUs per channel and call; 0 = samples_per_call / 8640 + 3 */
    int max_events;       /* handler decisions per channel and call; 0 = 4 * max_frames */
    int carry_symbols;    /* records carried into the next call; 0 = 896 (an LDU is 864 symbols) */
} ddn_p25_chain_config;

/* device pointers to the outputs of the most recent run (valid until the run after the next; S = n_channels * max_frames frame
 * slots, slot = channel * max_frames + k for the k-th decoded sync of the channel in this call) */
typedef struct ddn_p25_chain_results {
    size_t stride_symbols;     /* records per channel row = carry_symbols + ddn_p25_rx_max_symbols(samples_per_call) */
    const uint8_t* d_records10; /* [B][stride][10]: the carried records, then this call's */
    const uint8_t* d_flags;     /* [B][stride] */
    const int32_t* d_counts;    /* [B] records in the row (carried + new) */
    const int32_t* d_new;       /* [B] new records of this call (they start at index carry_symbols) */
    const int32_t* d_events;    /* [B][max_events][4] handler decisions of this call (ddn_p25_rx_set_events; index + carry_symbols) */
    const int32_t* d_n_events;  /* [B] */
    const int32_t* d_n_syncs;   /* [B] frame slots used */
    const int32_t* d_sync_pos;  /* [S] index of the sync's last symbol in the row */
    const int32_t* d_nid4;      /* [S][4] status, NAC, DUID, corrected bits */
    const uint8_t* d_tsbk;      /* [3][S][12] decoded TSDU blocks 0..2 (a slot's blocks after its last-block flag are not part of the TSDU) */
    const uint8_t* d_tsbk_crc;  /* [3][S] CRC16 good */
    const uint8_t* d_ldu_words[2];  /* [S][24][10] Hamming-corrected words of LDU1 / LDU2 */
    const uint8_t* d_ldu_rs_data[2]; /* [S][12][6] / [S][16][6] after Reed-Solomon */
    const uint8_t* d_ldu_rs_status[2]; /* [S] */
    const uint8_t* d_lsd_bits;  /* [S][2][16] low speed data words */
    const uint8_t* d_lsd_ok;    /* [S][2] */
    const uint8_t* d_hdu_rs_data;   /* [S][20][6] */
    const uint8_t* d_hdu_rs_status; /* [S] */
    const uint8_t* d_tdulc_rs_data; /* [S][12][6] */
    const uint8_t* d_tdulc_rs_status; /* [S] */
    const int32_t* d_n_ldu;     /* [B] voice LDUs of this call */
    const uint8_t* d_imbe_bits; /* [B][max_ldu * 9][88] voice parameter bits */
    const int32_t* d_imbe_result; /* [B][max_ldu * 9][5] */
    const float* d_pcm;         /* [B][max_ldu * 9][160] (vocoder = 1) */
} ddn_p25_chain_results;

typedef struct ddn_p25_chain ddn_p25_chain;

int ddn_p25_chain_create(const ddn_p25_chain_config* cfg, ddn_p25_chain** out);
void ddn_p25_chain_destroy(ddn_p25_chain* c);
/* every stage of one call on one stream (NULL = the default stream) */
int ddn_p25_chain_run(ddn_p25_chain* c, const void* d_iq, void* hip_stream);
/* the same work over the object's two streams: front end + receive loop on one, framer + FEC + voice on the other, so the decode
 * of call k runs beside the front end and loop of call k + 1.  Returns once everything is queued. */
int ddn_p25_chain_run_pipelined(ddn_p25_chain* c, const void* d_iq);
/* pipelined, from pinned host memory: the H2D copy of this call's I/Q and the D2H copy of the previous call's results (any of the
 * out pointers may be NULL) run on a third stream beside the kernels.  h_iq must stay untouched until the next call returns;
 * the outputs named at call k are complete when call k + 2 returns, or after ddn_p25_chain_wait(). */
typedef struct ddn_p25_chain_host_out {
    uint8_t* records10; /* [B][stride][10] */
    uint8_t* flags;     /* [B][stride] */
    int32_t* counts;    /* [B] */
    int32_t* events;    /* [B][max_events][4] */
    int32_t* n_events;  /* [B] */
    int32_t* nid4;      /* [S][4] */
    uint8_t* tsbk;      /* [3][S][12] */
    float* pcm;         /* [B][max_ldu * 9][160] */
} ddn_p25_chain_host_out;
int ddn_p25_chain_run_host(ddn_p25_chain* c, const void* h_iq, const ddn_p25_chain_host_out* out);
/* decode what the carry still holds back (end of a stream): one more decode pass without new samples */
int ddn_p25_chain_flush(ddn_p25_chain* c);
/* block until everything queued by the pipelined forms has run */
int ddn_p25_chain_wait(ddn_p25_chain* c);
int ddn_p25_chain_get_results(ddn_p25_chain* c, ddn_p25_chain_results* out);
/* sizes derived from the configuration */
size_t ddn_p25_chain_stride_symbols(const ddn_p25_chain* c);
int ddn_p25_chain_frame_slots(const ddn_p25_chain* c); /* max_frames as resolved */
int ddn_p25_chain_max_ldu(const ddn_p25_chain* c);
int ddn_p25_chain_max_events(const ddn_p25_chain* c);
/* the stage objects, for timing switches and state queries (ddn_batch*, ddn_p25_rx*, ddn_mbe_batch*) */
void* ddn_p25_chain_front_end(ddn_p25_chain* c);
void* ddn_p25_chain_rx(ddn_p25_chain* c);
void* ddn_p25_chain_mbe(ddn_p25_chain* c);
/* device / pinned-host memory for callers without a HIP binding of their own (synchronous copies) */
int ddn_device_alloc(size_t bytes, void** out);
void ddn_device_free(void* p);
int ddn_device_upload(void* d_dst, const void* h_src, size_t bytes);
int ddn_device_download(void* h_dst, const void* d_src, size_t bytes);
int ddn_host_alloc_pinned(size_t bytes, void** out);
void ddn_host_free_pinned(void* p);
#ifdef __cplusplus
}
#endif
#endif
