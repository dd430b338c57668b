/* ddn_chain.h - the whole P25 Phase 1 path as one C object: what a dsd-neo host would put in place of its demodulator thread
 * loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516: widen -> full_demod() per block) and of processFrame()'s P25p1 branch
 * (src/engine/protocol_dispatch.c:30-44 -> src/engine/dispatch/dispatch_p25p1.c) for B channels at once:
 *
 *   cu8 / cf32 I/Q  -> front end (ddn_front_end_run)                          widen, channel LPF, FSK discriminator
 *                   -> receive loop with the reference's handlers inside it   getSymbol / getFrameSync / getDibitSoft, per-DUID
 *                      (ddn_p25_rx_run, ddn_p25_rx_set_handlers)              in-frame lengths, TSDU last-block flag
 *                   -> framer (ddn_p25p1_framer_*)                            field gathers at fixed offsets from each sync
 *                      (the handlers' own decodes - NID BCH(63,16,11) + Chase, p25p1_nid_decode; TSDU blocks: list-8 half-rate decode,
 *                      first CRC16-clean candidate, tsbk_decode_repetition_bytes - are kept and filed by frame, not repeated)
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

#include "ddn_hip.h" /* DDN_OK ..., DDN_IN_*, ddn_last_error() */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddn_p25_chain_config {
    int n_channels;
    int samples_per_call; /* complex samples per channel and call (fixed per object) */
    int block_len;        /* front end: samples per reference full_demod() block (8192) */
    int input_format;     /* DDN_IN_CU8 / DDN_IN_CF32 */
    int vocoder;          /* 1 = IMBE synthesis to PCM, 0 = stop after the voice frames' FEC */
    int max_frames;       /* frame slots per channel and call; 0 = samples_per_call / 1800 + 6 (back-to-back single-block TSDUs, the
                             densest traffic a control channel carries; a run of 72-symbol TDUs or of false syncs can exceed it -
                             what then finds no slot is counted in d_dropped_syncs, and a host that expects such traffic sets
                             samples_per_call / 720 + 6) */
    int max_ldu;          /* voice LDUs per channel and call; 0 = samples_per_call / 8640 + 3 */
    int max_events;       /* handler decisions per channel and call; 0 = 4 * max_frames */
    int carry_symbols;    /* records carried into the next call; 0 = 960 (an LDU is 864 symbols, a data unit's eighth block ends 940 behind its sync) */
    /* appended in round 5 (a config zero-filled beyond carry_symbols is the C4FM chain as before): */
    int modulation;       /* DDN_P25_MOD_C4FM 0: front end -> FSK discriminator -> matched filter + sample-rate loop (ddn_p25_rx);
                             DDN_P25_MOD_CQPSK 1 (LSM / simulcast sites): CQPSK demodulator (ddn_cqpsk_run: channel LPF, RMS AGC, FLL,
                             Gardner, differential phasor, Costas) -> one symbol per call of the symbol-rate loop (ddn_cq_rx: 4-level
                             slice, sync incl. the rotated-constellation retries, running centre, the same per-DUID handlers) -
                             everything behind the records is the same decode */
    int sample_rate_hz;   /* CQPSK: demodulator rate, 0 = 48000 (10 samples per symbol); 24000 = 5 */
    float snr_cqpsk_db;   /* CQPSK: ddn_cq_rx_config.snr_cqpsk_db (0 = not available) */
    /* appended in round 6: */
    int d2h_blit;         /* _run_host's result copies: 0 = on an SDMA engine below HIP when the result buffers are pinned memory the
                             runtime knows (probed at the first call with result buffers, reported by ddn_p25_chain_d2h_route), 1 =
                             always hipMemcpyAsync on a copy stream (a shader blit: it cannot start while the loop holds every CU) */
} ddn_p25_chain_config;
enum { DDN_P25_MOD_C4FM = 0, DDN_P25_MOD_CQPSK = 1 };

/* device pointers to the outputs of the most recent run (S = n_channels * max_frames frame slots, slot = channel * max_frames + k
 * for the k-th decoded sync of the channel in this call).  Lifetime: d_records10 / d_flags / d_new / d_events / d_n_events /
 * d_event_data are double buffered and stay valid until the run after the next; everything else (counts, sync index, NIDs, TSDU
 * blocks, the per-frame-type outputs, voice bits, PCM) is single buffered and is overwritten by the NEXT run's decode stage - read
 * it (or have _run_host copy it out) before queueing another run, or after ddn_p25_chain_wait() of this one. */
typedef struct ddn_p25_chain_results {
    size_t stride_symbols;     /* records per channel row = carry_symbols + ddn_p25_rx_max_symbols(samples_per_call) */
    const uint8_t* d_records10; /* [B][stride][10]: the carried records, then this call's */
    const uint8_t* d_flags;     /* [B][stride] */
    const int32_t* d_counts;    /* [B] records in the row (carried + new) */
    const int32_t* d_new;       /* [B] new records of this call (they start at index carry_symbols) */
    const int32_t* d_events;    /* [B][max_events][4] handler decisions of this call (ddn_p25_rx_set_events; index + carry_symbols) */
    const int32_t* d_n_events;  /* [B] */
    const int32_t* d_event_data; /* [B][max_events][4] what each decision decoded (ddn_p25_rx_set_event_data) */
    const int32_t* d_n_syncs;   /* [B] frame slots used */
    const int32_t* d_dropped_syncs; /* [B] running count of accepted syncs that found no frame slot (0 unless max_frames is too small) */
    const int32_t* d_sync_pos;  /* [S] index of the sync's last symbol in the row */
    /* the NID and the TSDU blocks are decoded once, by the handlers inside the receive loop (with the reference's running NAC as the
     * decoder's observed NAC), and filed by frame here; a block the handler did not read (behind the last-block flag) is zero */
    const int32_t* d_nid4;      /* [S][4] status, NAC, DUID, corrected bits */
    const uint8_t* d_tsbk;      /* [3][S][12] decoded TSDU blocks 0..2 */
    const uint8_t* d_tsbk_crc;  /* [3][S] CRC16 good */
    /* the per-frame-type outputs below are written for the slots whose NID names that type (DUID 5 / A / 0 / F, status > 0) and are
     * left untouched for every other slot */
    const uint8_t* d_ldu_words[2];  /* [S][24][10] Hamming-corrected words of LDU1 / LDU2 */
    const uint8_t* d_ldu_rs_data[2]; /* [S][12][6] / [S][16][6] after Reed-Solomon */
    const uint8_t* d_ldu_rs_status[2]; /* [S] */
    const uint8_t* d_lsd_bits;  /* [S][2][16] low speed data words */
    const uint8_t* d_lsd_ok;    /* [S][2] */
    const uint8_t* d_hdu_rs_data;   /* [S][20][6] */
    const uint8_t* d_hdu_rs_status; /* [S] */
    const uint8_t* d_tdulc_rs_data; /* [S][12][6] */
    const uint8_t* d_tdulc_rs_status; /* [S] */
    /* data units (DUID 0xC, processMPDU() src/protocol/p25/phase1/p25p1_mdpu.c): entry = channel * pdu_per_channel + rank in air order.
     * The header is the one the loop's handler decoded (list-8 + CRC16, it decides the frame's length); the data blocks behind it are
     * decoded here: half-rate trellis, best path (p25_mpdu_decode_r12_block :263-273), CRC32 over the data (crc32mbf).  When the header
     * fails its CRC16 the reference reads three blocks and tries blocks 1 / 2 as repetitions of the header: so does this.  Confirmed
     * data (A/N = 1, format 0x16 in a header with a good CRC16): the blocks go through the rate 3/4 LLR list decoder (p25p1_mbf34.c),
     * first candidate with a good CRC9, CRC32 over the 16 payload bytes per block.  When none of the three repetitions passes: the
     * three blocks' LLRs summed (saturating) through the half-rate list decoder, first CRC16-clean candidate (:336-360, flag 32), else
     * the bitwise majority of the three decoded repetitions, kept whether or not its CRC16 holds (:362-379, flag 64; + 16 when not). */
    int pdu_per_channel, pdu_blocks;
    const int32_t* d_n_pdu;        /* [B] data units whose sync this call decodes (entries beyond pdu_per_channel are counted only) */
    const int32_t* d_pdu_slot;     /* [B][pdu_per_channel] frame slot, -1 = unused entry */
    const uint8_t* d_pdu_header;   /* [..][12] */
    const int32_t* d_pdu_info;     /* [..][4] {header CRC16 good, blocks read (header included), flags, CRC32 good}; flags: 1 / 2 header
                                      taken from repetition 1 / 2, 4 confirmed data (the blocks are in d_pdu_blocks18), 8 a block beyond the
                                      call's records or beyond pdu_blocks, 16 the header's CRC16 fails whatever was tried, 32 header from the summed LLRs of
                                      the three repetitions, 64 header = bitwise majority of the three decoded repetitions */
    const uint8_t* d_pdu_blocks;   /* [..][pdu_blocks][12] data blocks 1.. */
    const uint8_t* d_pdu_block_valid; /* [..][pdu_blocks] */
    const uint8_t* d_pdu_blocks18; /* [..][pdu_blocks][18] confirmed data (flag 4): DBSN(7) | CRC9, then 16 payload bytes per block */
    const uint8_t* d_pdu_crc9_ok;  /* [..][pdu_blocks] a candidate with a good CRC9 was found (else the cheapest one is stored) */
    const int32_t* d_n_ldu;     /* [B] voice LDUs of this call */
    const uint8_t* d_imbe_bits; /* [B][max_ldu * 9][88] voice parameter bits */
    const int32_t* d_imbe_result; /* [B][max_ldu * 9][5] */
    const float* d_pcm;         /* [B][max_ldu * 9][160] (vocoder = 1) */
    const int32_t* d_synth_result; /* [B][max_ldu * 9][5] result words after synthesis (repeat / mute flags added) */
} ddn_p25_chain_results;

typedef struct ddn_p25_chain ddn_p25_chain;

int ddn_p25_chain_create(const ddn_p25_chain_config* cfg, ddn_p25_chain** out);
void ddn_p25_chain_destroy(ddn_p25_chain* c);
/* every stage of one call on one stream (NULL = the default stream) */
int ddn_p25_chain_run(ddn_p25_chain* c, const void* d_iq, void* hip_stream);
/* the same call in its three stages (0: carry + front end, 1: matched filter + receive loop, 2: framer + FEC + voice; d_iq is read by
 * stage 0 only), for a host that lines the stages of several chain objects up itself - ddn_mixed_chain does */
int ddn_p25_chain_stage(ddn_p25_chain* c, int stage, const void* d_iq, void* hip_stream);
/* the same work over the object's two streams: front end + receive loop on one, framer + FEC + voice on the other, so the decode
 * of call k runs beside the front end and loop of call k + 1.  Returns once everything is queued. */
int ddn_p25_chain_run_pipelined(ddn_p25_chain* c, const void* d_iq);
/* pipelined, from pinned host memory (hipHostMalloc / hipHostRegister / ddn_host_alloc_pinned; any of the out pointers may be NULL; the
 * structure is copied, the buffers it names must stay valid): the H2D copy of this call's I/Q runs on a copy stream beside the
 * previous call's decode; the D2H copies of the PREVIOUS call's results are handed to an SDMA engine at the end of this call (below
 * HIP: hsa_amd_memory_async_copy_on_engine - a device -> host hipMemcpyAsync is a shader kernel on this ROCm and cannot start while
 * the receive loop holds every CU; the engine runs beside any kernel and duplex with the input copy).  This call first queues all
 * of its own work, then waits on the host for the previous call's decode (which runs beside this call's front end) and issues
 * those copies.  The outputs a result set names (counts, NIDs, TSDU blocks, PCM) exist once per buffer set on the device, so a
 * call's results can still be leaving while the next call is decoded.  cfg.d2h_blit = 1 (or result buffers the
 * runtime does not know as pinned) keeps the earlier route: hipMemcpyAsync on a copy stream, released beside this call's receive
 * loop.  ddn_p25_chain_wait() / _flush() issue the copies of the last call and wait for them.  h_iq must stay untouched until
 * the next call returns (that call waits on the host for the copy): two input buffers, used in turn, are enough.  The outputs
 * named at call k are complete when call k + 3 returns (it waits for them on the host before its loop overwrites the device-side
 * buffer set they come from - the object holds three), or after ddn_p25_chain_wait(): THREE output sets used in turn (a host that
 * reads call k's outputs after call k + 3 returned, while later calls run, needs four; one that calls ddn_p25_chain_wait() before
 * reading needs one).  The depth is what keeps the pipe full: an input copy (7 ms at 4096 x 48000), a step (9-10 ms) and a full
 * result set (10 ms on the engine) are in flight at once. */
typedef struct ddn_p25_chain_host_out {
    uint8_t* records10; /* [B][stride][10] */
    uint8_t* flags;     /* [B][stride] */
    int32_t* counts;    /* [B] */
    int32_t* events;    /* [B][max_events][4] */
    int32_t* n_events;  /* [B] */
    int32_t* event_data; /* [B][max_events][4] */
    int32_t* nid4;      /* [S][4] */
    uint8_t* tsbk;      /* [3][S][12] */
    float* pcm;         /* [B][max_ldu * 9][160] */
    uint8_t* records2;  /* [B][stride][2]: {dibit | flags << 2, reliability} - the records as a host consumer of the dibit stream
                           reads them, a fifth of records10 + flags over PCIe (the soft values and the float symbol stay on the device) */
    /* the synthesized frames only, dense and in slot order (slot = channel * max_ldu * 9 + frame slot), instead of every slot's
     * 640 bytes: up to pcm_dense_frames of them are copied (all three pointers set, pcm_dense_frames > 0); *pcm_count is the number
     * the call produced - if it exceeds pcm_dense_frames the rest is still in d_pcm (ddn_p25_chain_get_results) */
    float* pcm_dense;        /* [pcm_dense_frames][160] */
    int32_t* pcm_slot;       /* [pcm_dense_frames] */
    int32_t* pcm_count;      /* [1] */
    int64_t pcm_dense_frames;
} ddn_p25_chain_host_out;
int ddn_p25_chain_run_host(ddn_p25_chain* c, const void* h_iq, const ddn_p25_chain_host_out* out);
/* decode what the carry still holds back (end of a stream): one more decode pass without new samples.  Blocking; it first waits for
 * everything queued before it - the object's own streams and the caller's stream of the last ddn_p25_chain_run / _stage call */
int ddn_p25_chain_flush(ddn_p25_chain* c);
/* block until everything queued so far has run (the pipelined forms' streams and the last _run / _stage caller stream) */
int ddn_p25_chain_wait(ddn_p25_chain* c);
int ddn_p25_chain_get_results(ddn_p25_chain* c, ddn_p25_chain_results* out);
/* sizes derived from the configuration */
size_t ddn_p25_chain_stride_symbols(const ddn_p25_chain* c);
int ddn_p25_chain_frame_slots(const ddn_p25_chain* c); /* max_frames as resolved */
int ddn_p25_chain_max_ldu(const ddn_p25_chain* c);
int ddn_p25_chain_max_events(const ddn_p25_chain* c);
/* which way _run_host's result copies go, once a call with result buffers has been made: 0 = hipMemcpyAsync on a copy stream (shader
 * blit), otherwise an SDMA engine - the hsa_amd_sdma_engine_id_t bit in use, or 0x10000 when the runtime picks the engine */
int ddn_p25_chain_d2h_route(const ddn_p25_chain* c);
/* the stage objects, for timing switches and state queries (ddn_batch*, ddn_p25_rx*, ddn_mbe_batch*) */
void* ddn_p25_chain_front_end(ddn_p25_chain* c);
void* ddn_p25_chain_rx(ddn_p25_chain* c);
void* ddn_p25_chain_mbe(ddn_p25_chain* c);
/* HIP events at the stage boundaries of every call while enabled; _get_stage_ms waits for the most recent call and returns the
 * milliseconds of {front end, receive loop, framer + frame FEC, voice} (meaningful for ddn_p25_chain_run on one stream) */
int ddn_p25_chain_set_timing(ddn_p25_chain* c, int enable);
/* Channel 0 of this object is channel `first` of a set split over several objects (include/ddn_node.h): the one result that depends on
 * a channel's number - the vocoder's unvoiced-noise sequence - follows the global number.  Before the first call only. */
int ddn_p25_chain_set_first_channel(ddn_p25_chain* c, int first);
int ddn_p25_chain_get_stage_ms(ddn_p25_chain* c, float out4[4]);
/* ---- P25 Phase 2: the TDMA channel as one object -----------------------------------------------------------------------------------
 *   cu8 / cf32 I/Q -> CQPSK demodulator at 6000 symbols/s (ddn_cqpsk_run) -> symbol-rate receive loop (ddn_cq_rx, DDN_CQ_P25P2: S-ISCH
 *   sync exact or under the rotated constellations, 700 in-frame dibits per sync) -> the 700 dibits behind every sync
 *   (p2_dibit_buffer(), src/protocol/p25/phase2/p25p2_frame.c:352-370) -> processP2() (ddn_p25p2_groups_batch: I-ISCH, scramble
 *   offset, DUID dispatch, FACCH / SACCH / LCCH with RS(63,35) + MAC CRCs, 4V / 2V + ESS) -> vocoder = 1: the AMBE 3600x2450 frames
 *   of the two logical channels in air order through frame FEC + synthesis (process_4V / process_2V -> processMbeFrame, :1029-1047,
 *   :1435-1460; src/core/vocoder/dsd_mbe.c:172-190), one talk path per logical channel with its parameters carried.
 * One call per batch of samples_per_call samples; every state carries across calls; a group whose 700 dibits cross a call boundary is
 * decoded whole by the next call (the last 720 records are carried), ddn_p25p2_chain_flush() decodes what the carry still holds.
 * seed44[B] = wacn << 24 | sysid << 12 | colour code per channel (the scrambler's seed; 0 = no valid site: scrambled bursts and voice
 * are skipped as the reference skips them).  The calls wait for the sequencing pass on the host once per call (ddn_p25p2_groups_batch). */
typedef struct ddn_p25p2_chain_config {
    int n_channels;
    int samples_per_call;
    int block_len;       /* the demodulator's full_demod() block (8192) */
    int input_format;    /* DDN_IN_CU8 / DDN_IN_CF32 */
    int sample_rate_hz;  /* 0 = 48000 (8 samples per symbol) */
    int vocoder;         /* 1 = AMBE synthesis to PCM */
    int max_groups;      /* syncs decoded per channel and call; 0 = samples_per_call * 6000 / rate / 720 + 3 */
    float snr_cqpsk_db;  /* ddn_cq_rx_config.snr_cqpsk_db (0 = not available) */
} ddn_p25p2_chain_config;
typedef struct ddn_p25p2_chain_results { /* device pointers, valid until the next run */
    size_t stride_symbols;          /* records per channel row = carry_symbols + ddn_cqpsk_max_symbols(samples_per_call) */
    int carry_symbols, max_groups, voice_frames; /* 720; G; AMBE frame slots per talk path and call (8 G) */
    const uint8_t* d_records10;     /* [B][stride][10] the carried records, then this call's (ddn_cq_rx) */
    const uint8_t* d_flags;         /* [B][stride] */
    const int32_t* d_new;           /* [B] new records of this call */
    const int32_t* d_counts;        /* [B] records in the row */
    const int32_t* d_n_groups;      /* [B] syncs decoded in this call */
    const int32_t* d_group_pos;     /* [B][G] row index of each sync's last dibit (its group = the 700 records behind it) */
    const int32_t* d_dropped_syncs; /* [B] running count of syncs that found no place (max_groups too small) */
    const int32_t* d_info;          /* [B][G][4][8] as ddn_p25p2_groups_batch */
    const uint8_t* d_payload;       /* [B][G][4][180] MAC PDU bits */
    const uint8_t* d_ambe_fr;       /* [B][G][4][4][4][24] */
    const uint8_t* d_ambe_rel;
    const uint8_t* d_ess;           /* [B][G][4][96] */
    const int32_t* d_voice_src;     /* [2 B][voice_frames] timeslot row * 4 + frame of every synthesized frame, talk path = channel * 2 + slot */
    const int32_t* d_voice_count;   /* [2 B] */
    const uint8_t* d_voice_bits;    /* [2 B][voice_frames][49] */
    const int32_t* d_voice_result;  /* [2 B][voice_frames][5] */
    const float* d_pcm;             /* [2 B][voice_frames][160] */
} ddn_p25p2_chain_results;
typedef struct ddn_p25p2_chain ddn_p25p2_chain;
int ddn_p25p2_chain_create(const ddn_p25p2_chain_config* cfg, const uint64_t* seed44, ddn_p25p2_chain** out);
void ddn_p25p2_chain_destroy(ddn_p25p2_chain* c);
int ddn_p25p2_chain_run(ddn_p25p2_chain* c, const void* d_iq, void* hip_stream);
int ddn_p25p2_chain_flush(ddn_p25p2_chain* c, void* hip_stream);
int ddn_p25p2_chain_get_results(ddn_p25p2_chain* c, ddn_p25p2_chain_results* out);
/* ---- DMR / NXDN48: the same shape for BASELINE configs[3]'s other two protocols ------------------------------------------------
 *   cu8 / cf32 I/Q -> front end (12.5 kHz / 6.25 kHz channel filter) -> matched filter + receive loop (ddn_fsk4_rx_run; handlers = 1:
 *   dmr_data_sync / dmrBSBootstrap + dmrBS / nxdn_frame's LICH gate decide the in-frame lengths inside the loop)
 *   DMR:    burst gather -> slot type Golay(20,8) -> BPTC(196,96); the data bursts the handlers dispatch: type CRC / RS(12,9) full
 *           link control / rate 3/4 candidates; embedded link control BPTC(128,77) (handlers = 1); the voice bursts the BS handlers
 *           pass on (dmrBSBootstrap / dmrBS): three AMBE 3600x2450 frames each -> frame FEC -> synthesis, one talk path per time slot
 *           (vocoder = 1)
 *   NXDN48: frame gather -> SACCH / FACCH1 K=5 decode + CRC6 / CRC12 + the greedy SACCH retry -> the voice frames the LICHs announce:
 *           AMBE de-interleave -> AMBE 3600x2450 frame FEC -> synthesis (vocoder = 1)
 * One call per batch of samples_per_call samples; carried state streams from call to call.  Bursts / frames that cross a call
 * boundary decode whole: a row holds the last carry_symbols records of the previous call, then this call's, and a sync is decoded
 * in the call that brings the carry_symbols records behind it (ddn_fsk4_chain_flush at the end of a stream decodes the rest). */
typedef struct ddn_fsk4_chain_config {
    int n_channels;
    int samples_per_call;
    int block_len;
    int input_format; /* DDN_IN_CU8 / DDN_IN_CF32 */
    int protocol;     /* DDN_FSK4_DMR / DDN_FSK4_NXDN48 (include/ddn_fsk4.h) */
    int rf_mod;       /* 0 = C4FM rules, 2 = GFSK rules (what dsd-neo runs DMR with) */
    int inverted;     /* DMR: opts->inverted_dmr (handlers need 0) */
    int handlers;     /* 1 = the reference's handlers decide the in-frame lengths (ddn_fsk4_rx_set_handlers) */
    int vocoder;      /* 1 = AMBE synthesis to PCM (NXDN48 voice frames; DMR voice bursts when handlers = 1) */
} ddn_fsk4_chain_config;
typedef struct ddn_fsk4_chain_results { /* device pointers, S = n_channels * max_syncs sync slots */
    size_t stride_symbols, carry_symbols, max_syncs; /* records per row; carried records at its front; sync slots per channel */
    int voice_slots;                /* NXDN48: sync slots per channel the voice stage works on */
    const uint8_t* d_records10;     /* [B][stride][10]: the carried records, then this call's */
    const uint8_t* d_flags;         /* [B][stride] */
    const uint8_t* d_payload2;      /* [B][stride][2] (this call's records only, from index carry_symbols) */
    const int32_t* d_new;           /* [B] new records of this call */
    const int32_t* d_counts;        /* [B] records in the row (carried + new) */
    const int32_t* d_n_sync;        /* [B] syncs decoded in this call */
    const int32_t* d_dropped_syncs; /* [B] running count of accepted syncs that found no decode slot (max_syncs is sized for twice
                                       the densest burst / frame traffic when the handlers run in the loop; 0 in every test) */
    const int32_t* d_sync_pos;      /* [S] row index of the sync's last symbol */
    const uint8_t* d_sync_pat;      /* [S] */
    const uint8_t* d_pre;           /* [S][90] the payload history handed over at each sync */
    const uint8_t* d_valid;         /* [S] frame / burst complete inside the call */
    const uint8_t* d_dmr_slot_type; /* [S][20] after Golay(20,8) */
    const uint8_t* d_dmr_slot_type_ok; /* [S] */
    const uint8_t* d_dmr_pdu96;     /* [S][96] BPTC(196,96) payload bits */
    const uint32_t* d_dmr_bptc_errs; /* [S] */
    const uint8_t* d_nxdn_lich;     /* [S] 7-bit LICH, bit 7 = parity good */
    const uint8_t* d_nxdn_sacch;    /* [S][4] */
    const uint8_t* d_nxdn_sacch_ok; /* [S] CRC6 */
    const uint8_t* d_nxdn_sacch_hard; /* [S][32] the greedy retry's bits */
    const uint8_t* d_nxdn_sacch_hard_ok; /* [S] */
    const uint8_t* d_nxdn_facch;    /* [S][2][12] */
    const uint8_t* d_nxdn_facch_ok; /* [S][2] CRC12 */
    const uint8_t* d_nxdn_voice_skip; /* [B][voice_slots][4] 1 = not a voice frame */
    const uint8_t* d_nxdn_ambe_bits; /* [B][voice_slots * 4][49] */
    const float* d_nxdn_pcm;        /* [B][voice_slots * 4][160] */
    /* DMR voice (protocol DMR, handlers = 1, vocoder = 1; NULL otherwise): the bursts the reference's BS voice handlers hand to the
     * vocoder (dmrBSBootstrap / dmrBS, src/protocol/dmr/dmr_bs.c:585-640,697-760: decided inside the receive loop, event kind 6 with
     * VC >= 1), filed by talk path = 2 * channel + time slot in air order, three AMBE 3600x2450 frames each; a talk path's voice
     * history streams from call to call.  A burst is decoded in the call that holds its last symbol. */
    int dmr_voice_bursts;            /* burst slots per talk path and call */
    const int32_t* d_dmr_n_voice;    /* [2 B] voice bursts of this call */
    const int32_t* d_dmr_voice_start; /* [2 B][dmr_voice_bursts] row index of the burst's first CACH dibit (-1: unused) */
    const int32_t* d_dmr_voice_pre;  /* [2 B][dmr_voice_bursts] sync slot whose 90-dibit hand-over opens the burst (the bootstrap burst), else -1
                                        (>= n_channels * max_syncs: a sync that waits for the next call's decode pass) */
    const uint8_t* d_dmr_voice_skip; /* [2 B][dmr_voice_bursts][3] 0xFF = unused slot */
    const uint8_t* d_dmr_ambe_frames; /* [2 B][dmr_voice_bursts][3][4][24] */
    const uint8_t* d_dmr_ambe_bits;  /* [2 B][dmr_voice_bursts * 3][49] */
    const int32_t* d_dmr_ambe_result; /* [2 B][dmr_voice_bursts * 3][5] */
    const float* d_dmr_pcm;          /* [2 B][dmr_voice_bursts * 3][160] */
    const int32_t* d_events;         /* [B][max_events][4] the handlers' decisions of this call (include/ddn_fsk4.h) */
    const int32_t* d_n_events;       /* [B] */
    int max_events;
    /* DMR data bursts (protocol DMR, handlers = 1; NULL otherwise): the bursts dmr_data_dispatch_burst() hands to
     * dmr_data_burst_handler() (src/protocol/dmr/dmr_data.c:262-280 - found by the sync search or read inside dmrBS(); decided in the
     * receive loop, event kind 6 with VC = 0), in air order per channel, decoded in the call that holds the burst's last symbol.  What
     * the handler computes before it hands over to the protocol layer (src/protocol/dmr/dmr_dburst.c:323-650):
     *   type    the slot type's data type (0 PI, 1 VLC, 2 TLC, 3 CSBK, 4 MBC header, 5 MBC continuation, 6 data header, 7 rate 1/2,
     *           8 rate 3/4, 9 idle, 10 rate 1, 11 USBD; 0xFF: unused entry)
     *   bytes12 the BPTC(196,96) payload; for VLC / TLC after RS(12,9) (ComputeAndCorrectFullLinkControlCrc(), dmr_utils.c:291-351: a
     *           decodable word replaces the received bytes)
     *   crc     bit 0 = crc_correct as the handler computes it while state->data_conf_data = 0 (RS(12,9) for VLC / TLC, CRC-CCITT
     *           with the type's mask, 1 for unconfirmed rate 1/2, 3/4, 1 blocks); bit 1 = the CRC9 of a confirmed rate 1/2 / rate 1
     *           block (what crc_correct is while data_conf_data = 1); bit 2 = RS(12,9) corrected symbols
     *   rate 3/4 (type 8): unconfirmed = the cheaper of the hard and soft Viterbi paths (dmr_dburst_pick_trellis_payload() while
     *           data_conf_data = 0); confirmed = the pick over hard + soft + the list-32 candidates before the DBSN expectation is
     *           applied (cheapest candidate with a good CRC9, else the cheapest; confirmed_crc = a good CRC9 was found); pool = every
     *           candidate in the reference's order {metric, 18 bytes} with {CRC9 ok, DBSN} in the entry's two pad bytes, for the
     *           caller that tracks state->data_dbsn_expected.
     * The protocol layer's running state (confirmed / unconfirmed, DBSN sequence, block assembly: dmr_block.c) is the caller's. */
    int dmr_data_bursts;                 /* entries per channel and call */
    const int32_t* d_dmr_n_data;         /* [B] */
    const int32_t* d_dmr_data_start;     /* [B][dmr_data_bursts] row index of the burst's first CACH dibit (-1: unused) */
    const uint8_t* d_dmr_data_slot;      /* [B][dmr_data_bursts] time slot 0 / 1 */
    const uint8_t* d_dmr_data_type;      /* [..] */
    const uint8_t* d_dmr_data_info196;   /* [..][196] the burst's info bits as received (rate 1 payload) */
    const uint8_t* d_dmr_data_bits96;    /* [..][96] BPTC(196,96) output bits */
    const uint8_t* d_dmr_data_bytes12;   /* [..][12] */
    const uint32_t* d_dmr_data_errs;     /* [..] BPTC irrecoverable errors */
    const uint8_t* d_dmr_data_crc;       /* [..] */
    const uint8_t* d_dmr_r34_unconfirmed; /* [..][18] */
    const uint8_t* d_dmr_r34_confirmed;  /* [..][18] */
    const uint8_t* d_dmr_r34_confirmed_crc; /* [..] */
    const ddn_r34_candidate* d_dmr_r34_pool; /* [..][34] */
    const int32_t* d_dmr_r34_pool_n;     /* [..] */
    /* DMR embedded link control: dmr_data_burst_handler(.., 0xEB, ..) at every voice burst with VC 6 (handle_dmr_bs_slot_vc6_pre_link(),
     * dmr_bs.c:428-446) over the sync fields read under VC 2..5 (the store streams from call to call like
     * state->dmr_embedded_signalling): BPTC(128,77) -> 72 LC bits + 5 checksum bits, ok = ComputeCrc5Bit matches.  Per talk path. */
    int dmr_emb_lcs;                     /* entries per talk path and call */
    const int32_t* d_dmr_n_emb;          /* [2 B] */
    const int32_t* d_dmr_emb_pos;        /* [2 B][dmr_emb_lcs] row index of the VC 6 burst's last symbol (-1: unused) */
    const uint8_t* d_dmr_emb_lc77;       /* [2 B][dmr_emb_lcs][77] */
    const uint32_t* d_dmr_emb_errs;      /* [..] */
    const uint8_t* d_dmr_emb_ok;         /* [..] */
    /* M17 (protocol DDN_FSK4_M17; NULL otherwise), per sync slot [S] of this call's decode list (d_sync_pos / d_sync_pat; pattern index
     * 0 / 1 preamble, 2 / 3 EOT, 4 / 5 LSF, 6 / 7 BERT, 8 / 9 stream, 10 / 11 packet): what ddn_m17_lsf_decode_batch /
     * ddn_m17_str_decode_batch / ddn_m17_lich_assemble_batch give for it (include/ddn_fsk4.h); the LICH assembly buffer streams
     * from call to call like every other carried word */
    const float* d_sync_thr5;            /* [S][5] {center, umid, lmid, max, min} the sync left */
    const uint8_t* d_m17_lsf30;          /* [S][30] */
    const uint8_t* d_m17_lsf_status;     /* [S] 0 not an LSF frame, 1 CRC bad, 2 CRC good */
    const uint32_t* d_m17_lsf_cost;      /* [S] the decoder's path cost */
    const uint8_t* d_m17_lich6;          /* [S][6] */
    const uint8_t* d_m17_lich_cnt;       /* [S] */
    const uint8_t* d_m17_fn_payload18;   /* [S][18] frame number + payload */
    const uint8_t* d_m17_str_status;     /* [S] 0 not a stream frame, 1 LICH failed, 2 decoded */
    const uint8_t* d_m17_lich_lsf30;     /* [S][30] the LSF a chunk counter of 5 completed */
    const uint8_t* d_m17_lich_status;    /* [S] 0 none, 1 CRC bad, 2 CRC good */
    /* YSF (protocol DDN_FSK4_YSF; NULL otherwise): ddn_ysf_fich_decode_batch's outputs per sync slot (include/ddn_fsk4.h) */
    const uint8_t* d_ysf_fich4;          /* [S][4] the 32 FICH bits */
    const uint8_t* d_ysf_fich_status;    /* [S] 0 none, 1 good, 2 Golay failed, 3 CRC failed */
    const uint32_t* d_ysf_fich_cost;     /* [S] the decoder's path cost */
    /* ... and ddn_ysf_payload_decode_batch's (the frame type is carried per channel inside the object) */
    const uint8_t* d_ysf_info2;          /* [S][2] what was decoded (1 V/D1, 2 V/D2, 4 full-rate voice, 8 full-rate data) | FI, DT, flags */
    const uint8_t* d_ysf_dch40;          /* [S][2][20] data-channel bytes (DCH2: 10, DCH: 20; second block: full-rate data frames) */
    const uint8_t* d_ysf_dch_status2;    /* [S][2] 0 none, 1 CRC16 good, 3 CRC16 failed */
    const uint32_t* d_ysf_dch_cost2;     /* [S][2] the decoder's path cost */
    const uint8_t* d_ysf_ambe49x5;       /* [S][5][49] V/D mode 2: ambe_d of the five voice sub-frames */
    const uint8_t* d_ysf_errs2x5;        /* [S][5] their errs2 */
    const uint8_t* d_ysf_frames184x5;    /* [S][5][184] V/D mode 1: four ambe_fr[4][24]; full-rate voice: five (CSD3: two) imbe_fr[8][23] */
    const uint8_t* d_ysf_n_frames;       /* [S] how many of them */
    /* ... voice through the vocoder (vocoder = 1; 0 / NULL otherwise): talk path = channel, the frames of a call in stream order, five
     * positions per frame.  AMBE 3600x2450: V/D mode 2 frames (five sub-frames) and V/D mode 1 frames (the four ysf_ehr() decodes; the
     * fifth position skipped) share one talk-path history; IMBE 7200x4400 (full-rate voice: five frames, two in the CSD3 layout) has its
     * own - a stream that changed codec inside a call would share mbe_parms in the reference, a case these arrays keep apart */
    int ysf_voice_frames;                /* F: frames (of five positions) per channel and call the arrays below hold */
    const int32_t* d_ysf_n_voice;        /* [n_channels] AMBE frames (V/D mode 1 or 2) of this call */
    const int32_t* d_ysf_voice_slot;     /* [n_channels][F] the sync slot each came from */
    const int32_t* d_ysf_voice_result;   /* [n_channels][F * 5][5] mbe_process_result rows after synthesis */
    const float* d_ysf_pcm;              /* [n_channels][F * 5][160] 8 kHz PCM (silence behind the last frame) */
    const uint8_t* d_ysf_voice_skip;     /* [n_channels][F * 5] 1 = no frame at this position */
    const int32_t* d_ysf_imbe_n_voice;   /* the same five arrays for the full-rate (IMBE) frames */
    const int32_t* d_ysf_imbe_voice_slot;
    const uint8_t* d_ysf_imbe_voice_skip;
    const int32_t* d_ysf_imbe_voice_result;
    const float* d_ysf_imbe_pcm;
} ddn_fsk4_chain_results;
typedef struct ddn_fsk4_chain ddn_fsk4_chain;
int ddn_fsk4_chain_create(const ddn_fsk4_chain_config* cfg, ddn_fsk4_chain** out);
void ddn_fsk4_chain_destroy(ddn_fsk4_chain* c);
int ddn_fsk4_chain_run(ddn_fsk4_chain* c, const void* d_iq, void* hip_stream);
int ddn_fsk4_chain_flush(ddn_fsk4_chain* c, void* hip_stream); /* end of a stream: decode the syncs the carry still holds back */
int ddn_fsk4_chain_stage(ddn_fsk4_chain* c, int stage, const void* d_iq, void* hip_stream); /* 0 front end, 1 loop, 2 frame FEC + voice */
int ddn_fsk4_chain_get_results(ddn_fsk4_chain* c, ddn_fsk4_chain_results* out);
void* ddn_fsk4_chain_front_end(ddn_fsk4_chain* c); /* ddn_batch* */
void* ddn_fsk4_chain_rx(ddn_fsk4_chain* c);        /* ddn_fsk4_rx* */

/* ---- a mixed batch (BASELINE configs[3]): P25 Phase 1 + DMR + NXDN48 channel groups of one GPU, every receive loop with the
 * reference's handlers inside it; one stream per group inside the object, the groups' stages lined up (the three front ends, then the
 * three receive loops side by side), the frame FEC / voice stages on a fourth stream behind their loops so that the next call's
 * front ends need not wait for them (the next call's loop does).  _run queues one call of all three, _wait blocks; results of a
 * call are complete - and the d_iq buffers of that call free - after _wait.  When the batch is shared the P25 group runs eight
 * channels per workgroup so that the three loops' wavefronts are resident together. */
typedef struct ddn_mixed_chain_config {
    int n_p25, n_dmr, n_nxdn48; /* channels of each group on this GPU (a group may be empty) */
    int samples_per_call, block_len, input_format, vocoder;
    /* appended in round 6: */
    int overlap; /* 1 = the overlapped schedule: the front ends on streams of their own into two discriminator buffers per group (call
                    k + 1's front ends beside call k's loops), the fsk4 loops one channel per wavefront where a group has <= 1536
                    channels.  Seven streams: it pays only with GPU_MAX_HW_QUEUES >= 6 in the process environment (HIP's default
                    four hardware queues make streams share queues - slower than the default schedule).  Same results. */
} ddn_mixed_chain_config;
typedef struct ddn_mixed_chain ddn_mixed_chain;
int ddn_mixed_chain_create(const ddn_mixed_chain_config* cfg, ddn_mixed_chain** out);
void ddn_mixed_chain_destroy(ddn_mixed_chain* m);
int ddn_mixed_chain_run(ddn_mixed_chain* m, const void* d_iq_p25, const void* d_iq_dmr, const void* d_iq_nxdn48);
int ddn_mixed_chain_wait(ddn_mixed_chain* m);
void* ddn_mixed_chain_part(ddn_mixed_chain* m, int which); /* 0: ddn_p25_chain*, 1: DMR ddn_fsk4_chain*, 2: NXDN48 ddn_fsk4_chain* */
/* how a mixed batch of [P25 | DMR | NXDN48] channels is split over the GPUs of a node: rank r of `world` owns a contiguous block of
 * the global channel index, i.e. first3[k] / count3[k] of group k (no data-path collective: channels are independent streams) */
int ddn_mixed_partition(int n_p25, int n_dmr, int n_nxdn48, int rank, int world, int32_t first3[3], int32_t count3[3]);

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
