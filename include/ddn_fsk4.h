/* ddn_fsk4.h - C-ABI of the batched fixed-protocol receive loop for DMR and NXDN48 (BASELINE configs[3]);
 * P25 Phase 1 has its own entry points in ddn_hip.h (ddn_p25_rx_*).  Device pointers + stream, like the rest of the library.
 *
 * Replaces, per channel and per call, the loop a dsd-neo decoder thread runs between rtl_stream_read() and the protocol
 * handler for one enabled protocol: getFrameSync() (src/dsp/dsd_frame_sync.c:3098-3148; DMR matchers :1102-1314 with
 * dmr_resample_on_sync() src/dsp/dmr_sync.c:63-131; NXDN matcher :1507-1556) around getSymbol()
 * (src/dsp/dsd_symbol.c:1854-1880) and get_dibit_and_analog_signal() (src/core/frames/dsd_dibit.c:1045-1076).
 * What is fixed per batch instead of decided at run time: one protocol, the modulation lock (-mc: rf_mod 0, -mg: rf_mod 2),
 * signal polarity (inverted = the reference's -xr for DMR), and how many symbols the handler of a sync class consumes
 * (lock_symbols[]: DMR data 120 = 54 live + 66 skipped, src/protocol/dmr/dmr_data.c:213-302; DMR voice = 54 + 288 per
 * superframe pair the host expects, src/protocol/dmr/dmr_bs.c:840-948; NXDN 182).
 */
#ifndef DDN_FSK4_H
#define DDN_FSK4_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DDN_FSK4_DMR = 1, DDN_FSK4_NXDN48 = 2, DDN_FSK4_NXDN96 = 3, DDN_FSK4_M17 = 4, DDN_FSK4_YSF = 5 };
/* YSF (-fy): the 20-symbol FUSION_SYNC compared exactly in both polarities (frame_sync_try_ysf(), src/dsp/dsd_frame_sync.c:770-797;
 * pattern index 0 = +YSF, 1 = -YSF), 20-symbol warm start, the DMR matched filter, lock_symbols[0] = 460 = the 100 FICH + 360 payload
 * dibits processYSF() reads (every frame type but FI = 3 with DT != 1, which reads the FICH alone); no handler family. */
/* NXDN96: NXDN's rules at 4800 symbols/s (level ring 24, the DMR matched filter: src/dsp/dsd_frame_sync.c:1525-1556, dsd_symbol.c:323-335).
 * M17 (-fz): C4FM lock at 4800 symbols/s, no matched filter (use_matched_filter is ignored), frame_sync_try_m17()'s matcher
 * (src/dsp/dsd_frame_sync.c:865-1100: eight-symbol words with one error allowed, each accepted only after the sync type that may precede
 * it, polarity learnt from the preamble and cleared by EOT / carrier loss), fixed counts behind a sync (dispatch_m17.c:25-68):
 * lock_symbols[0] = 184 for every frame type and the EOT marker, lock_symbols[1] = 8 for the preamble; no handler family.
 * Sync pattern index: 0 / 1 preamble + / -, 2 / 3 EOT, 4 / 5 LSF, 6 / 7 BERT, 8 / 9 stream, 10 / 11 packet. */
enum { DDN_FSK4_CLASS_DATA = 0, DDN_FSK4_CLASS_VOICE = 1 }; /* index into lock_symbols[] */
/* sync pattern index reported in flags bits 3..6 / d_sync_pat.  DMR: 0 BS data word, 1 BS voice word, 2 MS data, 3 MS voice,
 * 4 / 5 direct-mode TS1 / TS2 data, 6 / 7 direct-mode TS1 / TS2 voice (with inverted = 1 the data words mark voice bursts and
 * vice versa, as with -xr).  NXDN48: 0..4 FSW variants positive, 5..9 inverted. */
#define DDN_FSK4_PRE 90 /* payload dibits handed over with every accepted sync */

typedef struct ddn_fsk4_rx_config {
    int n_channels;
    int out_rate_hz;        /* discriminator sample rate (48000) */
    int protocol;           /* DDN_FSK4_* */
    int rf_mod;             /* 0 = C4FM window / slip / clip rules (-mc), 2 = GFSK rules (-mg; what an unlocked dsd-neo
                               switches to on a DMR sync, dsd_frame_sync.c:595-600) */
    int inverted;           /* DMR only: opts->inverted_dmr */
    int use_matched_filter; /* opts->use_cosine_filter (default 1 in the reference) */
    int lock_symbols[4];    /* per sync class; all zero = the defaults above (DMR voice default 54 + 6 * 288) */
} ddn_fsk4_rx_config;
typedef struct ddn_fsk4_rx ddn_fsk4_rx;

int ddn_fsk4_rx_create(const ddn_fsk4_rx_config* cfg, ddn_fsk4_rx** out);
void ddn_fsk4_rx_destroy(ddn_fsk4_rx* b);
int ddn_fsk4_rx_reset(ddn_fsk4_rx* b);
/* enable != 0: the reference's own handlers decide how long a frame is read in frame (plain -fs / -fi, inverted = 0), inside the
 * loop - lock_symbols[] then only serves the sync types without a restated handler (DMR MS / direct mode words):
 *   DMR BS data word   dmr_data_sync() (src/protocol/dmr/dmr_data.c:117-343): TACT Hamming(7,4) on the cached CACH (fails: no live
 *                      dibit), 5 live dibits, slot type Golay(20,8) (fails: stop) + colour-code gate, 49 more; then skipDibit(66)
 *   DMR BS voice word  dmrBSBootstrap() + dmrBS() (src/protocol/dmr/dmr_bs.c:697-948): 54 live dibits, then 144 per burst with the
 *                      TACT / repeated-carrier / sync-word / EMB QR(16,7,6) / colour-code-gate decisions
 *                      (src/protocol/dmr/dmr_confidence.c) - this is the path that prints the reference's "Color Code=02" on its
 *                      dmr_voice / dmr_t3_cc captures (tests/CMakeLists.txt:8925-8930)
 *   NXDN               nxdn_frame() (src/protocol/nxdn/nxdn_frame.c:181-233,592-640): 8 LICH dibits, parity + profile; rejected ->
 *                      the handler returns and lastsynctype is cleared
 * ddn_fsk4_rx_set_events: device buffers for the handlers' decisions of the next runs, d_events i32 [B][max_events][4] =
 * {output index of the deciding symbol, kind, a, b | c << 16}, d_n_events i32 [B]:
 *   kind 4 NXDN LICH        a = accepted, b = LICH (7 bits), c = parity ok
 *   kind 5 DMR data burst   a = slot type ok, b = colour code (-1 Golay failed, -2 TACT failed), c = data type | reject << 8 | pending << 9
 *   kind 6 "Color Code="    a = the value the reference prints (16 = XX), b = VC (0: data burst), c = slot
 *   kind 7 DMR voice burst  a = slot, b = EMB colour code (25: none), c = voice sync word | action << 4 (1 on, 2 end) | the VC the
 *                           burst's sync segment was read under << 8 (2..6: filed as that burst's embedded signalling)
 *   kind 8 DMR voice end    a = 1 bootstrap / 0 loop, b = TACT ok, c = EMB / sync ok */
int ddn_fsk4_rx_set_handlers(ddn_fsk4_rx* b, int enable);
int ddn_fsk4_rx_set_events(ddn_fsk4_rx* b, int32_t* d_events, int32_t* d_n_events, size_t max_events);
/* host convenience: event buffers owned by the batch object (arm once), copied to host arrays [B][max_events][4] / [B] after a run */
int ddn_fsk4_rx_events_host_arm(ddn_fsk4_rx* b, size_t max_events);
int ddn_fsk4_rx_events_host_read(ddn_fsk4_rx* b, int32_t* events, int32_t* n_events);
size_t ddn_fsk4_rx_max_symbols(const ddn_fsk4_rx* b, size_t n);
size_t ddn_fsk4_rx_max_syncs(const ddn_fsk4_rx* b, size_t n);
/* per-channel handler lengths, host array int32 [n_channels][4] */
int ddn_fsk4_rx_set_lock_symbols(ddn_fsk4_rx* b, const int32_t* lock4);
/* d_disc f32 [B][n] -> d_records10 u8 [B][max_symbols][10] (the reference's symbol-capture record), d_flags u8 [B][max_symbols]
 * (1 in frame, 2 sync accepted on this symbol, 4 negative polarity, pattern index << 3 on the accepting symbol), d_payload2
 * u8 [B][max_symbols][2] = {payload dibit, reliability} (dmr_payload_buf / dmr_soft_buf contents), d_counts i32 [B];
 * per accepted sync k < d_n_sync[c] (capacity max_syncs per channel): d_sync_pos i32 = index of the sync's last symbol,
 * d_sync_pat u8, d_pre / d_pre_rel u8 [DDN_FSK4_PRE] = the payload dibits / reliabilities ending at that symbol, after DMR's
 * re-digitisation.  State carries across calls. */
int ddn_fsk4_rx_run(ddn_fsk4_rx* b, const float* d_disc, size_t n, uint8_t* d_records10, uint8_t* d_flags,
                    uint8_t* d_payload2, int32_t* d_counts, size_t max_symbols, int32_t* d_sync_pos, uint8_t* d_sync_pat,
                    uint8_t* d_pre, uint8_t* d_pre_rel, int32_t* d_n_sync, size_t max_syncs, void* hip_stream);
int ddn_fsk4_rx_run_host(ddn_fsk4_rx* b, const float* disc, size_t n, uint8_t* records10, uint8_t* flags, uint8_t* payload2,
                         int32_t* counts, size_t max_symbols, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre,
                         uint8_t* pre_rel, int32_t* n_sync, size_t max_syncs);
int ddn_fsk4_rx_get_thresholds(ddn_fsk4_rx* b, int channel, float out7[7]); /* center umid lmid max min maxref minref */
/* kernel shape: channels per recurrence wavefront (1, 2, 4 ... 32; results do not depend on it).  The default is the fewest that
 * keeps every workgroup resident for this batch alone; a host that runs other loops beside this one on the same GPU (the mixed
 * chain does) picks it for the whole device's channel count. */
int ddn_fsk4_rx_set_channels_per_wave(ddn_fsk4_rx* b, int channels_per_wave);
/* A/B selectors of the loop kernel's schedule (which of two equivalent code paths a launch takes; the results never depend on them):
 * 65536 = the bulk hunting passes one owner lane at a time.  For the parity tests and timing tools; 0 = the product's choice. */
int ddn_fsk4_rx_set_debug_flags(ddn_fsk4_rx* b, int flags);
/* optional output beside d_sync_pos / d_sync_pat: the slicer thresholds {center, umid, lmid, max, min} as every accepted sync leaves them
 * (after the warm start), [B][max_syncs][5] floats in device memory, NULL = off.  Thresholds are static inside a DMR / NXDN / M17 frame,
 * so these are what a soft-symbol frame decoder reads (M17 LSF: soft_symbol_to_viterbi_cost(), src/core/frames/dsd_dibit.c:1189-1242) */
int ddn_fsk4_rx_set_sync_thresholds(ddn_fsk4_rx* b, float* d_thr5);

/* ---- M17 link setup frames behind the loop (the consumer of the libM17-style K = 5 decoder, ddn_fec_viterbi_k5_*) ---------------------
 * == processM17LSF() (src/protocol/m17/m17.c:1395-1408) for every accepted LSF sync (pattern 4 / 5) of a DDN_FSK4_M17 loop call whose 184
 * payload symbols lie inside the call's records: soft symbols -> soft_symbol_to_viterbi_cost() per bit against the thresholds the sync left
 * (d_sync_thr5: ddn_fsk4_rx_set_sync_thresholds; src/core/frames/dsd_dibit.c:1189-1242 - its expf evaluated as the host's libm does) ->
 * de-randomise -> de-interleave -> de-puncture P1 -> viterbi_decode(488 costs) -> the 30 LSF bytes (DST 48, SRC 48, TYPE 16, META 112,
 * CRC 16: m17_parse_lsf(), src/protocol/m17/m17_parse.c:369-420) + CRC16 (m17_crc16, m17_algorithms.c:19-35).
 * d_records10 [B][stride_symbols][10], d_counts [B] = records of the call per channel, d_sync_pos / d_sync_pat [B][max_syncs], d_n_sync [B]:
 * the loop's outputs.  Out, per sync slot [B][max_syncs]: d_lsf30 [..][30], d_status (0 = not an LSF sync or its frame is not complete in
 * this call, 1 = decoded with a bad CRC, 2 = CRC good), d_path_cost (optional) = the decoder's path cost. */
int ddn_m17_lsf_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                             const uint8_t* d_sync_pat, const int32_t* d_n_sync, const float* d_sync_thr5, int n_channels, size_t max_syncs,
                             uint8_t* d_lsf30, uint8_t* d_status, uint32_t* d_path_cost, void* hip_stream);
/* == processM17STR() (src/protocol/m17/m17.c:1122-1176) for every accepted stream sync (pattern 8 / 9) whose frame is complete: hard dibits ->
 * de-randomise -> de-interleave -> LICH = four Golay(24,12) words (m17_lich_decode_bits) -> 48 content bits (d_lich6, packed) with the
 * 3-bit chunk counter (d_lich_cnt); when all four decode and the counter is < 6: P2 de-puncture -> CNXDNConvolution (148 steps) ->
 * d_fn_payload18 = frame number (2 bytes, big endian) + the 16 payload bytes.  d_status: 0 = not a stream sync / frame not complete,
 * 1 = LICH failed (no payload, as the reference), 2 = decoded. */
int ddn_m17_str_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                             const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_lich6,
                             uint8_t* d_lich_cnt, uint8_t* d_fn_payload18, uint8_t* d_status, void* hip_stream);
/* ---- YSF frame information channel behind the loop (the K = 5 decoder's second consumer) --------------------------------------------
 * == ysf_conv_fich() (src/protocol/ysf/ysf.c:357-424) for every accepted sync of a DDN_FSK4_YSF loop call whose 100 FICH dibits lie inside
 * the records: dibit de-interleave (20 x 5) -> dsd_ysf_soft_viterbi_decode (hard costs through viterbi_decode_punctured, ysf_frame.c:82-126)
 * -> four Golay(24,12) words -> CRC16.  d_fich4 [B][max_syncs][4] = the 32 information bits packed (FI 2, CS 2, CM 2, BN 2, BT 2, FN 3,
 * FT 3, reserved 2, MR 3, VoIP path 1, DT 2, SQL type 1, SQL code 7: ysf_parse_fich() :534-546); d_status 0 = no complete FICH behind this
 * sync in this call, 1 = good, 2 = a Golay word failed (err -1), 3 = CRC failed (err -2); d_v_error (optional) the decoder's path cost. */
int ddn_ysf_fich_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                              const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_fich4, uint8_t* d_status,
                              uint32_t* d_v_error, void* hip_stream);

/* The payload behind the FICH == ysf_dispatch_payload() (src/protocol/ysf/ysf.c:908-922) for every frame of the call (a sync of the
 * decode list with d_fich_status != 0), in stream order per channel:
 *   the type a frame is read as: FI / DT of its FICH, or - when the FICH failed - of the last good frame of the channel (ysf_parse_fich's
 *     statics, :553-555): d_last_dt_fi u8 [n_channels][2] {DT, FI}, zero before the first call, carried by the caller from call to call;
 *   d_info2 [S][2]: [0] = what was decoded, a bit mask: 1 V/D mode 1 (FI 1, DT 0), 2 V/D mode 2 (FI 1, DT 2), 4 full-rate voice
 *     (FI 1, DT 3), 8 full-rate data (DT 1 or FI 0 / 2) - 0 when the frame's 360 payload dibits are not inside the records;
 *     [1] = FI | DT << 2 | 16 (FICH failed: type taken over) | 32 (a frame) | 64 (more frames than a row holds 480 symbols apart:
 *     left undecoded) | 128 (full-rate voice in the CSD3 layout);
 *   V/D mode 2: the five voice sub-frames, ysf_read_type2_vech_bits + ysf_build_type2_ambe (:687-722): d_ambe49x5 [S][5][49] = ambe_d
 *     (what mbe_processAmbe2450Dataf takes), d_errs2x5 [S][5] = errs2; the data channel, ysf_conv_dch2 (:245-300): d_dch40 [S][2][20]
 *     block 0 bytes 0 .. 9, d_dch_status2 [S][2] (0 none, 1 CRC16 good, 3 CRC16 failed), d_dch_cost2 [S][2] = the decoder's path cost;
 *   V/D mode 1: the data channel, ysf_conv_dch (:302-355), block 0 bytes 0 .. 19; its voice blocks as ysf_ehr() (:425-476) hands them
 *     to processMbeFrame: d_frames184x5 [S][5][184], the first four frames, ambe_fr[4][24] as row * 24 + column (d_n_frames [S] = 4);
 *   full-rate voice (FI 1, DT 3): five IMBE 7200x4400 frames as dsd_ysf_unpack_full_rate_imbe (ysf_frame.c:138-163) builds them,
 *     imbe_fr[8][23] as row * 23 + column (d_n_frames = 5); with FT = 1 and FN = 0 (CSD3, d_info2[1] bit 128) two frames behind a data
 *     block that goes through ysf_conv_dch into block 0 (ysf_handle_full_rate_voice :824-842);
 *   full-rate data (a frame that is nothing else): both blocks, in turn (ysf_handle_full_rate_data :844-864).
 * S = n_channels x max_syncs, slots as d_sync_pos.  The frames of V/D mode 1 and of full-rate voice are handed back as frames here; the
 * chain object (include/ddn_chain.h, vocoder = 1) takes them through ddn_mbe_frame_decode_batch and ddn_mbe_synth_batch. */
int ddn_ysf_payload_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                                 const int32_t* d_n_sync, int n_channels, size_t max_syncs, const uint8_t* d_fich4,
                                 const uint8_t* d_fich_status, uint8_t* d_last_dt_fi, uint8_t* d_info2, uint8_t* d_dch40,
                                 uint8_t* d_dch_status2, uint32_t* d_dch_cost2, uint8_t* d_ambe49x5, uint8_t* d_errs2x5,
                                 uint8_t* d_frames184x5, uint8_t* d_n_frames, void* hip_stream);
/* The LSF reassembled from the LICH chunks, in the order of the syncs of a call: a decoded LSF frame seeds the buffer (m17_decode_lsf_soft_
 * bits), an EOT marker clears it (dispatch_m17.c:39), chunk c fills bytes 5 c .. 5 c + 4, chunk 5 closes it: d_lich_lsf30 [B][max_syncs][30]
 * + d_lich_status (0 none here, 1 CRC bad, 2 CRC good: M17finalizeLICH), then the buffer is cleared.  d_assembly32 [B][32] is the carried
 * buffer (zero it before a stream's first call).  d_lsf30 / d_lsf_status = ddn_m17_lsf_decode_batch's outputs, or both NULL. */
int ddn_m17_lich_assemble_batch(const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs, const uint8_t* d_lsf30,
                                const uint8_t* d_lsf_status, const uint8_t* d_lich6, const uint8_t* d_lich_cnt, const uint8_t* d_str_status,
                                uint8_t* d_assembly32, uint8_t* d_lich_lsf30, uint8_t* d_lich_status, void* hip_stream);
int ddn_fsk4_rx_set_timing(ddn_fsk4_rx* b, int enable);
int ddn_fsk4_rx_get_timing(ddn_fsk4_rx* b, float* ms2); /* {matched filter, receive loop} of the last run */

/* DMR burst fields from the per-sync hand-over + the live symbols after the sync (SURVEY 8f rank 3 seam: what
 * dmr_data_sync() / dmrBSBootstrap() assemble, src/protocol/dmr/dmr_data.c:117-262, dmr_bs.c:700-760).  For sync k of channel
 * c (slot c * max_syncs + k): d_slot_type u8 [20] bits (Golay(20,8) input order), d_info u8 [196] bits (BPTC input order,
 * first half from d_pre, second half live = the records' dibits, i.e. getDibitSoft()'s polarity-corrected return values),
 * d_cach u8 [24] bits de-interleaved (TACT first), d_valid = 1 when the 54 live dibits lie inside this call's records.
 * inverted != 0 applies the reference's dibit ^= 2 to the cached half (dmr_data.c:84-86). */
int ddn_dmr_burst_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                         const uint8_t* d_pre, const int32_t* d_n_sync, int n_channels, size_t max_syncs, int inverted,
                         uint8_t* d_slot_type, uint8_t* d_info, uint8_t* d_cach, uint8_t* d_valid, void* hip_stream);

/* NXDN frame fields from the records after each accepted frame sync (what nxdn_frame() assembles before its decoders,
 * src/protocol/nxdn/nxdn_frame.c:181-199,311-331,596-621 + nxdn_descramble.c + nxdn_deperm.c:123-172): per sync slot
 * (c * max_syncs + k) d_lich u8 = the 7-bit LICH, bit 7 set when its parity checks; d_sacch_sym / _rel u8 [36][2] and
 * d_facch_sym / _rel u8 [2][96][2] = de-scrambled, de-interleaved, de-punctured symbol / reliability pairs in the layout
 * ddn_fec_nxdn_conv_batch takes (36 steps -> 32 bits, 96 steps -> 92 bits; pass d_rel = NULL there for the hard-decision
 * retry the reference falls back to when the soft decode fails its CRC, nxdn_deperm.c:1128-1135,1195-1200);
 * d_valid = 1 when the 182 dibits lie inside this call's records. */
int ddn_nxdn_frame_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                          const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_lich, uint8_t* d_sacch_sym,
                          uint8_t* d_sacch_rel, uint8_t* d_facch_sym, uint8_t* d_facch_rel, uint8_t* d_valid, void* hip_stream);
/* AMBE 3600x2450 voice frames ("AMBE interleave"): the 36-dibit interleave schedule the reference's DMR,
 * NXDN and YSF voice paths share (include/dsd-neo/core/ambe_interleave.h:25-38; users dmr_bs.c:137-160, dmr_ms.c:73-77,
 * nxdn_voice.c:57-74).  Outputs feed ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, frames, reliabilities, ...):
 *   d_ambe_fr  u8 [..][4][24] one bit per byte (char ambe_fr[4][24]; the cells the schedule never writes are 0)
 *   d_ambe_rel u8 [..][4][24] per-bit reliability = its dibit's (dsd_vocoder_soft_bit.reliability), or NULL
 * ddn_ambe2450_deinterleave_batch: n frames of 36 dibits (+ optional reliabilities) that the caller has already cut out.
 * ddn_nxdn_voice_gather: for sync k of channel c (slot c * max_syncs + k, as ddn_nxdn_frame_gather) the four voice frames
 *   behind LICH + SACCH, de-scrambled (nxdn_frame.c:181-199, nxdn_voice.c:57-66): [slots][4][4][24]; d_valid (or NULL) = 1
 *   when the frame's 182 dibits lie inside this call's records.  Which of the four carry voice is the LICH's business.
 * ddn_dmr_voice_burst_gather: d_burst_start i32 [n_channels][max_bursts] = record index of a voice burst's first CACH dibit
 *   (< 0: unused slot) -> the burst's three voice frames [slots][3][4][24] (frame 2 straddles the 24 sync / EMB dibits),
 *   d_sync48 u8 [slots][48] sync / EMB bits, d_cach24 u8 [slots][24] de-interleaved CACH bits (TACT first) - either may be
 *   NULL; inverted != 0 applies the MS path's dibit ^= 2 (dmr_ms.c:55-64). */
int ddn_ambe2450_deinterleave_batch(const uint8_t* d_dibits36, const uint8_t* d_reliab36, size_t n, uint8_t* d_ambe_fr,
                                    uint8_t* d_ambe_rel, void* hip_stream);
int ddn_nxdn_voice_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols, const int32_t* d_sync_pos,
                          const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_ambe_fr, uint8_t* d_ambe_rel,
                          uint8_t* d_valid, void* hip_stream);
int ddn_dmr_voice_burst_gather(const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                               const int32_t* d_burst_start, int n_channels, size_t max_bursts, int inverted,
                               uint8_t* d_ambe_fr, uint8_t* d_ambe_rel, uint8_t* d_sync48, uint8_t* d_cach24, uint8_t* d_valid,
                               void* hip_stream);

/* CRC of decoded NXDN fields, rows = ddn_fec_nxdn_conv_batch output: kind 0 = SACCH (26 bits + CRC6, nxdn_deperm.c:1246-1261),
 * kind 1 = FACCH1 (80 bits + CRC12, nxdn_dcr_utils.c:21-42); kind + 2 = the same on rows of one bit per byte (what
 * ddn_fec_trellis_decode_batch writes); d_ok [n] = 1 when the field's CRC matches */
int ddn_nxdn_crc_check_batch(const uint8_t* d_bytes, int stride, size_t n, int kind, uint8_t* d_ok, void* hip_stream);
#ifdef __cplusplus
}
#endif
#endif
