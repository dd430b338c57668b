/*
 * include/ddn_mbe.h — C-ABI of the vocoder stage of libdsdneo_hip.so.
 *
 * dsd-neo does not carry the IMBE / AMBE arithmetic itself: src/core/vocoder/dsd_mbe.c calls the un-vendored
 * dependency mbelib-neo 2.0.0 @ 6138cce7091d90e4be9e889ac166006265d3e8fb (vcpkg-ports/mbe-neo/portfile.cmake:5-7),
 * which is NOT in the reference tree.  What this library implements is therefore the PUBLISHED algorithm of the
 * mbelib lineage (mbelib 1.3: imbe7200x4400.c, ambe3600x2450.c, mbelib.c, ecc.c) behind the mbelib-neo 2.x entry
 * points the reference is witnessed to call (CMakeLists.txt:626-657, src/core/vocoder/dsd_mbe.c:75-190,540-598):
 *
 *   frame FEC decode (Golay(23,12) x4 + Hamming(15,11) x3 + PN de-scramble -> 88 bits; Golay x2 + PN -> 49 bits)
 *       integer, bit-exact, PINNED by the four capture-derived vectors the reference holds
 *       (tests/core/test_core_mbe_transform_context.c:134-152: data words and correction counts);
 *   parameter unpack -> spectral amplitude enhancement -> voiced / unvoiced synthesis to f32[160] @ 8 kHz
 *       PARITY UNPINNED: structure per the public algorithm; the codec's quantiser tables (gain levels, bit
 *       allocation, bit order, AMBE codebooks) are DATA of the TIA-102.BABA standard that cannot be reproduced from
 *       the reference tree, so they are a loadable blob (ddn_mbe_tables).  The built-in default blob is SYNTHETIC
 *       (rule-generated, see dsd-neo_amd/csrc/ddn_mbe_tables.c) - it exercises every code path with the right
 *       shapes and rates but does not decode real traffic intelligibly; an integrator fills the blob from the
 *       mbelib-neo sources they already depend on (INTEGRATION.md shows the dump program) and audio then follows the
 *       standard's tables.  Random phases / noise use a counter-based generator instead of libc rand().
 *
 * Two families, as in ddn_hip.h:
 *   (1) ddn_mbe_*      batched, device pointers + stream: n frames of FEC decode per call; S independent talk paths
 *                      x F frames of synthesis per call with the cur / prev / prev_enhanced triple of every talk path
 *                      carried on the device (64 streams x 128 frames).
 *   (2) mbe_*          single-frame drop-ins with mbelib-neo's names, host pointers, caller-owned mbe_parms triple.
 *                      They run the same kernels with n = 1 (parity / link compatibility, not throughput).
 * No CPU fallback: without a gfx950 device the ddn_* calls return DDN_ENODEV and the mbe_* calls return
 * MBE_STATUS_NO_DEVICE (audio buffers are zeroed like mbe_synthesizeSilencef).
 */
#ifndef DDN_MBE_H
#define DDN_MBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mbelib-neo 2.x compatible types ---------------------------------------------------------------------------- */

/* one received bit + reliability (0 = erased .. 255 = certain); fields as used at src/core/vocoder/dsd_mbe.c:80-81 */
typedef struct mbe_soft_bit {
    uint8_t bit;
    uint8_t reliability;
} mbe_soft_bit;

/* decode / process outcome; field names as used at dsd_mbe.c:113-118 and src/core/mbe_result_context.h:33-47 */
typedef struct mbe_process_result {
    unsigned flags;       /* MBE_PROCESS_FLAG_* */
    int c0_errors;        /* data bits corrected in code word c0 (valid with MBE_PROCESS_FLAG_C0_VALID) */
    int c4_errors;        /* IMBE: bits corrected in c4 (valid with MBE_PROCESS_FLAG_C4_VALID) */
    int total_errors;     /* data bits corrected over the whole frame */
    int protected_errors; /* total_errors - c0_errors */
} mbe_process_result;

#define MBE_PROCESS_FLAG_C0_VALID   0x0001u
#define MBE_PROCESS_FLAG_C4_VALID   0x0002u
#define MBE_PROCESS_FLAG_SOFT_INPUT 0x0004u
#define MBE_PROCESS_FLAG_REPEAT     0x0008u /* parameters of the previous frame were reused */
#define MBE_PROCESS_FLAG_MUTE       0x0010u /* silence emitted, decoder history re-initialised */
#define MBE_PROCESS_FLAG_TONE       0x0020u /* AMBE tone frame (b0 126 / 127): muted, no tone synthesis here */
#define MBE_PROCESS_FLAG_ERASURE    0x0040u /* AMBE erasure frame (b0 120..123) */
#define MBE_PROCESS_FLAG_SILENCE    0x0080u /* AMBE silence frame (b0 124 / 125) */

#define MBE_STATUS_OK               0
#define MBE_STATUS_INVALID_ARGUMENT (-1)
#define MBE_STATUS_INVALID_BITS     (-2) /* a bit array held something other than 0 / 1 */
#define MBE_STATUS_NO_DEVICE        (-3)
#define MBE_STATUS_UNSUPPORTED      (-4) /* the rate is not restated here: nothing is decoded, silence is written */

/* decoder history of one talk path: the public mbelib layout (mbelib.h `struct mbe_parameters`); index 0 unused,
 * harmonics 1..L.  dsd-neo treats it as opaque (include/dsd-neo/core/state.h: cur_mp / prev_mp / prev_mp_enhanced). */
typedef struct mbe_parameters {
    float w0;
    int L;
    int K;
    int Vl[57];
    float Ml[57];
    float log2Ml[57];
    float PHIl[57];
    float PSIl[57];
    float gamma;
    int un;
    int repeat;
} mbe_parms;

/* ---- quantiser tables (loadable; the default blob is synthetic, see the header comment) -------------------------- */

typedef struct ddn_mbe_tables {
    uint32_t magic;    /* DDN_MBE_TABLES_MAGIC */
    uint32_t synthetic; /* 1 = rule-generated placeholder, 0 = filled from the standard's tables by the integrator */
    /* IMBE 7200x4400 (L = 9..56 -> row L - 9) */
    float imbe_gain_b2[64];        /* G1 levels indexed by b2 */
    float imbe_gain_step[11];      /* step multiplier by bit count 0..10 for G2..G6 */
    float imbe_gain_sigma[5];      /* std deviation of G2..G6 */
    float imbe_hoc_step[11];       /* step multiplier by bit count 0..10 for C(i,k), k >= 2 */
    float imbe_hoc_sigma[9];       /* std deviation of C(i,k) for k = 2..10 */
    uint8_t imbe_bits[48][58];     /* bit count of field b_m, m = 3..L+1 at [L-9][m] (0..10); other entries 0 */
    uint8_t imbe_bit_order[48][88][2]; /* imbe_d[p] -> {field m (0..L+2), bit weight index (0 = LSB)} */
    /* AMBE 3600x2450 */
    float ambe_f0[120];            /* fundamental in cycles / sample by b0 */
    uint8_t ambe_L[120];           /* harmonics by b0 */
    uint8_t ambe_vuv[32][8];       /* voicing of the 8 bands by b1 */
    float ambe_dg[32];             /* gain delta by b2 */
    float ambe_prba24[512][3];     /* G2..G4 by b3 */
    float ambe_prba58[128][4];     /* G5..G8 by b4 */
    float ambe_hoc5[32][4];        /* C(1,3..6) by b5 */
    float ambe_hoc6[16][4];        /* C(2,3..6) by b6 */
    float ambe_hoc7[16][4];        /* C(3,3..6) by b7 */
    float ambe_hoc8[8][4];         /* C(4,3..6) by b8 */
    uint8_t ambe_blocks[57][4];    /* block lengths J1..J4 by L */
} ddn_mbe_tables;

#define DDN_MBE_TABLES_MAGIC 0x4D424554u /* "MBET" */

/* fills *out with the built-in synthetic blob (host only, no device needed) */
int ddn_mbe_default_tables(ddn_mbe_tables* out);
/* checks shapes / ranges of a blob (bit counts sum to 73 - K per L, bit order is a permutation of the fields' bits,
 * block lengths sum to L ...); 0 = usable */
int ddn_mbe_validate_tables(const ddn_mbe_tables* t);
/* table blob files ("DDNMBET1", blob size, the struct, a checksum; host only): what an integrator who holds the standard's
 * quantiser tables writes once (synthetic = 0) and every run loads - _load_file validates like ddn_mbe_validate_tables */
int ddn_mbe_tables_save_file(const char* path, const ddn_mbe_tables* t);
int ddn_mbe_tables_load_file(const char* path, ddn_mbe_tables* out);

/* ---- (1) batched, device pointers ------------------------------------------------------------------------------ */

enum { DDN_MBE_IMBE_7200X4400 = 0, DDN_MBE_AMBE_3600X2450 = 1 };

/* result rows of the batched calls: int32[5] = {flags, c0_errors, c4_errors, total_errors, protected_errors} */
#define DDN_MBE_RESULT_WORDS 5

/* Frame FEC decode (mbe_decodeImbe7200x4400Frame / mbe_decodeAmbe3600x2450Frame, dsd_mbe.c:152-190).
 *   d_frames  u8 [n][8][23] (IMBE) or [n][4][24] (AMBE), one bit per byte, mbelib row / column order
 *             (= what ddn_p25p1_imbe_deinterleave_* writes)
 *   d_soft    optional u8, same shape: reliabilities (sets MBE_PROCESS_FLAG_SOFT_INPUT; hard decisions are d_frames)
 *   d_bits    u8 [n][88] / [n][49] decoded parameter bits (imbe_d / ambe_d order)
 *   d_result  i32 [n][5] */
int ddn_mbe_frame_decode_batch(int codec, const uint8_t* d_frames, const uint8_t* d_soft, size_t n, uint8_t* d_bits,
                               int32_t* d_result, void* hip_stream);

/* marks result rows whose frame must not reach the decoder (d_skip [n] != 0): ddn_mbe_synth_batch then emits silence for
 * that frame position and leaves the talk path's history untouched - how a batch expresses "no voice frame here"
 * (a slot that is not an LDU, a frame cut off by the end of the call) */
int ddn_mbe_result_skip_batch(const uint8_t* d_skip, size_t n, int32_t* d_result, void* hip_stream);

typedef struct ddn_mbe_batch ddn_mbe_batch;

/* S talk paths; every path's {cur, prev, prev_enhanced} starts as mbe_initMbeParms leaves it */
int ddn_mbe_batch_create(int codec, int n_streams, ddn_mbe_batch** out);
void ddn_mbe_batch_destroy(ddn_mbe_batch* b);
int ddn_mbe_batch_reset(ddn_mbe_batch* b, void* hip_stream);
/* Path s of this batch is path first_stream + s of a larger set split over several batches (devices): its unvoiced-noise sequence is
 * that path's, so a partition of the paths does not show in the PCM.  Resets every path (call before the first frame). */
int ddn_mbe_batch_set_first_stream(ddn_mbe_batch* b, uint32_t first_stream, void* hip_stream);
int ddn_mbe_batch_set_tables(ddn_mbe_batch* b, const ddn_mbe_tables* t);
int ddn_mbe_batch_load_tables_file(ddn_mbe_batch* b, const char* path); /* ddn_mbe_tables_load_file + _set_tables */
/* 1 while the batch synthesizes from the built-in placeholder tables (PCM is then not intelligible speech for real traffic),
 * 0 once a blob with synthetic = 0 has been loaded; -1 for a null batch.  bench.py reports it in config.vocoder_tables. */
int ddn_mbe_batch_tables_synthetic(const ddn_mbe_batch* b);
/* the same for the single-stream mbe_* entry points below (they run on two cached one-path batches, one per codec) */
int ddn_mbe_dropin_set_tables(const ddn_mbe_tables* t);
/* P25 Phase 1 teardown rule of mbe_process_p25p1 (dsd_mbe.c:447-463,540-566): a clear-mode frame that decodes to
 * FC.. with <= 24 set bits and >= 10 corrections is muted without touching the history.  Off by default. */
int ddn_mbe_batch_set_p25p1_tail_rule(ddn_mbe_batch* b, int enable);

/* mbe_processImbe4400Dataf / mbe_processAmbe2450Dataf over [n_streams][n_frames] frames, frame f of a talk path
 * after frame f - 1 (dsd_mbe.c:581, :685).
 *   d_bits       u8 [n_streams][n_frames][88 | 49]
 *   d_result_in  optional i32 [n_streams][n_frames][5] from the frame decode (total_errors drives repeat / mute);
 *                NULL = zero errors
 *   d_pcm        f32 [n_streams][n_frames][160]
 *   d_result_out optional i32 [n_streams][n_frames][5]: input result + REPEAT / MUTE / TONE / ERASURE / SILENCE */
int ddn_mbe_synth_batch(ddn_mbe_batch* b, const uint8_t* d_bits, const int32_t* d_result_in, size_t n_frames,
                        float* d_pcm, int32_t* d_result_out, void* hip_stream);

/* talk-path history <-> host (what the single-frame drop-ins use; also lets a caller migrate a call between batches) */
int ddn_mbe_batch_get_state(ddn_mbe_batch* b, int stream, mbe_parms* cur, mbe_parms* prev, mbe_parms* prev_enhanced);
int ddn_mbe_batch_set_state(ddn_mbe_batch* b, int stream, const mbe_parms* cur, const mbe_parms* prev,
                            const mbe_parms* prev_enhanced);
/* kernel times of the last ddn_mbe_synth_batch in ms: {parameter kernel, synthesis kernel}; enable first */
int ddn_mbe_batch_set_timing(ddn_mbe_batch* b, int enable);
int ddn_mbe_batch_get_timing(ddn_mbe_batch* b, float* ms2);

/* ---- (2) mbelib-neo 2.x names (host pointers, one frame) ------------------------------------------------------ */

void mbe_initMbeParms(mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
void mbe_initProcessResult(mbe_process_result* result);
void mbe_synthesizeSilencef(float* aout_buf);
void mbe_formatProcessResult(char* str, size_t size, const mbe_process_result* result);
int mbe_decodeImbe7200x4400Frame(const char imbe_fr[8][23], char imbe_d[88], mbe_process_result* result);
int mbe_decodeImbe7200x4400SoftFrame(const mbe_soft_bit imbe_fr[8][23], char imbe_d[88], mbe_process_result* result);
int mbe_decodeAmbe3600x2450Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result);
int mbe_decodeAmbe3600x2450SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result);
int mbe_processImbe4400Dataf(float* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
int mbe_processAmbe2450Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
int mbe_processAmbe3600x2450Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24],
                                   char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
int mbe_processAmbe3600x2450SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                       char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                                       mbe_parms* prev_mp_enhanced);
/* The two other symbols dsd-neo's configure step links against (CMakeLists.txt:626-657 probes them at :648 and :654; callers
 * src/core/vocoder/dsd_mbe.c:603 - ProVoice's IMBE 7100x4400 - and :300 - D-STAR AMBE 2400 data files).  They exist so that
 * the reference builds and runs against this library; neither rate is restated: the 7100 -> 7200 parameter-bit re-ordering
 * (mbelib's mbe_convertImbe7100to7200) and the 2400 rate's bit layout and table set are not in the reference tree and a guess
 * would be worse than saying so.  Both report MBE_STATUS_UNSUPPORTED (< 0: the reference clears its error display and plays
 * silence for the frame, dsd_mbe.c:127-148), zero what they would have written and set MBE_PROCESS_FLAG_MUTE. */
int mbe_decodeImbe7100x4400Frame(const char imbe_fr[7][24], char imbe_d[88], mbe_process_result* result);
int mbe_processAmbe2400Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                             mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
/* The three remaining mbelib-neo symbols the reference's sources call outside the configure probe (found by tools/gen_mbe_symbols.py,
 * which lists every mbe_* call in the reference's src/ that the reference does not define itself):
 *   mbe_processAmbe3600x2400Framef  src/core/vocoder/dsd_mbe.c:633 (D-STAR) - refused like the 2400 data call above
 *   mbe_floattoshort                src/core/audio/dsd_audio2.c:1376 - 160 floats -> shorts, gain 7, clip +-32760 (mbelib 1.3 rule)
 *   mbe_versionString               src/runtime/bootstrap/bootstrap.c:695 - the start-up banner */
int mbe_processAmbe3600x2400Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                                   mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced);
void mbe_floattoshort(const float* float_buf, short* aout_buf);
const char* mbe_versionString(void);

#ifdef __cplusplus
}
#endif
#endif /* DDN_MBE_H */
