/*
 * oracle/ddn_oracle.h — CPU restatement of the dsd-neo sample-streaming hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or called by the product
 * (dsd-neo_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).  The
 * restatement is pinned against the reference's own compiled sources (oracle/_ref/libdsdneo_ref.so, built
 * by oracle/Makefile from /root/reference in place) by tests/test_oracle_vs_ref.py, and against the
 * committed golden vectors under tests/golden/ (generated from that same library by
 * tests/golden/make_golden.py) everywhere else.
 */
#ifndef DDN_ORACLE_H
#define DDN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- front end: widen -> channel LPF -> power/squelch -> FSK discriminator ------------------------- */

#define ORC_MAX_TAPS 144 /* kChannelLpfTaps, src/dsp/demod_pipeline.cpp:129 */

/* DSD_CH_LPF_PROFILE_* ids, include/dsd-neo/dsp/demod_state.h:36-43 */
enum {
    ORC_LPF_WIDE = 0,
    ORC_LPF_6K25 = 1,
    ORC_LPF_12K5 = 2,
    ORC_LPF_PROVOICE = 3,
    ORC_LPF_P25_C4FM = 4,
    ORC_LPF_P25_CQPSK = 5,
};

void orc_widen_u8(const uint8_t* src, float* dst, size_t len);
int orc_channel_lpf_design(int rate_hz, int profile, float* taps_out, int max_taps);
int orc_firdes_low_pass_blackman(double gain, double fs, double cutoff, double transition, float* taps_out,
                                 int max_taps);

/* fma_order=1: the reference's AVX2 unit (vector FMA chain); 0: the scalar/SSE2 unit (mul, then add). */
void orc_fir_complex_apply(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                           int taps_len, int fma_order);
int orc_hb_decim2_complex(const float* in, int in_len, float* out, float* hist_i, float* hist_q, const float* taps,
                          int taps_len, int fma_order);
extern const float orc_hb15_taps[15];
extern const float orc_hb31_taps[31];

float orc_mean_power(const float* samples, int len, int step);

typedef struct orc_fsk_state {
    float prev_i, prev_q;
    int have_prev;
    float dc_est;
    float peak_est;
} orc_fsk_state;

void orc_fsk_reset(orc_fsk_state* st);
float orc_fsk_phase_delta(float cur_i, float cur_q, float prev_i, float prev_q);
int orc_fsk_discriminator(orc_fsk_state* st, const float* iq, int len_interleaved, float* out, int max_out);

/* one channel's carried front-end state (what struct demod_state carries for this path) */
typedef struct orc_front_end {
    int rate_hz;
    int lpf_enable;
    int taps_len;
    int downsample_passes;
    int fma_order;
    float squelch_level;
    float taps[ORC_MAX_TAPS];
    float hist_i[ORC_MAX_TAPS];
    float hist_q[ORC_MAX_TAPS];
    float hb_hist_i[10][30];
    float hb_hist_q[10][30];
    orc_fsk_state fsk;
    float channel_pwr;
    int channel_squelched;
    /* optional IQ conditioning between the channel LPF and the discriminator (default off) */
    int iq_dc_enable, iq_dc_shift, iqbal_enable;
    float iq_dc_r, iq_dc_i, iqbal_thr, iqbal_ema_a, iqbal_er, iqbal_ei;
} orc_front_end;
void orc_fe_set_iq_options(orc_front_end* fe, int dc_enable, int dc_shift, int bal_enable, float bal_thr,
                           float bal_ema_a);

void orc_fe_init(orc_front_end* fe, int rate_hz, int profile, int lpf_enable, float squelch_level,
                 int downsample_passes, int fma_order);
/* one full_demod() block: in = n_complex interleaved floats; returns discriminator samples written */
int orc_fe_block_f32(orc_front_end* fe, const float* iq, int n_complex, float* out, float* scratch /* >=4*n */);
long orc_fe_run_cu8(orc_front_end* fe, const uint8_t* iq, long n_complex, int block_len, float* out);
long orc_fe_run_f32(orc_front_end* fe, const float* iq, long n_complex, int block_len, float* out);
/* B independent channels, channel-major input/output (the batched shape the HIP path uses) */
void orc_fe_run_batch_cu8(int n_channels, const uint8_t* iq, long n_complex, int block_len, int rate_hz, int profile,
                          float squelch_level, float* out);

/* ---- trellis / Viterbi decoders (oracle/ddn_oracle_fec.c) -------------------------------------------- */
void orc_trellis_interleave_98(uint8_t tbl[98]);
const uint8_t* orc_tbl_p25_half_rate_nibble(void);
const uint8_t* orc_tbl_r34_point_to_nibble(void);
const uint8_t* orc_tbl_r34_nibble_to_point(void);
const uint8_t* orc_tbl_r34_fsm(void);
int orc_p25_12_soft_llr(const int16_t* llr196, uint8_t out12[12]);
int orc_p25_12_soft_llr_list(const int16_t* llr196, uint8_t* out_bytes, uint32_t* out_metric, int max);
int orc_r34_decode(const uint8_t* dibits98, const uint8_t* reliab98 /* NULL = hard */, uint8_t out18[18]);
/* P25 Phase 1 confirmed-data rate 3/4 blocks on LLR pairs (p25p1_mbf34.c): list of <= 8 {18 bytes, metric}; plain best path */
int orc_p25_mbf34_list(const int16_t* llr196, int max, uint8_t* out_bytes, uint32_t* out_metric);
int orc_p25_mbf34_best(const int16_t* llr196, uint8_t out18[18]);
int orc_r34_decode_list(const uint8_t* dibits98, const uint8_t* reliab98, int max, int32_t* out_metric, uint8_t* out_bytes);
void orc_nxdn_conv_decode(const uint8_t* sym, const uint8_t* rel /* NULL = hard */, int n_steps,
                          uint16_t metrics16[16], uint8_t* out, int n_bits);
uint32_t orc_m17_viterbi_decode(uint8_t* out, const uint16_t* in, int len);
uint32_t orc_m17_viterbi_decode_punctured(uint8_t* out, const uint16_t* in, const uint8_t* punct, int in_len,
                                          int p_len);

/* ---- Gardner timing recovery (oracle/ddn_oracle_ted.c) ------------------------------------------------ */
#define ORC_TED_DL 100 /* TED_DL_SIZE, include/dsd-neo/dsp/ted.h:19 */
typedef struct orc_ted_state {
    float mu, omega, omega_mid, omega_min, omega_max, omega_rel;
    float last_r, last_j;
    float lock_accum;
    int lock_count;
    float dl[ORC_TED_DL * 2 * 2];
    int dl_index, twice_sps, sps;
} orc_ted_state;
const float* orc_mmse_table(void);
void orc_mmse_interp_complex(const float* samples, float mu, float* re, float* im);
void orc_ted_init(orc_ted_state* t);
int orc_gardner_block(orc_ted_state* t, int sps, float ted_gain, int symbol_rate_hz, const float* iq, int n,
                      float* out);

/* ---- CQPSK chain after the channel LPF (oracle/ddn_oracle_cqpsk.c) ----------------------------------------- */
#define ORC_FLL_MAX_TAPS 48
typedef struct orc_fll {
    float phase, freq, alpha, beta, max_freq, min_freq;
    float tlr[ORC_FLL_MAX_TAPS], tli[ORC_FLL_MAX_TAPS], tur[ORC_FLL_MAX_TAPS], tui[ORC_FLL_MAX_TAPS];
    int n_taps;
    float dr[2 * ORC_FLL_MAX_TAPS], di[2 * ORC_FLL_MAX_TAPS];
    int delay_idx, sps, initialized;
} orc_fll;
typedef struct orc_costas {
    float phase, freq, alpha, beta, error, error_smooth;
    int initialized;
} orc_costas;
typedef struct orc_cqpsk {
    int sps, sym_rate;
    float ted_gain, agc_avg, diff_prev_r, diff_prev_j;
    orc_fll fll;
    orc_ted_state ted;
    orc_costas cos;
} orc_cqpsk;
void orc_cqpsk_rms_agc(float* avg_io, float* iq, int pairs);
void orc_fll_design(orc_fll* f, int sps);
void orc_fll_block(orc_fll* f, int sps, float* iq, int pairs);
void orc_diff_phasor(float* prev_r, float* prev_j, float* iq, int pairs);
void orc_costas_block(orc_costas* c, float* iq, int pairs);
float orc_atan2_qpsk(float y, float x);
void orc_cqpsk_init(orc_cqpsk* c, int sps, int symbol_rate_hz, float ted_gain);
int orc_cqpsk_block(orc_cqpsk* c, float* iq, int n, float* out, float* work);
size_t orc_cqpsk_sizeof(void);
typedef struct orc_cqpsk_fe {
    int taps_len;
    float taps[ORC_MAX_TAPS], hist_i[ORC_MAX_TAPS], hist_q[ORC_MAX_TAPS];
    orc_cqpsk chain;
} orc_cqpsk_fe;
void orc_cqpsk_fe_init(orc_cqpsk_fe* fe, int rate_hz, int symbol_rate_hz, int profile, int lpf_enable, float ted_gain);
long orc_cqpsk_fe_run_f32(orc_cqpsk_fe* fe, const float* iq, long n_complex, int block_len, float* out, float* scratch);
size_t orc_cqpsk_fe_sizeof(void);
void orc_cqpsk_fe_get_state(const orc_cqpsk_fe* fe, float out8[8]);

/* ---- P25p1 C4FM slicer / soft decisions / matched filter (oracle/ddn_oracle_sym.c) ------------------- */
#define ORC_SLICER_SSIZE 128  /* opts->ssize, src/core/util/dsd_init.c:169 */
#define ORC_SLICER_MSIZE 1024 /* opts->msize, src/core/util/dsd_init.c:170 */
typedef struct orc_slicer {
    int negative;
    float center, umid, lmid, max, min, maxref, minref;
    float sbuf[ORC_SLICER_SSIZE];
    int sidx;
    float minbuf[ORC_SLICER_MSIZE], maxbuf[ORC_SLICER_MSIZE];
    int midx, sums_valid;
    double min_sum, max_sum;
} orc_slicer;
void orc_slicer_init(orc_slicer* s, int negative_polarity);
void orc_slicer_step(orc_slicer* s, float sym, int rec4[4]);
void orc_slicer_run(orc_slicer* s, const float* sym, long n, int* rec4, float* thr5);
void orc_slicer_digitize(const orc_slicer* s, float sym, int rec4[4]);
void orc_slicer_step_static(orc_slicer* s, float sym, int rec4[4]);
int orc_slicer_reliability(const orc_slicer* s, float sym);
void orc_p25_filter_run(float hist[90], const float* in, long n, float* out);

/* ---- protocol handlers as per-symbol state machines: how many symbols a frame reads in frame (ddn_oracle_handlers.c) -- */
#define ORC_HEV_MAX 4096
enum {
    ORC_HEV_P25_NID = 1,     /* a = status, b = nac, c = duid (0xFF invalid) */
    ORC_HEV_P25_TSBK = 2,    /* a = block, b = crc ok | byte 1 << 8, c = last_block << 8 | selected candidate */
    ORC_HEV_P25_MPDU = 3,    /* a = header crc ok, b = blocks to read, c = byte 0 */
    ORC_HEV_NXDN_LICH = 4,   /* a = accepted, b = lich, c = parity ok */
    ORC_HEV_DMR_DATA = 5,    /* a = slot type ok, b = colour code (-1 Golay failed, -2 TACT failed), c = burst | reject << 8 | pending << 9 */
    ORC_HEV_DMR_CC_PRINT = 6, /* the reference prints "Color Code=%02d": a = value (16 = XX), b = VC (0 data burst), c = slot */
    ORC_HEV_DMR_VOICE_BURST = 7, /* a = slot, b = EMB colour code (25 none), c = voice sync | action << 4 */
    ORC_HEV_DMR_VOICE_END = 8    /* a = 1 bootstrap / 0 loop, b = tact ok, c = emb / sync ok */
};
typedef struct orc_hevent {
    int32_t pos;
    int16_t kind, a, b, c;
    /* what the decision decoded: P25 NID = p25p1_nid_decode's {status, nac, duid, error count}; P25 TSBK / PDU header block = its 12
     * bytes (little-endian words) + {CRC16 good | candidate index << 8 | block << 16}; zero for the other kinds */
    int32_t data[4];
} orc_hevent;
typedef struct orc_hevents {
    int n; /* events pushed (may exceed ORC_HEV_MAX; only the first ORC_HEV_MAX are kept) */
    orc_hevent ev[ORC_HEV_MAX];
} orc_hevents;

typedef struct orc_p25h {
    int phase, idx, left, duid, block, end, skipdibit, k;
    int nac, p2_cc, threshold, parity, parity_rel;
    uint8_t bch[63], bch_rel[63];
    uint8_t dib[98];
    int16_t llr[196];
} orc_p25h;
void orc_p25h_init(orc_p25h* h, int erasure_threshold);
void orc_p25h_no_carrier(orc_p25h* h);
int orc_p25h_begin(orc_p25h* h);
int orc_p25h_symbol(orc_p25h* h, long pos, int d, int l0, int l1, orc_hevents* ev);
/* p25_mpdu_finalize_header (p25p1_mdpu.c:336-410): repetitions -> header; 0/1/2 repetition, 32 combined LLRs, 64 majority (| 16 CRC bad) */
int orc_p25_mpdu_finalize_header(const uint8_t* rep_bytes, const int16_t* rep_llr, int hdr_reps, uint8_t out12[12]);

typedef struct orc_nxdnh {
    int idx, lich;
} orc_nxdnh;
void orc_nxdnh_init(orc_nxdnh* h);
int orc_nxdnh_begin(orc_nxdnh* h);
int orc_nxdnh_symbol(orc_nxdnh* h, long pos, int d, int* bad_sync, orc_hevents* ev);

enum { ORC_DMR_BS_DATA = 0, ORC_DMR_BS_VOICE = 1 };
typedef struct orc_dmrh {
    int phase, idx, left, stereo;
    uint8_t pay[144], rel[144];
    /* dmr_confidence.c */
    int locked, conf_cc, cand_cc, cand_count, mismatch;
    uint8_t vsync_seen[2], vopen[2], vcount[2];
    int dmr_color_code, color_code, currentslot;
    int conf_reject, conf_pending;
    /* dmr_bs_ctx */
    int vc1, vc2, skipcount, tact_okay, emb_ok, internalslot, boot_slot;
    uint8_t emb_err[2];
    uint8_t red_b[36];
} orc_dmrh;
void orc_dmrh_init(orc_dmrh* h);
void orc_dmrh_no_carrier(orc_dmrh* h);
int orc_dmrh_begin(orc_dmrh* h, long pos, int cls, const uint8_t* pre90, const uint8_t* rel90, orc_hevents* ev);
int orc_dmrh_begin_fixed(orc_dmrh* h, int symbols);
int orc_dmrh_symbol(orc_dmrh* h, long pos, int d, int rel, orc_hevents* ev);

/* ---- fixed-protocol P25p1 C4FM receive loop: symbolizer + sync hunt + warm start + slicer (ddn_oracle_rx.c) -- */
typedef struct orc_p25rx {
    int out_rate, sym_rate, lock_symbols, use_filter;
    /* getSymbol() */
    int sps_accum, jitter, in_symbol, span, centre, i, count;
    float sum, lastsample;
    int filter_on;
    float fhist[90];
    /* getFrameSync() */
    int have_sync, lock_left, lastsync; /* lastsync: 0 none, 1 +P25p1, 2 -P25p1 */
    int lidx, level_count, hist_count;
    uint32_t hist_bits;
    float lbuf[24], lmin, lmax;
    float shist[24];
    int shead, scount;
    orc_slicer sl;
    int hunt_pos;     /* rt.synctest_pos: symbols hunted by this getFrameSync() call */
    int need_reset;   /* noCarrier() zeroed the timing ratio: the next getSymbol() re-initialises timing and slicer */
    /* lock_symbols < 0: the reference's per-DUID handlers decide the in-frame length (ddn_oracle_handlers.c) */
    orc_p25h h;
    long n_sym;       /* symbols emitted since init (event positions) */
    orc_hevents* ev;  /* optional event log */
} orc_p25rx;

/* ---- symbol-rate receive loop behind the CQPSK demodulator: P25 Phase 1 (LSM) / Phase 2 (ddn_oracle_cqrx.c) ------------------ */
enum { ORC_CQ_P25P1 = 0, ORC_CQ_P25P2 = 1 };
typedef struct orc_cqrx {
    int protocol, sync_len, t_max, lock_symbols; /* lock_symbols < 0: the P25p1 handlers decide (Phase 2: 700 per sync) */
    double snr_db;                               /* what dsd_rtl_stream_metrics_hook_snr_cqpsk_db() would return; <= -50: no weight */
    int have_sync, lock_left, lastsync;          /* lastsync: 0 none, 1 positive, 2 inverted pattern */
    int map_idx;                                 /* DSD_P25_CQPSK_DIBIT_MAP_* of the last sync */
    int lidx, level_count, hist_count;
    uint64_t hist;                               /* the last sync_len raw dibits, oldest in the high bits */
    uint64_t target[2][4];                       /* [polarity][identity, X2400, N1200, P1200] raw-dibit images of the sync word */
    float lbuf[24], lmin, lmax;
    float shist[24];
    int shead, scount;
    orc_slicer sl;
    int hunt_pos;
    orc_p25h h;
    long n_sym;
    orc_hevents* ev;
} orc_cqrx;
void orc_cqrx_init(orc_cqrx* r, int protocol, int lock_symbols, double snr_db);
void orc_cqrx_set_events(orc_cqrx* r, orc_hevents* ev);
int orc_cqrx_symbol(orc_cqrx* r, float sym, int rec4[4]);
long orc_cqrx_run(orc_cqrx* r, const float* sym, long n, int* rec4, uint8_t* flags);
size_t orc_cqrx_sizeof(void);
size_t orc_cqrx_slicer_offset(void);
void orc_cqrx_get_state(const orc_cqrx* r, float out8[8]);
int orc_cq_reliability(float sym_c, double snr_db);
void orc_cq_digitize(const orc_slicer* s, float sym, int map_idx, int negative, double snr_db, int rec4[4]);
void orc_cq_inframe_step(orc_slicer* s, float sym, int map_idx, int negative, double snr_db, int rec4[4]);

/* ---- fixed-protocol 4-level FSK receive loop, profile-driven: P25p1 / DMR / NXDN48 (ddn_oracle_rx4.c) ------------- */
#define ORC_FSK4_MAX_PAT  20
#define ORC_FSK4_MAX_TAPS 135
#define ORC_FSK4_HIST     96 /* symbols of history kept per stream (>= 90: DMR re-digitisation reach) */
#define ORC_FSK4_PRE      90 /* payload dibits handed over with every accepted sync */
typedef struct orc_fsk4_profile {
    int out_rate, sym_rate;
    int rf_mod;          /* 0 = C4FM lock (-mc), 2 = GFSK lock (-mg) */
    int win_len;         /* sync window in symbols: 24 (P25p1, DMR) or 10 (NXDN) */
    int t_max;           /* level ring length: 24, NXDN48 12 */
    int warm_len;        /* symbols the warm start averages: 24 / 10 */
    int n_pat;
    uint32_t pat_bits[ORC_FSK4_MAX_PAT]; /* sign bits ('1' -> 1, '3' -> 0), oldest symbol in bit win_len-1 */
    uint8_t pat_type[ORC_FSK4_MAX_PAT];  /* sync-type id (> 0; lastsynctype value) */
    uint8_t pat_neg[ORC_FSK4_MAX_PAT];   /* digitize() polarity of that type */
    uint8_t pat_class[ORC_FSK4_MAX_PAT]; /* 0..3: which lock_symbols[] entry the handler of that type consumes */
    int confirm;         /* NXDN: a match is accepted only when lastsynctype already is that type */
    int live_thresholds; /* P25p1: use_symbol() keeps min / max / mids live in frame */
    int dmr_window;      /* DMR: C4FM window left edge 1 once a DMR type is the last sync */
    int redigitize;      /* DMR: dmr_resample_on_sync() */
    int slow_type;       /* type id exempt from the 1800-symbol timeout (P25p1 NEG), or 0 */
    int use_filter, nt;
    uint32_t taps[ORC_FSK4_MAX_TAPS];
    int lock_symbols[4];
    int handler; /* 0 = lock_symbols[] per sync class, 1 = the reference's handlers decide (ddn_oracle_handlers.c) */
    int proto;   /* handler family when handler = 1: 0 P25p1, 1 DMR, 2 NXDN */
    int m17;     /* 1: frame_sync_try_m17()'s matcher (8-symbol words, one error allowed, accepted by what came before) in place of
                    the exact pattern table; the table then only carries type / polarity / class of the twelve outcomes */
} orc_fsk4_profile;
typedef struct orc_fsk4rx {
    orc_fsk4_profile p;
    int sps_accum, jitter, in_symbol, span, centre, i, count;
    float sum, lastsample;
    int filter_on;
    float fhist[ORC_FSK4_MAX_TAPS];
    int have_sync, lock_left, lastsync, cur_pat; /* lastsync = lastsynctype id (0 none); cur_pat = index of state->synctype */
    int lidx, level_count, hist_count;
    uint32_t hist_bits;
    float lbuf[24], lmin, lmax;
    float shist[ORC_FSK4_HIST];
    uint8_t phist[ORC_FSK4_HIST], rhist[ORC_FSK4_HIST]; /* payload dibit / reliability history, same ring */
    int shead, scount;
    orc_slicer sl;
    int hunt_pos, need_reset;
    orc_p25h hp25;
    orc_dmrh hdmr;
    orc_nxdnh hnxdn;
    long n_sym;
    orc_hevents* ev;
    int m17_pol; /* state->m17_polarity: 0 unknown, 1 normal, 2 inverted (set by the preamble, cleared by EOT / no carrier) */
    float* sync_thr; /* optional: {center, umid, lmid, max, min} as every accepted sync leaves them, [sync_thr_max][5] */
    int sync_thr_max, sync_thr_n;
} orc_fsk4rx;
void orc_fsk4rx_set_sync_thresholds(orc_fsk4rx* r, float* buf, int max_syncs);
void orc_fsk4rx_set_events(orc_fsk4rx* r, orc_hevents* ev);
void orc_p25rx_set_events(orc_p25rx* r, orc_hevents* ev);
void orc_fsk4rx_init(orc_fsk4rx* r, const orc_fsk4_profile* p);
/* per symbol: out_sym, rec4 {dibit, rel, llr0, llr1}, flags (1 in frame, 2 sync accepted, 4 negative, pattern index << 3
 * on the accepting symbol), pay2 {payload dibit, reliability}.  per accepted sync (up to max_sync): sync_pos (index of the
 * sync's last symbol in this call's output), sync_pat, pre[90] / pre_rel[90] = the payload history ending at that symbol
 * after re-digitisation.  *n_sync = syncs accepted in this call.  returns the symbol count. */
long orc_fsk4rx_run(orc_fsk4rx* r, const float* in, long n, float* out_sym, int* rec4, uint8_t* flags, uint8_t* pay2,
                    long max_out, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre, uint8_t* pre_rel, int max_sync,
                    int* n_sync);
int orc_fsk4_adjust_timing(int sps, int centre, int rf_mod, int jitter, int have_sync, int span, int start_i, int* jitter_after);
void orc_fsk4_window(int rf_mod, int narrow, int* l_edge, int* r_edge);
size_t orc_fsk4rx_sizeof(void);
size_t orc_fsk4_profile_sizeof(void);
void orc_fsk4rx_get_thresholds(const orc_fsk4rx* r, float out7[7]);

/* ---- YSF frame information channel (oracle/ddn_oracle_ysf.c) ---------------------------------------------------------- */
uint16_t orc_ysf_crc16(const uint8_t* bits, int len);
uint32_t orc_ysf_soft_viterbi(const uint8_t* dibits, int n, int decoded_bytes, int offset_bits, int output_bits, uint8_t* out_bits);
int orc_ysf_fich(const uint8_t in100[100], uint8_t fich32[32], uint32_t* v_error);
int orc_ysf_vd2_index(int k);
int orc_ysf_pn95_bit(int i);
void orc_ysf_dewhiten(uint8_t* bits, int n);
int orc_ysf_vd2_voice(const uint8_t dibits52[52], uint8_t ambe_d[49]);
int orc_ysf_dch(const uint8_t* in, int n, uint8_t* out_bytes, uint32_t* v_error);
void orc_ysf_fr_unpack(const uint8_t dibits72[72], uint8_t imbe_fr[8 * 23]);
int orc_ysf_voice_frames(const uint8_t p[360], int kind, int csd3, uint8_t fr[5][184], uint8_t dch20[20], uint8_t* dch_status,
                         uint32_t* dch_cost);
int orc_ysf_payload(const uint8_t p[360], int fi, int dt, uint8_t dch[2][20], uint8_t dch_status[2], uint32_t dch_cost[2],
                    uint8_t ambe_d[5][49], uint8_t errs2[5]);

/* ---- M17 frames behind the loop (oracle/ddn_oracle_m17.c) ---------------------------------------------------------- */
int orc_m17_rand_bit(int i);
int orc_m17_interleave_index(int i);
uint16_t orc_m17_crc16(const uint8_t* in, int len);
uint16_t orc_m17_soft_cost(float symbol, const float thr5[5], int bit);
void orc_m17_lsf_costs(const float* sym184, const float thr5[5], uint16_t cost488[488]);
int orc_m17_lsf_decode(const uint16_t cost488[488], uint8_t lsf30[30], uint32_t* path_cost);
void orc_m17_payload_bits(const uint8_t* dibits184, uint8_t bits368[368]);
int orc_m17_str_decode(const uint8_t* dibits184, uint8_t lich6[6], int* lich_cnt, uint8_t fn_payload18[18]);
int orc_m17_callsign(uint64_t address, char out10[10]);

void orc_level_estimate(const float* sorted, int count, float* lo, float* hi);
int orc_slicer_warm_start(orc_slicer* s, const float* newest_first, int sync_len);
void orc_p25rx_init(orc_p25rx* r, int out_rate_hz, int sym_rate_hz, int lock_symbols, int use_matched_filter);
long orc_p25rx_run(orc_p25rx* r, const float* in, long n, float* out_sym, int* rec4, uint8_t* flags, long max_out);
void orc_p25rx_get_thresholds(const orc_p25rx* r, float out7[7]);
size_t orc_p25rx_sizeof(void);

/* ---- DMR / NXDN block codes (oracle/ddn_oracle_fec3.c) ------------------------------------------------------------ */
int orc_hamming_7_4_decode(uint8_t* rx);
int orc_hamming_multi_decode(int which, uint8_t* rx, uint8_t* dec, int nb); /* which: 0 (12,8) 1 (13,9) 2 (15,11) 3 (16,11,4) */
int orc_golay_dmr_decode(int n, uint8_t* rx);                               /* n = 20 or 24 */
int orc_qr_16_7_6_decode(uint8_t* rx);
uint32_t orc_bptc_196x96(const uint8_t* in196, int deinterleave, uint8_t out96[96], uint8_t r3[3]);
uint32_t orc_bptc_128x77(const uint8_t in128[128], uint8_t out77[77], int* row0_failed);
uint32_t orc_bptc_16x2(const uint8_t in32[32], uint8_t out32[32], uint32_t parity_odd, int* hamming_failed);
int orc_bptc_last_col0_failed(void);
void orc_trellis_decode(uint8_t* result, const uint8_t* source, int result_len);
int orc_rs_12_9(uint8_t cw[12], uint8_t syn3[3], uint8_t* found);

/* ---- getSymbol()'s sample loop in all its window / timing variants (oracle/ddn_oracle_symbolizer.c) ---------- */
typedef struct orc_symbolizer {
    int out_rate, sym_rate, rf_mod, l_edge, r_edge;
    int sps_accum, jitter, last_sps, last_centre;
    float lastsample, center, min, max, minref, maxref;
} orc_symbolizer;
void orc_symbolizer_init(orc_symbolizer* s, int out_rate_hz, int sym_rate_hz, int rf_mod, int l_edge, int r_edge);
long orc_symbolizer_symbol(orc_symbolizer* s, const float* in, long n, int have_sync, float* out_sym);

/* ---- P25p1 Golay(24,12,8) + RS GF(64) hard-decision decoders (oracle/ddn_oracle_rs.c) ---------------------- */
int orc_golay_24_decode(uint8_t* data, int len, const uint8_t* parity, int* fixed);
int orc_rs63_decode(int* word, int t);
int orc_rs63_decode_erasures(int* word, int t, const int* erasures, int n_er);
/* P25 Phase 2 RS(63,35) sections with caller-given erasures (== ez_rs28_ess / _facch / _sacch, src/fec/ez.cpp:104-281) */
int orc_ez_rs28(int kind, int* payload, const int* parity, const int* erasures, int n_erasures);
/* P25 Phase 2 FACCH (kind 0) / SACCH (kind 1) burst decode with the ranked soft erasures (p25p2_frame.c:408-495,652-671, p25p2_soft.c) */
int orc_p25p2_ess(const uint8_t* payload_bits96, const int16_t* payload_llr96, const uint8_t* parity_bits168, const int16_t* parity_llr168,
                  int threshold, uint8_t* payload_out96, int* ec);
int orc_p25p2_duid_hard(int received);
int orc_p25p2_duid_lookup_soft(int received, const uint8_t* reliab8, int threshold);
int orc_p25p2_xcch(int kind, const uint8_t* bits360, const int16_t* llr360, int threshold, uint8_t* payload_out, int* used_dynamic);
/* P25 Phase 2 I-ISCH lookup (== isch_lookup / isch_lookup_soft, src/fec/ez.cpp:325-384) */
/* short-integer voice path (processAudio -> hpf_dL -> agsm), oracle/ddn_oracle_audio.c */
float orc_hpf_d_coef(void);
void orc_audio_s16(const float* pcm, int n_frames, float audio_gain, int use_hpf_d, int use_agsm, int16_t* out, float* state32,
                   float* gain_a);
int orc_isch_lookup(uint64_t isch);
int orc_isch_lookup_soft(uint64_t isch, const uint8_t* reliab40);
int orc_p25_rs_decode_soft(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t, const int* erasures,
                           int n_er);
int orc_p25_rs_ranked_erasures(const uint8_t* data_rel, int n_data, const uint8_t* parity_rel, int n_par, int min_er,
                               int* out, int max_er);
int orc_p25_rs_soft_reliability(uint8_t* data6, const uint8_t* parity6, const uint8_t* data_rel,
                                const uint8_t* parity_rel, int n_par, int n_data, int t);
int orc_hamming_10_6_3_soft(const uint8_t* bits, const int* reliab, uint8_t* out);
/* process_IMBE() de-interleave of 72 dibits (+ skipped status symbols); returns 1 when c0 is the non-standard word,
 * 0 otherwise, -1 when fewer than `consumed` dibits are available */
int orc_p25p1_imbe_deinterleave(const uint8_t* dibits, const int16_t* llr0, const int16_t* llr1, long n_avail,
                                int status_count, uint8_t fr[8][23], uint8_t soft[8][23][2], int* status_count_out,
                                int* consumed);
int orc_golay_24_soft(uint8_t* data, int len, const uint8_t* parity, const int* reliab, int* fixed);
int orc_p25_rs_decode(uint8_t* data6, const uint8_t* parity6, int n_par, int n_data, int t);

/* ---- block codes (oracle/ddn_oracle_block.c) ---------------------------------------------------------- */
int orc_bch_63_16_decode(const uint8_t in63[63], uint8_t out16[16], int* err_count);
void orc_p25p1_nid_decode(const uint8_t code[63], const uint8_t* rel63, int observed_nac, int parity, int parity_rel,
                          int threshold, int out4[4]);
int orc_hamming_10_6_3(int word10, int* fixed6);
int orc_p25_crc16_ok(const uint8_t* bytes, int payload_bytes); /* 0 good, 65535 bad */
int orc_p25_lsd_parity(int data);
int orc_p25_lsd_fec_16x8(uint8_t* bits16);                           /* 1 = valid / corrected, 0 = uncorrectable */
int orc_p25_lsd_fec_16x8_soft(uint8_t* bits16, const int16_t* llr16);

/* ---- rational L/M polyphase resampler (ddn_oracle_resamp.c) ---- */
#define ORC_RESAMP_MAX_L 512
typedef struct orc_resamp {
    int L, M, phase;
    float win[16];
    float taps[16 * ORC_RESAMP_MAX_L];
} orc_resamp;
int orc_resamp_design(int L, int M, float* taps); /* taps[16 * L], per phase oldest tap first; returns 16 * L */
size_t orc_resamp_sizeof(void);
int orc_resamp_init(orc_resamp* r, int L, int M);
long orc_resamp_run(orc_resamp* r, const float* in, long n, float* out, long cap);


#ifdef __cplusplus
}
#endif
#endif
