// ddn_api_chain_fsk4.cpp - the DMR / NXDN48 chain object and the mixed-protocol object over the three chains
// (include/ddn_chain.h): stage order, buffers and streams on top of the library's own C-ABI stage calls.  Host-only code.
//
// What it stands in for in a dsd-neo host: the demodulator thread's per-block loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516) and
// processFrame()'s DMR / NXDN branches (src/engine/protocol_dispatch.c -> dmr_data.c / dmr_bs.c, nxdn_frame.c), B channels wide.
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_chain.h"
#include "ddn_device.h"
#include "ddn_fsk4.h"
#include "ddn_hip.h"
#include "ddn_mbe.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            ddn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                  \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu)             \
                       ? DDN_ENODEV                                                                                    \
                       : (e_ == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP);                                          \
        }                                                                                                              \
    } while (0)
#define DDN_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        const int rc_ = (expr);                                                                                        \
        if (rc_ != DDN_OK) {                                                                                           \
            return rc_;                                                                                                \
        }                                                                                                              \
    } while (0)

struct ddn_fsk4_chain {
    ddn_fsk4_chain_config cfg;
    int B, n, dmr, vf, T, myc, myd;
    size_t ms, my, stride, S, V;
    ddn_batch* fe;
    ddn_fsk4_rx* rx;
    ddn_mbe_batch* mbe;
    float* d_disc;
    float* d_disc2; // mixed chain only: odd steps' discriminator output (the next call's front end beside this call's loop)
    // (round 6) recorded inside the decode stage once its last reader of the loop's single buffers (sync lists, events) and of the
    // records has been queued: what the NEXT call's loop has to wait for (the frame FEC and the synthesis behind it work on gathered
    // copies) - the mixed chain gates the group's next loop on it instead of on the whole decode stage
    hipEvent_t ev_reads;
    // rows = T carried records + this call's (two sets: the carry reads the previous call's)
    uint8_t *d_rec[2], *d_fl[2], *d_pay;
    int32_t *d_new[2], *d_cnt_full, *d_cnt_scan;
    // what the loop reports per call, the syncs waiting for the next call (two sets), the syncs decoded in this one
    int32_t *s_pos, *s_n, *c_pos[2], *c_n[2], *d_spos, *d_ns, *d_dropped;
    uint8_t *s_pat, *s_pre, *s_prel, *c_pat[2], *c_pre[2], *c_prel[2], *d_spat, *d_pre, *d_prel;
    // DMR
    uint8_t *d_st, *d_info, *d_cach, *d_valid, *d_st_ok, *d_pdu, *d_r3;
    uint32_t* d_errs;
    // DMR voice (vocoder = 1): the loop's handler decisions of the call, the voice bursts they name filed by talk path (2 per
    // channel: time slots 1 / 2), three AMBE frames each
    int E, vb;
    int32_t *d_ev, *d_nev, *d_vstart, *d_vpre, *d_vnb;
    // DMR data bursts the handlers dispatch (handlers = 1): db per channel and call, D = B * db; embedded link control: lb per talk
    // path and call, L = 2 B * lb (ddn_dmr_data.hip)
    int db, lb;
    size_t D, L;
    int32_t *dd_start, *dd_pre, *dd_n, *dd_listn, *dd_pooln, *de_pos, *de_n;
    uint32_t *dd_errs, *de_errs;
    uint8_t *dd_slot, *dd_st, *dd_st_ok, *dd_info, *dd_td, *dd_rel, *dd_pdu, *dd_r3, *dd_type, *dd_bytes, *dd_cw, *dd_rsres, *dd_rsfound,
        *dd_crc, *dd_want, *dd_hard, *dd_soft, *dd_list, *dd_backs, *dd_pool, *dd_unconf, *dd_conf, *dd_confcrc, *de_sig, *de_in, *de_out,
        *de_ok;
    // NXDN48
    uint8_t *d_lich, *d_ss, *d_sr, *d_fs, *d_fr, *d_sacch, *d_sacch_ok, *d_hard_in, *d_sacch_hard, *d_sacch_hard_ok, *d_facch, *d_facch_ok;
    int32_t *d_vpos, *d_vn, *d_ambe_res, *d_res_out;
    uint8_t *d_ambe_fr, *d_ambe_rel, *d_ambe_d, *d_skip;
    float* d_pcm;
    // M17 (protocol DDN_FSK4_M17): the thresholds every sync left (loop's list, carried lists, decode list), the frame decoders' slot
    // arrays (ddn_m17_*_batch), the carried LICH assembly buffer
    bool m17;
    float *s_thr, *c_thr[2], *d_thr;
    uint8_t *m_lsf, *m_lsf_st, *m_l6, *m_cnt, *m_fp, *m_st, *m_asm, *m_ll, *m_ll_st;
    uint32_t* m_cost;
    // YSF (protocol DDN_FSK4_YSF): the frame information channel of every decoded sync
    bool ysf;
    uint8_t *y_fich4, *y_st;
    uint32_t* y_ve;
    // ... and the payload behind it (ddn_ysf_payload_decode_batch): the frame type carried per channel, the data channels, V/D2 voice bits
    uint8_t *y_last, *y_info, *y_dch, *y_dst, *y_ambe, *y_errs, *y_fr, *y_nfr;
    uint32_t* y_dcost;
    // ... V/D mode 2 voice (vocoder = 1): the sub-frames filed by talk path (= channel) -> AMBE 3600x2450 synthesis (d_ambe_d, d_ambe_res,
    // d_skip, d_pcm, d_res_out, d_vn as for NXDN48; yvf frames of five sub-frames per channel and call)
    int yvf;
    int32_t* y_vslot;
    // ... V/D mode 1 (four AMBE frames through the frame FEC, filed with the V/D mode 2 sub-frames in stream order) and full-rate voice
    // (IMBE 7200x4400: a vocoder batch, talk-path history and PCM of its own)
    ddn_mbe_batch* mbe_i;
    uint8_t *y_f96, *y_b49, *y_b88, *yi_bits, *yi_skip;
    int32_t *y_r49, *y_r88, *yi_res, *yi_res_out, *yi_vn, *yi_vslot;
    float* yi_pcm;
    long step;
    int last_set;
};
extern "C" int ddn_m17_lsf_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                                        const uint8_t* d_sync_pat, const int32_t* d_n_sync, const float* d_sync_thr5, int n_channels,
                                        size_t max_syncs, uint8_t* d_lsf30, uint8_t* d_status, uint32_t* d_path_cost, void* hip_stream);
extern "C" int ddn_m17_str_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                                        const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_lich6,
                                        uint8_t* d_lich_cnt, uint8_t* d_fn_payload18, uint8_t* d_status, void* hip_stream);
extern "C" int ddn_m17_lich_assemble_batch(const uint8_t* d_sync_pat, const int32_t* d_n_sync, int n_channels, size_t max_syncs,
                                           const uint8_t* d_lsf30, const uint8_t* d_lsf_status, const uint8_t* d_lich6, const uint8_t* d_lich_cnt,
                                           const uint8_t* d_str_status, uint8_t* d_assembly32, uint8_t* d_lich_lsf30, uint8_t* d_lich_status,
                                           void* hip_stream);
extern "C" int ddn_fsk4_rx_set_sync_thresholds(ddn_fsk4_rx* b, float* d_thr5);
extern "C" int ddn_ysf_fich_decode_batch(const uint8_t* d_records10, size_t stride_symbols, const int32_t* d_counts, const int32_t* d_sync_pos,
                                         const int32_t* d_n_sync, int n_channels, size_t max_syncs, uint8_t* d_fich4, uint8_t* d_status,
                                         uint32_t* d_v_error, void* hip_stream);

template <typename T>
static bool
dalloc(T** p, size_t count) {
    if (hipMalloc((void**)p, count * sizeof(T) + 16) != hipSuccess) {
        return false;
    }
    return hipMemset(*p, 0, count * sizeof(T)) == hipSuccess;
}

extern "C" void
ddn_fsk4_chain_destroy(ddn_fsk4_chain* c) {
    if (!c) {
        return;
    }
    (void)hipDeviceSynchronize();
    if (c->ev_reads) {
        (void)hipEventDestroy(c->ev_reads);
    }
    ddn_batch_destroy(c->fe);
    ddn_fsk4_rx_destroy(c->rx);
    ddn_mbe_batch_destroy(c->mbe);
    ddn_mbe_batch_destroy(c->mbe_i);
    void* all[] = {c->y_f96, c->y_b49, c->y_b88, c->yi_bits, c->yi_skip, c->y_r49, c->y_r88, c->yi_res, c->yi_res_out, c->yi_vn, c->yi_vslot, c->yi_pcm, c->y_fr, c->y_nfr, c->y_vslot, c->y_fich4, c->y_st, c->y_ve, c->y_last, c->y_info, c->y_dch, c->y_dst, c->y_ambe, c->y_errs, c->y_dcost, c->s_thr, c->c_thr[0], c->c_thr[1], c->d_thr, c->m_lsf, c->m_lsf_st, c->m_l6, c->m_cnt, c->m_fp, c->m_st, c->m_asm, c->m_ll,
                   c->m_ll_st, c->m_cost, c->d_disc, c->d_disc2, c->d_rec[0], c->d_rec[1], c->d_fl[0], c->d_fl[1], c->d_pay, c->d_new[0], c->d_new[1], c->d_cnt_full,
                   c->d_cnt_scan, c->d_dropped, c->s_pos, c->s_n, c->c_pos[0], c->c_pos[1], c->c_n[0], c->c_n[1], c->d_spos, c->d_ns, c->s_pat, c->s_pre,
                   c->s_prel, c->c_pat[0], c->c_pat[1], c->c_pre[0], c->c_pre[1], c->c_prel[0], c->c_prel[1], c->d_spat, c->d_pre,
                   c->d_prel, c->d_st, c->d_info, c->d_cach, c->d_valid, c->d_st_ok, c->d_pdu, c->d_r3, c->d_errs, c->d_lich, c->d_ss,
                   c->d_sr, c->d_fs, c->d_fr, c->d_sacch, c->d_sacch_ok, c->d_hard_in, c->d_sacch_hard, c->d_sacch_hard_ok, c->d_facch,
                   c->d_facch_ok, c->d_vpos, c->d_vn, c->d_ambe_res, c->d_res_out, c->d_ambe_fr, c->d_ambe_rel, c->d_ambe_d, c->d_skip,
                   c->d_pcm, c->d_ev, c->d_nev, c->d_vstart, c->d_vpre, c->d_vnb, c->dd_start, c->dd_pre, c->dd_n, c->dd_listn, c->dd_pooln, c->de_pos,
                   c->de_n, c->dd_errs, c->de_errs, c->dd_slot, c->dd_st, c->dd_st_ok, c->dd_info, c->dd_td, c->dd_rel, c->dd_pdu, c->dd_r3,
                   c->dd_type, c->dd_bytes, c->dd_cw, c->dd_rsres, c->dd_rsfound, c->dd_crc, c->dd_want, c->dd_hard, c->dd_soft, c->dd_list,
                   c->dd_backs, c->dd_pool, c->dd_unconf, c->dd_conf, c->dd_confcrc, c->de_sig, c->de_in, c->de_out, c->de_ok};
    for (void* p : all) {
        (void)hipFree(p);
    }
    delete c;
}

extern "C" hipError_t ddn_dev_ysf_voice_file(const int32_t* n_sync, int n_channels, int max_syncs, const uint8_t* info, const uint8_t* ambe49,
                                             const uint8_t* errs2, const uint8_t* bits_fd, const int32_t* res_fd, const uint8_t* n_frames,
                                             int mode, int vf, uint8_t* bits, int32_t* res, uint8_t* skip, int32_t* v_n, int32_t* v_slot,
                                             hipStream_t st);
extern "C" hipError_t ddn_dev_ysf_pack96(const uint8_t* frames184, size_t n, uint8_t* frames96, hipStream_t st);

extern "C" int
ddn_fsk4_chain_create(const ddn_fsk4_chain_config* cfg, ddn_fsk4_chain** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->samples_per_call <= 0 || cfg->block_len <= 0
        || (cfg->protocol != DDN_FSK4_DMR && cfg->protocol != DDN_FSK4_NXDN48 && cfg->protocol != DDN_FSK4_NXDN96 && cfg->protocol != DDN_FSK4_M17
            && cfg->protocol != DDN_FSK4_YSF)
        || ((cfg->protocol == DDN_FSK4_M17 || cfg->protocol == DDN_FSK4_YSF) && (cfg->handlers || cfg->inverted))) {
        ddn_set_error("ddn_fsk4_chain_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    ddn_fsk4_chain* c = new (std::nothrow) ddn_fsk4_chain();
    if (!c) {
        return DDN_ENOMEM;
    }
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->B = cfg->n_channels;
    c->n = cfg->samples_per_call;
    c->dmr = cfg->protocol == DDN_FSK4_DMR;
    c->m17 = cfg->protocol == DDN_FSK4_M17; // (an M17 frame ends 184 symbols after its sync: inside the same tail)
    c->ysf = cfg->protocol == DDN_FSK4_YSF; // (the FICH ends 100 symbols after its sync)
    c->T = 256;  // a DMR burst ends 54 symbols after its sync, an NXDN frame 182: the tail kept back for the next call
    c->myc = 16; // syncs that can lie inside that tail (a new sync needs 24 / 10 fresh symbols)
    if (c->ysf) {
        c->T = 480; // (a YSF frame's payload ends 460 symbols after its sync)
    }
    int rc = DDN_OK;
    do {
        // (NXDN96: a 12.5 kHz channel at 4800 symbols/s)
        const bool wide = c->dmr || cfg->protocol == DDN_FSK4_NXDN96 || c->m17 || c->ysf;
        ddn_front_end_config fc = {c->B, 48000, wide ? 4800 : 2400, 4, wide ? DDN_LPF_12K5 : DDN_LPF_6K25, cfg->input_format,
                                   cfg->block_len, 0.0f};
        if ((rc = ddn_batch_create(&fc, &c->fe)) != DDN_OK) {
            break;
        }
        if (hipEventCreateWithFlags(&c->ev_reads, hipEventDisableTiming) != hipSuccess) {
            rc = DDN_EHIP;
            break;
        }
        ddn_fsk4_rx_config rcfg;
        memset(&rcfg, 0, sizeof(rcfg));
        rcfg.n_channels = c->B;
        rcfg.out_rate_hz = 48000;
        rcfg.protocol = cfg->protocol;
        rcfg.rf_mod = cfg->rf_mod;
        rcfg.inverted = cfg->inverted;
        rcfg.use_matched_filter = 1;
        if ((rc = ddn_fsk4_rx_create(&rcfg, &c->rx)) != DDN_OK) {
            break;
        }
        if (cfg->handlers && (rc = ddn_fsk4_rx_set_handlers(c->rx, 1)) != DDN_OK) {
            break;
        }
        c->ms = ddn_fsk4_rx_max_symbols(c->rx, (size_t)c->n);
        c->my = ddn_fsk4_rx_max_syncs(c->rx, (size_t)c->n);
        c->stride = (size_t)c->T + c->ms;
        // Decode slots per channel and call.  The loop's own bound (a sync per window length) is what noise could do in theory;
        // with the handlers in the loop accepted syncs are bursts / frames (144 / 192 symbols apart), so twice the densest real
        // traffic + the carried ones is what every decode launch is sized for - a sync beyond that is counted in d_dropped_syncs.
        {
            const size_t dense = c->ms / 64 + 24 + (size_t)c->myc, loop_bound = c->my + (size_t)c->myc;
            c->myd = (int)(cfg->handlers && dense < loop_bound ? dense : loop_bound);
        }
        c->S = (size_t)c->B * (size_t)c->myd;
        const size_t B = (size_t)c->B, S = c->S, my = c->my, myc = (size_t)c->myc;
        bool ok = dalloc(&c->d_disc, B * (size_t)c->n) && dalloc(&c->d_pay, B * c->stride * 2) && dalloc(&c->d_cnt_full, B)
                  && dalloc(&c->d_cnt_scan, B) && dalloc(&c->d_dropped, B) && dalloc(&c->s_pos, B * my) && dalloc(&c->s_n, B) && dalloc(&c->s_pat, B * my)
                  && dalloc(&c->s_pre, B * my * 90) && dalloc(&c->s_prel, B * my * 90) && dalloc(&c->d_spos, S) && dalloc(&c->d_ns, B)
                  && dalloc(&c->d_spat, S) && dalloc(&c->d_pre, S * 90) && dalloc(&c->d_prel, S * 90);
        for (int k = 0; k < 2 && ok; k++) {
            ok = dalloc(&c->d_rec[k], B * c->stride * 10) && dalloc(&c->d_fl[k], B * c->stride) && dalloc(&c->d_new[k], B)
                 && dalloc(&c->c_pos[k], B * myc) && dalloc(&c->c_n[k], B) && dalloc(&c->c_pat[k], B * myc)
                 && dalloc(&c->c_pre[k], B * myc * 90) && dalloc(&c->c_prel[k], B * myc * 90);
        }
        if (ok && c->ysf) {
            ok = dalloc(&c->y_fich4, S * 4) && dalloc(&c->y_st, S) && dalloc(&c->y_ve, S) && dalloc(&c->y_last, B * 2) && dalloc(&c->y_info, S * 2)
                 && dalloc(&c->y_dch, S * 40) && dalloc(&c->y_dst, S * 2) && dalloc(&c->y_dcost, S * 2) && dalloc(&c->y_ambe, S * 5 * 49)
                 && dalloc(&c->y_errs, S * 5) && dalloc(&c->y_fr, S * 5 * 184) && dalloc(&c->y_nfr, S);
            if (ok && cfg->vocoder) {
                c->yvf = (int)(c->stride / 480 + 2);
                const size_t V5 = B * (size_t)c->yvf * 5;
                ok = dalloc(&c->y_vslot, B * (size_t)c->yvf) && dalloc(&c->d_vn, B) && dalloc(&c->d_ambe_d, V5 * 49) && dalloc(&c->d_ambe_res, V5 * 5)
                     && dalloc(&c->d_skip, V5) && dalloc(&c->d_pcm, V5 * 160) && dalloc(&c->d_res_out, V5 * 5);
                ok = ok && dalloc(&c->y_f96, S * 5 * 96) && dalloc(&c->y_b49, S * 5 * 49) && dalloc(&c->y_r49, S * 5 * 5) && dalloc(&c->y_b88, S * 5 * 88)
                     && dalloc(&c->y_r88, S * 5 * 5) && dalloc(&c->yi_bits, V5 * 88) && dalloc(&c->yi_res, V5 * 5) && dalloc(&c->yi_res_out, V5 * 5)
                     && dalloc(&c->yi_skip, V5) && dalloc(&c->yi_pcm, V5 * 160) && dalloc(&c->yi_vn, B) && dalloc(&c->yi_vslot, B * (size_t)c->yvf);
                if (ok && (rc = ddn_mbe_batch_create(DDN_MBE_AMBE_3600X2450, c->B, &c->mbe)) != DDN_OK) {
                    break;
                }
                if (ok && (rc = ddn_mbe_batch_create(DDN_MBE_IMBE_7200X4400, c->B, &c->mbe_i)) != DDN_OK) {
                    break;
                }
            }
        } else if (ok && c->m17) {
            ok = dalloc(&c->s_thr, B * my * 5) && dalloc(&c->c_thr[0], B * myc * 5) && dalloc(&c->c_thr[1], B * myc * 5) && dalloc(&c->d_thr, S * 5)
                 && dalloc(&c->m_lsf, S * 30) && dalloc(&c->m_lsf_st, S) && dalloc(&c->m_l6, S * 6) && dalloc(&c->m_cnt, S) && dalloc(&c->m_fp, S * 18)
                 && dalloc(&c->m_st, S) && dalloc(&c->m_asm, B * 32) && dalloc(&c->m_ll, S * 30) && dalloc(&c->m_ll_st, S) && dalloc(&c->m_cost, S);
            if (ok && (rc = ddn_fsk4_rx_set_sync_thresholds(c->rx, c->s_thr)) != DDN_OK) {
                break;
            }
        } else if (ok && c->dmr) {
            ok = dalloc(&c->d_st, S * 20) && dalloc(&c->d_info, S * 196) && dalloc(&c->d_cach, S * 24) && dalloc(&c->d_valid, S)
                 && dalloc(&c->d_st_ok, S) && dalloc(&c->d_pdu, S * 96) && dalloc(&c->d_r3, S * 3) && dalloc(&c->d_errs, S);
            if (ok && cfg->handlers) {
                // the handlers' decisions of every call (events): which bursts go to dmr_data_burst_handler(), which to the vocoder,
                // under which VC a burst's sync field was filed.  Data bursts: at most one per 144 symbols; an embedded link control
                // per six voice bursts of a time slot
                c->E = (int)(c->ms / 36 + 32);
                c->db = (int)(c->ms / 144 + 3);
                c->lb = (int)(c->ms / (288 * 6) + 2);
                c->D = B * (size_t)c->db;
                c->L = 2 * B * (size_t)c->lb;
                const size_t D = c->D, L = c->L;
                ok = dalloc(&c->d_ev, B * (size_t)c->E * 4) && dalloc(&c->d_nev, B) && dalloc(&c->dd_start, D) && dalloc(&c->dd_pre, D) && dalloc(&c->dd_n, B)
                     && dalloc(&c->dd_listn, D) && dalloc(&c->dd_pooln, D) && dalloc(&c->de_pos, L) && dalloc(&c->de_n, 2 * B)
                     && dalloc(&c->dd_errs, D) && dalloc(&c->de_errs, L) && dalloc(&c->dd_slot, D) && dalloc(&c->dd_st, D * 20)
                     && dalloc(&c->dd_st_ok, D) && dalloc(&c->dd_info, D * 196) && dalloc(&c->dd_td, D * 98) && dalloc(&c->dd_rel, D * 98)
                     && dalloc(&c->dd_pdu, D * 96) && dalloc(&c->dd_r3, D * 3) && dalloc(&c->dd_type, D) && dalloc(&c->dd_bytes, D * 12)
                     && dalloc(&c->dd_cw, D * 12) && dalloc(&c->dd_rsres, D) && dalloc(&c->dd_rsfound, D) && dalloc(&c->dd_crc, D)
                     && dalloc(&c->dd_want, D) && dalloc(&c->dd_hard, D * 18) && dalloc(&c->dd_soft, D * 18)
                     && dalloc(&c->dd_list, D * 32 * 24) && dalloc(&c->dd_backs, D * 49 * 8 * 32) && dalloc(&c->dd_pool, D * 34 * 24)
                     && dalloc(&c->dd_unconf, D * 18) && dalloc(&c->dd_conf, D * 18) && dalloc(&c->dd_confcrc, D)
                     && dalloc(&c->de_sig, B * 2 * 7 * 48) && dalloc(&c->de_in, L * 128) && dalloc(&c->de_out, L * 77) && dalloc(&c->de_ok, L);
                if (ok && (rc = ddn_fsk4_rx_set_events(c->rx, c->d_ev, c->d_nev, (size_t)c->E)) != DDN_OK) {
                    break;
                }
            }
            if (ok && cfg->vocoder && cfg->handlers) {
                // voice (dmrBSBootstrap / dmrBS -> processMbeFrame, dmr_bs.c:128-200,585-640): a time slot carries a burst every
                // 288 symbols, three AMBE 3600x2450 frames each; which bursts reach the vocoder is the handlers' decision (events)
                c->vb = (int)(c->ms / 288 + 3);
                c->V = 2 * B * (size_t)c->vb; // bursts
                const size_t V = c->V;
                ok = dalloc(&c->d_vstart, V) && dalloc(&c->d_vpre, V)
                     && dalloc(&c->d_vnb, 2 * B) && dalloc(&c->d_ambe_fr, V * 3 * 96) && dalloc(&c->d_ambe_d, V * 3 * 49)
                     && dalloc(&c->d_ambe_res, V * 3 * 5) && dalloc(&c->d_skip, V * 3) && dalloc(&c->d_pcm, V * 3 * 160)
                     && dalloc(&c->d_res_out, V * 3 * 5);
                if (ok && (rc = ddn_mbe_batch_create(DDN_MBE_AMBE_3600X2450, 2 * c->B, &c->mbe)) != DDN_OK) {
                    break;
                }
            }
        } else if (ok) {
            // voice: four AMBE frames per NXDN frame, one talk path per channel.  With the handlers deciding the frame length two
            // syncs are at least a 192-symbol frame apart: a call decodes n / (192 * 20) + 3 frames at most
            const size_t cap = (size_t)c->n / (192 * 20) + 3;
            c->vf = (int)(cfg->handlers ? (cap < (size_t)c->myd ? cap : (size_t)c->myd) : (size_t)c->myd);
            c->V = B * (size_t)c->vf;
            const size_t V = c->V;
            ok = dalloc(&c->d_lich, S) && dalloc(&c->d_valid, S) && dalloc(&c->d_ss, S * 72) && dalloc(&c->d_sr, S * 72)
                 && dalloc(&c->d_fs, S * 384) && dalloc(&c->d_fr, S * 384) && dalloc(&c->d_sacch, S * 4) && dalloc(&c->d_sacch_ok, S)
                 && dalloc(&c->d_hard_in, S * 72) && dalloc(&c->d_sacch_hard, S * 32) && dalloc(&c->d_sacch_hard_ok, S)
                 && dalloc(&c->d_facch, S * 2 * 12) && dalloc(&c->d_facch_ok, S * 2) && dalloc(&c->d_vpos, V) && dalloc(&c->d_vn, B)
                 && dalloc(&c->d_ambe_fr, V * 384) && dalloc(&c->d_ambe_rel, V * 384) && dalloc(&c->d_ambe_d, V * 4 * 49)
                 && dalloc(&c->d_ambe_res, V * 4 * 5) && dalloc(&c->d_skip, V * 4) && dalloc(&c->d_pcm, V * 4 * 160)
                 && dalloc(&c->d_res_out, V * 4 * 5);
            if (ok && (rc = ddn_mbe_batch_create(DDN_MBE_AMBE_3600X2450, c->B, &c->mbe)) != DDN_OK) {
                break;
            }
        }
        if (!ok) {
            ddn_set_error("ddn_fsk4_chain_create: device allocation failed");
            rc = DDN_ENOMEM;
        }
    } while (0);
    if (rc != DDN_OK) {
        ddn_fsk4_chain_destroy(c);
        return rc;
    }
    *out = c;
    return DDN_OK;
}

// frame FEC (+ voice) of the syncs this call decodes, out of buffer set `cur`
static int
fsk4_decode(ddn_fsk4_chain* c, int cur, int flush, hipStream_t st) {
    const size_t S = c->S;
    const int prev = cur ^ 1;
    const uint8_t* rec = c->d_rec[cur];
    bool reads_recorded = false;
    HIP_TRY(ddn_dev_chain_counts(c->d_new[cur], c->T, c->B, flush, c->d_cnt_scan, c->d_cnt_full, st));
    HIP_TRY(ddn_dev_fsk4_chain_syncs_thr(c->c_pos[prev], c->c_pat[prev], c->c_pre[prev], c->c_prel[prev], c->c_n[prev], c->myc, c->s_pos,
                                         c->s_pat, c->s_pre, c->s_prel, c->s_n, (int)c->my, c->d_new[cur], c->T, flush, c->d_spos, c->d_spat,
                                         c->d_pre, c->d_prel, c->d_ns, c->myd, c->c_pos[cur], c->c_pat[cur], c->c_pre[cur], c->c_prel[cur],
                                         c->c_n[cur], c->d_dropped, c->B, c->m17 ? c->c_thr[prev] : nullptr, c->m17 ? c->s_thr : nullptr,
                                         c->d_thr, c->m17 ? c->c_thr[cur] : nullptr, st));
    if (c->ysf) { // the frame information channel behind every sync of the decode list (row a17's second consumer)
        DDN_TRY(ddn_ysf_fich_decode_batch(rec, c->stride, c->d_cnt_full, c->d_spos, c->d_ns, c->B, (size_t)c->myd, c->y_fich4, c->y_st, c->y_ve, st));
        // ... and the payload of every frame: V/D mode 2 voice bits + DCH2, the DCH blocks of V/D mode 1 and of the full-rate data frames
        DDN_TRY(ddn_ysf_payload_decode_batch(rec, c->stride, c->d_cnt_full, c->d_spos, c->d_ns, c->B, (size_t)c->myd, c->y_fich4, c->y_st,
                                             c->y_last, c->y_info, c->y_dch, c->y_dst, c->y_dcost, c->y_ambe, c->y_errs, c->y_fr, c->y_nfr, st));
        if (c->mbe) { // mbe_processAmbe2450Dataf of every V/D mode 2 sub-frame, talk path = channel (ysf_handle_vd_type2, ysf.c:753-755)
            // the frames of V/D mode 1 and of full-rate voice through the frame FEC (processMbeFrame's hard decode, dsd_mbe.c:54-92), slot by slot
            const size_t S5 = c->S * 5, V5 = (size_t)c->B * (size_t)c->yvf * 5;
            HIP_TRY(ddn_dev_ysf_pack96(c->y_fr, S5, c->y_f96, st));
            DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, c->y_f96, nullptr, S5, c->y_b49, c->y_r49, st));
            DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_IMBE_7200X4400, c->y_fr, nullptr, S5, c->y_b88, c->y_r88, st));
            HIP_TRY(ddn_dev_ysf_voice_file(c->d_ns, c->B, c->myd, c->y_info, c->y_ambe, c->y_errs, c->y_b49, c->y_r49, c->y_nfr, 0, c->yvf,
                                           c->d_ambe_d, c->d_ambe_res, c->d_skip, c->d_vn, c->y_vslot, st));
            DDN_TRY(ddn_mbe_result_skip_batch(c->d_skip, V5, c->d_ambe_res, st));
            DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_ambe_d, c->d_ambe_res, (size_t)c->yvf * 5, c->d_pcm, c->d_res_out, st));
            HIP_TRY(ddn_dev_ysf_voice_file(c->d_ns, c->B, c->myd, c->y_info, c->y_ambe, c->y_errs, c->y_b88, c->y_r88, c->y_nfr, 1, c->yvf,
                                           c->yi_bits, c->yi_res, c->yi_skip, c->yi_vn, c->yi_vslot, st));
            DDN_TRY(ddn_mbe_result_skip_batch(c->yi_skip, V5, c->yi_res, st));
            DDN_TRY(ddn_mbe_synth_batch(c->mbe_i, c->yi_bits, c->yi_res, (size_t)c->yvf * 5, c->yi_pcm, c->yi_res_out, st));
        }
        HIP_TRY(hipEventRecord(c->ev_reads, st));
        return DDN_OK;
    }
    if (c->m17) {
        // the frames behind the syncs of this call's decode list (each complete inside the row): link setup frames through the K = 5
        // decoder of row a17, stream frames (LICH + payload), the LSF reassembled from the LICH chunks across calls
        DDN_TRY(ddn_m17_lsf_decode_batch(rec, c->stride, c->d_cnt_full, c->d_spos, c->d_spat, c->d_ns, c->d_thr, c->B, (size_t)c->myd, c->m_lsf,
                                         c->m_lsf_st, c->m_cost, st));
        DDN_TRY(ddn_m17_str_decode_batch(rec, c->stride, c->d_cnt_full, c->d_spos, c->d_spat, c->d_ns, c->B, (size_t)c->myd, c->m_l6, c->m_cnt,
                                         c->m_fp, c->m_st, st));
        DDN_TRY(ddn_m17_lich_assemble_batch(c->d_spat, c->d_ns, c->B, (size_t)c->myd, c->m_lsf, c->m_lsf_st, c->m_l6, c->m_cnt, c->m_st, c->m_asm,
                                            c->m_ll, c->m_ll_st, st));
        HIP_TRY(hipEventRecord(c->ev_reads, st));
        return DDN_OK;
    }
    if (c->dmr) {
        // burst gather -> slot type Golay(20,8) -> BPTC(196,96)
        DDN_TRY(ddn_dmr_burst_gather(rec, c->d_cnt_full, c->stride, c->d_spos, c->d_pre, c->d_ns, c->B, (size_t)c->myd, c->cfg.inverted,
                                     c->d_st, c->d_info, c->d_cach, c->d_valid, st));
        DDN_TRY(ddn_fec_block_code_batch(5 /* DDN_CODE_GOLAY_20_8 */, c->d_st, S, 1, nullptr, c->d_st_ok, st));
        DDN_TRY(ddn_fec_bptc_196x96_batch(c->d_info, 1, S, c->d_pdu, c->d_r3, c->d_errs, st));
        if (c->E && flush) { // no new records, no new decisions
            HIP_TRY(hipMemsetAsync(c->d_nev, 0, sizeof(int32_t) * (size_t)c->B, st));
        }
        if (c->E) {
            // the bursts the handlers dispatched to dmr_data_burst_handler() in this call (each ends inside it; one that began in the
            // previous call reaches back into the carried records): slot type, BPTC(196,96), the type's CRC / RS(12,9), and for
            // rate 3/4 bursts the three trellis decoders and the candidate pool (dmr_dburst.c:502-536)
            const size_t D = c->D, L = c->L;
            HIP_TRY(ddn_dev_dmr_data_select(c->d_ev, c->d_nev, c->E, c->T, c->B, c->db, c->d_spos, c->d_ns, c->myd, c->c_pos[cur], c->c_n[cur],
                                            c->myc, c->d_new[cur], c->dd_start, c->dd_slot, c->dd_pre, c->dd_n, st));
            HIP_TRY(ddn_dev_dmr_data_gather(rec, c->stride, c->dd_start, c->dd_pre, c->d_pre, c->d_prel, c->c_pre[cur], c->c_prel[cur],
                                            (long)c->S, c->db, c->B, c->dd_st, c->dd_info, c->dd_td, c->dd_rel, st));
            DDN_TRY(ddn_fec_block_code_batch(5 /* DDN_CODE_GOLAY_20_8 */, c->dd_st, D, 1, nullptr, c->dd_st_ok, st));
            DDN_TRY(ddn_fec_bptc_196x96_batch(c->dd_info, 1, D, c->dd_pdu, c->dd_r3, c->dd_errs, st));
            HIP_TRY(ddn_dev_dmr_data_prep(c->dd_start, c->dd_st, c->dd_st_ok, c->dd_pdu, (int)D, c->dd_type, c->dd_bytes, c->dd_cw, st));
            DDN_TRY(ddn_fec_rs_12_9_batch(c->dd_cw, D, c->dd_rsres, c->dd_rsfound, nullptr, st));
            HIP_TRY(ddn_dev_dmr_data_finish(c->dd_type, c->dd_pdu, c->dd_info, c->dd_cw, c->dd_rsres, (int)D, c->dd_bytes, c->dd_crc,
                                            c->dd_want, st));
            DDN_TRY(ddn_fec_r34_batch(c->dd_td, nullptr, D, c->dd_hard, st));
            DDN_TRY(ddn_fec_r34_batch(c->dd_td, c->dd_rel, D, c->dd_soft, st));
            HIP_TRY(ddn_dev_r34_list_wanted(c->dd_td, c->dd_rel, (int)D, 32, c->dd_want, c->dd_backs, (uint32_t*)c->dd_list, c->dd_listn, st));
            HIP_TRY(ddn_dev_dmr_r34_pick(c->dd_td, c->dd_rel, c->dd_want, c->dd_hard, c->dd_soft, c->dd_list, c->dd_listn, (int)D, c->dd_pool,
                                         c->dd_pooln, c->dd_unconf, c->dd_conf, c->dd_confcrc, st));
            // embedded link control: the sync fields filed under VC 2..6, BPTC(128,77) at every voice burst with VC 6
            HIP_TRY(ddn_dev_dmr_emb_collect(c->d_ev, c->d_nev, c->E, c->T, rec, c->stride, c->B, c->lb, c->de_sig, c->de_in, c->de_pos,
                                            c->de_n, st));
            DDN_TRY(ddn_fec_bptc_128x77_batch(c->de_in, L, c->de_out, c->de_errs, st));
            HIP_TRY(ddn_dev_dmr_emb_finish(c->de_out, c->de_pos, (int)L, c->de_ok, st));
        }
        if (c->mbe) {
            // voice: the bursts the handlers handed to the vocoder in this call (they end inside it; a burst that began in the
            // previous call reaches back into the carried records), filed by time slot -> 3 AMBE frames -> frame FEC -> synthesis.
            // (hard bits: the reference passes no soft frame here, processMbeFrame(opts, state, NULL, frame, NULL))
            const size_t V3 = c->V * 3;
            HIP_TRY(ddn_dev_dmr_voice_select(c->d_ev, c->d_nev, c->E, c->T, c->d_spos, c->d_ns, c->myd, c->B, c->vb, c->d_vstart,
                                             c->d_vpre, c->d_vnb, c->c_pos[cur], c->c_n[cur], c->myc, c->d_new[cur], st));
            HIP_TRY(ddn_dev_dmr_voice_gather_paths(rec, c->d_cnt_full, c->stride, c->d_vstart, c->d_vpre, c->d_pre, c->vb, c->B, 0,
                                                   c->d_ambe_fr, c->d_skip, c->c_pre[cur], (long)c->S, st));
            HIP_TRY(hipEventRecord(c->ev_reads, st)); // (everything below works on the gathered frames)
            reads_recorded = true;
            DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, c->d_ambe_fr, nullptr, V3, c->d_ambe_d, c->d_ambe_res, st));
            DDN_TRY(ddn_mbe_result_skip_batch(c->d_skip, V3, c->d_ambe_res, st));
            DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_ambe_d, c->d_ambe_res, (size_t)c->vb * 3, c->d_pcm, c->d_res_out, st));
        }
        if (!reads_recorded) {
            HIP_TRY(hipEventRecord(c->ev_reads, st));
        }
        return DDN_OK;
    }
    // NXDN48: frame gather -> SACCH / FACCH1 K=5 soft decode -> CRC6 / CRC12 -> the reference's greedy retry for the SACCH
    DDN_TRY(ddn_nxdn_frame_gather(rec, c->d_cnt_full, c->stride, c->d_spos, c->d_ns, c->B, (size_t)c->myd, c->d_lich, c->d_ss, c->d_sr,
                                  c->d_fs, c->d_fr, c->d_valid, st));
    if (c->cfg.vocoder) {
        // voice, first half: which frames the LICHs announce, and their AMBE words out of the records (both only need the frame
        // gather's LICHs; done here so that every reader of the loop's buffers sits at the head of the stage)
        HIP_TRY(ddn_dev_nxdn_voice_select(c->d_spos, c->d_ns, c->d_lich, c->d_valid, c->B, c->myd, c->vf, c->d_vpos, c->d_vn, c->d_skip, st));
        DDN_TRY(ddn_nxdn_voice_gather(rec, c->d_cnt_full, c->stride, c->d_vpos, c->d_vn, c->B, (size_t)c->vf, c->d_ambe_fr, c->d_ambe_rel,
                                      nullptr, st));
    }
    HIP_TRY(hipEventRecord(c->ev_reads, st)); // (the decoders and the synthesis below work on the gathered words)
    // (the decoders skip the slots that hold no complete frame - d_valid - and write zeros there: the slot arrays are sized for the
    // densest traffic, a call of the bench capture uses an eighth of them)
    HIP_TRY(ddn_dev_k5_nxdn_wanted(c->d_ss, c->d_sr, (int)S, 36, 32, nullptr, c->d_sacch, 4, c->d_valid, 1, st));
    DDN_TRY(ddn_nxdn_crc_check_batch(c->d_sacch, 4, S, 0, c->d_sacch_ok, st));
    HIP_TRY(ddn_dev_u8_shr1(c->d_ss, S * 72, c->d_hard_in, st));
    HIP_TRY(ddn_dev_trellis_greedy_wanted(c->d_hard_in, 72, S, 32, c->d_sacch_hard, 32, c->d_valid, st));
    DDN_TRY(ddn_nxdn_crc_check_batch(c->d_sacch_hard, 32, S, 2, c->d_sacch_hard_ok, st));
    HIP_TRY(ddn_dev_k5_nxdn_wanted(c->d_fs, c->d_fr, (int)(S * 2), 96, 92, nullptr, c->d_facch, 12, c->d_valid, 2, st));
    DDN_TRY(ddn_nxdn_crc_check_batch(c->d_facch, 12, S * 2, 1, c->d_facch_ok, st));
    if (c->cfg.vocoder) {
        // voice (nxdn_voice()): the frames the LICHs announce (selected and gathered above), through frame FEC -> synthesis
        const size_t V4 = c->V * 4;
        DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, c->d_ambe_fr, c->d_ambe_rel, V4, c->d_ambe_d, c->d_ambe_res, st));
        DDN_TRY(ddn_mbe_result_skip_batch(c->d_skip, V4, c->d_ambe_res, st));
        DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_ambe_d, c->d_ambe_res, (size_t)c->vf * 4, c->d_pcm, c->d_res_out, st));
    }
    return DDN_OK;
}

// stage 0: front end, 1: carry + matched filter + receive loop, 2: frame FEC (+ voice)
extern "C" int
ddn_fsk4_chain_stage(ddn_fsk4_chain* c, int stage, const void* d_iq, void* hip_stream) {
    if (!c || stage < 0 || stage > 2 || (stage == 0 && !d_iq)) {
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int cur = (int)(c->step & 1), prev = cur ^ 1;
    float* disc = (c->d_disc2 && (c->step & 1)) ? c->d_disc2 : c->d_disc;
    if (stage == 0) {
        return ddn_front_end_run(c->fe, d_iq, (size_t)c->n, disc, st);
    }
    if (stage == 1) {
        HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_rec[cur], c->d_fl[cur],
                                    c->stride, c->T, c->B, st));
        // the loop writes behind the T carried records: row pointers + T, row stride unchanged
        return ddn_fsk4_rx_run(c->rx, disc, (size_t)c->n, c->d_rec[cur] + (size_t)c->T * 10, c->d_fl[cur] + c->T,
                               c->d_pay + (size_t)c->T * 2, c->d_new[cur], c->stride, c->s_pos, c->s_pat, c->s_pre, c->s_prel, c->s_n,
                               c->my, st);
    }
    DDN_TRY(fsk4_decode(c, cur, 0, st));
    c->last_set = cur;
    c->step++;
    return DDN_OK;
}

// (internal, the mixed chain) the event the decode stage records once everything that reads the loop's buffers has been queued
extern "C" void*
ddn_fsk4_chain_reads_done_event(ddn_fsk4_chain* c) {
    return c ? (void*)c->ev_reads : nullptr;
}

// (internal, the mixed chain's shared front end) where this call's stage 0 would write its discriminator rows
extern "C" float*
ddn_fsk4_chain_disc_buffer(ddn_fsk4_chain* c) {
    return !c ? nullptr : ((c->d_disc2 && (c->step & 1)) ? c->d_disc2 : c->d_disc);
}

extern "C" int
ddn_fsk4_chain_run(ddn_fsk4_chain* c, const void* d_iq, void* hip_stream) {
    if (!c || !d_iq) {
        return DDN_EINVAL;
    }
    for (int stage = 0; stage < 3; stage++) {
        DDN_TRY(ddn_fsk4_chain_stage(c, stage, d_iq, hip_stream));
    }
    return DDN_OK;
}

// decode what the carry still holds back (end of a stream): one more decode pass without new samples
extern "C" int
ddn_fsk4_chain_flush(ddn_fsk4_chain* c, void* hip_stream) {
    if (!c) {
        return DDN_EINVAL;
    }
    if (c->step == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int cur = (int)(c->step & 1), prev = cur ^ 1;
    HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], 1, c->d_rec[cur], c->d_fl[cur], c->stride, c->T, c->B, st));
    HIP_TRY(hipMemsetAsync(c->d_new[cur], 0, sizeof(int32_t) * (size_t)c->B, st));
    HIP_TRY(hipMemsetAsync(c->s_n, 0, sizeof(int32_t) * (size_t)c->B, st));
    DDN_TRY(fsk4_decode(c, cur, 1, st));
    c->last_set = cur;
    c->step++;
    HIP_TRY(hipStreamSynchronize(st));
    return DDN_OK;
}

extern "C" int
ddn_fsk4_chain_get_results(ddn_fsk4_chain* c, ddn_fsk4_chain_results* r) {
    if (!c || !r) {
        return DDN_EINVAL;
    }
    const int cur = c->last_set;
    memset(r, 0, sizeof(*r));
    r->stride_symbols = c->stride;
    r->carry_symbols = (size_t)c->T;
    r->max_syncs = (size_t)c->myd;
    r->voice_slots = c->vf;
    r->d_records10 = c->d_rec[cur];
    r->d_flags = c->d_fl[cur];
    r->d_payload2 = c->d_pay;
    r->d_new = c->d_new[cur];
    r->d_counts = c->d_cnt_full;
    r->d_n_sync = c->d_ns;
    r->d_dropped_syncs = c->d_dropped;
    r->d_sync_pos = c->d_spos;
    r->d_sync_pat = c->d_spat;
    r->d_pre = c->d_pre;
    r->d_valid = c->d_valid;
    r->d_dmr_slot_type = c->d_st;
    r->d_dmr_slot_type_ok = c->d_st_ok;
    r->d_dmr_pdu96 = c->d_pdu;
    r->d_dmr_bptc_errs = c->d_errs;
    r->d_nxdn_lich = c->d_lich;
    r->d_nxdn_sacch = c->d_sacch;
    r->d_nxdn_sacch_ok = c->d_sacch_ok;
    r->d_nxdn_sacch_hard = c->d_sacch_hard;
    r->d_nxdn_sacch_hard_ok = c->d_sacch_hard_ok;
    r->d_nxdn_facch = c->d_facch;
    r->d_nxdn_facch_ok = c->d_facch_ok;
    if (c->dmr && c->E) {
        r->d_events = c->d_ev;
        r->d_n_events = c->d_nev;
        r->max_events = c->E;
        r->dmr_data_bursts = c->db;
        r->d_dmr_n_data = c->dd_n;
        r->d_dmr_data_start = c->dd_start;
        r->d_dmr_data_slot = c->dd_slot;
        r->d_dmr_data_type = c->dd_type;
        r->d_dmr_data_info196 = c->dd_info;
        r->d_dmr_data_bits96 = c->dd_pdu;
        r->d_dmr_data_bytes12 = c->dd_bytes;
        r->d_dmr_data_errs = c->dd_errs;
        r->d_dmr_data_crc = c->dd_crc;
        r->d_dmr_r34_unconfirmed = c->dd_unconf;
        r->d_dmr_r34_confirmed = c->dd_conf;
        r->d_dmr_r34_confirmed_crc = c->dd_confcrc;
        r->d_dmr_r34_pool = (const ddn_r34_candidate*)c->dd_pool;
        r->d_dmr_r34_pool_n = c->dd_pooln;
        r->dmr_emb_lcs = c->lb;
        r->d_dmr_n_emb = c->de_n;
        r->d_dmr_emb_pos = c->de_pos;
        r->d_dmr_emb_lc77 = c->de_out;
        r->d_dmr_emb_errs = c->de_errs;
        r->d_dmr_emb_ok = c->de_ok;
    }
    if (c->dmr) {
        r->dmr_voice_bursts = c->vb;
        r->d_dmr_voice_start = c->d_vstart;
        r->d_dmr_voice_pre = c->d_vpre;
        r->d_dmr_n_voice = c->d_vnb;
        r->d_dmr_voice_skip = c->d_skip;
        r->d_dmr_ambe_frames = c->d_ambe_fr;
        r->d_dmr_ambe_bits = c->d_ambe_d;
        r->d_dmr_ambe_result = c->d_res_out;
        r->d_dmr_pcm = c->d_pcm;
        r->d_events = c->d_ev;
        r->d_n_events = c->d_nev;
        r->max_events = c->E;
    } else {
        r->d_nxdn_voice_skip = c->d_skip;
        r->d_nxdn_ambe_bits = c->d_ambe_d;
        r->d_nxdn_pcm = c->d_pcm;
    }
    if (c->ysf) {
        r->d_ysf_fich4 = c->y_fich4;
        r->d_ysf_fich_status = c->y_st;
        r->d_ysf_fich_cost = c->y_ve;
        r->d_ysf_info2 = c->y_info;
        r->d_ysf_dch40 = c->y_dch;
        r->d_ysf_dch_status2 = c->y_dst;
        r->d_ysf_dch_cost2 = c->y_dcost;
        r->d_ysf_ambe49x5 = c->y_ambe;
        r->d_ysf_errs2x5 = c->y_errs;
        r->d_ysf_frames184x5 = c->y_fr;
        r->d_ysf_n_frames = c->y_nfr;
        r->ysf_voice_frames = c->mbe ? c->yvf : 0;
        r->d_ysf_n_voice = c->mbe ? c->d_vn : nullptr;
        r->d_ysf_voice_slot = c->mbe ? c->y_vslot : nullptr;
        r->d_ysf_voice_result = c->mbe ? c->d_res_out : nullptr;
        r->d_ysf_pcm = c->mbe ? c->d_pcm : nullptr;
        r->d_ysf_voice_skip = c->mbe ? c->d_skip : nullptr;
        r->d_ysf_imbe_n_voice = c->mbe_i ? c->yi_vn : nullptr;
        r->d_ysf_imbe_voice_slot = c->mbe_i ? c->yi_vslot : nullptr;
        r->d_ysf_imbe_voice_skip = c->mbe_i ? c->yi_skip : nullptr;
        r->d_ysf_imbe_voice_result = c->mbe_i ? c->yi_res_out : nullptr;
        r->d_ysf_imbe_pcm = c->mbe_i ? c->yi_pcm : nullptr;
    }
    if (c->m17) {
        r->d_sync_thr5 = c->d_thr;
        r->d_m17_lsf30 = c->m_lsf;
        r->d_m17_lsf_status = c->m_lsf_st;
        r->d_m17_lsf_cost = c->m_cost;
        r->d_m17_lich6 = c->m_l6;
        r->d_m17_lich_cnt = c->m_cnt;
        r->d_m17_fn_payload18 = c->m_fp;
        r->d_m17_str_status = c->m_st;
        r->d_m17_lich_lsf30 = c->m_ll;
        r->d_m17_lich_status = c->m_ll_st;
    }
    return DDN_OK;
}

extern "C" void*
ddn_fsk4_chain_front_end(ddn_fsk4_chain* c) {
    return c ? c->fe : nullptr;
}
extern "C" void*
ddn_fsk4_chain_rx(ddn_fsk4_chain* c) {
    return c ? c->rx : nullptr;
}

// ---- the three protocol groups of a mixed batch (BASELINE configs[3]) -------------------------------------------------------
struct ddn_mixed_chain {
    ddn_mixed_chain_config cfg;
    ddn_p25_chain* p25;
    ddn_fsk4_chain *dmr, *nxdn;
    hipStream_t st[3], st2[3]; // per group: front end + matched filter + loop / frame FEC + voice
    hipEvent_t ev_front[3], ev_loop[3], ev_dec[3];
    bool have_dec[3];
    // (round 5) front ends on streams of their own into two discriminator buffers per group: call k + 1's front end runs beside call
    // k's loop (ev_read[g][parity]: the loop that read that buffer has ended)
    bool overlap;
    hipStream_t stF[3];
    hipEvent_t ev_read[3][2];
    unsigned long long calls;
    // (round 6) one front-end launch for all the groups (ddn_batch_set_segments): the groups' channels share workgroups of sixteen, so a
    // 4096-channel mixed batch is one round of 256 workgroups instead of three launches of 171 eight-channel ones (two rounds and a half)
    ddn_batch* fe_all;
};
extern "C" int ddn_p25_chain_double_disc(ddn_p25_chain* c);

extern "C" void
ddn_mixed_chain_destroy(ddn_mixed_chain* m) {
    if (!m) {
        return;
    }
    (void)hipDeviceSynchronize();
    ddn_p25_chain_destroy(m->p25);
    ddn_fsk4_chain_destroy(m->dmr);
    ddn_fsk4_chain_destroy(m->nxdn);
    ddn_batch_destroy(m->fe_all);
    for (int k = 0; k < 3; k++) {
        for (hipStream_t s : {m->st[k], m->st2[k], m->stF[k]}) {
            if (s) {
                (void)hipStreamDestroy(s);
            }
        }
        for (hipEvent_t e : {m->ev_front[k], m->ev_loop[k], m->ev_dec[k], m->ev_read[k][0], m->ev_read[k][1]}) {
            if (e) {
                (void)hipEventDestroy(e);
            }
        }
    }
    delete m;
}

extern "C" int
ddn_mixed_chain_create(const ddn_mixed_chain_config* cfg, ddn_mixed_chain** out) {
    if (!cfg || !out || cfg->n_p25 < 0 || cfg->n_dmr < 0 || cfg->n_nxdn48 < 0 || cfg->n_p25 + cfg->n_dmr + cfg->n_nxdn48 <= 0
        || cfg->samples_per_call <= 0 || cfg->block_len <= 0) {
        ddn_set_error("ddn_mixed_chain_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    ddn_mixed_chain* m = new (std::nothrow) ddn_mixed_chain();
    if (!m) {
        return DDN_ENOMEM;
    }
    memset(m, 0, sizeof(*m));
    m->cfg = *cfg;
    int rc = DDN_OK;
    if (cfg->n_p25 > 0) {
        ddn_p25_chain_config pc;
        memset(&pc, 0, sizeof(pc));
        pc.n_channels = cfg->n_p25;
        pc.samples_per_call = cfg->samples_per_call;
        pc.block_len = cfg->block_len;
        pc.input_format = cfg->input_format;
        pc.vocoder = cfg->vocoder;
        rc = ddn_p25_chain_create(&pc, &m->p25);
    }
    if (rc == DDN_OK && cfg->n_dmr > 0) {
        ddn_fsk4_chain_config dc = {cfg->n_dmr, cfg->samples_per_call, cfg->block_len, cfg->input_format, DDN_FSK4_DMR, 2, 0, 1,
                                    cfg->vocoder};
        rc = ddn_fsk4_chain_create(&dc, &m->dmr);
    }
    if (rc == DDN_OK && cfg->n_nxdn48 > 0) {
        ddn_fsk4_chain_config nc = {cfg->n_nxdn48, cfg->samples_per_call, cfg->block_len, cfg->input_format, DDN_FSK4_NXDN48, 0, 0, 1,
                                    cfg->vocoder};
        rc = ddn_fsk4_chain_create(&nc, &m->nxdn);
    }
    { // the three loops share the device: the DMR / NXDN48 kernels take the shape that suits the whole batch
        const int total = cfg->n_p25 + cfg->n_dmr + cfg->n_nxdn48;
        int cpw = 32;
        for (int c = 1; c <= 32; c *= 2) {
            if ((total + c - 1) / c <= 1536) {
                cpw = c;
                break;
            }
        }
        int cpw_d = cpw, cpw_n = cpw;
        // (experiments: values the setters reject are ignored, not passed on)
        auto pow2_1_32 = [](const char* e, int dflt) {
            const int v = e ? atoi(e) : 0;
            return (v >= 1 && v <= 32 && (v & (v - 1)) == 0) ? v : dflt;
        };
        {   // the overlapped schedule runs the fsk4 loops one channel per wavefront where a group allows it (<= 1536 channels: the
            // loop's fastest shape - 2.7 / 3.1 ms alone against 5.2 / 6.8 at four; the two loops then take turns on the device)
            const char* e = DDN_EXP_ENV("DDN_MIX_OVERLAP");
            if (cfg->overlap || (e && e[0] == '1')) {
                cpw_d = cfg->n_dmr <= 1536 ? 1 : cpw_d;
                cpw_n = cfg->n_nxdn48 <= 1536 ? 1 : cpw_n;
            }
        }
        cpw_d = pow2_1_32(DDN_EXP_ENV("DDN_MIX_CPW_DMR"), cpw_d);
        cpw_n = pow2_1_32(DDN_EXP_ENV("DDN_MIX_CPW_NXDN"), cpw_n);
        // Residency decides the step: a CU holds 8 of these wavefronts (~200 registers each).  At 4096 channels in thirds the P25
        // loop's own choice (4 channels per workgroup of 4 waves: 342 workgroups) + 2 x 342 two-wave workgroups are 2736 waves for
        // 2048 places - the loop launched last waits for the first to finish (measured: NXDN48 loop 9 ms, step 15.1 ms).  With 8
        // channels per P25 workgroup it is 2052 waves: step 14.2 ms.
        int cpw_p = (total > 2048 && (m->dmr || m->nxdn)) ? 8 : 0;
        if (const char* e = DDN_EXP_ENV("DDN_MIX_CPW_P25")) {
            const int v = atoi(e);
            if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) {
                cpw_p = v;
            }
        }
        if (rc == DDN_OK && m->p25 && cpw_p) {
            rc = ddn_p25_rx_set_channels_per_wave((ddn_p25_rx*)ddn_p25_chain_rx(m->p25), cpw_p);
        }
        if (rc == DDN_OK && m->dmr) {
            rc = ddn_fsk4_rx_set_channels_per_wave(m->dmr->rx, cpw_d);
        }
        if (rc == DDN_OK && m->nxdn) {
            rc = ddn_fsk4_rx_set_channels_per_wave(m->nxdn->rx, cpw_n);
        }
    }
    {   // cfg.overlap = 1 (off by default): front ends on streams of their own, two discriminator buffers per group - call k + 1's
        // front ends beside call k's loops.  Seven streams: it needs GPU_MAX_HW_QUEUES=6 or 7 in the process environment (HIP's default
        // four hardware queues make streams share queues: 17-18 ms per step; with 6: 11.75 ms against 13.2 - profiles/README.md)
        const char* e = DDN_EXP_ENV("DDN_MIX_OVERLAP");
        m->overlap = cfg->overlap != 0 || (e && e[0] == '1');
    }
    if (rc == DDN_OK && m->overlap) {
        if (m->p25) {
            rc = ddn_p25_chain_double_disc(m->p25);
        }
        for (ddn_fsk4_chain* c : {m->dmr, m->nxdn}) {
            if (rc == DDN_OK && c && !c->d_disc2) {
                if (!dalloc(&c->d_disc2, (size_t)c->B * (size_t)c->n)) { // (zero-filled and padded like every other buffer)
                    rc = DDN_ENOMEM;
                }
            }
        }
        for (int k = 0; k < 3 && rc == DDN_OK; k++) {
            if (hipStreamCreateWithFlags(&m->stF[k], hipStreamNonBlocking) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_read[k][0], hipEventDisableTiming) != hipSuccess
                || hipEventCreateWithFlags(&m->ev_read[k][1], hipEventDisableTiming) != hipSuccess) {
                rc = DDN_EHIP;
            }
        }
    }
    // DDN_MIX_XCD="a,b,c" (experiment): the three groups' loop streams on disjoint sets of XCDs (a + b + c <= 8; CU-mask bit i is
    // CU i / 8 of XCD i % 8) - different loop kernels then never share a CU's instruction cache
    int xcd_n[3] = {0, 0, 0};
    if (const char* e = DDN_EXP_ENV("DDN_MIX_XCD")) {
        if (sscanf(e, "%d,%d,%d", &xcd_n[0], &xcd_n[1], &xcd_n[2]) != 3 || xcd_n[0] < 1 || xcd_n[1] < 1 || xcd_n[2] < 1
            || xcd_n[0] + xcd_n[1] + xcd_n[2] > 8) {
            xcd_n[0] = xcd_n[1] = xcd_n[2] = 0;
        }
    }
    for (int k = 0, x0 = 0; k < 3 && rc == DDN_OK; k++) {
        hipError_t se;
        if (xcd_n[k]) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 256; i++) {
                if (i % 8 >= x0 && i % 8 < x0 + xcd_n[k]) {
                    mask[i / 32] |= 1u << (i % 32);
                }
            }
            x0 += xcd_n[k];
            se = hipExtStreamCreateWithCUMask(&m->st[k], 8, mask);
        } else {
            se = hipStreamCreateWithFlags(&m->st[k], hipStreamNonBlocking);
        }
        if (se != hipSuccess
            || (k == 0 && hipStreamCreateWithFlags(&m->st2[0], hipStreamNonBlocking) != hipSuccess)
            || hipEventCreateWithFlags(&m->ev_front[k], hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&m->ev_loop[k], hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&m->ev_dec[k], hipEventDisableTiming) != hipSuccess) {
            rc = DDN_EHIP;
        }
    }
    if (rc == DDN_OK && !m->overlap && (m->p25 != nullptr) + (m->dmr != nullptr) + (m->nxdn != nullptr) >= 2) {
        // the shared front end: one batch object over all channels, a segment per group present (the profiles the groups' own chain
        // objects design: P25 C4FM / 12.5 kHz / 6.25 kHz - all 135 taps at 48 kHz; a set of profiles with different tap counts
        // keeps the groups' own front ends)
        int32_t cnt[3], prof[3];
        int ns = 0;
        const int gcnt[3] = {cfg->n_p25, cfg->n_dmr, cfg->n_nxdn48}, gprof[3] = {DDN_LPF_P25_C4FM, DDN_LPF_12K5, DDN_LPF_6K25};
        for (int g = 0; g < 3; g++) {
            if (gcnt[g] > 0) {
                cnt[ns] = gcnt[g];
                prof[ns++] = gprof[g];
            }
        }
        ddn_front_end_config fc = {cfg->n_p25 + cfg->n_dmr + cfg->n_nxdn48, 48000, 4800, 4, prof[0], cfg->input_format, cfg->block_len, 0.0f};
        if (!DDN_EXP_ENV("DDN_MIX_OWN_FE") && ddn_batch_create(&fc, &m->fe_all) == DDN_OK) {
            if (ddn_batch_set_segments(m->fe_all, ns, cnt, prof) != DDN_OK || cfg->samples_per_call < DDN_CARRY_LEN) {
                ddn_batch_destroy(m->fe_all);
                m->fe_all = nullptr;
            }
        }
    }
    if (rc != DDN_OK) {
        ddn_mixed_chain_destroy(m);
        return rc;
    }
    *out = m;
    return DDN_OK;
}



extern "C" int
ddn_mixed_chain_run(ddn_mixed_chain* m, const void* d_iq_p25, const void* d_iq_dmr, const void* d_iq_nxdn48) {
    if (!m || (m->p25 && !d_iq_p25) || (m->dmr && !d_iq_dmr) || (m->nxdn && !d_iq_nxdn48)) {
        return DDN_EINVAL;
    }
    // The protocol groups are independent channel sets, two streams each.  Their stages are lined up across the groups: the three
    // front ends first (kernels that would otherwise be starved by - and delay the workgroups of - another group's receive loop),
    // then the three receive loops side by side (latency chains that fit on the device together).  A group's frame FEC / voice stage
    // runs on its second stream behind its loop, so the NEXT call's front end does not queue up behind it (a front end of <= 2048
    // channels is a 3 ms latency chain whatever the batch: what it runs beside costs it little) - the next call's loop waits for it
    // (the loop's sync lists, handler events and payload rows are single buffers the decode stage reads).
    const void* iq[3] = {d_iq_p25, d_iq_dmr, d_iq_nxdn48};
    const bool on[3] = {m->p25 != nullptr, m->dmr != nullptr, m->nxdn != nullptr};
    auto stage = [&](int g, int st_no) -> int {
        // (one decode stream for the three groups: HIP maps streams onto four hardware queues, a fifth stream would share a queue
        // with one of the loops - measured: the NXDN48 loop then ran behind the P25 loop)
        hipStream_t s = st_no == 2 ? m->st2[0] : m->st[g];
        if (g == 0) {
            return ddn_p25_chain_stage(m->p25, st_no, iq[0], s);
        }
        return ddn_fsk4_chain_stage(g == 1 ? m->dmr : m->nxdn, st_no, iq[g], s);
    };
    // (round 5, measured and left off) DDN_MIX_PHASED=1: a call's front ends start when ALL loops of the call before have ended
    // instead of each behind its own group's loop (where it crawls beside the other groups' loops, 4-5 ms).  Lined up, the three
    // front ends take ~3 ms together, but the loops then have nothing beside them either: 14.4-15.4 ms per step against 13.2.
    static const bool phased = [] {
        const char* e = DDN_EXP_ENV("DDN_MIX_PHASED");
        return e && e[0] == '1';
    }();
    // (the discriminator buffer a group's call uses is picked by that chain's own step parity, the event that guards it by m->calls'.
    // A part-level flush through ddn_mixed_chain_part() advances the part's step and shifts the two against each other; it also
    // synchronises everything first, so the call after it has no reader to wait for, and from the call after that the event waited
    // for is that of a LATER loop than the buffer's last reader - an over-wait, never a race)
    const int par = (int)(m->calls & 1);
    if (m->overlap) {
        // (round 5) A group is a chain front end -> matched filter -> loop, and with one discriminator buffer the step could not be
        // shorter than the slowest group's chain (NXDN48: 4.4 + 1.4 + 6.8 ms).  With two buffers and the front ends on streams of
        // their own, call k + 1's front end runs beside call k's loop (the host runs ahead); it waits for the loop that read its
        // buffer two calls ago.  The carried record tails are copied at the head of stage 1, so stage 0 touches nothing a loop writes.
        for (int g = 0; g < 3; g++) {
            if (on[g]) {
                if (m->calls >= 2) {
                    HIP_TRY(hipStreamWaitEvent(m->stF[g], m->ev_read[g][par], 0));
                }
                if (g == 0) {
                    DDN_TRY(ddn_p25_chain_stage(m->p25, 0, iq[0], m->stF[0]));
                } else {
                    DDN_TRY(ddn_fsk4_chain_stage(g == 1 ? m->dmr : m->nxdn, 0, iq[g], m->stF[g]));
                }
                HIP_TRY(hipEventRecord(m->ev_front[g], m->stF[g]));
            }
        }
        for (int g = 0; g < 3; g++) {
            if (!on[g]) {
                continue;
            }
            HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_front[g], 0));
            if (m->have_dec[g]) {
                HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_dec[g], 0));
            }
            DDN_TRY(stage(g, 1));
            HIP_TRY(hipEventRecord(m->ev_loop[g], m->st[g]));
            HIP_TRY(hipEventRecord(m->ev_read[g][par], m->st[g]));
        }
    } else if (m->fe_all) {
        // one front-end launch for every group.  It writes every group's discriminator buffer, so it waits for all the loops of the
        // call before; every group's matched filter + loop then follows it on the group's stream.  It goes on the stream of the
        // LAST group present - the loop that ends last (NXDN48 6 ms, DMR 5, P25 4.8 side by side): queued right behind that loop
        // it is dispatched the moment the loop ends.  On another stream it is released by an event, in a race with that group's
        // decode stage (released by the same event), whose many small workgroups keep taking a little LDS on every CU while a front-end
        // workgroup needs a CU's whole LDS: measured 4.0 ms for the launch instead of 2.2.
        const int g0 = on[2] ? 2 : (on[1] ? 1 : 0);
        hipStream_t sf = m->st[g0];
        for (int g = 0; g < 3; g++) {
            if (on[g] && g != g0 && m->calls > 0) {
                HIP_TRY(hipStreamWaitEvent(sf, m->ev_loop[g], 0));
            }
        }
        const void* in[3];
        float* disc[3];
        int ns = 0;
        if (on[0]) {
            DDN_TRY(ddn_p25_chain_stage0_prepare(m->p25, sf, &disc[ns]));
            in[ns++] = iq[0];
        }
        if (on[1]) {
            disc[ns] = ddn_fsk4_chain_disc_buffer(m->dmr);
            in[ns++] = iq[1];
        }
        if (on[2]) {
            disc[ns] = ddn_fsk4_chain_disc_buffer(m->nxdn);
            in[ns++] = iq[2];
        }
        DDN_TRY(ddn_front_end_run_segments(m->fe_all, in, (size_t)m->cfg.samples_per_call, disc, sf));
        HIP_TRY(hipEventRecord(m->ev_front[g0], sf));
        for (int g = 0; g < 3; g++) {
            if (!on[g]) {
                continue;
            }
            if (g != g0) {
                HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_front[g0], 0));
            }
            if (m->have_dec[g]) {
                // the loop overwrites what the decode stage of the call before reads of it: P25 - the whole stage (its records and
                // events are triple-buffered, but a loop that starts beside LDS-hungry decode kernels is slowed for its whole
                // length); DMR / NXDN48 - the stage's gathers only (their loop's sync lists and events are single buffers), the
                // frame FEC and synthesis behind them run on beside the loop
                hipEvent_t gate = g == 0 ? m->ev_dec[0] : (hipEvent_t)ddn_fsk4_chain_reads_done_event(g == 1 ? m->dmr : m->nxdn);
                HIP_TRY(hipStreamWaitEvent(m->st[g], gate, 0));
            }
            DDN_TRY(stage(g, 1));
            HIP_TRY(hipEventRecord(m->ev_loop[g], m->st[g]));
        }
    } else {
    for (int g = 0; g < 3; g++) {
        if (on[g]) {
            if (phased) {
                for (int h = 0; h < 3; h++) {
                    if (h != g && on[h] && m->have_dec[h]) { // (have_dec: the group's events have been recorded once)
                        HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_loop[h], 0));
                    }
                }
            }
            DDN_TRY(stage(g, 0));
            HIP_TRY(hipEventRecord(m->ev_front[g], m->st[g]));
        }
    }
    for (int g = 0; g < 3; g++) {
        if (!on[g]) {
            continue;
        }
        for (int h = 0; h < 3; h++) {
            if (h != g && on[h]) {
                HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_front[h], 0));
            }
        }
        if (m->have_dec[g]) {
            HIP_TRY(hipStreamWaitEvent(m->st[g], m->ev_dec[g], 0));
        }
        DDN_TRY(stage(g, 1));
        HIP_TRY(hipEventRecord(m->ev_loop[g], m->st[g]));
    }
    }
    m->calls++;
    if (m->fe_all && !m->overlap) {
        // (round 6) With one decode stream for the three groups that stream was the step: its kernels run beside the front end and
        // the loops at a fraction of their speed (k_p25_lsd 1.6 ms for 0.05, k_mbe_synth 2 ms for 0.5), one group after the other -
        // 11.6 of a 12.5 ms step busy, whatever the front end and the loops did.  Now a group's decode stage follows its loop on the
        // loop's own stream (it runs beside the loops that are still going; the group's next loop comes behind the shared front
        // end anyway) - except the last group's, whose stream carries the front end of the next call right behind its loop: its
        // decode goes to the decode stream, beside that front end.
        const int g0 = on[2] ? 2 : (on[1] ? 1 : 0);
        for (int g = 0; g < 3; g++) {
            if (on[g]) {
                hipStream_t sd = g == g0 ? m->st2[0] : m->st[g];
                if (g == g0) {
                    HIP_TRY(hipStreamWaitEvent(sd, m->ev_loop[g], 0));
                }
                DDN_TRY(g == 0 ? ddn_p25_chain_stage(m->p25, 2, iq[0], sd) : ddn_fsk4_chain_stage(g == 1 ? m->dmr : m->nxdn, 2, iq[g], sd));
                HIP_TRY(hipEventRecord(m->ev_dec[g], sd));
                m->have_dec[g] = true;
            }
        }
        return DDN_OK;
    }
    for (int g = 0; g < 3; g++) {
        if (on[g]) {
            HIP_TRY(hipStreamWaitEvent(m->st2[0], m->ev_loop[g], 0));
            DDN_TRY(stage(g, 2));
            HIP_TRY(hipEventRecord(m->ev_dec[g], m->st2[0]));
            m->have_dec[g] = true;
        }
    }
    return DDN_OK;
}

extern "C" int
ddn_mixed_chain_wait(ddn_mixed_chain* m) {
    if (!m) {
        return DDN_EINVAL;
    }
    for (int k = 0; k < 3; k++) {
        if (m->stF[k]) {
            HIP_TRY(hipStreamSynchronize(m->stF[k]));
        }
        HIP_TRY(hipStreamSynchronize(m->st[k]));
        if (m->st2[k]) {
            HIP_TRY(hipStreamSynchronize(m->st2[k]));
        }
    }
    return DDN_OK;
}

extern "C" void*
ddn_mixed_chain_part(ddn_mixed_chain* m, int which) {
    if (!m) {
        return nullptr;
    }
    return which == 0 ? (void*)m->p25 : (which == 1 ? (void*)m->dmr : (which == 2 ? (void*)m->nxdn : nullptr));
}

// Block partition of a mixed batch over the ranks of a node: the global channel index is [P25 | DMR | NXDN48]; rank
// r of `world` owns a contiguous block of it (the first total % world ranks one channel more) and therefore a contiguous range of
// each protocol group.  Pure arithmetic: every rank computes the same table.
extern "C" int
ddn_mixed_partition(int n_p25, int n_dmr, int n_nxdn48, int rank, int world, int32_t first3[3], int32_t count3[3]) {
    if (n_p25 < 0 || n_dmr < 0 || n_nxdn48 < 0 || world <= 0 || rank < 0 || rank >= world || !first3 || !count3) {
        return DDN_EINVAL;
    }
    const long total = (long)n_p25 + n_dmr + n_nxdn48;
    const long base = total / world, extra = total % world;
    const long lo = rank * base + (rank < extra ? rank : extra), hi = lo + base + (rank < extra ? 1 : 0);
    const long start[3] = {0, n_p25, (long)n_p25 + n_dmr}, len[3] = {n_p25, n_dmr, n_nxdn48};
    for (int k = 0; k < 3; k++) {
        const long a = lo > start[k] ? lo : start[k], b = hi < start[k] + len[k] ? hi : start[k] + len[k];
        first3[k] = (int32_t)(b > a ? a - start[k] : 0);
        count3[k] = (int32_t)(b > a ? b - a : 0);
    }
    return DDN_OK;
}
