// ddn_api_chain.cpp - the P25 Phase 1 chain object (include/ddn_chain.h): stage order, buffers, double buffering and the
// streams / events of the pipelined forms, on top of the library's own C-ABI stage calls.  Host-only code.
//
// What it stands in for in a dsd-neo host: the demodulator thread's per-block loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516) and
// processFrame()'s P25p1 branch (src/engine/protocol_dispatch.c:30-44, src/engine/dispatch/dispatch_p25p1.c), B channels wide.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_chain.h"
#include "ddn_device.h"
#include "ddn_hip.h"
#include "ddn_internal.h"
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

struct ddn_p25_chain {
    ddn_p25_chain_config cfg;
    int B, n, T, F, Fv, E, EL;
    int off97[3]; // symbol a TSDU block's decision falls on, counted from the sync's last symbol
    size_t ms, stride, S, V;
    ddn_batch* fe;
    ddn_p25_rx* rx;
    ddn_p25p1_framer* fr;
    ddn_mbe_batch* mbe;
    float* d_disc;
    // receive-loop outputs, two sets: the loop of call k + 1 writes one while call k is decoded out of the other
    uint8_t *d_rec[2], *d_fl[2];
    uint8_t* d_rec2[2] = {nullptr, nullptr}; // host form of the records (run_host with records2), allocated on first use
    float* d_pcm_dense = nullptr;            // dense PCM (run_host with pcm_dense), allocated on first use: [V][160]
    int32_t *d_pcm_slot = nullptr, *d_pcm_bcnt = nullptr, *d_pcm_boff = nullptr, *d_pcm_total = nullptr;
    int32_t *d_new[2], *d_ev[2], *d_nev[2], *d_evd[2];
    // the decisions of the records a row holds (carried + new), by row index: what files NIDs and TSDU blocks by frame
    int32_t *d_evl[2], *d_evdl[2], *d_nevl[2];
    int32_t *d_cnt_scan, *d_cnt_full;
    // decode buffers
    int32_t* d_nid;
    int32_t *d_lists, *d_list_n; // per frame type: the slots holding a frame of it (k_chain_frames), what the decode launches walk
    uint8_t* d_cls; // frame type of every slot (DDN_CLS_*): the per-type decode launches only work on their own frames
    uint8_t *d_tsbk, *d_tsbk_crc;
    uint8_t *d_words[2], *d_wrel, *d_werrs, *d_vldu;
    uint8_t *d_rs_d[2], *d_rs_p[2], *d_rs_st[2];
    uint8_t *d_lsd, *d_lsd_ok;
    int16_t* d_lsd_llr;
    uint8_t *d_hdu_hex, *d_hdu_par, *d_hdu_st, *d_hdu_d, *d_hdu_p, *d_hdu_rs;
    uint8_t *d_td_d, *d_td_p, *d_td_st, *d_td_rd, *d_td_rp, *d_td_rs;
    // data units (DUID 0xC): PF entries per channel and call, PB data blocks each
    int PF, PB;
    int32_t *d_pdu_slot, *d_pdu_info, *d_n_pdu, *d_pdu_metric;
    uint8_t *d_pdu_hdr, *d_pdu_valid, *d_pdu_blocks, *d_pdu_wanted, *d_pdu_cand, *d_pdu_blocks18, *d_pdu_crc9;
    int32_t* d_pdu_cnt;
    int16_t* d_pdu_llr;
    int64_t* d_first;
    int32_t *d_sc, *d_nldu, *d_sc_out, *d_imbe_res, *d_res_out;
    uint8_t *d_imbe_fr, *d_imbe_soft, *d_imbe_fl, *d_imbe_d;
    float* d_pcm;
    // pipelining
    hipStream_t s_main, s_aux, s_copy, s_copy2; // front end + loop | decode | H2D | D2H
    hipStream_t s_voice = nullptr;              // the voice stage of a decode, beside its frame FEC
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_produced[2], ev_consumed[2], ev_in[2], ev_in_free[2], ev_out[2];
    hipEvent_t ev_loop[2] = {nullptr, nullptr}; // the receive loop of that set's call is next on s_main
    // _run_host: the result copies of a call are issued in the NEXT call (or by _wait / _flush), beside that call's receive loop
    ddn_p25_chain_host_out pending_out;
    int have_pending = 0, pending_set = 0;
    void* d_iq[2];
    size_t iq_bytes;
    long step;
    int last_set;
    // the caller's stream of the last _run / _stage call is not kept (the caller may destroy it): an event recorded on it at the end of
    // that call is what _wait and _flush order themselves behind
    hipEvent_t ev_user = nullptr;
    int have_user_stream;
    // stage timing (ddn_p25_chain_set_timing): events at the stage boundaries of the most recent call
    int timing;
    hipEvent_t ev_t[6];
};

template <typename T>
static bool
dalloc(T** p, size_t count) {
    if (hipMalloc((void**)p, count * sizeof(T) + 16) != hipSuccess) {
        return false;
    }
    return hipMemset(*p, 0, count * sizeof(T)) == hipSuccess;
}

extern "C" void
ddn_p25_chain_destroy(ddn_p25_chain* c) {
    if (!c) {
        return;
    }
    (void)hipDeviceSynchronize();
    ddn_batch_destroy(c->fe);
    ddn_p25_rx_destroy(c->rx);
    ddn_p25p1_framer_destroy(c->fr);
    ddn_mbe_batch_destroy(c->mbe);
    void* all[] = {c->d_disc, c->d_rec[0], c->d_rec[1], c->d_rec2[0], c->d_rec2[1], c->d_pcm_dense, c->d_pcm_slot, c->d_pcm_bcnt,
                   c->d_pcm_boff, c->d_pcm_total, c->d_fl[0], c->d_fl[1], c->d_new[0], c->d_new[1], c->d_ev[0], c->d_ev[1],
                   c->d_nev[0], c->d_nev[1], c->d_evd[0], c->d_evd[1], c->d_evl[0], c->d_evl[1], c->d_evdl[0], c->d_evdl[1],
                   c->d_nevl[0], c->d_nevl[1], c->d_cnt_scan, c->d_cnt_full, c->d_nid, c->d_cls, c->d_lists, c->d_list_n, c->d_tsbk, c->d_tsbk_crc, c->d_words[0],
                   c->d_words[1], c->d_wrel, c->d_werrs, c->d_vldu, c->d_rs_d[0], c->d_rs_d[1], c->d_rs_p[0], c->d_rs_p[1],
                   c->d_rs_st[0], c->d_rs_st[1], c->d_lsd, c->d_lsd_ok, c->d_lsd_llr, c->d_hdu_hex, c->d_hdu_par, c->d_hdu_st,
                   c->d_hdu_d, c->d_hdu_p, c->d_hdu_rs, c->d_td_d, c->d_td_p, c->d_td_st, c->d_td_rd, c->d_td_rp, c->d_td_rs,
                   c->d_first, c->d_sc, c->d_nldu, c->d_sc_out, c->d_imbe_res, c->d_res_out, c->d_imbe_fr, c->d_imbe_soft,
                   c->d_imbe_fl, c->d_imbe_d, c->d_pcm, c->d_iq[0], c->d_iq[1], c->d_pdu_slot, c->d_pdu_info, c->d_n_pdu,
                   c->d_pdu_metric, c->d_pdu_hdr, c->d_pdu_valid, c->d_pdu_blocks, c->d_pdu_llr, c->d_pdu_wanted, c->d_pdu_cand,
                   c->d_pdu_blocks18, c->d_pdu_crc9, c->d_pdu_cnt};
    for (void* p : all) {
        (void)hipFree(p);
    }
    if (c->s_main) {
        (void)hipStreamDestroy(c->s_main);
    }
    if (c->s_aux) {
        (void)hipStreamDestroy(c->s_aux);
    }
    if (c->s_copy) {
        (void)hipStreamDestroy(c->s_copy);
    }
    if (c->s_copy2) {
        (void)hipStreamDestroy(c->s_copy2);
    }
    if (c->s_voice) {
        (void)hipStreamDestroy(c->s_voice);
    }
    if (c->ev_fork) {
        (void)hipEventDestroy(c->ev_fork);
    }
    if (c->ev_join) {
        (void)hipEventDestroy(c->ev_join);
    }
    if (c->ev_user) {
        (void)hipEventDestroy(c->ev_user);
    }
    hipEvent_t evs[] = {c->ev_produced[0], c->ev_produced[1], c->ev_consumed[0], c->ev_consumed[1], c->ev_in[0], c->ev_in[1],
                        c->ev_in_free[0], c->ev_in_free[1], c->ev_out[0], c->ev_out[1], c->ev_loop[0], c->ev_loop[1], c->ev_t[0],
                        c->ev_t[1], c->ev_t[2], c->ev_t[3], c->ev_t[4], c->ev_t[5]};
    for (hipEvent_t e : evs) {
        if (e) {
            (void)hipEventDestroy(e);
        }
    }
    delete c;
}

extern "C" int
ddn_p25_chain_create(const ddn_p25_chain_config* cfg, ddn_p25_chain** out) {
    if (!cfg || !out || cfg->n_channels <= 0 || cfg->samples_per_call <= 0 || cfg->block_len <= 0) {
        ddn_set_error("ddn_p25_chain_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    ddn_p25_chain* c = new (std::nothrow) ddn_p25_chain();
    if (!c) {
        return DDN_ENOMEM;
    }
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->B = cfg->n_channels;
    c->n = cfg->samples_per_call;
    c->T = cfg->carry_symbols > 0 ? cfg->carry_symbols : 896;
    c->F = cfg->max_frames > 0 ? cfg->max_frames : cfg->samples_per_call / 1800 + 6;
    c->Fv = cfg->max_ldu > 0 ? cfg->max_ldu : cfg->samples_per_call / 8640 + 3;
    c->E = cfg->max_events > 0 ? cfg->max_events : 4 * c->F;
    c->EL = c->E + 64; // + the decisions inside a carried tail
    c->PF = 2;         // data units per channel and call (a second's worth of calls rarely holds one)
    // data blocks per unit: a sync decoded in this call is only guaranteed T symbols behind it, and data block b ends
    // (56 + 98 b + 97) dibits + one status symbol per 35 - 23 symbols behind its sync: 839 for b = 7, 940 for b = 8.  With the default
    // T = 896 the eighth block of a unit whose sync falls late in the scan range would lie beyond the call's records, so the
    // default reads seven blocks per unit (a longer unit is flagged 8, as before); a carry of 941+ symbols reads eight.
    c->PB = c->T >= 941 ? 8 : 7;
    int rc = DDN_OK;
    do {
        ddn_front_end_config fc = {c->B, 48000, 4800, 4, DDN_LPF_P25_C4FM, cfg->input_format, cfg->block_len, 0.0f};
        if ((rc = ddn_batch_create(&fc, &c->fe)) != DDN_OK) {
            break;
        }
        ddn_p25_rx_config rc_cfg = {c->B, 48000, 4800, 0, 1};
        if ((rc = ddn_p25_rx_create(&rc_cfg, &c->rx)) != DDN_OK || (rc = ddn_p25_rx_set_handlers(c->rx, 1, 64)) != DDN_OK) {
            break;
        }
        if ((rc = ddn_p25p1_framer_create(c->B, c->F, &c->fr)) != DDN_OK) {
            break;
        }
        // a TSDU block's decision falls on the last of the 101 symbols the handler reads for it (98 data dibits + the status
        // symbols among them; the third block's 101st symbol is a status symbol): 33 NID symbols + 101 per block after the sync
        for (int b = 0; b < 3; b++) {
            c->off97[b] = 33 + 101 * (b + 1);
        }
        if ((rc = ddn_mbe_batch_create(DDN_MBE_IMBE_7200X4400, c->B, &c->mbe)) != DDN_OK
            || (rc = ddn_mbe_batch_set_p25p1_tail_rule(c->mbe, 1)) != DDN_OK) {
            break;
        }
        c->ms = ddn_p25_rx_max_symbols(c->rx, (size_t)c->n);
        c->stride = (size_t)c->T + c->ms;
        c->S = (size_t)c->B * (size_t)c->F;
        c->V = (size_t)c->B * (size_t)c->Fv * 9;
        const size_t B = (size_t)c->B, S = c->S, V = c->V;
        bool ok = dalloc(&c->d_disc, B * (size_t)c->n);
        for (int k = 0; k < 2 && ok; k++) {
            ok = dalloc(&c->d_rec[k], B * c->stride * 10) && dalloc(&c->d_fl[k], B * c->stride) && dalloc(&c->d_new[k], B)
                 && dalloc(&c->d_ev[k], B * (size_t)c->E * 4) && dalloc(&c->d_nev[k], B) && dalloc(&c->d_evd[k], B * (size_t)c->E * 4)
                 && dalloc(&c->d_evl[k], B * (size_t)c->EL * 4) && dalloc(&c->d_evdl[k], B * (size_t)c->EL * 4) && dalloc(&c->d_nevl[k], B);
        }
        ok = ok && dalloc(&c->d_cnt_scan, B) && dalloc(&c->d_cnt_full, B) && dalloc(&c->d_nid, S * 4) && dalloc(&c->d_cls, S) && dalloc(&c->d_lists, S * DDN_LIST_COUNT) && dalloc(&c->d_list_n, 8)
             && dalloc(&c->d_tsbk, 3 * S * 12) && dalloc(&c->d_tsbk_crc, 3 * S) && dalloc(&c->d_words[0], S * 240)
             && dalloc(&c->d_words[1], S * 240) && dalloc(&c->d_wrel, S * 240) && dalloc(&c->d_werrs, S * 24) && dalloc(&c->d_vldu, S)
             && dalloc(&c->d_rs_d[0], S * 72) && dalloc(&c->d_rs_d[1], S * 96) && dalloc(&c->d_rs_p[0], S * 72)
             && dalloc(&c->d_rs_p[1], S * 48) && dalloc(&c->d_rs_st[0], S) && dalloc(&c->d_rs_st[1], S) && dalloc(&c->d_lsd, S * 32)
             && dalloc(&c->d_lsd_ok, S * 2) && dalloc(&c->d_lsd_llr, S * 32) && dalloc(&c->d_hdu_hex, S * 216)
             && dalloc(&c->d_hdu_par, S * 432) && dalloc(&c->d_hdu_st, S * 36) && dalloc(&c->d_hdu_d, S * 120)
             && dalloc(&c->d_hdu_p, S * 96) && dalloc(&c->d_hdu_rs, S) && dalloc(&c->d_td_d, S * 144) && dalloc(&c->d_td_p, S * 144)
             && dalloc(&c->d_td_st, S * 12) && dalloc(&c->d_td_rd, S * 72) && dalloc(&c->d_td_rp, S * 72) && dalloc(&c->d_td_rs, S)
             && dalloc(&c->d_first, V) && dalloc(&c->d_sc, V) && dalloc(&c->d_nldu, B) && dalloc(&c->d_sc_out, V)
             && dalloc(&c->d_imbe_res, V * 5) && dalloc(&c->d_res_out, V * 5) && dalloc(&c->d_imbe_fr, V * 184)
             && dalloc(&c->d_imbe_soft, V * 368) && dalloc(&c->d_imbe_fl, V) && dalloc(&c->d_imbe_d, V * 88)
             && dalloc(&c->d_pcm, V * 160) && dalloc(&c->d_pdu_slot, B * (size_t)c->PF) && dalloc(&c->d_pdu_info, B * (size_t)c->PF * 4)
             && dalloc(&c->d_n_pdu, B) && dalloc(&c->d_pdu_hdr, B * (size_t)c->PF * 12)
             && dalloc(&c->d_pdu_valid, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_blocks, B * (size_t)c->PF * (size_t)c->PB * 12)
             && dalloc(&c->d_pdu_metric, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_llr, B * (size_t)c->PF * (size_t)c->PB * 196)
             && dalloc(&c->d_pdu_wanted, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_cand, B * (size_t)c->PF * (size_t)c->PB * 8 * 24)
             && dalloc(&c->d_pdu_blocks18, B * (size_t)c->PF * (size_t)c->PB * 18) && dalloc(&c->d_pdu_crc9, B * (size_t)c->PF * (size_t)c->PB)
             && dalloc(&c->d_pdu_cnt, B * (size_t)c->PF * (size_t)c->PB);
        if (!ok) {
            ddn_set_error("ddn_p25_chain_create: device allocation failed");
            rc = DDN_ENOMEM;
            break;
        }
        if (hipStreamCreateWithPriority(&c->s_main, hipStreamNonBlocking, -1) != hipSuccess
            || hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking) != hipSuccess
            || hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking) != hipSuccess
            || hipStreamCreateWithFlags(&c->s_copy2, hipStreamNonBlocking) != hipSuccess
            || hipStreamCreateWithFlags(&c->s_voice, hipStreamNonBlocking) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&c->ev_user, hipEventDisableTiming) != hipSuccess) {
            rc = DDN_EHIP;
            break;
        }
        hipEvent_t* evs[] = {&c->ev_produced[0], &c->ev_produced[1], &c->ev_consumed[0], &c->ev_consumed[1], &c->ev_in[0],
                             &c->ev_in[1], &c->ev_in_free[0], &c->ev_in_free[1], &c->ev_out[0], &c->ev_out[1], &c->ev_loop[0],
                             &c->ev_loop[1]};
        for (hipEvent_t* e : evs) {
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
                rc = DDN_EHIP;
            }
        }
        for (hipEvent_t& e : c->ev_t) {
            if (hipEventCreate(&e) != hipSuccess) {
                rc = DDN_EHIP;
            }
        }
        c->iq_bytes = B * (size_t)c->n * (cfg->input_format == DDN_IN_CF32 ? 8 : 2);
    } while (0);
    if (rc != DDN_OK) {
        ddn_p25_chain_destroy(c);
        return rc;
    }
    *out = c;
    return DDN_OK;
}

// carry + front end of one call into buffer set `cur` on stream st
static int
chain_front(ddn_p25_chain* c, const void* d_iq, int cur, hipStream_t st) {
    const int prev = cur ^ 1;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[0], st));
    }
    HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_rec[cur], c->d_fl[cur],
                                c->stride, c->T, c->B, st));
    DDN_TRY(ddn_front_end_run(c->fe, d_iq, (size_t)c->n, c->d_disc, st));
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[1], st));
    }
    return DDN_OK;
}

// the receive loop of that call
static int
chain_loop(ddn_p25_chain* c, int cur, hipStream_t st) {
    DDN_TRY(ddn_p25_rx_set_events(c->rx, c->d_ev[cur], c->d_nev[cur], (size_t)c->E));
    DDN_TRY(ddn_p25_rx_set_event_data(c->rx, c->d_evd[cur]));
    // the loop writes its records behind the T carried ones: row pointer + T records, row stride unchanged
    DDN_TRY(ddn_p25_rx_run(c->rx, c->d_disc, (size_t)c->n, c->d_rec[cur] + (size_t)c->T * 10, c->d_fl[cur] + c->T, c->d_new[cur],
                           c->stride, st));
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[2], st));
    }
    return DDN_OK;
}

// front end + receive loop of one call into buffer set `cur` on stream st (the carried tail is copied in first)
static int
chain_receive(ddn_p25_chain* c, const void* d_iq, int cur, hipStream_t st, hipEvent_t before_loop = nullptr,
              hipEvent_t loop_next = nullptr) {
    DDN_TRY(chain_front(c, d_iq, cur, st));
    // The receive loop fills the device on its own (two workgroups per CU take its registers and LDS) and every workgroup runs
    // for the whole launch: one that has to wait for a CU another kernel still holds makes the launch half as long again.  In the
    // pipelined forms the loop therefore starts once the previous call's decode has drained; that decode overlaps this call's
    // carry, front end and matched filter instead.
    if (before_loop) {
        HIP_TRY(hipStreamWaitEvent(st, before_loop, 0));
    }
    if (loop_next) { // recorded inside ddn_p25_rx_run, after the matched filter
        DDN_TRY(ddn_p25_rx_mark_loop_start(c->rx, loop_next));
    }
    return chain_loop(c, cur, st);
}

// the device -> pinned-host copies of the results of the call that used buffer set `set`, on the second copy stream
static int
chain_copy_out(ddn_p25_chain* c, const ddn_p25_chain_host_out* out, int set) {
    const size_t B = (size_t)c->B, S = c->S, V = c->V;
    const bool dense_pcm = out->pcm_dense && out->pcm_slot && out->pcm_count && out->pcm_dense_frames > 0 && c->cfg.vocoder;
    if (out->records10) {
        HIP_TRY(hipMemcpyAsync(out->records10, c->d_rec[set], B * c->stride * 10, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->flags) {
        HIP_TRY(hipMemcpyAsync(out->flags, c->d_fl[set], B * c->stride, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->records2) {
        HIP_TRY(hipMemcpyAsync(out->records2, c->d_rec2[set], B * c->stride * 2, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->counts) {
        HIP_TRY(hipMemcpyAsync(out->counts, c->d_cnt_full, B * 4, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->events) {
        HIP_TRY(hipMemcpyAsync(out->events, c->d_ev[set], B * (size_t)c->E * 16, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->event_data) {
        HIP_TRY(hipMemcpyAsync(out->event_data, c->d_evd[set], B * (size_t)c->E * 16, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->n_events) {
        HIP_TRY(hipMemcpyAsync(out->n_events, c->d_nev[set], B * 4, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->nid4) {
        HIP_TRY(hipMemcpyAsync(out->nid4, c->d_nid, S * 16, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->tsbk) {
        HIP_TRY(hipMemcpyAsync(out->tsbk, c->d_tsbk, 3 * S * 12, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (out->pcm && c->cfg.vocoder) {
        HIP_TRY(hipMemcpyAsync(out->pcm, c->d_pcm, V * 160 * 4, hipMemcpyDeviceToHost, c->s_copy2));
    }
    if (dense_pcm) {
        const size_t nf = (size_t)out->pcm_dense_frames < V ? (size_t)out->pcm_dense_frames : V;
        HIP_TRY(hipMemcpyAsync(out->pcm_dense, c->d_pcm_dense, nf * 160 * 4, hipMemcpyDeviceToHost, c->s_copy2));
        HIP_TRY(hipMemcpyAsync(out->pcm_slot, c->d_pcm_slot, nf * 4, hipMemcpyDeviceToHost, c->s_copy2));
        HIP_TRY(hipMemcpyAsync(out->pcm_count, c->d_pcm_total, 4, hipMemcpyDeviceToHost, c->s_copy2));
    }
    return DDN_OK;
}

// _run_host defers the result copies of a call to the next call, where they run beside that call's receive loop (the loop is a
// latency chain that leaves the copy engines' shader waves room; beside the front end or the decode stage the copies and the
// kernels slow each other).  Whoever needs the results earlier - _wait, _flush - issues them here.
static int
chain_issue_pending(ddn_p25_chain* c, hipEvent_t beside) {
    if (!c->have_pending) {
        return DDN_OK;
    }
    const int set = c->pending_set;
    HIP_TRY(hipStreamWaitEvent(c->s_copy2, c->ev_consumed[set], 0)); // that call's decode (and its pack kernels) are done
    if (beside) {
        HIP_TRY(hipStreamWaitEvent(c->s_copy2, beside, 0));
    }
    DDN_TRY(chain_copy_out(c, &c->pending_out, set));
    HIP_TRY(hipEventRecord(c->ev_out[set], c->s_copy2));
    c->have_pending = 0;
    return DDN_OK;
}

// A device-form call (_run, _run_pipelined, _stage) after a _run_host call whose results have not left yet: the decode buffers are
// single (d_nid, d_tsbk, d_cnt_full, d_pcm ...), so that call's result copies are issued now and stream `st` - the one the next
// decode will run on - waits for them.
static int
chain_settle_pending(ddn_p25_chain* c, hipStream_t st) {
    if (!c->have_pending) {
        return DDN_OK;
    }
    const int set = c->pending_set;
    DDN_TRY(chain_issue_pending(c, nullptr));
    HIP_TRY(hipStreamWaitEvent(st, c->ev_out[set], 0));
    return DDN_OK;
}

// framer + every frame type's FEC + voice of buffer set `cur` on stream st
static int
chain_decode(ddn_p25_chain* c, int cur, int flush, hipStream_t st) {
    const size_t S = c->S, V = c->V, stride = c->stride;
    const uint8_t* rec = c->d_rec[cur];
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[3], st));
    }
    HIP_TRY(ddn_dev_chain_counts(c->d_new[cur], c->T, c->B, flush, c->d_cnt_scan, c->d_cnt_full, st));
    DDN_TRY(ddn_p25p1_framer_index(c->fr, c->d_fl[cur], c->d_cnt_scan, stride, st));
    // the NID and the TSDU blocks of every frame were decoded inside the loop by its handlers (p25p1_nid_decode,
    // tsbk_decode_repetition_bytes: ddn_p25_rx_set_event_data); they are filed by frame here, not decoded a second time
    {
        const int prev = cur ^ 1;
        HIP_TRY(ddn_dev_chain_events(c->d_evl[prev], c->d_evdl[prev], c->d_nevl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_ev[cur],
                                     c->d_evd[cur], c->d_nev[cur], c->E, c->EL, c->T, c->B, c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur],
                                     st));
        const int32_t *d_ns = nullptr, *d_sp = nullptr;
        DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &d_ns, &d_sp));
        HIP_TRY(hipMemsetAsync(c->d_list_n, 0, sizeof(int32_t) * 8, st));
        HIP_TRY(ddn_dev_chain_frames(c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur], c->EL, d_sp, d_ns, c->B, c->F, c->off97[0],
                                     c->off97[1], c->off97[2], c->d_nid, c->d_tsbk, c->d_tsbk_crc, c->d_cls, c->d_lists, c->d_list_n,
                                     st));
    }
    // data units (DUID 0xC): the header the loop decoded + the data blocks behind it (half-rate trellis, best path) + CRC32
    {
        const int32_t *d_ns = nullptr, *d_sp = nullptr;
        DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &d_ns, &d_sp));
        const size_t NE = (size_t)c->B * (size_t)c->PF, NB = NE * (size_t)c->PB;
        HIP_TRY(ddn_dev_chain_pdu_index(c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur], c->EL, d_sp, d_ns, c->d_nid, c->B, c->F, c->off97[0],
                                        c->PF, c->d_pdu_slot, c->d_pdu_hdr, c->d_pdu_info, c->d_n_pdu, st));
        HIP_TRY(ddn_dev_chain_pdu_gather(rec, c->d_cnt_full, stride, d_sp, c->d_pdu_slot, c->d_pdu_info, c->B, c->F, c->PF, c->PB,
                                         c->d_pdu_llr, c->d_pdu_valid, st));
        // (d_pdu_cand: scratch for the 1/2-rate candidates first, then the rate 3/4 ones - same stream)
        // (only the blocks that lie inside the call's records: groups of 32 without one leave at once)
        HIP_TRY(ddn_dev_p25_half_rate_list_wanted(c->d_pdu_llr, (int)NB, 8, c->d_pdu_valid, (uint32_t*)c->d_pdu_cand, c->d_pdu_cnt, st));
        HIP_TRY(ddn_dev_chain_pdu_take_first(c->d_pdu_cand, c->d_pdu_cnt, (int)NB, c->d_pdu_blocks, c->d_pdu_metric, st));
        // confirmed data: the same blocks through the rate 3/4 LLR list decoder, first candidate with a good CRC9 (:219-241)
        HIP_TRY(ddn_dev_chain_pdu_r34_wanted(c->d_pdu_slot, c->d_pdu_hdr, c->d_pdu_info, c->d_pdu_valid, (int)NB, c->PB, c->d_pdu_wanted, st));
        DDN_TRY(ddn_fec_p25_mbf34_list_batch(c->d_pdu_llr, NB, 8, c->d_pdu_wanted, (ddn_p25_mbf34_candidate*)c->d_pdu_cand, c->d_pdu_cnt, st));
        HIP_TRY(ddn_dev_chain_pdu_r34_select(c->d_pdu_cand, c->d_pdu_cnt, c->d_pdu_wanted, (int)NB, c->d_pdu_blocks18, c->d_pdu_crc9, st));
        HIP_TRY(ddn_dev_chain_pdu_finish(c->d_pdu_slot, c->d_pdu_blocks, c->d_pdu_valid, c->d_pdu_blocks18, (int)NE, c->PB, c->d_pdu_hdr,
                                         c->d_pdu_info, st));
    }
    // The frame FEC below (per-type work lists) and the voice stage (voice index by NID -> IMBE frames -> PCM) read the same records
    // and NIDs and write nothing the other reads: the voice stage runs on a stream of its own beside the FEC (0.7 ms of small
    // kernels) and joins this one at the end - in the pipelined forms (decode on the object's second stream); the one-stream form
    // and stage timing keep everything on the caller's stream.
    hipStream_t vst = st;
    if (!c->timing && st == c->s_aux) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(c->s_voice, c->ev_fork, 0));
        vst = c->s_voice;
    }
    // Every decode below walks the work list of its frame type (the slots whose NID names it, k_chain_frames): a slot's LDU / HDU /
    // TDULC outputs are meaningful for that type alone.  The selection is cleared on every way out.
    struct SelGuard {
        ~SelGuard() { ddn_sel_clear(); }
    } sel_guard;
    // LDU1 / LDU2: Hamming words -> Reed-Solomon; low speed data
    for (int i = 0; i < 2; i++) {
        const int ldu = i + 1;
        ddn_sel_set(c->d_lists + (size_t)i * S, c->d_list_n + i);
        DDN_TRY(ddn_p25p1_framer_gather_ldu_words(c->fr, ldu, rec, c->d_cnt_full, stride, c->d_words[i], c->d_wrel, c->d_vldu, st));
        DDN_TRY(ddn_fec_hamming_10_6_3_batch(c->d_words[i], S * 24, c->d_werrs, st));
        DDN_TRY(ddn_p25p1_framer_pack_ldu_rs(c->fr, ldu, c->d_words[i], c->d_rs_d[i], c->d_rs_p[i], st));
        DDN_TRY(ddn_fec_p25_rs_batch(i == 0 ? DDN_RS_24_12_13 : DDN_RS_24_16_9, c->d_rs_d[i], c->d_rs_p[i], S, c->d_rs_st[i], st));
    }
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_LSD * S, c->d_list_n + DDN_LIST_LSD);
    DDN_TRY(ddn_p25p1_framer_gather_lsd(c->fr, rec, c->d_cnt_full, stride, c->d_lsd, c->d_lsd_llr, c->d_vldu, st));
    DDN_TRY(ddn_fec_p25_lsd_batch(c->d_lsd, c->d_lsd_llr, S * 2, c->d_lsd_ok, st));
    // HDU: 36 Golay(24,6) words -> RS(36,20,17)
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_HDU * S, c->d_list_n + DDN_LIST_HDU);
    DDN_TRY(ddn_p25p1_framer_gather_hdu(c->fr, rec, c->d_cnt_full, stride, c->d_hdu_hex, c->d_hdu_par, nullptr, nullptr, c->d_vldu, st));
    DDN_TRY(ddn_fec_golay24_batch(6, c->d_hdu_hex, c->d_hdu_par, S * 36, c->d_hdu_st, nullptr, st));
    DDN_TRY(ddn_p25p1_framer_pack_hdu_rs(c->fr, c->d_hdu_hex, c->d_hdu_d, c->d_hdu_p, st));
    DDN_TRY(ddn_fec_p25_rs_batch(DDN_RS_36_20_17, c->d_hdu_d, c->d_hdu_p, S, c->d_hdu_rs, st));
    // TDULC: 12 Golay(24,12) words -> RS(24,12,13)
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_TDULC * S, c->d_list_n + DDN_LIST_TDULC);
    DDN_TRY(ddn_p25p1_framer_gather_tdulc(c->fr, rec, c->d_cnt_full, stride, c->d_td_d, c->d_td_p, nullptr, nullptr, c->d_vldu, st));
    DDN_TRY(ddn_fec_golay24_batch(12, c->d_td_d, c->d_td_p, S * 12, c->d_td_st, nullptr, st));
    DDN_TRY(ddn_p25p1_framer_pack_tdulc_rs(c->fr, c->d_td_d, c->d_td_rd, c->d_td_rp, st));
    DDN_TRY(ddn_fec_p25_rs_batch(DDN_RS_24_12_13, c->d_td_rd, c->d_td_rp, S, c->d_td_rs, st));
    ddn_sel_clear();
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[4], st));
    }
    // voice: nine IMBE frames per LDU
    DDN_TRY(ddn_p25p1_framer_voice_index(c->fr, c->d_nid, c->d_cnt_full, c->Fv, stride, c->d_first, c->d_sc, c->d_nldu, vst));
    DDN_TRY(ddn_p25p1_imbe_deinterleave_batch(rec, (size_t)c->B * stride, c->d_first, c->d_sc, V, c->d_imbe_fr, c->d_imbe_soft,
                                              c->d_imbe_fl, c->d_sc_out, vst));
    DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_IMBE_7200X4400, c->d_imbe_fr, nullptr, V, c->d_imbe_d, c->d_imbe_res, vst));
    DDN_TRY(ddn_mbe_result_skip_batch(c->d_imbe_fl, V, c->d_imbe_res, vst));
    if (c->cfg.vocoder) {
        DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_imbe_d, c->d_imbe_res, (size_t)c->Fv * 9, c->d_pcm, c->d_res_out, vst));
    }
    if (vst != st) {
        HIP_TRY(hipEventRecord(c->ev_join, vst));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[5], st));
    }
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_set_timing(ddn_p25_chain* c, int enable) {
    if (!c) {
        return DDN_EINVAL;
    }
    c->timing = enable ? 1 : 0;
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_get_stage_ms(ddn_p25_chain* c, float out4[4]) {
    if (!c || !out4) {
        return DDN_EINVAL;
    }
    DDN_TRY(ddn_p25_chain_wait(c));
    HIP_TRY(hipEventSynchronize(c->ev_t[5]));
    HIP_TRY(hipEventElapsedTime(&out4[0], c->ev_t[0], c->ev_t[1]));
    HIP_TRY(hipEventElapsedTime(&out4[1], c->ev_t[1], c->ev_t[2]));
    HIP_TRY(hipEventElapsedTime(&out4[2], c->ev_t[3], c->ev_t[4]));
    HIP_TRY(hipEventElapsedTime(&out4[3], c->ev_t[4], c->ev_t[5]));
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_run(ddn_p25_chain* c, const void* d_iq, void* hip_stream) {
    if (!c || !d_iq) {
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    DDN_TRY(chain_settle_pending(c, st));
    const int cur = (int)(c->step & 1);
    DDN_TRY(chain_receive(c, d_iq, cur, st));
    DDN_TRY(chain_decode(c, cur, 0, st));
    HIP_TRY(hipEventRecord(c->ev_user, st));
    c->have_user_stream = 1;
    c->last_set = cur;
    c->step++;
    return DDN_OK;
}

// the three stages of ddn_p25_chain_run as separate calls (the mixed-protocol object interleaves them across its groups)
extern "C" int
ddn_p25_chain_stage(ddn_p25_chain* c, int stage, const void* d_iq, void* hip_stream) {
    if (!c || stage < 0 || stage > 2 || (stage == 0 && !d_iq)) {
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int cur = (int)(c->step & 1);
    int rc = DDN_OK;
    if (stage == 0) {
        DDN_TRY(chain_settle_pending(c, st));
        rc = chain_front(c, d_iq, cur, st);
    } else if (stage == 1) {
        rc = chain_loop(c, cur, st);
    } else {
        DDN_TRY(chain_settle_pending(c, st)); // (a no-op after a stage 0 of the same step)
        rc = chain_decode(c, cur, 0, st);
        if (rc == DDN_OK) {
            c->last_set = cur;
            c->step++;
        }
    }
    if (rc == DDN_OK) {
        HIP_TRY(hipEventRecord(c->ev_user, st));
        c->have_user_stream = 1;
    }
    return rc;
}

extern "C" int
ddn_p25_chain_run_pipelined(ddn_p25_chain* c, const void* d_iq) {
    if (!c || !d_iq) {
        return DDN_EINVAL;
    }
    DDN_TRY(chain_settle_pending(c, c->s_aux));
    const int cur = (int)(c->step & 1);
    if (c->step >= 2) { // the decode of call k - 2 has read this set (and call k - 1's decode has read its carried tail source)
        HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_consumed[cur], 0));
    }
    DDN_TRY(chain_receive(c, d_iq, cur, c->s_main, c->step >= 1 ? c->ev_consumed[cur ^ 1] : nullptr));
    HIP_TRY(hipEventRecord(c->ev_produced[cur], c->s_main));
    HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_produced[cur], 0));
    DDN_TRY(chain_decode(c, cur, 0, c->s_aux));
    HIP_TRY(hipEventRecord(c->ev_consumed[cur], c->s_aux));
    c->last_set = cur;
    c->step++;
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_run_host(ddn_p25_chain* c, const void* h_iq, const ddn_p25_chain_host_out* out) {
    if (!c || !h_iq) {
        return DDN_EINVAL;
    }
    const int cur = (int)(c->step & 1);
    if (!c->d_iq[0]) {
        HIP_TRY(hipMalloc(&c->d_iq[0], c->iq_bytes + 16));
        HIP_TRY(hipMalloc(&c->d_iq[1], c->iq_bytes + 16));
    }
    if (c->step >= 2) {
        HIP_TRY(hipStreamWaitEvent(c->s_copy, c->ev_in_free[cur], 0)); // the front end of call k - 2 has read this input buffer
    }
    HIP_TRY(hipMemcpyAsync(c->d_iq[cur], h_iq, c->iq_bytes, hipMemcpyHostToDevice, c->s_copy));
    HIP_TRY(hipEventRecord(c->ev_in[cur], c->s_copy));
    // The header's contract, kept on the host side (stream-to-stream waits alone do not), once this call's input copy is queued and
    // before anything below records ev_out[cur] anew: the previous call's h_iq has left the host before this call returns, and the
    // results of the call before that are in the caller's buffers.  This also bounds what a host that never calls _wait can have
    // queued: two calls.
    if (c->step >= 1) {
        HIP_TRY(hipEventSynchronize(c->ev_in[cur ^ 1]));
    }
    if (c->step >= 2) {
        HIP_TRY(hipEventSynchronize(c->ev_out[cur]));
    }
    HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_in[cur], 0));
    if (c->step >= 2) {
        HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_consumed[cur], 0)); // call k - 2 decoded out of this set ...
        HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_out[cur], 0));      // ... and its results have left it
    }
    DDN_TRY(chain_receive(c, c->d_iq[cur], cur, c->s_main, c->step >= 1 ? c->ev_consumed[cur ^ 1] : nullptr, c->ev_loop[cur]));
    HIP_TRY(hipEventRecord(c->ev_in_free[cur], c->s_main));
    HIP_TRY(hipEventRecord(c->ev_produced[cur], c->s_main));
    // the previous call's results leave now, beside this call's receive loop
    DDN_TRY(chain_issue_pending(c, c->ev_loop[cur]));
    HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_produced[cur], 0));
    if (c->step >= 1) {
        HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_out[cur ^ 1], 0)); // the previous call's results have left the decode buffers
    }
    DDN_TRY(chain_decode(c, cur, 0, c->s_aux));
    if (out && out->records2) { // the records' host form, packed beside the decode stage
        if (!c->d_rec2[0]) {
            HIP_TRY(hipMalloc(&c->d_rec2[0], (size_t)c->B * c->stride * 2));
            HIP_TRY(hipMalloc(&c->d_rec2[1], (size_t)c->B * c->stride * 2));
        }
        HIP_TRY(ddn_dev_chain_pack2(c->d_rec[cur], c->d_fl[cur], (size_t)c->B * c->stride, c->d_rec2[cur], c->s_aux));
    }
    const bool dense_pcm = out && out->pcm_dense && out->pcm_slot && out->pcm_count && out->pcm_dense_frames > 0 && c->cfg.vocoder;
    if (dense_pcm) { // the synthesized frames only, compacted beside the decode stage
        const size_t V = c->V;
        if (!c->d_pcm_dense) {
            HIP_TRY(hipMalloc(&c->d_pcm_dense, V * 160 * sizeof(float)));
            HIP_TRY(hipMalloc(&c->d_pcm_slot, V * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&c->d_pcm_bcnt, ((V + 1023) / 1024) * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&c->d_pcm_boff, ((V + 1023) / 1024) * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&c->d_pcm_total, sizeof(int32_t)));
        }
        HIP_TRY(ddn_dev_chain_pcm_compact(c->d_imbe_res, c->d_pcm, (int)V, (long)V, c->d_pcm_bcnt, c->d_pcm_boff, c->d_pcm_dense,
                                          c->d_pcm_slot, c->d_pcm_total, c->s_aux));
    }
    HIP_TRY(hipEventRecord(c->ev_consumed[cur], c->s_aux));
    if (out) { // this call's results: copied out beside the next call's loop, or when _wait / _flush asks
        c->pending_out = *out;
        c->pending_set = cur;
        c->have_pending = 1;
    } else {
        HIP_TRY(hipStreamWaitEvent(c->s_copy2, c->ev_consumed[cur], 0));
        HIP_TRY(hipEventRecord(c->ev_out[cur], c->s_copy2));
    }
    c->last_set = cur;
    c->step++;
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_flush(ddn_p25_chain* c) {
    if (!c) {
        return DDN_EINVAL;
    }
    if (c->step == 0) {
        return DDN_OK;
    }
    DDN_TRY(ddn_p25_chain_wait(c));
    // a call without new samples: the held-back tail moves to the front of the other set and every sync in it is decoded
    const int cur = (int)(c->step & 1), prev = cur ^ 1;
    HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], 1, c->d_rec[cur], c->d_fl[cur], c->stride, c->T, c->B,
                                c->s_aux));
    HIP_TRY(hipMemsetAsync(c->d_new[cur], 0, sizeof(int32_t) * (size_t)c->B, c->s_aux));
    HIP_TRY(hipMemsetAsync(c->d_nev[cur], 0, sizeof(int32_t) * (size_t)c->B, c->s_aux));
    DDN_TRY(chain_decode(c, cur, 1, c->s_aux));
    DDN_TRY(ddn_p25_chain_wait(c));
    // what was flushed must not be decoded again should the stream go on: the carried stretch loses its sync marks
    HIP_TRY(hipMemsetAsync(c->d_fl[cur], 0, (size_t)c->B * c->stride, c->s_aux));
    c->last_set = cur;
    c->step++;
    return ddn_p25_chain_wait(c);
}

extern "C" int
ddn_p25_chain_wait(ddn_p25_chain* c) {
    if (!c) {
        return DDN_EINVAL;
    }
    DDN_TRY(chain_issue_pending(c, nullptr));
    if (c->have_user_stream) {
        HIP_TRY(hipEventSynchronize(c->ev_user));
    }
    HIP_TRY(hipStreamSynchronize(c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    HIP_TRY(hipStreamSynchronize(c->s_copy));
    HIP_TRY(hipStreamSynchronize(c->s_copy2));
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_get_results(ddn_p25_chain* c, ddn_p25_chain_results* r) {
    if (!c || !r) {
        return DDN_EINVAL;
    }
    const int cur = c->last_set;
    memset(r, 0, sizeof(*r));
    r->stride_symbols = c->stride;
    r->d_records10 = c->d_rec[cur];
    r->d_flags = c->d_fl[cur];
    r->d_counts = c->d_cnt_full;
    r->d_new = c->d_new[cur];
    r->d_events = c->d_ev[cur];
    r->d_n_events = c->d_nev[cur];
    r->d_event_data = c->d_evd[cur];
    DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &r->d_n_syncs, &r->d_sync_pos));
    DDN_TRY(ddn_p25p1_framer_device_dropped(c->fr, &r->d_dropped_syncs));
    r->d_nid4 = c->d_nid;
    r->d_tsbk = c->d_tsbk;
    r->d_tsbk_crc = c->d_tsbk_crc;
    for (int i = 0; i < 2; i++) {
        r->d_ldu_words[i] = c->d_words[i];
        r->d_ldu_rs_data[i] = c->d_rs_d[i];
        r->d_ldu_rs_status[i] = c->d_rs_st[i];
    }
    r->d_lsd_bits = c->d_lsd;
    r->d_lsd_ok = c->d_lsd_ok;
    r->d_hdu_rs_data = c->d_hdu_d;
    r->d_hdu_rs_status = c->d_hdu_rs;
    r->d_tdulc_rs_data = c->d_td_rd;
    r->d_tdulc_rs_status = c->d_td_rs;
    r->pdu_per_channel = c->PF;
    r->pdu_blocks = c->PB;
    r->d_n_pdu = c->d_n_pdu;
    r->d_pdu_slot = c->d_pdu_slot;
    r->d_pdu_header = c->d_pdu_hdr;
    r->d_pdu_info = c->d_pdu_info;
    r->d_pdu_blocks = c->d_pdu_blocks;
    r->d_pdu_block_valid = c->d_pdu_valid;
    r->d_pdu_blocks18 = c->d_pdu_blocks18;
    r->d_pdu_crc9_ok = c->d_pdu_crc9;
    r->d_n_ldu = c->d_nldu;
    r->d_imbe_bits = c->d_imbe_d;
    r->d_imbe_result = c->d_imbe_res;
    r->d_pcm = c->d_pcm;
    r->d_synth_result = c->d_res_out;
    return DDN_OK;
}

extern "C" size_t
ddn_p25_chain_stride_symbols(const ddn_p25_chain* c) {
    return c ? c->stride : 0;
}
extern "C" int
ddn_p25_chain_frame_slots(const ddn_p25_chain* c) {
    return c ? c->F : 0;
}
extern "C" int
ddn_p25_chain_max_ldu(const ddn_p25_chain* c) {
    return c ? c->Fv : 0;
}
extern "C" int
ddn_p25_chain_max_events(const ddn_p25_chain* c) {
    return c ? c->E : 0;
}
extern "C" void*
ddn_p25_chain_front_end(ddn_p25_chain* c) {
    return c ? c->fe : nullptr;
}
extern "C" void*
ddn_p25_chain_rx(ddn_p25_chain* c) {
    return c ? c->rx : nullptr;
}
extern "C" void*
ddn_p25_chain_mbe(ddn_p25_chain* c) {
    return c ? c->mbe : nullptr;
}

extern "C" int
ddn_fec_p25_tsbk_select_batch(const ddn_p25_12_candidate* d_candidates8, const int32_t* d_counts, size_t n, uint8_t* d_out12,
                              uint8_t* d_crc_ok, uint8_t* d_sel, void* hip_stream) {
    if (!d_candidates8 || !d_counts || !d_out12 || !d_crc_ok) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_tsbk_select((const uint8_t*)d_candidates8, d_counts, n, d_out12, d_crc_ok, d_sel, (hipStream_t)hip_stream));
    return DDN_OK;
}

// plain device-memory helpers for hosts without a HIP binding of their own (C callers, the ctypes tests)
extern "C" int
ddn_device_alloc(size_t bytes, void** out) {
    if (!out) {
        return DDN_EINVAL;
    }
    *out = nullptr;
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return DDN_OK;
}
extern "C" void
ddn_device_free(void* p) {
    (void)hipFree(p);
}
extern "C" int
ddn_device_upload(void* d_dst, const void* h_src, size_t bytes) {
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return DDN_OK;
}
extern "C" int
ddn_device_download(void* h_dst, const void* d_src, size_t bytes) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return DDN_OK;
}
extern "C" int
ddn_host_alloc_pinned(size_t bytes, void** out) {
    if (!out) {
        return DDN_EINVAL;
    }
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return DDN_OK;
}
extern "C" void
ddn_host_free_pinned(void* p) {
    (void)hipHostFree(p);
}
