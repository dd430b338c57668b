// ddn_api_chain.cpp - the P25 Phase 1 chain object (include/ddn_chain.h): stage order, buffers, double buffering and the
// streams / events of the pipelined forms, on top of the library's own C-ABI stage calls.  Host-only code.
//
// What it stands in for in a dsd-neo host: the demodulator thread's per-block loop (src/io/radio/rtl_sdr_fm.cpp:3458-3516) and
// processFrame()'s P25p1 branch (src/engine/protocol_dispatch.c:30-44, src/engine/dispatch/dispatch_p25p1.c), B channels wide.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

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

// Buffer sets: the receive loop of call k + 1 writes one set while call k is decoded out of another, and - _run_host - the results
// of call k - 1 may still be leaving a third (the engine copies complete by HSA signal, which a HIP stream cannot wait for: with
// three sets a call only ever has to wait, on the host, for copies issued two calls earlier, which are long done).
enum { NSET = 3 };
static inline int
set_prev(int cur) {
    return (cur + NSET - 1) % NSET;
}

struct ddn_p25_chain {
    ddn_p25_chain_config cfg;
    int B, n, T, F, Fv, E, EL;
    int off97[3]; // symbol a TSDU block's decision falls on, counted from the sync's last symbol
    size_t ms, stride, S, V;
    ddn_batch* fe;
    ddn_p25_rx* rx;
    // modulation = CQPSK: these two instead (symbols in d_disc, their per-channel counts in d_sym_cnt)
    ddn_cqpsk_batch* cq_fe = nullptr;
    ddn_cq_rx* cq = nullptr;
    int32_t* d_sym_cnt = nullptr;
    ddn_p25p1_framer* fr;
    ddn_mbe_batch* mbe;
    float* d_disc;
    float* d_disc2; // mixed chain only (ddn_p25_chain_double_disc): odd steps' discriminator output, so that the front end of call
                    // k + 1 (issued through _stage on a stream of its own) may run beside the loop of call k
    // receive-loop outputs, two sets: the loop of call k + 1 writes one while call k is decoded out of the other
    uint8_t *d_rec[NSET], *d_fl[NSET];
    uint8_t* d_rec2[NSET] = {nullptr, nullptr, nullptr}; // host form of the records (run_host with records2), allocated on first use
    float* d_pcm_dense[NSET] = {nullptr, nullptr, nullptr}; // dense PCM (run_host with pcm_dense), allocated on first use: [V][160]
    int32_t *d_pcm_slot[NSET] = {nullptr, nullptr, nullptr}, *d_pcm_bcnt = nullptr, *d_pcm_boff = nullptr,
            *d_pcm_total[NSET] = {nullptr, nullptr, nullptr};
    int32_t *d_new[NSET], *d_ev[NSET], *d_nev[NSET], *d_evd[NSET];
    // the decisions of the records a row holds (carried + new), by row index: what files NIDs and TSDU blocks by frame
    int32_t *d_evl[NSET], *d_evdl[NSET], *d_nevl[NSET];
    int32_t *d_cnt_scan, *d_cnt_full[NSET];
    // decode buffers; the ones a host result set names (counts, NIDs, TSDU blocks, PCM) exist per buffer set like the loop's outputs:
    // the copies of call k's results may still be leaving while call k + 1 is decoded (_run_host)
    int32_t* d_nid[NSET];
    int32_t *d_lists, *d_list_n; // per frame type: the slots holding a frame of it (k_chain_frames), what the decode launches walk
    uint8_t* d_cls; // frame type of every slot (DDN_CLS_*): the per-type decode launches only work on their own frames
    uint8_t *d_tsbk[NSET], *d_tsbk_crc;
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
    int16_t *d_pdu_llr, *d_pdu_hllr;
    int64_t* d_first;
    int32_t *d_sc, *d_nldu, *d_sc_out, *d_imbe_res, *d_res_out;
    uint8_t *d_imbe_fr, *d_imbe_soft, *d_imbe_fl, *d_imbe_d;
    float* d_pcm[NSET];
    // pipelining
    hipStream_t s_main, s_aux, s_copy, s_copy2; // front end + loop | decode | H2D | D2H
    hipStream_t s_voice = nullptr;              // the voice stage of a decode, beside its frame FEC
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_produced[NSET], ev_consumed[NSET], ev_out[NSET]; // per buffer set
    hipEvent_t ev_in[2], ev_in_free[2];                             // per input buffer (_run_host: two, used in turn)
    hipEvent_t ev_loop[NSET] = {nullptr, nullptr, nullptr}; // the receive loop of that set's call is next on s_main
    // what the NEXT call's receive loop waits for: that set's frame FEC and the voice stage up to the frame decode are done.  The
    // parameter + synthesis kernels behind them need next to no LDS (0 / 1.5 KB) and few registers (16 / 64), so they can run on
    // the same CUs as the loop's workgroups (which take 2 x 77 KB LDS and 2 x 214 registers per SIMD) - beside the loop instead of
    // ahead of it.  synth_beside_loop = 0 (DDN_CHAIN_SYNTH_BESIDE_LOOP=0) restores the order of round 4: the loop waits for it all.
    hipEvent_t ev_pre[NSET] = {nullptr, nullptr, nullptr}, ev_join_a = nullptr, ev_aux_done[NSET] = {nullptr, nullptr, nullptr};
    int synth_beside_loop = 1;
    // ... which takes queueing them in the NEXT call, behind the event that call records right before its loop kernel: the pipelined
    // forms leave a call's parameter + synthesis kernels (and the dense-PCM pack) "deferred"; the next call, _wait, _flush,
    // _get_results or a device-form call issues them
    int have_deferred = 0, deferred_set = 0, deferred_dense = 0;
    int want_rec2 = 0; // the decode being queued also packs records2 (_run_host)
    // _run_host: the result copies of a call are issued in the NEXT call (or by _wait / _flush), beside that call's receive loop
    ddn_p25_chain_host_out pending_out;
    int have_pending = 0, pending_set = 0;
    // _run_host's result copies on an SDMA engine (below HIP: hipMemcpyAsync device -> pinned host is a shader blit kernel on this
    // ROCm, which cannot start while the receive loop holds every CU; hsa_amd_memory_async_copy_on_engine runs beside any kernel -
    // tools/ubench/d2h_sdma.hip: 57 GB/s alone, 48.5 GB/s each way with an H2D copy in flight, identical beside a kernel that holds
    // every CU).  One completion signal per buffer set, counted up per copy issued and down by the engine.
    int sdma = 0;                 // 1 = in use (agents + engine resolved), 0 = not tried, -1 = unavailable / switched off
    hsa_agent_t sdma_gpu, sdma_cpu;
    uint32_t sdma_engine = 0;     // hsa_amd_sdma_engine_id_t bit, 0 = let the runtime pick (hsa_amd_memory_async_copy)
    hsa_signal_t sdma_sig[NSET];
    int sdma_sig_ok = 0;
    // The engine copies are ordered behind a call's decode on the HOST (an HSA copy takes HSA signals, not HIP events), and the
    // calling thread must not be the one that waits: _run_host has to return at once so that the next call's input copy is queued
    // early (measured: waiting in the call put the 8 ms H2D transfer on the critical path, 17.3 ms per step).  A worker thread per
    // chain object waits for the decode's event and hands the result set to the engine.
    struct OutJob {
        int set;
        ddn_p25_chain_host_out out;
    };
    std::thread* out_thread = nullptr;
    std::mutex* out_mu = nullptr;
    std::condition_variable* out_cv = nullptr;
    std::deque<OutJob>* out_q = nullptr;
    int out_stop = 0, out_rc = 0, out_dev = 0;
    char out_err[256] = {0}; // the worker's error text (ddn_last_error is per thread): handed to the caller's thread with out_rc
    long out_submitted[NSET] = {0, 0, 0}, out_issued[NSET] = {0, 0, 0};
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
    if (c->out_thread) {
        {   // jobs the worker has not picked up are dropped (their host buffers may be gone already); the one it is working on
            // is finished before the join returns
            std::lock_guard<std::mutex> lk(*c->out_mu);
            for (const ddn_p25_chain::OutJob& j : *c->out_q) {
                c->out_issued[j.set]++;
            }
            c->out_q->clear();
            c->out_stop = 1;
        }
        c->out_cv->notify_all();
        c->out_thread->join();
        delete c->out_thread;
        c->out_thread = nullptr;
    }
    if (c->sdma > 0 && c->sdma_sig_ok) {
        // hipDeviceSynchronize() does not cover hsa_amd_memory_async_copy(_on_engine): copies already on the engine still read the
        // device buffers freed below and write the caller's host buffers - wait for every set's completion signal
        for (int k = 0; k < NSET; k++) {
            while (hsa_signal_wait_scacquire(c->sdma_sig[k], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {
            }
        }
    }
    delete c->out_q;
    delete c->out_cv;
    delete c->out_mu;
    ddn_batch_destroy(c->fe);
    ddn_p25_rx_destroy(c->rx);
    ddn_cqpsk_batch_destroy(c->cq_fe);
    ddn_cq_rx_destroy(c->cq);
    (void)hipFree(c->d_sym_cnt);
    ddn_p25p1_framer_destroy(c->fr);
    ddn_mbe_batch_destroy(c->mbe);
    void* all[] = {c->d_disc, c->d_disc2, c->d_pcm_bcnt,
                   c->d_pcm_boff, c->d_cnt_scan, c->d_cls, c->d_lists, c->d_list_n, c->d_tsbk_crc, c->d_words[0],
                   c->d_words[1], c->d_wrel, c->d_werrs, c->d_vldu, c->d_rs_d[0], c->d_rs_d[1], c->d_rs_p[0], c->d_rs_p[1],
                   c->d_rs_st[0], c->d_rs_st[1], c->d_lsd, c->d_lsd_ok, c->d_lsd_llr, c->d_hdu_hex, c->d_hdu_par, c->d_hdu_st,
                   c->d_hdu_d, c->d_hdu_p, c->d_hdu_rs, c->d_td_d, c->d_td_p, c->d_td_st, c->d_td_rd, c->d_td_rp, c->d_td_rs,
                   c->d_first, c->d_sc, c->d_nldu, c->d_sc_out, c->d_imbe_res, c->d_res_out, c->d_imbe_fr, c->d_imbe_soft,
                   c->d_imbe_fl, c->d_imbe_d, c->d_iq[0], c->d_iq[1], c->d_pdu_slot, c->d_pdu_info, c->d_n_pdu,
                   c->d_pdu_metric, c->d_pdu_hdr, c->d_pdu_valid, c->d_pdu_blocks, c->d_pdu_llr, c->d_pdu_wanted, c->d_pdu_cand,
                   c->d_pdu_blocks18, c->d_pdu_crc9, c->d_pdu_cnt, c->d_pdu_hllr};
    for (void* p : all) {
        (void)hipFree(p);
    }
    for (int k = 0; k < NSET; k++) {
        void* per_set[] = {c->d_rec[k], c->d_rec2[k], c->d_pcm_dense[k], c->d_pcm_slot[k], c->d_pcm_total[k], c->d_fl[k], c->d_new[k], c->d_ev[k],
                           c->d_nev[k], c->d_evd[k], c->d_evl[k], c->d_evdl[k], c->d_nevl[k], c->d_cnt_full[k], c->d_nid[k], c->d_tsbk[k],
                           c->d_pcm[k]};
        for (void* p : per_set) {
            (void)hipFree(p);
        }
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
    if (c->sdma_sig_ok) {
        for (int k = 0; k < NSET; k++) {
            (void)hsa_signal_destroy(c->sdma_sig[k]);
        }
        (void)hsa_shut_down(); // (the reference count hsa_init() below took)
    }
    hipEvent_t evs[] = {c->ev_in[0], c->ev_in[1], c->ev_in_free[0], c->ev_in_free[1], c->ev_join_a, c->ev_t[0], c->ev_t[1], c->ev_t[2],
                        c->ev_t[3], c->ev_t[4], c->ev_t[5]};
    for (int k = 0; k < NSET; k++) {
        hipEvent_t per_set[] = {c->ev_produced[k], c->ev_consumed[k], c->ev_out[k], c->ev_loop[k], c->ev_pre[k], c->ev_aux_done[k]};
        for (hipEvent_t e : per_set) {
            if (e) {
                (void)hipEventDestroy(e);
            }
        }
    }
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
    c->T = cfg->carry_symbols > 0 ? cfg->carry_symbols : 960;
    c->F = cfg->max_frames > 0 ? cfg->max_frames : cfg->samples_per_call / 1800 + 6;
    c->Fv = cfg->max_ldu > 0 ? cfg->max_ldu : cfg->samples_per_call / 8640 + 3;
    c->E = cfg->max_events > 0 ? cfg->max_events : 4 * c->F;
    c->EL = c->E + 64; // + the decisions inside a carried tail
    c->PF = 2;         // data units per channel and call (a second's worth of calls rarely holds one)
    // data blocks per unit: a sync decoded in this call is only guaranteed T symbols behind it, and data block b ends
    // (56 + 98 b + 97) dibits + one status symbol per 35 - 23 symbols behind its sync: 839 for b = 7, 940 for b = 8.  The default
    // T = 960 (round 6; it was 896) guarantees the eighth block of a unit whose sync falls late in the scan range, so the default
    // reads eight blocks per unit (a longer unit is flagged 8); a caller's carry below 941 symbols reads seven.
    c->PB = c->T >= 941 ? 8 : 7;
    int rc = DDN_OK;
    do {
        const bool cqpsk = cfg->modulation == DDN_P25_MOD_CQPSK;
        if (cfg->modulation != DDN_P25_MOD_C4FM && !cqpsk) {
            ddn_set_error("ddn_p25_chain_create: modulation %d", cfg->modulation);
            rc = DDN_EINVAL;
            break;
        }
        if (cqpsk) {
            const int rate = cfg->sample_rate_hz > 0 ? cfg->sample_rate_hz : 48000;
            ddn_cqpsk_config qc = {c->B, rate, 4800, DDN_LPF_P25_CQPSK, 1, cfg->input_format, cfg->block_len, 0.0f};
            if ((rc = ddn_cqpsk_batch_create(&qc, &c->cq_fe)) != DDN_OK) {
                break;
            }
            ddn_cq_rx_config rq = {c->B, DDN_CQ_P25P1, 0, 64, cfg->snr_cqpsk_db};
            if ((rc = ddn_cq_rx_create(&rq, &c->cq)) != DDN_OK) {
                break;
            }
        } else {
            ddn_front_end_config fc = {c->B, 48000, 4800, 4, DDN_LPF_P25_C4FM, cfg->input_format, cfg->block_len, 0.0f};
            if ((rc = ddn_batch_create(&fc, &c->fe)) != DDN_OK) {
                break;
            }
            ddn_p25_rx_config rc_cfg = {c->B, 48000, 4800, 0, 1};
            if ((rc = ddn_p25_rx_create(&rc_cfg, &c->rx)) != DDN_OK || (rc = ddn_p25_rx_set_handlers(c->rx, 1, 64)) != DDN_OK) {
                break;
            }
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
        c->ms = c->cq ? ddn_cqpsk_max_symbols(c->cq_fe, (size_t)c->n) : ddn_p25_rx_max_symbols(c->rx, (size_t)c->n);
        c->stride = (size_t)c->T + c->ms;
        c->S = (size_t)c->B * (size_t)c->F;
        c->V = (size_t)c->B * (size_t)c->Fv * 9;
        const size_t B = (size_t)c->B, S = c->S, V = c->V;
        bool ok = dalloc(&c->d_disc, B * (size_t)c->n) && dalloc(&c->d_sym_cnt, B);
        for (int k = 0; k < NSET && ok; k++) {
            ok = dalloc(&c->d_rec[k], B * c->stride * 10) && dalloc(&c->d_fl[k], B * c->stride) && dalloc(&c->d_new[k], B)
                 && dalloc(&c->d_ev[k], B * (size_t)c->E * 4) && dalloc(&c->d_nev[k], B) && dalloc(&c->d_evd[k], B * (size_t)c->E * 4)
                 && dalloc(&c->d_evl[k], B * (size_t)c->EL * 4) && dalloc(&c->d_evdl[k], B * (size_t)c->EL * 4) && dalloc(&c->d_nevl[k], B)
                 && dalloc(&c->d_cnt_full[k], B) && dalloc(&c->d_nid[k], S * 4) && dalloc(&c->d_tsbk[k], 3 * S * 12) && dalloc(&c->d_pcm[k], V * 160);
        }
        ok = ok && dalloc(&c->d_cnt_scan, B) && dalloc(&c->d_cls, S) && dalloc(&c->d_lists, S * DDN_LIST_COUNT) && dalloc(&c->d_list_n, 8)
             && dalloc(&c->d_tsbk_crc, 3 * S) && dalloc(&c->d_words[0], S * 240)
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
             && dalloc(&c->d_pdu_slot, B * (size_t)c->PF) && dalloc(&c->d_pdu_info, B * (size_t)c->PF * 4)
             && dalloc(&c->d_n_pdu, B) && dalloc(&c->d_pdu_hdr, B * (size_t)c->PF * 12)
             && dalloc(&c->d_pdu_valid, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_blocks, B * (size_t)c->PF * (size_t)c->PB * 12)
             && dalloc(&c->d_pdu_metric, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_llr, B * (size_t)c->PF * (size_t)c->PB * 196)
             && dalloc(&c->d_pdu_wanted, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_cand, B * (size_t)c->PF * (size_t)c->PB * 8 * 24)
             && dalloc(&c->d_pdu_blocks18, B * (size_t)c->PF * (size_t)c->PB * 18) && dalloc(&c->d_pdu_crc9, B * (size_t)c->PF * (size_t)c->PB)
             && dalloc(&c->d_pdu_cnt, B * (size_t)c->PF * (size_t)c->PB) && dalloc(&c->d_pdu_hllr, B * (size_t)c->PF * 196);
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
        hipEvent_t* evs[] = {&c->ev_in[0], &c->ev_in[1], &c->ev_in_free[0], &c->ev_in_free[1], &c->ev_join_a,
                             &c->ev_produced[0], &c->ev_produced[1], &c->ev_produced[2], &c->ev_consumed[0], &c->ev_consumed[1],
                             &c->ev_consumed[2], &c->ev_out[0], &c->ev_out[1], &c->ev_out[2], &c->ev_loop[0], &c->ev_loop[1],
                             &c->ev_loop[2], &c->ev_pre[0], &c->ev_pre[1], &c->ev_pre[2], &c->ev_aux_done[0], &c->ev_aux_done[1],
                             &c->ev_aux_done[2]};
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
        c->synth_beside_loop = 1;
        if (const char* e = DDN_EXP_ENV("DDN_CHAIN_SYNTH_BESIDE_LOOP")) {
            c->synth_beside_loop = atoi(e) != 0;
        }
    } while (0);
    if (rc != DDN_OK) {
        ddn_p25_chain_destroy(c);
        return rc;
    }
    *out = c;
    return DDN_OK;
}

// the discriminator buffer of this step (two of them only in the mixed chain)
static float*
chain_disc(ddn_p25_chain* c) {
    return (c->d_disc2 && (c->step & 1)) ? c->d_disc2 : c->d_disc;
}

static int
chain_carry(ddn_p25_chain* c, int cur, hipStream_t st) {
    const int prev = set_prev(cur);
    HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_rec[cur], c->d_fl[cur],
                                c->stride, c->T, c->B, st));
    return DDN_OK;
}

// carry + front end of one call into buffer set `cur` on stream st (with_carry = false: the caller copies the carried tail
// itself, ahead of the loop - the mixed chain, whose front ends run on streams of their own beside the previous call's loops)
static int
chain_front(ddn_p25_chain* c, const void* d_iq, int cur, hipStream_t st, bool with_carry = true) {
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[0], st));
    }
    if (with_carry) {
        DDN_TRY(chain_carry(c, cur, st));
    }
    if (c->cq) { // CQPSK: I/Q -> one float per symbol (row stride ms <= n), counts per channel
        DDN_TRY(ddn_cqpsk_run(c->cq_fe, d_iq, (size_t)c->n, chain_disc(c), c->ms, c->d_sym_cnt, st));
    } else {
        DDN_TRY(ddn_front_end_run(c->fe, d_iq, (size_t)c->n, chain_disc(c), st));
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[1], st));
    }
    return DDN_OK;
}

// the receive loop of that call
static int
chain_loop(ddn_p25_chain* c, int cur, hipStream_t st) {
    if (c->cq) { // the symbol-rate loop: same records, flags, counts and event lists
        DDN_TRY(ddn_cq_rx_set_events(c->cq, c->d_ev[cur], c->d_nev[cur], c->d_evd[cur], (size_t)c->E));
        DDN_TRY(ddn_cq_rx_run(c->cq, chain_disc(c), c->d_sym_cnt, c->ms, c->ms, c->d_rec[cur] + (size_t)c->T * 10, c->d_fl[cur] + c->T, c->d_new[cur],
                              c->stride, st));
        if (c->timing) {
            HIP_TRY(hipEventRecord(c->ev_t[2], st));
        }
        return DDN_OK;
    }
    DDN_TRY(ddn_p25_rx_set_events(c->rx, c->d_ev[cur], c->d_nev[cur], (size_t)c->E));
    DDN_TRY(ddn_p25_rx_set_event_data(c->rx, c->d_evd[cur]));
    // the loop writes its records behind the T carried ones: row pointer + T records, row stride unchanged
    DDN_TRY(ddn_p25_rx_run(c->rx, chain_disc(c), (size_t)c->n, c->d_rec[cur] + (size_t)c->T * 10, c->d_fl[cur] + c->T, c->d_new[cur],
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
    if (c->cq) { // (no matched filter between the demodulator and the loop)
        if (before_loop) {
            HIP_TRY(hipStreamWaitEvent(st, before_loop, 0));
        }
        if (loop_next) {
            HIP_TRY(hipEventRecord(loop_next, st));
        }
        return chain_loop(c, cur, st);
    }
    if (before_loop) { // (waited for inside ddn_p25_rx_run, behind the matched filter: that one may run beside the decode kernels)
        DDN_TRY(ddn_p25_rx_gate_loop(c->rx, before_loop));
    }
    if (loop_next) { // recorded inside ddn_p25_rx_run, after the matched filter
        DDN_TRY(ddn_p25_rx_mark_loop_start(c->rx, loop_next));
    }
    return chain_loop(c, cur, st);
}


// ---- result copies on an SDMA engine ------------------------------------------------------------------------------------------
// Agents are taken from the pointers themselves (the device buffer's owner = this chain's GPU, whatever HIP_VISIBLE_DEVICES did to the
// ordinals; the pinned buffer's owner = its NUMA node's CPU agent).  cfg.d2h_blit = 1 keeps the hipMemcpyAsync path (A/B runs).
static bool
sdma_setup(ddn_p25_chain* c, const void* h_any) {
    if (c->sdma != 0) {
        return c->sdma > 0;
    }
    c->sdma = -1;
    const char* e = DDN_EXP_ENV("DDN_D2H");
    if (c->cfg.d2h_blit || (e && strcmp(e, "blit") == 0)) {
        return false;
    }
    if (hsa_init() != HSA_STATUS_SUCCESS) {
        return false;
    }
    bool ok = false;
    do {
        hsa_amd_pointer_info_t pd, ph;
        memset(&pd, 0, sizeof(pd));
        memset(&ph, 0, sizeof(ph));
        pd.size = sizeof(pd);
        ph.size = sizeof(ph);
        if (hsa_amd_pointer_info(c->d_rec[0], &pd, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pd.type == HSA_EXT_POINTER_TYPE_UNKNOWN
            || hsa_amd_pointer_info(const_cast<void*>(h_any), &ph, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS
            || ph.type == HSA_EXT_POINTER_TYPE_UNKNOWN) {
            break; // (a host buffer the runtime does not know - not pinned - keeps the hipMemcpyAsync path, which stages it)
        }
        hsa_device_type_t td, th;
        if (hsa_agent_get_info(pd.agentOwner, HSA_AGENT_INFO_DEVICE, &td) != HSA_STATUS_SUCCESS || td != HSA_DEVICE_TYPE_GPU
            || hsa_agent_get_info(ph.agentOwner, HSA_AGENT_INFO_DEVICE, &th) != HSA_STATUS_SUCCESS || th != HSA_DEVICE_TYPE_CPU) {
            break;
        }
        c->sdma_gpu = pd.agentOwner;
        c->sdma_cpu = ph.agentOwner;
        // An engine of the device -> host set that is NOT the one the input copies run on: hipMemcpyAsync host -> device takes the
        // preferred engine of that direction, and two copies on one engine run one after the other (measured: 16.4 ms per step on
        // the shared engine against 12.1-12.8 on separate ones).  The runtime only states a preference for the CPU agent nearest
        // to the device (for the other NUMA node's agent it answers "all 16" / "none"), so every CPU agent is asked; without any
        // answer the input copies are taken to run on engine 0 (what this ROCm does) and engines 1-3 to reach the host at full rate
        // (measured: 12.8 / 13.0 / 13.2 ms per step; engine 4 - an xGMI engine - 41.6).
        uint32_t avail = 0, mask = 0, h2d = 0;
        (void)hsa_amd_memory_copy_engine_status(c->sdma_cpu, c->sdma_gpu, &avail);
        struct Ask {
            hsa_agent_t gpu;
            uint32_t avail, d2h, h2d;
        } ask = {c->sdma_gpu, avail, 0, 0};
        (void)hsa_iterate_agents(
            [](hsa_agent_t a, void* p) {
                Ask* q = (Ask*)p;
                hsa_device_type_t t;
                if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_CPU) {
                    return HSA_STATUS_SUCCESS;
                }
                uint32_t m = 0;
                if (!q->d2h && hsa_amd_memory_get_preferred_copy_engine(a, q->gpu, &m) == HSA_STATUS_SUCCESS && m != 0 && m != q->avail) {
                    q->d2h = m;
                }
                m = 0;
                if (!q->h2d && hsa_amd_memory_get_preferred_copy_engine(q->gpu, a, &m) == HSA_STATUS_SUCCESS && m != 0 && m != q->avail) {
                    q->h2d = m;
                }
                return HSA_STATUS_SUCCESS;
            },
            &ask);
        h2d = ask.h2d ? ask.h2d : 0x1u;
        mask = ask.d2h ? ask.d2h : ((avail & 0xEu) ? (avail & 0xEu) : avail);
        const uint32_t h2d_engine = h2d & (~h2d + 1u);
        if (mask & ~h2d_engine) {
            mask &= ~h2d_engine;
        }
        if (DDN_EXP_ENV("DDN_D2H_VERBOSE")) {
            fprintf(stderr, "ddn_p25_chain: SDMA engines device->host preferred 0x%x available 0x%x, host->device preferred 0x%x -> 0x%x\n",
                    mask, avail, h2d, mask & (~mask + 1u));
        }
        if (const char* pick = DDN_EXP_ENV("DDN_D2H_ENGINE")) { // (experiments: an engine bit of hsa_amd_sdma_engine_id_t)
            const long v = strtol(pick, nullptr, 0);
            if (v > 0 && v <= 0x8000 && (v & (v - 1)) == 0) {
                mask = (uint32_t)v;
            }
        }
        c->sdma_engine = mask & (~mask + 1u); // lowest engine of the set
        int made = 0;
        for (; made < NSET; made++) {
            if (hsa_signal_create(0, 0, nullptr, &c->sdma_sig[made]) != HSA_STATUS_SUCCESS) {
                break;
            }
        }
        if (made < NSET) {
            while (made-- > 0) {
                (void)hsa_signal_destroy(c->sdma_sig[made]);
            }
            break;
        }
        c->sdma_sig_ok = 1;
        ok = true;
    } while (0);
    if (!ok) {
        (void)hsa_shut_down();
        return false;
    }
    c->sdma = 1;
    return true;
}

static int
sdma_copy(ddn_p25_chain* c, int set, void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) {
        return DDN_OK;
    }
    hsa_signal_add_relaxed(c->sdma_sig[set], 1);
    const hsa_status_t r =
        c->sdma_engine ? hsa_amd_memory_async_copy_on_engine(h_dst, c->sdma_cpu, d_src, c->sdma_gpu, bytes, 0, nullptr, c->sdma_sig[set],
                                                             (hsa_amd_sdma_engine_id_t)c->sdma_engine, false)
                       : hsa_amd_memory_async_copy(h_dst, c->sdma_cpu, d_src, c->sdma_gpu, bytes, 0, nullptr, c->sdma_sig[set]);
    if (r != HSA_STATUS_SUCCESS) {
        hsa_signal_subtract_relaxed(c->sdma_sig[set], 1);
        const char* m = "";
        (void)hsa_status_string(r, &m);
        ddn_set_error("ddn_p25_chain: SDMA result copy failed: %s", m);
        return DDN_EHIP;
    }
    return DDN_OK;
}

// the copies issued for that buffer set have all landed in the caller's buffers
static void
sdma_wait(ddn_p25_chain* c, int set) {
    if (c->sdma > 0) {
        if (c->out_thread) { // every result set handed to the worker for this buffer set has reached the engine
            std::unique_lock<std::mutex> lk(*c->out_mu);
            c->out_cv->wait(lk, [&] { return c->out_issued[set] == c->out_submitted[set]; });
        }
        while (hsa_signal_wait_scacquire(c->sdma_sig[set], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {
        }
    }
}

// the worker's verdict on the copies issued so far, taken over by the calling thread (error text included) and cleared
static int
out_take_rc(ddn_p25_chain* c) {
    if (!c->out_thread) {
        return DDN_OK;
    }
    std::lock_guard<std::mutex> lk(*c->out_mu);
    const int rc = c->out_rc;
    if (rc != DDN_OK) {
        ddn_set_error("%s", c->out_err[0] ? c->out_err : "ddn_p25_chain: a result copy failed");
        c->out_rc = DDN_OK;
        c->out_err[0] = 0;
    }
    return rc;
}

// the device -> pinned-host copies of the results of the call that used buffer set `set`: on an SDMA engine (the caller has
// waited for that call's decode on the host), else on the second copy stream
static int
chain_copy_out(ddn_p25_chain* c, const ddn_p25_chain_host_out* out, int set) {
    const size_t B = (size_t)c->B, S = c->S, V = c->V;
    auto cp = [&](void* h_dst, const void* d_src, size_t bytes) -> int {
        if (c->sdma > 0) {
            return sdma_copy(c, set, h_dst, d_src, bytes);
        }
        HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->s_copy2));
        return DDN_OK;
    };
    const bool dense_pcm = out->pcm_dense && out->pcm_slot && out->pcm_count && out->pcm_dense_frames > 0 && c->cfg.vocoder;
    if (out->records10) {
        DDN_TRY(cp(out->records10, c->d_rec[set], B * c->stride * 10));
    }
    if (out->flags) {
        DDN_TRY(cp(out->flags, c->d_fl[set], B * c->stride));
    }
    if (out->records2) {
        DDN_TRY(cp(out->records2, c->d_rec2[set], B * c->stride * 2));
    }
    if (out->counts) {
        DDN_TRY(cp(out->counts, c->d_cnt_full[set], B * 4));
    }
    if (out->events) {
        DDN_TRY(cp(out->events, c->d_ev[set], B * (size_t)c->E * 16));
    }
    if (out->event_data) {
        DDN_TRY(cp(out->event_data, c->d_evd[set], B * (size_t)c->E * 16));
    }
    if (out->n_events) {
        DDN_TRY(cp(out->n_events, c->d_nev[set], B * 4));
    }
    if (out->nid4) {
        DDN_TRY(cp(out->nid4, c->d_nid[set], S * 16));
    }
    if (out->tsbk) {
        DDN_TRY(cp(out->tsbk, c->d_tsbk[set], 3 * S * 12));
    }
    if (out->pcm && c->cfg.vocoder) {
        DDN_TRY(cp(out->pcm, c->d_pcm[set], V * 160 * 4));
    }
    if (dense_pcm) {
        const size_t nf = (size_t)out->pcm_dense_frames < V ? (size_t)out->pcm_dense_frames : V;
        DDN_TRY(cp(out->pcm_dense, c->d_pcm_dense[set], nf * 160 * 4));
        DDN_TRY(cp(out->pcm_slot, c->d_pcm_slot[set], nf * 4));
        DDN_TRY(cp(out->pcm_count, c->d_pcm_total[set], 4));
    }
    return DDN_OK;
}


static int chain_copy_out(ddn_p25_chain* c, const ddn_p25_chain_host_out* out, int set);

// the worker of the engine route: per job, wait for that call's decode (HIP event, on the host), then issue the set's copies
static void
out_worker(ddn_p25_chain* c) {
    (void)hipSetDevice(c->out_dev);
    for (;;) {
        ddn_p25_chain::OutJob job;
        {
            std::unique_lock<std::mutex> lk(*c->out_mu);
            c->out_cv->wait(lk, [&] { return c->out_stop || !c->out_q->empty(); });
            if (c->out_stop) {
                return;
            }
            job = c->out_q->front();
            c->out_q->pop_front();
        }
        int rc = DDN_OK;
        if (hipEventSynchronize(c->ev_consumed[job.set]) != hipSuccess) {
            rc = DDN_EHIP;
        } else {
            rc = chain_copy_out(c, &job.out, job.set);
        }
        {
            std::lock_guard<std::mutex> lk(*c->out_mu);
            if (rc != DDN_OK && c->out_rc == DDN_OK) {
                c->out_rc = rc;
                snprintf(c->out_err, sizeof(c->out_err), "%s", ddn_last_error());
            }
            c->out_issued[job.set]++;
        }
        c->out_cv->notify_all();
    }
}

static int
out_worker_start(ddn_p25_chain* c) {
    if (c->out_thread) {
        return DDN_OK;
    }
    HIP_TRY(hipGetDevice(&c->out_dev));
    c->out_mu = new (std::nothrow) std::mutex();
    c->out_cv = new (std::nothrow) std::condition_variable();
    c->out_q = new (std::nothrow) std::deque<ddn_p25_chain::OutJob>();
    if (!c->out_mu || !c->out_cv || !c->out_q) {
        return DDN_ENOMEM;
    }
    c->out_thread = new (std::nothrow) std::thread(out_worker, c);
    return c->out_thread ? DDN_OK : DDN_ENOMEM;
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
    if (c->sdma > 0) {
        // the engine copies are ordered against the kernels on the host, by the worker: it waits for that call's decode (and its
        // pack kernels) and issues the copies; this thread goes on at once
        DDN_TRY(out_worker_start(c));
        c->have_pending = 0;
        {
            std::lock_guard<std::mutex> lk(*c->out_mu);
            ddn_p25_chain::OutJob job;
            job.set = set;
            job.out = c->pending_out;
            c->out_q->push_back(job);
            c->out_submitted[set]++;
        }
        c->out_cv->notify_all();
        return out_take_rc(c); // (an earlier set's copies may have failed)
    }
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
static int chain_issue_deferred(ddn_p25_chain* c, hipEvent_t gate);

static int
chain_settle_pending(ddn_p25_chain* c, hipStream_t st, bool deferred_too = true) {
    if (c->have_deferred && deferred_too) { // a pipelined call's synthesis is still to be queued: now, and `st` goes on behind it
        const int set = c->deferred_set;
        DDN_TRY(chain_issue_deferred(c, nullptr));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_consumed[set], 0));
    }
    if (!c->have_pending) {
        return DDN_OK;
    }
    const int set = c->pending_set;
    DDN_TRY(chain_issue_pending(c, nullptr));
    if (c->sdma > 0) { // (engine copies complete by signal, not by event: this rare hand-over between the forms waits on the host)
        for (int k = 0; k < NSET; k++) {
            sdma_wait(c, k);
        }
        return DDN_OK;
    }
    HIP_TRY(hipStreamWaitEvent(st, c->ev_out[set], 0));
    return DDN_OK;
}

// framer + every frame type's FEC + voice of buffer set `cur` on stream st
static int
chain_decode(ddn_p25_chain* c, int cur, int flush, hipStream_t st, hipEvent_t ev_pre = nullptr, bool defer_synth = false) {
    const size_t S = c->S, V = c->V, stride = c->stride;
    const uint8_t* rec = c->d_rec[cur];
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[3], st));
    }
    HIP_TRY(ddn_dev_chain_counts(c->d_new[cur], c->T, c->B, flush, c->d_cnt_scan, c->d_cnt_full[cur], st));
    DDN_TRY(ddn_p25p1_framer_index(c->fr, c->d_fl[cur], c->d_cnt_scan, stride, st));
    // the NID and the TSDU blocks of every frame were decoded inside the loop by its handlers (p25p1_nid_decode,
    // tsbk_decode_repetition_bytes: ddn_p25_rx_set_event_data); they are filed by frame here, not decoded a second time
    {
        const int prev = set_prev(cur);
        HIP_TRY(ddn_dev_chain_events(c->d_evl[prev], c->d_evdl[prev], c->d_nevl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_ev[cur],
                                     c->d_evd[cur], c->d_nev[cur], c->E, c->EL, c->T, c->B, c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur],
                                     st));
        const int32_t *d_ns = nullptr, *d_sp = nullptr;
        DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &d_ns, &d_sp));
        HIP_TRY(ddn_dev_zero_words(c->d_list_n, 8, st)); // (not hipMemsetAsync: its fill kernel stalled 2 ms beside the front end)
        HIP_TRY(ddn_dev_chain_frames(c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur], c->EL, d_sp, d_ns, c->B, c->F, c->off97[0],
                                     c->off97[1], c->off97[2], c->d_nid[cur], c->d_tsbk[cur], c->d_tsbk_crc, c->d_cls, c->d_lists, c->d_list_n,
                                     st));
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
        DDN_TRY(ddn_p25p1_framer_gather_ldu_words(c->fr, ldu, rec, c->d_cnt_full[cur], stride, c->d_words[i], c->d_wrel, c->d_vldu, st));
        DDN_TRY(ddn_fec_hamming_10_6_3_batch(c->d_words[i], S * 24, c->d_werrs, st));
        DDN_TRY(ddn_p25p1_framer_pack_ldu_rs(c->fr, ldu, c->d_words[i], c->d_rs_d[i], c->d_rs_p[i], st));
        DDN_TRY(ddn_fec_p25_rs_batch(i == 0 ? DDN_RS_24_12_13 : DDN_RS_24_16_9, c->d_rs_d[i], c->d_rs_p[i], S, c->d_rs_st[i], st));
    }
    // HDU: 36 Golay(24,6) words -> RS(36,20,17)
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_HDU * S, c->d_list_n + DDN_LIST_HDU);
    DDN_TRY(ddn_p25p1_framer_gather_hdu(c->fr, rec, c->d_cnt_full[cur], stride, c->d_hdu_hex, c->d_hdu_par, nullptr, nullptr, c->d_vldu, st));
    DDN_TRY(ddn_fec_golay24_batch(6, c->d_hdu_hex, c->d_hdu_par, S * 36, c->d_hdu_st, nullptr, st));
    DDN_TRY(ddn_p25p1_framer_pack_hdu_rs(c->fr, c->d_hdu_hex, c->d_hdu_d, c->d_hdu_p, st));
    DDN_TRY(ddn_fec_p25_rs_batch(DDN_RS_36_20_17, c->d_hdu_d, c->d_hdu_p, S, c->d_hdu_rs, st));
    // TDULC: 12 Golay(24,12) words -> RS(24,12,13)
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_TDULC * S, c->d_list_n + DDN_LIST_TDULC);
    DDN_TRY(ddn_p25p1_framer_gather_tdulc(c->fr, rec, c->d_cnt_full[cur], stride, c->d_td_d, c->d_td_p, nullptr, nullptr, c->d_vldu, st));
    DDN_TRY(ddn_fec_golay24_batch(12, c->d_td_d, c->d_td_p, S * 12, c->d_td_st, nullptr, st));
    DDN_TRY(ddn_p25p1_framer_pack_tdulc_rs(c->fr, c->d_td_d, c->d_td_rd, c->d_td_rp, st));
    DDN_TRY(ddn_fec_p25_rs_batch(DDN_RS_24_12_13, c->d_td_rd, c->d_td_rp, S, c->d_td_rs, st));
    // Last on this stream, the kernels that take a large share of a CU's LDS (k_p25_lsd 49 KB, k_p25_half_rate_list 80 KB per
    // workgroup): in the pipelined forms the next call's front end (153 KB per CU) runs beside this decode, and a workgroup that needs
    // more LDS than the front end leaves (10 KB) waits for one of its workgroups to end - i.e. ~2 ms, with everything queued behind
    // it.  Everything above fits beside the front end (k_rs63 7.6 KB, k_golay24 8 KB).
    ddn_sel_set(c->d_lists + (size_t)DDN_LIST_LSD * S, c->d_list_n + DDN_LIST_LSD);
    DDN_TRY(ddn_p25p1_framer_gather_lsd(c->fr, rec, c->d_cnt_full[cur], stride, c->d_lsd, c->d_lsd_llr, c->d_vldu, st));
    DDN_TRY(ddn_fec_p25_lsd_batch(c->d_lsd, c->d_lsd_llr, S * 2, c->d_lsd_ok, st));
    ddn_sel_clear();
    // data units (DUID 0xC): the header the loop decoded + the data blocks behind it (half-rate trellis, best path) + CRC32
    {
        const int32_t *d_ns = nullptr, *d_sp = nullptr;
        DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &d_ns, &d_sp));
        const size_t NE = (size_t)c->B * (size_t)c->PF, NB = NE * (size_t)c->PB;
        HIP_TRY(ddn_dev_chain_pdu_index(c->d_evl[cur], c->d_evdl[cur], c->d_nevl[cur], c->EL, d_sp, d_ns, c->d_nid[cur], c->B, c->F, c->off97[0],
                                        c->PF, c->d_pdu_slot, c->d_pdu_hdr, c->d_pdu_info, c->d_n_pdu, st));
        HIP_TRY(ddn_dev_chain_pdu_gather(rec, c->d_cnt_full[cur], stride, d_sp, c->d_pdu_slot, c->d_pdu_info, c->B, c->F, c->PF, c->PB,
                                         c->d_pdu_llr, c->d_pdu_valid, st));
        // (d_pdu_cand: scratch for the 1/2-rate candidates first, then the rate 3/4 ones - same stream)
        // (only the blocks that lie inside the call's records: groups of 32 without one leave at once)
        HIP_TRY(ddn_dev_p25_half_rate_list_wanted(c->d_pdu_llr, (int)NB, 8, c->d_pdu_valid, (uint32_t*)c->d_pdu_cand, c->d_pdu_cnt, st));
        HIP_TRY(ddn_dev_chain_pdu_take_first(c->d_pdu_cand, c->d_pdu_cnt, (int)NB, c->d_pdu_blocks, c->d_pdu_metric, st));
        // confirmed data: the same blocks through the rate 3/4 LLR list decoder, first candidate with a good CRC9 (:219-241)
        HIP_TRY(ddn_dev_chain_pdu_r34_wanted(c->d_pdu_slot, c->d_pdu_hdr, c->d_pdu_info, c->d_pdu_valid, (int)NB, c->PB, c->d_pdu_wanted, st));
        DDN_TRY(ddn_fec_p25_mbf34_list_batch(c->d_pdu_llr, NB, 8, c->d_pdu_wanted, (ddn_p25_mbf34_candidate*)c->d_pdu_cand, c->d_pdu_cnt, st));
        HIP_TRY(ddn_dev_chain_pdu_r34_select(c->d_pdu_cand, c->d_pdu_cnt, c->d_pdu_wanted, (int)NB, c->d_pdu_blocks18, c->d_pdu_crc9, st));
        // a header that failed its CRC16 three times over: the repetitions' LLRs summed through the list decoder, then the bitwise
        // majority (p25_mpdu_finalize_header :381-410); the candidate / count / wanted scratch is free again here
        HIP_TRY(ddn_dev_chain_pdu_combine(rec, c->d_cnt_full[cur], stride, d_sp, c->d_pdu_slot, c->d_pdu_info, c->d_pdu_llr, c->d_pdu_valid,
                                          c->d_pdu_blocks, (int)NE, c->PF, c->PB, c->d_pdu_hllr, c->d_pdu_wanted, st));
        HIP_TRY(ddn_dev_p25_half_rate_list_wanted(c->d_pdu_hllr, (int)NE, 8, c->d_pdu_wanted, (uint32_t*)c->d_pdu_cand, c->d_pdu_cnt, st));
        HIP_TRY(ddn_dev_chain_pdu_finish(c->d_pdu_slot, c->d_pdu_blocks, c->d_pdu_valid, c->d_pdu_blocks18, c->d_pdu_cand, c->d_pdu_cnt,
                                         c->d_pdu_wanted, (int)NE, c->PB, c->d_pdu_hdr, c->d_pdu_info, st));
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[4], st));
    }
    if (c->want_rec2) { // _run_host with records2: the records' host form, packed first thing on the voice stream - beside the front
                        // end of the next call, not beside its receive loop (measured there: the loop 6.8 ms instead of 5.3)
        HIP_TRY(ddn_dev_chain_pack2(c->d_rec[cur], c->d_fl[cur], (size_t)c->B * c->stride, c->d_rec2[cur], vst));
        c->want_rec2 = 0;
    }
    // voice: nine IMBE frames per LDU
    DDN_TRY(ddn_p25p1_framer_voice_index(c->fr, c->d_nid[cur], c->d_cnt_full[cur], c->Fv, stride, c->d_first, c->d_sc, c->d_nldu, vst));
    DDN_TRY(ddn_p25p1_imbe_deinterleave_batch(rec, (size_t)c->B * stride, c->d_first, c->d_sc, V, c->d_imbe_fr, c->d_imbe_soft,
                                              c->d_imbe_fl, c->d_sc_out, vst));
    DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_IMBE_7200X4400, c->d_imbe_fr, nullptr, V, c->d_imbe_d, c->d_imbe_res, vst));
    DDN_TRY(ddn_mbe_result_skip_batch(c->d_imbe_fl, V, c->d_imbe_res, vst));
    if (ev_pre) { // frame FEC (st) + the voice stage so far (vst): what the next call's receive loop waits for
        if (vst != st) {
            HIP_TRY(hipEventRecord(c->ev_join_a, vst));
            HIP_TRY(hipStreamWaitEvent(st, c->ev_join_a, 0));
        }
        HIP_TRY(hipEventRecord(ev_pre, st));
    }
    if (defer_synth && vst != st && ev_pre && c->cfg.vocoder) { // the synthesis kernel: queued by chain_issue_deferred (the next call)
        // the parameter kernel stays here, ahead of the loop: one wave per talk path with the frame-to-frame recurrence - it is
        // latency-bound and lives on occupancy; beside the loop one wave per SIMD fits and it took 5.5 ms instead of 0.65
        DDN_TRY(ddn_mbe_params_only(c->mbe, c->d_imbe_d, c->d_imbe_res, (size_t)c->Fv * 9, c->d_res_out, vst));
        HIP_TRY(hipEventRecord(c->ev_join_a, vst));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join_a, 0));
        HIP_TRY(hipEventRecord(ev_pre, st)); // (again: the loop of the next call also waits for the parameter kernel)
        c->have_deferred = 1;
        c->deferred_set = cur;
        c->deferred_dense = 0;
        return DDN_OK;
    }
    if (c->cfg.vocoder) {
        DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_imbe_d, c->d_imbe_res, (size_t)c->Fv * 9, c->d_pcm[cur], c->d_res_out, vst));
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

// the deferred tail of a pipelined call's decode: parameter + synthesis kernels (and the dense-PCM pack of _run_host) on the voice
// stream, behind that call's frame FEC and - when `gate` is given - behind the event the CURRENT call recorded right before its
// loop kernel, so that they run beside the loop (they need no LDS to speak of; the loop leaves 84 registers per SIMD); then the
// set's "consumed" event
static int
chain_issue_deferred(ddn_p25_chain* c, hipEvent_t gate) {
    if (!c->have_deferred) {
        return DDN_OK;
    }
    const int set = c->deferred_set;
    c->have_deferred = 0;
    HIP_TRY(hipStreamWaitEvent(c->s_voice, c->ev_pre[set], 0));
    if (gate) {
        HIP_TRY(hipStreamWaitEvent(c->s_voice, gate, 0));
    }
    if (c->cfg.vocoder) {
        DDN_TRY(ddn_mbe_synth_only(c->mbe, (size_t)c->Fv * 9, c->d_pcm[set], c->s_voice));
        if (c->deferred_dense) {
            const size_t V = c->V;
            HIP_TRY(ddn_dev_chain_pcm_compact(c->d_imbe_res, c->d_pcm[set], (int)V, (long)V, c->d_pcm_bcnt, c->d_pcm_boff, c->d_pcm_dense[set],
                                              c->d_pcm_slot[set], c->d_pcm_total[set], c->s_voice));
        }
    }
    HIP_TRY(hipStreamWaitEvent(c->s_voice, c->ev_aux_done[set], 0)); // (the pack kernels _run_host put on the decode stream)
    HIP_TRY(hipEventRecord(c->ev_consumed[set], c->s_voice));
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
ddn_p25_chain_set_first_channel(ddn_p25_chain* c, int first) {
    if (!c || first < 0) {
        return DDN_EINVAL;
    }
    if (!c->mbe) {
        return DDN_OK; // no vocoder: nothing here depends on a channel's number
    }
    const int rc = ddn_mbe_batch_set_first_stream(c->mbe, (uint32_t)first, nullptr);
    return rc != DDN_OK ? rc : (hipDeviceSynchronize() == hipSuccess ? DDN_OK : DDN_EHIP);
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
    const int cur = (int)(c->step % NSET);
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
    const int cur = (int)(c->step % NSET);
    int rc = DDN_OK;
    if (stage == 0) {
        DDN_TRY(chain_settle_pending(c, st));
        rc = chain_front(c, d_iq, cur, st, c->d_disc2 == nullptr);
    } else if (stage == 1) {
        if (c->d_disc2) {
            DDN_TRY(chain_carry(c, cur, st));
        }
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

// (internal, the mixed chain's shared front end) stage 0 without the front end itself: what has to be on `st` ahead of a front end
// launch that writes this call's discriminator rows to *disc_out ([B][samples_per_call] f32)
extern "C" int
ddn_p25_chain_stage0_prepare(ddn_p25_chain* c, void* hip_stream, float** disc_out) {
    if (!c || !disc_out || c->cq) {
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int cur = (int)(c->step % NSET);
    DDN_TRY(chain_settle_pending(c, st));
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev_t[0], st));
    }
    if (c->d_disc2 == nullptr) {
        DDN_TRY(chain_carry(c, cur, st));
    }
    *disc_out = chain_disc(c);
    HIP_TRY(hipEventRecord(c->ev_user, st));
    c->have_user_stream = 1;
    return DDN_OK;
}

// (internal, the mixed chain) a second discriminator buffer: the staged form then keeps the front end (stage 0) free of anything the
// previous call's loop writes - the carried tail is copied at the head of stage 1 - and alternates the buffer by step
extern "C" int
ddn_p25_chain_double_disc(ddn_p25_chain* c) {
    if (!c || c->cq) {
        return DDN_EINVAL;
    }
    if (!c->d_disc2) {
        HIP_TRY(hipMalloc((void**)&c->d_disc2, sizeof(float) * (size_t)c->B * (size_t)c->n));
    }
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_run_pipelined(ddn_p25_chain* c, const void* d_iq) {
    if (!c || !d_iq) {
        return DDN_EINVAL;
    }
    const int cur = (int)(c->step % NSET);
    const bool beside = c->synth_beside_loop != 0 && c->cfg.vocoder && !c->timing;
    if (c->step >= NSET) { // the decode of call k - NSET has read this set (and call k - 1's decode has read its carried tail source);
                        // its synthesis kernel, when deferred, reads none of the loop's outputs and may still be running
        HIP_TRY(hipStreamWaitEvent(c->s_main, beside ? c->ev_aux_done[cur] : c->ev_consumed[cur], 0));
    }
    DDN_TRY(chain_receive(c, d_iq, cur, c->s_main, c->step >= 1 ? (beside ? c->ev_pre[set_prev(cur)] : c->ev_consumed[set_prev(cur)]) : nullptr,
                          c->ev_loop[cur]));
    HIP_TRY(hipEventRecord(c->ev_produced[cur], c->s_main));
    DDN_TRY(chain_issue_deferred(c, c->ev_loop[cur])); // the previous call's synthesis: beside this call's loop
    // a _run_host call before this one may still owe its result copies: they go now (behind that call's "consumed" event, which the
    // line above has just recorded if its synthesis was deferred)
    DDN_TRY(chain_settle_pending(c, c->s_aux, false));
    HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_produced[cur], 0));
    DDN_TRY(chain_decode(c, cur, 0, c->s_aux, c->ev_pre[cur], beside));
    HIP_TRY(hipEventRecord(c->ev_aux_done[cur], c->s_aux));
    if (!c->have_deferred) {
        HIP_TRY(hipEventRecord(c->ev_consumed[cur], c->s_aux));
    }
    c->last_set = cur;
    c->step++;
    return DDN_OK;
}

extern "C" int
ddn_p25_chain_run_host(ddn_p25_chain* c, const void* h_iq, const ddn_p25_chain_host_out* out) {
    if (!c || !h_iq) {
        return DDN_EINVAL;
    }
    const int cur = (int)(c->step % NSET);
    if (!c->d_iq[0]) {
        HIP_TRY(hipMalloc(&c->d_iq[0], c->iq_bytes + 16));
        HIP_TRY(hipMalloc(&c->d_iq[1], c->iq_bytes + 16));
    }
    if (out && c->sdma == 0) { // the first call that names result buffers decides the copy route
        const void* any = out->records10 ? (const void*)out->records10
                          : (out->records2 ? (const void*)out->records2
                             : (out->pcm ? (const void*)out->pcm : (out->nid4 ? (const void*)out->nid4 : (const void*)out->counts)));
        if (any) {
            (void)sdma_setup(c, any);
        }
    }
    const bool engine = c->sdma > 0;
    const int ic = (int)(c->step & 1); // input buffer (two, used in turn)
    if (c->step >= 2) {
        HIP_TRY(hipStreamWaitEvent(c->s_copy, c->ev_in_free[ic], 0)); // the front end of call k - 2 has read this input buffer
    }
    HIP_TRY(hipMemcpyAsync(c->d_iq[ic], h_iq, c->iq_bytes, hipMemcpyHostToDevice, c->s_copy));
    HIP_TRY(hipEventRecord(c->ev_in[ic], c->s_copy));
    // The header's contract, kept on the host side (stream-to-stream waits alone do not), once this call's input copy is queued: the
    // previous call's h_iq has left the host before this call returns, and the results of the call NSET calls back - whose buffer set
    // this call's loop is about to overwrite - are in the caller's buffers.  This also bounds what a host that never calls _wait can
    // have queued.
    if (c->step >= 1) {
        HIP_TRY(hipEventSynchronize(c->ev_in[ic ^ 1]));
    }
    if (c->step >= NSET) {
        if (engine) {
            sdma_wait(c, cur);
        } else {
            HIP_TRY(hipEventSynchronize(c->ev_out[cur]));
        }
    }
    HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_in[ic], 0));
    const bool beside = c->synth_beside_loop != 0 && c->cfg.vocoder && !c->timing;
    if (c->step >= NSET) {
        // call k - NSET decoded out of this set (its deferred synthesis kernel reads none of the loop's outputs) ...
        HIP_TRY(hipStreamWaitEvent(c->s_main, beside ? c->ev_aux_done[cur] : c->ev_consumed[cur], 0));
        if (!engine) {
            HIP_TRY(hipStreamWaitEvent(c->s_main, c->ev_out[cur], 0)); // ... and its results have left it (engine: waited above)
        }
    }
    DDN_TRY(chain_receive(c, c->d_iq[ic], cur, c->s_main, c->step >= 1 ? (beside ? c->ev_pre[set_prev(cur)] : c->ev_consumed[set_prev(cur)]) : nullptr,
                          c->ev_loop[cur]));
    HIP_TRY(hipEventRecord(c->ev_in_free[ic], c->s_main));
    HIP_TRY(hipEventRecord(c->ev_produced[cur], c->s_main));
    DDN_TRY(chain_issue_deferred(c, c->ev_loop[cur])); // the previous call's synthesis (+ dense-PCM pack): beside this call's loop
    // shader copies: the previous call's results leave now, beside this call's receive loop (engine copies: at the end of this call)
    if (!engine) {
        DDN_TRY(chain_issue_pending(c, c->ev_loop[cur]));
    }
    HIP_TRY(hipStreamWaitEvent(c->s_aux, c->ev_produced[cur], 0));
    if (out && out->records2) { // the records' host form (packed inside chain_decode)
        if (!c->d_rec2[cur]) {
            HIP_TRY(hipMalloc(&c->d_rec2[cur], (size_t)c->B * c->stride * 2));
        }
        c->want_rec2 = 1;
    }
    DDN_TRY(chain_decode(c, cur, 0, c->s_aux, c->ev_pre[cur], beside));
    const bool dense_pcm = out && out->pcm_dense && out->pcm_slot && out->pcm_count && out->pcm_dense_frames > 0 && c->cfg.vocoder;
    if (dense_pcm) { // the synthesized frames only, compacted beside the decode stage
        const size_t V = c->V;
        if (!c->d_pcm_bcnt) {
            HIP_TRY(hipMalloc(&c->d_pcm_bcnt, ((V + 1023) / 1024) * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&c->d_pcm_boff, ((V + 1023) / 1024) * sizeof(int32_t)));
        }
        if (!c->d_pcm_dense[cur]) {
            HIP_TRY(hipMalloc(&c->d_pcm_dense[cur], V * 160 * sizeof(float)));
            HIP_TRY(hipMalloc(&c->d_pcm_slot[cur], V * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&c->d_pcm_total[cur], sizeof(int32_t)));
        }
        if (c->have_deferred) { // (behind the deferred synthesis, on its stream)
            c->deferred_dense = 1;
        } else {
            HIP_TRY(ddn_dev_chain_pcm_compact(c->d_imbe_res, c->d_pcm[cur], (int)V, (long)V, c->d_pcm_bcnt, c->d_pcm_boff, c->d_pcm_dense[cur],
                                              c->d_pcm_slot[cur], c->d_pcm_total[cur], c->s_aux));
        }
    }
    HIP_TRY(hipEventRecord(c->ev_aux_done[cur], c->s_aux));
    if (!c->have_deferred) {
        HIP_TRY(hipEventRecord(c->ev_consumed[cur], c->s_aux));
    }
    if (engine) {
        // the previous call's results go to the worker: it waits for that call's decode (running beside this call's front end, or
        // done) and hands them to the SDMA engine - they leave beside whatever the device runs next
        DDN_TRY(chain_issue_pending(c, nullptr));
    }
    if (out) { // this call's results: copied out beside the next call's loop, or when _wait / _flush asks
        c->pending_out = *out;
        c->pending_set = cur;
        c->have_pending = 1;
    } else { // nothing to leave: the set is free once its decode stage is through
        HIP_TRY(hipStreamWaitEvent(c->s_copy2, c->ev_aux_done[cur], 0));
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
    const int cur = (int)(c->step % NSET), prev = set_prev(cur);
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
    DDN_TRY(chain_issue_deferred(c, nullptr));
    DDN_TRY(chain_issue_pending(c, nullptr));
    if (c->have_user_stream) {
        HIP_TRY(hipEventSynchronize(c->ev_user));
    }
    HIP_TRY(hipStreamSynchronize(c->s_main));
    HIP_TRY(hipStreamSynchronize(c->s_aux));
    HIP_TRY(hipStreamSynchronize(c->s_voice));
    HIP_TRY(hipStreamSynchronize(c->s_copy));
    HIP_TRY(hipStreamSynchronize(c->s_copy2));
    for (int k = 0; k < NSET; k++) {
        sdma_wait(c, k);
    }
    return out_take_rc(c); // a failed engine copy of the last sets is this call's error, not a later one's
}

extern "C" int
ddn_p25_chain_get_results(ddn_p25_chain* c, ddn_p25_chain_results* r) {
    if (!c || !r) {
        return DDN_EINVAL;
    }
    DDN_TRY(chain_issue_deferred(c, nullptr)); // (a pipelined call's synthesis is queued by the next call at the latest; or here)
    const int cur = c->last_set;
    memset(r, 0, sizeof(*r));
    r->stride_symbols = c->stride;
    r->d_records10 = c->d_rec[cur];
    r->d_flags = c->d_fl[cur];
    r->d_counts = c->d_cnt_full[cur];
    r->d_new = c->d_new[cur];
    r->d_events = c->d_ev[cur];
    r->d_n_events = c->d_nev[cur];
    r->d_event_data = c->d_evd[cur];
    DDN_TRY(ddn_p25p1_framer_device_syncs(c->fr, &r->d_n_syncs, &r->d_sync_pos));
    DDN_TRY(ddn_p25p1_framer_device_dropped(c->fr, &r->d_dropped_syncs));
    r->d_nid4 = c->d_nid[cur];
    r->d_tsbk = c->d_tsbk[cur];
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
    r->d_pcm = c->d_pcm[cur];
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
extern "C" int
ddn_p25_chain_d2h_route(const ddn_p25_chain* c) {
    return (c && c->sdma > 0) ? (c->sdma_engine ? (int)c->sdma_engine : 0x10000) : 0;
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
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocPortable));
    return DDN_OK;
}
extern "C" void
ddn_host_free_pinned(void* p) {
    (void)hipHostFree(p);
}
