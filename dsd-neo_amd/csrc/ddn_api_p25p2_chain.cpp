// ddn_api_p25p2_chain.cpp - the P25 Phase 2 chain object (include/ddn_chain.h): cu8 / cf32 I/Q of B TDMA channels -> CQPSK demodulator at
// 6000 symbols/s -> symbol-rate receive loop (S-ISCH sync incl. the rotated constellations, 700 in-frame dibits per sync) -> the groups
// behind every sync -> processP2(): I-ISCH, scramble offset, DUID dispatch, FACCH / SACCH / LCCH bursts with RS(63,35) + MAC CRCs, 4V / 2V
// voice + ESS -> the AMBE 3600x2450 frames of both logical channels through frame FEC and synthesis.  One call per batch of
// samples_per_call samples; state (demodulator loops, receive loop, scramble offset, 4V counters, ESS fragments, both vocoders) carries
// from call to call, and a group that crosses a call boundary is decoded whole in the next call (the records' carried tail).  Host-only.
//
// What it stands in for in a dsd-neo host: the demodulator thread's blocks (src/io/radio/rtl_sdr_fm.cpp:3458-3516, CQPSK branch) and
// processFrame()'s Phase 2 branch -> processP2() (src/protocol/p25/phase2/p25p2_frame.c:1760-1798) with the vocoder calls of process_4V /
// process_2V (:1029-1047,1435-1460 -> src/core/vocoder/dsd_mbe.c:172-190), B channels wide.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_chain.h"
#include "ddn_device.h"
#include "ddn_hip.h"
#include "ddn_internal.h"
#include "ddn_mbe.h"
#include "ddn_p25p2_seq.h"

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

struct ddn_p25p2_chain {
    ddn_p25p2_chain_config cfg;
    int B, n, T, G, cap;
    size_t ms, stride;
    ddn_cqpsk_batch* fe;
    ddn_cq_rx* rx;
    ddn_mbe_batch* mbe;
    float* d_sym;
    int32_t* d_sym_cnt;
    uint8_t *d_rec[2], *d_fl[2];
    int32_t* d_new[2];
    int32_t *d_cnt_scan, *d_cnt_full, *d_sync_pos, *d_n_sync, *d_dropped;
    uint8_t* d_bits;
    int16_t* d_llr;
    uint64_t* d_seed;
    ddn_p25p2_seq_state* d_state;
    int32_t* d_info;
    uint8_t *d_payload, *d_ambe_fr, *d_ambe_rel, *d_ess;
    // voice
    int32_t *d_vsrc, *d_vcount, *d_vres, *d_vres_out;
    uint8_t *d_vfr, *d_vrel, *d_vskip, *d_vbits;
    float* d_pcm;
    long step;
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
ddn_p25p2_chain_destroy(ddn_p25p2_chain* c) {
    if (!c) {
        return;
    }
    (void)hipDeviceSynchronize();
    ddn_cqpsk_batch_destroy(c->fe);
    ddn_cq_rx_destroy(c->rx);
    ddn_mbe_batch_destroy(c->mbe);
    void* all[] = {c->d_sym, c->d_sym_cnt, c->d_rec[0], c->d_rec[1], c->d_fl[0], c->d_fl[1], c->d_new[0], c->d_new[1], c->d_cnt_scan, c->d_cnt_full,
                   c->d_sync_pos, c->d_n_sync, c->d_dropped, c->d_bits, c->d_llr, c->d_seed, c->d_state, c->d_info, c->d_payload, c->d_ambe_fr,
                   c->d_ambe_rel, c->d_ess, c->d_vsrc, c->d_vcount, c->d_vres, c->d_vres_out, c->d_vfr, c->d_vrel, c->d_vskip, c->d_vbits, c->d_pcm};
    for (void* p : all) {
        (void)hipFree(p);
    }
    delete c;
}

extern "C" int
ddn_p25p2_chain_create(const ddn_p25p2_chain_config* cfg, const uint64_t* seed44, ddn_p25p2_chain** out) {
    if (!cfg || !out || !seed44 || cfg->n_channels <= 0 || cfg->samples_per_call <= 0 || cfg->block_len <= 0) {
        ddn_set_error("ddn_p25p2_chain_create: bad configuration");
        return DDN_EINVAL;
    }
    *out = nullptr;
    ddn_p25p2_chain* c = new (std::nothrow) ddn_p25p2_chain();
    if (!c) {
        return DDN_ENOMEM;
    }
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->B = cfg->n_channels;
    c->n = cfg->samples_per_call;
    c->T = 720; // a group = 20 sync dibits + 700: a sync is decoded in the call that brings the 720 records behind it
    const int rate = cfg->sample_rate_hz > 0 ? cfg->sample_rate_hz : 48000;
    c->G = cfg->max_groups > 0 ? cfg->max_groups : (int)((long)cfg->samples_per_call * 6000 / rate / 720) + 3;
    c->cap = 8 * c->G; // AMBE frames per logical channel and call: every other timeslot of a group, four frames each
    int rc = DDN_OK;
    do {
        ddn_cqpsk_config qc = {c->B, rate, 6000, DDN_LPF_P25_CQPSK, 1, cfg->input_format, cfg->block_len, 0.0f};
        if ((rc = ddn_cqpsk_batch_create(&qc, &c->fe)) != DDN_OK) {
            break;
        }
        ddn_cq_rx_config rq = {c->B, DDN_CQ_P25P2, 0, 0, cfg->snr_cqpsk_db};
        if ((rc = ddn_cq_rx_create(&rq, &c->rx)) != DDN_OK) {
            break;
        }
        if (cfg->vocoder && (rc = ddn_mbe_batch_create(DDN_MBE_AMBE_3600X2450, 2 * c->B, &c->mbe)) != DDN_OK) {
            break;
        }
        c->ms = ddn_cqpsk_max_symbols(c->fe, (size_t)c->n);
        c->stride = (size_t)c->T + c->ms;
        const size_t B = (size_t)c->B, R = B * (size_t)c->G * 4, V = 2 * B * (size_t)c->cap;
        bool ok = dalloc(&c->d_sym, B * c->ms) && dalloc(&c->d_sym_cnt, B) && dalloc(&c->d_cnt_scan, B) && dalloc(&c->d_cnt_full, B)
                  && dalloc(&c->d_sync_pos, B * (size_t)c->G) && dalloc(&c->d_n_sync, B) && dalloc(&c->d_dropped, B)
                  && dalloc(&c->d_bits, B * (size_t)c->G * 1400) && dalloc(&c->d_llr, B * (size_t)c->G * 1400) && dalloc(&c->d_seed, B)
                  && dalloc(&c->d_state, B) && dalloc(&c->d_info, R * 8) && dalloc(&c->d_payload, R * 180) && dalloc(&c->d_ambe_fr, R * 384)
                  && dalloc(&c->d_ambe_rel, R * 384) && dalloc(&c->d_ess, R * 96) && dalloc(&c->d_vsrc, V) && dalloc(&c->d_vcount, 2 * B)
                  && dalloc(&c->d_vres, V * 5) && dalloc(&c->d_vres_out, V * 5) && dalloc(&c->d_vfr, V * 96) && dalloc(&c->d_vrel, V * 96)
                  && dalloc(&c->d_vskip, V) && dalloc(&c->d_vbits, V * 49) && dalloc(&c->d_pcm, V * 160);
        for (int k = 0; k < 2 && ok; k++) {
            ok = dalloc(&c->d_rec[k], B * c->stride * 10) && dalloc(&c->d_fl[k], B * c->stride) && dalloc(&c->d_new[k], B);
        }
        if (!ok) {
            ddn_set_error("ddn_p25p2_chain_create: device allocation failed");
            rc = DDN_ENOMEM;
            break;
        }
        if (hipMemcpy(c->d_seed, seed44, sizeof(uint64_t) * B, hipMemcpyHostToDevice) != hipSuccess) {
            rc = DDN_EHIP;
        }
    } while (0);
    if (rc != DDN_OK) {
        ddn_p25p2_chain_destroy(c);
        return rc;
    }
    *out = c;
    return DDN_OK;
}

// groups of the records in set `cur` (syncs inside the scan range) -> processP2() -> voice
static int
p2_decode(ddn_p25p2_chain* c, int cur, int flush, hipStream_t st) {
    const size_t R = (size_t)c->B * (size_t)c->G * 4;
    HIP_TRY(ddn_dev_chain_counts(c->d_new[cur], c->T, c->B, 0, c->d_cnt_scan, c->d_cnt_full, st));
    if (flush) { // no new records: the carried 720 hold at most one more whole group, behind a sync among their first 20
        HIP_TRY(ddn_dev_fill_words(c->d_cnt_scan, c->B, c->T - 700, st));
    }
    HIP_TRY(ddn_dev_find_syncs(c->d_fl[cur], c->d_cnt_scan, c->B, c->stride, c->G, c->d_sync_pos, c->d_n_sync, c->d_dropped, st));
    HIP_TRY(ddn_dev_p2_cut_records(c->d_rec[cur], c->stride, c->d_sync_pos, c->d_n_sync, c->B, c->G, c->d_bits, c->d_llr, st));
    // rows no decoder writes keep what is there: clear the result arrays of this call first
    HIP_TRY(hipMemsetAsync(c->d_payload, 0, R * 180, st));
    HIP_TRY(hipMemsetAsync(c->d_ambe_fr, 0, R * 384, st));
    HIP_TRY(hipMemsetAsync(c->d_ambe_rel, 0, R * 384, st));
    HIP_TRY(hipMemsetAsync(c->d_ess, 0, R * 96, st));
    DDN_TRY(ddn_p25p2_groups_batch(c->d_bits, c->d_llr, c->B, c->G, c->d_n_sync, c->d_seed, c->d_state, 64, c->d_info, c->d_payload, c->d_ambe_fr,
                                   c->d_ambe_rel, c->d_ess, st));
    if (c->mbe) {
        const size_t V = 2 * (size_t)c->B * (size_t)c->cap;
        HIP_TRY(ddn_dev_p2_voice_gather(c->d_info, c->d_n_sync, c->B, c->G, c->cap, c->d_ambe_fr, c->d_ambe_rel, c->d_vsrc, c->d_vcount, c->d_vfr,
                                        c->d_vrel, c->d_vskip, st));
        DDN_TRY(ddn_mbe_frame_decode_batch(DDN_MBE_AMBE_3600X2450, c->d_vfr, c->d_vrel, V, c->d_vbits, c->d_vres, st));
        DDN_TRY(ddn_mbe_result_skip_batch(c->d_vskip, V, c->d_vres, st));
        DDN_TRY(ddn_mbe_synth_batch(c->mbe, c->d_vbits, c->d_vres, (size_t)c->cap, c->d_pcm, c->d_vres_out, st));
    }
    return DDN_OK;
}

extern "C" int
ddn_p25p2_chain_run(ddn_p25p2_chain* c, const void* d_iq, void* hip_stream) {
    if (!c || !d_iq) {
        return DDN_EINVAL;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const int cur = (int)(c->step & 1), prev = cur ^ 1;
    HIP_TRY(ddn_dev_chain_carry(c->d_rec[prev], c->d_fl[prev], c->d_new[prev], c->step > 0 ? 1 : 0, c->d_rec[cur], c->d_fl[cur], c->stride, c->T,
                                c->B, st));
    DDN_TRY(ddn_cqpsk_run(c->fe, d_iq, (size_t)c->n, c->d_sym, c->ms, c->d_sym_cnt, st));
    DDN_TRY(ddn_cq_rx_run(c->rx, c->d_sym, c->d_sym_cnt, c->ms, c->ms, c->d_rec[cur] + (size_t)c->T * 10, c->d_fl[cur] + c->T, c->d_new[cur],
                          c->stride, st));
    DDN_TRY(p2_decode(c, cur, 0, st));
    c->step++;
    return DDN_OK;
}

extern "C" int
ddn_p25p2_chain_flush(ddn_p25p2_chain* c, void* hip_stream) {
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
    DDN_TRY(p2_decode(c, cur, 1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemsetAsync(c->d_fl[cur], 0, (size_t)c->B * c->stride, st)); // what was flushed is not decoded again
    c->step++;
    return DDN_OK;
}

extern "C" int
ddn_p25p2_chain_get_results(ddn_p25p2_chain* c, ddn_p25p2_chain_results* r) {
    if (!c || !r || c->step == 0) {
        return DDN_EINVAL;
    }
    const int cur = (int)((c->step - 1) & 1);
    memset(r, 0, sizeof(*r));
    r->stride_symbols = c->stride;
    r->carry_symbols = c->T;
    r->max_groups = c->G;
    r->voice_frames = c->cap;
    r->d_records10 = c->d_rec[cur];
    r->d_flags = c->d_fl[cur];
    r->d_new = c->d_new[cur];
    r->d_counts = c->d_cnt_full;
    r->d_n_groups = c->d_n_sync;
    r->d_group_pos = c->d_sync_pos;
    r->d_dropped_syncs = c->d_dropped;
    r->d_info = c->d_info;
    r->d_payload = c->d_payload;
    r->d_ambe_fr = c->d_ambe_fr;
    r->d_ambe_rel = c->d_ambe_rel;
    r->d_ess = c->d_ess;
    r->d_voice_src = c->d_vsrc;
    r->d_voice_count = c->d_vcount;
    r->d_voice_bits = c->d_vbits;
    r->d_voice_result = c->d_vres_out;
    r->d_pcm = c->d_pcm;
    return DDN_OK;
}
