// ddn_api_p25p2.cpp - C-ABI of the P25 Phase 2 sequencing stage (include/ddn_hip.h: ddn_p25p2_groups_batch): processP2()'s work on
// the 700 dibits behind a sync, batched.  The stage strings the burst layer's own batched calls (ddn_api_fec.cpp) behind the
// sequencing pass of ddn_p25p2_seq.hip; nothing here computes on the CPU.
#include <hip/hip_runtime.h>

#include <new>

#include <cstdint>

#include "ddn_device.h"
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
        const int r_ = (expr);                                                                                         \
        if (r_ != DDN_OK) {                                                                                            \
            return r_;                                                                                                 \
        }                                                                                                              \
    } while (0)

namespace {
// stream-ordered scratch: arenas (one hipMallocAsync each, carved by get()), released when the call leaves (also on its error paths)
struct Scratch {
    hipStream_t st;
    void* p[4];
    int n = 0;
    uint8_t* cur = nullptr;
    size_t left = 0;
    explicit Scratch(hipStream_t s) : st(s) {}
    ~Scratch() {
        for (int i = 0; i < n; i++) {
            (void)hipFreeAsync(p[i], st);
        }
    }
    hipError_t arena(size_t bytes) {
        void* q = nullptr;
        const hipError_t e = n < 4 ? hipMallocAsync(&q, bytes + 256, st) : hipErrorOutOfMemory;
        if (e == hipSuccess) {
            p[n++] = q;
            cur = (uint8_t*)q;
            left = bytes + 256;
        }
        return e;
    }
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
    template <class T>
    hipError_t get(T** out, size_t count) {
        const size_t b = pad((count ? count : 1) * sizeof(T));
        if (b > left) {
            *out = nullptr;
            return hipErrorOutOfMemory;
        }
        *out = (T*)cur;
        cur += b;
        left -= b;
        return hipSuccess;
    }
};
// The decoder classes of a call are independent of each other and each is a chain of latency-bound launches (an RS decode is ~0.5 ms
// deep for its thread): FACCH, SACCH / LCCH and the voice bursts (with the ESS) run on three streams of the calling thread's own,
// side by side, between the call's host wait and its last kernel.
// They belong to ONE device: a host thread that drives several GPUs gets one set per device (aux_for_current_device).
struct Aux {
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t done[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr;
    int32_t* counts = nullptr; // pinned: the decoder lists' lengths come back without a staging copy
    bool ok = false;
    Aux() {
        ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess
             && hipHostMalloc((void**)&counts, 4 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
        for (int k = 0; k < 3; k++) {
            ok = ok && hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking) == hipSuccess
                 && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess;
        }
    }
    // no destructor: the objects live as long as the thread's first use of a device and are deliberately left to the runtime's own
    // teardown (a thread-exit destructor could run after the HIP runtime is gone)
};

static Aux*
aux_for_current_device() {
    enum { MAX_DEV = 64 };
    static thread_local Aux* per_dev[MAX_DEV] = {nullptr};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
        return nullptr; // the single-stream path
    }
    if (!per_dev[dev]) {
        per_dev[dev] = new (std::nothrow) Aux(); // created with `dev` current: its streams and events are that device's
    }
    return (per_dev[dev] && per_dev[dev]->ok) ? per_dev[dev] : nullptr;
}

// an early return (error path) must not free the arenas under work still queued on the side streams
struct AuxGuard {
    Aux* a;
    bool joined = false;
    explicit AuxGuard(Aux* x) : a(x) {}
    ~AuxGuard() {
        if (a && !joined) {
            for (int k = 0; k < 3; k++) {
                (void)hipStreamSynchronize(a->s[k]);
            }
        }
    }
};
} // namespace

extern "C" int
ddn_p25p2_groups_batch(const uint8_t* d_bits1400, const int16_t* d_llr1400, int n_channels, int n_groups, const int32_t* d_groups_of,
                       const uint64_t* d_seed44, ddn_p25p2_seq_state* d_state, int threshold, int32_t* d_info, uint8_t* d_payload, uint8_t* d_ambe_fr,
                       uint8_t* d_ambe_rel, uint8_t* d_ess, void* hip_stream) {
    if (!d_bits1400 || !d_llr1400 || !d_seed44 || !d_state || !d_info || !d_payload || !d_ambe_fr || !d_ambe_rel || !d_ess || n_channels < 0
        || n_groups < 0 || (size_t)n_channels * (size_t)n_groups * 4 > 0x7fffffffu / 360) {
        ddn_set_error("ddn_p25p2_groups_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n_channels == 0 || n_groups == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t n_gr = (size_t)n_channels * n_groups, n_rows = n_gr * 4;
    Scratch s(st);
    uint8_t *rb, *xb, *seq;
    int16_t *rl, *xl;
    int32_t *duid, *isch, *row_off, *counts, *list, *ess_src, *final_src, *seq_of;
    HIP_TRY(s.arena(n_rows * (360 * 6 + 4 * 4 + 16 + 16) + (size_t)n_channels * (4320 + 32) + 16 * 256));
    HIP_TRY(s.get(&rb, n_rows * 360));
    HIP_TRY(s.get(&rl, n_rows * 360));
    HIP_TRY(s.get(&xb, n_rows * 360));
    HIP_TRY(s.get(&xl, n_rows * 360));
    HIP_TRY(s.get(&seq, (size_t)n_channels * 4320));
    HIP_TRY(s.get(&duid, n_rows));
    HIP_TRY(s.get(&isch, n_rows));
    HIP_TRY(s.get(&row_off, n_rows));
    HIP_TRY(s.get(&counts, 4));
    HIP_TRY(s.get(&list, 4 * n_rows));
    HIP_TRY(s.get(&ess_src, n_rows * 4));
    HIP_TRY(s.get(&final_src, (size_t)n_channels * 8));
    HIP_TRY(s.get(&seq_of, n_rows));
    HIP_TRY(hipMemsetAsync(counts, 0, 16, st));
    // timeslot rows, their DUID / I-ISCH, the channels' scrambler sequences
    HIP_TRY(ddn_dev_p2_rows(d_bits1400, d_llr1400, n_gr, rb, rl, st));
    DDN_TRY(ddn_p25p2_burst_fields_batch(rb, rl, n_rows, threshold, duid, isch, st));
    DDN_TRY(ddn_p25p2_scramble_bits_batch(d_seed44, (size_t)n_channels, 4320, seq, st));
    // the sequencing pass: offsets, logical channels, actions, decoder lists
    HIP_TRY(ddn_dev_p2_sequence(duid, isch, n_channels, n_groups, d_groups_of, d_seed44, d_state, d_info, row_off, seq_of, counts, list, ess_src, final_src,
                                st));
    Aux* ax = aux_for_current_device();
    int32_t stack_counts[4] = {0, 0, 0, 0};
    int32_t* h_counts = ax ? ax->counts : stack_counts;
    HIP_TRY(hipMemcpyAsync(h_counts, counts, 16, hipMemcpyDeviceToHost, st));
    DDN_TRY(ddn_p25p2_descramble_batch(rb, rl, seq, row_off, seq_of, n_rows, 360, 360, xb, xl, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int n_f = h_counts[0], n_s = h_counts[1], n_4v = h_counts[2], n_2v = h_counts[3];
    const int rows_per_channel = n_groups * 4;
    HIP_TRY(s.arena((size_t)(n_f + n_s) * (360 * 3 + 180 + 8) + (size_t)(n_4v + n_2v) * (360 * 3 + 768) + (size_t)n_2v * (96 * 4 + 168 * 3 + 8)
                    + 40 * 256));
    AuxGuard guard(ax);
    if (ax) { // the arena exists (in the caller's stream's order) before the side streams touch it
        HIP_TRY(hipEventRecord(ax->fork, st));
        for (int k = 0; k < 3; k++) {
            HIP_TRY(hipStreamWaitEvent(ax->s[k], ax->fork, 0));
        }
    }
    // FACCH (class 0) and SACCH / LCCH (class 1) bursts
    for (int cls = 0; cls < 2; cls++) {
        const int cnt = cls == 0 ? n_f : n_s;
        if (cnt == 0) {
            continue;
        }
        hipStream_t cs = ax ? ax->s[cls] : st;
        const int n_pl = cls == 0 ? 156 : 180;
        uint8_t *db, *pl, *used, *c12, *c16;
        int16_t* dl;
        int32_t* ec;
        HIP_TRY(s.get(&db, (size_t)cnt * 360));
        HIP_TRY(s.get(&dl, (size_t)cnt * 360));
        HIP_TRY(s.get(&pl, (size_t)cnt * n_pl));
        HIP_TRY(s.get(&used, (size_t)cnt));
        HIP_TRY(s.get(&c12, (size_t)cnt));
        HIP_TRY(s.get(&c16, (size_t)cnt));
        HIP_TRY(s.get(&ec, (size_t)cnt));
        const int32_t* lst = list + (size_t)cls * n_rows;
        HIP_TRY(ddn_dev_p2_gather(cls, cnt, lst, d_info, rb, rl, xb, xl, db, dl, ess_src, d_state, rows_per_channel, nullptr, nullptr, nullptr,
                                  nullptr, cs));
        DDN_TRY(ddn_p25p2_xcch_batch(cls, db, dl, (size_t)cnt, threshold, pl, ec, used, cs));
        DDN_TRY(ddn_p25p2_mac_crc_batch(cls, pl, (size_t)cnt, c12, cls == 1 ? c16 : nullptr, cs));
        HIP_TRY(ddn_dev_p2_scatter(cls, cnt, lst, d_info, pl, n_pl, ec, used, c12, cls == 1 ? c16 : nullptr, nullptr, nullptr, 0, nullptr, d_payload,
                                   d_ambe_fr, d_ambe_rel, d_ess, cs));
    }
    // 4V (class 2) and 2V (class 3) bursts; the 2V bursts' ESS
    for (int cls = 2; cls < 4; cls++) {
        const int cnt = cls == 2 ? n_4v : n_2v;
        if (cnt == 0) {
            continue;
        }
        const int fc = cls == 2 ? 4 : 2;
        hipStream_t cs = ax ? ax->s[2] : st;
        uint8_t *db, *fr, *rel, *e_pl = nullptr, *e_pa = nullptr, *e_out = nullptr, *e_used = nullptr;
        int16_t *dl, *e_pll = nullptr, *e_pal = nullptr;
        int32_t* e_ec = nullptr;
        HIP_TRY(s.get(&db, (size_t)cnt * 360));
        HIP_TRY(s.get(&dl, (size_t)cnt * 360));
        HIP_TRY(s.get(&fr, (size_t)cnt * fc * 96));
        HIP_TRY(s.get(&rel, (size_t)cnt * fc * 96));
        if (cls == 3) {
            HIP_TRY(s.get(&e_pl, (size_t)cnt * 96));
            HIP_TRY(s.get(&e_pll, (size_t)cnt * 96));
            HIP_TRY(s.get(&e_pa, (size_t)cnt * 168));
            HIP_TRY(s.get(&e_pal, (size_t)cnt * 168));
            HIP_TRY(s.get(&e_out, (size_t)cnt * 96));
            HIP_TRY(s.get(&e_used, (size_t)cnt));
            HIP_TRY(s.get(&e_ec, (size_t)cnt));
        }
        const int32_t* lst = list + (size_t)cls * n_rows;
        HIP_TRY(ddn_dev_p2_gather(cls, cnt, lst, d_info, rb, rl, xb, xl, db, dl, ess_src, d_state, rows_per_channel, e_pl, e_pll, e_pa, e_pal, cs));
        DDN_TRY(ddn_p25p2_voice_frames_batch(db, dl, (size_t)cnt, fc, fr, rel, cs));
        if (cls == 3) {
            DDN_TRY(ddn_p25p2_ess_batch(e_pl, e_pll, e_pa, e_pal, (size_t)cnt, threshold, e_out, e_ec, e_used, cs));
        }
        HIP_TRY(ddn_dev_p2_scatter(cls, cnt, lst, d_info, nullptr, 0, e_ec, e_used, nullptr, nullptr, fr, rel, fc, e_out, d_payload, d_ambe_fr,
                                   d_ambe_rel, d_ess, cs));
    }
    if (ax) { // the caller's stream goes on behind the three
        for (int k = 0; k < 3; k++) {
            HIP_TRY(hipEventRecord(ax->done[k], ax->s[k]));
            HIP_TRY(hipStreamWaitEvent(st, ax->done[k], 0));
        }
        guard.joined = true;
    }
    // the carried ESS-B fragments (after every gather that still reads the old ones)
    HIP_TRY(ddn_dev_p2_state_ess(final_src, xb, xl, n_channels, d_state, st));
    return DDN_OK;
}

extern "C" int
ddn_p25p2_sync_cut_batch(const uint8_t* d_dibits, const int16_t* d_llr2, int n_channels, int n, size_t stride, const int32_t* d_cursor_in,
                         int max_groups, int32_t* d_n_groups, int32_t* d_group_pos, int32_t* d_cursor_out, uint8_t* d_bits1400, int16_t* d_llr1400,
                         void* hip_stream) {
    if (!d_dibits || !d_llr2 || !d_n_groups || !d_group_pos || !d_cursor_out || !d_bits1400 || !d_llr1400 || n_channels < 0 || n < 0
        || max_groups < 0 || stride < (size_t)n || max_groups > 65535) {
        ddn_set_error("ddn_p25p2_sync_cut_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n_channels == 0) {
        return DDN_OK;
    }
    HIP_TRY(ddn_dev_p2_sync_cut(d_dibits, d_llr2, n_channels, n, stride, d_cursor_in, max_groups, d_n_groups, d_group_pos, d_cursor_out,
                                d_bits1400, d_llr1400, (hipStream_t)hip_stream));
    return DDN_OK;
}

namespace {
struct Dev {
    void* p = nullptr;
    size_t bytes;
    explicit Dev(size_t b) : bytes(b) {
        if (hipMalloc(&p, b ? b : 4) != hipSuccess) {
            p = nullptr;
        }
    }
    ~Dev() { (void)hipFree(p); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    int up(const void* h) { return hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
    int down(void* h) { return hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
};

int
no_dev() {
    ddn_set_error("device allocation/copy failed (no HIP device?)");
    return DDN_ENODEV;
}
} // namespace

extern "C" int
ddn_p25p2_groups_host(const uint8_t* bits1400, const int16_t* llr1400, int n_channels, int n_groups, const int32_t* groups_of,
                      const uint64_t* seed44, ddn_p25p2_seq_state* state, int threshold, int32_t* info, uint8_t* payload, uint8_t* ambe_fr,
                      uint8_t* ambe_rel, uint8_t* ess) {
    if (!bits1400 || !llr1400 || !seed44 || !state || !info || !payload || !ambe_fr || !ambe_rel || !ess || n_channels < 0 || n_groups < 0) {
        ddn_set_error("ddn_p25p2_groups_host: bad argument");
        return DDN_EINVAL;
    }
    if (n_channels == 0 || n_groups == 0) {
        return DDN_OK;
    }
    const size_t ng = (size_t)n_channels * n_groups, nr = ng * 4;
    Dev b(ng * 1400), l(ng * 2800), go((size_t)n_channels * 4), sd((size_t)n_channels * 8), st((size_t)n_channels * sizeof(ddn_p25p2_seq_state));
    Dev oi(nr * 32), op(nr * 180), of(nr * 384), orl(nr * 384), oe(nr * 96);
    if (!b.p || !l.p || !go.p || !sd.p || !st.p || !oi.p || !op.p || !of.p || !orl.p || !oe.p || b.up(bits1400) || l.up(llr1400)
        || (groups_of && go.up(groups_of)) || sd.up(seed44) || st.up(state) || oi.up(info) || op.up(payload) || of.up(ambe_fr) || orl.up(ambe_rel)
        || oe.up(ess)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_groups_batch((const uint8_t*)b.p, (const int16_t*)l.p, n_channels, n_groups, groups_of ? (const int32_t*)go.p : nullptr,
                                          (const uint64_t*)sd.p, (ddn_p25p2_seq_state*)st.p, threshold, (int32_t*)oi.p, (uint8_t*)op.p,
                                          (uint8_t*)of.p, (uint8_t*)orl.p, (uint8_t*)oe.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (st.down(state) || oi.down(info) || op.down(payload) || of.down(ambe_fr) || orl.down(ambe_rel) || oe.down(ess)) {
        return no_dev();
    }
    return DDN_OK;
}

extern "C" int
ddn_p25p2_sync_cut_host(const uint8_t* dibits, const int16_t* llr2, int n_channels, int n, size_t stride, const int32_t* cursor_in, int max_groups,
                        int32_t* n_groups, int32_t* group_pos, int32_t* cursor_out, uint8_t* bits1400, int16_t* llr1400) {
    if (!dibits || !llr2 || !n_groups || !group_pos || !cursor_out || !bits1400 || !llr1400 || n_channels < 0 || n < 0 || max_groups < 0
        || stride < (size_t)n) {
        ddn_set_error("ddn_p25p2_sync_cut_host: bad argument");
        return DDN_EINVAL;
    }
    if (n_channels == 0) {
        return DDN_OK;
    }
    const size_t C = (size_t)n_channels, G = (size_t)max_groups;
    Dev d(C * stride), l(C * stride * 4), ci(C * 4), ng(C * 4), gp(C * G * 4), co(C * 4), gb(C * G * 1400), gl(C * G * 2800);
    if (!d.p || !l.p || !ci.p || !ng.p || !gp.p || !co.p || !gb.p || !gl.p || d.up(dibits) || l.up(llr2) || (cursor_in && ci.up(cursor_in))
        || gp.up(group_pos) || gb.up(bits1400) || gl.up(llr1400)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_sync_cut_batch((const uint8_t*)d.p, (const int16_t*)l.p, n_channels, n, stride, cursor_in ? (const int32_t*)ci.p : nullptr,
                                            max_groups, (int32_t*)ng.p, (int32_t*)gp.p, (int32_t*)co.p, (uint8_t*)gb.p, (int16_t*)gl.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (ng.down(n_groups) || gp.down(group_pos) || co.down(cursor_out) || gb.down(bits1400) || gl.down(llr1400)) {
        return no_dev();
    }
    return DDN_OK;
}
