// ddn_fsk4h_dev.h - the DMR and NXDN protocol handlers' hold on the receive loop (ddn_rx4.hip, k_fsk4_rx<.., HM = true>): how
// many symbols a frame is read in frame is decided by what the reference's handler decodes on the way.  Everything here works
// on hard dibits (inside a frame these protocols keep their thresholds still, so the recurrence lane slices them itself: three
// compares) and runs inline on the lane whose phase has ended - a few dozen integer instructions, a handful of times per burst.
//
//   DMR, plain -fs (dsd_dispatch_handle_dmr, src/engine/dispatch/dispatch_dmr.c:126-158)
//     BS data word   dmr_data_sync() (src/protocol/dmr/dmr_data.c:117-343): TACT Hamming(7,4) on the cached CACH (fails: no live
//                    dibit), 5 live dibits, slot type Golay(20,8) (fails: stop), colour-code gate, 49 more, then
//                    skipDibit(66) in every case
//     BS voice word  dmrBSBootstrap() + dmrBS() (src/protocol/dmr/dmr_bs.c:697-948): TACT and the sync word on the cached
//                    dibits, 54 live ones, then 144 per burst: TACT after 12, the repeated-carrier test after 48 (:216-222),
//                    and at the end the sync word (voice / data / neither), the frame-sync-miss counters (:311-335), QR(16,7,6)
//                    on the would-be EMB and the colour-code gate (:337-385), a data burst inside a call through
//                    dmr_data_sync() on the burst's own dibits (:286-309), the skip counter (:644-648)
//     colour-code gate  src/protocol/dmr/dmr_confidence.c
//     the other sync types (MS / direct mode, reverse channel) keep the configured count
//   NXDN  nxdn_frame() (src/protocol/nxdn/nxdn_frame.c:181-233,592-640): 8 LICH dibits de-scrambled (PN9 seed 228), parity and
//         profile table; accepted -> 174 more, rejected -> the handler returns and lastsynctype is cleared
//
// Code tables (syndrome -> positions to flip) are ddn_fec3.hip's per-device ones, built in the reference's init-loop order.
#ifndef DDN_FSK4H_DEV_H
#define DDN_FSK4H_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_fec3.h"
#include "ddn_tables_fec3.h"

namespace ddn_fsk4h {

// phases (DdnFsk4State.hmode)
enum { M_IDLE = 0, M_FIXED, M_DATA_SUFFIX, M_DATA_SECOND, M_SKIP66, M_BOOT54, M_BURST_CACH, M_BURST_RED, M_BURST_REST, M_NX_LICH, M_NX_REST };
enum { EV_NXDN_LICH = 4, EV_DMR_DATA = 5, EV_DMR_CC_PRINT = 6, EV_DMR_VOICE_BURST = 7, EV_DMR_VOICE_END = 8 };
// per-channel handler words kept in LDS (index = field, then lane)
enum {
    F_LOCKED = 0, F_CONF_CC, F_CAND_CC, F_CAND_COUNT, F_MISMATCH, F_VSEEN0, F_VSEEN1, F_VOPEN0, F_VOPEN1, F_VCOUNT0, F_VCOUNT1,
    F_COLOR, F_CURSLOT, F_VC1, F_VC2, F_SKIPCOUNT, F_EMBERR0, F_EMBERR1, F_TACT_OK, F_EMB_OK, F_ISLOT, F_BOOTSLOT, F_REJECT,
    F_PENDING, F_REDB, F_STEREO, F_NEV, F_COUNT = 32
};

template <int R>
__device__ __forceinline__ int
syndrome(uint32_t w, const uint32_t (&H)[R]) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < R; i++) {
        s |= (__popc(w & H[i]) & 1) << (R - 1 - i);
    }
    return s;
}

// Hamming_7_4_decode (fec.c:145-170) on a word with bit j = rxBits[j]; returns ok, *w corrected
__device__ __forceinline__ bool
hamming_7_4(uint32_t& w, const DdnFec3Tables* T) {
    const int s = syndrome<3>(w, ddn_hamming_7_4_H);
    if (s > 0) {
        const uint8_t p = T->h74[s];
        if (p == 0xFF) {
            return false;
        }
        w ^= 1u << p;
    }
    return true;
}
// Golay_20_8_decode (fec.c:514-561): up to two flips accepted (the flips of a three-position entry are applied, then refused)
__device__ __forceinline__ bool
golay_20_8(uint32_t& w, const DdnFec3Tables* T) {
    const int s = syndrome<12>(w, ddn_golay_20_8_H);
    if (s > 0) {
        int k = 0;
        for (; k < 3; k++) {
            const uint8_t p = T->g208[s][k];
            if (p == 0xFF) {
                break;
            }
            w ^= 1u << p;
        }
        return !(k == 0 || k > 2);
    }
    return true;
}
// QR_16_7_6_decode (fec.c:782-822)
__device__ __forceinline__ bool
qr_16_7_6(uint32_t& w, const DdnFec3Tables* T) {
    const int s = syndrome<9>(w, ddn_qr_16_7_6_H);
    if (s > 0) {
        int k = 0;
        for (; k < 2; k++) {
            const uint8_t p = T->qr[s][k];
            if (p == 0xFF) {
                break;
            }
            w ^= 1u << p;
        }
        return k != 0;
    }
    return true;
}

// The handler context of one lane: hs = its words in LDS (stride = lanes), pay = its burst dibits in LDS (stride = lanes)
struct Ctx {
    int* hs;
    uint8_t* pay;
    int stride;
    const DdnFec3Tables* T;
    int32_t* events; // this channel's event rows, or nullptr
    int max_events;
    __device__ __forceinline__ int& f(int field) const { return hs[field * stride]; }
    __device__ __forceinline__ int p(int i) const { return pay[i * stride]; }
    __device__ __forceinline__ void ev(int pos, int kind, int a, int b, int c) const {
        const int n = f(F_NEV);
        if (events && n < max_events) {
            int32_t* e = events + (size_t)n * 4;
            e[0] = pos;
            e[1] = kind;
            e[2] = a;
            e[3] = (b & 0xFFFF) | (c << 16);
        }
        f(F_NEV) = n + 1;
    }
};

// ---- dmr_confidence.c --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void
conf_clear_voice(const Ctx& x) {
    x.f(F_VSEEN0) = x.f(F_VSEEN1) = x.f(F_VOPEN0) = x.f(F_VOPEN1) = x.f(F_VCOUNT0) = x.f(F_VCOUNT1) = 0;
}
__device__ __forceinline__ void
conf_reset(const Ctx& x) {
    x.f(F_LOCKED) = 0;
    x.f(F_CONF_CC) = 16;
    x.f(F_CAND_CC) = 16;
    x.f(F_CAND_COUNT) = 0;
    x.f(F_MISMATCH) = 0;
    conf_clear_voice(x);
}
__device__ __forceinline__ void
conf_reset_slot(const Ctx& x, int slot) {
    x.f(F_VSEEN0 + slot) = 0;
    x.f(F_VOPEN0 + slot) = 0;
    x.f(F_VCOUNT0 + slot) = 0;
}
enum { CONF_REJECT = 0, CONF_PENDING = 1, CONF_LOCKED = 2 };
__device__ inline int
conf_observe(const Ctx& x, int cc, int may_lock) { // dmr_confidence_observe_cc(), :52-107
    if (cc < 0 || cc > 15) {
        return CONF_REJECT;
    }
    if (x.f(F_LOCKED)) {
        if (x.f(F_CONF_CC) == cc) {
            x.f(F_CAND_CC) = 16;
            x.f(F_CAND_COUNT) = 0;
            return CONF_LOCKED;
        }
        if (x.f(F_MISMATCH) < 255) {
            x.f(F_MISMATCH)++;
        }
        if (x.f(F_CAND_CC) != cc) {
            x.f(F_CAND_CC) = cc;
            x.f(F_CAND_COUNT) = 1;
        } else if (x.f(F_CAND_COUNT) < 255) {
            x.f(F_CAND_COUNT)++;
        }
        if (x.f(F_CAND_COUNT) >= 4) {
            x.f(F_CONF_CC) = cc;
            x.f(F_COLOR) = cc;
            x.f(F_CAND_CC) = 16;
            x.f(F_CAND_COUNT) = 0;
            x.f(F_MISMATCH) = 0;
            conf_clear_voice(x);
            return CONF_LOCKED;
        }
        return CONF_REJECT;
    }
    if (x.f(F_CAND_CC) != cc) {
        x.f(F_CAND_CC) = cc;
        x.f(F_CAND_COUNT) = 1;
    } else if (x.f(F_CAND_COUNT) < 255) {
        x.f(F_CAND_COUNT)++;
    }
    if (may_lock && x.f(F_CAND_COUNT) >= 2) {
        x.f(F_LOCKED) = 1;
        x.f(F_CONF_CC) = cc;
        x.f(F_COLOR) = cc;
        return CONF_LOCKED;
    }
    return CONF_PENDING;
}
__device__ __forceinline__ void
conf_note_voice_sync(const Ctx& x, int slot) {
    x.f(F_VSEEN0 + slot) = 1;
    if (!x.f(F_VOPEN0 + slot)) {
        x.f(F_VCOUNT0 + slot) = 0;
    }
}
__device__ inline int
conf_note_voice_burst(const Ctx& x, int slot, int cc) { // :120-148
    if (!x.f(F_VSEEN0 + slot) && !x.f(F_VOPEN0 + slot)) {
        return CONF_PENDING;
    }
    const int was_locked = x.f(F_LOCKED) != 0;
    const int r = conf_observe(x, cc, 1);
    if (r != CONF_LOCKED) {
        return r;
    }
    if (!was_locked && x.f(F_VSEEN0 + slot)) {
        x.f(F_VCOUNT0 + slot) = 2;
    } else if (x.f(F_VCOUNT0 + slot) < 255) {
        x.f(F_VCOUNT0 + slot)++;
    }
    if (x.f(F_VCOUNT0 + slot) >= 2) {
        x.f(F_VOPEN0 + slot) = 1;
        return CONF_LOCKED;
    }
    return CONF_PENDING;
}

// ---- burst fields from the burst's dibits -------------------------------------------------------------------------------------
__device__ inline bool
tact_decode(const Ctx& x, int& slot_out) { // CACH dibits 0..11 de-interleaved, TACT = its first seven bits
    // dmr_cach_interleave: position of the high / low bit of dibit i
    const uint64_t il_lo = 0x0C0B0A0109080700ull; // {0, 7, 8, 9, 1, 10, 11, 12} as bytes
    const uint64_t il_mid = 0x110410030F0E0D02ull; // {2, 13, 14, 15, 3, 16, 4, 17}
    const uint64_t il_hi = 0x1706161514051312ull; // {18, 19, 5, 20, 21, 22, 6, 23}
    uint32_t cach = 0;
    for (int i = 0; i < 12; i++) {
        const int d = x.p(i);
        for (int b = 0; b < 2; b++) {
            const int k = 2 * i + b;
            const uint64_t tab = k < 8 ? il_lo : (k < 16 ? il_mid : il_hi);
            const int pos = (int)((tab >> (8 * (k & 7))) & 0xFF);
            cach |= (uint32_t)((b == 0 ? (d >> 1) : d) & 1) << pos;
        }
    }
    uint32_t t = cach & 0x7Fu;
    if (!hamming_7_4(t, x.T)) {
        return false;
    }
    slot_out = (int)((t >> 1) & 1u);
    return true;
}
// 0 = neither, 1 = the BS voice word, 2 = the BS data word ((dibit | 1) + '0' against the pattern strings)
__device__ inline int
sync_kind(const Ctx& x) {
    // "131111333113313313113313" / "313333111331131131331131": bit k set where the pattern has '3'
    const uint32_t voice3 = 0xB2D9C2u, data3 = 0x4D263Du;
    uint32_t got = 0;
    for (int i = 0; i < 24; i++) {
        got |= (uint32_t)((x.p(66 + i) >> 1) & 1) << i; // (d | 1) = '3' <=> d >= 2
    }
    return got == voice3 ? 1 : (got == data3 ? 2 : 0);
}
// dmr_data_sync() on the burst's dibits once the slot type is complete (dibits 61..65 and 90..94); returns SlotTypeOk
__device__ inline bool
data_slot_type(const Ctx& x, int pos) {
    uint32_t st = 0;
    for (int i = 0; i < 5; i++) {
        const int a = x.p(61 + i), b = x.p(90 + i);
        st |= (uint32_t)((a >> 1) & 1) << (2 * i);
        st |= (uint32_t)(a & 1) << (2 * i + 1);
        st |= (uint32_t)((b >> 1) & 1) << (10 + 2 * i);
        st |= (uint32_t)(b & 1) << (11 + 2 * i);
    }
    if (!golay_20_8(st, x.T)) {
        x.ev(pos, EV_DMR_DATA, 0, -1, -1);
        return false;
    }
    const int cc = (int)(((st & 1) << 3) | (((st >> 1) & 1) << 2) | (((st >> 2) & 1) << 1) | ((st >> 3) & 1));
    const int burst = (int)((((st >> 4) & 1) << 3) | (((st >> 5) & 1) << 2) | (((st >> 6) & 1) << 1) | ((st >> 7) & 1));
    int rej = 0, pend = 0;
    const int c = conf_observe(x, cc, x.f(F_LOCKED) ? 0 : 1); // dmr_confidence_note_data_burst()
    if (c == CONF_REJECT) {
        rej = 1;
    } else if (c != CONF_LOCKED && burst != 9) {
        pend = 1;
    }
    x.f(F_REJECT) = rej;
    x.f(F_PENDING) = pend;
    x.ev(pos, EV_DMR_DATA, 1, cc, burst | (rej << 8) | (pend << 9));
    return true;
}
__device__ __forceinline__ void
data_dispatch(const Ctx& x, int pos) { // dmr_data_dispatch_burst(): the burst handler prints the colour code the gate holds
    if (x.f(F_REJECT) || x.f(F_PENDING)) {
        return;
    }
    x.ev(pos, EV_DMR_CC_PRINT, x.f(F_COLOR), 0, x.f(F_CURSLOT));
}
__device__ inline void
bs_finalize(const Ctx& x, int pos) { // finalize_dmr_bs()
    x.ev(pos, EV_DMR_VOICE_END, 0, x.f(F_TACT_OK), x.f(F_EMB_OK));
    x.f(F_EMBERR0) = x.f(F_EMBERR1) = 0;
    conf_reset(x);
}

// Entered on an accepted BS sync with the 90 cached dibits in pay[0..89].  voice = the BS voice word.  Sets mode / symbols to read
// next; returns false when the handler returns without reading a live dibit.
__device__ inline bool
dmr_begin(const Ctx& x, int pos, bool voice, int& mode, int& next) {
    int slot = 0;
    if (!voice) { // dmr_handle_other_data(): dmr_data_sync() with state->dmr_stereo = 0
        x.f(F_STEREO) = 0;
        if (!tact_decode(x, slot)) {
            x.ev(pos, EV_DMR_DATA, 0, -2, -1);
            mode = M_SKIP66;
            next = 66;
            return true;
        }
        x.f(F_CURSLOT) = slot;
        mode = M_DATA_SUFFIX;
        next = 5;
        return true;
    }
    x.f(F_STEREO) = 1;
    const bool tact_ok = tact_decode(x, slot);
    bool sync_ok = true;
    if (tact_ok) {
        x.f(F_CURSLOT) = slot;
        conf_note_voice_sync(x, slot);
        sync_ok = sync_kind(x) == 1;
    }
    if (!tact_ok || !sync_ok) {
        conf_reset(x);
        x.ev(pos, EV_DMR_VOICE_END, 1, tact_ok ? 1 : 0, sync_ok ? 1 : 0);
        mode = M_IDLE;
        next = 0;
        return false;
    }
    x.f(F_BOOTSLOT) = slot;
    mode = M_BOOT54;
    next = 54;
    return true;
}

// the decisions of process_dmr_bs_iteration() once all 144 dibits of a burst are in pay[]; true = on to the next burst
__device__ inline bool
bs_burst_done(const Ctx& x, int pos) {
    const int slot = x.f(F_ISLOT);
    const int kind = sync_kind(x);
    const bool is_voice = kind == 1, is_data = kind == 2;
    const int vc_read = x.f(F_VC1 + slot); // the VC read_dmr_bs_sync_segment() filed the sync segment under (dmr_bs.c:161-180)
    if (is_voice) { // note_dmr_bs_voice_sync()
        x.f(F_VC1 + slot) = 1;
        x.f(F_EMBERR0 + slot) = 0;
        conf_note_voice_sync(x, slot);
    }
    int action; // 1 SKIP, 2 END
    if (is_data) { // handle_dmr_bs_data_sync(): dmr_data_sync() on the stereo payload
        x.f(F_VC1 + slot) = 7;
        int ts = 0;
        if (tact_decode(x, ts)) {
            x.f(F_CURSLOT) = ts;
            if (data_slot_type(x, pos)) {
                data_dispatch(x, pos);
            }
        }
        x.f(F_SKIPCOUNT)++;
        action = 1;
    } else {
        action = 0;
        if (x.f(F_VC1 + slot) > 6) { // handle_dmr_bs_frame_sync_miss()
            x.f(F_VC1 + slot)++;
            action = (x.f(F_VC1 + slot) > 13) ? 2 : 1;
        }
        if (action == 0) { // process_dmr_bs_voice_burst()
            uint32_t emb = 0; // emb_pdu[i] = syncdata[i], emb_pdu[i + 8] = syncdata[i + 40]: dibits 66..69 and 86..89
            for (int i = 0; i < 4; i++) {
                const int a = x.p(66 + i), b = x.p(86 + i);
                emb |= (uint32_t)((a >> 1) & 1) << (2 * i);
                emb |= (uint32_t)(a & 1) << (2 * i + 1);
                emb |= (uint32_t)((b >> 1) & 1) << (8 + 2 * i);
                emb |= (uint32_t)(b & 1) << (9 + 2 * i);
            }
            int cc = 25;
            int emb_ok = qr_16_7_6(emb, x.T) ? 1 : 0;
            bool ended = false;
            if (emb_ok) {
                x.f(F_EMBERR0 + slot) = 0;
                cc = (int)(((emb & 1) << 3) | (((emb >> 1) & 1) << 2) | (((emb >> 2) & 1) << 1) | ((emb >> 3) & 1));
            } else if (!is_voice) {
                if (x.f(F_EMBERR0 + slot) < 0xFF) {
                    x.f(F_EMBERR0 + slot)++;
                }
                ended = x.f(F_EMBERR0 + slot) >= 2;
            } else {
                x.f(F_EMBERR0 + slot) = 0;
            }
            if (!ended) {
                bool open = x.f(F_VOPEN0 + slot) != 0;
                if (emb_ok) {
                    if (conf_note_voice_burst(x, slot, cc) == CONF_REJECT) {
                        emb_ok = 0;
                        conf_reset_slot(x, slot);
                        ended = true;
                    }
                    open = x.f(F_VOPEN0 + slot) != 0;
                }
                if (!ended && !open) {
                    if (!emb_ok && !is_voice) {
                        conf_reset_slot(x, slot);
                        ended = true;
                    } else {
                        x.f(F_VC1 + slot)++;
                        x.f(F_TACT_OK) = 0;
                        emb_ok = 0;
                        action = (x.f(F_VC1) > 14 || x.f(F_VC2) > 14) ? 2 : 1;
                    }
                } else if (!ended) { // the voice burst proper
                    x.f(F_SKIPCOUNT) = 0;
                    x.ev(pos, EV_DMR_CC_PRINT, x.f(F_COLOR), x.f(F_VC1 + slot), slot);
                    x.f(F_VC1 + slot)++;
                    x.f(F_TACT_OK) = 0;
                    emb_ok = 0;
                    action = (x.f(F_VC1) > 14 || x.f(F_VC2) > 14) ? 2 : 1;
                }
            }
            x.f(F_EMB_OK) = emb_ok;
            if (ended) {
                action = 2;
            }
            x.ev(pos, EV_DMR_VOICE_BURST, slot, cc, (is_voice ? 1 : 0) | (action << 4) | (vc_read << 8));
        }
    }
    if (action == 2) {
        return false;
    }
    if (x.f(F_SKIPCOUNT) > 3) { // run_dmr_bs_post_skip()
        x.f(F_TACT_OK) = 1;
        x.f(F_EMB_OK) = 1;
        return false;
    }
    return true;
}

// A phase has run out (its last dibit is in pay[]).  Sets the next phase and its length; false = the handler has returned.
__device__ inline bool
dmr_phase_end(const Ctx& x, int pos, int& mode, int& next) {
    switch (mode) {
        case M_DATA_SUFFIX:
            if (!data_slot_type(x, pos)) {
                mode = M_SKIP66;
                next = 66;
                return true;
            }
            mode = M_DATA_SECOND;
            next = 49;
            return true;
        case M_DATA_SECOND:
            data_dispatch(x, pos);
            mode = M_SKIP66;
            next = 66;
            return true;
        case M_BOOT54:
            if (x.f(F_VOPEN0 + x.f(F_BOOTSLOT))) { // process_dmr_bs_bootstrap_voice_if_open()
                x.ev(pos, EV_DMR_CC_PRINT, x.f(F_COLOR), 1, x.f(F_BOOTSLOT));
            }
            // dmrBS(): init_dmr_bs_ctx()
            x.f(F_VC1) = x.f(F_VC2) = 7;
            if (x.f(F_CURSLOT) == 0) {
                x.f(F_VC1) = 2;
            } else if (x.f(F_CURSLOT) == 1) {
                x.f(F_VC2) = 2;
            }
            x.f(F_SKIPCOUNT) = 0;
            x.f(F_TACT_OK) = 0;
            x.f(F_EMB_OK) = 0;
            x.f(F_ISLOT) = 0;
            x.f(F_REDB) = 0;
            x.f(F_EMBERR0) = x.f(F_EMBERR1) = 0;
            mode = M_BURST_CACH;
            next = 12;
            return true;
        case M_BURST_CACH: { // collect_dmr_bs_cach_and_tact()
            int slot = 0;
            const bool ok = tact_decode(x, slot);
            x.f(F_TACT_OK) = ok ? 1 : 0;
            if (!ok) {
                bs_finalize(x, pos);
                mode = M_IDLE;
                return false;
            }
            x.f(F_ISLOT) = slot;
            x.f(F_CURSLOT) = slot;
            mode = M_BURST_RED;
            next = 36;
            return true;
        }
        case M_BURST_RED: { // is_dmr_bs_redundant_carrier(): eight dibits of the first voice frame against the previous burst's
            uint32_t cur = 0;
            const int idx[8] = {16, 27, 1, 32, 3, 33, 13, 7};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                cur |= (uint32_t)x.p(12 + idx[k]) << (2 * k);
            }
            if (cur == (uint32_t)x.f(F_REDB)) {
                bs_finalize(x, pos);
                mode = M_IDLE;
                return false;
            }
            x.f(F_REDB) = (int)cur;
            mode = M_BURST_REST;
            next = 96;
            return true;
        }
        case M_BURST_REST:
            if (!bs_burst_done(x, pos)) {
                bs_finalize(x, pos);
                mode = M_IDLE;
                return false;
            }
            mode = M_BURST_CACH;
            next = 12;
            return true;
        default: // M_SKIP66, M_FIXED
            mode = M_IDLE;
            return false;
    }
}

// NXDN LICH after its 8th dibit: lich8 = the eight high bits after de-scrambling (first dibit in bit 7).  Returns accepted.
__device__ inline bool
nxdn_lich_ok(int lich8, int& lich7, int& parity_ok) {
    const int full = lich8, rx_par = full & 1;
    int par = ((full >> 7) + (full >> 6) + (full >> 5) + (full >> 4)) & 1;
    const int lich = full >> 1;
    if (lich == 0x08 || lich == 0x4A || lich == 0x48 || lich == 0x46) {
        par = ((full >> 7) + (full >> 6) + (full >> 5) + (full >> 4) + (full >> 3) + (full >> 2) + (full >> 1)) & 1;
    }
    lich7 = lich;
    parity_ok = rx_par == par;
    if (!parity_ok) {
        return false;
    }
    // k_nxdn_lich_profiles (nxdn_frame.c:117-161) as a 128-bit set
    // members: 01 05 08 20 21 28 29 2E 2F 30 31 32 33 34 35 36 37 38 39 40 41 46 48 49 4A 4E 4F 50 51 52 53 54 55 56 57
    //          60 61 62 63 68 69 6E 6F 70 71 72 73 75 76 77
    const uint32_t m0 = 0x00000122u, m1 = 0x03FFC303u, m2 = 0x00FFC743u, m3 = 0x00EFC30Fu;
    const uint32_t m = lich < 32 ? m0 : (lich < 64 ? m1 : (lich < 96 ? m2 : m3));
    return (m >> (lich & 31)) & 1u;
}

} // namespace ddn_fsk4h
#endif
