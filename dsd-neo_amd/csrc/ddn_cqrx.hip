// ddn_cqrx.hip - the symbol-rate receive loop behind the CQPSK demodulator (ddn_cqpsk_run hands over one float per symbol, levels
// +-1 / +-3): P25 Phase 1 (CQPSK / LSM) with the reference's per-DUID handlers in the loop, P25 Phase 2 with its 700-dibit lock.
//
// reference:
//   getSymbol() symbol-rate fast path        src/dsp/dsd_symbol.c:1581-1624, thresholds reset :744-765
//   getFrameSync() on a QPSK profile         src/dsp/dsd_frame_sync.c:3098-3148; 4-level slice :2061-2075; level window + extrema average
//                                            :2319-2336; exact compare :698-716 / :801-816; rotated retries :668-696; raw fit :417-548
//   in-frame symbol                          src/core/frames/dsd_dibit.c:243-275 (use_symbol), :330-350,951-1000 (CQPSK slice + map),
//                                            :376-430,548-556,609-721 (soft metrics)
//   handlers                                 src/engine/dispatch/dispatch_p25p1.c:86-143,206-225,391-403; p25p1_tsbk.c:117-161;
//                                            p25p1_mdpu.c:177-198,270-307; Phase 2 p25p2_frame.c:352-370
//
// Shape.  The loop is one recurrence per channel at 4800 / 6000 symbols per second - three orders of magnitude less work than the
// sample-rate C4FM loop - so it is written for clarity of the recurrence, not for the last cycle: ONE WAVEFRONT PER CHANNEL.  The
// channel's words are wave-uniform registers, its windows live in LDS (the 1024-deep extrema average, the 128-symbol slicer window,
// the 24-deep level ring and symbol history), and the wavefront is used where the reference loops: the two-smallest / two-largest
// scan of the slicer window (a butterfly of sorted pairs), the sort of the level ring (rank by counting), the NID / trellis
// decoders of the handlers (ddn_nid_dev.h, ddn_p25h_dev.h - the same device decoders the C4FM loop's handler wave runs).  Symbols
// come in and records go out 64 at a time, one per lane.  17 KB of LDS per workgroup: nine channels per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_p25h_dev.h"

namespace {
using ddn_p25h::Scratch;

constexpr int MSZ = 1024, SSZ = 128;
enum { PH_IDLE = 0, PH_NID = 1, PH_BODY = 2, PH_TSBK = 3, PH_MPDU = 4 };

struct Lds {
    // (round 6) the 1024-deep extrema rings stay where the carried state keeps them, in HBM (one slot read - a slot ahead, so off the
    // recurrence - and one written per in-frame symbol): 8 KB less LDS per channel, sixteen channels per CU instead of nine
    float sbuf[SSZ];
    float lbuf[24], shist[24];
    Scratch sc;
};

__device__ __forceinline__ void
wave_sync() { // LDS written by one lane, read by the others of the same (only) wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ float
bcast(float v, int src) {
    return __shfl(v, src);
}

__device__ __forceinline__ int
cq_slice(float s) {
    return s >= 2.0f ? 1 : (s >= 0.0f ? 0 : (s >= -2.0f ? 2 : 3));
}

__device__ __forceinline__ int
map_dibit(int map_idx, int raw) { // include/dsd-neo/core/p25_cqpsk_dibit.h: identity, reverse, X2400, N1200, P1200 (2 bits per entry)
    // the five maps 0xE4, 0x4E, 0x1B, 0x8D, 0x72 in one constant.  (As an indexed array the table went to constant memory: a global
    // load and a wait for EVERY outstanding load - the extrema ring's prefetch among them - in every in-frame symbol.)
    const uint64_t maps = 0x728D1B4EE4ull;
    return (int)((maps >> (8 * map_idx + 2 * raw)) & 3u);
}

__device__ __forceinline__ int
unmap_dibit(int map_idx, int corrected) {
    int raw = corrected;
    for (int q = 3; q >= 0; q--) {
        raw = (map_dibit(map_idx, q) == corrected) ? q : raw;
    }
    return raw;
}

__device__ __forceinline__ int
clamp255(int v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__device__ __forceinline__ int
bit_magnitude(float sym, const float ideal[4], int bit_index) { // soft_metric_for_bit()
    float best0 = 3.4028234663852886e38f, best1 = 3.4028234663852886e38f, spacing = 3.4028234663852886e38f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float d = (sym - ideal[i]) * (sym - ideal[i]);
        if ((i >> (1 - bit_index)) & 1) {
            best1 = d < best1 ? d : best1;
        } else {
            best0 = d < best0 ? d : best0;
        }
#pragma unroll
        for (int j = i + 1; j < 4; j++) {
            const float sp = fabsf(ideal[i] - ideal[j]);
            if (sp > 1e-6f && sp < spacing) {
                spacing = sp;
            }
        }
    }
    if (spacing == 3.4028234663852886e38f) {
        spacing = 2.0f;
    }
    const float scale = 255.0f / (spacing * spacing);
    return clamp255(__float2int_rn(fabsf(best0 - best1) * scale));
}

// the nominal level of each corrected dibit under the lock's dibit map and polarity: what the soft metrics measure against.  It only
// changes with a sync, so it is kept beside the lock's words instead of being unmapped again in every in-frame symbol.
__device__ __forceinline__ void
cq_levels(int map_idx, int negative, float level[4]) {
    const float base_ideal[4] = {1.0f, 3.0f, -1.0f, -3.0f};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int corrected = negative ? (d ^ 2) : d;
        const int raw = unmap_dibit(map_idx, corrected);
        level[d] = raw == 0 ? base_ideal[0] : (raw == 1 ? base_ideal[1] : (raw == 2 ? base_ideal[2] : base_ideal[3]));
    }
}

// digitize() + compute_dibit_soft_metric() on the CQPSK slice
__device__ __forceinline__ void
cq_digitize(float sym, float center, int map_idx, int negative, const float level[4], int snr_scale, int& dibit, int& rel8, int& l0,
            int& l1) {
    dibit = map_dibit(map_idx, cq_slice(sym - center));
    if (negative) {
        dibit ^= 2;
    }
    float ideal[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        ideal[d] = center + level[d];
    }
    int mag0 = bit_magnitude(sym, ideal, 0), mag1 = bit_magnitude(sym, ideal, 1);
    // dmr_compute_reliability(), rf_mod 1
    const float sc = sym - center;
    const float id = sc >= 2.0f ? 3.0f : (sc >= 0.0f ? 1.0f : (sc >= -2.0f ? -1.0f : -3.0f));
    float err = fabsf(sc - id);
    err = err > 1.0f ? 1.0f : err;
    int rel = clamp255((int)((1.0f - err) * 255.0f + 0.5f));
    if (snr_scale >= 0) {
        rel = clamp255((rel * snr_scale) >> 8);
    }
    const int mn = mag0 < mag1 ? mag0 : mag1;
    if (mn > 0 && rel < mn) {
        mag0 = (mag0 * rel) / mn;
        mag1 = (mag1 * rel) / mn;
    }
    l0 = ((dibit >> 1) & 1) ? clamp255(mag0) : -clamp255(mag0);
    l1 = (dibit & 1) ? clamp255(mag1) : -clamp255(mag1);
    const int a0 = l0 < 0 ? -l0 : l0, a1 = l1 < 0 ? -l1 : l1;
    rel8 = clamp255(a1 < a0 ? a1 : a0);
}

// (round 6) three wavefronts per SIMD: left to itself the kernel compiled to 256 + registers (the handlers' decoders, inlined) - ONE
// wavefront per SIMD, four channels per CU, 4096 channels in four rounds (29 ms per 4096 x 4800 symbols).  The per-symbol path needs a
// fraction of that.  Capped at 168 registers twelve channels are resident per CU (the chain objects at 4096 channels: P25 CQPSK 72.6
// -> 55.8 ms, Phase 2 - with the 17-tap band-edge instance - 138 -> 57 ms); at 128 (sixteen per CU) the spills reach the per-symbol
// path and it is slower again (84.8 / 94.3 ms); two per SIMD measures the same as three.
#ifndef DDN_CQ_WAVES
#define DDN_CQ_WAVES 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DDN_CQ_WAVES, DDN_CQ_WAVES))) void
k_cq_rx(const float* __restrict__ symbols, const int32_t* __restrict__ counts_in, size_t sym_stride, int n_fixed, int n_channels,
        DdnCqConfig cfg, DdnCqState* __restrict__ states, uint8_t* __restrict__ rec, uint8_t* __restrict__ flags,
        int32_t* __restrict__ counts_out, size_t max_sym, int32_t* __restrict__ events, int32_t* __restrict__ n_events,
        int32_t* __restrict__ event_data) {
    __shared__ Lds L;
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    DdnCqState* gs = states + ch;
    // ---- carried state in -----------------------------------------------------------------------------------------------------
    float* const gmin = gs->minbuf;
    float* const gmax = gs->maxbuf;
    for (int i = lane; i < SSZ; i += 64) {
        L.sbuf[i] = gs->sbuf[i];
    }
    if (lane < 24) {
        L.lbuf[lane] = gs->lbuf[lane];
        L.shist[lane] = gs->shist[lane];
    }
    L.sc.nb[lane] = gs->nb[lane];
    L.sc.nr[lane] = gs->nr[lane];
    for (int i = lane; i < 100; i += 64) {
        L.sc.d[i] = gs->d[i];
    }
    ddn_p25h::crc_cols_fill(L.sc.crc_cols, lane);
    if (lane == 0) {
        ddn_nid::gf_fill(L.sc.ex, L.sc.lg);
        ddn_nid::chase_masks_fill(L.sc.masks);
    }
    float s_max = gs->max, s_min = gs->min, lmin = gs->lmin, lmax = gs->lmax;
    double min_sum = gs->min_sum, max_sum = gs->max_sum;
    int sidx = gs->sidx, midx = gs->midx, sums_valid = gs->sums_valid;
    bool win_ok = false;                                          // the slicer window's carried extrema below are current
    float w_lo1 = 0.0f, w_lo2 = 0.0f, w_hi1 = 0.0f, w_hi2 = 0.0f; // its two smallest and two largest values
    int have_sync = gs->have_sync, lock_left = gs->lock_left, lastsync = gs->lastsync, map_idx = gs->map_idx;
    int lidx = gs->lidx, level_count = gs->level_count, hist_count = gs->hist_count, shead = gs->shead, scount = gs->scount;
    int hunt_pos = gs->hunt_pos;
    uint64_t hist = gs->hist;
    int h_phase = gs->h_phase, h_idx = gs->h_idx, h_left = gs->h_left, h_block = gs->h_block, h_end = gs->h_end, h_skip = gs->h_skip,
        h_k = gs->h_k, h_nac = gs->h_nac, h_p2cc = gs->h_p2cc;
    int nev = 0;
    wave_sync();
    // the ring slot the next push replaces, fetched ahead (refreshed by whatever rewrites the ring)
    float pre_lo = gmin[(midx >= 0 && midx < MSZ) ? midx : 0], pre_hi = gmax[(midx >= 0 && midx < MSZ) ? midx : 0];

    const int sync_len = cfg.sync_len, t_max = cfg.t_max;
    const uint64_t hmask = sync_len == 24 ? 0xFFFFFFFFFFFFull : 0xFFFFFFFFFFull;
    int n = counts_in ? counts_in[ch] : n_fixed;
    n = n < 0 ? 0 : n;
    const float* src = symbols + (size_t)ch * sym_stride;
    uint8_t* orec = rec + (size_t)ch * max_sym * 10;
    uint8_t* ofl = flags + (size_t)ch * max_sym;

    auto push_minmax = [&](float lo, float hi) { // dsd_state_push_minmax_window(state, 1024, lo, hi)
        if (!sums_valid) { // (after a raw fit seeded the window: every slot holds the same pair)
            double a = 0.0, b = 0.0;
            for (int i = 0; i < MSZ; i++) {
                a += (double)gmin[i];
                b += (double)gmax[i];
            }
            min_sum = a;
            max_sum = b;
            sums_valid = 1;
            if (midx < 0 || midx >= MSZ) {
                midx = 0;
            }
            pre_lo = gmin[midx];
            pre_hi = gmax[midx];
        }
        min_sum += (double)lo - (double)pre_lo;
        max_sum += (double)hi - (double)pre_hi;
        if (lane == 0) {
            gmin[midx] = lo;
            gmax[midx] = hi;
        }
        midx = (midx + 1 >= MSZ) ? 0 : midx + 1;
        pre_lo = gmin[midx]; // (a slot this wave wrote 1023 pushes ago, or never: no store of this wave is in flight to it)
        pre_hi = gmax[midx];
        s_min = (float)(min_sum / (double)MSZ);
        s_max = (float)(max_sum / (double)MSZ);
    };
    auto hunt_enter = [&]() {
        hunt_pos = 0;
        have_sync = 0;
        lidx = 0;
        level_count = 0;
        hist_count = 0;
        hist = 0;
        lmin = s_min;
        lmax = s_max;
    };
    auto no_carrier = [&]() {
        lastsync = 0;
        s_max = 15000.0f;
        s_min = -15000.0f;
        h_nac = 0; // engine.c:1889; p2_cc stays
    };
    auto push_event = [&](int pos, int kind, int a, int b, int p0, int p1, int p2, int p3) {
        if (lane == 0 && events && nev < cfg.max_events) {
            int32_t* e = events + ((size_t)ch * cfg.max_events + nev) * 4;
            e[0] = pos;
            e[1] = kind;
            e[2] = a;
            e[3] = b;
            if (event_data) {
                *reinterpret_cast<int4*>(event_data + ((size_t)ch * cfg.max_events + nev) * 4) = make_int4(p0, p1, p2, p3);
            }
        }
        nev++;
    };
    // block reader shared by the TSDU and data-unit handlers (processTSBK, p25_mpdu_read_repetition)
    auto block_take = [&](int l0, int l1) {
        if ((h_skip / 36) == 0) {
            if (h_k < 98 && lane == 0) {
                L.sc.d[ddn_p25h::deinterleave98(h_k)] = (int32_t)((uint32_t)(uint16_t)(int16_t)l0 | ((uint32_t)(uint16_t)(int16_t)l1 << 16));
            }
            h_k++;
        } else {
            h_skip = 0;
        }
        h_skip++;
    };
    // tsbk_decode_repetition_bytes(): list-8, the first CRC16-clean candidate, else the best one
    auto half_rate_select = [&](uint32_t by[3], int& crc_ok) -> int {
        wave_sync();
        ddn_p25h::half_rate_best_wave(L.sc, lane, by);
        crc_ok = ddn_p25h::crc16_ok_wave(L.sc, by, lane);
        int sel = 0;
        if (!crc_ok) {
            wave_sync();
            ddn_p25h::half_rate_list_wave(L.sc, lane);
            wave_sync();
            const int nout = L.sc.n_out;
            for (int q = 0; q < nout; q++) {
                const uint32_t cw[3] = {L.sc.outl[q][0], L.sc.outl[q][1], L.sc.outl[q][2]};
                if (ddn_p25h::crc16_ok(cw)) {
                    sel = q;
                    crc_ok = 1;
                    break;
                }
            }
            by[0] = L.sc.outl[sel][0];
            by[1] = L.sc.outl[sel][1];
            by[2] = L.sc.outl[sel][2];
        }
        return sel;
    };
    // one in-frame symbol through the P25 Phase 1 handlers (ddn_p25h_dev.h lists their source lines); false = the handler has returned
    auto handler_symbol = [&](int pos, int d, int l0, int l1) -> bool {
        if (h_phase == PH_NID) {
            const int i = h_idx++;
            if (i != 11 && lane == 0) { // dibit 11 is the status symbol inside the NID
                const int b = (i < 11) ? 2 * i : 2 * (i - 1);
                const int a0 = l0 < 0 ? -l0 : l0, a1 = l1 < 0 ? -l1 : l1;
                L.sc.nb[b] = (uint8_t)((d >> 1) & 1);
                L.sc.nr[b] = (uint8_t)(a0 > 255 ? 255 : a0);
                L.sc.nb[b + 1] = (uint8_t)(d & 1); // the last dibit: index 62 + the parity bit at 63
                L.sc.nr[b + 1] = (uint8_t)(a1 > 255 ? 255 : a1);
            }
            if (i < 32) {
                return true;
            }
            wave_sync();
            const uint64_t w = __ballot(lane < 63 && L.sc.nb[lane] != 0);
            const int par = L.sc.nb[63], prel = L.sc.nr[63];
            const int observed = (h_nac > 0 && h_nac < 0xFFF) ? h_nac : ((h_p2cc > 0 && h_p2cc < 0xFFF) ? h_p2cc : 0);
            const ddn_nid::Gf gf = {L.sc.ex, L.sc.lg};
            const ddn_nid::Work wk = {L.sc.work + lane, L.sc.work + 23 * 64 + lane, L.sc.work + 47 * 64 + lane, L.sc.work + 71 * 64 + lane};
            const ddn_nid::NidRes r = ddn_nid::nid_decode_wave(gf, wk, w, L.sc.nr, par, prel, observed, cfg.nid_threshold, L.sc.masks, lane);
            wave_sync();
            int duid = 0xFF;
            if (r.status > 0) {
                const bool valid = r.nac != 0 && r.nac != 0xFFF;
                if (r.nac != h_nac && valid) {
                    h_nac = r.nac;
                    h_p2cc = r.nac;
                }
                duid = r.duid;
            }
            push_event(pos, ddn_p25h::EV_NID, r.status, (r.nac & 0xFFFF) | (duid << 16), r.status, r.nac, r.duid, r.errs);
            if (duid == 0x7 || duid == 0xC) {
                h_phase = (duid == 0x7) ? PH_TSBK : PH_MPDU;
                h_block = 0;
                h_end = 3;
                h_skip = 36 - 14;
                h_idx = 0;
                h_k = 0;
                return true;
            }
            h_left = duid == 0x0 ? 339 : ((duid == 0x5 || duid == 0xA) ? 807 : (duid == 0x3 ? 15 : (duid == 0xF ? 159 : 0)));
            if (h_left <= 0) {
                h_phase = PH_IDLE;
                return false;
            }
            h_phase = PH_BODY;
            return true;
        }
        if (h_phase == PH_BODY) {
            if (--h_left <= 0) {
                h_phase = PH_IDLE;
                return false;
            }
            return true;
        }
        if (h_phase == PH_TSBK) {
            block_take(l0, l1);
            if (++h_idx < 101) {
                return true;
            }
            uint32_t by[3];
            int crc_ok;
            const int sel = half_rate_select(by, crc_ok);
            const int last = (int)((by[0] >> 7) & 1u);
            push_event(pos, ddn_p25h::EV_TSBK, h_block, crc_ok | (int)(((by[0] >> 8) & 0xFF) << 8) | (((last << 8) | sel) << 16), (int)by[0],
                       (int)by[1], (int)by[2], (crc_ok & 1) | ((sel & 0xFF) << 8) | (h_block << 16));
            h_block++;
            h_idx = 0;
            h_k = 0;
            if (last || h_block >= 3) {
                h_phase = PH_IDLE;
                return false;
            }
            return true;
        }
        if (h_phase == PH_MPDU) {
            block_take(l0, l1);
            h_idx++;
            if (h_k < 98 && h_idx < 101) {
                return true;
            }
            if (h_block == 0) {
                uint32_t by[3];
                int crc_ok;
                const int sel = half_rate_select(by, crc_ok);
                if (crc_ok) {
                    const int sap = (int)((by[0] >> 8) & 0x3F), blks = (int)((by[1] >> 16) & 0x7F);
                    h_end = blks + 1;
                    if ((sap == 61 || sap == 63) && blks > 10) {
                        h_end = 4;
                    }
                }
                push_event(pos, ddn_p25h::EV_MPDU, crc_ok, (h_end & 0xFFFF) | (int)((by[0] & 0xFF) << 16), (int)by[0], (int)by[1], (int)by[2],
                           (crc_ok & 1) | ((sel & 0xFF) << 8));
            }
            h_block++;
            h_idx = 0;
            h_k = 0;
            if (h_block >= h_end) {
                h_phase = PH_IDLE;
                return false;
            }
            return true;
        }
        return false;
    };

    float lv[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // cq_levels() of (lv_map, lv_neg)
    int lv_map = -1, lv_neg = -1;
    int o = 0; // records written by this call
    for (int base = 0; base < n; base += 64) {
        const int cnt = (n - base) < 64 ? (n - base) : 64;
        const float mine = lane < cnt ? src[base + lane] : 0.0f;
        uint32_t r_lo = 0, r_hi = 0; // this lane's record: {dibit, rel, llr0 (i16)} , {llr1 (i16), flags << 16}
        float r_sym = 0.0f;
        for (int j = 0; j < cnt; j++) {
            const float x = bcast(mine, j);
            int dibit = 0, rel8 = 0, l0 = 0, l1 = 0, fl = 0;
            // dsd_symbol_history_push()
            if (lane == 0) {
                L.shist[shead] = x;
            }
            shead = (shead + 1) % 24;
            scount = scount < 24 ? scount + 1 : 24;
            if (have_sync) {
                // ---- in frame: window slot, use_symbol(), digitize() --------------------------------------------------------
                // the window's two smallest and two largest are carried from symbol to symbol: the slot's old value leaves, the symbol
                // enters.  If the value that leaves lies strictly between the second smallest and the second largest the four stay
                // what they were but for the newcomer, which takes its place by plain compares; otherwise (it is, or ties with, one
                // of the four: a few per cent of the symbols; a zero or a NaN coming in; the first symbol after anything else wrote
                // the window) the whole window is scanned as before
                const float old = L.sbuf[sidx];
                if (lane == 0) {
                    L.sbuf[sidx] = x;
                }
                float lo1 = w_lo1, lo2 = w_lo2, hi1 = w_hi1, hi2 = w_hi2;
                if (!win_ok || !(old > w_lo2 && old < w_hi2) || !(x != 0.0f)) {
                    wave_sync();
                    const float a = L.sbuf[lane], b = L.sbuf[lane + 64];
                    lo1 = a < b ? a : b, lo2 = a < b ? b : a, hi1 = lo2, hi2 = lo1;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const float o1 = __shfl_xor(lo1, off), o2 = __shfl_xor(lo2, off);
                        const float n1 = lo1 < o1 ? lo1 : o1, mx = lo1 < o1 ? o1 : lo1, m2 = lo2 < o2 ? lo2 : o2;
                        lo2 = mx < m2 ? mx : m2;
                        lo1 = n1;
                        const float p1 = __shfl_xor(hi1, off), p2 = __shfl_xor(hi2, off);
                        const float g1 = hi1 > p1 ? hi1 : p1, mn = hi1 > p1 ? p1 : hi1, g2 = hi2 > p2 ? hi2 : p2;
                        hi2 = mn > g2 ? mn : g2;
                        hi1 = g1;
                    }
                    // (a scan that met a NaN is not carried: its compares have no order to update from)
                    win_ok = (lo1 == lo1) && (lo2 == lo2) && (hi1 == hi1) && (hi2 == hi2);
                } else {
                    if (x < lo1) {
                        lo2 = lo1;
                        lo1 = x;
                    } else if (x < lo2) {
                        lo2 = x;
                    }
                    if (x > hi1) {
                        hi2 = hi1;
                        hi1 = x;
                    } else if (x > hi2) {
                        hi2 = x;
                    }
                }
                w_lo1 = lo1, w_lo2 = lo2, w_hi1 = hi1, w_hi2 = hi2;
                const float wlo = (lo1 + lo2) * 0.5f, whi = (hi1 + hi2) * 0.5f;
                push_minmax(wlo, whi);
                const float center = (s_max + s_min) / 2.0f;
                sidx = (sidx >= SSZ - 1) ? 0 : sidx + 1;
                const int neg = lastsync == 2;
                if (lv_map != map_idx || lv_neg != neg) {
                    cq_levels(map_idx, neg, lv);
                    lv_map = map_idx;
                    lv_neg = neg;
                }
                cq_digitize(x, center, map_idx, neg, lv, cfg.snr_scale, dibit, rel8, l0, l1);
                fl = 1 | (neg ? 4 : 0);
                if (cfg.lock_symbols < 0) {
                    if (!handler_symbol(o, dibit, l0, l1)) {
                        hunt_enter();
                    }
                } else if (--lock_left <= 0) {
                    hunt_enter();
                }
            } else {
                // ---- hunting ------------------------------------------------------------------------------------------------
                if (lane == 0) {
                    L.lbuf[lidx] = x;
                    L.sbuf[sidx] = x;
                }
                win_ok = false;
                level_count = level_count < t_max ? level_count + 1 : level_count;
                lidx = (lidx == t_max - 1) ? 0 : lidx + 1;
                sidx = (sidx == SSZ - 1) ? 0 : sidx + 1;
                const int raw = cq_slice(x); // the fast path has just put the centre at 0
                hist = ((hist << 2) | (uint64_t)raw) & hmask;
                hist_count = hist_count < 24 ? hist_count + 1 : 24;
                dibit = raw;
                bool accepted = false;
                if (hist_count >= 8) {
                    wave_sync();
                    // frame_sync_window_levels(): sort the ring (rank by counting), mean of three from either end
                    const bool act = lane < level_count;
                    const float v = act ? L.lbuf[lane] : 0.0f;
                    int rank = 0;
                    for (int q = 0; q < level_count; q++) {
                        const float vq = L.lbuf[q];
                        rank += (vq < v || (vq == v && q < lane)) ? 1 : 0;
                    }
                    auto sorted_at = [&](int k) {
                        const unsigned long long m = __ballot(act && rank == k);
                        return bcast(v, __ffsll((long long)m) - 1);
                    };
                    if (level_count < 3) {
                        float sum = 0.0f;
                        for (int q = 0; q < level_count; q++) {
                            sum += sorted_at(q);
                        }
                        lmin = lmax = sum / (float)level_count;
                    } else {
                        const int a0 = level_count >= 13 ? 2 : 0, b0 = level_count >= 13 ? level_count - 5 : level_count - 3;
                        lmin = (sorted_at(a0) + sorted_at(a0 + 1) + sorted_at(a0 + 2)) / 3.0f;
                        lmax = (sorted_at(b0) + sorted_at(b0 + 1) + sorted_at(b0 + 2)) / 3.0f;
                    }
                    push_minmax(lmin, lmax); // QPSK profile: the hunting levels feed the extrema average too
                    if (hist_count >= sync_len) {
                        int pol = 0, m = -1;
                        if (hist == cfg.target[0][0]) {
                            pol = 1, m = 0;
                        } else if (hist == cfg.target[1][0]) {
                            pol = 2, m = 0;
                        } else {
                            for (int p = 0; p < 2 && m < 0; p++) {
                                for (int k = 1; k < 4; k++) {
                                    if (hist == cfg.target[p][k]) {
                                        pol = p + 1;
                                        m = k;
                                        break;
                                    }
                                }
                            }
                        }
                        float fc = 0.0f, fg = 0.0f;
                        bool fit = false;
                        if (m > 0 && scount >= sync_len) { // frame_sync_fit_p25_cqpsk_raw_sync()
                            wave_sync();
                            float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                            int cn[4] = {0, 0, 0, 0};
                            for (int i = 0; i < sync_len; i++) {
                                const int rw = (int)((hist >> (2 * (sync_len - 1 - i))) & 3u);
                                const int back = sync_len - 1 - i;
                                const float sv = L.shist[(shead - 1 - back + 48) % 24];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    if (rw == q) {
                                        sum[q] += sv;
                                        cn[q]++;
                                    }
                                }
                            }
                            const float unit[4] = {1.0f, 3.0f, -1.0f, -3.0f};
                            float sx = 0.0f, sy = 0.0f, sxx = 0.0f, sxy = 0.0f;
                            int nn = 0;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (cn[q]) {
                                    const float xx = unit[q], yy = sum[q] / (float)cn[q];
                                    sx += xx;
                                    sy += yy;
                                    sxx += xx * xx;
                                    sxy += xx * yy;
                                    nn++;
                                }
                            }
                            if (nn >= 2) {
                                const float den = ((float)nn * sxx) - (sx * sx);
                                if (!(fabsf(den) < 1.0e-6f)) {
                                    const float g = (((float)nn * sxy) - (sx * sy)) / den;
                                    if (!(fabsf(g) * 2.0f < 1.0f)) {
                                        fc = (sy - (g * sx)) / (float)nn;
                                        fg = g;
                                        fit = true;
                                    }
                                }
                            }
                        }
                        if ((m == 2 || m == 3) && !fit) { // N1200 / P1200 need the centre fit
                            pol = 0;
                        }
                        if (pol) {
                            map_idx = m == 0 ? 0 : m + 1; // identity | X2400 = 2, N1200 = 3, P1200 = 4
                            s_max = (s_max + lmax) / 2;
                            s_min = (s_min + lmin) / 2;
                            lastsync = pol;
                            if (m > 0 && fit) { // frame_sync_apply_p25_cqpsk_raw_fit(): seeds the extrema average and the window
                                const float half = fabsf(fg) * 3.0f;
                                s_min = fc - half;
                                s_max = fc + half;
                                wave_sync();
                                for (int i = lane; i < MSZ; i += 64) {
                                    gmin[i] = s_min;
                                    gmax[i] = s_max;
                                }
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // (read back by every lane at the next push)
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                pre_lo = s_min;
                                pre_hi = s_max;
                                for (int i = lane; i < SSZ; i += 64) {
                                    L.sbuf[i] = (i & 1) ? s_max : s_min;
                                }
                                win_ok = false;
                                sums_valid = 0;
                                wave_sync();
                            }
                            have_sync = 1;
                            lock_left = cfg.lock_symbols;
                            fl = 2 | (pol == 2 ? 4 : 0) | (map_idx << 4);
                            accepted = true;
                            if (cfg.lock_symbols < 0) {
                                h_phase = PH_NID; // orc_p25h_begin()
                                h_idx = 0;
                            } else if (lock_left <= 0) {
                                hunt_enter();
                            }
                        }
                    }
                }
                if (!accepted) {
                    if (hunt_pos < 10200) {
                        hunt_pos++;
                    } else {
                        hunt_pos = 0;
                        no_carrier();
                    }
                    if (!(cfg.protocol == 0 && lastsync == 2) && hunt_pos >= 1800) {
                        no_carrier();
                        hunt_enter();
                    }
                }
            }
            if (lane == j) {
                r_lo = (uint32_t)(dibit & 0xFF) | ((uint32_t)(rel8 & 0xFF) << 8) | ((uint32_t)(uint16_t)(int16_t)l0 << 16);
                r_hi = (uint32_t)(uint16_t)(int16_t)l1 | ((uint32_t)fl << 16);
                r_sym = x;
            }
            o++;
        }
        // 64 records out: {dibit, reliability, llr0 i16, llr1 i16, symbol f32}
        if (lane < cnt && (size_t)(base + lane) < max_sym) {
            uint8_t* q = orec + (size_t)(base + lane) * 10;
            const uint32_t sb = __float_as_uint(r_sym);
            *reinterpret_cast<uint16_t*>(q) = (uint16_t)(r_lo & 0xFFFF);
            *reinterpret_cast<uint16_t*>(q + 2) = (uint16_t)(r_lo >> 16);
            *reinterpret_cast<uint16_t*>(q + 4) = (uint16_t)(r_hi & 0xFFFF);
            *reinterpret_cast<uint16_t*>(q + 6) = (uint16_t)(sb & 0xFFFF);
            *reinterpret_cast<uint16_t*>(q + 8) = (uint16_t)(sb >> 16);
            ofl[base + lane] = (uint8_t)(r_hi >> 16);
        }
    }
    // ---- carried state out ------------------------------------------------------------------------------------------------------
    wave_sync();
    for (int i = lane; i < SSZ; i += 64) {
        gs->sbuf[i] = L.sbuf[i];
    }
    if (lane < 24) {
        gs->lbuf[lane] = L.lbuf[lane];
        gs->shist[lane] = L.shist[lane];
    }
    gs->nb[lane] = L.sc.nb[lane];
    gs->nr[lane] = L.sc.nr[lane];
    for (int i = lane; i < 100; i += 64) {
        gs->d[i] = L.sc.d[i];
    }
    if (lane == 0) {
        gs->max = s_max, gs->min = s_min, gs->lmin = lmin, gs->lmax = lmax;
        gs->min_sum = min_sum, gs->max_sum = max_sum;
        gs->sidx = sidx, gs->midx = midx, gs->sums_valid = sums_valid;
        gs->have_sync = have_sync, gs->lock_left = lock_left, gs->lastsync = lastsync, gs->map_idx = map_idx;
        gs->lidx = lidx, gs->level_count = level_count, gs->hist_count = hist_count, gs->shead = shead, gs->scount = scount;
        gs->hunt_pos = hunt_pos;
        gs->hist = hist;
        gs->h_phase = h_phase, gs->h_idx = h_idx, gs->h_left = h_left, gs->h_block = h_block, gs->h_end = h_end, gs->h_skip = h_skip;
        gs->h_k = h_k, gs->h_nac = h_nac, gs->h_p2cc = h_p2cc;
        counts_out[ch] = n;
        if (n_events) {
            n_events[ch] = nev;
        }
    }
}

__global__ void
k_cq_state_init(DdnCqState* st, int n) {
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= n) {
        return;
    }
    DdnCqState* s = st + ch;
    // initState(): src/core/util/dsd_init.c:519-539
    for (int i = lane; i < MSZ; i += 64) {
        s->minbuf[i] = -15000.0f;
        s->maxbuf[i] = 15000.0f;
    }
    for (int i = lane; i < SSZ; i += 64) {
        s->sbuf[i] = 0.0f;
    }
    if (lane < 24) {
        s->lbuf[lane] = 0.0f;
        s->shist[lane] = 0.0f;
    }
    s->nb[lane] = 0;
    s->nr[lane] = 0;
    for (int i = lane; i < 100; i += 64) {
        s->d[i] = 0;
    }
    if (lane == 0) {
        s->max = 15000.0f, s->min = -15000.0f, s->lmin = -15000.0f, s->lmax = 15000.0f;
        s->min_sum = 0.0, s->max_sum = 0.0;
        s->sidx = 0, s->midx = 0, s->sums_valid = 0;
        s->have_sync = 0, s->lock_left = 0, s->lastsync = 0, s->map_idx = 0;
        s->lidx = 0, s->level_count = 0, s->hist_count = 0, s->shead = 0, s->scount = 0, s->hunt_pos = 0;
        s->hist = 0;
        s->h_phase = 0, s->h_idx = 0, s->h_left = 0, s->h_block = 0, s->h_end = 0, s->h_skip = 0, s->h_k = 0, s->h_nac = 0, s->h_p2cc = 0;
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_cq_rx_init(DdnCqState* states, int n_channels, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_cq_state_init, dim3((unsigned)n_channels), dim3(64), 0, st, states, n_channels);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_cq_rx(const float* symbols, const int32_t* counts_in, size_t sym_stride, int n_fixed, int n_channels, const DdnCqConfig* cfg,
              DdnCqState* states, uint8_t* rec, uint8_t* flags, int32_t* counts_out, size_t max_sym, int32_t* events, int32_t* n_events,
              int32_t* event_data, hipStream_t st) {
    if (n_channels <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_cq_rx, dim3((unsigned)n_channels), dim3(64), 0, st, symbols, counts_in, sym_stride, n_fixed, n_channels, *cfg, states, rec,
                       flags, counts_out, max_sym, events, n_events, event_data);
    return hipGetLastError();
}
