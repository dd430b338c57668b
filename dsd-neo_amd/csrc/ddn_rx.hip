// ddn_rx.hip — batched fixed-protocol P25 Phase 1 C4FM receive loop: discriminator samples -> capture records.
//
// One lane = one channel running what the reference's decoder thread runs per stream:
//   getSymbol() (RTL-FSK discriminator path)   src/dsp/dsd_symbol.c:1343-1387,197-211,347-358,360-397,436-460,489-517,
//                                              1769-1805,1839-1851; matched-filter gating :301-338
//   getFrameSync() hunting loop                src/dsp/dsd_frame_sync.c:3098-3148 (ring :1747-1764, sign dibit
//                                              :2110-2127, level window :2316-2336 + src/dsp/frame_sync_level.c:10-44,
//                                              P25p1 pattern :698-716, accept :385-392,603-625)
//   threshold warm start                       src/dsp/sync_calibration.c:156-233
//   in-frame symbols                           get_dibit_and_analog_signal, src/core/frames/dsd_dibit.c (ddn_slicer_dev.h)
// Scope and deviations: P25p1 only, modulation locked to C4FM.  The in-frame symbol count comes from the reference's
// per-DUID handlers running inside the loop (HANDLERS = true, ddn_p25h_dev.h / ddn_nid_dev.h; ddn_p25_rx_set_handlers) or, for the
// stage tests, from a caller-given lock length; 1800 sync-less symbols reset the loop as the reference's noCarrier() does.
//
// GPU shape.  Everything here is a per-sample / per-symbol recurrence, so the only parallelism is across channels,
// and one wavefront issues about one VALU instruction every ~5 cycles however many of its lanes are active.  At the
// batch sizes this runs (thousands of channels) spreading channels over MORE wavefronts with FEWER active lanes each is
// what shortens the critical path, so the kernel is templated on channels-per-wavefront (CPW = 16/32/64) and the
// launcher picks the smallest CPW that still fits the grid.  Per workgroup: wave 0 runs the recurrence, wave 1 streams
// the next 64-sample tile of every channel's row into LDS with coalesced 256-B row loads (raw discriminator samples and
// the always-on matched-filter output F computed beforehand by k_p25_matched_filter).  The matched filter is a pure FIR,
// so once it has been on for 90 samples its output IS F; only the 90 samples after its first enable are computed
// inline (history = zeros before the enable sample, as the reference's static filter memory).
// The two 1024-deep extrema rings stay in HBM; a ring refill (reset, warm start) is recorded as {fill value,
// pushes since fill} instead of 2048 stores, and the refilled ring's binary64 sum is 1024*v exactly (k*v is exact in
// binary64 for k <= 1024 and a binary32 v, so the reference's sequential rebuild gives the same bits).

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_p25h_dev.h"
#include "ddn_slicer_dev.h"
#include "ddn_tables_p25.h"

using ddn_sl::two_max_insert;
using ddn_sl::two_min_insert;

#define DDN_RXW_RAW (1 << 30)
namespace {
constexpr int TS = 64;       // samples per staged tile
constexpr int SS = 128;      // opts->ssize
constexpr int MS = 1024;     // opts->msize
constexpr int NT = DDN_P25_FILTER_TAPS;
constexpr int QCAP = 10;     // symbols one lane can finish inside one 64-sample tile when sps >= 8 (64 / 7 + 1)
constexpr uint32_t kSyncBits = 0xFB30A0u; // P25P1_SYNC as sign bits, oldest symbol in bit 23

__constant__ uint32_t c_taps[NT];

template <int CPW>
struct Lds {
    float sb[SS][CPW];
    float gs[8][4][CPW];
    float lb[24][CPW];
    float sh[24][CPW];
    float raw[2][CPW][TS + 1];
    float flt[2][CPW][TS + 1];
    // symbol queue wave 0 -> wave 1 (soft decisions + record stores happen off the recurrence's critical path):
    // per tile and lane up to QCAP symbols of {symbol, center, umid, lmid, max, min, flags}; qn = count, qo = first index
    float q[2][QCAP][7][CPW];
    int qn[2][CPW];
    int qo[2][CPW];
};
} // namespace

// Carrier loss (src/dsp/dsd_frame_sync.c:2753-2760,3037-3053): a hunt that saw 1800 symbols without a sync - 10200 while
// the last sync was the inverted pattern - runs noCarrier() (src/engine/engine.c:1838-1847).  For this loop that is: the
// crossing latch cleared, lastsynctype NONE (so the matched filter is gated off again; its memory stays as it is), max /
// min / centre parked, and - because the timing ratio is zeroed - the next getSymbol() re-initialises the samples-per-
// symbol accumulator and the whole slicer (src/dsp/dsd_symbol.c:1306-1341).
__device__ __forceinline__ void
rx_no_carrier(DdnRxState& s) {
    s.jitter = -1;
    s.lastsync = 0;
    s.filter_on = 0;
    s.max = 15000.0f;
    s.min = -15000.0f;
    s.center = 0.0f;
    s.need_reset = 1;
}
__device__ __forceinline__ void
rx_timing_reset(DdnRxState& s) {
    s.need_reset = 0;
    s.sps_accum = 0;
    s.jitter = -1;
    s.center = 0.0f;
    s.min = -30000.0f;
    s.max = 30000.0f;
    s.lmid = -20000.0f;
    s.umid = 20000.0f;
    s.minref = -24000.0f;
    s.maxref = 24000.0f;
    s.fill_min = s.min; // the extrema rings are refilled: {value, pushes since} instead of 2048 stores
    s.fill_max = s.max;
    s.since_fill = 0;
    s.midx = 0;
    s.min_sum = (double)s.min * 1024.0;
    s.max_sum = (double)s.max * 1024.0;
}
__device__ __forceinline__ void
rx_hunt_restart(DdnRxState& s) { // frame_sync_runtime_init() of the next getFrameSync() call
    s.hunt_pos = 0;
    s.lidx = 0;
    s.level_count = 0;
    s.hist_count = 0;
    s.hist_bits = 0;
    s.lmin = s.min;
    s.lmax = s.max;
}

template <int CPW>
__global__ __launch_bounds__(128) void
k_p25_rx(const float* __restrict__ raw, const float* __restrict__ filt, const float* __restrict__ prev_tail,
          float* __restrict__ fstale, long n,
         size_t stride, int n_channels, DdnRxConfig cfg, DdnRxState* __restrict__ state, float* __restrict__ sbuf_store,
         float* __restrict__ lbuf_store, float* __restrict__ shist_store, float* __restrict__ minring,
         float* __restrict__ maxring, uint8_t* __restrict__ rec, uint8_t* __restrict__ flags, int32_t* __restrict__ counts,
         size_t max_sym, const int32_t* __restrict__ lock_cfg) {
    extern __shared__ unsigned char smem_raw[];
    Lds<CPW>& L = *reinterpret_cast<Lds<CPW>*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;
    const int ch0 = blockIdx.x * CPW;
    const int ch = ch0 + lane;
    const bool live = !loader && lane < CPW && ch < n_channels;
    const int ln = lane < CPW ? lane : 0; // LDS column (idle lanes alias column 0 but never write)
    const bool use_flt = cfg.use_filter != 0;

    DdnRxState s;
    if (live) {
        s = state[ch];
        for (int k = 0; k < SS; k++) {
            L.sb[k][ln] = sbuf_store[(size_t)k * n_channels + ch];
        }
        for (int k = 0; k < 24; k++) {
            L.lb[k][ln] = lbuf_store[(size_t)k * n_channels + ch];
            L.sh[k][ln] = shist_store[(size_t)k * n_channels + ch];
        }
    } else {
        s = DdnRxState{};
    }
    auto refresh_group = [&](int g) {
        float a1 = L.sb[g * 16][ln], a2 = L.sb[g * 16 + 1][ln];
        float b1 = a1, b2 = a2;
        if (a2 < a1) {
            const float t = a1;
            a1 = a2;
            a2 = t;
        }
        if (b2 > b1) {
            const float t = b1;
            b1 = b2;
            b2 = t;
        }
#pragma unroll
        for (int k = 2; k < 16; k++) {
            const float v = L.sb[g * 16 + k][ln];
            two_min_insert(v, a1, a2);
            two_max_insert(v, b1, b2);
        }
        L.gs[g][0][ln] = a1;
        L.gs[g][1][ln] = a2;
        L.gs[g][2][ln] = b1;
        L.gs[g][3][ln] = b2;
    };
    if (live) {
        for (int g = 0; g < 8; g++) {
            refresh_group(g);
        }
    }

    // ---- coalesced staging by the loader wave -----------------------------------------------------------------
    auto stage = [&](long t0, int buf) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
#pragma unroll
        for (int h = 0; h < CPW / 16; h++) {
            float r[16], f[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int cc = 16 * h + c;
                const bool ok = (ch0 + cc < n_channels) && lane < tn;
                const size_t off = (size_t)(ch0 + cc) * stride + (size_t)t0 + lane;
                r[c] = ok ? raw[off] : 0.0f;
                f[c] = (ok && use_flt) ? filt[off] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < 16; c++) {
                L.raw[buf][16 * h + c][lane] = r[c];
                L.flt[buf][16 * h + c][lane] = f[c];
            }
        }
    };
    if (loader && n > 0) {
        stage(0, 0);
    }
    __syncthreads();

    const int whole0 = cfg.out_rate / cfg.sym_rate, rem0 = cfg.out_rate % cfg.sym_rate;
    const int whole = whole0 < 2 ? 2 : (whole0 > 64 ? 64 : whole0);
    const int rem = (whole0 < 2 || whole0 > 64) ? 0 : rem0;
    int o = 0;
    uint8_t* rp = rec + (size_t)(live ? ch : 0) * max_sym * 10;
    uint8_t* fp = flags + (size_t)(live ? ch : 0) * max_sym;
    const long long abs0 = s.n_abs;

    // With >= 8 samples per symbol a tile yields at most QCAP symbols per lane, and the slice / soft-decision /
    // record-store work (a third of the per-symbol instructions, none of it feeding back) moves to wave 1.
    const bool offload = whole >= 8 && !(cfg.dbg & 16);
    const bool hlive = loader && lane < CPW && ch < n_channels;
    uint8_t* hrp = rec + (size_t)(hlive ? ch : 0) * max_sym * 10;
    uint8_t* hfp = flags + (size_t)(hlive ? ch : 0) * max_sym;
    auto drain = [&](int qb) {
        if (!hlive) {
            return;
        }
        const int cnt = L.qn[qb][ln], o0 = L.qo[qb][ln];
        for (int k = 0; k < QCAP; k++) {
            if (k >= cnt) {
                break;
            }
            const float sym = L.q[qb][k][0][ln];
            const int fl = __float_as_int(L.q[qb][k][6][ln]);
            int dibit, relb = 0, l0 = 0, l1 = 0;
            if (fl & 1) {
                const ddn_sl::Thr th = {L.q[qb][k][1][ln], L.q[qb][k][2][ln], L.q[qb][k][3][ln], L.q[qb][k][4][ln],
                                        L.q[qb][k][5][ln]};
                ddn_sl::slice_soft(sym, th, (fl >> 2) & 1, dibit, relb, l0, l1);
            } else {
                dibit = sym > 0.0f ? 1 : 3;
            }
            const size_t oo = (size_t)(o0 + k);
            if (oo < max_sym) {
                uint8_t* r = hrp + oo * 10;
                const uint32_t xb = __float_as_uint(sym);
                ((uint16_t*)r)[0] = (uint16_t)((dibit & 3) | (relb << 8));
                ((uint16_t*)r)[1] = (uint16_t)(int16_t)l0;
                ((uint16_t*)r)[2] = (uint16_t)(int16_t)l1;
                ((uint16_t*)r)[3] = (uint16_t)(xb & 0xFFFFu);
                ((uint16_t*)r)[4] = (uint16_t)(xb >> 16);
                hfp[oo] = (uint8_t)fl;
            }
        }
    };
    int buf = 0;
    int it = 0;
    for (long t0 = 0; t0 < n; t0 += TS, buf ^= 1, it++) {
        const int tn = (int)((n - t0) < TS ? (n - t0) : TS);
        if (loader) {
            if (t0 + TS < n) {
                stage(t0 + TS, buf ^ 1);
            }
            if (offload && it > 0) {
                drain((it - 1) & 1);
            }
        } else {
            int qk = 0;
            if (live && offload) {
                L.qo[it & 1][ln] = o;
            }
            int sp = 0; // this lane's cursor in the tile
            int guard = 0;
            // the matched filter's memory at the moment it is gated off (see rx_no_carrier): the last 90 samples it was fed
            auto snapshot_filter = [&]() {
                if (!s.filter_on) {
                    return;
                }
                const long long tnext = abs0 + t0 + sp;
                float* fs = fstale + (size_t)ch * (NT - 1);
                for (int k = 0; k < NT - 1; k++) {
                    const long long ja = tnext - (NT - 1) + k;
                    float v;
                    if (ja >= s.filt_start) {
                        const long jc = (long)(ja - abs0);
                        v = (jc >= 0) ? raw[(size_t)ch * stride + jc] : prev_tail[(size_t)ch * (NT - 1) + (NT - 1) + jc];
                    } else {
                        v = fs[(int)(ja - s.filt_start) + (NT - 1)];
                    }
                    fs[k] = v;
                }
            };
            // frame_sync_advance_sync_window() + frame_sync_handle_no_sync_timeout() after a hunting symbol without a sync
            auto hunt_advance = [&]() {
                if (s.hunt_pos < 10200) {
                    s.hunt_pos++;
                } else {
                    s.hunt_pos = 0;
                    snapshot_filter();
                    rx_no_carrier(s);
                }
                if (s.lastsync != 2 && s.hunt_pos >= 1800) {
                    snapshot_filter();
                    rx_no_carrier(s);
                    rx_hunt_restart(s);
                }
            };
            while (true) {
                // ---- sample loop: run every lane to the end of its current symbol (or of the tile) ---------------
                if (live && sp < tn && !s.in_symbol) {
                    if (s.need_reset) {
                        rx_timing_reset(s);
                    }
                    int sps = whole;
                    if (rem > 0) {
                        int acc = s.sps_accum + rem;
                        if (acc >= cfg.sym_rate) {
                            sps++;
                            acc -= cfg.sym_rate;
                        }
                        s.sps_accum = acc;
                        sps = sps > 64 ? 64 : sps;
                    }
                    s.span = sps;
                    s.centre = (sps - 1) / 2;
                    s.i = 0;
                    s.sum = 0.0f;
                    s.count = 0;
                    s.in_symbol = 1;
                    if (sps > 1 && s.have_sync == 0 && s.jitter >= 0) {
                        if (s.jitter > 0 && s.jitter <= s.centre) {
                            s.i--;
                        } else if (s.jitter > s.centre && s.jitter < sps) {
                            s.i++;
                        }
                        s.jitter = -1;
                    }
                }
                // In-frame fast path.  With have_sync = 1 there is no slip, and once jitter is latched (>= 0) the
                // per-sample crossing test cannot change it, so the symbol is just the mean of the five clipped
                // window samples (added in sample order) and lastsample the clipped last sample.  Taken when the whole
                // symbol lies inside the staged tile and the filter (if on) is past its 90-sample cold start.
                if (live && s.in_symbol && s.i == 0 && s.have_sync && s.jitter >= 0 && s.span >= 6 && s.span != 20
                    && sp + s.span <= tn && !(cfg.dbg & 8)
                    && (!s.filter_on || (abs0 + t0 + sp - s.filt_start) >= (long long)(NT - 1))) {
                    const bool fo = s.filter_on != 0;
                    const int c = s.centre;
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int j = sp + c - 2 + k;
                        float x = fo ? L.flt[buf][ln][j] : L.raw[buf][ln][j];
                        x = x > s.max ? s.max : (x < s.min ? s.min : x);
                        acc += x;
                    }
                    const int jl = sp + s.span - 1;
                    float xl = fo ? L.flt[buf][ln][jl] : L.raw[buf][ln][jl];
                    xl = xl > s.max ? s.max : (xl < s.min ? s.min : xl);
                    s.sum = acc;
                    s.count = 5;
                    s.lastsample = xl;
                    sp += s.span;
                    s.i = s.span;
                }
                bool act = live && sp < tn && s.i < s.span;
                while (__any(act)) {
                    if (act && (cfg.dbg & 4)) {
                        s.i++;
                        sp++;
                    } else if (act) {
                        float x = L.raw[buf][ln][sp];
                        if (s.filter_on) {
                            const long long a = abs0 + t0 + sp;
                            if (a - s.filt_start >= (long long)(NT - 1)) {
                                x = L.flt[buf][ln][sp];
                            } else {
                                // first 90 samples after the enable: FIR over a zero-extended history
                                const long k = t0 + sp;
                                float acc = 0.0f;
                                for (int i = 0; i < NT; i++) {
                                    const long j = k - (NT - 1) + i;
                                    float v; // before the enable sample: the filter's memory as the last hunt left it
                                    if (abs0 + j >= s.filt_start) {
                                        v = (j >= 0) ? raw[(size_t)ch * stride + j]
                                                     : prev_tail[(size_t)ch * (NT - 1) + (NT - 1) + j];
                                    } else {
                                        v = fstale[(size_t)ch * (NT - 1) + (size_t)((abs0 + j - s.filt_start) + (NT - 1))];
                                    }
                                    acc += __uint_as_float(c_taps[i]) * v;
                                }
                                x = acc;
                            }
                        }
                        if (s.have_sync) {
                            x = x > s.max ? s.max : (x < s.min ? s.min : x);
                        }
                        const int i = s.i;
                        if (s.jitter < 0) {
                            if (x > s.center) {
                                if (!(x > s.maxref * 1.25f) && s.lastsample < s.center) {
                                    s.jitter = i;
                                }
                            } else if (!(x < s.minref * 1.25f) && s.lastsample > s.center) {
                                s.jitter = i;
                            }
                        }
                        if (s.span == 20 && i >= 7 && i <= 13) {
                            s.sum += x;
                            s.count++;
                        }
                        if ((s.span == 5 && i == 2) || (i >= s.centre - 2 && i <= s.centre + 2)) {
                            s.sum += x;
                            s.count++;
                        }
                        s.lastsample = x;
                        s.i++;
                        sp++;
                    }
                    act = live && sp < tn && s.i < s.span;
                }
                // ---- symbol commit ----------------------------------------------------------------------------
                const bool done = live && s.in_symbol && s.i >= s.span;
                if (done && (cfg.dbg & 1)) {
                    s.in_symbol = 0;
                    o++;
                } else if (done) {
                    const float sym = (s.count > 0) ? (s.sum / (float)s.count) : 0.0f;
                    s.in_symbol = 0;
                    L.sh[s.shead][ln] = sym;
                    s.shead = (s.shead + 1 >= 24) ? 0 : s.shead + 1;
                    s.scount = s.scount < 24 ? s.scount + 1 : 24;
                    int dibit, relb = 0, l0 = 0, l1 = 0, fl = 0;
                    if (s.have_sync) {
                        // get_dibit_and_analog_signal(): window, extrema rings, thresholds, slice + soft decision
                        const int neg = (s.lastsync == 2);
                        L.sb[s.sidx][ln] = sym;
                        refresh_group(s.sidx >> 4);
                        float m1 = L.gs[0][0][ln], m2 = L.gs[0][1][ln], x1 = L.gs[0][2][ln], x2 = L.gs[0][3][ln];
#pragma unroll
                        for (int g = 1; g < 8; g++) {
                            if (cfg.dbg & 64) {
                                break;
                            }
                            two_min_insert(L.gs[g][0][ln], m1, m2);
                            two_min_insert(L.gs[g][1][ln], m1, m2);
                            two_max_insert(L.gs[g][2][ln], x1, x2);
                            two_max_insert(L.gs[g][3][ln], x1, x2);
                        }
                        const float lo = (m1 + m2) * 0.5f, hi = (x1 + x2) * 0.5f;
                        const size_t ro = (size_t)ch * MS + s.midx; // [channel][slot], see k_p25_rxw
                        float old_lo = s.fill_min, old_hi = s.fill_max;
                        if (s.since_fill >= MS && !(cfg.dbg & 32)) {
                            old_lo = minring[ro];
                            old_hi = maxring[ro];
                        } else {
                            s.since_fill++;
                        }
                        s.min_sum += (double)lo - (double)old_lo;
                        s.max_sum += (double)hi - (double)old_hi;
                        minring[ro] = lo;
                        maxring[ro] = hi;
                        s.midx = (s.midx + 1 >= MS) ? 0 : s.midx + 1;
                        s.min = (float)(s.min_sum / (double)MS);
                        s.max = (float)(s.max_sum / (double)MS);
                        s.center = (s.max + s.min) / 2.0f;
                        s.umid = ((s.max - s.center) * 5.0f / 8.0f) + s.center;
                        s.lmid = ((s.min - s.center) * 5.0f / 8.0f) + s.center;
                        s.maxref = s.max * 0.80f;
                        s.minref = s.min * 0.80f;
                        s.sidx = (s.sidx >= SS - 1) ? 0 : s.sidx + 1;
                        fl = 1 | (neg ? 4 : 0);
                        if (offload) {
                            L.q[it & 1][qk][1][ln] = s.center;
                            L.q[it & 1][qk][2][ln] = s.umid;
                            L.q[it & 1][qk][3][ln] = s.lmid;
                            L.q[it & 1][qk][4][ln] = s.max;
                            L.q[it & 1][qk][5][ln] = s.min;
                            dibit = 0;
                        } else {
                            const ddn_sl::Thr th = {s.center, s.umid, s.lmid, s.max, s.min};
                            ddn_sl::slice_soft(sym, th, neg, dibit, relb, l0, l1);
                        }
                        if (--s.lock_left <= 0) {
                            s.have_sync = 0;
                            s.hunt_pos = 0;
                            s.lidx = 0;
                            s.level_count = 0;
                            s.hist_count = 0;
                            s.hist_bits = 0;
                            s.lmin = s.min;
                            s.lmax = s.max;
                        }
                    } else {
                        // getFrameSync(): one hunting iteration
                        L.lb[s.lidx][ln] = sym;
                        s.level_count = s.level_count < 24 ? s.level_count + 1 : 24;
                        L.sb[s.sidx][ln] = sym;
                        refresh_group(s.sidx >> 4);
                        s.lidx = (s.lidx == 23) ? 0 : s.lidx + 1;
                        s.sidx = (s.sidx >= SS - 1) ? 0 : s.sidx + 1;
                        const uint32_t bit = sym > 0.0f ? 1u : 0u;
                        s.hist_bits = ((s.hist_bits << 1) | bit) & 0xFFFFFFu;
                        s.hist_count = s.hist_count < 24 ? s.hist_count + 1 : 24;
                        dibit = bit ? 1 : 3;
                        if (s.hist_count >= 8) {
                            s.maxref = s.max;
                            s.minref = s.min;
                            int pol = 0;
                            if (s.hist_count >= 24) {
                                pol = (s.hist_bits == kSyncBits) ? 1 : ((s.hist_bits == (~kSyncBits & 0xFFFFFFu)) ? 2 : 0);
                            }
                            // The level window (lmin / lmax of the last <= 24 hunting symbols) is recomputed from
                            // scratch by the reference on every hunting symbol but only consumed when a sync is
                            // accepted, so it is evaluated here only then: same values at the only point of use.
                            if (pol) {
                                // five smallest (ascending) / five largest (descending) of the level window
                                const float big = 3.4028234663852886e38f;
                                float a0 = big, a1 = big, a2 = big, a3 = big, a4 = big;
                                float b0 = -big, b1 = -big, b2 = -big, b3 = -big, b4 = -big;
                                const int cnt = s.level_count;
                                for (int k = 0; k < 24; k++) {
                                    if (k < cnt) {
                                        float v = L.lb[k][ln], t;
                                        float w = v;
                                        t = fminf(a0, v); v = fmaxf(a0, v); a0 = t;
                                        t = fminf(a1, v); v = fmaxf(a1, v); a1 = t;
                                        t = fminf(a2, v); v = fmaxf(a2, v); a2 = t;
                                        t = fminf(a3, v); v = fmaxf(a3, v); a3 = t;
                                        a4 = fminf(a4, v);
                                        t = fmaxf(b0, w); w = fminf(b0, w); b0 = t;
                                        t = fmaxf(b1, w); w = fminf(b1, w); b1 = t;
                                        t = fmaxf(b2, w); w = fminf(b2, w); b2 = t;
                                        t = fmaxf(b3, w); w = fminf(b3, w); b3 = t;
                                        b4 = fmaxf(b4, w);
                                    }
                                }
                                if (cnt >= 13) {
                                    s.lmin = (a2 + a3 + a4) / 3.0f;
                                    s.lmax = (b4 + b3 + b2) / 3.0f;
                                } else {
                                    s.lmin = (a0 + a1 + a2) / 3.0f;
                                    s.lmax = (b2 + b1 + b0) / 3.0f;
                                }
                            }
                            if (pol) {
                                s.max = (s.max + s.lmax) / 2;
                                s.min = (s.min + s.lmin) / 2;
                                s.lastsync = pol;
                                if (use_flt && !s.filter_on) {
                                    s.filter_on = 1;
                                    s.filt_start = abs0 + t0 + sp; // first sample the filter sees
                                }
                                if (s.scount >= 24) {
                                    float sp_ = 0.0f, sn_ = 0.0f;
                                    int np = 0, nn = 0;
                                    int idx = s.shead;
                                    for (int k = 0; k < 24; k++) {
                                        idx = idx == 0 ? 23 : idx - 1;
                                        const float v = L.sh[idx][ln];
                                        if (v > 0.0f) {
                                            sp_ += v;
                                            np++;
                                        } else {
                                            sn_ += v;
                                            nn++;
                                        }
                                    }
                                    if (np != 0 && nn != 0) {
                                        const float mp = sp_ / (float)np, mn = sn_ / (float)nn;
                                        if (!(fabsf(mp - mn) < 1.0f)) {
                                            s.max = mp;
                                            s.min = mn;
                                            s.center = (s.max + s.min) / 2.0f;
                                            s.umid = s.center + (s.max - s.center) * 0.625f;
                                            s.lmid = s.center + (s.min - s.center) * 0.625f;
                                            s.maxref = s.max * 0.80f;
                                            s.minref = s.min * 0.80f;
                                            s.fill_max = s.max;
                                            s.fill_min = s.min;
                                            s.since_fill = 0;
                                            s.max_sum = (double)s.max * (double)MS;
                                            s.min_sum = (double)s.min * (double)MS;
                                        }
                                    }
                                }
                                s.have_sync = 1;
                                s.lock_left = lock_cfg[ch]; // in-frame symbols after a sync, per channel
                                fl = 2 | (pol == 2 ? 4 : 0);
                                if (s.lock_left <= 0) {
                                    s.have_sync = 0;
                                    s.lidx = 0;
                                    s.level_count = 0;
                                    s.hist_count = 0;
                                    s.hist_bits = 0;
                                    s.lmin = s.min;
                                    s.lmax = s.max;
                                    s.hunt_pos = 0;
                                }
                            }
                        }
                        if (!(fl & 2)) {
                            hunt_advance();
                        }
                    }
                    if (offload) {
                        L.q[it & 1][qk][0][ln] = sym;
                        L.q[it & 1][qk][6][ln] = __int_as_float(fl);
                        qk++;
                    } else if ((size_t)o < max_sym && !(cfg.dbg & 2)) {
                        uint8_t* r = rp + (size_t)o * 10;
                        const uint32_t xb = __float_as_uint(sym);
                        ((uint16_t*)r)[0] = (uint16_t)((dibit & 3) | (relb << 8));
                        ((uint16_t*)r)[1] = (uint16_t)(int16_t)l0;
                        ((uint16_t*)r)[2] = (uint16_t)(int16_t)l1;
                        ((uint16_t*)r)[3] = (uint16_t)(xb & 0xFFFFu);
                        ((uint16_t*)r)[4] = (uint16_t)(xb >> 16);
                        fp[o] = (uint8_t)fl;
                    }
                    o++;
                }
                const bool busy = live && sp < tn;
                if (!__any(busy) || ++guard > 2 * TS) {
                    break;
                }
            }
            if (live && offload) {
                L.qn[it & 1][ln] = qk;
            }
        }
        __syncthreads();
    }
    if (loader && offload && it > 0) {
        drain((it - 1) & 1);
    }
    if (live) {
        s.n_abs = abs0 + n;
        state[ch] = s;
        counts[ch] = o;
        for (int k = 0; k < SS; k++) {
            sbuf_store[(size_t)k * n_channels + ch] = L.sb[k][ln];
        }
        for (int k = 0; k < 24; k++) {
            lbuf_store[(size_t)k * n_channels + ch] = L.lb[k][ln];
            shist_store[(size_t)k * n_channels + ch] = L.sh[k][ln];
        }
    }
}


// =====================================================================================================================
// k_p25_rxw — the same loop for CPW <= 32 with these changes that cut the instructions on the per-channel chain:
//  * the staged tiles form a ring of three (previous, current, next being loaded), so a symbol that would straddle a
//    tile edge is simply deferred to the next tile and every symbol is evaluated whole: the five-sample latched path in
//    frame, a straight per-sample pass (crossing search included) while hunting.  The sample-at-a-time loop is left for
//    the matched filter's 90-sample cold start, spans > 24 and the last tile of a call (partial symbols are carried);
//  * the 128-symbol window statistics never leave the tile rhythm.  The window after a push = (ring as it stood at a
//    checkpoint, minus the m oldest entries) + (the m symbols pushed since).  The first part does not depend on the
//    symbols being produced, so a third wavefront (wave 2) prepares, while tile t is processed, the suffix summaries
//    S_m = {two smallest, two largest} of ring entries m+1..128 as of the START of tile t, for every m the next tile can
//    ask for; tile t+1 uses them with checkpoint = start of tile t.  On the recurrence wave a push is then: insert the
//    symbol into this tile's running summary (pc), and - in frame - merge {S_m, summary of the previous tile's pushes
//    (pp), pc}: 4 LDS reads and ~30 ALU operations, no rescans, no per-group events, no handshake besides the tile
//    barrier.  Same multiset statistics as the reference's rescan of the 128 values (src/core/frames/dsd_dibit.c:194-241);
//  * the queue to wave 1 carries {symbol, max, min, flags}; centre / mid thresholds are recomputed there with the
//    reference's expressions;
//  * (round 2) three kinds of trip, the cheapest one every live lane qualifies for: the LEAN trip (every lane in frame with
//    its crossing latched: clipped mean from operands fetched one trip ahead, window push, rings, thresholds, queue entry),
//    the STANDARD trip (in-frame and hunting lanes with ordinary whole symbols, straight-line, the crossing search shared out
//    over the wave's 64 / CPW lanes per column) and the GENERAL trip (everything else, and the trip that ends a tile);
//  * (round 2) 128-sample tiles at 8 lanes per wave (LdsW::TW): half the tile prologues and closing trips.
constexpr int WMAX = 24;

// three-operand min / max / median as single VALU instructions.  fminf / fmaxf go through the compiler's IEEE canonicalisation
// (an extra v_max_f32 x, x, x per operand that comes from memory); on the recurrence wave every instruction is ~5 cycles of issue and
// a dependent one ~9 (tools/ubench/chain_latency.hip), so the lean run spells its extrema bookkeeping out.  Operands are never NaN on
// the paths that use these (a NaN channel's statistics are unspecified, tests/test_nonfinite_gpu.py).
__device__ __forceinline__ float
v_min2(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float
v_max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float
v_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float
v_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float
v_med3(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// What a helper wave does while it has nothing to do.  A dependent chain runs up to three times slower on a SIMD whose vector unit
// falls idle between its instructions than beside a wave that keeps it issuing (tools/ubench/lonely_wave.hip: 150 ns per step alone,
// 56 beside a sleeping wave, 47 beside a wave that spins on VALU instructions), so the staging and handler waves - one of them shares
// every SIMD with a recurrence wave - idle on a short burst of register-only VALU instructions at the lowest priority instead of
// s_sleep.  cfg.dbg bit 8388608 restores the sleep for A/B timing.
__device__ __forceinline__ void
helper_idle(float& spin, bool sleep_instead, int sleep_arg) {
    if (sleep_instead) {
        if (sleep_arg >= 4) {
            __builtin_amdgcn_s_sleep(4);
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
        asm volatile("v_max_f32 %0, %0, %0" : "+v"(spin));
    }
}
// x / 5.0f, correctly rounded, in three instructions: q = RN(x * RN(1/5)), r = x - 5 q (exact in the fma), q + r * RN(1/5).  Equal to
// the IEEE quotient for every finite binary32 x except -0 (checked exhaustively on the host, tests/test_oracle_rx_kat.py holds the
// sampled form); the callers' sums start from +0 and can never be -0.
__device__ __forceinline__ float
div5_exact(float x) {
    const float y = 0.2f;
    const float q = x * y;
    const float r = __builtin_fmaf(-5.0f, q, x);
    return __builtin_fmaf(r, y, q);
}


// 1: per-wave cycle counters (tile body / barrier wait / trips by kind) written over the tail of each workgroup's first
// record area when cfg.dbg bit 8192 is set - timing experiments only (tools/bench_rx_handlers.py with DDN_RX_DBG=8192 and a library built with EXTRA=-DDDN_RX_CYCLES=1)
#ifndef DDN_RX_CYCLES
#define DDN_RX_CYCLES 0
#endif

template <int CPW, bool SMALL = false>
struct LdsW {
    // tile shape: 128-sample tiles where the rows fit (half the tile prologues / closing trips), 64 at 32 lanes per wave
    // (8 lanes per wave: 66 KB, two workgroups per CU still fit; at 16 lanes the 128-sample shape would take 132 KB and halve
    // the resident workgroups of an 8192-channel batch)
    static constexpr int TW = CPW <= 8 ? 128 : 64, RTW = 3 * TW;
    // suffix summaries per checkpoint: pushes per two tiles, 2 * (ceil((TW + sps) / (sps - 1)) + 1).  SMALL (handler mode, which
    // needs the LDS for its history ring): sized for at least 9 samples per symbol
    static constexpr int WMW = TW == 128 ? (SMALL ? 40 : 64) : (SMALL ? 24 : 32);
    static constexpr int QTW = TW == 128 ? 40 : 20; // queue slots = trips of a tile that can hand a symbol to wave 1
    float sb[SS][CPW];
    float lb[24][CPW];
    float sh[24][CPW];
    // a row = [mirror of slot 2 | slot 0 | slot 1 | slot 2] + pad: sample j (-TW <= j < TW + 12) of the tile in slot b
    // is row[TW + b * TW + j] with no wrap test (slot 2 precedes slot 0 in the ring)
    float raw[CPW][RTW + TW + 13];
    float flt[CPW][RTW + TW + 13];
    alignas(16) float q[2][QTW][CPW][4]; // [tile parity][trip][lane] = {symbol, max, min, flags | output index << 8}
    int qn[2][4];                       // trips of that tile, per recurrence wave
    int qo[2][CPW];                     // output index of the lane's first symbol of that tile
    alignas(16) float sfx[2][WMW][CPW][4]; // [checkpoint parity][m - 1][lane] = {min1, min2, max1, max2} of ring entries m+1..128
    int sidx0[2][CPW];        // [tile parity] ring slot of the oldest entry at the start of that tile
};

// Handler mode (HM): a fourth wave answers, one decision at a time, how long the frame a lane is in goes on (ddn_p25h_dev.h).
// The recurrence wave keeps every in-frame symbol of a phase that ends in a decision - {symbol, max, min, flags}, what wave 1
// slices from - in a per-channel history ring, posts a request when the phase's last symbol is in, and that lane sits out until
// the answer is there (the other lanes go on; the tile does not end while a lane waits).
template <int CPW>
struct LdsH {
    float hh[ddn_p25h::HN][CPW][3]; // {symbol, max, min} of the phase's in-frame symbols, slot = count mod HN
    int req_seq[CPW], req_kind[CPW], req_hw[CPW], req_n[CPW], req_o[CPW], req_neg[CPW], req_nc[CPW];
    int rsp_seq[CPW], rsp_ext[CPW], rsp_more[CPW];
#if DDN_RX_CYCLES
    int req_t[CPW], rsp_t[CPW], rsp_pick[CPW]; // timing experiments: when the request was posted / answered, how long it lay unserved
#endif
    int tile_done[4]; // per recurrence wave: tiles it has finished; bit 30 (DDN_RXW_RAW): the tile its staging picks up next may be read
                      // unfiltered by one of its channels (see stage_half_load)
    int hwid[8];      // HW_ID of the workgroup's waves (role placement)
    int ready[4];     // per recurrence wave: tiles the staging wave has made enterable for its channels (staged + window
                      // summaries of the checkpoint before)
    int fready[4];    // per recurrence wave: tiles whose matched-filter row the handler wave has computed (filter in the loop)
    int staged[4];    // per recurrence wave: tiles whose raw samples are in the ring (what the filter pass waits for; `ready` follows
                      // once the window summaries are written too)
    float hh_dummy[CPW][4]; // where the lean run's history store goes for a lane whose phase does not end in a decision
    ddn_p25h::Scratch sc;
    // (at the end: the members above keep the LDS addresses they had - the handler wave's scratch sits right below the 64 KB that an
    // LDS instruction's immediate offset reaches, and the loop's time moved by 2.5 % with these 128 bytes in front of it)
};

// NRW_T = recurrence waves per workgroup.  Handler mode runs two (four lanes each at eight channels per workgroup: four waves, two
// workgroups per CU, one recurrence and one light wave per SIMD) or four (two lanes each: six waves per workgroup, three waves per
// SIMD = two recurrences and one light wave, 168 registers per wave) - a recurrence is a latency chain that leaves its SIMD idle
// most cycles, two of them interleave almost for free (tools/ubench/simd_share.hip), and every special trip (a sync's warm start, a
// bulk hunting pass, a handler's answer) then holds up one other channel instead of three.
template <int CPW, bool HM, int NRW_T = (HM ? 2 : 1)>
__global__ __launch_bounds__(HM ? 64 * (NRW_T + 2) * (NRW_T == 4 ? 2 : 1) : 192) __attribute__((amdgpu_waves_per_eu(NRW_T == 4 ? 3 : 2))) void
k_p25_rxw(const float* __restrict__ raw, const float* __restrict__ filt, const float* __restrict__ prev_tail,
          float* __restrict__ fstale, long n,
          size_t stride, int n_channels, DdnRxConfig cfg, DdnRxState* __restrict__ state, float* __restrict__ sbuf_store,
          float* __restrict__ lbuf_store, float* __restrict__ shist_store, float* __restrict__ minring,
          float* __restrict__ maxring, uint8_t* __restrict__ rec, uint8_t* __restrict__ flags,
          int32_t* __restrict__ counts, size_t max_sym, const int32_t* __restrict__ lock_cfg,
          DdnP25HState* __restrict__ hstate, float* __restrict__ hh_store, int32_t* __restrict__ events,
          int32_t* __restrict__ n_events) {
    using LW = LdsW<CPW, HM>;
    constexpr int TW = LW::TW, RTW = LW::RTW, WMW = LW::WMW, QTW = LW::QTW;
    (void)RTW;
    extern __shared__ unsigned char smem_all[];
    // NRW_T = 4: a launch workgroup is TWO logical ones of six waves, each with its own half of the LDS and its own channels.  The
    // hardware deals a workgroup's waves round the four SIMDs in order, so twelve waves land three per SIMD (two six-wave workgroups
    // of their own would both start at SIMD 0 and ask two SIMDs for four waves' registers: only one would be resident), and waves
    // 0-5 / 6-11 sit on SIMDs (0 1 2 3 0 1) / (2 3 0 1 2 3): every SIMD gets two recurrences and one helper.  The halves share
    // nothing but the three workgroup barriers of the prologue.
    constexpr int LTHREADS = HM ? 64 * (NRW_T + 2) : 192;   // threads of a logical workgroup
    constexpr size_t LBYTES = HM ? ((((sizeof(LW) + 15) & ~(size_t)15) + sizeof(LdsH<CPW>) + 15) & ~(size_t)15) : sizeof(LW);
    const int lhalf = (NRW_T == 4) ? (int)(threadIdx.x / LTHREADS) : 0;
    const int ltid = (int)threadIdx.x - lhalf * LTHREADS;
    const int lblock = (NRW_T == 4) ? (int)blockIdx.x * 2 + lhalf : (int)blockIdx.x;
    unsigned char* smem_raw = smem_all + (size_t)lhalf * LBYTES;
    LW& L = *reinterpret_cast<LW*>(smem_raw);
    LdsH<CPW>& H = *reinterpret_cast<LdsH<CPW>*>(smem_raw + ((sizeof(LW) + 15) & ~(size_t)15));
    const int lane = ltid & 63;
    // (round 6) The matched filter inside the loop: with `filt` == NULL the always-on filter output of a tile is computed from the raw
    // tile where it already sits in LDS - by the handler wave, between decisions - instead of by a kernel of its own that writes
    // a second f32 row to HBM for this one to read back (src/dsp/dsd_filters.c:173-200,299-324: same taps, same order of sums).
    // (128-sample tiles only: the 90 samples ahead of a tile are then all in the ring's previous slot)
    const bool fuse_mf = HM && LdsW<CPW, HM>::TW == 128 && cfg.use_filter != 0 && filt == nullptr;
    // Which wave takes which role.  The dispatcher puts the four waves of a workgroup on the four SIMDs of its CU in an order that
    // changes from workgroup to workgroup, and two workgroups share a CU: with the roles tied to the wave index a quarter of the
    // SIMDs ended up with two recurrence waves (two latency chains taking turns) and a quarter with none.  So the roles go by SIMD:
    // the workgroup in the SIMDs' wave slot 0 runs its recurrences on SIMDs 0 / 1, the staging wave on 2, the handlers on 3, the
    // workgroup in slot 1 the other way round - every SIMD hosts one recurrence and one light wave.
    int wave_role = ltid >> 6;
    if (HM && !(cfg.dbg & 524288)) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (lane == 0) {
            H.hwid[ltid >> 6] = (int)hwid;
        }
        __syncthreads();
        int seen = 0, slot_of_simd0 = 0;
        for (int w = 0; w < 4; w++) {
            const int v = H.hwid[w];
            seen |= 1 << ((v >> 4) & 3);
            if (((v >> 4) & 3) == 0) {
                slot_of_simd0 = v & 15;
            }
        }
        if (NRW_T == 2 && seen == 15) { // one wave per SIMD (else: roles by wave index, as before)
            wave_role = ((int)((hwid >> 4) & 3) - 2 * (slot_of_simd0 & 1)) & 3;
        }
        if (NRW_T == 4) {
            // six waves over four SIMDs: two SIMDs get two of them.  The first wave of the workgroup on a SIMD runs a recurrence, a
            // second one is a helper (staging, then handlers) - with two workgroups per CU placed (2, 2, 1, 1) and (1, 1, 2, 2)
            // every SIMD hosts two recurrences and one helper.  Any other placement: roles by wave index.
            constexpr int NW = NRW_T + 2;
            const int me = ltid >> 6, my_simd = (int)((hwid >> 4) & 3);
            int cnt[4] = {0, 0, 0, 0}, my_rank = 0;
            for (int w = 0; w < NW; w++) {
                const int sd = (H.hwid[w] >> 4) & 3;
                cnt[sd]++;
                my_rank += (w < me && sd == my_simd) ? 1 : 0;
            }
            const bool good = cnt[0] >= 1 && cnt[1] >= 1 && cnt[2] >= 1 && cnt[3] >= 1 && cnt[0] <= 2 && cnt[1] <= 2 && cnt[2] <= 2 && cnt[3] <= 2;
            if (good) {
                int before = 0; // waves of the same kind (first / second on their SIMD) on lower SIMDs
                for (int sd = 0; sd < my_simd; sd++) {
                    before += my_rank == 0 ? 1 : (cnt[sd] == 2 ? 1 : 0);
                }
                wave_role = my_rank == 0 ? before : NRW_T + before;
            }
        }
    }
    const bool hwave = HM && wave_role == NRW_T + 1; // the handlers' decisions
    if (HM && hwave) {
        // The handler wave runs its own path from here on (its decoders would otherwise be allocated on top of the
        // recurrence's live registers); it meets the other waves at the same workgroup barriers: two before the tile loop,
        // one per tile.
        const int ch0 = lblock * CPW;
        const int ch = ch0 + lane;
        auto wg_barrier = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        __builtin_amdgcn_s_setprio(3); // a lane of the recurrence wave (and with it the wave) waits on every decision
    // handler wave: its per-channel words (lane = channel), the history ring, the mailboxes, the decoder tables
    DdnP25HState hs = DdnP25HState{};
    int h_served = 0, h_nev = 0;
    const bool hlive = hwave && lane < CPW && ch < n_channels;
    {
        if (hlive) {
            hs = hstate[ch];
        }
        for (int k = lane; k < ddn_p25h::HN * CPW * 3; k += 64) {
            const int w = k % 3, c = (k / 3) % CPW, slot = k / (3 * CPW);
            H.hh[slot][c][w] = (ch0 + c < n_channels) ? hh_store[((size_t)(ch0 + c) * ddn_p25h::HN + slot) * 3 + w] : 0.0f;
        }
        if (lane < CPW) {
            H.req_seq[lane] = 0;
            H.rsp_seq[lane] = 0;
        }
        ddn_p25h::crc_cols_fill(H.sc.crc_cols, lane);
        if (lane == 0) {
            for (int w = 0; w < 4; w++) {
                H.tile_done[w] = DDN_RXW_RAW; // (tiles 0 and 1 of a call are staged whole)
                H.ready[w] = 1; // tile 0 is staged and summarised by the prologue
                H.fready[w] = 0;
                H.staged[w] = 1;
            }
            ddn_nid::gf_fill(H.sc.ex, H.sc.lg);
            ddn_nid::chase_masks_fill(H.sc.masks);
        }
    }
        wg_barrier();
        wg_barrier();
    // ---- handler wave: one decision, the whole wavefront on it (c and seq are wave-uniform) -------------------------------
    int dbg_pend = 0, dbg_free = 0; // timing experiments: requests open when this one was picked up; when the last service ended
    auto serve = [&](int c, int seq) {
        using namespace ddn_p25h;
        Scratch& sc = H.sc;
        const int kind = H.req_kind[c], hw = H.req_hw[c], nsym = H.req_n[c], o_dec = H.req_o[c], neg = H.req_neg[c];
        const int gch = ch0 + c;
        const long long dbg_t0 = (cfg.dbg & 65536) ? (long long)clock64() : 0; // timing experiments: cycles per decision
#if DDN_RX_CYCLES
        if (lane == c) {
            const int since_req = (int)dbg_t0 - H.req_t[c];
            const int since_free = dbg_free == 0 ? 0x7fffffff : (int)((unsigned)(int)dbg_t0 - (unsigned)dbg_free);
            H.rsp_pick[c] = (cfg.dbg & 268435456) ? 1000 * dbg_pend : ((cfg.dbg & 536870912) ? (since_req < since_free ? since_req : since_free) : since_req);
        }
#endif
        int dbg_path = 0;
        long long dbg_s[3] = {0, 0, 0};
        auto dbg_stamp = [&](int k) {
            if (cfg.dbg & 131072) {
                dbg_s[k] = (long long)clock64() - dbg_t0;
            }
        };
        // channel c's handler words live in lane c's registers
        int phase = __shfl(hs.phase, c), block = __shfl(hs.block, c), end = __shfl(hs.end, c), sk0 = __shfl(hs.skipdibit, c);
        int nac = __shfl(hs.nac, c), p2cc = __shfl(hs.p2_cc, c);
        int nev = __shfl(h_nev, c);
        if (H.req_nc[c]) {
            nac = 0; // noCarrier(): state->nac = 0, p2_cc stays (engine.c:1889)
        }
        if (kind == 1) {
            phase = PH_NID;
        }
        int ext = 0, more = 0;
        int ev_kind = 0, ev_a = 0, ev_b = 0;
        int32_t pay0 = 0, pay1 = 0, pay2 = 0, pay3 = 0; // what the decision decoded (cfg.event_data)
        auto slice_at = [&](int i, int& d, int& relb, int& l0, int& l1) {
            int slot = hw - nsym + i; // hw = the ring slot after the phase's last symbol
            slot += slot < 0 ? HN : 0;
            const float* e = &H.hh[slot][c][0];
            const float mx = e[1], mn = e[2];
            const float center = (mx + mn) / 2.0f;
            const ddn_sl::Thr th = {center, ((mx - center) * 5.0f / 8.0f) + center, ((mn - center) * 5.0f / 8.0f) + center, mx, mn};
            ddn_sl::slice_soft(e[0], th, neg, d, relb, l0, l1);
        };
        if (phase == PH_NID) {
            // dispatch_p25p1.c:86-143: dibit 11 of the 33 is the status symbol; the last dibit = BCH bit 62 + the parity bit.
            // First the hard dibits alone (three compares each): when they spell a code word with a defined DUID the decoder's answer
            // is that word with an error count of 0 whatever the reliabilities - the soft slice and the ladder are for the rest.
            bool nid_done = false;
            ddn_nid::NidRes r = {0, 0, 0, 0};
            {
                int hd = 0;
                if (lane < 33 && lane != 11) {
                    int slot = hw - nsym + lane;
                    slot += slot < 0 ? HN : 0;
                    const float* e = &H.hh[slot][c][0];
                    const float mx = e[1], mn = e[2], x = e[0];
                    const float center = (mx + mn) / 2.0f;
                    const float umid = ((mx - center) * 5.0f / 8.0f) + center, lmid = ((mn - center) * 5.0f / 8.0f) + center;
                    hd = (x > center) ? ((x > umid) ? (neg ? 3 : 1) : (neg ? 2 : 0)) : ((x < lmid) ? (neg ? 1 : 3) : (neg ? 0 : 2));
                }
                // lanes 0..10 and 12..32 -> bit pairs 0..31 of the word (the status symbol's lane squeezed out)
                const unsigned long long bh = __ballot(hd & 2), bl = __ballot(hd & 1);
                const uint32_t ph = (uint32_t)((bh & 0x7FFull) | ((bh >> 12) << 11)), pl = (uint32_t)((bl & 0x7FFull) | ((bl >> 12) << 11));
                auto spread = [](uint32_t v) { // bit k -> bit 2 k
                    uint64_t x = v;
                    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
                    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
                    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
                    x = (x | (x << 2)) & 0x3333333333333333ull;
                    x = (x | (x << 1)) & 0x5555555555555555ull;
                    return x;
                };
                const uint64_t both = spread(ph) | (spread(pl) << 1); // bit 2 k = high bit of dibit k, bit 2 k + 1 = its low bit
                const uint64_t w0 = both & 0x7FFFFFFFFFFFFFFFull;
                const int par0 = (int)(both >> 63);
                dbg_stamp(0);
                if (ddn_nid::bch_63_16_is_codeword(w0)) {
                    r = ddn_nid::nid_fields(w0, 0, par0);
                    nid_done = r.status > 0;
                }
                dbg_stamp(2);
            }
            if (!nid_done) {
            if (lane < 33 && lane != 11) {
                int d, relb, l0, l1;
                slice_at(lane, d, relb, l0, l1);
                const int a0 = l0 < 0 ? -l0 : l0, a1 = l1 < 0 ? -l1 : l1; // p25p1_llr_reliability()
                const int b = (lane < 11) ? 2 * lane : 2 * (lane - 1);
                sc.nb[b] = (uint8_t)((d >> 1) & 1);
                sc.nr[b] = (uint8_t)(a0 > 255 ? 255 : a0);
                sc.nb[b + 1] = (uint8_t)(d & 1); // for lane 32: index 63 = the parity bit
                sc.nr[b + 1] = (uint8_t)(a1 > 255 ? 255 : a1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            dbg_stamp(0);
            const uint64_t w = __ballot(lane < 63 && sc.nb[lane] != 0);
            const int par = sc.nb[63], prel = sc.nr[63];
            const int observed = (nac > 0 && nac < 0xFFF) ? nac : ((p2cc > 0 && p2cc < 0xFFF) ? p2cc : 0);
            const ddn_nid::Gf gf = {sc.ex, sc.lg};
            const ddn_nid::Work wk = {sc.work + lane, sc.work + 23 * 64 + lane, sc.work + 47 * 64 + lane, sc.work + 71 * 64 + lane};
            r = ddn_nid::nid_decode_wave(gf, wk, w, sc.nr, par, prel, observed, cfg.nid_threshold, sc.masks, lane);
            }
            dbg_stamp(1);
            int duid = 0xFF;
            if (r.status > 0) { // p25p1_handle_nid_decode_success(): NAC / DUID updates
                const bool valid = r.nac != 0 && r.nac != 0xFFF;
                if (r.nac != nac && valid) {
                    nac = r.nac;
                    p2cc = r.nac;
                }
                duid = r.duid;
            }
            ev_kind = EV_NID;
            ev_a = r.status;
            ev_b = (r.nac & 0xFFFF) | (duid << 16);
            pay0 = r.status, pay1 = r.nac, pay2 = r.duid, pay3 = r.errs;
            phase = PH_IDLE;
            if (duid == 0x7 || duid == 0xC) {
                phase = (duid == 0x7) ? PH_TSBK : PH_MPDU;
                block = 0;
                end = 3; // TSBK_MAX_BLOCKS; p25_mpdu_context_init()
                sk0 = 36 - 14;
                int ska;
                ext = (duid == 0x7) ? 101 : mpdu_block_symbols(sk0, ska);
                more = 1;
            } else {
                ext = duid == 0x0 ? 339 : ((duid == 0x5 || duid == 0xA) ? 807 : (duid == 0x3 ? 15 : (duid == 0xF ? 159 : 0)));
            }
        } else if (phase == PH_TSBK || (phase == PH_MPDU && block == 0)) {
            // the block's data dibits, de-interleaved LLR pairs (lo 16 bits = the dibit's first bit)
            for (int r0 = 0; r0 < 2; r0++) {
                const int i = lane + 64 * r0;
                if (i < nsym) {
                    int k;
                    if (!block_is_status(sk0, i, k) && k < 98) {
                        int d, relb, l0, l1;
                        slice_at(i, d, relb, l0, l1);
                        sc.d[deinterleave98(k)] = (int32_t)((uint32_t)(uint16_t)(int16_t)l0 | ((uint32_t)(uint16_t)(int16_t)l1 << 16));
                    }
                }
            }
            int n_data;
            const int sk_after = block_counter_after(sk0, nsym, n_data);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // tsbk_decode_repetition_bytes(): list-8 decode, the first CRC16-clean candidate, else the best one.  The list's
            // first candidate is the plain best path (ddn_p25h_dev.h), so the list proper is only run when that fails its CRC.
            dbg_stamp(0);
            uint32_t by[3];
            half_rate_best_wave(sc, lane, by);
            dbg_stamp(1);
            int crc_ok = crc16_ok_wave(sc, by, lane), sel = 0;
            dbg_stamp(2);
            if (!crc_ok) {
                dbg_path = 1;
                half_rate_list_wave(sc, lane);
                const int nout = sc.n_out;
                for (int q = 0; q < nout; q++) {
                    const uint32_t cw[3] = {sc.outl[q][0], sc.outl[q][1], sc.outl[q][2]};
                    if (crc16_ok(cw)) {
                        sel = q;
                        crc_ok = 1;
                        break;
                    }
                }
                by[0] = sc.outl[sel][0];
                by[1] = sc.outl[sel][1];
                by[2] = sc.outl[sel][2];
            }
            const int byte0 = by[0] & 0xFF;
            pay0 = (int32_t)by[0], pay1 = (int32_t)by[1], pay2 = (int32_t)by[2];
            pay3 = (crc_ok & 1) | ((sel & 0xFF) << 8) | ((phase == PH_TSBK ? block : 0) << 16);
            sk0 = sk_after;
            if (phase == PH_TSBK) {
                const int last = (byte0 >> 7) & 1;
                ev_kind = EV_TSBK;
                ev_a = block;
                ev_b = crc_ok | (((by[0] >> 8) & 0xFF) << 8) | (((last << 8) | sel) << 16); // bits 8..15: byte 1 of the block
                block++;
                if (last || block >= 3) {
                    phase = PH_IDLE;
                } else {
                    ext = 101;
                    more = 1;
                }
            } else {
                // p25_mpdu_update_header_from_first_block(): aggressive_framesync = 1 keeps the defaults on a bad header CRC
                if (crc_ok) {
                    const int sap = (by[0] >> 8) & 0x3F, blks = (by[1] >> 16) & 0x7F;
                    end = blks + 1;
                    if ((sap == 61 || sap == 63) && blks > 10) {
                        end = 4;
                    }
                }
                ev_kind = EV_MPDU;
                ev_a = crc_ok;
                ev_b = (end & 0xFFFF) | (byte0 << 16);
                block = 1;
                // the remaining repetitions are read without a decision: their symbol counts follow from the status counter
                int total = 0;
                for (int bi = 1; bi < end; bi++) {
                    int ska;
                    total += mpdu_block_symbols(sk0, ska);
                    sk0 = ska;
                }
                ext = total;
                more = 0;
                phase = PH_IDLE;
            }
        } else {
            phase = PH_IDLE; // a request without a phase (not reachable): the handler returns
        }
        if (lane == c) {
            hs.phase = phase;
            hs.block = block;
            hs.end = end;
            hs.skipdibit = sk0;
            hs.nac = nac;
            hs.p2_cc = p2cc;
            h_served = seq;
            H.rsp_ext[c] = ext;
            H.rsp_more[c] = more;
#if DDN_RX_CYCLES
            H.rsp_t[c] = (int)clock64();
#endif
        }
#if DDN_RX_CYCLES
        dbg_free = (int)clock64();
#endif
        if (lane == c) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __hip_atomic_store(&H.rsp_seq[c], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ev_kind) {
                if (nev < cfg.max_events && gch < n_channels) {
                    int32_t* e = events + ((size_t)gch * cfg.max_events + nev) * 4;
                    e[0] = o_dec;
                    e[1] = ev_kind;
                    e[2] = ev_a;
                    e[3] = ev_b;
                    if (cfg.event_data) {
                        *reinterpret_cast<int4*>(cfg.event_data + ((size_t)gch * cfg.max_events + nev) * 4) = make_int4(pay0, pay1, pay2, pay3);
                    }
                    if (cfg.dbg & 65536) {
                        e[2] = dbg_path;
                        e[3] = (int)((long long)clock64() - dbg_t0);
                        if (cfg.dbg & 131072) {
                            e[2] = dbg_path | ((int)(dbg_s[0] >> 4) << 4);
                            e[3] = (int)(dbg_s[1] >> 4) | ((int)(dbg_s[2] >> 4) << 16);
                        }
#if DDN_RX_CYCLES
                        if (cfg.dbg & 1073741824) { // how long the request lay unserved, per decision
                            e[2] = H.rsp_pick[c];
                        }
#endif
                    }
                }
                h_nev = nev + 1;
            }
        }
    };

        // ---- handler wave: the matched filter of one pass = 128 outputs (one channel's tile at 128-sample tiles, two channels' at
        // 64).  Lane k takes outputs k and k + 64 as one packed pair: tap i reads x[k + i - 90] and x[k + 64 + i - 90] - consecutive
        // lanes, consecutive words, no bank conflict, one LDS read + one packed multiply + one packed add per tap; the products are
        // added oldest sample first, multiply and add rounded apart, as apply_sps_fir does.  The 90 samples before a tile are the
        // previous tile's (the ring keeps it; the call's first tile has the carried filter memory there, staged by the prologue).
        constexpr int LPR_H = CPW / NRW_T;
        constexpr int NPASS = LPR_H / 2; // two channels a pass (fuse_mf implies 128-sample tiles)
        static_assert(LPR_H % 2 == 0, "the filter pass takes the channels of a recurrence wave in pairs");
        typedef float mf2h __attribute__((ext_vector_type(2)));
        auto mf_pass = [&](int h, int t, int p) {
            // two channels a pass: their accumulator chains take turns on the vector unit (multiply A, multiply B, add A, add B - every
            // operand was produced at least two instructions earlier, so nothing waits for a dependent result: ~4.7 cycles per
            // instruction instead of ~14 per tap with one chain)
            const int slot = t % 3;
            const int cA = h * LPR_H + 2 * p, cB = cA + 1;
            const float* ra = &L.raw[cA][TW + slot * TW + 2 * lane - (NT - 1)];
            const float* rb = &L.raw[cB][TW + slot * TW + 2 * lane - (NT - 1)];
            mf2h accA = {0.0f, 0.0f}, accB = {0.0f, 0.0f};
            {
                // lane k takes outputs 2 k and 2 k + 1 of both channels: {x[2 k + i - 90], x[2 k + 1 + i - 90]} is ONE ds_read2_b32 (offsets i
                // and i + 1 words; sixteen lanes = thirty-two consecutive words = every bank once.  Outputs k and k + 64 - offsets i and
                // i + 64 - put a lane's two words in one bank: measured 3.6 k cycles a pass).  Left to itself the compiler pairs adjacent
                // taps of one output instead and re-sorts them with moves, waiting on every read; so the reads are spelled out,
                // seven taps of both channels a group (13 x 7 = 91), the next group's in flight under this group's arithmetic.  LDS
                // operations of a wave complete in order: lgkmcnt(14) = everything but the fourteen reads issued last.
                constexpr int G = 7;
                static_assert(NT == 13 * G, "thirteen groups of seven taps");
                uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)ra;
                uint32_t lb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)rb;
#define DDN_MF_LOAD(d, e, a, b)                                                                                                    \
    asm volatile("ds_read2_b32 %0, %14 offset1:1\n\tds_read2_b32 %7, %15 offset1:1\n\t"                                          \
                 "ds_read2_b32 %1, %14 offset0:1 offset1:2\n\tds_read2_b32 %8, %15 offset0:1 offset1:2\n\t"                     \
                 "ds_read2_b32 %2, %14 offset0:2 offset1:3\n\tds_read2_b32 %9, %15 offset0:2 offset1:3\n\t"                     \
                 "ds_read2_b32 %3, %14 offset0:3 offset1:4\n\tds_read2_b32 %10, %15 offset0:3 offset1:4\n\t"                    \
                 "ds_read2_b32 %4, %14 offset0:4 offset1:5\n\tds_read2_b32 %11, %15 offset0:4 offset1:5\n\t"                    \
                 "ds_read2_b32 %5, %14 offset0:5 offset1:6\n\tds_read2_b32 %12, %15 offset0:5 offset1:6\n\t"                    \
                 "ds_read2_b32 %6, %14 offset0:6 offset1:7\n\tds_read2_b32 %13, %15 offset0:6 offset1:7"                        \
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(e[0]),         \
                   "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6])                                    \
                 : "v"(a), "v"(b)                                                                                                  \
                 : "memory")
#define DDN_MF_WAIT(d, e, cnt)                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(" cnt ")"                                                                                      \
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(e[0]), "+v"(e[1]),     \
                   "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]))
                mf2h xa0[G], xb0[G], xa1[G], xb1[G];
                DDN_MF_LOAD(xa0, xb0, la, lb);
#pragma unroll
                for (int g = 0; g < 13; g++) {
                    mf2h* ca = (g & 1) ? xa1 : xa0;
                    mf2h* cb = (g & 1) ? xb1 : xb0;
                    if (g + 1 < 13) {
                        la += 4u * G;
                        lb += 4u * G;
                        if (g & 1) {
                            DDN_MF_LOAD(xa0, xb0, la, lb);
                        } else {
                            DDN_MF_LOAD(xa1, xb1, la, lb);
                        }
                        DDN_MF_WAIT(ca, cb, "14");
                    } else {
                        DDN_MF_WAIT(ca, cb, "0");
                    }
#pragma unroll
                    for (int k = 0; k < G; k++) {
                        const float tp = __uint_as_float(ddn_p25_filter_bits[g * G + k]);
                        const mf2h tt = {tp, tp};
                        const mf2h pa = tt * ca[k], pb = tt * cb[k];
                        accA += pa;
                        accB += pb;
                    }
                }
#undef DDN_MF_LOAD
#undef DDN_MF_WAIT
            }
            float* fa = &L.flt[cA][TW + slot * TW + 2 * lane];
            float* fb = &L.flt[cB][TW + slot * TW + 2 * lane];
            fa[0] = accA.x;
            fa[1] = accA.y;
            fb[0] = accB.x;
            fb[1] = accB.y;
            if (slot == 2) {
                fa[-3 * TW] = accA.x;
                fa[1 - 3 * TW] = accA.y;
                fb[-3 * TW] = accB.x;
                fb[1 - 3 * TW] = accB.y;
            }
        };
        const int n_tiles = (int)((n + TW - 1) / TW);
        int mfj[NRW_T], mfp[NRW_T];
        for (int w = 0; w < NRW_T; w++) {
            mfj[w] = 0;
            mfp[w] = 0;
        }
        float idle_spin = 0.0f;
#if DDN_RX_CYCLES
        long long dbg_mf[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // filter passes: cycles, count; decisions: cycles, count; idle polls
#endif
        while (true) {
            // requests first (a lane of a recurrence wave - and whoever shares its trips - waits on every decision), then one pass of
            // whichever half has a staged tile without its filter row, then nothing
            const int rq = (lane < CPW) ? __hip_atomic_load(&H.req_seq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
            const unsigned long long pend = __ballot(lane < CPW && rq != h_served);
            if (pend != 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int c = __ffsll((long long)pend) - 1;
                dbg_pend = __popcll(pend);
#if DDN_RX_CYCLES
                const long long sv_t0 = (long long)clock64();
#endif
                serve(c, __shfl(rq, c));
#if DDN_RX_CYCLES
                dbg_mf[2] += (long long)clock64() - sv_t0;
                dbg_mf[3]++;
#endif
                continue;
            }
            bool did = false;
            if (fuse_mf) {
                // the half whose filter rows are furthest behind goes first (its recurrence wave is the one that may be waiting)
                int hsel = -1, jsel = 0x7fffffff;
#pragma unroll
                for (int h = 0; h < NRW_T; h++) {
                    if (mfj[h] < n_tiles && mfj[h] < jsel
                        && __hip_atomic_load(&H.staged[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > mfj[h]) {
                        hsel = h;
                        jsel = mfj[h];
                    }
                }
                if (hsel >= 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    int pj = 0;
#pragma unroll
                    for (int h = 0; h < NRW_T; h++) {
                        pj = h == hsel ? mfp[h] : pj;
                    }
#if DDN_RX_CYCLES
                    const long long mf_t0 = (long long)clock64();
#endif
                    mf_pass(hsel, jsel, pj);
#if DDN_RX_CYCLES
                    dbg_mf[0] += (long long)clock64() - mf_t0;
                    dbg_mf[1]++;
#endif
                    pj++;
                    const bool tile_filtered = pj == NPASS;
#pragma unroll
                    for (int h = 0; h < NRW_T; h++) {
                        if (h == hsel) {
                            mfp[h] = tile_filtered ? 0 : pj;
                            mfj[h] += tile_filtered ? 1 : 0;
                        }
                    }
                    if (tile_filtered) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            __hip_atomic_store(&H.fready[hsel], jsel + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    did = true;
                }
            }
            if (did) {
                continue;
            }
            bool all_done = true;
            for (int w = 0; w < NRW_T; w++) {
                all_done = all_done
                           && (__hip_atomic_load(&H.tile_done[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & (DDN_RXW_RAW - 1)) >= n_tiles;
            }
            if (all_done) { // (a recurrence wave never leaves a tile with a request open)
                break;
            }
            helper_idle(idle_spin, (cfg.dbg & 8388608) != 0, 4);
#if DDN_RX_CYCLES
            dbg_mf[4]++;
#endif
        }
#if DDN_RX_CYCLES
        if ((cfg.dbg & 8192) && lane == 0) { // (over the tail of the workgroup's first record area, below the other waves' blocks)
            uint8_t* d = rec + ((size_t)ch0 + 1) * max_sym * 10 - 448;
            for (int k = 0; k < 64; k++) {
                d[k] = reinterpret_cast<const uint8_t*>(dbg_mf)[k];
            }
        }
#endif
    {
        if (hlive) {
            hstate[ch] = hs;
            n_events[ch] = h_nev;
        }
        for (int k = lane; k < ddn_p25h::HN * CPW * 3; k += 64) {
            const int w = k % 3, c = (k / 3) % CPW, slot = k / (3 * CPW);
            if (ch0 + c < n_channels) {
                hh_store[((size_t)(ch0 + c) * ddn_p25h::HN + slot) * 3 + w] = H.hh[slot][c][w];
            }
        }
    }
        return;
    }
    // Wave roles.  Plain mode: wave 0 the recurrence of all CPW channels, wave 1 tile staging + slice + record stores, wave 2 the
    // window's suffix summaries.  Handler mode: the channels are split over TWO recurrence waves (waves 0 and 1, LPR = CPW / 2
    // lanes each) - a trip is only lean when every lane of its wave is, a handler decision only holds up the wave of the lane
    // that asked, and the crossing search has twice the lanes per channel - and wave 2 does both the staging and the summaries
    // (together about half a tile's time), wave 3 the handlers' decisions.
    constexpr int NRW = NRW_T;        // recurrence waves
    constexpr int LPR = CPW / NRW;    // lanes (channels) per recurrence wave
    const int wave = wave_role;
    const bool loader = wave == NRW;                      // tile staging, slice + record stores
    const bool winprep = wave == (HM ? NRW : NRW + 1);    // suffix summaries of the symbol window
    const bool recur = wave < NRW;                        // the per-channel recurrence
    const int rw = recur ? wave : 0;
    const int ch0 = lblock * CPW;
    const int ln = rw * LPR + (lane < LPR ? lane : 0);    // this lane's channel column in the workgroup's LDS arrays
    const int ch = ch0 + ln;
    const bool live = recur && lane < LPR && ch < n_channels;
    const bool use_flt = cfg.use_filter != 0;
    const bool ld_flt = use_flt && filt != nullptr; // the filter row comes from HBM (else: computed in place by the handler wave)
    const bool raw_always = !ld_flt || (cfg.dbg & 16384) != 0; // the unfiltered row is staged whatever the channels' filters do
    const float inf = __builtin_inff();
    // the recurrence wave is a latency chain: when other kernels' wavefronts share its SIMD (the front end of the next batch
    // runs beside this loop, bindings/ddn_chain.py run_pipelined3) its instructions go first
    if (recur && !(cfg.dbg & 32768)) {
        __builtin_amdgcn_s_setprio(3);
    }

    DdnRxState s;
    if (live) {
        s = state[ch];
        for (int k = 0; k < SS; k++) {
            L.sb[k][ln] = sbuf_store[(size_t)k * n_channels + ch];
        }
        for (int k = 0; k < 24; k++) {
            L.lb[k][ln] = lbuf_store[(size_t)k * n_channels + ch];
            L.sh[k][ln] = shist_store[(size_t)k * n_channels + ch];
        }
    } else {
        s = DdnRxState{};
    }
    // a bound on the magnitude of every value in the 128-symbol window (kept as a running maximum: the lean run's entry test)
    float wabs = 0.0f;
    if (live) {
        L.sidx0[0][ln] = s.sidx;
        for (int k = 0; k < SS; k++) {
            wabs = fmaxf(wabs, fabsf(L.sb[k][ln]));
        }
    }
    auto stage = [&](long t0, int slot) {
        const int tn = (int)((n - t0) < TW ? (n - t0) : TW);
        constexpr int RPP = CPW < 16 ? CPW : 16; // rows in flight per pass
#pragma unroll
        for (int half = 0; half < TW / 64; half++) { // 64 samples of a row per pass
            const int j = lane + 64 * half;          // sample of the tile
#pragma unroll
            for (int h = 0; h < CPW / RPP; h++) {
                float r[RPP], f[RPP];
#pragma unroll
                for (int c = 0; c < RPP; c++) {
                    const int cc = RPP * h + c;
                    const bool ok = (ch0 + cc < n_channels) && j < tn;
                    const size_t off = (size_t)(ch0 + cc) * stride + (size_t)t0 + j;
                    r[c] = ok ? raw[off] : 0.0f;
                    f[c] = (ok && ld_flt) ? filt[off] : 0.0f;
                }
#pragma unroll
                for (int c = 0; c < RPP; c++) {
                    L.raw[RPP * h + c][TW + slot * TW + j] = r[c];
                    if (slot == 2) {
                        L.raw[RPP * h + c][j] = r[c];
                    }
                    if (!fuse_mf) {
                        L.flt[RPP * h + c][TW + slot * TW + j] = f[c];
                        if (slot == 2) {
                            L.flt[RPP * h + c][j] = f[c];
                        }
                    }
                }
            }
        }
    };
    if (loader && n > 0) {
        stage(0, 0);
        if (fuse_mf) { // the filter's memory ahead of the call's first tile: the last 90 raw samples of the call before
            for (int k = lane; k < CPW * (NT - 1); k += 64) {
                const int c = k / (NT - 1), i = k % (NT - 1);
                L.raw[c][TW - (NT - 1) + i] = (ch0 + c < n_channels) ? prev_tail[(size_t)(ch0 + c) * (NT - 1) + i] : 0.0f;
            }
        }
    }
    if (loader) { // every queue slot starts as "nothing handed over" (and returns to that when drained)
        for (int k = lane; k < 2 * QTW * CPW; k += 64) {
            (&L.q[0][0][0][0])[4 * k + 3] = __int_as_float(-1);
        }
    }
    __syncthreads();
    const int whole0 = cfg.out_rate / cfg.sym_rate, rem0 = cfg.out_rate % cfg.sym_rate;
    const int whole = whole0 < 2 ? 2 : (whole0 > 64 ? 64 : whole0);
    const int rem = (whole0 < 2 || whole0 > 64) ? 0 : rem0;
    // most symbols two consecutive tiles can push (every symbol consumes at least sps - 1 samples)
    const int mn_raw = 2 * ((TW + whole + whole - 2) / (whole > 1 ? whole - 1 : 1) + 1);
    const int Mn = mn_raw > WMW ? WMW : mn_raw;
    // wave 2: S_m for m = Mn .. 1 from the ring as it stands at L.sidx0 (entries being overwritten meanwhile are the
    // oldest ones, which only feed summaries nobody will ask for)
    auto compute_sfx = [&](int buf, int tile_parity) {
        constexpr int EPL = 64 / CPW;
        const int c = lane % CPW, part = lane / CPW;
        const int s0 = L.sidx0[tile_parity][c];
        const int total = SS - Mn, per = (total + EPL - 1) / EPL;
        const int i_lo = Mn + 1 + part * per;
        const int i_hi = (i_lo + per - 1) < SS ? (i_lo + per - 1) : SS;
        float a1 = inf, a2 = inf, b1 = -inf, b2 = -inf;
        for (int i = i_lo; i <= i_hi; i++) {
            const float v = L.sb[(s0 + i - 1) & (SS - 1)][c];
            two_min_insert(v, a1, a2);
            two_max_insert(v, b1, b2);
        }
#pragma unroll
        for (int d = CPW; d < 64; d <<= 1) {
            const float o1 = __shfl_xor(a1, d), o2 = __shfl_xor(a2, d), p1 = __shfl_xor(b1, d), p2 = __shfl_xor(b2, d);
            two_min_insert(o1, a1, a2);
            two_min_insert(o2, a1, a2);
            two_max_insert(p1, b1, b2);
            two_max_insert(p2, b1, b2);
        }
        for (int m = Mn; m >= 1; m--) {
            if (m < Mn) {
                const float v = L.sb[(s0 + m) & (SS - 1)][c]; // entry m + 1
                two_min_insert(v, a1, a2);
                two_max_insert(v, b1, b2);
            }
            if (part == 0) {
                *reinterpret_cast<float4*>(&L.sfx[buf][m - 1][c][0]) = make_float4(a1, a2, b1, b2);
            }
        }
    };
    if (winprep) {
        compute_sfx(0, 0);
    }
    __syncthreads();

    int o = 0;
    uint8_t* rp = rec + (size_t)(live ? ch : 0) * max_sym * 10;
    uint8_t* fp = flags + (size_t)(live ? ch : 0) * max_sym;
    const long long abs0 = s.n_abs;

    const bool offload = whole >= 8;
    // the ordinary symbol length: constant span with the five-sample window (src/dsp/dsd_symbol.c:405-426 special-cases 5 / 20)
    const bool stdspan = rem == 0 && whole >= 6 && whole <= 11; // whole + 1 samples at most with a late slip: the search covers 12
    const bool std_ok = stdspan && !(cfg.dbg & 1024); // see "standard trip" in the trip loop
    const bool lean_ok = offload && stdspan && !(cfg.dbg & 2048); // see "lean trip" in the trip loop
    const bool bulk_ok = std_ok && offload && !(cfg.dbg & 1048576); // see "bulk hunting pass" in the trip loop
    // bulk hunting pass: the slip of a symbol's start by the latched crossing jit (-1 .. whole), as (slip + 1) in two bits per jit + 1
    uint32_t i0tab = 0;
    for (int jt = -1; jt <= 14; jt++) {
        const int i0 = (jt > 0 && jt <= (whole - 1) / 2) ? -1 : ((jt > (whole - 1) / 2 && jt < whole) ? 1 : 0);
        i0tab |= (uint32_t)(i0 + 1) << (2 * (jt + 1));
    }
    i0tab = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0tab);
    auto store_record = [&](uint8_t* r, uint8_t* f, float sym, int dibit, int relb, int l0, int l1, int fl) {
        const uint32_t xb = __float_as_uint(sym);
        ((uint16_t*)r)[0] = (uint16_t)((dibit & 3) | (relb << 8));
        ((uint16_t*)r)[1] = (uint16_t)(int16_t)l0;
        ((uint16_t*)r)[2] = (uint16_t)(int16_t)l1;
        ((uint16_t*)r)[3] = (uint16_t)(xb & 0xFFFFu);
        ((uint16_t*)r)[4] = (uint16_t)(xb >> 16);
        *f = (uint8_t)fl;
    };
    // slice + soft decision + record store of the symbols queued during one tile: entries are independent, so the wave's 64
    // lanes take 64 / CPW queue entries of every channel at once (lane -> channel lane % CPW, entry lane / CPW + ...)
    auto drain = [&](int qb) {
        if (!loader) {
            return;
        }
        constexpr int EPL = 64 / CPW;
        const int dc = lane % CPW, de = lane / CPW;
        const int dch = ch0 + dc;
        if (dch >= n_channels) {
            return;
        }
        const int cnt = L.qn[qb][dc / LPR], o0 = L.qo[qb][dc];
        uint8_t* drp = rec + (size_t)dch * max_sym * 10;
        uint8_t* dfp = flags + (size_t)dch * max_sym;
        for (int k = de; k < QTW; k += EPL) {
            if (k >= cnt) {
                break;
            }
            const float4 e = *reinterpret_cast<const float4*>(&L.q[qb][k][dc][0]);
            const int fw = __float_as_int(e.w);
            if (fw < 0) {
                continue; // this lane handed nothing over in that trip
            }
            L.q[qb][k][dc][3] = __int_as_float(-1); // slots read "nothing" until written again: a lean run only writes its own lanes'
            const float sym = e.x;
            const int fl = fw & 0xFF;
            int dibit, relb = 0, l0 = 0, l1 = 0;
            if (fl & 1) {
                const float mx = e.y, mn = e.z;
                const float center = (mx + mn) / 2.0f;
                const ddn_sl::Thr th = {center, ((mx - center) * 5.0f / 8.0f) + center,
                                        ((mn - center) * 5.0f / 8.0f) + center, mx, mn};
                ddn_sl::slice_soft(sym, th, (fl >> 2) & 1, dibit, relb, l0, l1);
            } else {
                dibit = sym > 0.0f ? 1 : 3;
            }
            const size_t oo = (size_t)(o0 + (fw >> 8));
            if (oo < max_sym) {
                store_record(drp + oo * 10, dfp + oo, sym, dibit, relb, l0, l1, fl);
            }
        }
    };
    // Handler mode: the staging wave's three jobs for the channels of ONE recurrence wave (h) - the two recurrence waves share
    // nothing but this wave's time, so each runs as far ahead of the other as its own channels let it.
    // (round 5) in two parts - the loads of a tile go out before the staging wave's other two jobs and land in the LDS after them:
    // ~3 k cycles of memory latency per half and tile that the wave used to sit out (the staging wave is what a tile of quiet
    // traffic waits for)
    struct StageRegs {
        float r[TW / 64][LPR], f[TW / 64][LPR];
    };
    auto stage_half_load = [&](long t0, int h, StageRegs& g, bool need) {
        const int tn = (int)((n - t0) < TW ? (n - t0) : TW);
        // A channel whose matched filter is on and warm reads the filter's row only, so the unfiltered samples of a half whose channels
        // are all in that state stay in HBM (half the loop's fetches on a batch in sync).  The recurrence lanes say so a tile ahead
        // (bit 30 of tile_done, written at the end of every tile for the tile picked up here), and they can: the filter is gated off only by
        // rx_no_carrier, which takes 1800 hunting symbols without a sync to reach - a lane asks for the unfiltered row again 64
        // symbols (more than four tiles' worth) before that count can be reached, keeps asking while the filter is off and through
        // its next cold start, and always asks where the row is needed whatever the filter does (no filter row in HBM; cfg.dbg bit
        // 16384: staged always, as before).  One decision for the half: both forms of the staging are straight-line code.
#pragma unroll
        for (int half = 0; half < TW / 64; half++) { // every load of the tile in flight before the first LDS write
            const int j = lane + 64 * half;
#pragma unroll
            for (int c = 0; c < LPR; c++) {
                const int cc = h * LPR + c;
                const bool ok = (ch0 + cc < n_channels) && j < tn;
                const size_t off = (size_t)(ch0 + cc) * stride + (size_t)t0 + j;
                g.f[half][c] = (ok && ld_flt) ? filt[off] : 0.0f;
                g.r[half][c] = 0.0f;
            }
        }
        if (need) {
#pragma unroll
            for (int half = 0; half < TW / 64; half++) {
                const int j = lane + 64 * half;
#pragma unroll
                for (int c = 0; c < LPR; c++) {
                    const int cc = h * LPR + c;
                    const bool ok = (ch0 + cc < n_channels) && j < tn;
                    const size_t off = (size_t)(ch0 + cc) * stride + (size_t)t0 + j;
                    g.r[half][c] = ok ? raw[off] : 0.0f;
                }
            }
        }
    };
    auto stage_half_store = [&](int tile, int h, const StageRegs& g) {
        const int slot = tile % 3;
#pragma unroll
        for (int half = 0; half < TW / 64; half++) {
            const int j = lane + 64 * half;
#pragma unroll
            for (int c = 0; c < LPR; c++) {
                const int cc = h * LPR + c;
                L.raw[cc][TW + slot * TW + j] = g.r[half][c]; // (zeros where the row was left in HBM: never read)
                if (slot == 2) {
                    L.raw[cc][j] = g.r[half][c];
                }
                if (!fuse_mf) {
                    L.flt[cc][TW + slot * TW + j] = g.f[half][c];
                    if (slot == 2) {
                        L.flt[cc][j] = g.f[half][c];
                    }
                }
            }
        }
    };
    auto compute_sfx_half = [&](int buf, int tile_parity, int h) {
        // S_m = summary of ring entries m + 1 .. SS for m = Mn .. 1, all of them at once: the 64 / LPR lanes of a channel's
        // column first share the entries past Mn (B), then each takes K consecutive m: its own entries' summary, a suffix scan
        // of those over the column (what the parts after it hold), and its K results on top of B and that.  (Two smallest /
        // two largest of a union of disjoint sets do not depend on the order of the merges.)
        constexpr int EPL = 64 / LPR;
        const int c = h * LPR + lane % LPR, part = lane / LPR;
        const int s0 = L.sidx0[tile_parity][c];
        const int total = SS - Mn, per = (total + EPL - 1) / EPL;
        const int i_lo = Mn + 1 + part * per;
        const int i_hi = (i_lo + per - 1) < SS ? (i_lo + per - 1) : SS;
        float a1 = inf, a2 = inf, b1 = -inf, b2 = -inf;
        for (int i = i_lo; i <= i_hi; i++) {
            const float v = L.sb[(s0 + i - 1) & (SS - 1)][c];
            two_min_insert(v, a1, a2);
            two_max_insert(v, b1, b2);
        }
#pragma unroll
        for (int d = LPR; d < 64; d <<= 1) {
            const float o1 = __shfl_xor(a1, d), o2 = __shfl_xor(a2, d), p1 = __shfl_xor(b1, d), p2 = __shfl_xor(b2, d);
            two_min_insert(o1, a1, a2);
            two_min_insert(o2, a1, a2);
            two_max_insert(p1, b1, b2);
            two_max_insert(p2, b1, b2);
        }
        const int K = (Mn + EPL - 1) / EPL;   // m values per part: m = part * K + 1 .. part * K + K (those <= Mn)
        const int m_lo = part * K + 1;
        // own entries: e_(m + 1) for the part's m, as far as they lie inside 2 .. Mn
        float ev[3] = {0.0f, 0.0f, 0.0f};
        float c1 = inf, c2 = inf, d1 = -inf, d2 = -inf;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int m = m_lo + k;
            if (k < K && m + 1 <= Mn) {
                ev[k] = L.sb[(s0 + m) & (SS - 1)][c]; // entry m + 1
                two_min_insert(ev[k], c1, c2);
                two_max_insert(ev[k], d1, d2);
            }
        }
        // inclusive suffix over the parts, then shifted by one part
#pragma unroll
        for (int d = 1; d < EPL; d <<= 1) {
            const float o1 = __shfl_down(c1, d * LPR), o2 = __shfl_down(c2, d * LPR);
            const float p1 = __shfl_down(d1, d * LPR), p2 = __shfl_down(d2, d * LPR);
            if (part + d < EPL) {
                two_min_insert(o1, c1, c2);
                two_min_insert(o2, c1, c2);
                two_max_insert(p1, d1, d2);
                two_max_insert(p2, d1, d2);
            }
        }
        {
            const float o1 = __shfl_down(c1, LPR), o2 = __shfl_down(c2, LPR), p1 = __shfl_down(d1, LPR), p2 = __shfl_down(d2, LPR);
            if (part + 1 < EPL) {
                two_min_insert(o1, a1, a2);
                two_min_insert(o2, a1, a2);
                two_max_insert(p1, b1, b2);
                two_max_insert(p2, b1, b2);
            }
        }
        // the part's own m, from the highest down: each adds entry m + 1
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            const int m = m_lo + k;
            if (k < K && m <= Mn) {
                if (m + 1 <= Mn) {
                    two_min_insert(ev[k], a1, a2);
                    two_max_insert(ev[k], b1, b2);
                }
                *reinterpret_cast<float4*>(&L.sfx[buf][m - 1][c][0]) = make_float4(a1, a2, b1, b2);
            }
        }
    };
    auto drain_half = [&](int qb, int h) {
        constexpr int EPL = 64 / LPR;
        const int dc = h * LPR + lane % LPR, de = lane / LPR;
        const int dch = ch0 + dc;
        if (dch >= n_channels) {
            return;
        }
        const int cnt = L.qn[qb][h], o0 = L.qo[qb][dc];
        uint8_t* drp = rec + (size_t)dch * max_sym * 10;
        uint8_t* dfp = flags + (size_t)dch * max_sym;
        for (int k = de; k < QTW; k += EPL) {
            if (k >= cnt) {
                break;
            }
            const float4 e = *reinterpret_cast<const float4*>(&L.q[qb][k][dc][0]);
            const int fw = __float_as_int(e.w);
            if (fw < 0) {
                continue; // this lane handed nothing over in that trip
            }
            L.q[qb][k][dc][3] = __int_as_float(-1); // slots read "nothing" until written again: a lean run only writes its own lanes'
            const float sym = e.x;
            const int fl = fw & 0xFF;
            int dibit, relb = 0, l0 = 0, l1 = 0;
            if (fl & 1) {
                const float mx = e.y, mn = e.z;
                const float center = (mx + mn) / 2.0f;
                const ddn_sl::Thr th = {center, ((mx - center) * 5.0f / 8.0f) + center,
                                        ((mn - center) * 5.0f / 8.0f) + center, mx, mn};
                ddn_sl::slice_soft(sym, th, (fl >> 2) & 1, dibit, relb, l0, l1);
            } else {
                dibit = sym > 0.0f ? 1 : 3;
            }
            const size_t oo = (size_t)(o0 + (fw >> 8));
            if (oo < max_sym) {
                store_record(drp + oo * 10, dfp + oo, sym, dibit, relb, l0, l1, fl);
            }
        }
    };
    // window state of this lane: summaries {min1, min2, max1, max2} of the symbols pushed during this tile (pc) and during the
    // previous one (pp), their counts, and which checkpoint's suffix summaries apply
    float pc1 = inf, pc2 = inf, pc3 = -inf, pc4 = -inf;
    float pp1 = inf, pp2 = inf, pp3 = -inf, pp4 = -inf;
    int npc = 0, npp = 0, sbuf_sel = 0;
    // window push of one symbol at slot s.sidx; returns the whole-window extrema pairs when `global` is set
    auto window_push = [&](float sym, bool global, float& m1, float& m2, float& x1, float& x2, int) {
        L.sb[s.sidx][ln] = sym;
        two_min_insert(sym, pc1, pc2);
        two_max_insert(sym, pc3, pc4);
        npc++;
        wabs = fmaxf(wabs, fabsf(sym));
        if (global) {
            int m = npp + npc; // ring entries replaced since the checkpoint
            m = m > WMW ? WMW : m;
            const float4 sv = *reinterpret_cast<const float4*>(&L.sfx[sbuf_sel][m - 1][ln][0]);
            const float s1 = sv.x, s2 = sv.y, s3 = sv.z, s4 = sv.w;
            // two smallest of three sorted pairs, two largest likewise (values are finite: min / max pick the same multiset)
            float t1 = fminf(s1, pp1), t2 = fminf(fmaxf(s1, pp1), fminf(s2, pp2));
            m1 = fminf(t1, pc1);
            m2 = fminf(fmaxf(t1, pc1), fminf(t2, pc2));
            float u1 = fmaxf(s3, pp3), u2 = fmaxf(fminf(s3, pp3), fmaxf(s4, pp4));
            x1 = fmaxf(u1, pc3);
            x2 = fmaxf(fminf(u1, pc3), fmaxf(u2, pc4));
        }
    };

    const float* rrow = &L.raw[ln][0];
    const float* frow = &L.flt[ln][0];
    int sp = 0; // this lane's cursor relative to the current tile (negative: a deferred symbol begins in the previous one)
    int it = 0;
    double fill_min_d = (double)s.fill_min, fill_max_d = (double)s.fill_max;
    // this lane's slot in the extrema rings as a 32-bit byte offset (uniform base + lane offset addressing)
    // (round 6) rings laid out [channel][slot]: a channel's pushes of successive symbols are successive words of one cache line, which
    // the L2 merges before it writes them back.  [slot][channel] made every push a 16-byte piece (the four lanes of a recurrence wave)
    // of a line it shares with other workgroups: written back piece by piece, 3.4 x the payload in WRITE_SIZE.
    const uint32_t ro_step = 4u, ro_first = 4u * (uint32_t)MS * (uint32_t)ch;
    uint32_t ro = (uint32_t)s.midx * ro_step + ro_first;
    auto ring_at = [](float* ring, uint32_t off) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(ring) + off); };
    // ---- per-symbol commit, shared by the fast paths and the generic path of the trip loop ---------------------------
    long t0 = 0;  // call-relative index of the current tile's first sample (the commit stamps filt_start with it)
    int cold_until = 0; // see cold_limit()
    int tk = 0;   // trip of the current tile (= queue slot handed to wave 1)
    int o_tile = 0; // output index at the start of the current tile
    float4 qv = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1)); // this trip's queue entry (-1: nothing handed over)
    int itq = 0;  // queue half of the current tile
    auto timing_reset = [&]() {
        rx_timing_reset(s);
        fill_min_d = (double)s.fill_min;
        fill_max_d = (double)s.fill_max;
        ro = ro_first;
    };
    // the matched filter's memory at the moment it is gated off: the last 90 samples it was fed (older slots keep what
    // an earlier memory held), kept per channel for the cold start of the next enable
    auto snapshot_filter = [&]() {
        if (!s.filter_on) {
            return;
        }
        const long long tnext = abs0 + t0 + sp;
        float* fs = fstale + (size_t)ch * (NT - 1);
        for (int k = 0; k < NT - 1; k++) {
            const long long ja = tnext - (NT - 1) + k;
            float v;
            if (ja >= s.filt_start) {
                const long jc = (long)(ja - abs0);
                v = (jc >= 0) ? raw[(size_t)ch * stride + jc] : prev_tail[(size_t)ch * (NT - 1) + (NT - 1) + jc];
            } else {
                v = fs[(int)(ja - s.filt_start) + (NT - 1)];
            }
            fs[k] = v;
        }
    };
    // frame_sync_advance_sync_window() + frame_sync_handle_no_sync_timeout() after a hunting symbol without a sync
    auto hunt_advance = [&]() {
        if (s.hunt_pos < 10200) {
            s.hunt_pos++;
        } else {
            s.hunt_pos = 0;
            snapshot_filter();
            rx_no_carrier(s);
            s.hnc = 1;
            cold_until = -2147483647;
        }
        if (s.lastsync != 2 && s.hunt_pos >= 1800) {
            snapshot_filter();
            rx_no_carrier(s);
            s.hnc = 1;
            cold_until = -2147483647;
            rx_hunt_restart(s);
        }
    };
    // dsd_symbol_history_push()
    auto commit_pre = [&](float sym) {
        L.sh[s.shead][ln] = sym;
        s.shead = (s.shead + 1 >= 24) ? 0 : s.shead + 1;
        s.scount = s.scount < 24 ? s.scount + 1 : 24;
    };
    // the handler has returned (or the configured count ran out): the next getFrameSync() call starts
    auto frame_end = [&]() {
        s.hunt_pos = 0;
        // the mid thresholds are read by nobody inside a frame (wave 1 derives its own from max / min): they are
        // brought up to date when the frame ends (and at the end of the call, below)
        s.umid = ((s.max - s.center) * 5.0f / 8.0f) + s.center;
        s.lmid = ((s.min - s.center) * 5.0f / 8.0f) + s.center;
        s.have_sync = 0;
        s.lidx = 0;
        s.level_count = 0;
        s.hist_count = 0;
        s.hist_bits = 0;
        s.lmin = s.min;
        s.lmax = s.max;
    };
    // handler mode, recurrence side: hpost = this symbol ends a phase, hwait = the lane sits out until the answer is there
    bool hpost = false, hwait = false;
    int hseq = 0;
    auto post_request = [&](int o_last) { // o_last: output index of the phase's last symbol
        {
            hpost = false;
            hwait = true;
            hseq++;
            if (cfg.dbg & 65536) {
                s.dbg_nreq++;
                if (!(cfg.dbg & (67108864 | 134217728))) {
                    s.dbg_wait -= (long long)clock64();
                }
#if DDN_RX_CYCLES
                H.req_t[ln] = (int)clock64();
#endif
            }
            H.req_kind[ln] = s.hphase;
            H.req_hw[ln] = s.hw;
            H.req_n[ln] = s.hn;
            H.req_o[ln] = o_last;
            H.req_neg[ln] = (s.lastsync == 2) ? 1 : 0;
            H.req_nc[ln] = s.hnc;
            s.hnc = 0;
            // the request is LDS traffic only (a wave's LDS operations complete in order): no wait for the ring stores in flight
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __hip_atomic_store(&H.req_seq[ln], hseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto hist_push = [&](float sym, float q_max, float q_min, int fl) {
        if (HM && live && s.hphase != 0) {
            float* e = &H.hh[s.hw][ln][0];
            e[0] = sym;
            e[1] = q_max;
            e[2] = q_min;
            s.hw = (s.hw + 1 >= ddn_p25h::HN) ? 0 : s.hw + 1;
        }
        if (HM && hpost) {
            post_request(o);
        }
    };
    auto commit_inframe = [&](float sym, int done_snap, int& fl, float& q_max, float& q_min) {
        // get_dibit_and_analog_signal(): window, extrema rings, thresholds (slice + soft decision on wave 1)
        const int neg = (s.lastsync == 2);
        float m1, m2, x1, x2;
        window_push(sym, true, m1, m2, x1, x2, done_snap);
        const float lo = (m1 + m2) * 0.5f, hi = (x1 + x2) * 0.5f;
        double old_lo = fill_min_d, old_hi = fill_max_d; // a ring refilled by a warm start holds one value (k * v exact, §5b)
        if (s.since_fill >= MS) {
            old_lo = (double)ring_at(minring, ro);
            old_hi = (double)ring_at(maxring, ro);
        } else {
            s.since_fill++;
        }
        s.min_sum += (double)lo - old_lo;
        s.max_sum += (double)hi - old_hi;
        ring_at(minring, ro) = lo;
        ring_at(maxring, ro) = hi;
        ro += ro_step;
        if (++s.midx >= MS) {
            s.midx = 0;
            ro = ro_first;
        }
        s.min = (float)(s.min_sum / (double)MS);
        s.max = (float)(s.max_sum / (double)MS);
        s.center = (s.max + s.min) / 2.0f;
        s.maxref = s.max * 0.80f;
        s.minref = s.min * 0.80f;
        s.sidx = (s.sidx >= SS - 1) ? 0 : s.sidx + 1;
        fl = 1 | (neg ? 4 : 0);
        q_max = s.max;
        q_min = s.min;
        if (--s.lock_left <= 0) {
            if (HM && s.hphase != 0) {
                hpost = true; // the handler decides how the frame goes on: asked once this symbol is in the history ring
            } else {
                frame_end();
            }
        }
    };
    auto commit_hunt = [&](float sym, int done_snap, int& fl) {
        // getFrameSync(): one hunting iteration
        L.lb[s.lidx][ln] = sym;
        s.level_count = s.level_count < 24 ? s.level_count + 1 : 24;
        float u0, u1, u2, u3;
        window_push(sym, false, u0, u1, u2, u3, done_snap);
        s.lidx = (s.lidx == 23) ? 0 : s.lidx + 1;
        s.sidx = (s.sidx >= SS - 1) ? 0 : s.sidx + 1;
        const uint32_t bit = sym > 0.0f ? 1u : 0u;
        s.hist_bits = ((s.hist_bits << 1) | bit) & 0xFFFFFFu;
        s.hist_count = s.hist_count < 24 ? s.hist_count + 1 : 24;
        if (s.hist_count >= 8) {
            s.maxref = s.max;
            s.minref = s.min;
            int pol = 0;
            if (s.hist_count >= 24) {
                pol = (s.hist_bits == kSyncBits) ? 1 : ((s.hist_bits == (~kSyncBits & 0xFFFFFFu)) ? 2 : 0);
            }
            // The level window (lmin / lmax of the last <= 24 hunting symbols) is recomputed from
            // scratch by the reference on every hunting symbol but only consumed when a sync is
            // accepted, so it is evaluated here only then: same values at the only point of use.
            if (pol) {
                const float big = 3.4028234663852886e38f;
                float a0 = big, a1 = big, a2 = big, a3 = big, a4 = big;
                float b0 = -big, b1 = -big, b2 = -big, b3 = -big, b4 = -big;
                const int lc = s.level_count;
                // (straight-line: an entry past the window's fill goes in as +big / -big, which leaves the five unchanged, so the 24
                // insertions carry no branch and stage i of entry k + 1 only waits for stage i of entry k; four entries at a time -
                // more in flight costs the kernel its register budget)
#pragma unroll 4
                for (int k = 0; k < 24; k++) {
                    const float x = L.lb[k][ln];
                    float v = k < lc ? x : big, t;
                    float w = k < lc ? x : -big;
                    t = fminf(a0, v); v = fmaxf(a0, v); a0 = t;
                    t = fminf(a1, v); v = fmaxf(a1, v); a1 = t;
                    t = fminf(a2, v); v = fmaxf(a2, v); a2 = t;
                    t = fminf(a3, v); v = fmaxf(a3, v); a3 = t;
                    a4 = fminf(a4, v);
                    t = fmaxf(b0, w); w = fminf(b0, w); b0 = t;
                    t = fmaxf(b1, w); w = fminf(b1, w); b1 = t;
                    t = fmaxf(b2, w); w = fminf(b2, w); b2 = t;
                    t = fmaxf(b3, w); w = fminf(b3, w); b3 = t;
                    b4 = fmaxf(b4, w);
                }
                if (lc >= 13) {
                    s.lmin = (a2 + a3 + a4) / 3.0f;
                    s.lmax = (b4 + b3 + b2) / 3.0f;
                } else {
                    s.lmin = (a0 + a1 + a2) / 3.0f;
                    s.lmax = (b2 + b1 + b0) / 3.0f;
                }
                s.max = (s.max + s.lmax) / 2;
                s.min = (s.min + s.lmin) / 2;
                s.lastsync = pol;
                if (use_flt && !s.filter_on) {
                    s.filter_on = 1;
                    s.filt_start = abs0 + t0 + sp; // first sample the filter sees
                    cold_until = sp + (NT - 1);
                }
                if (s.scount >= 24) {
                    float sp_ = 0.0f, sn_ = 0.0f;
                    int np = 0, nn = 0;
                    int idx = s.shead;
                    // (no branch: x + (+0) == x for every x but -0, and neither sum can be -0 - both start at +0 - so adding +0 to
                    // the sum a value does not belong to leaves it as the reference's conditional add does; the loads run ahead)
#pragma unroll 4
                    for (int k = 0; k < 24; k++) {
                        idx = idx == 0 ? 23 : idx - 1;
                        const float v = L.sh[idx][ln];
                        const bool pos = v > 0.0f;
                        sp_ += pos ? v : 0.0f;
                        sn_ += pos ? 0.0f : v;
                        np += pos ? 1 : 0;
                        nn += pos ? 0 : 1;
                    }
                    if (np != 0 && nn != 0) {
                        const float mp = sp_ / (float)np, mn = sn_ / (float)nn;
                        if (!(fabsf(mp - mn) < 1.0f)) {
                            s.max = mp;
                            s.min = mn;
                            s.center = (s.max + s.min) / 2.0f;
                            s.umid = s.center + (s.max - s.center) * 0.625f;
                            s.lmid = s.center + (s.min - s.center) * 0.625f;
                            s.maxref = s.max * 0.80f;
                            s.minref = s.min * 0.80f;
                            s.fill_max = s.max;
                            s.fill_min = s.min;
                            fill_max_d = (double)s.max;
                            fill_min_d = (double)s.min;
                            s.since_fill = 0;
                            s.max_sum = (double)s.max * (double)MS;
                            s.min_sum = (double)s.min * (double)MS;
                        }
                    }
                }
                s.have_sync = 1;
                s.lock_left = lock_cfg[ch]; // in-frame symbols after a sync, per channel
                if (HM) { // the NID: 32 dibits + the status symbol inside it, then p25p1_nid_decode decides
                    s.lock_left = 33;
                    s.hphase = 1;
                    s.hn = 33;
                }
                fl = 2 | (pol == 2 ? 4 : 0);
                if (s.lock_left <= 0) {
                    s.have_sync = 0;
                    s.lidx = 0;
                    s.level_count = 0;
                    s.hist_count = 0;
                    s.hist_bits = 0;
                    s.lmin = s.min;
                    s.lmax = s.max;
                    s.hunt_pos = 0;
                }
            }
        }
        if (!(fl & 2)) {
            hunt_advance();
        }
    };
    auto emit = [&](float sym, int fl, float q_max, float q_min) {
    if (fl & 1) {
        hist_push(sym, q_max, q_min, fl);
    }
    if (offload && tk <= QTW) {
        qv = make_float4(sym, q_max, q_min, __int_as_float(fl | ((o - o_tile) << 8)));
    } else if ((size_t)o < max_sym) {
        int dibit, relb = 0, l0 = 0, l1 = 0;
        if (fl & 1) {
            const float center = (q_max + q_min) / 2.0f;
            const ddn_sl::Thr th = {center, ((q_max - center) * 5.0f / 8.0f) + center,
                                    ((q_min - center) * 5.0f / 8.0f) + center, q_max, q_min};
            ddn_sl::slice_soft(sym, th, (fl >> 2) & 1, dibit, relb, l0, l1);
        } else {
            dibit = sym > 0.0f ? 1 : 3;
        }
        store_record(rp + (size_t)o * 10, fp + o, sym, dibit, relb, l0, l1, fl);
    }
        o++;
    };

    long long dbg_busy = 0, dbg_wait = 0, dbg_cyc[3] = {0, 0, 0}, dbg_prev = 0, dbg_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_st = 0;
#define DBG_SEC(k) do { if (DDN_RX_CYCLES && (cfg.dbg & 8192)) { const long long n_ = (long long)clock64(); dbg_sec[k] += n_ - dbg_st; dbg_st = n_; } } while (0)
    int dbg_n[3] = {0, 0, 0}, dbg_kind = -1;
    long long dbg_blk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_bt = 0; // bulk hunting pass, by section
#define DBG_BLK(k) do { if (DDN_RX_CYCLES && (cfg.dbg & 8192)) { const long long n_ = (long long)clock64(); dbg_blk[k] += n_ - dbg_bt; dbg_bt = n_; } } while (0)
    long long dbg_run[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // lean runs: count, trips, -, -, -, runs polling a mailbox, cycles inside the run, phases
    if (HM && loader) {
        // Handler mode, staging wave: job j of recurrence wave h's channels (stage tile j + 1, drain tile j - 1's queue, the
        // window summaries of checkpoint j) falls due when that wave has finished tile j - 1, and lets it into tile j + 1; job
        // n_tiles is the last tile's drain.  Whichever half is due is served.
        const int n_tiles = (int)((n + TW - 1) / TW);
        int job[NRW];
        for (int h = 0; h < NRW; h++) {
            job[h] = 0;
        }
        float idle_spin = 0.0f;
        auto jobs_left = [&]() {
            bool any = false;
            for (int h = 0; h < NRW; h++) {
                any = any || job[h] <= n_tiles;
            }
            return any;
        };
        while (jobs_left()) {
            bool served = false;
#pragma unroll
            for (int h = 0; h < NRW; h++) {
                const int j = job[h];
                const int tdw = __hip_atomic_load(&H.tile_done[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (j > n_tiles || (tdw & (DDN_RXW_RAW - 1)) < j) {
                    continue;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const long long w0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
                if (j < n_tiles) {
                    long long c0 = w0;
                    StageRegs sg;
                    if (j + 1 < n_tiles) {
                        stage_half_load((long)(j + 1) * TW, h, sg, (tdw & DDN_RXW_RAW) != 0);
                    }
                    if (DDN_RX_CYCLES && (cfg.dbg & 8192)) { // the staging wave's three jobs, timed apart
                        const long long c1 = (long long)clock64();
                        dbg_cyc[0] += c1 - c0;
                        c0 = c1;
                    }
                    if (offload && j > 0 && !(cfg.dbg & 512)) {
                        drain_half((j - 1) & 1, h);
                    }
                    if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                        const long long c1 = (long long)clock64();
                        dbg_cyc[1] += c1 - c0;
                        c0 = c1;
                    }
                    if (fuse_mf) { // the raw tile goes into the ring ahead of the summaries: the handler wave filters it meanwhile
                        if (j + 1 < n_tiles) {
                            stage_half_store(j + 1, h, sg);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            __hip_atomic_store(&H.staged[h], j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                            const long long c1 = (long long)clock64();
                            dbg_cyc[0] += c1 - c0;
                            c0 = c1;
                        }
                    }
                    if (j >= 1 && !(cfg.dbg & 256)) {
                        compute_sfx_half(j & 1, j & 1, h);
                    }
                    if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                        const long long c1 = (long long)clock64();
                        dbg_cyc[2] += c1 - c0;
                        c0 = c1;
                    }
                    if (!fuse_mf && j + 1 < n_tiles) {
                        stage_half_store(j + 1, h, sg);
                    }
                    if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                        dbg_cyc[0] += (long long)clock64() - c0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) {
                        __hip_atomic_store(&H.ready[h], j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else if (offload && n_tiles > 0) {
                    drain_half((n_tiles - 1) & 1, h);
                }
                if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                    dbg_busy += (long long)clock64() - w0;
                }
                job[h] = j + 1;
                served = true;
            }
            if (!served) {
                const long long w0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
                helper_idle(idle_spin, (cfg.dbg & 8388608) != 0, 1);
                if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                    dbg_wait += (long long)clock64() - w0;
                }
            }
        }
        it = n_tiles;
    } else
    for (t0 = 0; t0 < n; t0 += TW, it++) {
        const long long dbg_t0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
        const int tn = (int)((n - t0) < TW ? (n - t0) : TW);
        const bool more = (t0 + TW) < n;
        // Handler mode has no barrier per tile: the staging wave prepares tile it + 1 (and drains tile it - 1's queue) once BOTH
        // recurrence waves have finished tile it - 1, a recurrence wave enters tile it once that preparation is published - so
        // the two recurrence waves may be up to a tile apart and a tile that is slow for one of them is not waited out by the
        // other (with the barrier each waited ~10 k of 62 k cycles per tile for its sister).
        long long dbg_spin = 0;
        if (HM) {
            const long long w0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
            while (__hip_atomic_load(&H.ready[rw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= it
                   || (fuse_mf && __hip_atomic_load(&H.fready[rw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= it)) {
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                dbg_spin = (long long)clock64() - w0;
            }
        }
        if (loader || winprep) { // handler mode: one wave does both, one after the other
            if (loader) {
                if (more) {
                    stage(t0 + TW, (it + 1) % 3);
                }
                if (offload && it > 0 && !(cfg.dbg & 512)) {
                    drain((it - 1) & 1);
                }
            }
            if (winprep) {
                if (it >= 1 && !(cfg.dbg & 256)) {
                    compute_sfx(it & 1, it & 1);
                }
            }
        } else {
            const int base = TW + (it % 3) * TW;
            auto rd = [&](const float* row, int j) { return row[base + j]; };
            // tile-relative sample index from which the matched filter's output is usable (INT_MIN: filter off or warm)
            auto cold_limit = [&]() {
                if (!s.filter_on) {
                    return -2147483647;
                }
                const long long d = (long long)(NT - 1) - (abs0 + t0 - s.filt_start);
                return d <= -2147483647LL ? -2147483647 : (d > 1000000LL ? 1000000 : (int)d);
            };
            cold_until = cold_limit();
            if (it >= 1) { // the checkpoint moves to the start of the previous tile
                pp1 = pc1, pp2 = pc2, pp3 = pc3, pp4 = pc4;
                npp = npc;
                sbuf_sel = (it - 1) & 1;
            }
            pc1 = inf, pc2 = inf, pc3 = -inf, pc4 = -inf;
            npc = 0;
            tk = 0;
            itq = it & 1;
            o_tile = o;
            if (live && offload) {
                L.qo[it & 1][ln] = o;
            }
            int guard = 0;
            bool gblocked = false; // generic path: this lane's symbol waits for the next tile
            // lean trip operands fetched one trip ahead: the five window samples of this lane's next symbol and the suffix
            // summary its window push will ask for (valid only from one lean trip to the next inside a tile)
            bool respin = false; // handler mode: the last pass only waited for an answer (no trip was spent)
            int blk_o = -1;      // bulk hunting pass: output index at which it left this lane's next symbol to the standard trip
            while (true) {
                const long long dbg_p0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
                long long dbg_p1 = 0;
                if (HM && __any(hwait)) {
                    if (hwait
                        && __hip_atomic_load(&H.rsp_seq[ln], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == hseq) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                        const int ext = H.rsp_ext[ln];
                        hwait = false;
                        if ((cfg.dbg & 65536) && !(cfg.dbg & (67108864 | 134217728))) {
                            s.dbg_wait += (long long)clock64();
                        }
#if DDN_RX_CYCLES
                        if (cfg.dbg & 67108864) { // how long the answer lay unread
                            s.dbg_wait += (long long)((int)clock64() - H.rsp_t[ln]);
                        }
                        if (cfg.dbg & 134217728) { // how long the request lay unserved
                            s.dbg_wait += (long long)H.rsp_pick[ln];
                        }
#endif
                        if (ext > 0) { // the handler reads on
                            s.lock_left = ext;
                            s.hn = ext;
                            s.hphase = H.rsp_more[ln] ? 2 : 0;
                        } else {       // it has returned
                            s.hphase = 0;
                            frame_end();
                        }
                    }
                    // With eight lanes per wave the wave waits with a waiting lane: the lanes share their trips, and one that falls
                    // behind has to make up its symbols in trips of its own afterwards (measured: letting the others run ahead costs
                    // about three times the decision's latency, standing still costs it once).  With four lanes per wave the others
                    // go on (measured 9.47 against 9.66 ms): fewer lanes share the trips a straggler adds.
                    if (__any(hwait) && (LPR > 4) != ((cfg.dbg & 262144) != 0)) {
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                }
                const bool alive = live & !hwait;
                // ---- bulk hunting pass ---------------------------------------------------------------------------------------
                // A lane that hunts with its thresholds parked (no sync: max / min / centre stand still) runs a recurrence with
                // three words of state - where the next symbol starts, the latched crossing, the sign history - so its symbols
                // up to the end of the tile are found at once by the whole wavefront instead of one standard trip each (which the
                // in-frame lanes of the wave would sit through at 2.7 times the cost of a lean trip): the crossing test of every
                // staged sample as a 192-bit mask (three ballots), the chain symbol start -> first crossing inside the symbol
                // -> one-sample slip of the next start as scalar bit operations, then lane j takes symbol j: five-sample mean,
                // sign, its 24-symbol history word and the sync compare.  The pass stops in front of the symbol that completes a
                // sync pattern (the standard trip takes that one: warm start, lock) and at the eighth symbol of a hunt (the
                // commit of that symbol moves the crossing limits to the parked max / min).
                // hunt_wait: such a lane has nothing left that fits this tile - it waits for the next one like a lean lane does.
                bool hunt_wait = false;
                if (bulk_ok) {
                    const int jt = s.jitter;
                    const int i0n = (jt > 0 && jt <= (whole - 1) / 2) ? -1 : ((jt > (whole - 1) / 2 && jt < whole) ? 1 : 0);
                    const bool be = alive & (s.have_sync == 0) & (s.in_symbol == 0) & (s.need_reset == 0) & (sp >= cold_until)
                                    & (s.hunt_pos + 20 < 1800) & (blk_o != o) & ((size_t)(o + 20) < max_sym);
                    const bool fitsn = sp + whole - i0n <= tn;
                    hunt_wait = be & !fitsn & more;
                    unsigned long long bm = __ballot(be & fitsn);
                    // ---- (round 5) two / four channels per recurrence wave: every hunting lane's pass at once ------------------------
                    // A channel of such a wave has a ROW of 64 / LPR >= 16 lanes and a pass is at most 15 symbols, so every owner's
                    // pass fits its own row: the owner's words are broadcast over its row, lane l of a row is symbol l of that row's
                    // owner (mean, sign, history word, sync compare, stores, extrema), the owner lanes read their rows' results.  The
                    // slip chain: one owner - the crossing mask by three wave-wide ballots and the chain on scalars (as before);
                    // several owners (1.4 per pass on the bench traffic, 2.4 on voice calls whose frames end together) - the mask
                    // 64 / LPR samples per owner and ballot, the chain per row on the vector unit (its words are uniform inside a row).
                    if constexpr (LPR == 2 || LPR == 4) {
                    if (__builtin_expect(bm != 0, 0)) {
                        const long long bt0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
                        constexpr int OW = 64 / LPR;
                        constexpr int CAP = OW < 17 ? OW - 1 : 16; // (lane `m` of the row holds the start after the pass)
                        constexpr unsigned long long GM = OW == 32 ? 0xFFFFFFFFull : 0xFFFFull;
                        const int g = lane / OW, l = lane % OW;
                        const bool gact = ((bm >> g) & 1ull) != 0;
                        const int cln = rw * LPR + g;
                        const int sp0 = __shfl(sp, g);
                        const float cen_o = __shfl(s.center, g), ls_o = __shfl(s.lastsample, g);
                        const float* pr = (__shfl(s.filter_on, g) ? &L.flt[cln][0] : &L.raw[cln][0]) + base;
                        int m = 0, mypk = 0; // mypk: {start + 256, slip code, latch on entry + 1} of this lane's symbol
                        if ((bm & (bm - 1)) == 0 || (cfg.dbg & 4096)) {
                            // one owner at a time: wave-wide masks, scalar chain, the symbols' words written into the owner's row
                            unsigned long long b2 = bm;
                            while (b2) {
                                const int ow = __ffsll((long long)b2) - 1;
                                b2 &= b2 - 1;
                                const int sp0s = __builtin_amdgcn_readlane(sp, ow), c0s = __builtin_amdgcn_readlane(s.hist_count, ow);
                                const float cen_s = __shfl(s.center, ow), ls_s = __shfl(s.lastsample, ow);
                                float hl = __shfl(s.maxref * 1.25f, ow), ll = __shfl(s.minref * 1.25f, ow);
                                const float hl_n = __shfl(s.max * 1.25f, ow), ll_n = __shfl(s.min * 1.25f, ow);
                                const float* prs = (__builtin_amdgcn_readlane(s.filter_on, ow) ? &L.flt[rw * LPR + ow][0] : &L.raw[rw * LPR + ow][0]) + base;
                                int q = sp0s, sm = 0, jit = __builtin_amdgcn_readlane(s.jitter, ow);
                                int lim = c0s < 8 ? 8 - c0s : CAP;
                                for (int ph = 0; ph < 2; ph++) {
                                    unsigned long long cm[3];
                                    q = __builtin_amdgcn_readfirstlane(q);
                                    sm = __builtin_amdgcn_readfirstlane(sm);
                                    jit = __builtin_amdgcn_readfirstlane(jit);
#pragma unroll
                                    for (int r = 0; r < 3; r++) {
                                        cm[r] = 0ull;
                                        if (q + 64 * r < tn) {
                                            const int a = q + lane + 64 * r;
                                            bool hit = false;
                                            if (a < tn) {
                                                const float x = prs[a];
                                                const float xp = (a == sp0s) ? ls_s : prs[a - 1];
                                                hit = (x > cen_s) ? (!(x > hl) && xp < cen_s) : (!(x < ll) && xp > cen_s);
                                            }
                                            cm[r] = __ballot(hit);
                                        }
                                    }
                                    bool full = false;
                                    auto uni64 = [](unsigned long long v) {
                                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                                        return ((unsigned long long)hi << 32) | lo;
                                    };
                                    cm[0] = uni64(cm[0]);
                                    cm[1] = uni64(cm[1]);
                                    cm[2] = uni64(cm[2]);
                                    while (sm < lim) {
                                        const int c = (int)((i0tab >> (2 * (jit + 1))) & 3u), i0 = c - 1;
                                        const int cnt = whole - i0;
                                        if (q + cnt > tn) {
                                            full = true;
                                            break;
                                        }
                                        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mypk) : "s"((q + 256) | (c << 10) | ((jit + 1) << 12)), "s"(ow * OW + sm) : "m0");
                                        const int k0 = i0 < 0 ? 1 : 0; // a crossing at symbol index -1 latches nothing
                                        const uint32_t wv = (uint32_t)(cm[0] >> k0) & ((1u << (cnt - k0)) - 1u);
                                        jit = wv ? i0 + k0 + (__ffs((int)wv) - 1) : -1;
                                        cm[0] = (cm[0] >> cnt) | (cm[1] << (64 - cnt));
                                        cm[1] = (cm[1] >> cnt) | (cm[2] << (64 - cnt));
                                        cm[2] >>= cnt;
                                        q += cnt;
                                        sm++;
                                    }
                                    if (full || lim >= CAP) {
                                        break;
                                    }
                                    lim = CAP;
                                    hl = hl_n;
                                    ll = ll_n;
                                }
                                // where the symbol after the pass starts, and the latch it starts with
                                asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mypk) : "s"((q + 256) | ((jit + 1) << 12)), "s"(ow * OW + sm) : "m0");
                                m = g == ow ? sm : m;
                            }
                        } else {
                            int jit = __shfl(s.jitter, g), q = sp0;
                            float hl = __shfl(s.maxref * 1.25f, g), ll = __shfl(s.minref * 1.25f, g);
                            int lim = __shfl(s.hist_count, g);
                            lim = lim < 8 ? 8 - lim : CAP;
                            bool gdone = !gact; // this row's chain has ended
                            for (int ph = 0; ph < 2; ph++) {
                                unsigned long long cm0 = 0ull, cm1 = 0ull, cm2 = 0ull;
                                constexpr int PW = 64 / OW; // ballots per 64-bit word
#pragma unroll 1
                                for (int wd = 0; wd < 3; wd++) { // (one word at a time: PW loads in flight, not 192 / OW of them)
                                    unsigned long long word = 0ull;
#pragma unroll
                                    for (int r = 0; r < PW; r++) {
                                        const int a = q + l + OW * (wd * PW + r);
                                        bool hit = false;
                                        if (!gdone && a < tn) {
                                            const float x = pr[a];
                                            const float xp = (a == sp0) ? ls_o : pr[a - 1];
                                            hit = (x > cen_o) ? (!(x > hl) && xp < cen_o) : (!(x < ll) && xp > cen_o);
                                        }
                                        word |= ((__ballot(hit) >> (g * OW)) & GM) << (OW * r);
                                    }
                                    cm0 = wd == 0 ? word : cm0;
                                    cm1 = wd == 1 ? word : cm1;
                                    cm2 = wd == 2 ? word : cm2;
                                }
                                bool full = false;
                                while (true) {
                                    const bool go = !gdone && !full && m < lim;
                                    if (!__any(go)) {
                                        break;
                                    }
                                    if (go) {
                                        const int c = (int)((i0tab >> (2 * (jit + 1))) & 3u), i0 = c - 1;
                                        const int cnt = whole - i0;
                                        if (q + cnt > tn) {
                                            full = true;
                                        } else {
                                            if (l == m) {
                                                mypk = (q + 256) | (c << 10) | ((jit + 1) << 12);
                                            }
                                            const int k0 = i0 < 0 ? 1 : 0;
                                            const uint32_t wv = (uint32_t)(cm0 >> k0) & ((1u << (cnt - k0)) - 1u);
                                            jit = wv ? i0 + k0 + (__ffs((int)wv) - 1) : -1;
                                            cm0 = (cm0 >> cnt) | (cm1 << (64 - cnt));
                                            cm1 = (cm1 >> cnt) | (cm2 << (64 - cnt));
                                            cm2 >>= cnt;
                                            q += cnt;
                                            m++;
                                        }
                                    }
                                }
                                if (ph == 1) {
                                    break;
                                }
                                const float hl_n = __shfl(s.max * 1.25f, g), ll_n = __shfl(s.min * 1.25f, g);
                                if (!gdone) {
                                    if (full || lim >= CAP) {
                                        gdone = true;
                                    } else { // the eighth symbol of the hunt moved the crossing limits: the mask is taken again
                                        lim = CAP;
                                        hl = hl_n;
                                        ll = ll_n;
                                    }
                                }
                                if (!__any(!gdone)) {
                                    break;
                                }
                            }
                            if (l == m) {
                                mypk = (q + 256) | ((jit + 1) << 12);
                            }
                        }
                        // lane l of row g: symbol l of owner g
                        const int myq = (mypk & 1023) - 256, myi0 = ((mypk >> 10) & 3) - 1, myjin = ((mypk >> 12) & 15) - 1;
                        const int c0 = __shfl(s.hist_count, g);
                        const uint32_t h0 = (uint32_t)__shfl((int)s.hist_bits, g);
                        float sym = 0.0f;
                        if (gact && l < m) {
                            const float* pw = pr + myq + ((whole - 1) / 2 - 2 - myi0);
                            float acc = 0.0f;
#pragma unroll
                            for (int w = 0; w < 5; w++) {
                                acc += __builtin_amdgcn_fmed3f(pw[w], -inf, inf);
                            }
                            sym = acc / 5.0f;
                        }
                        const uint32_t sw = (uint32_t)((__ballot(gact && l < m && sym > 0.0f) >> (g * OW)) & GM);
                        const int lj = l < 31 ? l : 31;
                        const uint32_t hj = ((h0 << (lj + 1)) | __brev(sw << (31 - lj))) & 0xFFFFFFu;
                        const bool syn = gact && l < m && (c0 + l + 1 >= 24) && (hj == kSyncBits || hj == (~kSyncBits & 0xFFFFFFu));
                        const uint32_t smg = (uint32_t)((__ballot(syn) >> (g * OW)) & GM);
                        if (smg) {
                            m = __ffs((int)smg) - 1;
                        }
                        const int o_o = __shfl(o, g);
                        const int sh_o = __shfl(s.shead, g), li_o = __shfl(s.lidx, g), si_o = __shfl(s.sidx, g);
                        if (gact && l < m) { // symbol history, level window, extrema window, record
                            int k = sh_o + l;
                            L.sh[k >= 24 ? k - 24 : k][cln] = sym;
                            k = li_o + l;
                            L.lb[k >= 24 ? k - 24 : k][cln] = sym;
                            L.sb[(si_o + l) & (SS - 1)][cln] = sym;
                            const size_t oo = (size_t)(o_o + l);
                            if (oo < max_sym) {
                                const size_t cho = (size_t)(ch0 + cln);
                                store_record(rec + (cho * max_sym + oo) * 10, flags + cho * max_sym + oo, sym, sym > 0.0f ? 1 : 3, 0, 0, 0, 0);
                            }
                        }
                        float a1 = (gact && l < m) ? sym : inf, a2 = inf, b1 = (gact && l < m) ? sym : -inf, b2 = -inf;
                        // (a pass is at most 16 symbols: lanes 0 .. 15 of the row = one DPP row; quad swaps, then the row rotated by 4 and
                        // by 8 - every step joins disjoint sets of lanes, so no value is counted twice)
#define DDN_ROW_STEP(ctl)                                                                                                                   \
    do {                                                                                                                                    \
        const float o1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), ctl, 0xF, 0xF, true));                           \
        const float o2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), ctl, 0xF, 0xF, true));                           \
        const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b1), ctl, 0xF, 0xF, true));                           \
        const float p2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b2), ctl, 0xF, 0xF, true));                           \
        two_min_insert(o1, a1, a2);                                                                                                         \
        two_min_insert(o2, a1, a2);                                                                                                         \
        two_max_insert(p1, b1, b2);                                                                                                         \
        two_max_insert(p2, b1, b2);                                                                                                         \
    } while (0)
                        DDN_ROW_STEP(0xB1);  // quad_perm [1,0,3,2]
                        DDN_ROW_STEP(0x4E);  // quad_perm [2,3,0,1]
                        DDN_ROW_STEP(0x124); // row_ror:4
                        DDN_ROW_STEP(0x128); // row_ror:8
#undef DDN_ROW_STEP
                        // the owner lanes (lane = channel of this wave) read their row's results
                        const int orow = (lane < LPR ? lane : 0) * OW;
                        const int m_own = __shfl(m, orow);
                        const int src1 = orow + (m_own > 0 ? m_own - 1 : 0);
                        const int qf = __shfl(myq, orow + m_own), jf = __shfl(myjin, orow + m_own);
                        const int il = __shfl(myi0, src1);
                        const uint32_t hf = (uint32_t)__shfl((int)hj, src1);
                        a1 = __shfl(a1, orow), a2 = __shfl(a2, orow), b1 = __shfl(b1, orow), b2 = __shfl(b2, orow);
                        if (lane < LPR && (be & fitsn)) {
                            if (m_own == 0) { // the next symbol completes a sync: the standard trip's
                                blk_o = o;
                            } else {
                                const float* prow = (s.filter_on ? &L.flt[ln][0] : &L.raw[ln][0]) + base;
                                const float lsf = prow[qf - 1];
                                two_min_insert(a1, pc1, pc2);
                                two_min_insert(a2, pc1, pc2);
                                two_max_insert(b1, pc3, pc4);
                                two_max_insert(b2, pc3, pc4);
                                npc += m_own;
                                wabs = fmaxf(wabs, fmaxf(fabsf(a1), fabsf(b1))); // the pass's smallest and largest symbol
                                s.shead = (s.shead + m_own) % 24;
                                s.scount = s.scount + m_own < 24 ? s.scount + m_own : 24;
                                s.lidx = (s.lidx + m_own) % 24;
                                s.level_count = s.level_count + m_own < 24 ? s.level_count + m_own : 24;
                                s.sidx = (s.sidx + m_own) & (SS - 1);
                                s.hist_bits = hf;
                                s.hist_count = s.hist_count + m_own < 24 ? s.hist_count + m_own : 24;
                                if (s.hist_count >= 8) {
                                    s.maxref = s.max;
                                    s.minref = s.min;
                                }
                                s.hunt_pos += m_own;
                                s.span = whole;
                                s.centre = (whole - 1) / 2;
                                s.i = il;
                                s.sum = 0.0f;
                                s.count = 0;
                                s.in_symbol = 0;
                                s.jitter = jf;
                                s.lastsample = lsf;
                                sp = qf;
                                o += m_own;
                            }
                        }
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) { // the pass is no trip: its cycles are kept apart
                            const long long now = (long long)clock64();
                            dbg_sec[6] += now - bt0;
                            dbg_sec[7]++;
                            dbg_prev += now - bt0;
                        }
                        continue;
                    }
                    } else
                    if (__builtin_expect(bm != 0, 0)) {
                        const long long bt0 = (DDN_RX_CYCLES && (cfg.dbg & 8192)) ? (long long)clock64() : 0;
                        while (bm) {
                            const int ow = __ffsll((long long)bm) - 1; // owner lane of this pass (wave-uniform)
                            bm &= bm - 1;
                            if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                                dbg_bt = (long long)clock64();
                                dbg_blk[7]++;
                            }
                            const int cln = rw * LPR + ow;
                            const int sp0 = __builtin_amdgcn_readlane(sp, ow);
                            const int c0 = __builtin_amdgcn_readlane(s.hist_count, ow);
                            const int flt_o = __builtin_amdgcn_readlane(s.filter_on, ow);
                            int jit = __builtin_amdgcn_readlane(s.jitter, ow);
                            const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)s.hist_bits, ow);
                            const int o_o = __builtin_amdgcn_readlane(o, ow);
                            const float cen_o = __shfl(s.center, ow), hl_o = __shfl(s.maxref * 1.25f, ow);
                            const float ll_o = __shfl(s.minref * 1.25f, ow), ls_o = __shfl(s.lastsample, ow);
                            const float* pr = (flt_o ? &L.flt[cln][0] : &L.raw[cln][0]) + base;
                            // the chain: symbol start -> first crossing inside the symbol -> slip of the next start, on scalars.
                            // cm[] = crossing test of the samples q .. tn - 1 (bit a - q), shifted along with q.  The eighth
                            // symbol of a hunt moves the crossing limits to the parked max / min: the mask is taken again there.
                            float hl = hl_o, ll = ll_o;
                            const float hl_n = __shfl(s.max * 1.25f, ow), ll_n = __shfl(s.min * 1.25f, ow);
                            int q = sp0, m = 0, mypk = 0;
                            int lim = c0 < 8 ? 8 - c0 : 16;
                            DBG_BLK(0);
                            for (int ph = 0; ph < 2; ph++) {
                                unsigned long long cm[3];
                                // (the chain's words are wave-uniform; said so explicitly - left to itself the compiler keeps them in
                                // vector registers and runs the loop under an execution mask: ~680 cycles per symbol against scalar code)
                                q = __builtin_amdgcn_readfirstlane(q);
                                m = __builtin_amdgcn_readfirstlane(m);
                                jit = __builtin_amdgcn_readfirstlane(jit);
#pragma unroll
                                for (int r = 0; r < 3; r++) {
                                    cm[r] = 0ull;
                                    if (q + 64 * r < tn) {
                                        const int a = q + lane + 64 * r;
                                        bool hit = false;
                                        if (a < tn) {
                                            const float x = pr[a];
                                            const float xp = (a == sp0) ? ls_o : pr[a - 1];
                                            hit = (x > cen_o) ? (!(x > hl) && xp < cen_o) : (!(x < ll) && xp > cen_o);
                                        }
                                        cm[r] = __ballot(hit);
                                    }
                                }
                                bool full = false;
                                auto uni64 = [](unsigned long long v) {
                                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
                                    return ((unsigned long long)hi << 32) | lo;
                                };
                                cm[0] = uni64(cm[0]);
                                cm[1] = uni64(cm[1]);
                                cm[2] = uni64(cm[2]);
                                while (m < lim) {
                                    // slip of this symbol's start by the latched crossing: -1 / 0 / +1 as a two-bit code from a table
                                    // over the latch value (i0tab, one entry per jit + 1)
                                    const int c = (int)((i0tab >> (2 * (jit + 1))) & 3u), i0 = c - 1;
                                    const int cnt = whole - i0;
                                    if (q + cnt > tn) {
                                        full = true;
                                        break;
                                    }
                                    // lane m's symbol: {start, slip code, latch on entry} as one word written into that lane
                                    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mypk) : "s"((q + 256) | (c << 10) | ((jit + 1) << 12)), "s"(m) : "m0");
                                    const int k0 = i0 < 0 ? 1 : 0; // a crossing at symbol index -1 latches nothing
                                    const uint32_t wv = (uint32_t)(cm[0] >> k0) & ((1u << (cnt - k0)) - 1u);
                                    jit = wv ? i0 + k0 + (__ffs((int)wv) - 1) : -1;
                                    cm[0] = (cm[0] >> cnt) | (cm[1] << (64 - cnt));
                                    cm[1] = (cm[1] >> cnt) | (cm[2] << (64 - cnt));
                                    cm[2] >>= cnt;
                                    q += cnt;
                                    m++;
                                }
                                if (full || lim >= 16) {
                                    break;
                                }
                                lim = 16;
                                hl = hl_n;
                                ll = ll_n;
                            }
                            DBG_BLK(1);
                            // where the symbol after the pass starts, and the latch it starts with
                            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(mypk) : "s"((q + 256) | ((jit + 1) << 12)), "s"(m) : "m0");
                            const int myq = (mypk & 1023) - 256, myi0 = ((mypk >> 10) & 3) - 1, myjin = ((mypk >> 12) & 15) - 1; // (starts may lie in the previous tile)
                            float sym = 0.0f;
                            if (lane < m) {
                                const float* pw = pr + myq + ((whole - 1) / 2 - 2 - myi0);
                                float acc = 0.0f;
#pragma unroll
                                for (int w = 0; w < 5; w++) {
                                    acc += __builtin_amdgcn_fmed3f(pw[w], -inf, inf);
                                }
                                sym = acc / 5.0f;
                            }
                            const uint32_t sw = (uint32_t)__ballot(lane < m && sym > 0.0f);
                            const int lj = lane < 31 ? lane : 31;
                            const uint32_t hj = ((h0 << (lj + 1)) | __brev(sw << (31 - lj))) & 0xFFFFFFu;
                            const bool syn = lane < m && (c0 + lane + 1 >= 24) && (hj == kSyncBits || hj == (~kSyncBits & 0xFFFFFFu));
                            const unsigned long long sm = __ballot(syn);
                            if (sm) {
                                m = __ffsll((long long)sm) - 1;
                            }
                            DBG_BLK(2);
                            if (m == 0) { // the next symbol completes a sync: the standard trip's
                                if (lane == ow) {
                                    blk_o = o;
                                }
                                continue;
                            }
                            const int qf = __builtin_amdgcn_readlane(myq, m), jf = __builtin_amdgcn_readlane(myjin, m);
                            const int il = __builtin_amdgcn_readlane(myi0, m - 1);
                            const uint32_t hf = (uint32_t)__builtin_amdgcn_readlane((int)hj, m - 1);
                            const int sh_o = __builtin_amdgcn_readlane(s.shead, ow), li_o = __builtin_amdgcn_readlane(s.lidx, ow);
                            const int si_o = __builtin_amdgcn_readlane(s.sidx, ow);
                            if (lane < m) { // symbol history, level window, extrema window, record
                                int k = sh_o + lane;
                                L.sh[k >= 24 ? k - 24 : k][cln] = sym;
                                k = li_o + lane;
                                L.lb[k >= 24 ? k - 24 : k][cln] = sym;
                                L.sb[(si_o + lane) & (SS - 1)][cln] = sym;
                                const size_t oo = (size_t)(o_o + lane);
                                if (oo < max_sym) {
                                    const size_t cho = (size_t)(ch0 + cln);
                                    store_record(rec + (cho * max_sym + oo) * 10, flags + cho * max_sym + oo, sym, sym > 0.0f ? 1 : 3, 0, 0, 0, 0);
                                }
                            }
                            DBG_BLK(3);
                            float a1 = lane < m ? sym : inf, a2 = inf, b1 = lane < m ? sym : -inf, b2 = -inf;
#pragma unroll
                            for (int d = 1; d < 16; d <<= 1) {
                                const float o1 = __shfl_xor(a1, d), o2 = __shfl_xor(a2, d), p1 = __shfl_xor(b1, d), p2 = __shfl_xor(b2, d);
                                two_min_insert(o1, a1, a2);
                                two_min_insert(o2, a1, a2);
                                two_max_insert(p1, b1, b2);
                                two_max_insert(p2, b1, b2);
                            }
                            a1 = __shfl(a1, 0), a2 = __shfl(a2, 0), b1 = __shfl(b1, 0), b2 = __shfl(b2, 0);
                            const float lsf = pr[qf - 1];
                            DBG_BLK(4);
                            if (lane == ow) {
                                two_min_insert(a1, pc1, pc2);
                                two_min_insert(a2, pc1, pc2);
                                two_max_insert(b1, pc3, pc4);
                                two_max_insert(b2, pc3, pc4);
                                npc += m;
                                wabs = fmaxf(wabs, fmaxf(fabsf(a1), fabsf(b1))); // the pass's smallest and largest symbol
                                s.shead = (s.shead + m) % 24;
                                s.scount = s.scount + m < 24 ? s.scount + m : 24;
                                s.lidx = (s.lidx + m) % 24;
                                s.level_count = s.level_count + m < 24 ? s.level_count + m : 24;
                                s.sidx = (s.sidx + m) & (SS - 1);
                                s.hist_bits = hf;
                                s.hist_count = c0 + m < 24 ? c0 + m : 24;
                                if (s.hist_count >= 8) {
                                    s.maxref = s.max;
                                    s.minref = s.min;
                                }
                                s.hunt_pos += m;
                                s.span = whole;
                                s.centre = (whole - 1) / 2;
                                s.i = il;
                                s.sum = 0.0f;
                                s.count = 0;
                                s.in_symbol = 0;
                                s.jitter = jf;
                                s.lastsample = lsf;
                                sp = qf;
                                o += m;
                            }
                            DBG_BLK(5);
                            if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                                dbg_blk[6] += m;
                            }
                        }
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) { // the pass is no trip: its cycles are kept apart
                            const long long now = (long long)clock64();
                            dbg_sec[6] += now - bt0;
                            dbg_sec[7]++;
                            dbg_prev += now - bt0;
                        }
                        continue;
                    }
                }
                if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                    dbg_p1 = (long long)clock64();
                }
                // hand the previous trip's symbols (one per lane at most) to wave 1: one 16-byte LDS write at a wave-uniform slot
                if (!respin) {
                    if (offload && tk > 0 && tk <= QTW && lane < LPR) {
                        *reinterpret_cast<float4*>(&L.q[itq][tk - 1][ln][0]) = qv;
                    }
                    qv.w = __int_as_float(-1);
                    tk++;
                }
                respin = false;
                if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                    const long long now = (long long)clock64();
                    if (dbg_kind >= 0) {
                        dbg_cyc[dbg_kind] += now - dbg_prev;
                        dbg_n[dbg_kind]++;
                    }
                    dbg_prev = now;
                    dbg_kind = 1;
                }
                // ---- standard trip -------------------------------------------------------------------------------------
                // Symbols of the ordinary length that start fresh (or were deferred whole to this tile) with the matched filter
                // warm: A = in frame (five-sample clipped mean, in-frame commit; crossing search only while the latch is open),
                // B = hunting (one-sample timing slip, crossing search, plain mean, hunting commit).  When every live lane is in
                // one of the two states - or waits in one of them for the next tile - the trip is this straight-line block and
                // none of the general code below is on the wave's instruction stream.  Everything else (filter cold start,
                // partly consumed symbols at a call's edges, odd spans, degenerate thresholds) takes the general trip.
                // ---- lean trip -----------------------------------------------------------------------------------------
                // Every live lane sits inside a frame with its crossing latched, a whole fresh symbol staged, the matched filter
                // warm and more than one symbol of the lock left (or waits in that state for the next tile): nothing per-sample
                // can change state and nothing reads the last sample or the 24-symbol history (the history is only read by the
                // warm start of a sync, 24 hunting symbols after the frame), so the trip is: clipped five-sample mean, window
                // push, extrema rings, thresholds, queue entry.  Its LDS operands were fetched by the previous lean trip, which
                // takes the two LDS round trips off the recurrence.
                bool all_lean_wait = false;
                if (lean_ok && tk <= QTW) {
                    // (bitwise on purpose: one compare each, no short-circuit branches on the recurrence wave)
                    // (round 5) a phase that ends in a handler's decision takes its last symbol inside the run too (the request is posted
                    // after the run, below); any other lock keeps its last symbol for the standard trip (frame_end, the last sample)
                    const bool hpl = HM && (s.hphase != 0) && (s.hn >= 2) && !(cfg.dbg & 33554432);
                    bool lean_state = alive & (s.have_sync != 0) & (s.jitter >= 0) & (s.lock_left > (hpl ? 0 : 1)) & (sp >= cold_until)
                                      & ((s.in_symbol == 0) | ((s.i == 0) & (s.count == 0))) & (s.min < s.max)
                                      & (s.since_fill < MS) & (npp + npc < WMW);
                    if (__any(lean_state)) {
                        // A lean run clips with the median of three and never looks at min < max again, so a lane only enters one when
                        // that order provably holds to the run's end (QTW trips at most).  With D = max_sum - min_sum, G = the rings'
                        // fill gap (every ring slot a run replaces still holds the fill value: since_fill < MS), A = the larger sum's
                        // magnitude and S a bound on every window value, threshold and fill value: a push changes D by (hi - lo) - G
                        // >= -max(G, 0) and the sums by at most 6 S, so D stays above D - 40 max(G, 0) and max / 1024 and min / 1024
                        // round to different binary32 values while that exceeds 2^-22 of the sums.  (Real traffic: D ~ 4e7, the
                        // right side ~ 50.)  Anything else takes the standard trip, which tests min < max symbol by symbol.
                        const double d0 = s.max_sum - s.min_sum;
                        const double gap = fill_max_d - fill_min_d;
                        const double gp = __builtin_fmax(gap, 0.0);
                        const double am = __builtin_fmax(__builtin_fabs(s.max_sum), __builtin_fabs(s.min_sum));
                        const float sf = fmaxf(fmaxf(wabs, fmaxf(fabsf(s.min), fabsf(s.max))), fmaxf(fabsf(s.fill_min), fabsf(s.fill_max)));
                        lean_state &= (d0 - 40.0 * gp) > (am * 0x1p-19 + (double)sf * 0x1p-12);
                    }
                    const bool lean = lean_state & (sp + whole <= tn);
                    const bool lean_wait = lean_state & !(sp + whole <= tn) & more;
                    const bool all_lean = !__any(alive & !(lean | lean_wait | hunt_wait));
                    if (all_lean && __any(lean)) {
                        dbg_kind = 2;
                        const int k0 = (whole - 1) / 2 - 2;
                        // A lean run: K trips back to back, K known before the first one - the fewest symbols any lean lane still has
                        // in this tile, in its lock, before a ring index wraps (window 128, extrema 1024, history HN), before the
                        // extrema rings' fill runs out or the queue is full.  Nothing inside the run depends on a lane's data any more
                        // (no test, no select): a scalar trip counter, straight-line trips, the state words that only count brought
                        // up to date once at the end.  The lanes that wait for the next tile, or sit out for a handler, are masked
                        // off for the whole run.  (Tried and dropped: going on in phases with the lanes that have symbols left, and
                        // polling a waiting lane's mailbox once per trip instead of capping the run - 1.5 runs per tile instead of
                        // 2.7, the same kernel time, 30 more registers.)
                        int kl = 0x7fffffff;
                        const bool hp = HM && (s.hphase != 0);
                        if (lean) {
                            kl = ((tn - sp) * (65536 / whole + 1)) >> 16; // == (tn - sp) / whole for these small numbers
                            kl = min(kl, s.lock_left - (hpl ? 0 : 1));
                            kl = min(kl, MS - s.since_fill);
                            kl = min(kl, WMW - (npp + npc));
                            kl = min(kl, SS - s.sidx);
                            kl = min(kl, MS - s.midx);
                            kl = hp ? min(kl, ddn_p25h::HN - s.hw) : kl;
                        }
                        int K = 0x7fffffff;
                        {
                            const unsigned long long lm = __ballot(lean);
                            for (int c = 0; c < LPR; c++) {
                                if ((lm >> c) & 1) {
                                    K = min(K, __builtin_amdgcn_readlane(kl, c));
                                }
                            }
                        }
                        K = min(K, QTW - tk + 1);
                        if (HM && __any(hwait)) { // an answer may arrive: look again soon
                            K = min(K, 2);
                        }
                        if (cfg.dbg & 2097152) {
                            K = 1;
                        }
                        K = __builtin_amdgcn_readfirstlane(K); // (tk is uniform, but not provably so: keep the counter scalar)
                        long long dbg_l0 = 0;
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                            dbg_run[0]++;
                            dbg_run[1] += K;
                            dbg_run[5] += (HM && __any(hwait)) ? 1 : 0;
                            dbg_run[7]++;
                            dbg_l0 = (long long)clock64();
                            dbg_run[2] += dbg_p1 - dbg_p0;  // pass top: handler answers, bulk-pass test
                            dbg_run[3] += dbg_l0 - dbg_p1;  // queue hand-over, lean tests, entry proof, run length
                        }
                        if (lean) {
                            const float* pw = (s.filter_on ? frow : rrow) + base + sp + k0;
                            const float4* psp = reinterpret_cast<const float4*>(&L.sfx[sbuf_sel][npp + npc][ln][0]);
                            float* sbp = &L.sb[s.sidx][ln];
                            float4* qp = reinterpret_cast<float4*>(&L.q[itq][tk - 1][ln][0]);
                            float* hhp = &L.sb[0][ln];
                            int hstep = 0;
                            if (HM) {
                                hhp = hp ? &H.hh[s.hw][ln][0] : &H.hh_dummy[ln][0];
                                hstep = hp ? CPW * 3 : 0;
                            }
                            uint32_t rof = ro;
                            float mn = s.min, mx = s.max;
                            double smin = s.min_sum, smax = s.max_sum;
                            float x0 = pw[0], x1 = pw[1], x2 = pw[2], x3 = pw[3], x4 = pw[4];
                            float4 sf4 = *psp;
                            float sym = 0.0f;
                            int fw = (1 | (s.lastsync == 2 ? 4 : 0)) | ((o - o_tile) << 8);
                            for (int j = 0; j < K; j++) {
                                // min < max and finite samples: the reference's two-sided clip is the median of three (a zero's sign
                                // may differ, which a sum that starts at +0 cannot show)
                                float acc = 0.0f;
                                acc += v_med3(x0, mn, mx);
                                acc += v_med3(x1, mn, mx);
                                acc += v_med3(x2, mn, mx);
                                acc += v_med3(x3, mn, mx);
                                acc += v_med3(x4, mn, mx);
                                sym = div5_exact(acc);
                                // the window summaries that do not depend on this symbol: suffix summary + the previous tile's pushes
                                const float t1 = v_min2(sf4.x, pp1), t2 = v_min3(v_max2(sf4.x, pp1), sf4.y, pp2);
                                const float u1 = v_max2(sf4.z, pp3), u2 = v_max3(v_min2(sf4.z, pp3), sf4.w, pp4);
                                // the next trip's operands (a run's last trip fetches past what it may use: inside the arrays, unread)
                                pw += whole;
                                psp += CPW;
                                x0 = pw[0], x1 = pw[1], x2 = pw[2], x3 = pw[3], x4 = pw[4];
                                sf4 = *psp;
                                *sbp = sym;
                                sbp += CPW;
                                // this tile's pushes + the symbol (two smallest / two largest of a sorted pair and one value)
                                const float n2 = v_med3(pc1, pc2, sym), n4 = v_med3(pc3, pc4, sym);
                                pc1 = v_min2(pc1, sym);
                                pc3 = v_max2(pc3, sym);
                                pc2 = n2;
                                pc4 = n4;
                                const float m1 = v_min2(t1, pc1), m2 = v_min3(v_max2(t1, pc1), t2, pc2);
                                const float w1 = v_max2(u1, pc3), w2 = v_max3(v_min2(u1, pc3), u2, pc4);
                                const float lo = (m1 + m2) * 0.5f, hi = (w1 + w2) * 0.5f;
                                ring_at(minring, rof) = lo;
                                ring_at(maxring, rof) = hi;
                                rof += ro_step;
                                smin += (double)lo - fill_min_d; // every slot a run replaces still holds the fill value
                                smax += (double)hi - fill_max_d;
                                mn = (float)(smin / (double)MS);
                                mx = (float)(smax / (double)MS);
                                *qp = make_float4(sym, mx, mn, __int_as_float(fw));
                                qp += CPW;
                                fw += 256;
                                if (HM) {
                                    hhp[0] = sym;
                                    hhp[1] = mx;
                                    hhp[2] = mn;
                                    hhp += hstep;
                                }
                            }
                            // the words that only count
                            sp += K * whole;
                            s.in_symbol = 0;
                            npc += K;
                            s.since_fill += K;
                            s.sidx = (s.sidx + K) & (SS - 1);
                            s.midx += K;
                            ro = rof;
                            if (s.midx >= MS) {
                                s.midx = 0;
                                ro = ro_first;
                            }
                            s.lock_left -= K;
                            o += K;
                            if (hp) {
                                s.hw = (s.hw + K >= ddn_p25h::HN) ? 0 : s.hw + K;
                            }
                            s.min_sum = smin;
                            s.max_sum = smax;
                            s.min = mn;
                            s.max = mx;
                            wabs = fmaxf(wabs, fmaxf(fabsf(pc1), fabsf(pc3)));
                            // the thresholds that follow max / min: nothing inside a lean run reads them
                            s.center = (s.max + s.min) / 2.0f;
                            s.maxref = s.max * 0.80f;
                            s.minref = s.min * 0.80f;
                            qv = make_float4(sym, mx, mn, __int_as_float(fw - 256)); // the run's last entry, as the trip loop's top hands it over
                            if (HM && hpl && s.lock_left == 0) {
                                // The phase's last symbol was one of the run's.  What the standard trip does beyond a lean one: the
                                // symbol's last sample, clipped to max / min as they stood BEFORE this symbol (= the previous history
                                // entry of the phase; read by the first hunting symbol if the handler returns), and the request.
                                const int pv = s.hw - 2 + (s.hw < 2 ? ddn_p25h::HN : 0);
                                const float pmx = H.hh[pv][ln][1], pmn = H.hh[pv][ln][2];
                                const float xl = ((s.filter_on ? frow : rrow) + base)[sp - 1];
                                s.lastsample = xl > pmx ? pmx : (xl < pmn ? pmn : xl);
                                post_request(o - 1);
                            }
                        }
                        tk += K - 1;
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                            dbg_n[2] += K - 1;
                            dbg_run[6] += (long long)clock64() - dbg_l0;
                        }
                        // every lane of the run now waits for the next tile (the others already did) and no lane sits out for a
                        // handler: the tile is over - its last trip handed over as the trip loop's top would, no empty pass
                        const bool st = lean & (s.lock_left > 1) & (s.min < s.max);
                        const bool now_waits = st & !(sp + whole <= tn) & more;
                        if (!__any(lean & !now_waits) && !(HM && __any(hwait)) && !(cfg.dbg & 4194304)) {
                            if (tk <= QTW && lane < LPR) {
                                *reinterpret_cast<float4*>(&L.q[itq][tk - 1][ln][0]) = qv;
                            }
                            qv.w = __int_as_float(-1);
                            tk++;
                            break;
                        }
                        continue;
                    }
                    all_lean_wait = all_lean; // nobody can go on and nobody needs another kind of trip: the tile is over
                }
                bool all_std_wait = all_lean_wait;
                if (std_ok && !all_lean_wait) {
                    // (bitwise on purpose: one compare each, no short-circuit branches on the recurrence wave)
                    const bool warm = alive & (sp >= cold_until); // == !(filter_on && (abs0 + t0 + sp - filt_start) < NT - 1)
                    const bool hunting = s.have_sync == 0;
                    if (warm & hunting & (s.in_symbol == 0) & (sp < tn) & !hunt_wait) { // symbol start while hunting: slip by the latched crossing
                        if (s.need_reset) {
                            timing_reset();
                        }
                        s.span = whole;
                        s.centre = (whole - 1) / 2;
                        s.i = 0;
                        s.sum = 0.0f;
                        s.count = 0;
                        s.in_symbol = 1;
                        if (s.jitter >= 0) {
                            if (s.jitter > 0 && s.jitter <= s.centre) {
                                s.i = -1;
                            } else if (s.jitter > s.centre && s.jitter < whole) {
                                s.i = 1;
                            }
                            s.jitter = -1;
                        }
                    }
                    const bool in_a = warm & !hunting & ((s.in_symbol == 0) | ((s.i == 0) & (s.count == 0))) & (s.min < s.max);
                    const bool in_b = warm & hunting & (s.in_symbol != 0) & (s.i >= -1) & (s.i <= 1) & (s.count == 0);
                    const bool idle_b = warm & hunting & (s.in_symbol == 0); // tile used up exactly: nothing started
                    const int i0 = in_b ? s.i : 0;
                    const int cnt = whole - i0;
                    const bool fits = sp + cnt <= tn;
                    const bool can = (in_a | in_b) & fits;
                    const bool wt = (((in_a | in_b) & !fits) | idle_b) & more; // waits for the next tile (both stay staged)
                    const bool all_ok = !__any(alive & !(can | wt));
                    if (all_ok && __any(can)) {
                        dbg_kind = 0;
                        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
                            dbg_sec[0] += (long long)clock64() - dbg_prev; // trip top .. here (pause, queue hand-over, lean / std tests)
                            dbg_st = (long long)clock64();
                        }
                        const float* p = (s.filter_on ? frow : rrow) + base + sp;
                        int jit = s.jitter;
                        if (__any(can & (jit < 0))) {
                            // The crossing test of sample k reads x[k], x[k - 1] and thresholds that are fixed for the whole symbol,
                            // so the 64 / CPW lanes that share a channel's column (lane = channel + CPW * slot) each test the
                            // samples k = slot, slot + 64 / CPW, ... and the owner takes the lowest set bit of the ballots: the
                            // first crossing, as the in-order search finds it.
                            constexpr int EPL = 64 / LPR;
                            const int oc = lane % LPR, slot = lane / LPR;
                            const float hi_lim = s.maxref * 1.25f, lo_lim = s.minref * 1.25f;
                            const int need_o = __shfl((int)(can & (jit < 0)), oc);
                            const int clip_o = __shfl((int)in_a, oc);
                            const int cnt_o = __shfl(cnt, oc);
                            const int i0_o = __shfl(i0, oc);
                            const int flt_o = __shfl(s.filter_on, oc);
                            const int sp_o = __shfl(sp, oc);
                            const float cen_o = __shfl(s.center, oc), hl_o = __shfl(hi_lim, oc), ll_o = __shfl(lo_lim, oc);
                            const float mx_o = __shfl(s.max, oc), mn_o = __shfl(s.min, oc), ls_o = __shfl(s.lastsample, oc);
                            const float* po = (flt_o ? &L.flt[rw * LPR + oc][0] : &L.raw[rw * LPR + oc][0]) + base + sp_o;
                            int found = -1;
#pragma unroll
                            for (int r = 0; r < (12 + EPL - 1) / EPL; r++) {
                                const int k = slot + r * EPL;
                                bool hit = false;
                                if (k < 12) {
                                    float x = po[k], xp = po[k - 1];
                                    const float xc = x > mx_o ? mx_o : (x < mn_o ? mn_o : x);
                                    const float xpc = xp > mx_o ? mx_o : (xp < mn_o ? mn_o : xp);
                                    x = clip_o ? xc : x;
                                    xp = k == 0 ? ls_o : (clip_o ? xpc : xp);
                                    const bool cross = (x > cen_o) ? (!(x > hl_o) && xp < cen_o) : (!(x < ll_o) && xp > cen_o);
                                    // a crossing at symbol index -1 (sample 0 of a symbol that slipped early) latches nothing: the
                                    // in-order search stores -1 and keeps looking
                                    hit = need_o && k < cnt_o && cross && (i0_o + k >= 0);
                                }
                                const unsigned long long bal = __ballot(hit);
                                // this channel's column: bits oc, oc + CPW, ... of the ballot
                                unsigned long long col = bal >> oc;
                                unsigned long long m = 0;
#pragma unroll
                                for (int e = 0; e < EPL; e++) {
                                    m |= 1ull << (e * LPR);
                                }
                                col &= m;
                                if (found < 0 && col != 0) {
                                    found = (__ffsll((long long)col) - 1) / LPR + r * EPL;
                                }
                            }
                            jit = (can & (jit < 0) & (found >= 0)) ? i0 + found : jit;
                        }
                        DBG_SEC(1);
                        if (can) {
                            // the five window samples (indices centre - 2 .. centre + 2 of the symbol) and its last sample.  In frame
                            // min < max and the samples are finite, so the reference's two-sided clip is the median of three (a
                            // zero's sign may differ, which a sum that starts at +0 cannot show); hunting symbols are not clipped.
                            const float lo = in_a ? s.min : -inf, hi = in_a ? s.max : inf;
                            const float* pw = p + ((whole - 1) / 2 - 2 - i0);
                            float acc = 0.0f;
#pragma unroll
                            for (int w = 0; w < 5; w++) {
                                acc += __builtin_amdgcn_fmed3f(pw[w], lo, hi);
                            }
                            const float xl = p[cnt - 1];
                            const float xlc = xl > s.max ? s.max : (xl < s.min ? s.min : xl);
                            const float sym = acc / 5.0f;
                            s.jitter = jit;
                            s.lastsample = in_a ? xlc : xl;
                            sp += cnt;
                            s.in_symbol = 0;
                            commit_pre(sym);
                            DBG_SEC(2);
                            int fl = 0;
                            float q_max = 0.0f, q_min = 0.0f;
                            if (in_a) {
                                commit_inframe(sym, 0, fl, q_max, q_min);
                            }
                            DBG_SEC(3);
                            if (!in_a) {
                                commit_hunt(sym, 0, fl);
                            }
                            DBG_SEC(4);
                            emit(sym, fl, q_max, q_min);
                            DBG_SEC(5);
                        }
                        continue;
                    }
                    all_std_wait = all_ok; // nobody can go on and nobody needs the general trip: the tile is over
                }
                // ---- general trip ------------------------------------------------------------------------------------------
                const int done_snap = 0;
                const bool glive = alive && !gblocked;
                const bool gneed = glive && (sp < tn || s.in_symbol);
                {
                    const bool tile_over = all_std_wait || !__any(gneed && (sp < tn));
                    if (HM && tile_over && __any(hwait)) { // nobody can go on, but a lane waits for its handler: not the tile's end
                        __builtin_amdgcn_s_sleep(2);
                        respin = true;
                        continue;
                    }
                    if (tile_over || ++guard > 4 * TW) {
                        break;
                    }
                }
                if (!__any(gneed)) {
                    continue;
                }
                // ---- symbol start ----------------------------------------------------------------------------------
                if (glive && sp < tn && !s.in_symbol) {
                    if (s.need_reset) {
                        timing_reset();
                    }
                    int sps = whole;
                    if (rem > 0) {
                        int acc = s.sps_accum + rem;
                        if (acc >= cfg.sym_rate) {
                            sps++;
                            acc -= cfg.sym_rate;
                        }
                        s.sps_accum = acc;
                        sps = sps > 64 ? 64 : sps;
                    }
                    s.span = sps;
                    s.centre = (sps - 1) / 2;
                    s.i = 0;
                    s.sum = 0.0f;
                    s.count = 0;
                    s.in_symbol = 1;
                    if (sps > 1 && s.have_sync == 0 && s.jitter >= 0) {
                        if (s.jitter > 0 && s.jitter <= s.centre) {
                            s.i--;
                        } else if (s.jitter > s.centre && s.jitter < sps) {
                            s.i++;
                        }
                        s.jitter = -1;
                    }
                }
                // ---- whole-symbol evaluation -----------------------------------------------------------------------
                const int cnt = s.span - s.i; // samples this symbol still consumes
                const bool cold = s.filter_on && (abs0 + t0 + sp - s.filt_start) < (long long)(NT - 1);
                const bool wholeok = glive && s.in_symbol && cnt > 0 && cnt <= WMAX && !cold;
                const bool fits = sp + cnt <= tn;
                const bool blocked = wholeok && !fits && more; // wait for the next tile, both tiles stay staged
                const bool fo = s.filter_on != 0;
                const float* row = fo ? frow : rrow;
                const bool latched = wholeok && fits && s.i == 0 && s.have_sync && s.jitter >= 0 && s.span >= 6
                                     && s.span != 20;
                if (latched) {
                    // in frame with the crossing latched: nothing per-sample can change state, the symbol is the mean
                    // of the five clipped window samples (added in sample order)
                    const int c = s.centre;
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        float x = rd(row, sp + c - 2 + k);
                        x = x > s.max ? s.max : (x < s.min ? s.min : x);
                        acc += x;
                    }
                    float xl = rd(row, sp + s.span - 1);
                    xl = xl > s.max ? s.max : (xl < s.min ? s.min : xl);
                    s.sum = acc;
                    s.count = 5;
                    s.lastsample = xl;
                    sp += s.span;
                    s.i = s.span;
                }
                // Unlatched symbol of ordinary length (every hunting symbol): the crossing search needs each sample once,
                // in order, and nothing else; the window mean is the five centre samples.  Straight-line code, one
                // compare chain per sample instead of the general per-sample body.
                const bool genf = wholeok && fits && !latched && cnt <= 12 && s.span != 20 && s.span != 5;
                if (__any(genf)) {
                    const bool clip = s.have_sync != 0;
                    const float hi_lim = s.maxref * 1.25f, lo_lim = s.minref * 1.25f;
                    const float* rp0 = row + base + sp;
                    float last = s.lastsample;
                    int jit = s.jitter;
                    const bool anyclip = __any(genf && clip); // hunting lanes never clip: skip the selects for them
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        // the search stops at the first crossing; once every lane has one, only the last sample matters
                        if ((k & 3) == 0 && k > 0 && !__any(genf && k < cnt && jit < 0)) {
                            break;
                        }
                        float x = rp0[k];
                        if (anyclip) {
                            const float xc = x > s.max ? s.max : (x < s.min ? s.min : x);
                            x = clip ? xc : x;
                        }
                        const bool in = genf && k < cnt;
                        const bool cross = (x > s.center) ? (!(x > hi_lim) && last < s.center)
                                                          : (!(x < lo_lim) && last > s.center);
                        jit = (in && jit < 0 && cross) ? s.i + k : jit;
                        last = in ? x : last;
                    }
                    if (genf) { // lastsample = the symbol's final sample, whether or not the loop ran that far
                        float x = rp0[cnt - 1];
                        const float xc = x > s.max ? s.max : (x < s.min ? s.min : x);
                        last = clip ? xc : x;
                    }
                    float acc = s.sum;
                    int cw = s.count;
#pragma unroll
                    for (int w = 0; w < 5; w++) {
                        const int k = s.centre - 2 + w - s.i;
                        if (genf && k >= 0 && k < cnt) {
                            float x = rp0[k];
                            const float xc = x > s.max ? s.max : (x < s.min ? s.min : x);
                            acc += clip ? xc : x;
                            cw++;
                        }
                    }
                    if (genf) {
                        s.jitter = jit;
                        s.lastsample = last;
                        s.sum = acc;
                        s.count = cw;
                        sp += cnt;
                        s.i = s.span;
                    }
                }
                const bool gen = wholeok && fits && !latched && !genf;
                if (__any(gen)) {
                    for (int k = 0; k < WMAX; k++) {
                        const bool on = gen && k < cnt;
                        if (!__any(on)) {
                            break;
                        }
                        if (on) {
                            float x = rd(row, sp + k);
                            if (s.have_sync) {
                                x = x > s.max ? s.max : (x < s.min ? s.min : x);
                            }
                            const int i = s.i + k;
                            if (s.jitter < 0) {
                                if (x > s.center) {
                                    if (!(x > s.maxref * 1.25f) && s.lastsample < s.center) {
                                        s.jitter = i;
                                    }
                                } else if (!(x < s.minref * 1.25f) && s.lastsample > s.center) {
                                    s.jitter = i;
                                }
                            }
                            if (s.span == 20 && i >= 7 && i <= 13) {
                                s.sum += x;
                                s.count++;
                            }
                            if ((s.span == 5 && i == 2) || (i >= s.centre - 2 && i <= s.centre + 2)) {
                                s.sum += x;
                                s.count++;
                            }
                            s.lastsample = x;
                        }
                    }
                    if (gen) {
                        sp += cnt;
                        s.i = s.span;
                    }
                }
                // ---- sample-at-a-time path (filter cold start, long spans, last tile of the call) --------------------
                bool act = glive && s.in_symbol && sp < tn && s.i < s.span && !blocked;
                while (__any(act)) {
                    if (act) {
                        float x = rd(rrow, sp);
                        if (s.filter_on) {
                            const long long a = abs0 + t0 + sp;
                            if (a - s.filt_start >= (long long)(NT - 1)) {
                                x = rd(frow, sp);
                            } else {
                                // first 90 samples after the enable: taps that reach before the enable sample see the
                                // filter's memory (zeros on a fresh stream, the samples of the hunt that ended in a
                                // carrier loss otherwise); samples of this and the previous tile come from the LDS ring,
                                // older ones from HBM.
                                const long k = t0 + sp;
                                float acc = 0.0f;
                                for (int i = 0; i < NT; i++) {
                                    const int jr = sp - (NT - 1) + i; // tile-relative
                                    const long j = k - (NT - 1) + i;  // call-relative
                                    float v;
                                    if (abs0 + j < s.filt_start) { // the filter's memory as the last hunt left it (zeros at first)
                                        v = fstale[(size_t)ch * (NT - 1) + (size_t)((abs0 + j - s.filt_start) + (NT - 1))];
                                    } else if (jr >= (it > 0 ? -TW : 0)) { // the ring has no previous tile on a call's first
                                        v = rd(rrow, jr);
                                    } else {
                                        v = (j >= 0) ? raw[(size_t)ch * stride + j]
                                                     : prev_tail[(size_t)ch * (NT - 1) + (NT - 1) + j];
                                    }
                                    acc += __uint_as_float(c_taps[i]) * v;
                                }
                                x = acc;
                            }
                        }
                        if (s.have_sync) {
                            x = x > s.max ? s.max : (x < s.min ? s.min : x);
                        }
                        const int i = s.i;
                        if (s.jitter < 0) {
                            if (x > s.center) {
                                if (!(x > s.maxref * 1.25f) && s.lastsample < s.center) {
                                    s.jitter = i;
                                }
                            } else if (!(x < s.minref * 1.25f) && s.lastsample > s.center) {
                                s.jitter = i;
                            }
                        }
                        if (s.span == 20 && i >= 7 && i <= 13) {
                            s.sum += x;
                            s.count++;
                        }
                        if ((s.span == 5 && i == 2) || (i >= s.centre - 2 && i <= s.centre + 2)) {
                            s.sum += x;
                            s.count++;
                        }
                        s.lastsample = x;
                        s.i++;
                        sp++;
                    }
                    act = glive && s.in_symbol && sp < tn && s.i < s.span && !blocked;
                }
                // ---- symbol commit ---------------------------------------------------------------------------------
                const bool done = glive && s.in_symbol && s.i >= s.span;
                if (done) {
                    const float sym = (s.count > 0) ? (s.sum / (float)s.count) : 0.0f;
                    s.in_symbol = 0;
                    commit_pre(sym);
                    int fl = 0;
                    float q_max = 0.0f, q_min = 0.0f;
                    if (s.have_sync) {
                        commit_inframe(sym, done_snap, fl, q_max, q_min);
                    } else {
                        commit_hunt(sym, done_snap, fl);
                    }
                    emit(sym, fl, q_max, q_min);
                }
                gblocked = gblocked || blocked;
            }
            if ((DDN_RX_CYCLES && (cfg.dbg & 8192)) && dbg_kind >= 0) {
                dbg_cyc[dbg_kind] += (long long)clock64() - dbg_prev;
                dbg_n[dbg_kind]++;
                dbg_kind = -1;
            }
            if (offload && lane == 0) {
                L.qn[it & 1][rw] = (tk - 1) < QTW ? (tk - 1) : QTW; // trips that may have queued (the last one broke out at its top)
            }
            if (live) {
                L.sidx0[(it + 1) & 1][ln] = s.sidx;
                sp -= TW;
            }
            if (HM) { // this wave's queue half, trip count and checkpoint slots are in LDS: the tile is done
                // ... and with it goes the word on the unfiltered row: wanted with the filter off, in the tile a cold start began in
                // or reaches into (the hint of tile it serves tile it + 1 or it + 2, a cold start reaches one tile on), from 64
                // hunting symbols before the count that gates the filter off (1800; a lane with lastsync == 2 goes on to 10200 and
                // is served all the way), and always where there is no filter row to read instead
                const bool want = live
                                  && (!s.filter_on || cold_until > 0 || (!s.have_sync && s.hunt_pos + 64 >= 1800) || raw_always);
                const int raw_bit = __any(want) ? DDN_RXW_RAW : 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) {
                    __hip_atomic_store(&H.tile_done[rw], (it + 1) | raw_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (DDN_RX_CYCLES && (cfg.dbg & 8192)) {
            const long long t1 = (long long)clock64();
            if (!HM) {
                __syncthreads();
            }
            dbg_busy += t1 - dbg_t0 - dbg_spin;
            dbg_wait += (long long)clock64() - t1 + dbg_spin;
        } else if (!HM) {
            __syncthreads();
        }
    }
    if ((DDN_RX_CYCLES && (cfg.dbg & 8192)) && lane == 0) { // timing experiment only: cycles per wave in the tile body / at the tile barrier
        // (written over the unused tail of the workgroup's first channel's record area)
        uint8_t* d = rec + ((size_t)ch0 + 1) * max_sym * 10 - 192 + (wave < NRW ? (wave ? 1 : 0) : 2) * 64; // (recurrence 0, another one, staging)
        const long long v[8] = {dbg_busy, dbg_wait, dbg_cyc[0], dbg_cyc[1], dbg_cyc[2], dbg_n[0], dbg_n[1], dbg_n[2]};
        for (int k = 0; k < 64; k++) {
            d[k] = reinterpret_cast<const uint8_t*>(v)[k];
        }
        if (wave == 0) { // the recurrence wave's standard-trip sections, one block further down; its lean-run figures, another one
            uint8_t* d2 = rec + ((size_t)ch0 + 1) * max_sym * 10 - 256;
            for (int k = 0; k < 64; k++) {
                d2[k] = reinterpret_cast<const uint8_t*>(dbg_sec)[k];
                d2[k - 64] = reinterpret_cast<const uint8_t*>(dbg_run)[k];
                d2[k - 128] = reinterpret_cast<const uint8_t*>(dbg_blk)[k];
            }
        }
    }
    if (!HM && loader && offload && it > 0) { // (handler mode: the staging wave's last job)
        drain((it - 1) & 1);
    }
    if (live) {
        if (s.have_sync) {
            s.umid = ((s.max - s.center) * 5.0f / 8.0f) + s.center;
            s.lmid = ((s.min - s.center) * 5.0f / 8.0f) + s.center;
        }
        s.n_abs = abs0 + n;
        state[ch] = s;
        counts[ch] = o;
        for (int k = 0; k < SS; k++) {
            sbuf_store[(size_t)k * n_channels + ch] = L.sb[k][ln];
        }
        for (int k = 0; k < 24; k++) {
            lbuf_store[(size_t)k * n_channels + ch] = L.lb[k][ln];
            shist_store[(size_t)k * n_channels + ch] = L.sh[k][ln];
        }
    }
}

template <int CPW, bool HM, int NRW_T = (HM ? 2 : 1)>
static hipError_t
launch_rxw(const float* raw, const float* filt, const float* prev_tail, float* fstale, long n, size_t stride, int n_channels,
           const DdnRxConfig& cfg, DdnRxState* state, float* sbuf_store, float* lbuf_store, float* shist_store,
           float* minring, float* maxring, uint8_t* rec, uint8_t* flags, int32_t* counts, size_t max_sym, const int32_t* lock_cfg,
           DdnP25HState* hstate, float* hh_store, int32_t* events, int32_t* n_events, hipStream_t st) {
    const size_t shm1 = HM ? (((((sizeof(LdsW<CPW, true>) + 15) & ~(size_t)15) + sizeof(LdsH<CPW>)) + 15) & ~(size_t)15) : sizeof(LdsW<CPW>);
    const size_t shm = shm1 * (NRW_T == 4 ? 2 : 1); // (two logical workgroups per launch workgroup, see the kernel)
    const unsigned lblocks = (unsigned)((n_channels + CPW - 1) / CPW);
    const unsigned nblocks = NRW_T == 4 ? (lblocks + 1) / 2 : lblocks;
    const unsigned nthreads = HM ? 64 * (NRW_T + 2) * (NRW_T == 4 ? 2 : 1) : 192;
    if (HM && (cfg.sym_rate <= 0 || cfg.out_rate / cfg.sym_rate < 9)) {
        return hipErrorInvalidValue; // the handler-mode window bookkeeping is sized for >= 9 samples per symbol
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p25_rxw<CPW, HM, NRW_T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) {
        return e;
    }
    if (DDN_EXP_ENV("DDN_RX_OCC")) {
        int nb = -1;
        hipFuncAttributes fa;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_p25_rxw<CPW, HM, NRW_T>), (int)nthreads, shm);
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_p25_rxw<CPW, HM, NRW_T>));
        fprintf(stderr, "k_p25_rxw<%d,%d,%d>: %d blocks per CU, %d regs, %zu B dynamic LDS, %zu B static\n", CPW, (int)HM, NRW_T, nb, fa.numRegs, shm, fa.sharedSizeBytes);
    }
    hipLaunchKernelGGL((k_p25_rxw<CPW, HM, NRW_T>), dim3(nblocks), dim3(nthreads), shm, st, raw,
                       filt, prev_tail, fstale, n, stride, n_channels, cfg, state, sbuf_store, lbuf_store, shist_store, minring,
                       maxring, rec, flags, counts, max_sym, lock_cfg, hstate, hh_store, events, n_events);
    return hipGetLastError();
}

template <int CPW>
static hipError_t
launch_rx(const float* raw, const float* filt, const float* prev_tail, float* fstale, long n, size_t stride, int n_channels,
          const DdnRxConfig& cfg, DdnRxState* state, float* sbuf_store, float* lbuf_store, float* shist_store,
          float* minring, float* maxring, uint8_t* rec, uint8_t* flags, int32_t* counts, size_t max_sym, const int32_t* lock_cfg,
          hipStream_t st) {
    const size_t shm = sizeof(Lds<CPW>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_p25_rx<CPW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_p25_rx<CPW>, dim3((unsigned)((n_channels + CPW - 1) / CPW)), dim3(128), shm, st, raw, filt,
                       prev_tail, fstale, n, stride, n_channels, cfg, state, sbuf_store, lbuf_store, shist_store, minring,
                       maxring, rec, flags, counts, max_sym, lock_cfg);
    return hipGetLastError();
}

static inline bool
use_filter_row_missing(const DdnRxConfig* cfg, const float* filt) {
    return cfg->use_filter && !filt;
}

// channels per wavefront of a handler-mode launch (0 = by batch size)
static int
handler_mode_cpw(int channels_per_wave, int n_channels) {
    int cpw = channels_per_wave;
    if (const char* e = DDN_EXP_ENV("DDN_RX_CPW")) { // (experiments)
        cpw = atoi(e);
    }
    if (cpw != 4 && cpw != 8 && cpw != 16) {
        // two workgroups per CU are resident: up to 2048 channels four per workgroup (two lanes per recurrence wave - a lane's
        // standard trip, bulk pass or wait then holds up one other lane instead of three), beyond that eight.  Larger batches
        // keep eight and run in rounds of 512 resident workgroups (a workgroup never waits for another): measured on the bench
        // traffic, 8192 channels take 19.6 ms sixteen per workgroup (eight lanes per recurrence wave) against two rounds of
        // the 5.3 ms the eight-channel shape takes for 4096 (bench.py batch_sweep, round 5)
        cpw = n_channels <= 4 * 512 ? 4 : 8;
    }
    return cpw;
}

// 1: this configuration's loop kernel computes the matched filter itself (handler mode with 128-sample tiles: the handler wave filters
// each staged tile in LDS) - the caller may pass filt = NULL and leave k_p25_matched_filter out.
extern "C" int
ddn_dev_p25_rx_fuses_filter(const DdnRxConfig* cfg, int channels_per_wave, int n_channels) {
    return cfg && cfg->handlers && cfg->use_filter && handler_mode_cpw(channels_per_wave, n_channels) <= 8
               ? 1
               : 0;
}

extern "C" hipError_t
ddn_dev_p25_rx(const float* raw, const float* filt, const float* prev_tail, float* fstale, long n, size_t stride, int n_channels,
               const DdnRxConfig* cfg, DdnRxState* state, float* sbuf_store, float* lbuf_store, float* shist_store,
               float* minring, float* maxring, uint8_t* rec, uint8_t* flags, int32_t* counts, size_t max_sym,
               int channels_per_wave, const int32_t* lock_cfg, DdnP25HState* hstate, float* hh_store, int32_t* events,
               int32_t* n_events, hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    if ((unsigned long long)n_channels * (unsigned long long)MS * 4ull > 0xFFFFFFFFull) {
        return hipErrorInvalidValue; // the extrema rings are addressed with 32-bit byte offsets
    }
    {   // the matched-filter taps are a __constant__ of this code object: one upload per device, under a lock
        static std::mutex mu;
        static bool taps_up[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) {
            return e;
        }
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64 || !taps_up[dev]) {
            e = hipMemcpyToSymbol(HIP_SYMBOL(c_taps), ddn_p25_filter_bits, sizeof(uint32_t) * NT);
            if (e != hipSuccess) {
                return e;
            }
            if (dev >= 0 && dev < 64) {
                taps_up[dev] = true;
            }
        }
    }
    int cpw = channels_per_wave;
    if (const char* e = DDN_EXP_ENV("DDN_RX_CPW")) { // (experiments)
        cpw = atoi(e);
    }
    const int whole = cfg->sym_rate > 0 ? cfg->out_rate / cfg->sym_rate : 0;
    if (cfg->handlers) {
        // handler mode: the windowed kernel with its fourth wave, 8 or 16 lanes per wave (the in-frame history ring is LDS)
        if (whole < 6 || !hstate || !hh_store || !events || !n_events) {
            return hipErrorInvalidValue;
        }
        cpw = handler_mode_cpw(channels_per_wave, n_channels);
        if (use_filter_row_missing(cfg, filt) && cpw > 8) {
            return hipErrorInvalidValue; // sixteen channels per workgroup run 64-sample tiles: the filter row has to be given
        }
        if (cpw == 4) {
            return launch_rxw<4, true>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                       shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, hstate, hh_store,
                                       events, n_events, st);
        }
        if (cpw == 8) {
            // cfg.dbg bit 16777216: four recurrence waves of two lanes (six-wave logical workgroups, two per launch workgroup).  Measured
            // at 4096 x 48000 of the bench traffic: 6.4 ms against 5.4 ms for the two-wave shape - the one staging wave then serves four
            // halves a tile (35 k cycles of work for a tile the recurrences finish in 29 k) and becomes the critical path; the shape
            // pays once the slice / record stores leave that wave.  Kept selectable, off by default.
            if (cfg->dbg & 16777216) {
                return launch_rxw<8, true, 4>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                              shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, hstate, hh_store,
                                              events, n_events, st);
            }
            return launch_rxw<8, true>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                       shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, hstate, hh_store,
                                       events, n_events, st);
        }
        return launch_rxw<16, true>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                    shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, hstate, hh_store,
                                    events, n_events, st);
    }
    if (cpw != 8 && cpw != 16 && cpw != 32 && cpw != 64) {
        // fewest lanes per wavefront that still gives every CU (256) no more than ~2 workgroups: a trip is only a lean trip
        // when every lane of the wave is in the lean state, so fewer lanes per wave means fewer mixed trips (measured on the
        // bench traffic at 4096 channels: 6.07 ms with 8 lanes per wave, 6.39 ms with 16)
        cpw = n_channels <= 8 * 512 ? 8 : (n_channels <= 16 * 512 ? 16 : (n_channels <= 32 * 512 ? 32 : 64));
    }
    // CPW 16 / 32 run the windowed variant (k_p25_rxw); 64 lanes per wavefront keeps the two-tile kernel, whose LDS
    // footprint still fits (cfg.dbg bit 128 forces it for A/B timing)
    // k_p25_rxw sizes its window bookkeeping for symbols of at least 6 samples; shorter ones keep the two-tile kernel
    if (whole < 6) {
        if (cpw <= 16) {
            return launch_rx<16>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                 shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
        }
        if (cpw == 32) {
            return launch_rx<32>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                 shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
        }
        return launch_rx<64>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                             shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
    }
    if (cpw == 8) {
        return launch_rxw<8, false>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                    shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, nullptr, nullptr, nullptr,
                                    nullptr, st);
    }
    if (cpw == 16 && !(cfg->dbg & 128)) {
        return launch_rxw<16, false>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                     shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, nullptr, nullptr,
                                     nullptr, nullptr, st);
    }
    if (cpw == 32 && !(cfg->dbg & 128)) {
        return launch_rxw<32, false>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                     shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, nullptr, nullptr,
                                     nullptr, nullptr, st);
    }
    switch (cpw) {
        case 16:
            return launch_rx<16>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                 shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
        case 32:
            return launch_rx<32>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                 shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
        default:
            return launch_rx<64>(raw, filt, prev_tail, fstale, n, stride, n_channels, *cfg, state, sbuf_store, lbuf_store,
                                 shist_store, minring, maxring, rec, flags, counts, max_sym, lock_cfg, st);
    }
}
