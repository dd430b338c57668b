// ddn_rx4.hip - batched fixed-protocol receive loop for the other 4-level FSK protocols of BASELINE configs[3]: DMR and
// NXDN48.  One lane = one channel running what the reference's decoder thread runs per stream between
// rtl_stream_read() and the protocol handler, driven by a profile (DdnFsk4Config):
//   getSymbol()      src/dsp/dsd_symbol.c:1343-1387,1769-1805; window selection :197-224 (C4FM left edge 1 once a DMR type
//                    is the last sync; GFSK = the two samples next to the centre), accumulation :404-460 (sps-20 7..13 window
//                    on top), slip rules :462-517 (sps 20 / GFSK / C4FM), clip only on C4FM :347-358, matched-filter family
//                    by lastsynctype :301-338 (dmr_filter / nxdn_filter, src/dsp/dsd_filters.c:173-200,348-356)
//   getFrameSync()   src/dsp/dsd_frame_sync.c:3098-3148; ring :1729-1764; sign dibit :2110-2127; 4-level payload dibit +
//                    reliability stored while hunting :2161-2189; level window :2316-2336; timeouts :2753-2760,3037-3053
//     DMR accept     :1102-1106,1108-1314: basic lock :385-392 + dmr_resample_on_sync() src/dsp/dmr_sync.c:63-131 (warm start
//                    over the 24 sync symbols, re-digitisation of the 66 payload dibits before them)
//     NXDN accept    :1507-1556 (10-symbol window, five patterns per polarity, accepted on the second consecutive match)
//   in-frame symbol  get_dibit_and_analog_signal, src/core/frames/dsd_dibit.c:1045-1076; use_symbol :243-299 (no continuous
//                    threshold update outside P25p1); digitize :1018-1043
// Deviations from a full dsd-neo run (include/ddn_fsk4.h): modulation locked, one protocol, handler = configured symbol count.
//
// Outputs per symbol: the 10-byte capture record + flags as ddn_rx.hip, plus {payload dibit, reliability} - what the
// reference appends to dmr_payload_buf / dmr_soft_buf for every symbol, hunting or not.  Per accepted sync: the index of its
// last symbol and the 90 payload dibits ending there (CACH + first half + slot-type prefix + sync for a DMR burst) AFTER the
// re-digitisation, so a burst decoder needs nothing from before the current call.
//
// GPU shape: as k_p25_rx (ddn_rx.hip) - wave 0 of a workgroup runs CPW channel recurrences, wave 1 stages the next
// 64-sample tile of every channel (raw + always-on matched-filter output) with coalesced row loads; lanes advance symbol by
// symbol.  P25p1 stays on its own kernels (live thresholds need the window / ring machinery of ddn_rx.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#include "ddn_device.h"
#include "ddn_slicer_dev.h"
#include "ddn_fsk4h_dev.h"
#include "ddn_tables_fsk4.h"
#include "ddn_tables_ambe.h"

namespace {
constexpr int RTILES = 4;               // staged tiles per channel (power of two: ring index = sample index & RMASK)
constexpr int HN = DDN_FSK4_HIST;

template <int CPW>
struct Lds4 {
    // tile shape: 128-sample tiles where the rows are few (half the round overhead per symbol: a 64-sample tile holds only
    // 3.2 NXDN48 symbols), 64 from 16 lanes per wave on (LDS)
    static constexpr int TSW = CPW <= 8 ? 128 : 64, RMASKW = RTILES * TSW - 1;
    static constexpr int QCAPW = CPW <= 8 ? 24 : 12; // symbols a lane can finish in one round (TSW / 7 + 1) + sync entries + slack
    float lb[24][CPW];
    float sh[HN][CPW];
    uint8_t ph[HN][CPW];
    uint8_t rh[HN][CPW];
    float raw[CPW][RTILES * TSW + 1]; // ring of RTILES tiles per channel; the + 1 skews the rows over the LDS banks
    float flt[CPW][RTILES * TSW + 1];
    // hand-off to the helper wave, one buffer per round parity: per lane up to QCAPW entries {symbol, centre, umid, lmid,
    // max, min, meta, aux}.  meta = flags | slot << 8 | kind << 16; a sync entry (kind 1) follows the entry of the accepting
    // symbol and carries the thresholds AFTER the warm start, meta |= redigitise << 17 | scount << 24, aux = sync index
    float q[2][QCAPW][8][CPW];
    int qn[2][CPW];
    int qo[2][CPW];
    uint32_t pat_bits[DDN_FSK4_MAX_PAT];
    uint32_t pat_meta[DDN_FSK4_MAX_PAT]; // type | neg << 8 | class << 16
};

// handler mode: the handlers' per-channel words and the dibits of the burst / frame being read (ddn_fsk4h_dev.h)
template <int CPW>
struct Lds4H {
    int hs[ddn_fsk4h::F_COUNT][CPW];
    uint8_t pay[144][CPW];
};

__device__ __forceinline__ void
no_carrier(DdnFsk4State& s) { // src/engine/engine.c:1838-1847 as far as this loop sees it
    s.jitter = -1;
    s.lastsync = 0;
    s.filter_on = 0;
    s.max = 15000.0f;
    s.min = -15000.0f;
    s.center = 0.0f;
    s.need_reset = 1;
}
__device__ __forceinline__ void
timing_reset(DdnFsk4State& s) { // dsd_symbol.c:1306-1341
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
}
__device__ __forceinline__ void
hunt_restart(DdnFsk4State& s) {
    s.hunt_pos = 0;
    s.have_sync = 0;
    s.lidx = 0;
    s.level_count = 0;
    s.hist_count = 0;
    s.hist_bits = 0;
    s.lmin = s.min;
    s.lmax = s.max;
}
template <int PROTO>
struct Fsk4Cfg {
    // PROTO 1 DMR, 2 NXDN48, 3 NXDN96: NXDN's sync words, two-match confirmation and LICH gate at 4800 symbols/s on the 4800_4 hunt
    // profile (level ring 24, src/dsp/dsd_frame_sync.c:1525-1556,1729-1744) behind the DMR matched filter
    // (symbol_apply_matched_filter(), src/dsp/dsd_symbol.c:323-335)
    // PROTO 4 M17 (round 5): C4FM lock at 4800 symbols/s without a matched filter (decode_mode_apply_m17(),
    // src/runtime/decode_mode.c:486-510), eight-symbol words matched with one error allowed by frame_sync_try_m17()
    // (src/dsp/dsd_frame_sync.c:865-1100: m17_hit() below; the pattern table only names the twelve outcomes), 8-symbol warm start,
    // fixed counts behind a sync (dispatch_m17.c:25-68: preamble 8, everything else 184)
    // PROTO 5 YSF (round 5): the 20-symbol FUSION_SYNC compared exactly in both polarities (frame_sync_try_ysf(),
    // src/dsp/dsd_frame_sync.c:770-797), 20-symbol warm start, the DMR matched filter (src/dsp/dsd_symbol.c:306-309), a fixed count
    // behind a sync (processYSF() reads 100 + 360 dibits for every frame type but FI = 3 with DT != 1, src/protocol/ysf/ysf.c)
    static constexpr int sym_rate = PROTO == 2 ? 2400 : 4800;
    static constexpr int win_len = PROTO == 1 ? 24 : (PROTO == 4 ? 8 : (PROTO == 5 ? 20 : 10)), t_max = PROTO == 2 ? 12 : 24;
    static constexpr int warm_len = PROTO == 1 ? 24 : (PROTO == 4 ? 8 : (PROTO == 5 ? 20 : 10));
    static constexpr int n_pat = PROTO == 1 ? 8 : (PROTO == 4 ? 12 : (PROTO == 5 ? 2 : 10));
    static constexpr int confirm = (PROTO == 1 || PROTO == 4 || PROTO == 5) ? 0 : 1, dmr_window = PROTO == 1 ? 1 : 0, redigitize = PROTO == 1 ? 1 : 0;
    static constexpr bool m17 = PROTO == 4;
    static constexpr int slow_type = 0;
    static constexpr int nt = PROTO == 2 ? DDN_NXDN48_FILTER_TAPS : DDN_DMR_FILTER_TAPS;
    int out_rate, rf_mod, use_filter, dbg;
};

// how often sample i of the current symbol enters the sum (symbol_accumulate_sample)
__device__ __forceinline__ int
adds(int i, int span, int c, int rf_mod, int l_edge) {
    int k = (span == 20 && i >= 7 && i <= 13) ? 1 : 0;
    if (span == 5 && i == 2) {
        return k + 1;
    }
    if (rf_mod == 0) {
        return k + ((i >= c - l_edge && i <= c + 2) ? 1 : 0);
    }
    if (span <= 4) {
        return k + (i == c ? 1 : 0);
    }
    return k + ((i == c - 1 || i == c + 1) ? 1 : 0);
}

// frame_sync_try_m17() with only M17 enabled (src/dsp/dsd_frame_sync.c:865-1100; max_hamming 1 for the preamble, no repeated-marker rule):
// w8 = the last eight sign dibits, oldest first; last = lastsynctype; pol = state->m17_polarity (0 unknown, 1 normal, 2 inverted);
// ty[k] = the type id of outcome k (0 / 1 preamble + / -, 2 / 3 EOT, 4 / 5 LSF, 6 / 7 BERT, 8 / 9 stream, 10 / 11 packet).
// Returns the outcome or -1; pol_after = the polarity the match leaves (preamble sets it, EOT clears it).
__device__ __forceinline__ int
m17_hit(uint32_t w8, int last, int pol, const uint32_t* pat_meta, int& pol_after) {
    enum { W_LSF = 0xF2, W_STR = 0x0D, W_PRE = 0x55, W_PIV = 0xAA, W_BRT = 0x4F, W_PKT = 0xB0, W_EOT = 0xFD, W_EOT_INV = 0x02 };
    auto ham1 = [&](uint32_t word) { return __popc((w8 ^ word) & 0xFFu) <= 1; };
    auto ty = [&](int k) { return (int)(pat_meta[k] & 0xFFu); };
    const bool inv = pol == 2;
    pol_after = pol;
    if (ham1(W_PRE)) {
        pol_after = 1;
        return 0;
    }
    if (ham1(W_PIV)) {
        pol_after = 2;
        return 1;
    }
    const bool after_frame = last == ty(4) || last == ty(5) || last == ty(8) || last == ty(9) || last == ty(10) || last == ty(11)
                             || last == ty(6) || last == ty(7);
    if (ham1(inv ? W_EOT_INV : W_EOT) && after_frame) {
        pol_after = 0;
        return inv ? 3 : 2;
    }
    const bool after_pre = (!inv && last == ty(0)) || (inv && last == ty(1));
    const bool after_brt = (!inv && last == ty(6)) || (inv && last == ty(7));
    if (after_pre || after_brt) {
        if (after_pre && ham1(inv ? W_STR : W_LSF)) {
            return inv ? 5 : 4;
        }
        if (ham1(inv ? W_PKT : W_BRT)) {
            return inv ? 7 : 6;
        }
    }
    if (ham1(W_STR) && !inv) {
        if (last == ty(4) || last == ty(8)) {
            return 8;
        }
    } else if (ham1(W_LSF) && inv) {
        if (last == ty(5) || last == ty(9)) {
            return 9;
        }
    }
    if (ham1(W_PKT) && !inv) {
        if (last == ty(4) || last == ty(10)) {
            return 10;
        }
    } else if (ham1(W_BRT) && inv) {
        if (last == ty(5) || last == ty(11)) {
            return 11;
        }
    }
    return -1;
}

// (one channel per wave is what a batch of <= 1536 channels runs: three waves per SIMD keep all of its workgroups resident at once)
template <int CPW, int MAXW, int PROTO, bool HM>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(CPW == 1 ? 3 : 2))) void
k_fsk4_rx(const float* __restrict__ raw, const float* __restrict__ filt, const float* __restrict__ prev_tail,
          float* __restrict__ fstale, const float* __restrict__ taps, long n_long, size_t stride, int n_channels,
          const DdnFsk4Config* __restrict__ cfgp, DdnFsk4State* __restrict__ state, float* __restrict__ lbuf_store,
          float* __restrict__ shist_store, uint8_t* __restrict__ phist_store, uint8_t* __restrict__ rhist_store,
          uint8_t* __restrict__ rec, uint8_t* __restrict__ flags, uint8_t* __restrict__ pay, int32_t* __restrict__ counts,
          size_t max_sym, const int32_t* __restrict__ lock4, int32_t* __restrict__ sync_pos, uint8_t* __restrict__ sync_pat,
          uint8_t* __restrict__ pre, uint8_t* __restrict__ pre_rel, int32_t* __restrict__ n_sync, int max_sync,
          int32_t* __restrict__ hwords, uint8_t* __restrict__ hpay_store, const DdnFec3Tables* __restrict__ htab,
          int32_t* __restrict__ events, int32_t* __restrict__ n_events) {
    constexpr int TSW = Lds4<CPW>::TSW, RMASKW = Lds4<CPW>::RMASKW, QCAPW = Lds4<CPW>::QCAPW;
    extern __shared__ unsigned char smem_raw[];
    Lds4<CPW>& L = *reinterpret_cast<Lds4<CPW>*>(smem_raw);
    Lds4H<CPW>& LH = *reinterpret_cast<Lds4H<CPW>*>(smem_raw + ((sizeof(Lds4<CPW>) + 15) & ~(size_t)15));
    const int n = (int)n_long; // the C-ABI keeps a call below 2^31 samples
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;
    const int ch0 = blockIdx.x * CPW;
    const int ch = ch0 + lane;
    const bool live = !loader && lane < CPW && ch < n_channels;
    const int ln = lane < CPW ? lane : 0;
    // The protocol's constants are compile-time (PROTO 1 = DMR, 2 = NXDN48; the host checks the profile against them), the
    // batch's choices are read once from the profile in device memory, its pattern table sits in LDS.
    using Cfg = Fsk4Cfg<PROTO>;
    Cfg cfg;
    cfg.out_rate = cfgp->out_rate;
    cfg.rf_mod = cfgp->rf_mod;
    cfg.use_filter = cfgp->use_filter;
    cfg.dbg = cfgp->dbg;
    if (threadIdx.x < DDN_FSK4_MAX_PAT) {
        L.pat_bits[threadIdx.x] = cfgp->pat_bits[threadIdx.x];
        L.pat_meta[threadIdx.x] = (uint32_t)cfgp->pat_type[threadIdx.x] | ((uint32_t)cfgp->pat_neg[threadIdx.x] << 8)
                                  | ((uint32_t)cfgp->pat_class[threadIdx.x] << 16);
    }
    const bool use_flt = cfg.use_filter != 0;
    float* const sync_thr_out = cfgp->sync_thr;
    constexpr int NT = Cfg::nt;

    DdnFsk4State s;
    if (live) {
        s = state[ch];
        if (HM) {
            for (int k = 0; k < ddn_fsk4h::F_COUNT; k++) {
                LH.hs[k][ln] = hwords[(size_t)ch * ddn_fsk4h::F_COUNT + k];
            }
            LH.hs[ddn_fsk4h::F_NEV][ln] = 0; // events are per call
            for (int k = 0; k < 144; k++) {
                LH.pay[k][ln] = hpay_store[(size_t)ch * 144 + k];
            }
        }
        for (int k = 0; k < 24; k++) {
            L.lb[k][ln] = lbuf_store[(size_t)k * n_channels + ch];
        }
        for (int k = 0; k < HN; k++) {
            L.sh[k][ln] = shist_store[(size_t)k * n_channels + ch];
        }
    } else {
        s = DdnFsk4State{};
        if (loader && lane < CPW && ch < n_channels) {
            for (int k = 0; k < HN; k++) {
                L.ph[k][ln] = phist_store[(size_t)k * n_channels + ch];
                L.rh[k][ln] = rhist_store[(size_t)k * n_channels + ch];
            }
        }
    }
    // Staged input = a ring of four 64-sample tiles per channel (sample g of the call sits at ring index g & 255): while the
    // symbols STARTING in tile t are evaluated, tile t + 1 is already complete and the loader wave fills t + 2, so a symbol is
    // always read whole - no symbol straddles a staging boundary (with 16 unsynchronised channels per wavefront some lane
    // would, on nearly every trip, and drag the whole wavefront through the sample-at-a-time loop).
    auto stage = [&](int t) {
        const long t0 = (long)t * TSW;
        if (t0 >= n) {
            return;
        }
        const int tn = (int)((n - t0) < TSW ? (n - t0) : TSW);
        const int slot = (t & (RTILES - 1)) * TSW;
        constexpr int RB = CPW < 16 ? CPW : 16; // rows staged per pass
#pragma unroll
        for (int half = 0; half < TSW / 64; half++) { // 64 samples of a row per pass
            const int j = lane + 64 * half;
#pragma unroll
            for (int h = 0; h < CPW / RB; h++) {
                float r[RB], f[RB];
#pragma unroll
                for (int c = 0; c < RB; c++) {
                    const int cc = RB * h + c;
                    const bool ok = (ch0 + cc < n_channels) && j < tn;
                    const size_t off = (size_t)(ch0 + cc) * stride + (size_t)t0 + j;
                    r[c] = ok ? raw[off] : 0.0f;
                    f[c] = (ok && use_flt) ? filt[off] : 0.0f;
                }
#pragma unroll
                for (int c = 0; c < RB; c++) {
                    L.raw[RB * h + c][slot + j] = r[c];
                    L.flt[RB * h + c][slot + j] = f[c];
                }
            }
        }
    };
    if (loader) {
        stage(0);
        stage(1);
    }
    __syncthreads();

    const int whole0 = cfg.out_rate / cfg.sym_rate, rem0 = cfg.out_rate % cfg.sym_rate;
    const int whole = whole0 < 2 ? 2 : (whole0 > 64 ? 64 : whole0);
    const int rem = (whole0 < 2 || whole0 > 64) ? 0 : rem0;
    // the lean in-frame trip: fixed symbol length of ordinary size (the 5-sample symbol has its own accumulation rule)
    const bool lean_ok = rem == 0 && whole >= 6 && whole <= MAXW && !(cfg.dbg & 1024);
    const uint32_t wmask = cfg.win_len >= 24 ? 0xFFFFFFu : ((1u << cfg.win_len) - 1u);
    // bulk hunting pass (see the trip loop): the slip of a hunting symbol's start by the latch the one before left
    // (symbol start in the general trip below), + 1, two bits per latch value -1 .. 30
    const bool bulk_ok = lean_ok && !(cfg.dbg & 4096);
    unsigned long long i0lut = 0;
    for (int j = -1; j < 31; j++) {
        const int c = (whole - 1) / 2;
        int i0 = 0;
        if (j >= 0 && whole > 1) {
            if (whole == 20) {
                i0 = (j >= 7 && j <= 10) ? -1 : ((j >= 11 && j <= 14) ? 1 : 0);
            } else if (cfg.rf_mod == 2) {
                i0 = (j >= c - 1 && j <= c) ? -1 : ((j >= c + 1 && j <= c + 2) ? 1 : 0);
            } else {
                i0 = (j > 0 && j <= c) ? -1 : ((j > c && j < whole) ? 1 : 0);
            }
        }
        i0lut |= (unsigned long long)(i0 + 1) << (2 * (j + 1));
    }
    int o = 0, ns = 0;
    const long long abs0 = s.n_abs;
    // ---- helper wave: slice / soft decision / record + payload stores / payload history / per-sync hand-over ----------------
    // None of it feeds back into the recurrence, so it runs one round behind on wave 1 (lane = channel) while wave 0 is
    // already on the next tile.  The payload history ring (ph / rh) belongs to this wave alone.
    // One lane per queue entry: 64 / CPW entries of every channel per pass (the entries of a tile are independent but for the output
    // index - a prefix count of the symbol entries - and for a sync's hand-over, which reads what the entries before it wrote: the
    // symbol entries of a pass come first, then its sync entries in queue order, each spread over the wavefront).  With one lane
    // per channel working through its entries one after the other this wave was the critical path of the whole kernel (measured
    // at 1365 DMR channels: 5.3 ms, 2.35 with the drain switched off).
    auto drain = [&](int qb) {
        if (cfg.dbg & 32768) { // (timing experiment: no records)
            return;
        }
        constexpr int EPP = CPW >= 64 ? 1 : 64 / CPW; // entries per channel and pass
        const int dc = lane % CPW, e0 = lane / CPW;
        const int dch = ch0 + dc;
        const bool dok = dch < n_channels && e0 < EPP;
        const int cnt = dok ? L.qn[qb][dc] : 0;
        int oo = dok ? L.qo[qb][dc] : 0;
        uint8_t* hrp = rec + (size_t)(dok ? dch : 0) * max_sym * 10;
        uint8_t* hfp = flags + (size_t)(dok ? dch : 0) * max_sym;
        uint8_t* hpp = pay + (size_t)(dok ? dch : 0) * max_sym * 2;
        unsigned long long chm = 0; // the lanes of this lane's channel
#pragma unroll
        for (int e = 0; e < EPP; e++) {
            chm |= 1ull << (e * CPW + dc);
        }
        const unsigned long long below = chm & ((1ull << lane) - 1ull);
        for (int kb = 0; kb < QCAPW; kb += EPP) {
            const int k = kb + e0;
            const bool have = k < cnt;
            if (!__any(have)) {
                break;
            }
            float sym = 0.0f;
            ddn_sl::Thr th = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            int meta = 0, aux = 0;
            if (have) {
                sym = L.q[qb][k][0][dc];
                th = {L.q[qb][k][1][dc], L.q[qb][k][2][dc], L.q[qb][k][3][dc], L.q[qb][k][4][dc], L.q[qb][k][5][dc]};
                meta = __float_as_int(L.q[qb][k][6][dc]);
                aux = __float_as_int(L.q[qb][k][7][dc]);
            }
            const int fl = meta & 0xFF, slot = (meta >> 8) & 0x7F;
            const bool is_sym = have && ((meta >> 16) & 1) == 0;
            const unsigned long long sb = __ballot(is_sym);
            const int my_oo = oo + __popcll(sb & below); // symbol entries of this channel ahead of this one
            oo += __popcll(sb & chm);
            if (is_sym) {
                int dibit, relb = 0, l0 = 0, l1 = 0, pd, pr;
                if (fl & 1) {
                    const int neg = (fl >> 2) & 1;
                    ddn_sl::slice_soft(sym, th, neg, dibit, relb, l0, l1);
                    pd = neg ? (dibit ^ 2) : dibit;
                    pr = relb;
                } else {
                    dibit = sym > 0.0f ? 1 : 3;
                    pd = sym > th.center ? (sym > th.umid ? 1 : 0) : (sym < th.lmid ? 3 : 2);
                    pr = ddn_sl::rel_from_thresholds(sym, th);
                }
                L.ph[slot][dc] = (uint8_t)pd;
                L.rh[slot][dc] = (uint8_t)pr;
                if ((size_t)my_oo < max_sym) {
                    uint8_t* r = hrp + (size_t)my_oo * 10;
                    const uint32_t xb = __float_as_uint(sym);
                    ((uint16_t*)r)[0] = (uint16_t)((dibit & 3) | (relb << 8));
                    ((uint16_t*)r)[1] = (uint16_t)(int16_t)l0;
                    ((uint16_t*)r)[2] = (uint16_t)(int16_t)l1;
                    ((uint16_t*)r)[3] = (uint16_t)(xb & 0xFFFFu);
                    ((uint16_t*)r)[4] = (uint16_t)(xb >> 16);
                    hfp[my_oo] = (uint8_t)fl;
                    ((uint16_t*)hpp)[my_oo] = (uint16_t)((pd & 3) | (pr << 8));
                }
            }
            // accepted syncs of this pass: `slot` = the sync's last symbol (its own entry sits just before), my_oo - 1 its index
            unsigned long long yb = __ballot(have && !is_sym);
            while (yb) {
                const int sl = __ffsll((long long)yb) - 1;
                yb &= yb - 1;
                const int s_meta = __builtin_amdgcn_readlane(meta, sl), s_aux = __builtin_amdgcn_readlane(aux, sl);
                const int s_oo = __builtin_amdgcn_readlane(my_oo, sl), s_c = sl % CPW;
                const float s_cen = __shfl(th.center, sl), s_um = __shfl(th.umid, sl), s_lm = __shfl(th.lmid, sl);
                const int s_slot = (s_meta >> 8) & 0x7F, scount = (s_meta >> 24) & 0xFF, s_fl = s_meta & 0xFF;
                if ((s_meta >> 17) & 1) { // dmr_resample_cach(): the 66 dibits before the sync, against the warm-started thresholds
                    for (int i = lane; i < 66; i += 64) {
                        const int qi = (s_slot + 1 - 90 + i) & (HN - 1);
                        const float v = L.sh[qi][s_c];
                        L.ph[qi][s_c] = (uint8_t)(v > s_cen ? (v > s_um ? 1 : 0) : (v < s_lm ? 3 : 2));
                    }
                }
                if (s_aux < max_sync) {
                    const size_t so = (size_t)(ch0 + s_c) * max_sync + s_aux;
                    if (lane == 0) {
                        sync_pos[so] = s_oo - 1;
                        sync_pat[so] = (uint8_t)((s_fl >> 3) & 31);
                    }
                    if (sync_thr_out) { // the sync entry carries the thresholds after the warm start
                        const float s_mx = __shfl(th.max, sl), s_mn = __shfl(th.min, sl);
                        if (lane < 5) {
                            sync_thr_out[so * 5 + lane] = lane == 0 ? s_cen : (lane == 1 ? s_um : (lane == 2 ? s_lm : (lane == 3 ? s_mx : s_mn)));
                        }
                    }
                    for (int i = lane; i < DDN_FSK4_PRE; i += 64) {
                        const int qi = (s_slot + 1 - DDN_FSK4_PRE + i) & (HN - 1);
                        const bool hv = (DDN_FSK4_PRE - i) <= scount;
                        pre[so * DDN_FSK4_PRE + i] = hv ? L.ph[qi][s_c] : 0;
                        pre_rel[so * DDN_FSK4_PRE + i] = hv ? L.rh[qi][s_c] : 0;
                    }
                }
            }
        }
    };
    // ---- handler mode (recurrence lane) ---------------------------------------------------------------------------------
    const int cfg_max_events = cfgp->max_events;
    auto hctx = [&]() {
        ddn_fsk4h::Ctx x;
        x.hs = &LH.hs[0][ln];
        x.pay = &LH.pay[0][ln];
        x.stride = CPW;
        x.T = htab;
        x.events = events ? events + (size_t)ch * cfg_max_events * 4 : nullptr;
        x.max_events = cfg_max_events;
        return x;
    };
    // the dibit getDibit() hands the handler: sliced against the frame's (frozen) thresholds, polarity undone
    auto hard_dibit = [&](float sym, int neg) {
        const int d = sym > s.center ? (sym > s.umid ? 1 : 0) : (sym < s.lmid ? 3 : 2);
        return neg ? (d ^ 2) : d;
    };
    // one in-frame symbol has been read: keep its dibit for the handler
    auto hsymbol = [&](float sym, int neg) {
        using namespace ddn_fsk4h;
        if (PROTO == 1) {
            if (s.hmode >= M_DATA_SUFFIX && s.hmode != M_SKIP66 && s.hidx < 144) {
                LH.pay[s.hidx][ln] = (uint8_t)hard_dibit(sym, neg);
            }
            s.hidx++;
        } else if (s.hmode == M_NX_LICH) {
            // nxdn_descramble_with_seed(.., 8, 228): the PN9 sequence starts 0 0 1 0 0 1 1 1
            const int dd = hard_dibit(sym, neg) ^ (((0xE4 >> s.hidx) & 1) << 1);
            s.hlich |= ((dd >> 1) & 1) << (7 - s.hidx);
            s.hidx++;
        }
    };
    int pos = 0; // call-relative index of this lane's next sample
    const float* rrow = &L.raw[ln][0];
    const float* frow = &L.flt[ln][0];
    const int n_tiles = (n + TSW - 1) / TSW;
    // Matched-filter cold start.  For the NT - 1 samples after the filter is gated on its output is a FIR over the filter's stale
    // memory followed by the new samples (the always-on stream in `filt` assumes the raw samples before it instead).  Those
    // outputs do not depend on anything the recurrence does, so the whole wavefront computes them - each one the same ordered sum
    // as the reference's filter, one output per lane - straight into the staged LDS rows, tile by tile as the tiles arrive,
    // instead of one lane working through NT taps per sample (in noise a first sync-word match, which gates the NXDN filter on
    // before the second match confirms it, comes along every ~100 symbols: that path was most of the hunting time).
    long long cold_fs = -1; // filt_start these outputs belong to (-1: none; a cold start inherited from the previous call is not)
    int cold_p0 = 0, cold_next = 0, cold_end = 0;
    auto cold_fill = [&](int limit) {
        unsigned long long m = __ballot(live && cold_fs >= 0 && cold_next < (cold_end < limit ? cold_end : limit));
        while (m) {
            const int c = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int p0 = __shfl(cold_p0, c), a = __shfl(cold_next, c), e0 = __shfl(cold_end, c);
            const int e = e0 < limit ? e0 : limit;
            const float* fsrc = fstale + (size_t)(ch0 + c) * (DDN_FSK4_MAX_TAPS - 1);
            const float* rw = raw + (size_t)(ch0 + c) * stride;
            for (int jp = a + lane; jp < e; jp += 64) {
                float acc = 0.0f;
                for (int i = 0; i < NT; i++) {
                    const int idx = jp - (NT - 1) + i; // call-relative; new samples from p0 on, the stale memory before it
                    const float v = idx >= p0 ? rw[idx] : fsrc[(idx - p0) + (NT - 1)];
                    acc += taps[i] * v;
                }
                L.flt[c][jp & RMASKW] = acc;
            }
            if (lane == c) {
                cold_next = e;
            }
        }
    };
    for (int t = 0; t < n_tiles; t++) {
        const int tile_end = (t + 1) * TSW < n ? (t + 1) * TSW : n; // symbols starting before this index belong to this round
        const int lim = (t + 2) * TSW < n ? (t + 2) * TSW : n;       // samples staged so far
        if (loader) {
            stage(t + 2);
            if (t > 0) {
                drain((t - 1) & 1);
            }
        } else {
            int guard = 0, qk = 0;
            int blk_o = -1; // bulk hunting pass: output index at which it left this lane's next symbol to the general trip
            if (live) {
                L.qo[t & 1][ln] = o;
            }
            if (use_flt) {
                cold_fill(lim); // the part of a pending cold start that lies in the tile staged during the previous round
            }
            auto snapshot_filter = [&]() { // the filter's memory at the moment it is gated off
                if (!s.filter_on) {
                    return;
                }
                const long long tnext = abs0 + pos;
                float* fs = fstale + (size_t)ch * (DDN_FSK4_MAX_TAPS - 1);
                for (int k = 0; k < NT - 1; k++) {
                    const long long ja = tnext - (NT - 1) + k;
                    float v;
                    if (ja >= s.filt_start) {
                        const long jc = (long)(ja - abs0);
                        v = (jc >= 0) ? raw[(size_t)ch * stride + jc]
                                      : prev_tail[(size_t)ch * (DDN_FSK4_MAX_TAPS - 1) + (NT - 1) + jc];
                    } else {
                        v = fs[(int)(ja - s.filt_start) + (NT - 1)];
                    }
                    // in-place is safe: entry k is read from index >= k (ja - filt_start + NT-1 >= k  <=>  tnext >= filt_start)
                    fs[k] = v;
                }
            };
            auto hunt_advance = [&]() {
                if (s.hunt_pos < 10200) {
                    s.hunt_pos++;
                } else {
                    s.hunt_pos = 0;
                    snapshot_filter();
                    no_carrier(s);
                    if (Cfg::m17) {
                        s.hlich = 0; // state->m17_polarity
                    }
                    if (HM && PROTO == 1) {
                        ddn_fsk4h::conf_reset(hctx()); // noCarrier(): dmr_confidence_reset(), engine.c:1856
                    }
                }
                if (!(cfg.slow_type && s.lastsync == cfg.slow_type) && s.hunt_pos >= 1800) {
                    snapshot_filter();
                    no_carrier(s);
                    if (Cfg::m17) {
                        s.hlich = 0;
                    }
                    if (HM && PROTO == 1) {
                        ddn_fsk4h::conf_reset(hctx());
                    }
                    hunt_restart(s);
                }
            };
            // handler mode: the phase's last symbol is in - the handler decides (lock_left set anew, or the frame is over)
            auto hphase_end = [&]() {
                using namespace ddn_fsk4h;
                int mode = s.hmode, next = 0;
                bool on;
                if (PROTO == 1) {
                    const Ctx x = hctx();
                    on = dmr_phase_end(x, o, mode, next);
                    if (on && mode == M_BURST_CACH) {
                        s.hidx = 0;
                    }
                } else {
                    on = false;
                    if (mode == M_NX_LICH) {
                        int lich7, par_ok;
                        on = nxdn_lich_ok(s.hlich, lich7, par_ok);
                        const Ctx x = hctx();
                        x.ev(o, EV_NXDN_LICH, on ? 1 : 0, lich7, par_ok);
                        if (on) {
                            mode = M_NX_REST;
                            next = 174;
                        } else { // nxdn_mark_bad_sync(): lastsynctype NONE - the NXDN filter gate and the two-match rule see it
                            snapshot_filter();
                            s.filter_on = 0;
                            s.lastsync = 0;
                        }
                    }
                }
                if (on) {
                    s.hmode = mode;
                    s.lock_left = next;
                } else {
                    s.hmode = M_IDLE;
                    hunt_restart(s);
                }
            };
            while (true) {
                // ---- bulk hunting pass ---------------------------------------------------------------------------------------
                // As in the P25 loop (ddn_rx.hip): a lane that hunts has its thresholds parked, so its recurrence is the symbol
                // start, the latched crossing and the sign history - its symbols that start in this tile are found at once by the
                // whole wavefront (crossing mask by ballots, the start -> crossing -> slip chain on scalars, lane j = symbol j:
                // window sum, sign, history word, the compare with every sync pattern) and handed to the helper wave through the
                // queue like any other symbol.  The pass stops in front of the symbol that matches a pattern (the general trip
                // takes that one: level window, filter gate, confirm rule, warm start, handler) and at the eighth symbol of a hunt
                // (whose commit moves the crossing limits to the parked max / min: the mask is taken again).  Without it a lane
                // that hunts drags its wave's in-frame lanes through general trips (54 % of the trips on the DMR capture).
                if (bulk_ok) {
                    const bool fo_b = s.filter_on != 0;
                    const bool be = live & (s.have_sync == 0) & (s.in_symbol == 0) & (s.need_reset == 0) & (pos < tile_end) & (blk_o != o)
                                    & (s.hunt_pos + 20 < 1800) & (qk + 4 < QCAPW) & (pos + whole + 1 <= n)
                                    & (!fo_b | ((abs0 + pos - s.filt_start) >= (long long)(NT - 1)) | (cold_fs == s.filt_start));
                    unsigned long long bm = __ballot(be);
                    // ---- (round 5) the same pass for every hunting lane of the wave at once ------------------------------------------
                    // Several lanes that hunt (idle channels; a group of channels whose bursts end together) used to take the pass one
                    // after the other, each with the whole wavefront: four lanes = four times ~8 k cycles per tile.  With two or four
                    // channels per wavefront a lane owns a ROW of 64 / CPW >= 16 lanes, a pass handles at most 16 symbols, so every owner's
                    // pass fits its own row: the owner's words are broadcast to its row, the crossing mask is taken 64 / CPW samples per
                    // owner and ballot, the slip chain runs per row on the vector unit (its words are uniform inside a row), lane l of a
                    // row is symbol l of that row's owner.  Same results lane for lane; DDN_RX4_DBG bit 65536 keeps the serial order.
                    if constexpr (CPW == 2 || CPW == 4) {
                    if (__popcll(bm) >= 2 && !(cfg.dbg & 65536)) {
                        constexpr int OW = 64 / CPW;
                        constexpr unsigned long long GM = OW == 32 ? 0xFFFFFFFFull : 0xFFFFull;
                        const int g = lane / OW, l = lane % OW;
                        const bool gact = (bm >> g) & 1ull;
                        const int sp0 = __shfl(pos, g), c0 = __shfl(s.hist_count, g), flt_o = __shfl(s.filter_on, g);
                        int jit = __shfl(s.jitter, g);
                        const uint32_t h0 = (uint32_t)__shfl((int)s.hist_bits, g);
                        const int sh_o = __shfl(s.shead, g), li_o = __shfl(s.lidx, g), qk_o = __shfl(qk, g), ls_type = __shfl(s.lastsync, g);
                        const int m17_pol_o = Cfg::m17 ? __shfl(s.hlich, g) : 0;
                        const float cen_o = __shfl(s.center, g), ls_o = __shfl(s.lastsample, g);
                        const float um_o = __shfl(s.umid, g), lm_o = __shfl(s.lmid, g), mx_o = __shfl(s.max, g), mn_o = __shfl(s.min, g);
                        float hl = __shfl(s.maxref * 1.25f, g), ll = __shfl(s.minref * 1.25f, g);
                        const float* pr = flt_o ? &L.flt[g][0] : &L.raw[g][0];
                        const int a_end = lim < n ? lim : n; // samples staged
                        int q = sp0, m = 0, myq = 0, myi0 = 0, myjin = 0;
                        int cap = QCAPW - 2 - qk_o;
                        cap = cap > (OW < 17 ? OW - 1 : 16) ? (OW < 17 ? OW - 1 : 16) : cap; // (lane `m` of the row holds the start after the pass)
                        cap = cap > cfg.t_max ? cfg.t_max : cap;
                        const bool refs_stale = __shfl((int)((s.maxref != s.max) | (s.minref != s.min)), g) != 0;
                        int lm = c0 < 8 ? 8 - c0 : (refs_stale ? 1 : cap);
                        lm = lm > cap ? cap : lm;
                        bool gdone = !gact; // this row's chain has ended
                        for (int ph = 0; ph < 2; ph++) {
                            unsigned long long cm0 = 0ull, cm1 = 0ull, cm2 = 0ull;
#pragma unroll
                            for (int r = 0; r < 192 / OW; r++) {
                                const int a = q + l + OW * r;
                                bool hit = false;
                                if (!gdone && a < a_end) {
                                    const float x = pr[a & RMASKW];
                                    const float xp = (a == sp0) ? ls_o : pr[(a - 1) & RMASKW];
                                    const bool up = x > cen_o;
                                    const bool within = up ? !(x > hl) : !(x < ll);
                                    const bool crossed = up ? (xp < cen_o) : (xp > cen_o);
                                    hit = within && crossed;
                                }
                                const unsigned long long bits = (__ballot(hit) >> (g * OW)) & GM;
                                constexpr int PW = 64 / OW; // rounds per 64-bit word
                                const unsigned long long put = bits << (OW * (r % PW));
                                if (r / PW == 0) {
                                    cm0 |= put;
                                } else if (r / PW == 1) {
                                    cm1 |= put;
                                } else {
                                    cm2 |= put;
                                }
                            }
                            bool full = false;
                            while (true) {
                                const bool go = !gdone && !full && m < lm;
                                if (!__any(go)) {
                                    break;
                                }
                                if (go) {
                                    const int i0 = (int)((i0lut >> (2 * (jit + 1))) & 3ull) - 1;
                                    const int cnt = whole - i0;
                                    if (q >= tile_end || q + cnt > n) {
                                        full = true;
                                    } else {
                                        if (l == m) {
                                            myq = q;
                                            myi0 = i0;
                                            myjin = jit;
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
                            if (!gdone) {
                                if (full || lm >= cap) {
                                    gdone = true;
                                } else {
                                    lm = cap;
                                    hl = mx_o * 1.25f;
                                    ll = mn_o * 1.25f;
                                }
                            }
                            if (!__any(!gdone)) {
                                break;
                            }
                        }
                        if (l == m) { // where the symbol after the pass starts, and the latch it starts with
                            myq = q;
                            myjin = jit;
                        }
                        float sym = 0.0f, wsum = 0.0f;
                        int wc = 0;
                        if (gact && l < m) {
                            const int cw = (whole - 1) / 2;
                            const int l_e = (Cfg::dmr_window && ls_type != 0) ? 1 : 2;
                            const bool rf0b = cfg.rf_mod == 0;
                            const int wlo = rf0b ? cw - l_e : cw - 1, whi = rf0b ? cw + 2 : cw + 1;
                            const bool has20 = whole == 20;
                            const int i_lo = has20 && 7 < wlo ? 7 : wlo, i_hi = has20 && 13 > whi ? 13 : whi;
                            const int leftj = whole - myi0;
#pragma unroll
                            for (int qq = 0; qq < 12; qq++) {
                                const int i = i_lo + qq, k = i - myi0;
                                if (i <= i_hi && k >= 0 && k < leftj) {
                                    const float x = pr[(myq + k) & RMASKW];
                                    const bool k1 = rf0b ? (i >= wlo && i <= whi) : (i == wlo || i == whi);
                                    const bool k2 = has20 && i >= 7 && i <= 13;
                                    if (k2) {
                                        wsum += x;
                                    }
                                    if (k1) {
                                        wsum += x;
                                    }
                                    wc += (k1 ? 1 : 0) + (k2 ? 1 : 0);
                                }
                            }
                            sym = (wc > 0) ? (wsum / (float)wc) : 0.0f;
                        }
                        const uint32_t sw = (uint32_t)((__ballot(gact && l < m && sym > 0.0f) >> (g * OW)) & GM);
                        const int lj = l < 31 ? l : 31;
                        const uint32_t hj = ((h0 << (lj + 1)) | __brev(sw << (31 - lj))) & 0xFFFFFFu;
                        bool syn = false;
                        if (gact && l < m && (c0 + l + 1 >= cfg.win_len) && !(cfg.dbg & 8)) {
                            const uint32_t w = hj & wmask;
                            if (Cfg::m17) {
                                int pa;
                                syn = m17_hit(w, ls_type, m17_pol_o, L.pat_meta, pa) >= 0;
                            } else {
                                for (int k = 0; k < cfg.n_pat; k++) {
                                    syn |= (w == L.pat_bits[k]);
                                }
                            }
                        }
                        const uint32_t smg = (uint32_t)((__ballot(syn) >> (g * OW)) & GM);
                        if (smg) {
                            m = __ffs((int)smg) - 1;
                        }
                        // the owner lanes (lane = channel column < CPW) read their row's results
                        const int orow = (lane < CPW ? lane : 0) * OW;
                        const int m_own = __shfl(m, orow);
                        const int qf = __shfl(myq, orow + m_own), jf = __shfl(myjin, orow + m_own);
                        const int src1 = orow + (m_own > 0 ? m_own - 1 : 0);
                        const float sumf = __shfl(wsum, src1);
                        const int cntf = __shfl(wc, src1);
                        const uint32_t hf = (uint32_t)__shfl((int)hj, src1);
                        if (gact && l < m) { // symbol history, level window, the queue entry the helper wave slices and stores
                            const int slot = (sh_o + l) & (HN - 1);
                            L.sh[slot][g] = sym;
                            int k = li_o + l;
                            k = k >= cfg.t_max ? k - cfg.t_max : k;
                            L.lb[k][g] = sym;
                            const int qb = t & 1, qe = qk_o + l;
                            L.q[qb][qe][0][g] = sym;
                            L.q[qb][qe][1][g] = cen_o;
                            L.q[qb][qe][2][g] = um_o;
                            L.q[qb][qe][3][g] = lm_o;
                            L.q[qb][qe][4][g] = mx_o;
                            L.q[qb][qe][5][g] = mn_o;
                            L.q[qb][qe][6][g] = __int_as_float(slot << 8);
                        }
                        if (lane < CPW && be) {
                            if (m_own == 0) { // the next symbol matches a sync pattern: the general trip's
                                blk_o = o;
                            } else {
                                const float* prow = s.filter_on ? &L.flt[lane][0] : &L.raw[lane][0];
                                const float lsf = prow[(qf - 1) & RMASKW];
                                s.shead = (s.shead + m_own) & (HN - 1);
                                s.scount = s.scount + m_own < HN ? s.scount + m_own : HN;
                                int k = s.lidx + m_own;
                                s.lidx = k >= cfg.t_max ? k - cfg.t_max : k;
                                s.level_count = s.level_count + m_own < cfg.t_max ? s.level_count + m_own : cfg.t_max;
                                s.hist_count = s.hist_count + m_own < 24 ? s.hist_count + m_own : 24;
                                s.hist_bits = hf;
                                if (s.hist_count >= 8) {
                                    s.maxref = s.max;
                                    s.minref = s.min;
                                }
                                s.hunt_pos += m_own;
                                s.span = whole;
                                s.centre = (whole - 1) / 2;
                                s.i = whole;
                                s.sum = sumf;
                                s.count = cntf;
                                s.in_symbol = 0;
                                s.jitter = jf;
                                s.lastsample = lsf;
                                pos = qf;
                                o += m_own;
                                qk += m_own;
                            }
                        }
                        continue;
                    }
                    }
                    if (__builtin_expect(bm != 0, 0)) {
                        while (bm) {
                            const int ow = __ffsll((long long)bm) - 1; // owner lane (= channel column) of this pass
                            bm &= bm - 1;
                            const int sp0 = __builtin_amdgcn_readlane(pos, ow);
                            const int c0 = __builtin_amdgcn_readlane(s.hist_count, ow);
                            const int flt_o = __builtin_amdgcn_readlane(s.filter_on, ow);
                            int jit = __builtin_amdgcn_readlane(s.jitter, ow);
                            const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)s.hist_bits, ow);
                            const int sh_o = __builtin_amdgcn_readlane(s.shead, ow), li_o = __builtin_amdgcn_readlane(s.lidx, ow);
                            const int qk_o = __builtin_amdgcn_readlane(qk, ow), ls_type = __builtin_amdgcn_readlane(s.lastsync, ow);
                            const int m17_pol_o = Cfg::m17 ? __builtin_amdgcn_readlane(s.hlich, ow) : 0; // (M17 keeps its polarity in hlich)
                            const float cen_o = __shfl(s.center, ow), ls_o = __shfl(s.lastsample, ow);
                            const float um_o = __shfl(s.umid, ow), lm_o = __shfl(s.lmid, ow), mx_o = __shfl(s.max, ow), mn_o = __shfl(s.min, ow);
                            float hl = __shfl(s.maxref * 1.25f, ow), ll = __shfl(s.minref * 1.25f, ow);
                            const float* pr = flt_o ? &L.flt[ow][0] : &L.raw[ow][0];
                            const int a_end = lim < n ? lim : n; // samples staged
                            int q = sp0, m = 0, myq = 0, myi0 = 0, myjin = 0;
                            int cap = QCAPW - 2 - qk_o;
                            cap = cap > 16 ? 16 : cap;
                            cap = cap > cfg.t_max ? cfg.t_max : cap;
                            // crossing limits: a symbol is searched against maxref / minref as the commit before it left them -
                            // from the eighth symbol of a hunt on that is the parked max / min, except right after a pattern match
                            // that was not accepted (it moved max / min behind the commit's back): that one symbol stands alone
                            const bool refs_stale = __shfl((int)((s.maxref != s.max) | (s.minref != s.min)), ow) != 0;
                            int lm = c0 < 8 ? 8 - c0 : (refs_stale ? 1 : cap);
                            lm = lm > cap ? cap : lm;
                            for (int ph = 0; ph < 2; ph++) {
                                unsigned long long cm[3];
#pragma unroll
                                for (int r = 0; r < 3; r++) {
                                    cm[r] = 0ull;
                                    if (q + 64 * r < a_end) {
                                        const int a = q + lane + 64 * r;
                                        bool hit = false;
                                        if (a < a_end) {
                                            const float x = pr[a & RMASKW];
                                            const float xp = (a == sp0) ? ls_o : pr[(a - 1) & RMASKW];
                                            const bool up = x > cen_o;
                                            const bool within = up ? !(x > hl) : !(x < ll);
                                            const bool crossed = up ? (xp < cen_o) : (xp > cen_o);
                                            hit = within && crossed;
                                        }
                                        cm[r] = __ballot(hit);
                                    }
                                }
                                bool full = false;
                                while (m < lm) {
                                    const int i0 = (int)((i0lut >> (2 * (jit + 1))) & 3ull) - 1;
                                    const int cnt = whole - i0;
                                    if (q >= tile_end || q + cnt > n) {
                                        full = true;
                                        break;
                                    }
                                    if (lane == m) {
                                        myq = q;
                                        myi0 = i0;
                                        myjin = jit;
                                    }
                                    const int k0 = i0 < 0 ? 1 : 0; // a crossing at symbol index -1 latches nothing
                                    const uint32_t wv = (uint32_t)(cm[0] >> k0) & ((1u << (cnt - k0)) - 1u);
                                    jit = wv ? i0 + k0 + (__ffs((int)wv) - 1) : -1;
                                    cm[0] = (cm[0] >> cnt) | (cm[1] << (64 - cnt));
                                    cm[1] = (cm[1] >> cnt) | (cm[2] << (64 - cnt));
                                    cm[2] >>= cnt;
                                    q += cnt;
                                    m++;
                                }
                                if (full || lm >= cap) {
                                    break;
                                }
                                lm = cap;
                                hl = mx_o * 1.25f;
                                ll = mn_o * 1.25f;
                            }
                            if (lane == m) { // where the symbol after the pass starts, and the latch it starts with
                                myq = q;
                                myjin = jit;
                            }
                            // lane j: the window sum of symbol j (symbol_accumulate_sample's rule, in sample order; no clip while hunting)
                            float sym = 0.0f, wsum = 0.0f;
                            int wc = 0;
                            if (lane < m) {
                                const int cw = (whole - 1) / 2;
                                const int l_e = (Cfg::dmr_window && ls_type != 0) ? 1 : 2;
                                const bool rf0b = cfg.rf_mod == 0;
                                const int wlo = rf0b ? cw - l_e : cw - 1, whi = rf0b ? cw + 2 : cw + 1;
                                const bool has20 = whole == 20;
                                const int i_lo = has20 && 7 < wlo ? 7 : wlo, i_hi = has20 && 13 > whi ? 13 : whi;
                                const int leftj = whole - myi0;
#pragma unroll
                                for (int qq = 0; qq < 12; qq++) {
                                    const int i = i_lo + qq, k = i - myi0;
                                    if (i <= i_hi && k >= 0 && k < leftj) {
                                        const float x = pr[(myq + k) & RMASKW];
                                        const bool k1 = rf0b ? (i >= wlo && i <= whi) : (i == wlo || i == whi);
                                        const bool k2 = has20 && i >= 7 && i <= 13;
                                        if (k2) {
                                            wsum += x;
                                        }
                                        if (k1) {
                                            wsum += x;
                                        }
                                        wc += (k1 ? 1 : 0) + (k2 ? 1 : 0);
                                    }
                                }
                                sym = (wc > 0) ? (wsum / (float)wc) : 0.0f;
                            }
                            const uint32_t sw = (uint32_t)__ballot(lane < m && sym > 0.0f);
                            const int lj = lane < 31 ? lane : 31;
                            const uint32_t hj = ((h0 << (lj + 1)) | __brev(sw << (31 - lj))) & 0xFFFFFFu;
                            bool syn = false;
                            if (lane < m && (c0 + lane + 1 >= cfg.win_len) && !(cfg.dbg & 8)) {
                                const uint32_t w = hj & wmask;
                                if (Cfg::m17) { // (the owner's sync type and polarity stand still while it hunts)
                                    int pa;
                                    syn = m17_hit(w, ls_type, m17_pol_o, L.pat_meta, pa) >= 0;
                                } else {
                                    for (int k = 0; k < cfg.n_pat; k++) {
                                        syn |= (w == L.pat_bits[k]);
                                    }
                                }
                            }
                            const unsigned long long sm = __ballot(syn);
                            if (sm) {
                                m = __ffsll((long long)sm) - 1;
                            }
                            if (m == 0) { // the next symbol matches a sync pattern: the general trip's
                                if (lane == ow) {
                                    blk_o = o;
                                }
                                continue;
                            }
                            const int qf = __builtin_amdgcn_readlane(myq, m), jf = __builtin_amdgcn_readlane(myjin, m);
                            const float sumf = __shfl(wsum, m - 1);
                            const int cntf = __builtin_amdgcn_readlane(wc, m - 1);
                            const uint32_t hf = (uint32_t)__builtin_amdgcn_readlane((int)hj, m - 1);
                            if (lane < m) { // symbol history, level window, the queue entry the helper wave slices and stores
                                const int slot = (sh_o + lane) & (HN - 1);
                                L.sh[slot][ow] = sym;
                                int k = li_o + lane;
                                k = k >= cfg.t_max ? k - cfg.t_max : k;
                                L.lb[k][ow] = sym;
                                const int qb = t & 1, qe = qk_o + lane;
                                L.q[qb][qe][0][ow] = sym;
                                L.q[qb][qe][1][ow] = cen_o;
                                L.q[qb][qe][2][ow] = um_o;
                                L.q[qb][qe][3][ow] = lm_o;
                                L.q[qb][qe][4][ow] = mx_o;
                                L.q[qb][qe][5][ow] = mn_o;
                                L.q[qb][qe][6][ow] = __int_as_float(slot << 8);
                            }
                            const float lsf = pr[(qf - 1) & RMASKW];
                            if (lane == ow) {
                                s.shead = (s.shead + m) & (HN - 1);
                                s.scount = s.scount + m < HN ? s.scount + m : HN;
                                int k = s.lidx + m;
                                s.lidx = k >= cfg.t_max ? k - cfg.t_max : k;
                                s.level_count = s.level_count + m < cfg.t_max ? s.level_count + m : cfg.t_max;
                                s.hist_bits = hf;
                                s.hist_count = c0 + m < 24 ? c0 + m : 24;
                                if (s.hist_count >= 8) {
                                    s.maxref = s.max;
                                    s.minref = s.min;
                                }
                                s.hunt_pos += m;
                                s.span = whole;
                                s.centre = (whole - 1) / 2;
                                s.i = whole;
                                s.sum = sumf;
                                s.count = cntf;
                                s.in_symbol = 0;
                                s.jitter = jf;
                                s.lastsample = lsf;
                                pos = qf;
                                o += m;
                                qk += m;
                            }
                        }
                        continue;
                    }
                }
                // ---- bulk in-frame pass ------------------------------------------------------------------------------------
                // Inside a frame these protocols leave the thresholds alone and the crossing search is off (see the lean trip below),
                // so the symbols of a lane up to the end of its tile - or of its phase, less the phase's last symbol, whose commit
                // belongs to the handler - do not depend on one another at all: symbol j starts at pos + j * whole.  The wavefront
                // takes them at once, lane j = symbol j (window sum in sample order, history ring, queue entry, the handler's dibit),
                // and the owner lane's counters move by K.  Without it every in-frame symbol is one trip of the whole loop body
                // (one lane per wave at 1365 DMR channels: 3.1 k cycles per symbol).
                if (bulk_ok && !(cfg.dbg & 16384)) {
                    const bool fo_i = s.filter_on != 0;
                    const bool warm_i = !fo_i | ((abs0 + pos - s.filt_start) >= (long long)(NT - 1)) | (cold_fs == s.filt_start);
                    bool ie = live & (s.in_symbol == 0) & (s.have_sync != 0) & (s.jitter >= 0) & (s.lock_left >= 3) & (s.need_reset == 0)
                              & (pos < tile_end) & warm_i;
                    if (HM && PROTO != 1) {
                        ie = ie & (s.hmode != ddn_fsk4h::M_NX_LICH); // (the LICH symbols feed a word of the recurrence lane)
                    }
                    int kfit = 0;
                    if (ie) {
                        const int a = (tile_end - pos + whole - 1) / whole; // symbols that start in this tile
                        const int b = (lim - pos) / whole;                 // ... with every sample staged
                        kfit = a < b ? a : b;
                        kfit = kfit < s.lock_left - 1 ? kfit : s.lock_left - 1;
                        kfit = kfit < QCAPW - 2 - qk ? kfit : QCAPW - 2 - qk;
                        kfit = kfit < 32 ? kfit : 32;
                    }
                    // lane -> (owner = channel column, symbol index): 64 / CPW lanes per owner, all owners at once
                    constexpr int G = CPW >= 64 ? 1 : 64 / CPW;
                    if (ie) {
                        kfit = kfit < G ? kfit : G;
                    }
                    const unsigned long long im = __ballot(ie && kfit >= 2);
                    if (im != 0) {
                        const int ow = lane / G, jj = lane % G;
                        const bool ow_ok = ow < CPW && ((im >> ow) & 1ull) != 0;
                        const int k_ow = __shfl(kfit, ow); // (every lane takes part in the shuffles: the owners' lanes are sources)
                        const int K = ow_ok ? k_ow : 0;
                        const int p0 = __shfl(pos, ow), flt_o = __shfl(s.filter_on, ow), sh_o = __shfl(s.shead, ow);
                        const int qk_o = __shfl(qk, ow), ls_type = __shfl(s.lastsync, ow), pat_o = __shfl(s.cur_pat, ow);
                        const float cen_o = __shfl(s.center, ow), um_o = __shfl(s.umid, ow), lm_o = __shfl(s.lmid, ow);
                        const float mx_o = __shfl(s.max, ow), mn_o = __shfl(s.min, ow);
                        int hm_o = 0, hidx_o = 0;
                        if (HM && PROTO == 1) {
                            hm_o = __shfl(s.hmode, ow);
                            hidx_o = __shfl(s.hidx, ow);
                        }
                        if (jj < K) {
                            const int owc = ow < CPW ? ow : 0;
                            const int neg = (cfg.dbg & 32) ? 0 : (int)((L.pat_meta[pat_o] >> 8) & 1);
                            const float* pr = flt_o ? &L.flt[owc][0] : &L.raw[owc][0];
                            const int p = p0 + jj * whole;
                            const int cw = (whole - 1) / 2;
                            const int l_e = (Cfg::dmr_window && ls_type != 0) ? 1 : 2;
                            const bool rf0l = cfg.rf_mod == 0;
                            const int wlo = rf0l ? cw - l_e : cw - 1, whi = rf0l ? cw + 2 : cw + 1;
                            const bool has20 = whole == 20;
                            const int i_lo = has20 && 7 < wlo ? 7 : wlo, i_hi = has20 && 13 > whi ? 13 : whi;
                            float sum = 0.0f;
                            int c = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                const int i = i_lo + k;
                                if (i <= i_hi) {
                                    float x = pr[(p + i) & RMASKW];
                                    if (rf0l) { // the sync-time clip (C4FM rules only)
                                        x = x > mx_o ? mx_o : (x < mn_o ? mn_o : x);
                                    }
                                    const bool k1 = rf0l ? (i >= wlo && i <= whi) : (i == wlo || i == whi);
                                    const bool k2 = has20 && i >= 7 && i <= 13;
                                    if (k2) {
                                        sum += x;
                                    }
                                    if (k1) {
                                        sum += x;
                                    }
                                    c += (k1 ? 1 : 0) + (k2 ? 1 : 0);
                                }
                            }
                            const float sym = (c > 0) ? (sum / (float)c) : 0.0f;
                            const int slot = (sh_o + jj) & (HN - 1);
                            L.sh[slot][owc] = sym;
                            const int qb = t & 1, qe = qk_o + jj;
                            L.q[qb][qe][0][owc] = sym;
                            L.q[qb][qe][1][owc] = cen_o;
                            L.q[qb][qe][2][owc] = um_o;
                            L.q[qb][qe][3][owc] = lm_o;
                            L.q[qb][qe][4][owc] = mx_o;
                            L.q[qb][qe][5][owc] = mn_o;
                            L.q[qb][qe][6][owc] = __int_as_float((1 | (neg ? 4 : 0)) | (slot << 8));
                            if (HM && PROTO == 1) {
                                using namespace ddn_fsk4h;
                                if (hm_o >= M_DATA_SUFFIX && hm_o != M_SKIP66 && hidx_o + jj < 144) {
                                    const int d = sym > cen_o ? (sym > um_o ? 1 : 0) : (sym < lm_o ? 3 : 2);
                                    LH.pay[hidx_o + jj][owc] = (uint8_t)(neg ? (d ^ 2) : d);
                                }
                            }
                        }
                        if (ie && kfit >= 2) { // the owners move their counters
                            const int K2 = kfit;
                            pos += K2 * whole;
                            s.shead = (s.shead + K2) & (HN - 1);
                            s.scount = s.scount + K2 < HN ? s.scount + K2 : HN;
                            qk += K2;
                            if (HM && PROTO == 1) {
                                s.hidx += K2;
                            }
                            s.lock_left -= K2;
                            s.maxref = s.max;
                            s.minref = s.min;
                            o += K2;
                        }
                        continue;
                    }
                }
                // ---- lean trip -----------------------------------------------------------------------------------------
                // Every live lane sits inside a frame with its crossing latched, a whole fresh symbol of the fixed length
                // staged and the matched filter warm (or has used up its tile): inside a
                // frame these protocols leave the thresholds alone (use_symbol()'s "no continuous update" branch,
                // dsd_dibit.c:264-276), the crossing search is off and nothing reads the last sample before the frame's last symbol, so
                // the symbol is the clipped window sum over its count, pushed to the history ring and the queue - none of the
                // per-sample pass, the start-up or the hunting commit below is on the wave's instruction stream.
                if (lean_ok) {
                    const bool fo_l = s.filter_on != 0;
                    const bool lean = live & (s.in_symbol == 0) & (s.have_sync != 0) & (s.jitter >= 0) & (s.lock_left >= 1)
                                      & (s.need_reset == 0) & (pos < tile_end) & (pos + whole <= lim) & (qk < QCAPW - 2)
                                      & (!fo_l | ((abs0 + pos - s.filt_start) >= (long long)(NT - 1)) | (cold_fs == s.filt_start));
                    const bool idle = live & (s.in_symbol == 0) & !(pos < tile_end);
                    if (!__any(live & !(lean | idle)) && __any(lean)) {
                        if (lean) {
                            const float* rowl = fo_l ? frow : rrow;
                            const int cw = (whole - 1) / 2;
                            const int l_e = (Cfg::dmr_window && s.lastsync != 0) ? 1 : 2;
                            const bool rf0l = cfg.rf_mod == 0;
                            const int wlo = rf0l ? cw - l_e : cw - 1, whi = rf0l ? cw + 2 : cw + 1;
                            const bool has20 = whole == 20;
                            const int i_lo = has20 && 7 < wlo ? 7 : wlo, i_hi = has20 && 13 > whi ? 13 : whi;
                            float sum = 0.0f;
                            int c = 0;
                            // (at most 8 samples: 7..13 of a 20-sample symbol, or the window of 2..5)
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                const int i = i_lo + k;
                                if (i <= i_hi) {
                                    float x = rowl[(pos + i) & RMASKW];
                                    if (rf0l) { // the sync-time clip (C4FM rules only)
                                        x = x > s.max ? s.max : (x < s.min ? s.min : x);
                                    }
                                    const bool k1 = rf0l ? (i >= wlo && i <= whi) : (i == wlo || i == whi);
                                    const bool k2 = has20 && i >= 7 && i <= 13;
                                    if (k2) {
                                        sum += x;
                                    }
                                    if (k1) {
                                        sum += x;
                                    }
                                    c += (k1 ? 1 : 0) + (k2 ? 1 : 0);
                                }
                            }
                            const float sym = (c > 0) ? (sum / (float)c) : 0.0f;
                            if (s.lock_left == 1) { // the frame's last symbol: the hunt that follows starts from its last sample
                                float x = rowl[(pos + whole - 1) & RMASKW];
                                if (rf0l) {
                                    x = x > s.max ? s.max : (x < s.min ? s.min : x);
                                }
                                s.lastsample = x;
                            }
                            pos += whole;
                            const int slot = s.shead;
                            L.sh[slot][ln] = sym;
                            s.shead = (s.shead + 1 >= HN) ? 0 : s.shead + 1;
                            s.scount = s.scount < HN ? s.scount + 1 : HN;
                            const int neg = (cfg.dbg & 32) ? 0 : ((L.pat_meta[s.cur_pat] >> 8) & 1);
                            const int qb = t & 1;
                            L.q[qb][qk][0][ln] = sym;
                            L.q[qb][qk][1][ln] = s.center;
                            L.q[qb][qk][2][ln] = s.umid;
                            L.q[qb][qk][3][ln] = s.lmid;
                            L.q[qb][qk][4][ln] = s.max;
                            L.q[qb][qk][5][ln] = s.min;
                            L.q[qb][qk][6][ln] = __int_as_float((1 | (neg ? 4 : 0)) | (slot << 8));
                            qk++;
                            s.maxref = s.max;
                            s.minref = s.min;
                            if (HM) {
                                hsymbol(sym, neg);
                            }
                            if (--s.lock_left <= 0) {
                                if (HM) {
                                    hphase_end();
                                } else {
                                    if (Cfg::m17 && (s.cur_pat == 2 || s.cur_pat == 3)) { // dsd_dispatch_handle_m17(): EOT ends the transmission
                                        s.lastsync = 0;
                                        s.hlich = 0;
                                    }
                                    hunt_restart(s);
                                }
                            }
                            o++;
                        }
                        continue;
                    }
                }
                bool began = false;
                if (live && pos < tile_end && !s.in_symbol) {
                    began = true;
                    if (s.need_reset) {
                        timing_reset(s);
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
                        const int j = s.jitter, c = s.centre;
                        if (sps == 20) {
                            s.i += (j >= 7 && j <= 10) ? -1 : ((j >= 11 && j <= 14) ? 1 : 0);
                        } else if (cfg.rf_mod == 2) {
                            s.i += (j >= c - 1 && j <= c) ? -1 : ((j >= c + 1 && j <= c + 2) ? 1 : 0);
                        } else {
                            s.i += (j > 0 && j <= c) ? -1 : ((j > c && j < sps) ? 1 : 0);
                        }
                        s.jitter = -1;
                    }
                }
                const int l_edge = (cfg.dmr_window && s.lastsync != 0) ? 1 : 2;
                const bool clip = s.have_sync && cfg.rf_mod == 0;
                // per-symbol constants: which staged row feeds the symbol, the window as integer bounds
                const bool fo_now = s.filter_on != 0;
                const bool steady = !fo_now || (cfg.dbg & 128) || (abs0 + pos - s.filt_start) >= (long long)(NT - 1)
                                    || cold_fs == s.filt_start;
                const float* rowp = fo_now ? frow : rrow;
                const bool rf0 = cfg.rf_mod == 0;
                const bool lean = s.span >= 6;
                const int cw = s.centre;
                const int wlo = rf0 ? cw - l_edge : cw - 1, whi = rf0 ? cw + 2 : cw + 1;
                const bool two = !rf0; // GFSK: only the two edge samples wlo and whi
                const bool has20 = s.span == 20;
                const float up_lim = s.maxref * 1.25f, dn_lim = s.minref * 1.25f;
                const int left = s.span - s.i; // samples this symbol still needs
                const bool whole_ok = live && began && lean && steady && pos + left <= n && left <= MAXW && !(cfg.dbg & 256);
                if (__any(whole_ok)) {
                    // Whole-symbol pass (hunting, or in frame before the latch is set) - the arithmetic of the sample-at-a-time loop
                    // below, laid out over the wave: the crossing test of sample k reads x[k], x[k - 1] and values that are fixed
                    // for the whole symbol, so the 64 / CPW lanes that share a channel's column (lane = channel + CPW * slot) each
                    // test the samples k = slot, slot + 64 / CPW, ... and the owner takes the lowest set bit of the ballots: the
                    // first crossing, as the in-order search latches it.  The window sums stay on the owner lane, in sample order.
                    constexpr int EPL = 64 / CPW;
                    const int oc = lane % CPW, slot = lane / CPW;
                    const int i0 = s.i;
                    const int need_o = __shfl((int)(whole_ok && s.jitter < 0), oc);
                    const int left_o = __shfl(left, oc), pos_o = __shfl(pos, oc), i0_o = __shfl(i0, oc);
                    const int flt_o = __shfl((int)fo_now, oc), clip_o = __shfl((int)clip, oc);
                    const float cen_o = __shfl(s.center, oc), ul_o = __shfl(up_lim, oc), dl_o = __shfl(dn_lim, oc);
                    const float mx_o = __shfl(s.max, oc), mn_o = __shfl(s.min, oc), ls_o = __shfl(s.lastsample, oc);
                    const float* po = flt_o ? &L.flt[oc][0] : &L.raw[oc][0];
                    int found = -1;
                    if (__any(need_o != 0)) {
#pragma unroll
                        for (int r = 0; r < (MAXW + EPL - 1) / EPL; r++) {
                            const int k = slot + r * EPL;
                            bool hit = false;
                            if (need_o && k < left_o) {
                                float x = po[(pos_o + k) & RMASKW];
                                float xp = k == 0 ? ls_o : po[(pos_o + k - 1) & RMASKW];
                                if (clip_o) {
                                    x = x > mx_o ? mx_o : (x < mn_o ? mn_o : x);
                                    if (k > 0) { // (the symbol's first sample compares with lastsample as it was stored)
                                        xp = xp > mx_o ? mx_o : (xp < mn_o ? mn_o : xp);
                                    }
                                }
                                const bool up = x > cen_o;
                                const bool within = up ? !(x > ul_o) : !(x < dl_o);
                                const bool crossed = up ? (xp < cen_o) : (xp > cen_o);
                                // a crossing at symbol index -1 (sample 0 of a symbol that slipped early) stores -1: nothing latched
                                hit = within && crossed && (i0_o + k >= 0);
                            }
                            unsigned long long col = __ballot(hit) >> oc; // this channel's column: bits oc, oc + CPW, ...
                            unsigned long long m = 0;
#pragma unroll
                            for (int e = 0; e < EPL; e++) {
                                m |= 1ull << (e * CPW);
                            }
                            col &= m;
                            if (found < 0 && col != 0) {
                                found = (__ffsll((long long)col) - 1) / CPW + r * EPL;
                            }
                        }
                    }
                    if (whole_ok) {
                        // window sums in sample order: index i of the symbol adds once when it lies in 7..13 of a 20-sample symbol,
                        // once more when it lies in the modulation's window (symbol_accumulate_sample)
                        const int i_lo = has20 && 7 < wlo ? 7 : wlo, i_hi = has20 && 13 > whi ? 13 : whi;
                        float sum = 0.0f;
                        int c = 0;
#pragma unroll
                        for (int q = 0; q < 12; q++) {
                            const int i = i_lo + q, k = i - i0;
                            if (i <= i_hi && k >= 0 && k < left) {
                                float x = rowp[(pos + k) & RMASKW];
                                if (clip) {
                                    x = x > s.max ? s.max : (x < s.min ? s.min : x);
                                }
                                const bool k1 = two ? (i == wlo || i == whi) : (i >= wlo && i <= whi);
                                const bool k2 = has20 && i >= 7 && i <= 13;
                                if (k2) {
                                    sum += x;
                                }
                                if (k1) {
                                    sum += x;
                                }
                                c += (k1 ? 1 : 0) + (k2 ? 1 : 0);
                            }
                        }
                        float last = rowp[(pos + left - 1) & RMASKW];
                        if (clip) {
                            last = last > s.max ? s.max : (last < s.min ? s.min : last);
                        }
                        s.sum = sum;
                        s.count = c;
                        s.jitter = (s.jitter < 0 && found >= 0) ? i0 + found : s.jitter;
                        s.lastsample = last;
                        s.i = s.span;
                        pos += left;
                    }
                }
                // sample-at-a-time loop: a symbol carried over a call boundary, the filter's cold start, very short symbols
                bool act = live && s.in_symbol && pos < lim && s.i < s.span;
                while (__any(act)) {
                    if (act) {
                        float x;
                        if (!fo_now) {
                            x = rrow[pos & RMASKW];
                        } else if ((cfg.dbg & 128) || (abs0 + pos - s.filt_start) >= (long long)(NT - 1) || cold_fs == s.filt_start) {
                            x = frow[pos & RMASKW];
                        } else { // first NT-1 samples after the enable: FIR over the filter's stale memory + new samples
                            float acc = 0.0f;
                            for (int i = 0; i < NT; i++) {
                                const long j = (long)pos - (NT - 1) + i;
                                float v;
                                if (abs0 + j >= s.filt_start) {
                                    v = (j >= 0) ? raw[(size_t)ch * stride + j]
                                                 : prev_tail[(size_t)ch * (DDN_FSK4_MAX_TAPS - 1) + (NT - 1) + j];
                                } else {
                                    v = fstale[(size_t)ch * (DDN_FSK4_MAX_TAPS - 1)
                                               + (size_t)((abs0 + j - s.filt_start) + (NT - 1))];
                                }
                                acc += taps[i] * v;
                            }
                            x = acc;
                        }
                        if (clip) {
                            x = x > s.max ? s.max : (x < s.min ? s.min : x);
                        }
                        const int i = s.i;
                        {
                            const bool up = x > s.center;
                            const bool within = up ? !(x > up_lim) : !(x < dn_lim);
                            const bool crossed = up ? (s.lastsample < s.center) : (s.lastsample > s.center);
                            s.jitter = (s.jitter < 0 && within && crossed) ? i : s.jitter;
                        }
                        const int a = adds(i, s.span, s.centre, cfg.rf_mod, l_edge);
                        if (a) {
                            s.sum += x;
                            if (a > 1) {
                                s.sum += x;
                            }
                            s.count += a;
                        }
                        s.lastsample = x;
                        s.i++;
                        pos++;
                    }
                    act = live && s.in_symbol && pos < lim && s.i < s.span;
                }
                // ---- symbol commit ----------------------------------------------------------------------------------
                if (live && s.in_symbol && s.i >= s.span) {
                    const float sym = (s.count > 0) ? (s.sum / (float)s.count) : 0.0f;
                    s.in_symbol = 0;
                    const int slot = s.shead;
                    L.sh[slot][ln] = sym;
                    s.shead = (s.shead + 1 >= HN) ? 0 : s.shead + 1;
                    s.scount = s.scount < HN ? s.scount + 1 : HN;
                    int fl = 0;
                    bool sync_entry = false;
                    // thresholds the symbol is sliced against: as they stand before a sync on this symbol changes them
                    const float q1 = s.center, q2 = s.umid, q3 = s.lmid, q4 = s.max, q5 = s.min;
                    if (s.have_sync) {
                        const int neg = (cfg.dbg & 32) ? 0 : ((L.pat_meta[s.cur_pat] >> 8) & 1);
                        s.maxref = s.max;
                        s.minref = s.min;
                        fl = 1 | (neg ? 4 : 0);
                        if (HM) {
                            hsymbol(sym, neg);
                        }
                        if (--s.lock_left <= 0) {
                            if (HM) {
                                hphase_end();
                            } else {
                                if (Cfg::m17 && (s.cur_pat == 2 || s.cur_pat == 3)) {
                                    s.lastsync = 0;
                                    s.hlich = 0;
                                }
                                hunt_restart(s);
                            }
                        }
                    } else {
                        L.lb[s.lidx][ln] = sym;
                        s.level_count = s.level_count < cfg.t_max ? s.level_count + 1 : cfg.t_max;
                        s.lidx = (s.lidx == cfg.t_max - 1) ? 0 : s.lidx + 1;
                        const uint32_t bit = sym > 0.0f ? 1u : 0u;
                        s.hist_bits = ((s.hist_bits << 1) | bit) & 0xFFFFFFu;
                        s.hist_count = s.hist_count < 24 ? s.hist_count + 1 : 24;
                        bool accepted = false;
                        if (s.hist_count >= 8) {
                            s.maxref = s.max;
                            s.minref = s.min;
                            int hit = -1;
                            if (s.hist_count >= cfg.win_len && !(cfg.dbg & 8)) {
                                const uint32_t w = s.hist_bits & wmask;
                                if (Cfg::m17) {
                                    int pa;
                                    hit = m17_hit(w, s.lastsync, s.hlich, L.pat_meta, pa);
                                    s.hlich = pa;
                                } else {
                                    for (int k = cfg.n_pat - 1; k >= 0; k--) {
                                        hit = (w == L.pat_bits[k]) ? k : hit; // lowest matching index wins, as a forward scan
                                    }
                                }
                            }
                            if (hit >= 0) {
                                // level window of the last level_count hunting symbols (frame_sync_level.c:10-44)
                                const float big = 3.4028234663852886e38f;
                                float a0 = big, a1 = big, a2 = big, a3 = big, a4 = big;
                                float b0 = -big, b1 = -big, b2 = -big, b3 = -big, b4 = -big;
                                const int cnt = s.level_count;
                                // (straight-line, as in the P25 loop: an entry past the fill goes in as +big / -big, which leaves the
                                // five unchanged; stage i of entry k + 1 then only waits for stage i of entry k, four entries at a time)
#pragma unroll 4
                                for (int k = 0; k < 24; k++) {
                                    const float x = L.lb[k][ln];
                                    float v = k < cnt ? x : big, t;
                                    float w = k < cnt ? x : -big;
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
                                if (cnt >= 13) {
                                    s.lmin = (a2 + a3 + a4) / 3.0f;
                                    s.lmax = (b4 + b3 + b2) / 3.0f;
                                } else {
                                    s.lmin = (a0 + a1 + a2) / 3.0f;
                                    s.lmax = (b2 + b1 + b0) / 3.0f;
                                }
                                const uint32_t pm = L.pat_meta[hit];
                                const int type = (int)(pm & 0xFF);
                                s.max = (s.max + s.lmax) / 2;
                                s.min = (s.min + s.lmin) / 2;
                                accepted = !(cfg.confirm && s.lastsync != type);
                                s.lastsync = type;
                                if (use_flt && !s.filter_on) {
                                    s.filter_on = 1;
                                    s.filt_start = abs0 + pos;
                                    if (!(cfg.dbg & 2048)) {
                                        cold_fs = s.filt_start;
                                        cold_p0 = pos;
                                        cold_next = pos;
                                        cold_end = (pos + (NT - 1)) < n ? (pos + (NT - 1)) : n;
                                    }
                                }
                                if (accepted) {
                                    const int wl = cfg.redigitize ? 24 : cfg.warm_len;
                                    // (M17: the EOT marker takes the basic lock only, no warm start - dsd_frame_sync.c:905-933)
                                    if (s.scount >= wl && !(Cfg::m17 && (hit == 2 || hit == 3))) { // dsd_sync_warm_start_thresholds_outer_only(opts, state, wl)
                                        float sp_ = 0.0f, sn_ = 0.0f;
                                        int np = 0, nn = 0, idx = s.shead;
                                        // (no branch: adding +0 to the sum a value does not belong to is exact - neither sum can be -0)
#pragma unroll 4
                                        for (int k = 0; k < wl; k++) {
                                            idx = idx == 0 ? HN - 1 : idx - 1;
                                            const float v = L.sh[idx][ln];
                                            const bool posv = v > 0.0f;
                                            sp_ += posv ? v : 0.0f;
                                            sn_ += posv ? 0.0f : v;
                                            np += posv ? 1 : 0;
                                            nn += posv ? 0 : 1;
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
                                            }
                                        }
                                    }
                                    s.have_sync = 1;
                                    s.cur_pat = hit;
                                    s.lock_left = lock4[(size_t)ch * 4 + ((pm >> 16) & 3)];
                                    fl = 2 | (((pm >> 8) & 1) ? 4 : 0) | (hit << 3);
                                    sync_entry = true;
                                    if (HM) {
                                        using namespace ddn_fsk4h;
                                        if (PROTO == 1 && hit < 2) {
                                            // the 90 dibits the handler finds at dmr_payload_p - 90: the 66 before the sync as
                                            // dmr_resample_on_sync() has just re-sliced them (needs 90 symbols of history), the 24 of
                                            // the sync as they were sliced while hunting
                                            const bool redig = s.scount >= 90;
#pragma unroll 6
                                            for (int i = 0; i < 90; i++) { // (independent entries: six in flight)
                                                int idx = s.shead - 90 + i;
                                                idx += idx < 0 ? HN : 0;
                                                const float v = L.sh[idx][ln];
                                                const bool nw = redig && i < 66;
                                                const float c0 = nw ? s.center : q1, u0 = nw ? s.umid : q2, l0 = nw ? s.lmid : q3;
                                                const int d = v > c0 ? (v > u0 ? 1 : 0) : (v < l0 ? 3 : 2);
                                                LH.pay[i][ln] = (uint8_t)(((90 - i) <= s.scount) ? d : 0);
                                            }
                                            const Ctx x = hctx();
                                            int mode = M_IDLE, next = 0;
                                            const bool on = dmr_begin(x, o, hit == 1, mode, next);
                                            s.hmode = mode;
                                            s.hidx = 90;
                                            s.lock_left = on ? next : 0;
                                        } else if (PROTO == 1) {
                                            s.hmode = M_FIXED; // MS / direct-mode words: the configured count
                                        } else {
                                            s.hmode = M_NX_LICH;
                                            s.hidx = 0;
                                            s.hlich = 0;
                                            s.lock_left = 8;
                                        }
                                    }
                                    if (s.lock_left <= 0) {
                                        s.hmode = 0;
                                        hunt_restart(s);
                                    }
                                }
                            }
                        }
                        if (!accepted) {
                            hunt_advance();
                        }
                    }
                    {
                        const int qb = t & 1;
                        L.q[qb][qk][0][ln] = sym;
                        L.q[qb][qk][1][ln] = q1;
                        L.q[qb][qk][2][ln] = q2;
                        L.q[qb][qk][3][ln] = q3;
                        L.q[qb][qk][4][ln] = q4;
                        L.q[qb][qk][5][ln] = q5;
                        L.q[qb][qk][6][ln] = __int_as_float(fl | (slot << 8));
                        qk++;
                        if (sync_entry) {
                            const bool redig = cfg.redigitize && s.scount >= 90 && s.scount >= 24;
                            L.q[qb][qk][1][ln] = s.center;
                            L.q[qb][qk][2][ln] = s.umid;
                            L.q[qb][qk][3][ln] = s.lmid;
                            L.q[qb][qk][4][ln] = s.max;
                            L.q[qb][qk][5][ln] = s.min;
                            L.q[qb][qk][6][ln] = __int_as_float(fl | (slot << 8) | (1 << 16) | (redig ? (1 << 17) : 0)
                                                                | ((s.scount > 255 ? 255 : s.scount) << 24));
                            L.q[qb][qk][7][ln] = __int_as_float(ns);
                            qk++;
                            ns++;
                        }
                    }
                    o++;
                }
                if (use_flt) {
                    cold_fill(lim); // a filter gated on in this trip: its cold outputs inside the tiles staged so far
                }
                const bool busy = live && pos < tile_end && !s.in_symbol;
                if (!__any(busy) || ++guard > 4 * TSW) {
                    break;
                }
            }
            if (live) {
                L.qn[t & 1][ln] = qk;
            }
        }
        __syncthreads();
    }
    if (loader && n_tiles > 0) {
        drain((n_tiles - 1) & 1);
    }
    __syncthreads();
    if (loader && lane < CPW && ch < n_channels) {
        for (int k = 0; k < HN; k++) {
            phist_store[(size_t)k * n_channels + ch] = L.ph[k][ln];
            rhist_store[(size_t)k * n_channels + ch] = L.rh[k][ln];
        }
    }
    if (live) {
        if (HM) {
            if (n_events) {
                n_events[ch] = LH.hs[ddn_fsk4h::F_NEV][ln];
            }
            for (int k = 0; k < ddn_fsk4h::F_COUNT; k++) {
                hwords[(size_t)ch * ddn_fsk4h::F_COUNT + k] = LH.hs[k][ln];
            }
            for (int k = 0; k < 144; k++) {
                hpay_store[(size_t)ch * 144 + k] = LH.pay[k][ln];
            }
        }
        s.n_abs = abs0 + n;
        state[ch] = s;
        counts[ch] = o;
        n_sync[ch] = ns;
        for (int k = 0; k < 24; k++) {
            lbuf_store[(size_t)k * n_channels + ch] = L.lb[k][ln];
        }
        for (int k = 0; k < HN; k++) {
            shist_store[(size_t)k * n_channels + ch] = L.sh[k][ln];
        }
    }
}

// always-on matched-filter stream (apply_sps_fir order: products added oldest first, mul and add rounded separately);
// hist = the NT-1 samples before this call, rows of DDN_FSK4_MAX_TAPS-1 floats, right-aligned use as in ddn_slicer.hip
template <int NT>
__global__ __launch_bounds__(256) void
k_fsk4_matched_filter(const float* __restrict__ in, long n, size_t stride, const float* __restrict__ hist,
                      float* __restrict__ out) {
    // y[n] = sum_i taps[i] * x[n - (NT - 1) + i], products added oldest first, mul and add rounded separately (as the reference's
    // loop compiles).  (round 5) two outputs per packed f32 operation, as k_p25_matched_filter (ddn_slicer.hip): the tile's input span
    // sits in LDS as pairs at both alignments - E = (x[2m], x[2m+1]), O = (x[2m+1], x[2m+2]) - a thread owns eight consecutive outputs
    // = four pairs, an even tap multiplies the E pairs, an odd tap the O pairs, and the pairs slide through registers: one LDS read
    // and four packed mul + four packed add per tap and thread instead of eight reads, eight mul and eight add.  Every output's sum is
    // formed in the same order as before.
    typedef float mf2 __attribute__((ext_vector_type(2)));
    constexpr int T = 1024, NP = (T + NT + 3) / 2, NH = NP / 4 + 2;
    __shared__ mf2 E[4][NH], O[4][NH];
    const unsigned int* bits = NT == DDN_DMR_FILTER_TAPS ? ddn_dmr_filter_bits : ddn_nxdn48_filter_bits;
    const int ch = blockIdx.y;
    const long t0 = (long)blockIdx.x * T;
    const int tid = threadIdx.x; // 128 threads
    auto sample = [&](int i) -> float { // x[i] of the tile's input span (past the call's end: 0)
        const long j = t0 - (NT - 1) + i;
        if (j < 0) {
            return hist[(size_t)ch * (DDN_FSK4_MAX_TAPS - 1) + (NT - 1) + j];
        }
        return j < n ? in[(size_t)ch * stride + j] : 0.0f;
    };
    // (round 6, as k_p25_matched_filter) a tile whose whole input span lies inside the call's samples on an 8-byte boundary is read as
    // the aligned pairs E is made of: two 8-byte loads per pair index instead of three tested 4-byte loads
    const long j0 = t0 - (NT - 1);
    const float* span = in + (size_t)ch * stride + j0;
    if (j0 >= 0 && j0 + 2 * NP + 2 <= n && (((size_t)span) & 7) == 0) {
        const mf2* pr = (const mf2*)span;
        for (int m = tid; m < NP; m += 128) {
            const mf2 e = pr[m], nx = pr[m + 1];
            E[m & 3][m >> 2] = e;
            O[m & 3][m >> 2] = mf2{e.y, nx.x};
        }
    } else {
        for (int m = tid; m < NP; m += 128) {
            const float a = sample(2 * m), b = sample(2 * m + 1), c = sample(2 * m + 2);
            E[m & 3][m >> 2] = mf2{a, b};
            O[m & 3][m >> 2] = mf2{b, c};
        }
    }
    __syncthreads();
    // this thread's outputs o .. o + 7, o = 8 * tid: pair index m0 = 4 * tid
    mf2 e[4], q[4], acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        e[r] = E[r][tid];
        q[r] = O[r][tid];
        acc[r] = mf2{0.0f, 0.0f};
    }
#pragma unroll
    for (int j = 0; j < (NT + 1) / 2; j++) {
        {
            const float t = __uint_as_float(bits[2 * j]);
            const mf2 tt = {t, t};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[r] += tt * e[r];
            }
        }
        if (2 * j + 1 < NT) {
            const float t = __uint_as_float(bits[2 * j + 1]);
            const mf2 tt = {t, t};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[r] += tt * q[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            e[r] = e[r + 1];
            q[r] = q[r + 1];
        }
        if (j + 1 < (NT + 1) / 2) { // pair 4 * tid + j + 4
            e[3] = E[j & 3][tid + (j + 4) / 4];
            q[3] = O[j & 3][tid + (j + 4) / 4];
        }
    }
    const long o = t0 + 8 * tid;
    float* dst = out + (size_t)ch * stride + o;
    if (o + 7 < n) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            *(mf2*)&dst[2 * r] = acc[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (o + 2 * r < n) {
                dst[2 * r] = acc[r].x;
            }
            if (o + 2 * r + 1 < n) {
                dst[2 * r + 1] = acc[r].y;
            }
        }
    }
}

__global__ void
k_fsk4_filter_hist(int nt, const float* __restrict__ in, long n, size_t stride, float* __restrict__ hist) {
    const int H = nt - 1;
    const int ch = blockIdx.x, i = threadIdx.x;
    float* h = hist + (size_t)ch * (DDN_FSK4_MAX_TAPS - 1);
    float v = 0.0f;
    if (i < H) {
        const long j = n - H + i;
        v = (j >= 0) ? in[(size_t)ch * stride + j] : h[H + j];
    }
    __syncthreads();
    if (i < H) {
        h[i] = v;
    }
}

// DMR burst fields for every accepted sync: slot type (10 + 10 bits around the sync), the 196 info bits (98 + 98), the
// CACH de-interleaved (ETSI TS 102 361-1 CACH interleave: TACT bits first) - dmr_data.c:117-262.  One workgroup per sync slot.
__constant__ uint8_t c_cach_il[24] = {0, 7, 8, 9, 1, 10, 11, 12, 2, 13, 14, 15, 3, 16, 4, 17, 18, 19, 5, 20, 21, 22, 6, 23};
__global__ void
k_dmr_burst_gather(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym,
                   const int32_t* __restrict__ sync_pos, const uint8_t* __restrict__ pre, const int32_t* __restrict__ n_sync,
                   int max_sync, int inverted, uint8_t* __restrict__ slot_type, uint8_t* __restrict__ info,
                   uint8_t* __restrict__ cach, uint8_t* __restrict__ valid) {
    const int k = blockIdx.x, ch = blockIdx.y, t = threadIdx.x; // 128 threads: t < 90 cached dibits, 90 <= t < 144 live
    const size_t so = (size_t)ch * max_sync + k;
    const bool have = k < n_sync[ch] && k < max_sync;
    const long pos = have ? sync_pos[so] : 0;
    const bool live_ok = have && (pos + 54 < (long)counts[ch]) && ((size_t)(pos + 54) < max_sym);
    if (t == 0) {
        valid[so] = live_ok ? 1 : 0;
    }
    for (int d = t; d < 144; d += blockDim.x) {
        int dibit = 0;
        if (d < 90) {
            dibit = have ? pre[so * DDN_FSK4_PRE + d] : 0;
            if (inverted) {
                dibit ^= 2;
            }
        } else if (live_ok) {
            dibit = rec[((size_t)ch * max_sym + (size_t)(pos + 1 + (d - 90))) * 10] & 3; // getDibitSoft()'s return value
        }
        const uint8_t hi = (uint8_t)((dibit >> 1) & 1), lo = (uint8_t)(dibit & 1);
        if (d < 12) {
            cach[so * 24 + c_cach_il[2 * d]] = hi;
            cach[so * 24 + c_cach_il[2 * d + 1]] = lo;
        } else if (d < 61) {
            info[so * 196 + 2 * (d - 12)] = hi;
            info[so * 196 + 2 * (d - 12) + 1] = lo;
        } else if (d < 66) {
            slot_type[so * 20 + 2 * (d - 61)] = hi;
            slot_type[so * 20 + 2 * (d - 61) + 1] = lo;
        } else if (d < 90) {
            // the sync itself: not part of any field
        } else if (d < 95) {
            slot_type[so * 20 + 10 + 2 * (d - 90)] = hi;
            slot_type[so * 20 + 10 + 2 * (d - 90) + 1] = lo;
        } else {
            info[so * 196 + 98 + 2 * (d - 95)] = hi;
            info[so * 196 + 98 + 2 * (d - 95) + 1] = lo;
        }
    }
}

// NXDN frame fields for every accepted sync (nxdn_frame(), src/protocol/nxdn/nxdn_frame.c:181-199,311-331,596-621): the 182
// dibits after the frame sync word are de-scrambled (PN9 x^9 + x^4 + 1 from 0x0E4: a set bit flips the dibit's sign bit,
// nxdn_descramble.c:35-57), the LICH is read from the first 8, the SACCH (60 bits) and the two FACCH1 fields (144 bits each)
// are de-interleaved (12 x 5 / 16 x 9 block interleavers, nxdn_const.h:29-40: bit i goes to (i mod R) * C + i / R) and
// de-punctured (nxdn_deperm.c:139-172) into the symbol / reliability pairs the K = 5 decoder takes (bit << 1; a punctured
// position is symbol 0 with reliability 0).  One workgroup per sync slot.
__global__ void
k_nxdn_frame_gather(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym,
                    const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, int max_sync,
                    uint8_t* __restrict__ lich, uint8_t* __restrict__ sacch_sym, uint8_t* __restrict__ sacch_rel,
                    uint8_t* __restrict__ facch_sym, uint8_t* __restrict__ facch_rel, uint8_t* __restrict__ valid) {
    __shared__ uint8_t bit[364], rel[364];
    __shared__ uint8_t pn[182];
    const int k = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
    const size_t so = (size_t)ch * max_sync + k;
    const bool have = k < n_sync[ch] && k < max_sync;
    const long pos = have ? sync_pos[so] : 0;
    const bool ok = have && (pos + 182 < (long)counts[ch]) && ((size_t)(pos + 182) < max_sym);
    if (!ok) { // an unused slot (most of them: the slots are sized for the densest traffic) or a frame the records do not hold yet:
        if (t == 0) { // flagged, its fields are not defined (the decoders behind skip what is not valid)
            valid[so] = 0;
            lich[so] = 0;
        }
        return;
    }
    if (t == 0) {
        unsigned l = 228u;
        for (int i = 0; i < 182; i++) {
            pn[i] = (uint8_t)(l & 1u);
            const unsigned b = ((l >> 4) ^ l) & 1u;
            l = (l >> 1) | (b << 8);
        }
        valid[so] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int i = t; i < 182; i += blockDim.x) {
        int d = 0, r = 0;
        if (ok) {
            const uint8_t* rr = rec + ((size_t)ch * max_sym + (size_t)(pos + 1 + i)) * 10;
            d = rr[0] & 3; // getDibitSoft()'s return value
            r = rr[1];
        }
        d ^= pn[i] << 1;
        bit[2 * i] = (uint8_t)(d >> 1);
        bit[2 * i + 1] = (uint8_t)(d & 1);
        rel[2 * i] = rel[2 * i + 1] = (uint8_t)r;
    }
    __syncthreads();
    if (t == 0) {
        int l = 0;
        for (int i = 0; i < 8; i++) {
            l |= bit[2 * i] << (7 - i);
        }
        int par = ((l >> 7) + (l >> 6) + (l >> 5) + (l >> 4)) & 1;
        const int l7 = l >> 1;
        if (l7 == 0x08 || l7 == 0x4A || l7 == 0x48 || l7 == 0x46) {
            par = ((l >> 7) + (l >> 6) + (l >> 5) + (l >> 4) + (l >> 3) + (l >> 2) + (l >> 1)) & 1;
        }
        lich[so] = (uint8_t)(l7 | (((l & 1) == par && ok) ? 0x80 : 0)); // bit 7: parity good
    }
    // SACCH: output position q of 72 <- de-punctured index; groups of 12 from 10
    for (int q = t; q < 72; q += blockDim.x) {
        const int g = q / 12, m = q % 12;
        const bool punct = (m == 5 || m == 11);
        const int dp = g * 10 + (m < 5 ? m : m - 1); // index into the de-interleaved 60
        // de-interleave: deperm[(i % 5) * 12 + i / 5] = in[i]  ->  in index of deperm position p: i = (p % 12) * 5 + p / 12
        const int src = (dp % 12) * 5 + dp / 12;
        sacch_sym[so * 72 + q] = punct ? 0 : (uint8_t)(bit[16 + src] << 1);
        sacch_rel[so * 72 + q] = punct ? 0 : rel[16 + src];
    }
    // FACCH1 a / b: 144 -> 192: every group of 3 becomes {b0, punctured, b1, b2}
    for (int q = t; q < 384; q += blockDim.x) {
        const int f = q / 192, qq = q % 192;
        const int g = qq / 4, m = qq % 4;
        const bool punct = (m == 1);
        const int dp = g * 3 + (m == 0 ? 0 : m - 1);
        const int src = (dp % 16) * 9 + dp / 16; // deperm[(i % 9) * 16 + i / 9] = in[i]
        const int base = 16 + 60 + 144 * f;
        facch_sym[so * 384 + q] = punct ? 0 : (uint8_t)(bit[base + src] << 1);
        facch_rel[so * 384 + q] = punct ? 0 : rel[base + src];
    }
}

// CRC of a decoded NXDN field, bit-serial as the reference computes it (crc6: nxdn_deperm.c:1246-1261, x^6+x^5+x^2+x+1;
// crc12f: nxdn_dcr_utils.c:21-42, x^12+x^11+x^3+x^2+x+1; registers start all ones): kind 0 = SACCH (26 + 6 bits), 1 = FACCH1
// (80 + 12).  bytes = the K = 5 decoder's output rows, MSB first.
__global__ void
k_nxdn_crc(const uint8_t* __restrict__ bytes, int stride, int n, int kind, uint8_t* __restrict__ ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const uint8_t* b = bytes + (size_t)i * stride;
    const bool bitrows = (kind & 2) != 0; // rows hold one bit per byte (trellis_decode output) instead of packed bytes
    kind &= 1;
    const int nd = kind == 0 ? 26 : 80, nc = kind == 0 ? 6 : 12;
    const unsigned poly = kind == 0 ? 0x27u : 0x80Fu; // x^5+x^2+x+1 / x^11+x^3+x^2+x+1 below the leading term
    const unsigned top = 1u << (nc - 1), mask = (1u << nc) - 1u;
    unsigned crc = mask;
    for (int k = 0; k < nd; k++) {
        const unsigned in = bitrows ? (b[k] & 1u) : ((b[k >> 3] >> (7 - (k & 7))) & 1u);
        const unsigned fb = ((crc & top) ? 1u : 0u) ^ in;
        crc = (crc << 1) & mask;
        if (fb) {
            crc ^= poly;
        }
    }
    unsigned got = 0;
    for (int k = nd; k < nd + nc; k++) {
        got = (got << 1) | (bitrows ? (b[k] & 1u) : ((b[k >> 3] >> (7 - (k & 7))) & 1u));
    }
    ok[i] = crc == got ? 1 : 0;
}

template <int CPW, int MAXW, int PROTO, bool HM>
hipError_t
launch(const float* raw, const float* filt, const float* prev_tail, float* fstale, const float* taps, long n, size_t stride,
       int n_channels, const DdnFsk4Config* cfg, DdnFsk4State* state, float* lbuf_store, float* shist_store,
       uint8_t* phist_store, uint8_t* rhist_store, uint8_t* rec, uint8_t* flags, uint8_t* pay, int32_t* counts,
       size_t max_sym, const int32_t* lock4, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre, uint8_t* pre_rel,
       int32_t* n_sync, int max_sync, int32_t* hwords, uint8_t* hpay, const DdnFec3Tables* htab, int32_t* events,
       int32_t* n_events, hipStream_t st) {
    const size_t shmem = HM ? (((sizeof(Lds4<CPW>) + 15) & ~(size_t)15) + sizeof(Lds4H<CPW>)) : sizeof(Lds4<CPW>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fsk4_rx<CPW, MAXW, PROTO, HM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) {
        return e;
    }
    if (DDN_EXP_ENV("DDN_RX_OCC")) {
        int nb = -1;
        hipFuncAttributes fa;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_fsk4_rx<CPW, MAXW, PROTO, HM>), 128, shmem);
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_fsk4_rx<CPW, MAXW, PROTO, HM>));
        fprintf(stderr, "k_fsk4_rx<%d,%d,%d,%d>: %d blocks per CU alone, %d regs, %zu B dynamic LDS, %d blocks\n", CPW, MAXW, PROTO, (int)HM, nb,
                fa.numRegs, shmem, (n_channels + CPW - 1) / CPW);
    }
    hipLaunchKernelGGL((k_fsk4_rx<CPW, MAXW, PROTO, HM>), dim3((unsigned)((n_channels + CPW - 1) / CPW)), dim3(128), shmem, st,
                       raw, filt, prev_tail, fstale, taps, n, stride, n_channels, cfg, state, lbuf_store, shist_store,
                       phist_store, rhist_store, rec, flags, pay, counts, max_sym, lock4, sync_pos, sync_pat, pre, pre_rel,
                       n_sync, max_sync, hwords, hpay, htab, events, n_events);
    return hipGetLastError();
}
} // namespace

extern "C" hipError_t
ddn_dev_fsk4_rx(const float* raw, const float* filt, const float* prev_tail, float* fstale, const float* taps, long n,
                size_t stride, int n_channels, const DdnFsk4Config* cfg, DdnFsk4State* state, float* lbuf_store,
                float* shist_store, uint8_t* phist_store, uint8_t* rhist_store, uint8_t* rec, uint8_t* flags, uint8_t* pay,
                int32_t* counts, size_t max_sym, const int32_t* lock4, int32_t* sync_pos, uint8_t* sync_pat, uint8_t* pre,
                uint8_t* pre_rel, int32_t* n_sync, int max_sync, int channels_per_wave, int cfg_sps, int protocol, int handlers,
                int32_t* hwords, uint8_t* hpay, int32_t* events, int32_t* n_events, hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    if (protocol < 1 || protocol > 5) {
        return hipErrorInvalidValue;
    }
    if ((protocol == 4 || protocol == 5) && handlers) {
        return hipErrorInvalidValue; // M17 / YSF frames are fixed counts: no handler family
    }
    const DdnFec3Tables* htab = nullptr;
    if (handlers) {
        if (!hwords || !hpay) {
            return hipErrorInvalidValue;
        }
        const hipError_t e = ddn_dev_fec3_tables(&htab, st);
        if (e != hipSuccess) {
            return e;
        }
    }
    // MAXW: the longest whole symbol the straight pass takes (samples per symbol + one slip sample); 12 covers 4800 baud at
    // 48 ksps, 22 covers 2400 baud
    const int sps = cfg_sps > 0 ? cfg_sps : 64;
#define DDN_RX4_ARGS                                                                                                       \
    raw, filt, prev_tail, fstale, taps, n, stride, n_channels, cfg, state, lbuf_store, shist_store, phist_store, rhist_store, rec,  \
        flags, pay, counts, max_sym, lock4, sync_pos, sync_pat, pre, pre_rel, n_sync, max_sync, hwords, hpay, htab, events,     \
        n_events, st
#define DDN_RX4_GO(CPW_, MAXW_)                                                                                            \
    do {                                                                                                                   \
        if (protocol == 1) {                                                                                               \
            return handlers ? launch<CPW_, MAXW_, 1, true>(DDN_RX4_ARGS) : launch<CPW_, MAXW_, 1, false>(DDN_RX4_ARGS);     \
        }                                                                                                                  \
        if (protocol == 3) { /* NXDN96: 10 samples per symbol at 48 ksps - the straight pass of 12 */                      \
            return handlers ? launch<CPW_, 12, 3, true>(DDN_RX4_ARGS) : launch<CPW_, 12, 3, false>(DDN_RX4_ARGS);           \
        }                                                                                                                  \
        if (protocol == 4) { /* M17: 4800 symbols/s, fixed counts */                                                       \
            return launch<CPW_, 12, 4, false>(DDN_RX4_ARGS);                                                                \
        }                                                                                                                  \
        if (protocol == 5) { /* YSF: 4800 symbols/s, fixed counts */                                                       \
            return launch<CPW_, 12, 5, false>(DDN_RX4_ARGS);                                                                \
        }                                                                                                                  \
        return handlers ? launch<CPW_, MAXW_, 2, true>(DDN_RX4_ARGS) : launch<CPW_, MAXW_, 2, false>(DDN_RX4_ARGS);         \
    } while (0)
    if (channels_per_wave <= 1) {
        if (sps <= 11) {
            DDN_RX4_GO(1, 12);
        }
        DDN_RX4_GO(1, 22);
    }
    if (channels_per_wave <= 2) {
        if (sps <= 11) {
            DDN_RX4_GO(2, 12);
        }
        DDN_RX4_GO(2, 22);
    }
    if (channels_per_wave <= 4) {
        if (sps <= 11) {
            DDN_RX4_GO(4, 12);
        }
        DDN_RX4_GO(4, 22);
    }
    if (channels_per_wave <= 8) {
        if (sps <= 11) {
            DDN_RX4_GO(8, 12);
        }
        DDN_RX4_GO(8, 22);
    }
    if (channels_per_wave <= 16) {
        if (sps <= 11) {
            DDN_RX4_GO(16, 12);
        }
        DDN_RX4_GO(16, 22);
    }
    if (sps <= 11) {
        DDN_RX4_GO(32, 12);
    }
    DDN_RX4_GO(32, 22);
#undef DDN_RX4_GO
#undef DDN_RX4_ARGS
}

extern "C" hipError_t
ddn_dev_dmr_burst_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos, const uint8_t* pre,
                         const int32_t* n_sync, int n_channels, int max_sync, int inverted, uint8_t* slot_type, uint8_t* info,
                         uint8_t* cach, uint8_t* valid, hipStream_t st) {
    if (n_channels <= 0 || max_sync <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_burst_gather, dim3((unsigned)max_sync, (unsigned)n_channels), dim3(64), 0, st, rec, counts, max_sym,
                       sync_pos, pre, n_sync, max_sync, inverted, slot_type, info, cach, valid);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_nxdn_frame_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos, const int32_t* n_sync,
                          int n_channels, int max_sync, uint8_t* lich, uint8_t* sacch_sym, uint8_t* sacch_rel, uint8_t* facch_sym,
                          uint8_t* facch_rel, uint8_t* valid, hipStream_t st) {
    if (n_channels <= 0 || max_sync <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_nxdn_frame_gather, dim3((unsigned)max_sync, (unsigned)n_channels), dim3(128), 0, st, rec, counts, max_sym,
                       sync_pos, n_sync, max_sync, lich, sacch_sym, sacch_rel, facch_sym, facch_rel, valid);
    return hipGetLastError();
}

// ---- AMBE 3600x2450 voice frames: the 36-dibit interleave schedule DMR, NXDN and YSF share ------------------------------
// include/dsd-neo/core/ambe_interleave.h:25-38 (table measured from the compiled reference, ddn_tables_ambe.h): dibit i puts
// its high bit at ambe_fr[map[i][0]][map[i][1]] and its low bit at ambe_fr[map[i][2]][map[i][3]]; both bits take the dibit's
// reliability (nxdn_voice.c:57-74); the 24 cells of the 4 x 24 array the schedule never writes stay 0.
__constant__ uint8_t c_ambe2450_map[36][4] = DDN_AMBE2450_MAP_INIT;

__device__ __forceinline__ void
ambe2450_put(uint8_t* fr, uint8_t* rl, int i, int dibit, int reliab) {
    const int hi = c_ambe2450_map[i][0] * 24 + c_ambe2450_map[i][1], lo = c_ambe2450_map[i][2] * 24 + c_ambe2450_map[i][3];
    fr[hi] = (uint8_t)((dibit >> 1) & 1);
    fr[lo] = (uint8_t)(dibit & 1);
    if (rl) {
        rl[hi] = (uint8_t)reliab;
        rl[lo] = (uint8_t)reliab;
    }
}

// n frames of 36 dibits (+ optional reliabilities) -> ambe_fr [n][4][24] (+ per-bit reliabilities): one frame per 64 threads
__global__ __launch_bounds__(64) void
k_ambe2450_deinterleave(const uint8_t* __restrict__ dibits, const uint8_t* __restrict__ reliab, int n,
                        uint8_t* __restrict__ fr, uint8_t* __restrict__ rl) {
    const int f = blockIdx.x, t = threadIdx.x;
    if (f >= n) {
        return;
    }
    uint8_t* o = fr + (size_t)f * 96;
    uint8_t* r = rl ? rl + (size_t)f * 96 : nullptr;
    for (int k = t; k < 96; k += 64) {
        o[k] = 0;
        if (r) {
            r[k] = 0;
        }
    }
    __syncthreads();
    if (t < 36) {
        ambe2450_put(o, r, t, dibits[(size_t)f * 36 + t] & 3, reliab ? reliab[(size_t)f * 36 + t] : 0);
    }
}

// NXDN voice: the four 36-dibit AMBE frames behind LICH (8 dibits) + SACCH (30) of the frame that follows sync k of channel
// ch, de-scrambled with the same PN9 sequence as the control fields (nxdn_frame.c:181-199: every one of the 182 dibits after
// the sync word; nxdn_voice.c:57-66: frame v starts at de-scrambled dibit 38 + 36 v).  Which of the four carry voice is the
// LICH's business (k_nxdn_frame_gather); all four are de-interleaved here.
__global__ __launch_bounds__(192) void
k_nxdn_voice_gather(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym,
                    const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, int max_sync,
                    uint8_t* __restrict__ fr, uint8_t* __restrict__ rl, uint8_t* __restrict__ valid) {
    __shared__ uint8_t pn[182];
    const int k = blockIdx.x, ch = blockIdx.y, t = threadIdx.x;
    const size_t so = (size_t)ch * max_sync + k;
    const bool have = k < n_sync[ch] && k < max_sync;
    const long pos = have ? sync_pos[so] : 0;
    const bool ok = have && (pos + 182 < (long)counts[ch]) && ((size_t)(pos + 182) < max_sym);
    if (t == 0) {
        unsigned l = 228u;
        for (int i = 0; i < 182; i++) {
            pn[i] = (uint8_t)(l & 1u);
            const unsigned b = ((l >> 4) ^ l) & 1u;
            l = (l >> 1) | (b << 8);
        }
        if (valid) {
            valid[so] = ok ? 1 : 0;
        }
    }
    uint8_t* o = fr + so * 4 * 96;
    uint8_t* r = rl ? rl + so * 4 * 96 : nullptr;
    for (int q = t; q < 4 * 96; q += blockDim.x) {
        o[q] = 0;
        if (r) {
            r[q] = 0;
        }
    }
    __syncthreads();
    if (ok && t < 144) {
        const int i = 38 + t;
        const uint8_t* rr = rec + ((size_t)ch * max_sym + (size_t)(pos + 1 + i)) * 10;
        const int d = (rr[0] & 3) ^ (pn[i] << 1);
        ambe2450_put(o + (t / 36) * 96, r ? r + (t / 36) * 96 : nullptr, t % 36, d, rr[1]);
    }
}

// DMR voice burst (dmr_bs.c:128-200, dmr_ms.c:96-140): 144 dibits from the burst's first CACH dibit: CACH 0..11, voice frame
// 1 = dibits 12..47, frame 2 = 48..65 + 90..107 (either side of the 24 sync / EMB dibits 66..89), frame 3 = 108..143.
// burst_start [n_channels][max_bursts] = record index of the first CACH dibit (< 0: unused).  inverted != 0 applies the MS
// path's dibit ^= 2 (dmr_ms.c:55-64).
__global__ __launch_bounds__(160) void
k_dmr_voice_gather(const uint8_t* __restrict__ rec, const int32_t* __restrict__ counts, size_t max_sym,
                   const int32_t* __restrict__ burst_start, int max_bursts, int inverted, uint8_t* __restrict__ fr,
                   uint8_t* __restrict__ rl, uint8_t* __restrict__ sync48, uint8_t* __restrict__ cach24,
                   uint8_t* __restrict__ valid, int rows_per_channel, const int32_t* __restrict__ pre_slot,
                   const uint8_t* __restrict__ pre90, uint8_t* __restrict__ skip3, const uint8_t* __restrict__ pre90_b, long split) {
    // (the chain's talk paths: row = 2 * channel + time slot, rows_per_channel = 2; a burst whose sync the frame sync search found
    // takes its first 90 dibits from that sync's hand-over, pre90[pre_slot] - dmrBSBootstrap(), dmr_bs.c:697-760)
    const int k = blockIdx.x, row = blockIdx.y, ch = row / rows_per_channel, t = threadIdx.x;
    const size_t so = (size_t)row * max_bursts + k;
    const long s0 = burst_start[so];
    const long ps = (pre_slot && s0 >= 0) ? (long)pre_slot[so] : -1;
    const bool ok = s0 >= 0 && s0 + 144 <= (long)counts[ch] && (size_t)(s0 + 144) <= max_sym;
    if (skip3 && t < 3) {
        skip3[so * 3 + t] = ok ? 0 : 0xFF;
    }
    uint8_t* o = fr + so * 3 * 96;
    uint8_t* r = rl ? rl + so * 3 * 96 : nullptr;
    for (int q = t; q < 3 * 96; q += blockDim.x) {
        o[q] = 0;
        if (r) {
            r[q] = 0;
        }
    }
    if (t == 0 && valid) {
        valid[so] = ok ? 1 : 0;
    }
    __syncthreads();
    if (t >= 144) {
        return;
    }
    int d = 0, q = 0;
    if (ok) {
        const uint8_t* rr = rec + ((size_t)ch * max_sym + (size_t)(s0 + t)) * 10;
        d = ((rr[0] & 3) ^ (inverted ? 2 : 0)) & 3;
        q = rr[1];
        if (ps >= 0 && t < 90) {
            // seed_dmr_bs_bootstrap_payload(): already turned round when inverted.  (pre_slot >= split: the sync waits for the
            // next call's decode pass - its hand-over is in the carry-out list)
            d = (pre90_b && ps >= split ? pre90_b[(size_t)(ps - split) * 90 + t] : pre90[(size_t)ps * 90 + t]) & 3;
            q = 0;
        }
    }
    if (t < 12) {
        if (cach24) {
            cach24[so * 24 + c_cach_il[2 * t]] = (uint8_t)(d >> 1);
            cach24[so * 24 + c_cach_il[2 * t + 1]] = (uint8_t)(d & 1);
        }
    } else if (t >= 66 && t < 90) {
        if (sync48) {
            sync48[so * 48 + 2 * (t - 66)] = (uint8_t)(d >> 1);
            sync48[so * 48 + 2 * (t - 66) + 1] = (uint8_t)(d & 1);
        }
    } else {
        const int f = t < 48 ? 0 : (t < 108 ? 1 : 2);
        const int i = t < 48 ? t - 12 : (t < 66 ? t - 48 : (t < 108 ? 18 + t - 90 : t - 108));
        ambe2450_put(o + f * 96, r ? r + f * 96 : nullptr, i, d, q);
    }
}

// P25 Phase 2 4V / 2V bursts: p25p2_unpack_voice_frames() (src/protocol/p25/phase2/p25p2_frame.c:250-262,849-900) - frame f's 72 bits
// start at bit 2 / 76 / 172 / 246 of the (de-scrambled) timeslot; bit x goes to ambe_fr[w][b] by the csubset / c0..c3 schedule, which is
// the AMBE 3600x2450 dibit schedule read bit by bit (bit 2 i = dibit i's high bit, 2 i + 1 its low bit: checked entry by entry
// against the table measured from the compiled reference, tests/test_oracle_p25p2_xcch.py); soft bit = {bit, min(|LLR|, 255)}
// (p25p2_soft_bit_from_abs_bit()).  One workgroup per burst, thread = (frame, bit).
__global__ __launch_bounds__(288) void
k_p2_voice_unpack(const uint8_t* __restrict__ xbits360, const int16_t* __restrict__ xllr360, int frame_count, uint8_t* __restrict__ fr,
                  uint8_t* __restrict__ rl) {
    const int i = blockIdx.x, t = threadIdx.x, f = t / 72, x = t % 72;
    uint8_t* o = fr + (size_t)i * frame_count * 96;
    uint8_t* r = rl + (size_t)i * frame_count * 96;
    for (int q = t; q < frame_count * 96; q += blockDim.x) {
        o[q] = 0;
        r[q] = 0;
    }
    __syncthreads();
    if (f >= frame_count) {
        return;
    }
    const int off = f == 0 ? 2 : (f == 1 ? 76 : (f == 2 ? 172 : 246));
    const int bit = xbits360[(size_t)i * 360 + off + x] & 1;
    int v = xllr360[(size_t)i * 360 + off + x];
    v = v < 0 ? -v : v;
    v = v > 255 ? 255 : v;
    const int d = x >> 1, lo = x & 1;
    const int cell = lo ? c_ambe2450_map[d][2] * 24 + c_ambe2450_map[d][3] : c_ambe2450_map[d][0] * 24 + c_ambe2450_map[d][1];
    o[f * 96 + cell] = (uint8_t)bit;
    r[f * 96 + cell] = (uint8_t)v;
}

extern "C" hipError_t
ddn_dev_p25p2_voice_unpack(const uint8_t* xbits360, const int16_t* xllr360, int n, int frame_count, uint8_t* fr, uint8_t* rl, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_voice_unpack, dim3((unsigned)n), dim3(288), 0, st, xbits360, xllr360, frame_count, fr, rl);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_ambe2450_deinterleave(const uint8_t* dibits, const uint8_t* reliab, int n, uint8_t* fr, uint8_t* rl, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_ambe2450_deinterleave, dim3((unsigned)n), dim3(64), 0, st, dibits, reliab, n, fr, rl);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_nxdn_voice_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* sync_pos,
                          const int32_t* n_sync, int max_sync, int n_channels, uint8_t* fr, uint8_t* rl, uint8_t* valid,
                          hipStream_t st) {
    if (n_channels <= 0 || max_sync <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_nxdn_voice_gather, dim3((unsigned)max_sync, (unsigned)n_channels), dim3(192), 0, st, rec, counts, max_sym,
                       sync_pos, n_sync, max_sync, fr, rl, valid);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_dmr_voice_gather(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* burst_start, int max_bursts,
                         int n_channels, int inverted, uint8_t* fr, uint8_t* rl, uint8_t* sync48, uint8_t* cach24,
                         uint8_t* valid, hipStream_t st) {
    if (n_channels <= 0 || max_bursts <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_voice_gather, dim3((unsigned)max_bursts, (unsigned)n_channels), dim3(160), 0, st, rec, counts, max_sym,
                       burst_start, max_bursts, inverted, fr, rl, sync48, cach24, valid, 1, nullptr, nullptr, nullptr, nullptr, 0L);
    return hipGetLastError();
}

// ---- the DMR chain's voice stage -------------------------------------------------------------------------------------------------
// Which bursts the reference's BS voice handlers hand to the vocoder is decided inside the receive loop (ddn_fsk4h_dev.h): every
// "voice burst proper" of dmrBS() and the bootstrap burst of dmrBSBootstrap() while the slot's voice gate is open leave a kind-6
// event {position of the burst's last symbol, 6, colour code, VC >= 1 | slot << 16} (process_dmr_bs_voice_burst(), dmr_bs.c:585-640;
// process_dmr_bs_bootstrap_voice_if_open()).  One thread per channel files them by time slot: talk path = 2 * channel + slot, in air
// order; start = row index of the burst's first CACH dibit (the row holds `carry` records of the previous call, then this call's);
// pre = the sync slot whose 90-dibit hand-over are the burst's first 90 dibits when the burst is the one the sync search found
// (the sync's last symbol is the burst's dibit 89), else -1.  The sync is either in this call's decode list (slot index) or, when
// the records behind it are not all in yet, in the list carried to the next call (n_channels * max_syncs + its index there).
__global__ void
k_dmr_voice_select(const int32_t* __restrict__ events, const int32_t* __restrict__ n_events, int max_events, int carry,
                   const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync, int max_syncs, int n_channels,
                   int max_bursts, int32_t* __restrict__ vstart, int32_t* __restrict__ vpre, int32_t* __restrict__ vn,
                   const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_n, int max_out,
                   const int32_t* __restrict__ n_new) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= n_channels) {
        return;
    }
    int n[2] = {0, 0};
    const int ne = n_events[ch] < max_events ? n_events[ch] : max_events;
    const int ns = n_sync[ch] < max_syncs ? n_sync[ch] : max_syncs;
    for (int i = 0; i < ne; i++) {
        const int32_t* e = events + ((size_t)ch * max_events + i) * 4;
        const int vc = e[3] & 0xFFFF, slot = (e[3] >> 16) & 1;
        if (e[1] != 6 || vc < 1) {
            continue;
        }
        const int k = n[slot]++;
        if (k >= max_bursts) {
            continue;
        }
        const size_t so = ((size_t)ch * 2 + slot) * max_bursts + k;
        vstart[so] = carry + e[0] - 143;
        int pre = -1;
        for (int j = 0; j < ns; j++) {
            if (sync_pos[(size_t)ch * max_syncs + j] == carry + e[0] - 54) {
                pre = ch * max_syncs + j;
            }
        }
        if (pre < 0 && out_pos) {
            const int no = out_n[ch] < max_out ? out_n[ch] : max_out;
            for (int j = 0; j < no; j++) {
                if (out_pos[(size_t)ch * max_out + j] == carry + e[0] - 54 - n_new[ch]) {
                    pre = n_channels * max_syncs + ch * max_out + j;
                }
            }
        }
        vpre[so] = pre;
    }
    for (int slot = 0; slot < 2; slot++) {
        vn[ch * 2 + slot] = n[slot] < max_bursts ? n[slot] : max_bursts;
        for (int k = n[slot]; k < max_bursts; k++) {
            vstart[((size_t)ch * 2 + slot) * max_bursts + k] = -1;
            vpre[((size_t)ch * 2 + slot) * max_bursts + k] = -1;
        }
    }
}

extern "C" hipError_t
ddn_dev_dmr_voice_select(const int32_t* events, const int32_t* n_events, int max_events, int carry, const int32_t* sync_pos,
                         const int32_t* n_sync, int max_syncs, int n_channels, int max_bursts, int32_t* vstart, int32_t* vpre,
                         int32_t* vn, const int32_t* out_pos, const int32_t* out_n, int max_out, const int32_t* n_new, hipStream_t st) {
    if (n_channels <= 0 || max_bursts <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_voice_select, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, events, n_events, max_events, carry,
                       sync_pos, n_sync, max_syncs, n_channels, max_bursts, vstart, vpre, vn, out_pos, out_n, max_out, n_new);
    return hipGetLastError();
}

// the bursts k_dmr_voice_select filed -> three AMBE frames each [2 B][max_bursts][3][4][24] + a skip flag per frame
extern "C" hipError_t
ddn_dev_dmr_voice_gather_paths(const uint8_t* rec, const int32_t* counts, size_t max_sym, const int32_t* vstart, const int32_t* vpre,
                               const uint8_t* pre90, int max_bursts, int n_channels, int inverted, uint8_t* fr, uint8_t* skip3,
                               const uint8_t* pre90_out, long split, hipStream_t st) {
    if (n_channels <= 0 || max_bursts <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_dmr_voice_gather, dim3((unsigned)max_bursts, (unsigned)(2 * n_channels)), dim3(160), 0, st, rec, counts, max_sym,
                       vstart, max_bursts, inverted, fr, nullptr, nullptr, nullptr, nullptr, 2, vpre, pre90, skip3, pre90_out, split);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_nxdn_crc(const uint8_t* bytes, int stride, int n, int kind, uint8_t* ok, hipStream_t st) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_nxdn_crc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bytes, stride, n, kind, ok);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_fsk4_matched_filter(int nt, const float* in, long n, size_t stride, int n_channels, const float* hist, float* out,
                            hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    const dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)n_channels);
    if (nt == DDN_DMR_FILTER_TAPS) {
        hipLaunchKernelGGL((k_fsk4_matched_filter<DDN_DMR_FILTER_TAPS>), grid, dim3(128), 0, st, in, n, stride, hist, out);
    } else if (nt == DDN_NXDN48_FILTER_TAPS) {
        hipLaunchKernelGGL((k_fsk4_matched_filter<DDN_NXDN48_FILTER_TAPS>), grid, dim3(128), 0, st, in, n, stride, hist, out);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_fsk4_filter_hist_update(int nt, const float* in, long n, size_t stride, int n_channels, float* hist,
                                hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_fsk4_filter_hist, dim3((unsigned)n_channels), dim3(192), 0, st, nt, in, n, stride, hist);
    return hipGetLastError();
}
