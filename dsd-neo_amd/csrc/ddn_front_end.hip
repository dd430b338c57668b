// ddn_front_end.hip — gfx950 kernel for the dsd-neo FSK front end, one persistent launch per call:
//
//   cu8/cf32 widen -> zero-latency symmetric complex channel LPF -> phase delta   (sample-parallel, VALU-bound)
//   dc centring + asymmetric peak AGC                                              (per-channel serial chains)
//   scale to +-30000, clip, store                                                  (sample-parallel)
//
// Reference behaviour reproduced (paths relative to the dsd-neo tree):
//   widen                     src/dsp/simd_widen.cpp:139-149
//   channel LPF               src/dsp/simd_fir_avx2.cpp:119-143 (per-output FMA chain: centre tap, then
//                             k = 0..centre-1 of fma(h[k], x[n-d] + x[n+d], acc)); block-edge sample
//                             replication src/dsp/simd_fir.cpp:66-85
//   power / squelch           src/dsp/demod_pipeline.cpp:926-945,1003-1020,1173-1190
//   discriminator             src/dsp/fsk_modem.c:23-35,89-164
//
// Arithmetic contract: every float op is an IEEE-754 binary32 op in the reference's order; the file is compiled
// with -ffp-contract=off and fused multiply-adds appear only where the reference's AVX2 unit has
// _mm256_fmadd_ps.  No MFMA: the symmetric pre-add makes the contraction a VALU (v_pk_add/v_pk_fma) job.
//
// Work decomposition (MI355X-first, not a translation of the reference's per-stream loop):
//   * one workgroup owns G channels for the whole call and walks time in tiles of TT = 256 samples;
//   * G*32 "filter" threads (half a wave per channel, R = 8 outputs per thread, sliding register windows fed
//     from a padded, bank-conflict-free LDS window) do widen + LPF + phase delta for tile i;
//   * one extra "recurrence" wave (lane = channel) runs the dc / peak recurrences of tile i-1 sample by sample
//     in the exact reference order while the filter threads are busy — the recurrences are only parallel
//     across channels, so they ride on an otherwise idle issue slot instead of a second kernel;
//   * the filter threads finish tile i-2 (30000/peak scaling, clip, coalesced store) before staging tile i.
//   HBM traffic is the algorithmic minimum: input read once (+ a 2*CENTER halo per tile from L2), output
//   written once.  Two barriers per tile; everything else stays in LDS / registers.
//
// Data layout in HBM: input [B][n] interleaved I/Q (2 B or 8 B per complex sample), output [B][n] f32, both
// channel-major so one channel's stream is contiguous (what the stream-read hook hands out).  Per-channel
// carried state: DDN_CARRY_LEN widened samples of FIR look-back + 5 words of modem state.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_atan2f.h"
#include "ddn_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define DDN_TT 256
#ifndef DDN_GROUP
#define DDN_GROUP 16 /* channels per workgroup */
#endif
#define DDN_R  4
#ifndef DDN_FE_EXTRA
#define DDN_FE_EXTRA 0 /* filter waves beyond the G / 2 that stage and finish tiles (sixteen-channel workgroups only): they take the
                          filter's work items and nothing else.  With 2 the workgroup has twelve waves, three on every SIMD: alone the
                          kernel is 1-2 % faster on clean C4FM and 6 % on noise (2.42 against 2.59 ms) - and the step 0.35 ms SLOWER
                          (8.17 against 7.82 ms; the mixed step 11.8 against 10.8): ten waves leave 176 registers free on two SIMDs of
                          every CU, which is where the previous call's decode kernels run beside the front end; with twelve they
                          run beside the loop instead and that costs the loop 0.27 ms */
#endif
#define DDN_FE_THREADS(G) ((G) * 32 + 128 + ((G) == 16 ? DDN_FE_EXTRA * 64 : 0))

// ------------------------------------------------------------------------------------------------------
// helpers

__device__ __forceinline__ f2
ddn_load_iq(const void* base, int fmt, size_t ch_stride_samples, int ch, long p) {
    if (fmt == DDN_IN_CU8) {
        const uint16_t* s = (const uint16_t*)base + (size_t)ch * ch_stride_samples + p;
        const uint16_t v = *s;
        const float inv = 1.0f / 127.5f;
        f2 r;
        r.x = ((float)(v & 0xFF) - 127.5f) * inv;
        r.y = ((float)(v >> 8) - 127.5f) * inv;
        return r;
    }
    const f2* s = (const f2*)base + (size_t)ch * ch_stride_samples + p;
    return *s;
}

// src/dsp/fsk_modem.c:23-35 + :89-94
__device__ __forceinline__ float
ddn_phase_delta(f2 cur, f2 prev) {
    const float re = cur.x * prev.x + cur.y * prev.y;
    const float im = cur.y * prev.x - cur.x * prev.y;
    if (re > 1.0e-7f && fabsf(im) <= (0.35f * re)) {
        const float x = im / re;
        const float x2 = x * x;
        return x * (1.0f + x2 * (-0.3333333333333333f + x2 * 0.2f));
    }
    return ddn_atan2f_fast(im, re);
}

// One sample of the peak-tracking AGC recurrence (src/dsp/fsk_modem.c:116-133) on the centred value c.
// Returns the peak to divide by; the division itself is done by the finishing threads.
__device__ __forceinline__ float
ddn_peak_step(float c, float& peak) {
    const float mag = fabsf(c);
    if (mag > 1.0e-7f) {
        if (peak <= 1.0e-7f) {
            peak = mag;
        } else if (mag > peak) {
            peak += 0.125f * (mag - peak);
        } else {
            peak += 0.00005f * (mag - peak);
        }
    }
    return (peak <= 1.0e-7f) ? 1.0f : peak;
}

// Common-case form with both rare guards (|c| <= 1e-7, peak <= 1e-7) hoisted out; the caller checks once per
// group whether a guard could have fired and replays the group through ddn_peak_step if so.
//   (mag > peak ? 0.125f : 0.00005f) * d == max(0.125f*d, 0.00005f*d)   with d = mag - peak:
// d > 0 makes the 0.125 product the larger one, d < 0 the 0.00005 product, d == 0 gives +0 for both — the same
// rounded product without a compare/select in the dependent chain.
__device__ __forceinline__ float
ddn_peak_step_fast(float c, float& peak) {
    const float d = fabsf(c) - peak;
    // (both products from one packed multiply: the peak wave's time is its instruction count - a wave that shares its SIMD with busy
    // filter waves is served about every 2.4th issue slot)
    const f2 k = {0.125f, 0.00005f}, dd = {d, d};
    const f2 p = k * dd;
    peak = peak + fmaxf(p.x, p.y);
    return peak;
}

// ------------------------------------------------------------------------------------------------------

struct DdnTapsK {
    float centre;
    float side[DDN_MAX_CENTER]; // side[k] multiplies x[n - (C-k)] + x[n + (C-k)]
};

struct TileDesc {
    long start;   // first sample of the tile (call-relative)
    long blk_start;
    long blk_end; // end of the reference block the tile belongs to
    int valid;    // samples in the tile (<= TT)
    int first;    // 1 = first tile of its block
};

// Tiles are enumerated block by block (tiles never straddle a reference block); the walk is incremental so no
// 64-bit division sits in the per-tile path.  Tiles past the end of a short last block have valid == 0.
struct TileWalk {
    long blk_start;
    int j;
};

__device__ __forceinline__ TileDesc
ddn_tile_next(TileWalk& w, const DdnFusedArgs& a) {
    TileDesc t;
    long be = w.blk_start + a.block_len;
    if (be > a.n) {
        be = a.n;
    }
    t.start = w.blk_start + (long)w.j * DDN_TT;
    t.blk_start = w.blk_start;
    t.blk_end = be;
    const long v = be - t.start;
    t.valid = (int)(v < 0 ? 0 : (v > DDN_TT ? DDN_TT : v));
    t.first = (w.j == 0);
    if (++w.j == a.tiles_per_block) {
        w.j = 0;
        w.blk_start += a.block_len;
    }
    return t;
}

// CENTER_T > 0: fully unrolled sliding-window FIR for that half-length; CENTER_T == 0: run-time half-length
// (taps in LDS, no register windows) for sample rates without an unrolled instance.
//
// Pipeline per tile index i (two barriers per iteration `it`):
//   phase A  filter threads: finish tile it-3 (scale, clip, store), stage the window of tile it
//   phase B  filter threads: LPF + phase delta of tile it           -> Fb[it % 3]
//            wave S1 (lane = channel): dc centring of tile it-1     -> Fb[(it-1) % 3] in place (centred value)
//            wave S2 (lane = channel): peak AGC recurrence, tile it-2 -> Pb[(it-2) % 2] (peak to divide by)
// A single wave issues roughly one VALU instruction per 5 cycles whatever the lane count, so the two
// recurrences are split over two waves to keep each under the filter threads' time per tile.
template <int CENTER_T, int G, bool SKIPZ, int FMT, bool SEGS = false>
__global__ __launch_bounds__(DDN_FE_THREADS(G), 3) void
k_front_end_fused(DdnFusedArgs a) {
    constexpr int TT = DDN_TT;
    constexpr int R = DDN_R;
    constexpr int CMAX = (CENTER_T > 0) ? CENTER_T : DDN_MAX_CENTER;
    constexpr int W = TT + 2 * CMAX;
    constexpr int WP = W + W / R + 1;
    constexpr int FS = TT + 4; // row stride of the tile buffers (floats), keeps 16-B alignment
    constexpr int NPRE = (W + 31) / 32; // window samples each filter thread stages per tile

    __shared__ f2 win[G][WP];
    __shared__ __attribute__((aligned(16))) float Fb[3][G][FS]; // raw phase delta -> centred value (in place)
    __shared__ __attribute__((aligned(16))) float Pb[2][G][FS]; // peak to divide by
    __shared__ f2 chan_last[2][G];                              // last LPF output of the previous tile
    __shared__ int tflag[3][G];
    __shared__ float gmin[3][G][TT / 8 + 1]; // min |centred| of each 8-sample group, written by S1 for S2's guard test
    __shared__ int next_item; // filter work-item counter of the current tile // per tile/channel: 1 = squelched (zeros + modem reset), 2 = first sample skipped
    constexpr int STAP = DDN_MAX_CENTER + 5; // (odd + 1 = even: every segment's tap row starts 8-byte aligned)
    static_assert(STAP % 2 == 0, "tap rows are read in 8-byte pairs");
    __shared__ __attribute__((aligned(8))) float stap[3 * STAP]; // one row of taps per segment (a plain batch uses row 0)
    extern __shared__ f2 ysq[]; // [1 or 2][G][256] first LPF outputs of a block (squelch builds only)

    // The two recurrence waves are the FIRST waves of the workgroup: VALU issue on a SIMD is arbitrated by age
    // (measured: a youngest-wave dependent chain runs 2x slower beside busy filter waves, an oldest-wave one at
    // full speed; s_setprio makes no difference), and they are the per-tile critical path.
    const int tid = threadIdx.x;
    const bool is_filter = tid >= 128;
    const int role = is_filter ? 0 : (tid < 64 ? 1 : 2);      // 0 filter, 1 = S1 (dc), 2 = S2 (peak)
    const int ft = tid - 128;                                  // filter-thread index
    const bool stager = is_filter && ft < G * 32;              // a filter thread with a share of the staging and finishing (the extra
                                                               // waves only take work items of the filter)
    const int g = is_filter ? (ft >> 5) : (tid & 63);          // channel slot
    const int u = ft & 31;
    const int lane64 = tid & 63;
    const int ch0 = blockIdx.x * G;
    const int nch = (a.n_channels - ch0) < G ? (a.n_channels - ch0) : G;
    const int C = (CENTER_T > 0) ? CENTER_T : a.center;
    const int ch = ch0 + (g < nch ? g : 0); // clamp: surplus slots recompute channel ch0, never store
    const bool ch_ok = g < nch;
    // segment of this thread's channel: its input / output arrays and its row inside them (a plain batch: segment 0 = the batch)
    const bool segs = SEGS && a.n_seg > 1; // (an instance of its own: the plain batch's kernel carries none of the look-ups)
    const int seg = segs ? ((ch >= a.seg_first1 ? 1 : 0) + ((a.n_seg > 2 && ch >= a.seg_first2) ? 1 : 0)) : 0;
    const int chl = ch - (seg == 0 ? 0 : (seg == 1 ? a.seg_first1 : a.seg_first2));
    const void* const in_s = segs ? (seg == 0 ? a.seg_in0 : (seg == 1 ? a.seg_in1 : a.seg_in2)) : a.in;
    float* const out_s = segs ? (seg == 0 ? a.seg_out0 : (seg == 1 ? a.seg_out1 : a.seg_out2)) : a.out;
    const long NT = a.n_tiles;

    // recurrence state (S1: dc / have_prev / squelched, S2: peak)
    float dc = 0.f, peak = 0.f;
    int have_prev = 0;
    int squelched = 0;
    if (role != 0 && ch_ok) {
        const DdnFskState s = a.state[ch];
        dc = s.dc_est;
        peak = s.peak_est;
        have_prev = s.have_prev;
        if (role == 1) {
            const f2 pv = {s.prev_i, s.prev_q};
            chan_last[0][g] = pv;
        }
    }
    if (tid <= C) {
        stap[tid] = a.taps_dev[tid];
        if (a.n_seg > 1) {
            stap[STAP + tid] = a.seg_taps1[tid];
        }
        if (a.n_seg > 2) {
            stap[2 * STAP + tid] = a.seg_taps2[tid];
        }
    }
    // raw samples of the NEXT tile, fetched while the current one is filtered (hides HBM latency)
    uint32_t pre_u[(FMT == DDN_IN_CU8) ? NPRE : 1];
    f2 pre_f[(FMT == DDN_IN_CF32) ? NPRE : 1];
    __syncthreads();

    // tile descriptors in flight: tq[0] = tile it+1 (prefetch), tq[1] = it, tq[2] = it-1, tq[3] = it-2, tq[4] = it-3
    TileWalk walk = {0, 0};
    TileDesc tq[5];
    for (int k = 0; k < 5; k++) {
        tq[k].start = 0;
        tq[k].blk_start = 0;
        tq[k].blk_end = 0;
        tq[k].valid = 0;
        tq[k].first = 0;
    }
    tq[0] = ddn_tile_next(walk, a); // tile 0
    long long tA = 0, tB = 0, tW = 0, tm0 = 0, tm1 = 0, tm2 = 0;
    for (long it = 0; it < NT + 3; it++) {
        tq[4] = tq[3];
        tq[3] = tq[2];
        tq[2] = tq[1];
        tq[1] = tq[0];
        tq[0] = ddn_tile_next(walk, a);
        if (a.dbg & 64) {
            tm0 = __builtin_readcyclecounter();
        }
        // ================= phase A: finish tile it-3, stage tile it =====================================
        if (stager) {
            if (it >= 3 && !(a.dbg & 4)) {
                const TileDesc t3 = tq[4];
                const int bf = (int)((it - 3) % 3);
                const int bp = (int)((it - 3) & 1);
                float* o = out_s + (size_t)chl * a.out_stride + t3.start;
                const bool al = ((((size_t)o) & 15) == 0);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int t = half * 128 + u * 4;
                    const f4 c4 = *(const f4*)&Fb[bf][g][t];
                    const f4 p4 = *(const f4*)&Pb[bp][g][t];
                    f4 y;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float v = c4[k] * (30000.0f / p4[k]);
                        v = (v > 32767.0f) ? 32767.0f : ((v < -32768.0f) ? -32768.0f : v);
                        y[k] = v;
                    }
                    if (ch_ok) {
                        if (al && t + 4 <= t3.valid) {
                            *(f4*)(o + t) = y;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                if (t + k < t3.valid) {
                                    o[t + k] = y[k];
                                }
                            }
                        }
                    }
                }
            }
            if (it < NT && !(a.dbg & 8)) {
                const int Weff = TT + 2 * C;
                if (it == 0) {
                    // first tile of the call: look-back comes from the carried history, fetched synchronously
                    const TileDesc tc = tq[1];
                    const f2* carry = a.carry + (size_t)ch * DDN_CARRY_LEN;
                    for (int i = u; i < Weff; i += 32) {
                        long p = tc.start - C + i;
                        f2 v;
                        if (p < 0) {
                            v = carry[DDN_CARRY_LEN + p];
                        } else {
                            if (p > tc.blk_end - 1) {
                                p = tc.blk_end - 1; // the reference replicates the block's last sample
                            }
                            v = ddn_load_iq(in_s, FMT, a.ch_stride, chl, p);
                        }
                        win[g][i + i / R] = v;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NPRE; k++) {
                        const int i = u + 32 * k;
                        if (i < Weff) {
                            f2 v;
                            if constexpr (FMT == DDN_IN_CU8) {
                                const uint32_t r = pre_u[k];
                                const float inv = 1.0f / 127.5f;
                                v.x = ((float)(r & 0xFF) - 127.5f) * inv;
                                v.y = ((float)(r >> 8) - 127.5f) * inv;
                            } else {
                                v = pre_f[k];
                            }
                            win[g][i + i / R] = v;
                        }
                    }
                }
            }
        }
        if (tid == 128) {
            next_item = 0;
        }
        if (a.dbg & 64) {
            tm1 = __builtin_readcyclecounter();
            tA += tm1 - tm0;
        }
        __syncthreads();
        if (a.dbg & 64) {
            tm2 = __builtin_readcyclecounter();
            tW += tm2 - tm1;
        }
        // ================= phase B ======================================================================
        if (role == 0) {
            // ---- filter threads: prefetch tile it+1, LPF + phase delta of tile it ------------------------
            if (stager && it + 1 < NT && !(a.dbg & 8)) {
                const TileDesc tn = tq[0];
                const int Weff = TT + 2 * C;
                if (tn.valid > 0) {
#pragma unroll
                    for (int k = 0; k < NPRE; k++) {
                        const int i = u + 32 * k;
                        long p = tn.start - C + i;
                        if (p > tn.blk_end - 1) {
                            p = tn.blk_end - 1;
                        }
                        if (i < Weff) {
                            if constexpr (FMT == DDN_IN_CU8) {
                                pre_u[k] = *((const uint16_t*)in_s + (size_t)chl * a.ch_stride + p);
                            } else {
                                pre_f[k] = *((const f2*)in_s + (size_t)chl * a.ch_stride + p);
                            }
                        }
                    }
                }
            }
            const TileDesc tc = tq[1];
            if (stager && it < NT && tc.valid <= 0 && u == 0) {
                chan_last[(it & 1) ^ 1][g] = chan_last[it & 1][g]; // surplus tile of a short last block
            }
            // Filter work is handed out dynamically: one item = one channel's tile on a full wave (64 lanes x R = 4
            // outputs).  Waves that share their SIMD with a recurrence wave simply come back for fewer items, so all
            // four SIMDs finish together.  (Results do not depend on which wave computes which channel.)
            // (round 6, DDN_FE_EXTRA = 2: two more waves come here for items only.  VALU issue on a SIMD goes to the oldest wave that can
            // issue, so the first waves of each SIMD take two items a tile and the last one what is left; capping the second items
            // per wave made no difference, the split 5 / 5 / 3 / 3 items over the SIMDs is what the item size allows.)
            while (it < NT && tc.valid > 0 && !(a.dbg & 2)) {
                int item = 0;
                if (lane64 == 0) {
                    item = atomicAdd(&next_item, 1);
                }
                item = __shfl(item, 0);
                if (item >= nch) {
                    break;
                }
                const int g = item;      // channel slot of this item
                const int u = lane64;    // position inside the tile: outputs u*R .. u*R+R-1
                // the item's tap row (one channel per item: uniform over the wave)
                const int ich = ch0 + item;
                const int irow = segs ? STAP * ((ich >= a.seg_first1 ? 1 : 0) + ((a.n_seg > 2 && ich >= a.seg_first2) ? 1 : 0)) : 0;
                const float* const stp = &stap[irow];
                f2 acc[R];
                const f2 zero = {0.0f, 0.0f};
                // a block shorter than taps_len samples goes to the reference's non-fused scalar unit
                // (src/dsp/simd_fir.cpp:303-306,350-356): separate multiply and add roundings
                const bool short_blk = (tc.blk_end - tc.blk_start) < (long)(2 * C + 1);
                if (short_blk) {
                    const f2* w = &win[g][0];
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        const int o = u * R + j + C;
                        acc[j] = zero + stp[C] * w[o + o / R];
                    }
                    for (int k = 0; k < C; k++) {
                        const float h = stp[k];
                        if (h == 0.0f) {
                            continue;
                        }
                        const int d = C - k;
#pragma unroll
                        for (int j = 0; j < R; j++) {
                            const int om = u * R + j + C - d;
                            const int op = u * R + j + C + d;
                            acc[j] = acc[j] + h * (w[om + om / R] + w[op + op / R]);
                        }
                    }
                } else if constexpr (CENTER_T > 0) {
#define PH(off) ((off) + (off) / R)
                    // Software-pipelined tap loop: the two window elements a step brings in are fetched from LDS
                    // PF steps ahead (queues qm/qp) and a scheduling barrier per step keeps the compiler from sinking
                    // those reads back next to their use, so no step waits on LDS latency.  Taps come from a uniform
                    // global pointer (scalar loads), not from 68 kernel-argument SGPRs.
                    constexpr int PF = 4;
                    const f2* w = &win[g][u * (R + 1)];
                    // The descending and the ascending window are read through two base registers the compiler cannot
                    // prove equal (opaque index copies; the pointers stay LDS pointers): with one base it fuses each step's pair of 8-byte reads into ds_read2_b64, which
                    // gfx950 services at 128 B/clk (8 LDS cycles) where two ds_read_b64 take 2 + 2 cycles at 256 B/clk —
                    // and at R = 4 the fused form makes the LDS, not the VALU, the busiest unit of the CU.
                    int olo = u * (R + 1), ohi = u * (R + 1), otap = irow;
                    asm volatile("" : "+v"(olo));
                    asm volatile("" : "+v"(ohi));
                    asm volatile("" : "+v"(otap));
                    const f2* wlo = &win[g][olo];
                    const f2* whi = &win[g][ohi];
                    const float* tp = &stap[otap]; // a base register + immediate offsets instead of one v_mov per tap
                    f2 xm[R], xp[R], qm[PF], qp[PF];
                    float qh[PF]; // taps ride the same prefetch queue (LDS broadcast reads of stap[])
                    const float hcs = stp[CENTER_T];
                    const f2 hc = {hcs, hcs};
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        acc[j] = __builtin_elementwise_fma(hc, w[PH(CENTER_T + j)], zero);
                        xm[j] = w[PH(j)];
                        xp[j] = w[PH(2 * CENTER_T + j)];
                    }
#pragma unroll
                    for (int q = 0; q < PF; q++) {
                        if (q + 1 < CENTER_T) {
                            qm[q] = wlo[PH(q + 1 + R - 1)];
                            qp[q] = whi[PH(2 * CENTER_T - (q + 1))];
                            qh[q] = tp[q + 1];
                        }
                    }
                    float h = tp[0];
                    float qh_odd = tp[PF + 1]; // second tap of the last 8-byte tap read (PF + 1 is odd: first use is step 0)
#pragma unroll
                    for (int k = 0; k < CENTER_T; k++) {
                        if (!SKIPZ || h != 0.0f) {
                            const f2 hh = {h, h};
                            f2 t[R];
#pragma unroll
                            for (int j = 0; j < R; j++) {
                                t[j] = xm[j] + xp[j];
                            }
#pragma unroll
                            for (int j = 0; j < R; j++) {
                                acc[j] = __builtin_elementwise_fma(hh, t[j], acc[j]);
                            }
                        }
                        if (k + 1 < CENTER_T) {
#pragma unroll
                            for (int j = 0; j < R - 1; j++) {
                                xm[j] = xm[j + 1];
                            }
                            xm[R - 1] = qm[0];
#pragma unroll
                            for (int j = R - 1; j > 0; j--) {
                                xp[j] = xp[j - 1];
                            }
                            xp[0] = qp[0];
                            h = qh[0];
#pragma unroll
                            for (int q = 0; q < PF - 1; q++) {
                                qm[q] = qm[q + 1];
                                qp[q] = qp[q + 1];
                                qh[q] = qh[q + 1];
                            }
                            if (k + 1 + PF < CENTER_T) {
                                qm[PF - 1] = wlo[PH(k + 1 + PF + R - 1)];
                                qp[PF - 1] = whi[PH(2 * CENTER_T - (k + 1 + PF))];
                                // taps are fetched two at a time (one ds_read_b64 every other step)
                                if (((k + 1 + PF) & 1) == 0) {
                                    const f2 t2 = *(const f2*)&tp[k + 1 + PF];
                                    qh[PF - 1] = t2.x;
                                    qh_odd = t2.y;
                                } else {
                                    qh[PF - 1] = qh_odd;
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#undef PH
                } else {
                    const f2* w = &win[g][0];
                    const f2 hc = {stp[C], stp[C]};
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        const int o = u * R + j + C;
                        acc[j] = __builtin_elementwise_fma(hc, w[o + o / R], zero);
                    }
                    for (int k = 0; k < C; k++) {
                        const float h = stp[k];
                        if (h == 0.0f) {
                            continue;
                        }
                        const int d = C - k;
                        const f2 hh = {h, h};
#pragma unroll
                        for (int j = 0; j < R; j++) {
                            const int om = u * R + j + C - d;
                            const int op = u * R + j + C + d;
                            acc[j] = __builtin_elementwise_fma(hh, w[om + om / R] + w[op + op / R], acc[j]);
                        }
                    }
                }
                // phase delta against the previous output (previous lane / previous tile)
                const int b = (int)(it & 1);
                const int bf = (int)(it % 3);
                f2 prev;
                prev.x = __shfl_up(acc[R - 1].x, 1);
                prev.y = __shfl_up(acc[R - 1].y, 1);
                if (u == 0) {
                    prev = chan_last[b][g];
                }
                const int lastv = tc.valid - 1;
                if (u == lastv / R) {
                    f2 yl = acc[0];
#pragma unroll
                    for (int j = 1; j < R; j++) {
                        if (j == lastv % R) {
                            yl = acc[j];
                        }
                    }
                    chan_last[b ^ 1][g] = yl;
                }
                f4 q0;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const float fq = ddn_phase_delta(acc[j], prev);
                    prev = acc[j];
                    q0[j] = fq;
                }
                *(f4*)&Fb[bf][g][u * R] = q0;
                if (a.squelch_on && tc.first) {
                    // one-tile blocks: every tile is a block's first, and S1 is still summing tile it-1's copy while
                    // this one is written, so the copies alternate between two halves
                    f2* yq = ysq + (a.tiles_per_block == 1 ? (int)(it & 1) * (G * 256) : 0);
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        yq[g * 256 + u * R + j] = acc[j];
                    }
                }
            }
        } else if (role == 1) {
            // ---- S1: dc centring (src/dsp/fsk_modem.c:98-105) of tile it-1, squelch gate, stream restart ----
            if (it >= 1 && it <= NT && ch_ok && g < G && !(a.dbg & 1)) {
                const TileDesc tp = tq[2];
                const int bf = (int)((it - 1) % 3);
                float* F = &Fb[bf][g][0];
                int flag = 0;
                if (a.squelch_on && tp.first && tp.valid > 0) {
                    // block power: first <=512 floats of the block's LPF output, sequential binary64 sums
                    // exactly like mean_power() (src/dsp/demod_pipeline.cpp:926-945)
                    const int len = (int)((tp.blk_end - tp.start) * 2 > 512 ? 512 : (tp.blk_end - tp.start) * 2);
                    const float* sq =
                        (const float*)&ysq[(a.tiles_per_block == 1 ? (int)((it - 1) & 1) * (G * 256) : 0) + g * 256];
                    double pw = 0.0, tot = 0.0;
                    for (int i = 0; i < len; i++) {
                        const double v = (double)sq[i];
                        tot += v;
                        pw += v * v;
                    }
                    const double dcc = (tot * tot) / (double)len;
                    double e = pw - dcc;
                    if (e < 0.0) {
                        e = 0.0;
                    }
                    const float chp = (float)(e / (double)len);
                    squelched = (chp < a.squelch_level) ? 1 : 0;
                }
                if (squelched) {
                    // zeros out + modem reset (src/dsp/demod_pipeline.cpp:1179-1184)
                    flag = 1;
                    have_prev = 0;
                    dc = 0.f;
                    for (int t = 0; t < tp.valid; t++) {
                        F[t] = 0.0f;
                    }
                } else {
                    int t = 0;
                    if (!have_prev && tp.valid > 0) {
                        // first sample of a (re)started stream: output 0, remember it (src/dsp/fsk_modem.c:148-153)
                        have_prev = 1;
                        flag = 2;
                        F[0] = 0.0f;
                        t = 1;
                    }
                    for (; t < tp.valid && (t & 3) != 0; t++) {
                        const float fr = F[t];
                        dc += 0.00025f * (fr - dc);
                        F[t] = fr - dc;
                    }
                    // 8 samples per trip; the next trip's LDS reads are issued ahead of the dependent chain
                    const long long tl0 = (a.dbg & 64) ? __builtin_readcyclecounter() : 0;
                    f4 fa = {0.f, 0.f, 0.f, 0.f}, fb = {0.f, 0.f, 0.f, 0.f};
                    if (t + 8 <= tp.valid) {
                        fa = *(const f4*)&F[t];
                        fb = *(const f4*)&F[t + 4];
                    }
                    for (; t + 8 <= tp.valid; t += 8) {
                        f4 na = fa, nb = fb;
                        if (t + 16 <= tp.valid) {
                            na = *(const f4*)&F[t + 8];
                            nb = *(const f4*)&F[t + 12];
                        }
                        f4 ca, cb;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            dc += 0.00025f * (fa[k] - dc);
                            ca[k] = fa[k] - dc;
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            dc += 0.00025f * (fb[k] - dc);
                            cb[k] = fb[k] - dc;
                        }
                        *(f4*)&F[t] = ca;
                        *(f4*)&F[t + 4] = cb;
                        // the peak wave's rare-guard test needs min |centred| of this group; this wave has the slack
                        gmin[bf][g][t >> 3] =
                            fminf(fminf(fminf(fabsf(ca[0]), fabsf(ca[1])), fminf(fabsf(ca[2]), fabsf(ca[3]))),
                                  fminf(fminf(fabsf(cb[0]), fabsf(cb[1])), fminf(fabsf(cb[2]), fabsf(cb[3]))));
                        fa = na;
                        fb = nb;
                    }
                    if (a.dbg & 64) {
                        tA += __builtin_readcyclecounter() - tl0;
                    }
                    for (; t < tp.valid; t++) {
                        const float fr = F[t];
                        dc += 0.00025f * (fr - dc);
                        F[t] = fr - dc;
                    }
                }
                tflag[bf][g] = flag;
            }
        } else {
            // ---- S2: asymmetric peak AGC recurrence (src/dsp/fsk_modem.c:116-133) of tile it-2 -------------
            if (it >= 2 && it <= NT + 1 && ch_ok && g < G && !(a.dbg & 1)) {
                const TileDesc tp = tq[3];
                const int bf = (int)((it - 2) % 3);
                const int bp = (int)((it - 2) & 1);
                const float* Cc = &Fb[bf][g][0];
                float* P = &Pb[bp][g][0];
                const int flag = tflag[bf][g];
                if (flag == 1) {
                    peak = 0.f;
                    for (int t = 0; t < tp.valid; t++) {
                        P[t] = 1.0f;
                    }
                } else {
                    int t = 0;
                    if (flag == 2) {
                        P[0] = 1.0f;
                        t = 1;
                    }
                    for (; t < tp.valid && (t & 3) != 0; t++) {
                        P[t] = ddn_peak_step(Cc[t], peak);
                    }
                    f4 fa = {0.f, 0.f, 0.f, 0.f}, fb = {0.f, 0.f, 0.f, 0.f};
                    if (t + 8 <= tp.valid) {
                        fa = *(const f4*)&Cc[t];
                        fb = *(const f4*)&Cc[t + 4];
                    }
                    for (; t + 8 <= tp.valid; t += 8) {
                        f4 na = fa, nb = fb;
                        if (t + 16 <= tp.valid) {
                            na = *(const f4*)&Cc[t + 8];
                            nb = *(const f4*)&Cc[t + 12];
                        }
                        const float pk0 = peak;
                        f4 pa, pb;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            pa[k] = ddn_peak_step_fast(fa[k], peak);
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            pb[k] = ddn_peak_step_fast(fb[k], peak);
                        }
                        // Did a rare guard fire in this group?  Every new peak lies between the old peak and
                        // |centred| (monotone rounding), so "old peak > 1e-7 and every |centred| > 1e-7" rules
                        // both guards out; non-finite values also replay (fmaxf drops NaNs).
                        const float gm = gmin[bf][g][t >> 3]; // min |centred| of this group, from the dc wave
                        if (!(fminf(gm, pk0) > 1.0e-7f) || !(peak <= 3.0e38f)) {
                            peak = pk0;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                pa[k] = ddn_peak_step(fa[k], peak);
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                pb[k] = ddn_peak_step(fb[k], peak);
                            }
                        }
                        *(f4*)&P[t] = pa;
                        *(f4*)&P[t + 4] = pb;
                        fa = na;
                        fb = nb;
                    }
                    for (; t < tp.valid; t++) {
                        P[t] = ddn_peak_step(Cc[t], peak);
                    }
                }
            }
        }
        if (a.dbg & 64) {
            tm0 = __builtin_readcyclecounter();
            tB += tm0 - tm2;
        }
        __syncthreads();
        if (a.dbg & 64) {
            tW += __builtin_readcyclecounter() - tm0;
        }
    }
    if ((a.dbg & 64) && a.dbg_out && blockIdx.x == 0 && (tid & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        long long* o = a.dbg_out + (is_filter ? (ft >> 6) : (10 + (tid >> 6))) * 4;
        o[0] = (hw >> 4) & 3;
        o[1] = tA;
        o[2] = tB;
        o[3] = tW;
    }

    if (stager && ch_ok && a.carry_out) {
        // next call's FIR look-back: the channel's last DDN_CARRY_LEN widened samples.  Only this half-wave ever reads
        // or writes this channel's row (it read it at tile 0), so no other ordering is needed.
        f2* c = a.carry_out + (size_t)ch * DDN_CARRY_LEN;
        for (int i = u; i < DDN_CARRY_LEN; i += 32) {
            c[i] = ddn_load_iq(in_s, FMT, a.ch_stride, chl, a.n - DDN_CARRY_LEN + i);
        }
    }
    if (role == 1 && ch_ok && g < G && a.n > 0) {
        const f2 yl = chan_last[(int)(NT & 1)][g];
        DdnFskState* s = &a.state[ch];
        s->prev_i = have_prev ? yl.x : 0.f;
        s->prev_q = have_prev ? yl.y : 0.f;
        s->have_prev = have_prev;
        s->dc_est = dc;
    }
    if (role == 2 && ch_ok && g < G && a.n > 0) {
        a.state[ch].peak_est = peak;
    }
}

// FIR history carry: the last DDN_CARRY_LEN widened input samples of each channel, for the next call.
__global__ void
k_carry_update(const void* in, int in_fmt, size_t ch_stride, long n, f2* carry) {
    const int ch = blockIdx.x;
    const int i = threadIdx.x; // 0..DDN_CARRY_LEN-1
    f2* c = carry + (size_t)ch * DDN_CARRY_LEN;
    const long p = n - DDN_CARRY_LEN + i;
    f2 v;
    if (p >= 0) {
        v = ddn_load_iq(in, in_fmt, ch_stride, ch, p);
    } else {
        v = c[i + n]; // shift older history down
    }
    __syncthreads();
    c[i] = v;
}

__global__ void
k_zero_u32(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        p[i] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------------
// launchers (called from ddn_api.cpp)

template <int CENTER_T, int G, int FMT>
static hipError_t
launch_fused_t(const DdnFusedArgs& a, const DdnTapsK& tp, bool has_zero, hipStream_t st) {
    dim3 grid((unsigned)((a.n_channels + G - 1) / G));
    dim3 block(DDN_FE_THREADS(G));
    // squelch builds: the block's first 256 LPF outputs, twice when a block is a single tile (see the filter's store)
    const size_t dyn = a.squelch_on ? (size_t)G * 256 * sizeof(f2) * (a.tiles_per_block == 1 ? 2 : 1) : 0;
    if (dyn) {
        const void* fn = has_zero ? (const void*)k_front_end_fused<CENTER_T, G, true, FMT>
                                  : (const void*)k_front_end_fused<CENTER_T, G, false, FMT>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) {
            return e;
        }
    }
    if (a.n_seg > 1) {
        if (dyn) {
            return hipErrorInvalidValue; // (the squelch route has no segmented form)
        }
        if (has_zero) {
            hipLaunchKernelGGL((k_front_end_fused<CENTER_T, G, true, FMT, true>), grid, block, 0, st, a);
        } else {
            hipLaunchKernelGGL((k_front_end_fused<CENTER_T, G, false, FMT, true>), grid, block, 0, st, a);
        }
        return hipGetLastError();
    }
    if (has_zero) {
        hipLaunchKernelGGL((k_front_end_fused<CENTER_T, G, true, FMT>), grid, block, dyn, st, a);
    } else {
        hipLaunchKernelGGL((k_front_end_fused<CENTER_T, G, false, FMT>), grid, block, dyn, st, a);
    }
    return hipGetLastError();
}

template <int G, int FMT>
static hipError_t
launch_fused_c(const DdnFusedArgs& a, const DdnTapsK& tp, bool has_zero, hipStream_t st) {
    switch (a.center) {
        case 67: return launch_fused_t<67, G, FMT>(a, tp, has_zero, st);
        case 33: return launch_fused_t<33, G, FMT>(a, tp, has_zero, st);
        default: return launch_fused_t<0, G, FMT>(a, tp, true, st);
    }
}

// group: channels per workgroup, 8 / 16, or 0 = by batch size.  A workgroup walks its channels' whole call, so the grid is
// n_channels / G workgroups: below ~2048 channels the 16-channel shape leaves CUs idle and the 8-channel one (twice the workgroups, same
// work per thread) wins - when the launch has the device to itself.  (round 6) A caller that runs several front ends side by side (the
// mixed chain) asks for 16: a workgroup's time does not depend on how many of its channel slots are filled (the two recurrence waves
// run lane = channel), so 8-channel workgroups of three launches fill the CUs twice over (measured: 6.3 ms for three groups of 1365
// where one launch of 4096 takes 2.1).
extern "C" hipError_t
ddn_dev_launch_fused_ex(const DdnFusedArgs* a, bool has_zero, int group, hipStream_t st) {
    DdnTapsK tp = {}; // (the kernels take their taps from a->taps_dev; the structure only selects the instance)
    if (a->squelch_on) {
        // squelch builds keep the first 256 LPF outputs of each block in LDS for the block-power sum: use the
        // 8-channel workgroup so everything still fits in 160 KB
        if (a->in_fmt == DDN_IN_CU8) {
            return launch_fused_c<8, DDN_IN_CU8>(*a, tp, has_zero, st);
        }
        return launch_fused_c<8, DDN_IN_CF32>(*a, tp, has_zero, st);
    }
    const bool small = group == 8 || (group != 16 && a->n_channels <= 2048);
    if (small) {
        if (a->in_fmt == DDN_IN_CU8) {
            return launch_fused_c<8, DDN_IN_CU8>(*a, tp, has_zero, st);
        }
        return launch_fused_c<8, DDN_IN_CF32>(*a, tp, has_zero, st);
    }
    if (a->in_fmt == DDN_IN_CU8) {
        return launch_fused_c<DDN_GROUP, DDN_IN_CU8>(*a, tp, has_zero, st);
    }
    return launch_fused_c<DDN_GROUP, DDN_IN_CF32>(*a, tp, has_zero, st);
}

extern "C" hipError_t
ddn_dev_launch_fused(const DdnFusedArgs* a, const float* taps_host, int group, hipStream_t st) {
    bool has_zero = false;
    for (int k = 0; k < a->center && k < DDN_MAX_CENTER; k++) {
        has_zero = has_zero || taps_host[k] == 0.0f;
    }
    return ddn_dev_launch_fused_ex(a, has_zero, group, st);
}

extern "C" hipError_t
ddn_dev_launch_carry(const void* in, int in_fmt, size_t ch_stride, long n, void* carry, int n_channels,
                     hipStream_t st) {
    hipLaunchKernelGGL(k_carry_update, dim3((unsigned)n_channels), dim3(DDN_CARRY_LEN), 0, st, in, in_fmt, ch_stride,
                       n, (f2*)carry);
    return hipGetLastError();
}

extern "C" hipError_t
ddn_dev_zero(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n);
    return hipGetLastError();
}
