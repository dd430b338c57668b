// ddn_p25p2_seq.hip - P25 Phase 2 above the burst layer: what processP2() does with the 700 dibits behind a sync
// (src/protocol/p25/phase2/p25p2_frame.c:1760-1798), batched over channels x groups.
//
//   p2_dibit_buffer()            :354-370    700 dibits = 1400 bits + metrics; bits 1400..1439 of the fourth timeslot (its ISCH) are
//                                            never captured: p2bit stays 0 there, p25p2_reliability_for_abs_bit() answers 0
//   p25p2_process_isch()         :708-745    the four ISCH words -> state->p2_scramble_offset (only a channel-1 I-ISCH moves it)
//   currentslot                  :1774-1779  offset % 2, toggled behind every timeslot (p25p2_duid_post_timeslot(), :1690-1695)
//   process_Frame_Scramble()     :372-392    xbit[i] = bit[i] ^ sequence[i + 20 + 360 * offset]  (ddn_dev_p25p2_descramble)
//   p25p2_process_duid()         :1742-1760  per timeslot: DUID -> dispatch (:1580-1640); an unknown DUID counts, the second one ends
//                                            the group and zeroes both 4V counters (p25p2_duid_should_abort(), :1642-1655)
//   valid site                   :1455-1459  4V / 2V / scrambled SACCH, FACCH, LCCH need wacn, sysid, cc neither 0 nor all ones
//   ESS                          :902-925    a 4V burst files bits 148..171 under fourv_counter[slot] (counter 0 clears the four),
//                                :1377-1411  a 2V burst decodes them with its own ESS-A and zeroes the counter
//
// GPU shape.  The sequencing is a handful of integer decisions per timeslot, serial along a channel's groups (the offset, the 4V
// counters and the ESS-B fragments are carried): one lane per channel walks its groups (k_p2_sequence) over DUID / I-ISCH values the
// burst layer's kernels computed for every timeslot at once, and files each timeslot under the decoder it needs; the decoders
// (RS(63,35) with ranked retries, RS(44,16), the voice unpack) then run over dense lists of exactly those timeslots, and a scatter
// pass writes the per-timeslot results.  The ESS-B fragments are de-scrambled bits, which exist only after the sequencing fixed the
// offsets: the sequencing lane records where each fragment comes from (a row of this call, zeros, or the carried state) and the
// gather pass materialises them.
#include <hip/hip_runtime.h>

#include "ddn_device.h"
#include "ddn_p25p2_seq.h"

namespace {

// [C][G][1400] -> timeslot rows [C * G * 4][360], the uncaptured tail zero
__global__ __launch_bounds__(256) void
k_p2_rows(const uint8_t* __restrict__ bits1400, const int16_t* __restrict__ llr1400, size_t n_groups_total, uint8_t* __restrict__ rb,
          int16_t* __restrict__ rl) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_groups_total * 1440) {
        return;
    }
    const size_t g = i / 1440;
    const int k = (int)(i - g * 1440);
    const bool in = k < 1400;
    rb[i] = in ? bits1400[g * 1400 + k] : (uint8_t)0;
    rl[i] = in ? llr1400[g * 1400 + k] : (int16_t)0;
}

__device__ __forceinline__ int
action_of_duid(int d) {
    switch (d) {
        case 0: return DDN_P2_4V;
        case 6: return DDN_P2_2V;
        case 3: return DDN_P2_SACCH_S;
        case 12: return DDN_P2_SACCH_C;
        case 15: return DDN_P2_FACCH_C;
        case 9: return DDN_P2_FACCH_S;
        case 13: return DDN_P2_LCCH_C;
        case 4: return DDN_P2_LCCH_S;
        default: return DDN_P2_ERR;
    }
}

__global__ __launch_bounds__(64) void
k_p2_sequence(const int32_t* __restrict__ duid, const int32_t* __restrict__ isch, int n_channels, int n_groups,
              const int32_t* __restrict__ groups_of,
              const uint64_t* __restrict__ seed44, ddn_p25p2_seq_state* __restrict__ state, int32_t* __restrict__ info, int32_t* __restrict__ row_off,
              int32_t* __restrict__ seq_of, int32_t* __restrict__ counts, int32_t* __restrict__ list, size_t n_rows,
              int32_t* __restrict__ ess_src, int32_t* __restrict__ final_src) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_channels) {
        return;
    }
    const uint64_t seed = seed44[c];
    const uint32_t wacn = (uint32_t)((seed >> 24) & 0xFFFFF), sysid = (uint32_t)((seed >> 12) & 0xFFF), cc = (uint32_t)(seed & 0xFFF);
    const bool valid = wacn != 0 && cc != 0 && sysid != 0 && wacn != 0xFFFFF && cc != 0xFFF && sysid != 0xFFF;
    // the carried state is the caller's memory: a stale or never-initialised block must not index out of the private arrays below
    // (offsets the I-ISCH rule can produce are 0..12; the 4V counter runs 0..3)
    int off = min(max(state[c].offset, 0), 12);
    int fourv[2] = {state[c].fourv[0] & 3, state[c].fourv[1] & 3};
    int src[2][4];                                            // where each ESS-B fragment lives: row >= 0, -1 zeros, -2 - j carried fragment j
    for (int s = 0; s < 2; s++) {
        for (int j = 0; j < 4; j++) {
            src[s][j] = -2 - j;
        }
    }
    const int held = groups_of ? min(max(groups_of[c], 0), n_groups) : n_groups;
    for (int g = held; g < n_groups; g++) {                   // places without a group: reported "not reached"
        for (int ts = 0; ts < 4; ts++) {
            const size_t row = ((size_t)c * n_groups + g) * 4 + ts;
            int32_t* o = info + row * 8;
            o[0] = -3;
            o[1] = -2;
            o[2] = off;
            o[3] = -1;
            o[4] = DDN_P2_NONE;
            o[5] = o[6] = o[7] = 0;
            row_off[row] = 0;
            seq_of[row] = c;
        }
    }
    for (int g = 0; g < held; g++) {
        const size_t row0 = ((size_t)c * n_groups + g) * 4;
        for (int f = 0; f < 4; f++) {
            const int v = isch[row0 + f];
            if (v > -1 && ((v >> 5) & 3) == 1) {
                const int loc = (v >> 3) & 3;
                if (loc == 0) {
                    off = 12 - f;
                } else if (loc == 1) {
                    off = 4 - f;
                } else if (loc == 2) {
                    off = 8 - f;
                }
            }
        }
        int slot = (off % 2) != 0 ? 1 : 0; // any non-zero remainder is slot 1 (p25p2_frame.c:1774-1779)
        int errs = 0;
        bool dead = false;
        for (int ts = 0; ts < 4; ts++) {
            const size_t row = row0 + ts;
            int32_t* o = info + row * 8;
            int d = -3, act = DDN_P2_NONE, fv = 0, sl = -1;
            int cls = -1;
            if (!dead) {
                d = duid[row];
                sl = slot;
                act = action_of_duid(d);
                if (!valid && (act == DDN_P2_4V || act == DDN_P2_2V || act == DDN_P2_SACCH_S || act == DDN_P2_FACCH_S || act == DDN_P2_LCCH_S)) {
                    act = DDN_P2_NOSITE;
                }
                if (act == DDN_P2_ERR) {
                    errs++;
                    if (errs > 1) {
                        fourv[0] = fourv[1] = 0;
                        dead = true;
                    }
                } else if (act == DDN_P2_4V) {
                    fv = fourv[slot];
                    if (fv == 0) {
                        src[slot][0] = src[slot][1] = src[slot][2] = src[slot][3] = -1;
                    }
                    src[slot][fv] = (int)row;
                    fourv[slot] = (fv + 1) & 3;
                    cls = 2;
                } else if (act == DDN_P2_2V) {
                    fv = fourv[slot];
                    for (int j = 0; j < 4; j++) {
                        ess_src[row * 4 + j] = src[slot][j];
                    }
                    fourv[slot] = 0;
                    cls = 3;
                } else if (act == DDN_P2_FACCH_C || act == DDN_P2_FACCH_S) {
                    cls = 0;
                } else if (act != DDN_P2_NOSITE) {
                    cls = 1;
                }
                if (!dead) {
                    slot ^= 1;
                }
            }
            o[0] = d;
            o[1] = isch[row];
            o[2] = off;
            o[3] = sl;
            o[4] = act;
            o[5] = 0;
            o[6] = fv;
            o[7] = 0;
            row_off[row] = off + ts;
            seq_of[row] = c;                                  // the de-scrambler's sequence row
            // a place in its decoder's list: one atomic per class and wavefront step (the lanes of a class share it by rank); the order
            // inside a list is whatever the wavefronts made it - results go back by row
            for (int k = 0; k < 4; k++) {
                const unsigned long long m = __ballot(cls == k);
                if (m == 0 || cls != k) {
                    continue;
                }
                const int leader = __ffsll((long long)m) - 1;
                int base = 0;
                if ((int)threadIdx.x == leader) {
                    base = atomicAdd(&counts[k], __popcll(m));
                }
                base = __shfl(base, leader);
                const int pos = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
                list[(size_t)k * n_rows + pos] = (int32_t)row;
            }
        }
    }
    state[c].offset = off;
    state[c].fourv[0] = fourv[0];
    state[c].fourv[1] = fourv[1];
    for (int s = 0; s < 2; s++) {
        for (int j = 0; j < 4; j++) {
            final_src[(c * 2 + s) * 4 + j] = src[s][j];
        }
    }
}

__device__ __forceinline__ void
ess_fragment(int src, int j, const uint8_t* xb, const int16_t* xl, const ddn_p25p2_seq_state* st, int slot, int k, uint8_t* b, int16_t* l) {
    if (src >= 0) {
        *b = xb[(size_t)src * 360 + 148 + k];
        *l = xl[(size_t)src * 360 + 148 + k];
    } else if (src == -1) {
        *b = 0;
        *l = 0;
    } else {
        *b = st->ess_b[slot][24 * j + k];
        *l = st->ess_b_llr[slot][24 * j + k];
    }
}

// one workgroup per listed timeslot: its decoder's dense input row
__global__ __launch_bounds__(128) void
k_p2_gather(int cls, const int32_t* __restrict__ list, const int32_t* __restrict__ info, const uint8_t* __restrict__ rb,
            const int16_t* __restrict__ rl, const uint8_t* __restrict__ xb, const int16_t* __restrict__ xl, uint8_t* __restrict__ db,
            int16_t* __restrict__ dl, const int32_t* __restrict__ ess_src, const ddn_p25p2_seq_state* __restrict__ state, int rows_per_channel,
            uint8_t* __restrict__ ess_pl, int16_t* __restrict__ ess_pll, uint8_t* __restrict__ ess_pa, int16_t* __restrict__ ess_pal) {
    const int p = blockIdx.x;
    const size_t row = (size_t)list[p];
    const int act = info[row * 8 + 4];
    const bool scr = !(act == DDN_P2_SACCH_C || act == DDN_P2_FACCH_C || act == DDN_P2_LCCH_C);
    const uint8_t* sb = (scr ? xb : rb) + row * 360;
    const int16_t* sl = (scr ? xl : rl) + row * 360;
    for (int k = threadIdx.x; k < 360; k += 128) {
        db[(size_t)p * 360 + k] = sb[k];
        dl[(size_t)p * 360 + k] = sl[k];
    }
    if (cls == 3) {
        const int slot = info[row * 8 + 3];
        const ddn_p25p2_seq_state* st = state + row / (size_t)rows_per_channel;
        for (int k = threadIdx.x; k < 96; k += 128) {
            const int j = k / 24;
            ess_fragment(ess_src[row * 4 + j], j, xb, xl, st, slot, k - 24 * j, ess_pl + (size_t)p * 96 + k, ess_pll + (size_t)p * 96 + k);
        }
        for (int k = threadIdx.x; k < 168; k += 128) {
            const int at = k < 96 ? 148 + k : 246 + (k - 96);
            ess_pa[(size_t)p * 168 + k] = sb[at];
            ess_pal[(size_t)p * 168 + k] = sl[at];
        }
    }
}

// the carried ESS-B fragments after the call (fragments still "carried" stay as they are)
__global__ __launch_bounds__(64) void
k_p2_state_ess(const int32_t* __restrict__ final_src, const uint8_t* __restrict__ xb, const int16_t* __restrict__ xl, int n_channels,
               ddn_p25p2_seq_state* __restrict__ state) {
    const int cs = blockIdx.x;                                // channel * 2 + slot
    const int c = cs >> 1, s = cs & 1;
    for (int k = threadIdx.x; k < 96; k += 64) {
        const int j = k / 24;
        const int src = final_src[cs * 4 + j];
        if (src >= -1) {
            uint8_t b;
            int16_t l;
            ess_fragment(src, j, xb, xl, state + c, s, k - 24 * j, &b, &l);
            state[c].ess_b[s][k] = b;
            state[c].ess_b_llr[s][k] = l;
        }
    }
}

// dense decoder outputs -> the per-timeslot result arrays
__global__ __launch_bounds__(128) void
k_p2_scatter(int cls, const int32_t* __restrict__ list, int32_t* __restrict__ info, const uint8_t* __restrict__ x_payload, int n_pl,
             const int32_t* __restrict__ ec, const uint8_t* __restrict__ used, const uint8_t* __restrict__ c12, const uint8_t* __restrict__ c16,
             const uint8_t* __restrict__ fr, const uint8_t* __restrict__ rel, int frame_count, const uint8_t* __restrict__ ess_out,
             uint8_t* __restrict__ o_payload, uint8_t* __restrict__ o_fr, uint8_t* __restrict__ o_rel, uint8_t* __restrict__ o_ess) {
    const int p = blockIdx.x;
    const size_t row = (size_t)list[p];
    if (cls <= 1) {
        for (int k = threadIdx.x; k < n_pl; k += 128) {
            o_payload[row * 180 + k] = x_payload[(size_t)p * n_pl + k];
        }
        if (threadIdx.x == 0) {
            info[row * 8 + 5] = ec[p];
            info[row * 8 + 7] = (used[p] ? 1 : 0) | (c12[p] ? 2 : 0) | ((c16 && c16[p]) ? 4 : 0);
        }
    } else {
        const int n = frame_count * 96;
        for (int k = threadIdx.x; k < n; k += 128) {
            o_fr[row * 384 + k] = fr[(size_t)p * n + k];
            o_rel[row * 384 + k] = rel[(size_t)p * n + k];
        }
        if (cls == 3) {
            for (int k = threadIdx.x; k < 96; k += 128) {
                o_ess[row * 96 + k] = ess_out[(size_t)p * 96 + k];
            }
            if (threadIdx.x == 0) {
                info[row * 8 + 5] = ec[p];
                info[row * 8 + 7] = (used[p] ? 1 : 0) | (ec[p] >= 0 ? 8 : 0);
            }
        }
    }
}

// ---- the dibit-level sync cut -----------------------------------------------------------------------------------------------------
// One wavefront per channel.  A search position is independent of its neighbours (an exact 20-dibit compare), so the wavefront tests
// 64 candidate sync ends at a time and takes the first; only the choice "which sync, then skip 700 dibits" is serial.
__device__ __forceinline__ int
sync_at(const uint8_t* d, int e) {                           // sync ending at dibit e: 1 = P25P2_SYNC, 2 = its inverse, 0 = neither
    const uint64_t want = 0x575D57F7FFull;                    // "11131131111333133333" as 20 dibits, first dibit on top
    uint64_t w = 0;
    for (int k = 0; k < 20; k++) {
        w = (w << 2) | (uint64_t)(d[e - 19 + k] & 3);
    }
    return w == want ? 1 : (w == (want ^ 0xAAAAAAAAAAull) ? 2 : 0);
}

__global__ __launch_bounds__(64) void
k_p2_sync_cut(const uint8_t* __restrict__ dibits, int n, size_t stride, const int32_t* __restrict__ cursor_in, int max_groups,
              int32_t* __restrict__ n_groups, int32_t* __restrict__ group_pos, int32_t* __restrict__ cursor_out) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* d = dibits + (size_t)c * stride;
    int start = cursor_in ? max(cursor_in[c], 0) : 0;        // the search window fills from here
    int count = 0;
    int out = -1;
    while (out < 0) {
        int found = -1, kind = 0;
        for (int base = start + 19; base < n && found < 0; base += 64) {
            const int e = base + lane;
            const int k = e < n ? sync_at(d, e) : 0;
            const unsigned long long m = __ballot(k != 0);
            if (m) {
                const int first = __ffsll((long long)m) - 1;
                found = base + first;
                kind = __shfl(k, first);
            }
        }
        if (found < 0) {
            out = max(start, n - 19);                         // nothing open: the last 19 dibits may begin a sync
        } else if (found + 700 >= n || count >= max_groups) {
            out = found - 19;                                 // the group is not all here (or has no place): find this sync again
        } else {
            if (lane == 0) {
                group_pos[(size_t)c * max_groups + count] = (kind == 2 ? -(found + 1) - 1 : found + 1);   // inverted: -(pos) - 1
            }
            count++;
            start = found + 701;
        }
    }
    if (lane == 0) {
        n_groups[c] = count;
        cursor_out[c] = min(out, n);
    }
}

// a group's 700 dibits -> 1400 bits + metrics (first bit = the dibit's high bit), through invert_dibit() behind an inverted sync
__global__ __launch_bounds__(256) void
k_p2_cut_copy(const uint8_t* __restrict__ dibits, const int16_t* __restrict__ llr2, size_t stride, int max_groups,
              const int32_t* __restrict__ n_groups, int32_t* __restrict__ group_pos, uint8_t* __restrict__ bits1400, int16_t* __restrict__ llr1400) {
    const int c = blockIdx.y, g = blockIdx.x;
    if (g >= n_groups[c]) {
        return;
    }
    const size_t gi = (size_t)c * max_groups + g;
    int pos = group_pos[gi];
    const bool inv = pos < 0;
    if (inv) {
        pos = -(pos + 1);
    }
    for (int k = threadIdx.x; k < 700; k += 256) {
        const size_t at = (size_t)c * stride + pos + k;
        const int v = (dibits[at] & 3) ^ (inv ? 2 : 0);
        const int16_t l0 = llr2[at * 2], l1 = llr2[at * 2 + 1];
        bits1400[gi * 1400 + 2 * k] = (uint8_t)(v >> 1);
        bits1400[gi * 1400 + 2 * k + 1] = (uint8_t)(v & 1);
        llr1400[gi * 1400 + 2 * k] = inv ? (int16_t)(l0 == -32768 ? 32767 : -l0) : l0;
        llr1400[gi * 1400 + 2 * k + 1] = l1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && inv) {
        group_pos[gi] = pos;                                  // reported positions are plain; the polarity is in the copy
    }
}

} // namespace

hipError_t
ddn_dev_p2_rows(const uint8_t* bits1400, const int16_t* llr1400, size_t n_groups_total, uint8_t* rb, int16_t* rl, hipStream_t st) {
    const size_t n = n_groups_total * 1440;
    hipLaunchKernelGGL(k_p2_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bits1400, llr1400, n_groups_total, rb, rl);
    return hipGetLastError();
}

// the 700 dibits behind every sync the symbol-rate loop marked (ddn_cq_rx, protocol Phase 2: flag bit 1 on the sync's last symbol, the
// in-frame records behind it already polarity-corrected with their soft decisions): p2_dibit_buffer()'s p2bit / p2llr
// (src/protocol/p25/phase2/p25p2_frame.c:352-370), one workgroup per (group, channel)
__global__ __launch_bounds__(256) void
k_p2_cut_records(const uint8_t* __restrict__ rec, size_t stride, const int32_t* __restrict__ sync_pos, const int32_t* __restrict__ n_sync,
                 int max_groups, uint8_t* __restrict__ bits1400, int16_t* __restrict__ llr1400) {
    const int c = blockIdx.y, g = blockIdx.x;
    if (g >= n_sync[c]) {
        return;
    }
    const size_t gi = (size_t)c * max_groups + g;
    const int p = sync_pos[gi];
    for (int k = threadIdx.x; k < 700; k += 256) {
        const uint8_t* q = rec + ((size_t)c * stride + (size_t)(p + 1 + k)) * 10;
        const int d = q[0] & 3;
        bits1400[gi * 1400 + 2 * k] = (uint8_t)(d >> 1);
        bits1400[gi * 1400 + 2 * k + 1] = (uint8_t)(d & 1);
        llr1400[gi * 1400 + 2 * k] = (int16_t)((uint16_t)q[2] | ((uint16_t)q[3] << 8));
        llr1400[gi * 1400 + 2 * k + 1] = (int16_t)((uint16_t)q[4] | ((uint16_t)q[5] << 8));
    }
}

// Voice of a call's timeslots by logical channel, in air order: talk path tp = channel * 2 + slot takes the 4 (4V) / 2 (2V) AMBE frames
// of every timeslot the sequencing filed under its slot (process_4V / process_2V hand them to the vocoder one by one,
// src/protocol/p25/phase2/p25p2_frame.c:1029-1047,1435-1460).  src[tp][k] = row * 4 + frame of the k-th frame, -1 beyond the count.
__global__ void
k_p2_voice_index(const int32_t* __restrict__ info, const int32_t* __restrict__ groups_of, int n_channels, int n_groups, int cap,
                 int32_t* __restrict__ src, int32_t* __restrict__ count) {
    const int tp = blockIdx.x * blockDim.x + threadIdx.x;
    if (tp >= 2 * n_channels) {
        return;
    }
    const int c = tp >> 1, slot = tp & 1;
    const int held = groups_of ? min(max(groups_of[c], 0), n_groups) : n_groups;
    int k = 0;
    for (int g = 0; g < held; g++) {
        for (int ts = 0; ts < 4; ts++) {
            const int row = (c * n_groups + g) * 4 + ts;
            const int32_t* i8 = info + (size_t)row * 8;
            const int act = i8[4];
            if (i8[3] != slot || (act != DDN_P2_4V && act != DDN_P2_2V)) {
                continue;
            }
            const int nf = act == DDN_P2_4V ? 4 : 2;
            for (int f = 0; f < nf && k < cap; f++) {
                src[(size_t)tp * cap + k++] = row * 4 + f;
            }
        }
    }
    count[tp] = k;
    for (; k < cap; k++) {
        src[(size_t)tp * cap + k] = -1;
    }
}

__global__ __launch_bounds__(64) void
k_p2_voice_gather(const int32_t* __restrict__ src, size_t n_slots, const uint8_t* __restrict__ fr, const uint8_t* __restrict__ rel,
                  uint8_t* __restrict__ o_fr, uint8_t* __restrict__ o_rel, uint8_t* __restrict__ skip) {
    const size_t i = blockIdx.x;
    if (i >= n_slots) {
        return;
    }
    const int s = src[i];
    for (int k = threadIdx.x; k < 96; k += 64) {
        o_fr[i * 96 + k] = s >= 0 ? fr[(size_t)s * 96 + k] : (uint8_t)0;
        o_rel[i * 96 + k] = s >= 0 ? rel[(size_t)s * 96 + k] : (uint8_t)0;
    }
    if (threadIdx.x == 0) {
        skip[i] = s >= 0 ? 0 : 1;
    }
}

hipError_t
ddn_dev_p2_voice_gather(const int32_t* info, const int32_t* groups_of, int n_channels, int n_groups, int cap, const uint8_t* fr, const uint8_t* rel,
                        int32_t* src, int32_t* count, uint8_t* o_fr, uint8_t* o_rel, uint8_t* skip, hipStream_t st) {
    if (n_channels <= 0 || cap <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_voice_index, dim3((unsigned)((2 * n_channels + 63) / 64)), dim3(64), 0, st, info, groups_of, n_channels, n_groups, cap, src,
                       count);
    hipLaunchKernelGGL(k_p2_voice_gather, dim3((unsigned)((size_t)2 * n_channels * cap)), dim3(64), 0, st, src, (size_t)2 * n_channels * cap, fr, rel,
                       o_fr, o_rel, skip);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_cut_records(const uint8_t* rec, size_t stride, const int32_t* sync_pos, const int32_t* n_sync, int n_channels, int max_groups,
                       uint8_t* bits1400, int16_t* llr1400, hipStream_t st) {
    if (n_channels <= 0 || max_groups <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_cut_records, dim3((unsigned)max_groups, (unsigned)n_channels), dim3(256), 0, st, rec, stride, sync_pos, n_sync, max_groups,
                       bits1400, llr1400);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_sequence(const int32_t* duid, const int32_t* isch, int n_channels, int n_groups, const int32_t* groups_of, const uint64_t* seed44,
                    ddn_p25p2_seq_state* state,
                    int32_t* info, int32_t* row_off, int32_t* seq_of, int32_t* counts, int32_t* list, int32_t* ess_src, int32_t* final_src,
                    hipStream_t st) {
    const size_t n_rows = (size_t)n_channels * n_groups * 4;
    hipLaunchKernelGGL(k_p2_sequence, dim3((unsigned)((n_channels + 63) / 64)), dim3(64), 0, st, duid, isch, n_channels, n_groups, groups_of, seed44, state,
                       info, row_off, seq_of, counts, list, n_rows, ess_src, final_src);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_gather(int cls, int count, const int32_t* list, const int32_t* info, const uint8_t* rb, const int16_t* rl, const uint8_t* xb,
                  const int16_t* xl, uint8_t* db, int16_t* dl, const int32_t* ess_src, const ddn_p25p2_seq_state* state, int rows_per_channel,
                  uint8_t* ess_pl, int16_t* ess_pll, uint8_t* ess_pa, int16_t* ess_pal, hipStream_t st) {
    if (count <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_gather, dim3((unsigned)count), dim3(128), 0, st, cls, list, info, rb, rl, xb, xl, db, dl, ess_src, state,
                       rows_per_channel, ess_pl, ess_pll, ess_pa, ess_pal);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_state_ess(const int32_t* final_src, const uint8_t* xb, const int16_t* xl, int n_channels, ddn_p25p2_seq_state* state, hipStream_t st) {
    hipLaunchKernelGGL(k_p2_state_ess, dim3((unsigned)(n_channels * 2)), dim3(64), 0, st, final_src, xb, xl, n_channels, state);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_scatter(int cls, int count, const int32_t* list, int32_t* info, const uint8_t* x_payload, int n_pl, const int32_t* ec,
                   const uint8_t* used, const uint8_t* c12, const uint8_t* c16, const uint8_t* fr, const uint8_t* rel, int frame_count,
                   const uint8_t* ess_out, uint8_t* o_payload, uint8_t* o_fr, uint8_t* o_rel, uint8_t* o_ess, hipStream_t st) {
    if (count <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)count), dim3(128), 0, st, cls, list, info, x_payload, n_pl, ec, used, c12, c16, fr, rel,
                       frame_count, ess_out, o_payload, o_fr, o_rel, o_ess);
    return hipGetLastError();
}

hipError_t
ddn_dev_p2_sync_cut(const uint8_t* dibits, const int16_t* llr2, int n_channels, int n, size_t stride, const int32_t* cursor_in, int max_groups,
                    int32_t* n_groups, int32_t* group_pos, int32_t* cursor_out, uint8_t* bits1400, int16_t* llr1400, hipStream_t st) {
    hipLaunchKernelGGL(k_p2_sync_cut, dim3((unsigned)n_channels), dim3(64), 0, st, dibits, n, stride, cursor_in, max_groups, n_groups, group_pos,
                       cursor_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || max_groups == 0) {
        return e;
    }
    hipLaunchKernelGGL(k_p2_cut_copy, dim3((unsigned)max_groups, (unsigned)n_channels), dim3(256), 0, st, dibits, llr2, stride, max_groups,
                       n_groups, group_pos, bits1400, llr1400);
    return hipGetLastError();
}
