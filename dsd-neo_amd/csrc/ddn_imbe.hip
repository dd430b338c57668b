// ddn_imbe.hip — batched IMBE de-interleave of P25 Phase 1 voice frames (LDU1 / LDU2 carry nine each).
//
// reference: process_IMBE(), src/protocol/p25/phase1/p25p1_ldu.c:89-120 (store loop :42-48, mid-frame status symbol
// :27-39, non-standard c0 word :55-66); soft bit include/dsd-neo/core/vocoder.h:30-38.
//
// A pure gather, so it is written output-side: one workgroup per voice frame, one thread per cell of the 8 x 23 code
// vector array.  Each thread inverts the interleave schedule for its cell (the 144 code bits laid end to end are sent
// twelve per row: columns 0..5 carry stream bits 24k + 2r, columns 6..11 carry 24k + 2r + 1 for k = 1,0,3,2,5,4),
// finds the dibit that carries it, steps over the status symbols the reference skips on the way (one whenever its
// running dibit counter shows 35) and reads one 10-byte capture record {dibit, rel, llr0, llr1, f32 symbol}.  Cells
// the schedule never writes (c4..c6 hold 15 bits, c7 holds 7) stay zero like the reference's memset.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_device.h"

namespace {
__global__ __launch_bounds__(64) void
k_imbe_deinterleave(const uint8_t* __restrict__ rec, long n_records, const int64_t* __restrict__ first,
                    const int32_t* __restrict__ status_count, int n_frames, uint8_t* __restrict__ fr,
                    uint8_t* __restrict__ soft, uint8_t* __restrict__ flags, int32_t* __restrict__ status_out) {
    __shared__ uint8_t c0[23];
    __shared__ int short_read;
    const int f = blockIdx.x;
    const int t0 = threadIdx.x; // one wavefront per frame, three passes over the 184 cells (a single-wave workgroup finds room beside
                                // the front-end kernel of the next call, a three-wave one waited for it: DDN_WG, ddn_device.h)
    if (t0 == 0) {
        short_read = 0;
    }
    __syncthreads();
    const long base = first[f];
    const int sc0 = status_count[f];
    // status symbols are skipped before dibit steps t1, t1 + 35, ... (the counter shows 35 at step 35 - sc0, then
    // every 35 steps); dibit step j therefore sits j + skips(j) records after the frame's first record
    const int t1 = 35 - sc0;
    auto skips = [&](int j) { return (sc0 <= 35 && j >= t1) ? 1 + (j - t1) / 35 : 0; };
    for (int t = t0; t < 184; t += 64) {
        const int v = t / 23, i = t % 23;
        const int len = v < 4 ? 23 : (v < 7 ? 15 : 7);
        const int off = v < 4 ? 23 * v : (v < 7 ? 92 + 15 * (v - 4) : 137);
        int bit = 0, rel = 0;
        if (i < len) {
            const int L = off + (len - 1 - i);
            const int k = L / 24, rr = L % 24;
            const int r = rr >> 1;
            const int m = (rr & 1) ? 6 + (k ^ 1) : k;
            const int p = 12 * r + m;
            const int j = p >> 1, low = p & 1;
            const long ri = base + j + skips(j);
            if (base < 0 || ri >= n_records) {
                short_read = 1;
            } else {
                const uint8_t* q = rec + (size_t)ri * 10;
                const int d = q[0];
                const int llr = (int16_t)((uint16_t)q[2 + 2 * low] | ((uint16_t)q[3 + 2 * low] << 8));
                bit = low ? (d & 1) : ((d >> 1) & 1);
                rel = llr < 0 ? -llr : llr;
                rel = rel > 255 ? 255 : rel;
            }
        }
        fr[(size_t)f * 184 + t] = (uint8_t)bit;
        soft[((size_t)f * 184 + t) * 2] = (uint8_t)bit;
        soft[((size_t)f * 184 + t) * 2 + 1] = (uint8_t)rel;
        if (t < 23) {
            c0[t] = (uint8_t)bit;
        }
    }
    __syncthreads();
    if (t0 == 0) {
        int ns = 1;
        for (int i = 0; i < 23; i++) {
            ns &= (c0[i] == ((i >= 15 && i <= 17) ? 1 : 0));
        }
        flags[f] = short_read ? 0xFF : (uint8_t)ns;
        // counter after 72 dibit steps: sc0 + 72 without a skip, else the steps since the last skip, plus one
        const int s71 = skips(71);
        status_out[f] = s71 == 0 ? sc0 + 72 : (71 - (t1 + 35 * (s71 - 1))) + 1;
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_imbe_deinterleave(const uint8_t* rec, long n_records, const int64_t* first, const int32_t* status_count,
                          int n_frames, uint8_t* fr, uint8_t* soft, uint8_t* flags, int32_t* status_out, hipStream_t st) {
    if (n_frames <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(k_imbe_deinterleave, dim3((unsigned)n_frames), dim3(64), 0, st, rec, n_records, first,
                       status_count, n_frames, fr, soft, flags, status_out);
    return hipGetLastError();
}
