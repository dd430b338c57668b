// ddn_mbe_dev.h - device-side records of the vocoder stage (internal; the C-ABI is include/ddn_mbe.h)
#ifndef DDN_MBE_DEV_H
#define DDN_MBE_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_mbe.h"

#define DDN_MBE_RESULT_INVALID 0x80000000u /* result flags: the frame held a byte other than 0 / 1 */
#define DDN_MBE_REC_SILENCE    1u

// one talk path's decoder history, exactly the caller-visible triple + the frame counter that keys the generator
struct DdnMbeStream {
    mbe_parms cur, prev, enh;
    uint32_t frame_no;
    uint32_t seed;
};

// everything k_mbe_synth needs for one frame: both sides of the cross-fade (previous enhanced frame / this frame)
struct DdnMbeFrameRec {
    float cw0, pw0;
    int32_t maxl;
    uint32_t flags;
    uint32_t cv_lo, cv_hi, pv_lo, pv_hi; // voicing bit l of the current / previous side
    uint32_t fbase;                      // generator key of (talk path, frame)
    uint32_t pad[7];
    float cMl[64], pMl[64], cPHI[64], pPHI[64];
};

extern "C" {
hipError_t ddn_dev_mbe_frame_decode(int codec, const uint8_t* frames, const uint8_t* soft, size_t n, uint8_t* bits,
                                    int32_t* result, hipStream_t st);
hipError_t ddn_dev_mbe_result_skip(const uint8_t* skip, size_t n, int32_t* result, hipStream_t st);
hipError_t ddn_dev_mbe_stream_init(DdnMbeStream* streams, int n_streams, uint32_t seed0, hipStream_t st);
hipError_t ddn_dev_mbe_params(int codec, const uint8_t* bits, const int32_t* res_in, int n_streams, int n_frames,
                              const ddn_mbe_tables* d_tables, const float* d_half_log2, DdnMbeStream* streams,
                              int tail_rule, DdnMbeFrameRec* recs, int32_t* res_out, hipStream_t st);
hipError_t ddn_dev_mbe_synth(const DdnMbeFrameRec* recs, size_t n_frames_total, float* pcm, hipStream_t st);
}
#endif
