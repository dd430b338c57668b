/* ddn_device.h — kernel argument structs shared by the .hip kernels and the C-ABI host code. */
#ifndef DDN_DEVICE_H
#define DDN_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ddn_internal.h"

#define DDN_CARRY_LEN 72 /* >= DDN_MAX_CENTER widened samples of FIR look-back per channel */
#define DDN_FIR_R     8  /* outputs per thread in the unrolled FIR kernel */

typedef struct DdnFskState { /* == the fields of dsd_fsk_modem_state the path carries */
    float prev_i, prev_q;
    int have_prev;
    float dc_est, peak_est;
} DdnFskState;

typedef float ddn_f2 __attribute__((ext_vector_type(2)));

typedef struct DdnFirArgs {
    const void* in;    /* [B][ch_stride] complex samples (cu8 pairs or float pairs) */
    float* out;        /* [B][out_stride] raw phase deltas */
    const ddn_f2* carry;
    ddn_f2* tile_edge; /* [B][n_tiles][2]: first / last LPF output of each FIR tile */
    float* blk_pwr;    /* [B][n_blocks] (squelch only) */
    size_t ch_stride;
    size_t out_stride;
    long n;            /* complex samples per channel in this call */
    int in_fmt;
    int block_len;
    int tiles_per_block;
    int n_tiles;       /* n_blocks * tiles_per_block */
    int n_blocks;
    int squelch_on;
} DdnFirArgs;

typedef struct DdnSerialArgs {
    float* buf; /* in: raw phase deltas, out: discriminator samples (in place) */
    const ddn_f2* tile_edge;
    const float* blk_pwr;
    DdnFskState* state;
    size_t stride;
    long n;
    int n_channels;
    int block_len;
    int fir_tile;
    int tiles_per_block;
    int n_tiles;
    int n_blocks;
    int squelch_on;
    float squelch_level;
} DdnSerialArgs;

#ifdef __cplusplus
extern "C" {
#endif
int ddn_dev_fir_tile(int center);
hipError_t ddn_dev_launch_fir(const DdnFirArgs* a, const float* taps_host, const float* taps_dev, int center,
                              int n_channels, hipStream_t st);
hipError_t ddn_dev_launch_serial(const DdnSerialArgs* a, hipStream_t st);
hipError_t ddn_dev_launch_carry(const void* in, int in_fmt, size_t ch_stride, long n, void* carry, int n_channels,
                                hipStream_t st);
hipError_t ddn_dev_zero(void* p, size_t bytes, hipStream_t st);
#ifdef __cplusplus
}
#endif
#endif
