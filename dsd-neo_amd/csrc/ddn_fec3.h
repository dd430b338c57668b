// ddn_fec3.h - internal launch prototypes of ddn_fec3.hip (the C-ABI is in include/ddn_hip.h)
#ifndef DDN_FEC3_H
#define DDN_FEC3_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_internal.h"

/* syndrome -> positions-to-flip tables of the DMR / NXDN block codes, one copy per device, built in the reference's init-loop
 * order (ddn_fec3.hip) */
typedef struct DdnFec3Tables {
    uint8_t h74[8], h128[16], h139[16], h1511[16], h16114[32];
    uint8_t g208[4096][3], g2412[4096][3], qr[512][2];
} DdnFec3Tables;

extern "C" {
hipError_t ddn_dev_fec3_tables(const DdnFec3Tables** out, hipStream_t st);
hipError_t ddn_dev_block_code(int code, uint8_t* bits, size_t n_items, int nb, uint8_t* decoded, uint8_t* ok, hipStream_t st);
hipError_t ddn_dev_bptc_128x77(const uint8_t* in, size_t n, uint8_t* out77, uint32_t* errs, hipStream_t st);
hipError_t ddn_dev_bptc_16x2(const uint8_t* in, size_t n, int parity_odd, uint8_t* out32, uint32_t* errs, hipStream_t st);
hipError_t ddn_dev_bptc_196x96(const uint8_t* in, int deinterleave, size_t n, uint8_t* out96, uint8_t* r3, uint32_t* errs,
                               hipStream_t st);
hipError_t ddn_dev_trellis_greedy(const uint8_t* src, int src_stride, size_t n, int result_len, uint8_t* out, int out_stride,
                                  hipStream_t st);
hipError_t ddn_dev_rs_12_9(uint8_t* cw, size_t n, uint8_t* result, uint8_t* found, uint8_t* syn_out, hipStream_t st);
}
#endif
