// ddn_api_fec3.cpp - C-ABI of the DMR / NXDN block codes (include/ddn_hip.h, "block codes downstream of the receive
// loop"): batched device-pointer calls, host-buffer variants and the drop-ins with the reference's names
// (include/dsd-neo/fec/block_codes.h:19-43, bptc.h:20-26, rs_12_9.h:38-42).

#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ddn_device.h"
#include "ddn_fec3.h"

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            ddn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                  \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorNoBinaryForGpu)             \
                       ? DDN_ENODEV                                                                                    \
                       : (e_ == hipErrorOutOfMemory ? DDN_ENOMEM : DDN_EHIP);                                          \
        }                                                                                                              \
    } while (0)

static int
code_shape(int code, int* n, int* k) {
    static const int N[] = {7, 12, 13, 15, 16, 20, 24, 16}, K[] = {4, 8, 9, 11, 11, 8, 12, 7};
    if (code < 0 || code > DDN_CODE_QR_16_7_6) {
        return DDN_EINVAL;
    }
    *n = N[code];
    *k = K[code];
    return DDN_OK;
}

static int
have_device() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    return DDN_OK;
}

extern "C" int
ddn_fec_block_code_batch(int code, uint8_t* d_bits, size_t n_items, int nb_codewords, uint8_t* d_decoded, uint8_t* d_ok,
                         void* hip_stream) {
    int n, k;
    if (code_shape(code, &n, &k) != DDN_OK || nb_codewords < 1 || (n_items && !d_bits)) {
        ddn_set_error("ddn_fec_block_code_batch: bad argument");
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    const bool multi = code >= DDN_CODE_HAMMING_12_8 && code <= DDN_CODE_HAMMING_16_11_4;
    HIP_TRY(ddn_dev_block_code(code, d_bits, n_items, multi ? nb_codewords : 1, multi ? d_decoded : nullptr, d_ok,
                               (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_block_code_host(int code, uint8_t* bits, size_t n_items, int nb_codewords, uint8_t* decoded, uint8_t* ok) {
    int n, k;
    if (code_shape(code, &n, &k) != DDN_OK || nb_codewords < 1 || (n_items && !bits)) {
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    const bool multi = code >= DDN_CODE_HAMMING_12_8 && code <= DDN_CODE_HAMMING_16_11_4;
    const int nb = multi ? nb_codewords : 1;
    const size_t nbits = n_items * (size_t)n * (size_t)nb, ndec = n_items * (size_t)k * (size_t)nb;
    uint8_t *d_b = nullptr, *d_d = nullptr, *d_o = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_b, nbits + 4) != hipSuccess || hipMalloc(&d_d, ndec + 4) != hipSuccess
        || hipMalloc(&d_o, n_items + 4) != hipSuccess) {
        rc = DDN_ENOMEM;
    } else if (hipMemcpy(d_b, bits, nbits, hipMemcpyHostToDevice) != hipSuccess
               || hipMemset(d_d, 0, ndec + 4) != hipSuccess
               || ddn_dev_block_code(code, d_b, n_items, nb, (multi && decoded) ? d_d : nullptr, d_o, nullptr) != hipSuccess
               || hipMemcpy(bits, d_b, nbits, hipMemcpyDeviceToHost) != hipSuccess
               || (multi && decoded && hipMemcpy(decoded, d_d, ndec, hipMemcpyDeviceToHost) != hipSuccess)
               || (ok && hipMemcpy(ok, d_o, n_items, hipMemcpyDeviceToHost) != hipSuccess)) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_b);
    (void)hipFree(d_d);
    (void)hipFree(d_o);
    return rc;
}

extern "C" int
ddn_fec_bptc_196x96_batch(const uint8_t* d_in196, int deinterleave, size_t n, uint8_t* d_out96, uint8_t* d_r3,
                          uint32_t* d_errs, void* hip_stream) {
    if (n && (!d_in196 || !d_out96)) {
        ddn_set_error("ddn_fec_bptc_196x96_batch: null argument");
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    HIP_TRY(ddn_dev_bptc_196x96(d_in196, deinterleave, n, d_out96, d_r3, d_errs, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_bptc_196x96_host(const uint8_t* in196, int deinterleave, size_t n, uint8_t* out96, uint8_t* r3, uint32_t* errs) {
    if (n && (!in196 || !out96)) {
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    uint8_t *d_i = nullptr, *d_o = nullptr, *d_r = nullptr;
    uint32_t* d_e = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_i, n * 196 + 4) != hipSuccess || hipMalloc(&d_o, n * 96 + 4) != hipSuccess
        || hipMalloc(&d_r, n * 3 + 4) != hipSuccess || hipMalloc(&d_e, n * 4 + 4) != hipSuccess) {
        rc = DDN_ENOMEM;
    } else if (hipMemcpy(d_i, in196, n * 196, hipMemcpyHostToDevice) != hipSuccess
               || ddn_dev_bptc_196x96(d_i, deinterleave, n, d_o, d_r, d_e, nullptr) != hipSuccess
               || hipMemcpy(out96, d_o, n * 96, hipMemcpyDeviceToHost) != hipSuccess
               || (r3 && hipMemcpy(r3, d_r, n * 3, hipMemcpyDeviceToHost) != hipSuccess)
               || (errs && hipMemcpy(errs, d_e, n * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_i);
    (void)hipFree(d_o);
    (void)hipFree(d_r);
    (void)hipFree(d_e);
    return rc;
}

// BPTC 128x77 (kind 0: item 128 bytes -> 77) and reverse-channel BPTC 16x2 (kind 1 even / 2 odd parity: 32 -> 32)
static int
bptc_small(int kind, const uint8_t* d_in, size_t n, uint8_t* d_out, uint32_t* d_errs, hipStream_t st) {
    hipError_t e = kind == 0 ? ddn_dev_bptc_128x77(d_in, n, d_out, d_errs, st) : ddn_dev_bptc_16x2(d_in, n, kind == 2, d_out, d_errs, st);
    if (e != hipSuccess) {
        ddn_set_error("bptc kernel launch failed: %s", hipGetErrorString(e));
        return DDN_EHIP;
    }
    return DDN_OK;
}

static int
bptc_small_host(int kind, const uint8_t* in, size_t n, uint8_t* out, uint32_t* errs) {
    const size_t ni = kind == 0 ? 128 : 32, no = kind == 0 ? 77 : 32;
    if (n && (!in || !out)) {
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    uint8_t *d_i = nullptr, *d_o = nullptr;
    uint32_t* d_e = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_i, n * ni + 4) != hipSuccess || hipMalloc(&d_o, n * no + 4) != hipSuccess
        || hipMalloc(&d_e, n * 4 + 4) != hipSuccess) {
        rc = DDN_ENOMEM;
    } else if (hipMemcpy(d_i, in, n * ni, hipMemcpyHostToDevice) != hipSuccess || bptc_small(kind, d_i, n, d_o, d_e, nullptr) != DDN_OK
               || hipMemcpy(out, d_o, n * no, hipMemcpyDeviceToHost) != hipSuccess
               || (errs && hipMemcpy(errs, d_e, n * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_i);
    (void)hipFree(d_o);
    (void)hipFree(d_e);
    return rc;
}

extern "C" int
ddn_fec_bptc_128x77_batch(const uint8_t* d_in128, size_t n, uint8_t* d_out77, uint32_t* d_errs, void* stream) {
    if (n && (!d_in128 || !d_out77)) {
        ddn_set_error("ddn_fec_bptc_128x77_batch: null argument");
        return DDN_EINVAL;
    }
    return bptc_small(0, d_in128, n, d_out77, d_errs, (hipStream_t)stream);
}

extern "C" int
ddn_fec_bptc_128x77_host(const uint8_t* in128, size_t n, uint8_t* out77, uint32_t* errs) {
    return bptc_small_host(0, in128, n, out77, errs);
}

extern "C" int
ddn_fec_bptc_16x2_batch(const uint8_t* d_in32, size_t n, int parity_odd, uint8_t* d_out32, uint32_t* d_errs, void* stream) {
    if (n && (!d_in32 || !d_out32)) {
        ddn_set_error("ddn_fec_bptc_16x2_batch: null argument");
        return DDN_EINVAL;
    }
    return bptc_small(parity_odd ? 2 : 1, d_in32, n, d_out32, d_errs, (hipStream_t)stream);
}

extern "C" int
ddn_fec_bptc_16x2_host(const uint8_t* in32, size_t n, int parity_odd, uint8_t* out32, uint32_t* errs) {
    return bptc_small_host(parity_odd ? 2 : 1, in32, n, out32, errs);
}

extern "C" uint32_t
BPTC_128x77_Extract_Data(uint8_t InputDataMatrix[8][16], uint8_t DMRDataExtracted[77]) {
    uint32_t errs = 0xFFFFFFFFu;
    if (!InputDataMatrix || !DMRDataExtracted
        || bptc_small_host(0, &InputDataMatrix[0][0], 1, DMRDataExtracted, &errs) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    return errs;
}

extern "C" uint32_t
BPTC_16x2_Extract_Data(uint8_t InputInterleavedData[32], uint8_t DMRDataExtracted[32], uint32_t ParityCheckTypeOdd) {
    uint32_t errs = 0xFFFFFFFFu;
    if (!InputInterleavedData || !DMRDataExtracted
        || bptc_small_host(ParityCheckTypeOdd ? 2 : 1, InputInterleavedData, 1, DMRDataExtracted, &errs) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    return errs;
}

extern "C" int
ddn_fec_trellis_decode_batch(const uint8_t* d_source_bits, int source_stride, size_t n, int result_len, uint8_t* d_result_bits,
                             int result_stride, void* hip_stream) {
    if (n && (!d_source_bits || !d_result_bits)) {
        ddn_set_error("ddn_fec_trellis_decode_batch: null argument");
        return DDN_EINVAL;
    }
    if (result_len <= 0 || source_stride < 2 * result_len + 6 || result_stride < result_len) {
        ddn_set_error("ddn_fec_trellis_decode_batch: a row needs 2 * result_len + 6 source bits (the last bit looks 8 ahead)");
        return DDN_ERANGE;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    HIP_TRY(ddn_dev_trellis_greedy(d_source_bits, source_stride, n, result_len, d_result_bits, result_stride, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_trellis_decode_host(const uint8_t* source_bits, int source_stride, size_t n, int result_len, uint8_t* result_bits,
                            int result_stride) {
    if (n && (!source_bits || !result_bits)) {
        return DDN_EINVAL;
    }
    if (result_len <= 0 || source_stride < 2 * result_len + 6 || result_stride < result_len) {
        return DDN_ERANGE;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    uint8_t *d_s = nullptr, *d_o = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_s, n * (size_t)source_stride + 4) != hipSuccess || hipMalloc(&d_o, n * (size_t)result_stride + 4) != hipSuccess) {
        rc = DDN_ENOMEM;
    } else if (hipMemcpy(d_s, source_bits, n * (size_t)source_stride, hipMemcpyHostToDevice) != hipSuccess
               || hipMemset(d_o, 0, n * (size_t)result_stride) != hipSuccess
               || ddn_dev_trellis_greedy(d_s, source_stride, n, result_len, d_o, result_stride, nullptr) != hipSuccess
               || hipMemcpy(result_bits, d_o, n * (size_t)result_stride, hipMemcpyDeviceToHost) != hipSuccess) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_s);
    (void)hipFree(d_o);
    return rc;
}

// drop-in: include/dsd-neo/fec/trellis.h:22.  The reference reads source[2p .. 2p+7] for p < result_len, i.e. the caller's buffer
// holds 2 * result_len + 6 bits (its callers pass the zero-padded de-punctured field).
extern "C" void
trellis_decode(uint8_t result[], const uint8_t source[], int result_len) {
    if (!result || !source || result_len <= 0) {
        return;
    }
    (void)ddn_fec_trellis_decode_host(source, 2 * result_len + 6, 1, result_len, result, result_len);
}

extern "C" int
ddn_fec_rs_12_9_batch(uint8_t* d_codewords12, size_t n, uint8_t* d_result, uint8_t* d_errors_found, uint8_t* d_syndrome3,
                      void* hip_stream) {
    if (n && (!d_codewords12 || !d_result)) {
        ddn_set_error("ddn_fec_rs_12_9_batch: null argument");
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    HIP_TRY(ddn_dev_rs_12_9(d_codewords12, n, d_result, d_errors_found, d_syndrome3, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_rs_12_9_host(uint8_t* codewords12, size_t n, uint8_t* result, uint8_t* errors_found, uint8_t* syndrome3) {
    if (n && (!codewords12 || !result)) {
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    uint8_t *d_c = nullptr, *d_r = nullptr, *d_f = nullptr, *d_s = nullptr;
    int rc = DDN_OK;
    if (hipMalloc(&d_c, n * 12 + 4) != hipSuccess || hipMalloc(&d_r, n + 4) != hipSuccess
        || hipMalloc(&d_f, n + 4) != hipSuccess || hipMalloc(&d_s, n * 3 + 4) != hipSuccess) {
        rc = DDN_ENOMEM;
    } else if (hipMemcpy(d_c, codewords12, n * 12, hipMemcpyHostToDevice) != hipSuccess
               || ddn_dev_rs_12_9(d_c, n, d_r, d_f, d_s, nullptr) != hipSuccess
               || hipMemcpy(codewords12, d_c, n * 12, hipMemcpyDeviceToHost) != hipSuccess
               || hipMemcpy(result, d_r, n, hipMemcpyDeviceToHost) != hipSuccess
               || (errors_found && hipMemcpy(errors_found, d_f, n, hipMemcpyDeviceToHost) != hipSuccess)
               || (syndrome3 && hipMemcpy(syndrome3, d_s, n * 3, hipMemcpyDeviceToHost) != hipSuccess)) {
        rc = DDN_EHIP;
    }
    (void)hipFree(d_c);
    (void)hipFree(d_r);
    (void)hipFree(d_f);
    (void)hipFree(d_s);
    return rc;
}

// ---- the reference's names ---------------------------------------------------------------------------------------------
// The *_init() functions build file-static tables in the reference; here the tables are built with the library's first
// call on a device, so they are empty.
extern "C" void Hamming_7_4_init(void) {}
extern "C" void Hamming_12_8_init(void) {}
extern "C" void Hamming_13_9_init(void) {}
extern "C" void Hamming_15_11_init(void) {}
extern "C" void Hamming_16_11_4_init(void) {}
extern "C" void Golay_20_8_init(void) {}
extern "C" void Golay_24_12_init(void) {}
extern "C" void QR_16_7_6_init(void) {}
extern "C" void InitAllFecFunction(void) {}

static bool
one(int code, unsigned char* rx, unsigned char* dec, int nb) {
    uint8_t ok = 0;
    if (!rx || ddn_fec_block_code_host(code, rx, 1, nb, dec, &ok) != DDN_OK) {
        return false;
    }
    return ok != 0;
}

extern "C" bool Hamming_7_4_decode(unsigned char* rxBits) { return one(DDN_CODE_HAMMING_7_4, rxBits, nullptr, 1); }
extern "C" bool
Hamming_12_8_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords) {
    return nbCodewords < 1 ? true : one(DDN_CODE_HAMMING_12_8, rxBits, decodedBits, nbCodewords);
}
extern "C" bool
Hamming_13_9_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords) {
    return nbCodewords < 1 ? true : one(DDN_CODE_HAMMING_13_9, rxBits, decodedBits, nbCodewords);
}
extern "C" bool
Hamming_15_11_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords) {
    return nbCodewords < 1 ? true : one(DDN_CODE_HAMMING_15_11, rxBits, decodedBits, nbCodewords);
}
extern "C" bool
Hamming_16_11_4_decode(unsigned char* rxBits, unsigned char* decodedBits, int nbCodewords) {
    return nbCodewords < 1 ? true : one(DDN_CODE_HAMMING_16_11_4, rxBits, decodedBits, nbCodewords);
}
extern "C" bool Golay_20_8_decode(unsigned char* rxBits) { return one(DDN_CODE_GOLAY_20_8, rxBits, nullptr, 1); }
extern "C" bool Golay_24_12_decode(unsigned char* rxBits) { return one(DDN_CODE_GOLAY_24_12, rxBits, nullptr, 1); }
extern "C" bool QR_16_7_6_decode(unsigned char* rxBits) { return one(DDN_CODE_QR_16_7_6, rxBits, nullptr, 1); }

extern "C" void
BPTCDeInterleaveDMRData(const uint8_t* Input, uint8_t* Output) { // pure index permutation: host (src/fec/bptc.c:27-46)
    if (!Input || !Output) {
        return;
    }
    for (uint32_t i = 0; i < 196; i++) {
        Output[(i * 13u) % 196u] = Input[i] & 1u; // BPTCDeInterleavingIndex[i] = 13 i mod 196
    }
}

extern "C" uint32_t
BPTC_196x96_Extract_Data(uint8_t InputDeInteleavedData[196], uint8_t DMRDataExtracted[96], uint8_t R[3]) {
    uint32_t errs = 0xFFFFFFFFu;
    uint8_t r3[3] = {0, 0, 0};
    if (!InputDeInteleavedData || !DMRDataExtracted
        || ddn_fec_bptc_196x96_host(InputDeInteleavedData, 0, 1, DMRDataExtracted, r3, &errs) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    if (R) {
        memcpy(R, r3, 3);
    }
    return errs;
}

extern "C" void
rs_12_9_calc_syndrome(const rs_12_9_codeword_t* codeword, rs_12_9_poly_t* syndrome) {
    if (!codeword || !syndrome) {
        return;
    }
    uint8_t cw[12], res = 0, syn[3] = {0, 0, 0};
    memcpy(cw, codeword->data, 12);
    memset(syndrome->data, 0, sizeof(syndrome->data));
    if (ddn_fec_rs_12_9_host(cw, 1, &res, nullptr, syn) == DDN_OK) {
        memcpy(syndrome->data, syn, 3);
    }
}

extern "C" uint8_t
rs_12_9_check_syndrome(const rs_12_9_poly_t* syndrome) {
    return (syndrome && (syndrome->data[0] | syndrome->data[1] | syndrome->data[2])) ? 1 : 0;
}

// `syndrome` must be the one rs_12_9_calc_syndrome returned for this code word (what every caller passes): the device
// kernel recomputes it
extern "C" rs_12_9_correct_errors_result_t
rs_12_9_correct_errors(rs_12_9_codeword_t* codeword, const rs_12_9_poly_t* syndrome, uint8_t* errors_found) {
    (void)syndrome;
    uint8_t res = RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CANT_BE_CORRECTED, found = 0;
    if (!codeword || ddn_fec_rs_12_9_host(codeword->data, 1, &res, &found, nullptr) != DDN_OK) {
        return RS_12_9_CORRECT_ERRORS_RESULT_ERRORS_CANT_BE_CORRECTED;
    }
    if (errors_found) {
        *errors_found = found;
    }
    return res;
}
