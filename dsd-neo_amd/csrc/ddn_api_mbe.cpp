// ddn_api_mbe.cpp - C-ABI of the vocoder stage (include/ddn_mbe.h): batched frame decode / synthesis objects and the
// single-frame drop-ins with mbelib-neo 2.x names that src/core/vocoder/dsd_mbe.c calls (:75-190, :540-598).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "ddn_device.h"
#include "ddn_mbe.h"
#include "ddn_mbe_dev.h"

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

struct ddn_mbe_batch {
    int codec, n_streams, tail_rule;
    uint32_t first_stream; // number of stream 0 in the caller's numbering (the unvoiced-noise sequence of a path is a function of its number)
    int synthetic; // the loaded table blob's `synthetic` word (1 until ddn_mbe_batch_set_tables() brings real tables)
    ddn_mbe_tables* d_tables;
    float* d_half_log2;     // [57] 0.5 * log2(L)
    DdnMbeStream* d_streams;
    DdnMbeFrameRec* d_recs; // [n_streams][rec_frames]
    size_t rec_frames;
    bool timing;
    hipEvent_t ev[3];
    float last_ms[2];
};

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
ddn_mbe_frame_decode_batch(int codec, const uint8_t* d_frames, const uint8_t* d_soft, size_t n, uint8_t* d_bits,
                           int32_t* d_result, void* hip_stream) {
    if ((codec != DDN_MBE_IMBE_7200X4400 && codec != DDN_MBE_AMBE_3600X2450) || (n && (!d_frames || !d_bits || !d_result))) {
        ddn_set_error("ddn_mbe_frame_decode_batch: bad argument");
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    HIP_TRY(ddn_dev_mbe_frame_decode(codec, d_frames, d_soft, n, d_bits, d_result, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_mbe_result_skip_batch(const uint8_t* d_skip, size_t n, int32_t* d_result, void* hip_stream) {
    if (n && (!d_skip || !d_result)) {
        ddn_set_error("ddn_mbe_result_skip_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_mbe_result_skip(d_skip, n, d_result, (hipStream_t)hip_stream));
    return DDN_OK;
}

static void
mbe_free(ddn_mbe_batch* b) {
    (void)hipFree(b->d_tables);
    (void)hipFree(b->d_half_log2);
    (void)hipFree(b->d_streams);
    (void)hipFree(b->d_recs);
    for (int i = 0; i < 3; i++) {
        if (b->ev[i]) {
            (void)hipEventDestroy(b->ev[i]);
        }
    }
}

extern "C" int
ddn_mbe_batch_create(int codec, int n_streams, ddn_mbe_batch** out) {
    if (!out || n_streams <= 0 || (codec != DDN_MBE_IMBE_7200X4400 && codec != DDN_MBE_AMBE_3600X2450)) {
        ddn_set_error("ddn_mbe_batch_create: bad argument");
        return DDN_EINVAL;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    ddn_mbe_batch* b = new (std::nothrow) ddn_mbe_batch();
    if (!b) {
        return DDN_ENOMEM;
    }
    memset(b, 0, sizeof(*b));
    b->codec = codec;
    b->n_streams = n_streams;
    ddn_mbe_tables* t = new (std::nothrow) ddn_mbe_tables;
    float hl[57];
    hl[0] = 0.0f;
    for (int L = 1; L <= 56; L++) {
        hl[L] = 0.5f * (logf((float)L) / logf(2.0f)); // ambe3600x2450.c "BigGamma": gamma - 0.5 log2(L) - mean(T)
    }
    if (!t || ddn_mbe_default_tables(t) != DDN_OK || hipMalloc(&b->d_tables, sizeof(ddn_mbe_tables)) != hipSuccess
        || hipMalloc(&b->d_half_log2, sizeof(hl)) != hipSuccess
        || hipMalloc(&b->d_streams, sizeof(DdnMbeStream) * (size_t)n_streams) != hipSuccess
        || hipMemcpy(b->d_tables, t, sizeof(ddn_mbe_tables), hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(b->d_half_log2, hl, sizeof(hl), hipMemcpyHostToDevice) != hipSuccess
        || ddn_dev_mbe_stream_init(b->d_streams, n_streams, 0u, nullptr) != hipSuccess
        || hipDeviceSynchronize() != hipSuccess) {
        ddn_set_error("ddn_mbe_batch_create: device allocation failed");
        delete t;
        mbe_free(b);
        delete b;
        return DDN_ENOMEM;
    }
    b->synthetic = t->synthetic ? 1 : 0;
    delete t;
    *out = b;
    return DDN_OK;
}

extern "C" void
ddn_mbe_batch_destroy(ddn_mbe_batch* b) {
    if (!b) {
        return;
    }
    mbe_free(b);
    delete b;
}

extern "C" int
ddn_mbe_batch_reset(ddn_mbe_batch* b, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_mbe_stream_init(b->d_streams, b->n_streams, b->first_stream, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_set_first_stream(ddn_mbe_batch* b, uint32_t first_stream, void* hip_stream) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->first_stream = first_stream;
    return ddn_mbe_batch_reset(b, hip_stream);
}

extern "C" int
ddn_mbe_batch_set_tables(ddn_mbe_batch* b, const ddn_mbe_tables* t) {
    if (!b || !t) {
        return DDN_EINVAL;
    }
    const int rc = ddn_mbe_validate_tables(t);
    if (rc != DDN_OK) {
        ddn_set_error("ddn_mbe_batch_set_tables: table blob failed validation");
        return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b->d_tables, t, sizeof(ddn_mbe_tables), hipMemcpyHostToDevice));
    b->synthetic = t->synthetic ? 1 : 0;
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_tables_synthetic(const ddn_mbe_batch* b) {
    return b ? b->synthetic : -1;
}

extern "C" int
ddn_mbe_batch_load_tables_file(ddn_mbe_batch* b, const char* path) {
    if (!b || !path) {
        return DDN_EINVAL;
    }
    ddn_mbe_tables* t = new (std::nothrow) ddn_mbe_tables;
    if (!t) {
        return DDN_ENOMEM;
    }
    int rc = ddn_mbe_tables_load_file(path, t);
    if (rc == DDN_OK) {
        rc = ddn_mbe_batch_set_tables(b, t);
    }
    delete t;
    return rc;
}

extern "C" int
ddn_mbe_batch_set_p25p1_tail_rule(ddn_mbe_batch* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    b->tail_rule = enable ? 1 : 0;
    return DDN_OK;
}

extern "C" int
ddn_mbe_synth_batch(ddn_mbe_batch* b, const uint8_t* d_bits, const int32_t* d_result_in, size_t n_frames, float* d_pcm,
                    int32_t* d_result_out, void* hip_stream) {
    if (!b || (n_frames && (!d_bits || !d_pcm)) || n_frames > (size_t)1 << 24) {
        ddn_set_error("ddn_mbe_synth_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n_frames == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    if (b->rec_frames < n_frames) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(b->d_recs);
        b->d_recs = nullptr;
        b->rec_frames = 0;
        HIP_TRY(hipMalloc(&b->d_recs, sizeof(DdnMbeFrameRec) * (size_t)b->n_streams * n_frames));
        b->rec_frames = n_frames;
    }
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[0], st));
    }
    HIP_TRY(ddn_dev_mbe_params(b->codec, d_bits, d_result_in, b->n_streams, (int)n_frames, b->d_tables, b->d_half_log2,
                               b->d_streams, b->tail_rule, b->d_recs, d_result_out, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[1], st));
    }
    HIP_TRY(ddn_dev_mbe_synth(b->d_recs, (size_t)b->n_streams * n_frames, d_pcm, st));
    if (b->timing) {
        HIP_TRY(hipEventRecord(b->ev[2], st));
        HIP_TRY(hipEventSynchronize(b->ev[2]));
        HIP_TRY(hipEventElapsedTime(&b->last_ms[0], b->ev[0], b->ev[1]));
        HIP_TRY(hipEventElapsedTime(&b->last_ms[1], b->ev[1], b->ev[2]));
    }
    return DDN_OK;
}

// internal (ddn_internal.h): the two kernels of ddn_mbe_synth_batch as separate calls - the chain object runs the parameter kernel
// (one wave per talk path, latency-bound, wants occupancy) ahead of the next call's receive loop and the synthesis kernel (16
// registers, no LDS, throughput-bound) beside it.  Same order on one stream = ddn_mbe_synth_batch.
extern "C" int
ddn_mbe_params_only(ddn_mbe_batch* b, const uint8_t* d_bits, const int32_t* d_result_in, size_t n_frames, int32_t* d_result_out,
                    void* hip_stream) {
    if (!b || (n_frames && !d_bits) || n_frames > (size_t)1 << 24) {
        ddn_set_error("ddn_mbe_params_only: bad argument");
        return DDN_EINVAL;
    }
    if (n_frames == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    if (b->rec_frames < n_frames) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(b->d_recs);
        b->d_recs = nullptr;
        b->rec_frames = 0;
        HIP_TRY(hipMalloc(&b->d_recs, sizeof(DdnMbeFrameRec) * (size_t)b->n_streams * n_frames));
        b->rec_frames = n_frames;
    }
    HIP_TRY(ddn_dev_mbe_params(b->codec, d_bits, d_result_in, b->n_streams, (int)n_frames, b->d_tables, b->d_half_log2, b->d_streams,
                               b->tail_rule, b->d_recs, d_result_out, st));
    return DDN_OK;
}

extern "C" int
ddn_mbe_synth_only(ddn_mbe_batch* b, size_t n_frames, float* d_pcm, void* hip_stream) {
    if (!b || (n_frames && !d_pcm) || n_frames > b->rec_frames) {
        ddn_set_error("ddn_mbe_synth_only: bad argument");
        return DDN_EINVAL;
    }
    if (n_frames == 0) {
        return DDN_OK;
    }
    HIP_TRY(ddn_dev_mbe_synth(b->d_recs, (size_t)b->n_streams * n_frames, d_pcm, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_set_timing(ddn_mbe_batch* b, int enable) {
    if (!b) {
        return DDN_EINVAL;
    }
    if (enable && !b->ev[0]) {
        for (int i = 0; i < 3; i++) {
            HIP_TRY(hipEventCreate(&b->ev[i]));
        }
    }
    b->timing = enable != 0;
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_get_timing(ddn_mbe_batch* b, float* ms2) {
    if (!b || !ms2) {
        return DDN_EINVAL;
    }
    ms2[0] = b->last_ms[0];
    ms2[1] = b->last_ms[1];
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_get_state(ddn_mbe_batch* b, int stream, mbe_parms* cur, mbe_parms* prev, mbe_parms* enh) {
    if (!b || stream < 0 || stream >= b->n_streams) {
        return DDN_EINVAL;
    }
    DdnMbeStream s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_streams + stream, sizeof(s), hipMemcpyDeviceToHost));
    if (cur) {
        *cur = s.cur;
    }
    if (prev) {
        *prev = s.prev;
    }
    if (enh) {
        *enh = s.enh;
    }
    return DDN_OK;
}

extern "C" int
ddn_mbe_batch_set_state(ddn_mbe_batch* b, int stream, const mbe_parms* cur, const mbe_parms* prev, const mbe_parms* enh) {
    if (!b || stream < 0 || stream >= b->n_streams || !cur || !prev || !enh) {
        return DDN_EINVAL;
    }
    DdnMbeStream s;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&s, b->d_streams + stream, sizeof(s), hipMemcpyDeviceToHost));
    s.cur = *cur;
    s.prev = *prev;
    s.enh = *enh;
    s.frame_no = (uint32_t)cur->un; // the talk path's frame counter travels in the struct's spare `un` field
    HIP_TRY(hipMemcpy(b->d_streams + stream, &s, sizeof(s), hipMemcpyHostToDevice));
    return DDN_OK;
}

// ---- mbelib-neo 2.x names ---------------------------------------------------------------------------------------------
extern "C" void
mbe_initMbeParms(mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    if (!cur_mp || !prev_mp || !prev_mp_enhanced) {
        return;
    }
    memset(prev_mp, 0, sizeof(*prev_mp)); // also un = 0: the frame counter that keys the random generator
    prev_mp->w0 = 0.09378f;
    prev_mp->L = 30;
    prev_mp->K = 10;
    for (int l = 0; l <= 56; l++) {
        prev_mp->PSIl[l] = 3.14159265358979323846f / 2.0f;
    }
    *cur_mp = *prev_mp;
    *prev_mp_enhanced = *prev_mp;
}

extern "C" void
mbe_initProcessResult(mbe_process_result* result) {
    if (result) {
        memset(result, 0, sizeof(*result));
    }
}

extern "C" void
mbe_synthesizeSilencef(float* aout_buf) {
    if (aout_buf) {
        memset(aout_buf, 0, sizeof(float) * 160);
    }
}

// "=" per corrected bit, then R (repeat) / M (mute) / T (tone) / E (erasure), as mbelib's err_str
extern "C" void
mbe_formatProcessResult(char* str, size_t size, const mbe_process_result* result) {
    if (!str || size == 0) {
        return;
    }
    size_t o = 0;
    if (result) {
        for (int i = 0; i < result->total_errors && o + 1 < size; i++) {
            str[o++] = '=';
        }
        const struct {
            unsigned f;
            char c;
        } marks[] = {{MBE_PROCESS_FLAG_ERASURE, 'E'}, {MBE_PROCESS_FLAG_TONE, 'T'}, {MBE_PROCESS_FLAG_REPEAT, 'R'}, {MBE_PROCESS_FLAG_MUTE, 'M'}};
        for (const auto& m : marks) {
            if ((result->flags & m.f) && o + 1 < size) {
                str[o++] = m.c;
            }
        }
    }
    str[o] = '\0';
}

namespace {
struct Scratch { // one-frame device buffers shared by the drop-ins (serialised by `mu`)
    std::mutex mu;
    uint8_t* d_in = nullptr;   // 184 B frame
    uint8_t* d_soft = nullptr; // 184 B reliabilities
    uint8_t* d_bits = nullptr; // 88 B
    int32_t* d_res = nullptr;  // 2 x 5
    float* d_pcm = nullptr;    // 160
    ddn_mbe_batch* batch[2] = {nullptr, nullptr};
    bool ok = false;
};

Scratch&
scratch() {
    static Scratch s;
    return s;
}

int
scratch_ready(Scratch& s) {
    if (s.ok) {
        return DDN_OK;
    }
    if (have_device() != DDN_OK) {
        return DDN_ENODEV;
    }
    if (hipMalloc(&s.d_in, 256) != hipSuccess || hipMalloc(&s.d_soft, 256) != hipSuccess
        || hipMalloc(&s.d_bits, 128) != hipSuccess || hipMalloc(&s.d_res, sizeof(int32_t) * 10) != hipSuccess
        || hipMalloc(&s.d_pcm, sizeof(float) * 160) != hipSuccess) {
        return DDN_ENOMEM;
    }
    s.ok = true;
    return DDN_OK;
}

void
unpack_result(const int32_t r[5], mbe_process_result* result) {
    if (result) {
        result->flags = (unsigned)r[0] & ~DDN_MBE_RESULT_INVALID;
        result->c0_errors = r[1];
        result->c4_errors = r[2];
        result->total_errors = r[3];
        result->protected_errors = r[4];
    }
}

int
decode_one(int codec, const uint8_t* bits_in, const uint8_t* soft_in, size_t nbytes, char* out, size_t nout,
           mbe_process_result* result) {
    if (!bits_in || !out) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    Scratch& s = scratch();
    std::lock_guard<std::mutex> lock(s.mu);
    if (scratch_ready(s) != DDN_OK) {
        return MBE_STATUS_NO_DEVICE;
    }
    int32_t r[5];
    if (hipMemcpy(s.d_in, bits_in, nbytes, hipMemcpyHostToDevice) != hipSuccess
        || (soft_in && hipMemcpy(s.d_soft, soft_in, nbytes, hipMemcpyHostToDevice) != hipSuccess)
        || ddn_dev_mbe_frame_decode(codec, s.d_in, soft_in ? s.d_soft : nullptr, 1, s.d_bits, s.d_res, nullptr) != hipSuccess
        || hipMemcpy(out, s.d_bits, nout, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(r, s.d_res, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) {
        return MBE_STATUS_NO_DEVICE;
    }
    if ((unsigned)r[0] & DDN_MBE_RESULT_INVALID) {
        return MBE_STATUS_INVALID_BITS;
    }
    unpack_result(r, result);
    return MBE_STATUS_OK;
}

} // namespace

// The single-stream mbe_* entry points run on two cached one-path batches (one per codec); this loads the integrator's
// table blob into both (creating them if needed).  Until it is called they use the built-in synthetic blob.
extern "C" int
ddn_mbe_dropin_set_tables(const ddn_mbe_tables* t) {
    if (!t) {
        return DDN_EINVAL;
    }
    Scratch& s = scratch();
    std::lock_guard<std::mutex> lock(s.mu);
    const int rc0 = scratch_ready(s);
    if (rc0 != DDN_OK) {
        return rc0;
    }
    for (int codec = 0; codec < 2; codec++) {
        if (!s.batch[codec]) {
            const int rc = ddn_mbe_batch_create(codec, 1, &s.batch[codec]);
            if (rc != DDN_OK) {
                return rc;
            }
        }
        const int rc = ddn_mbe_batch_set_tables(s.batch[codec], t);
        if (rc != DDN_OK) {
            return rc;
        }
    }
    return DDN_OK;
}

namespace {
int
process_one(int codec, float* aout_buf, mbe_process_result* result, const char* bits, size_t nbits, mbe_parms* cur,
            mbe_parms* prev, mbe_parms* enh) {
    if (!aout_buf) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    if (!bits || !cur || !prev || !enh) {
        mbe_synthesizeSilencef(aout_buf);
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    Scratch& s = scratch();
    std::lock_guard<std::mutex> lock(s.mu);
    if (scratch_ready(s) != DDN_OK) {
        mbe_synthesizeSilencef(aout_buf);
        return MBE_STATUS_NO_DEVICE;
    }
    if (!s.batch[codec] && ddn_mbe_batch_create(codec, 1, &s.batch[codec]) != DDN_OK) {
        mbe_synthesizeSilencef(aout_buf);
        return MBE_STATUS_NO_DEVICE;
    }
    ddn_mbe_batch* b = s.batch[codec];
    int32_t r[10] = {0};
    if (result) {
        r[0] = (int32_t)result->flags;
        r[1] = result->c0_errors;
        r[2] = result->c4_errors;
        r[3] = result->total_errors;
        r[4] = result->protected_errors;
    }
    if (ddn_mbe_batch_set_state(b, 0, cur, prev, enh) != DDN_OK
        || hipMemcpy(s.d_bits, bits, nbits, hipMemcpyHostToDevice) != hipSuccess
        || hipMemcpy(s.d_res, r, sizeof(int32_t) * 5, hipMemcpyHostToDevice) != hipSuccess
        || ddn_mbe_synth_batch(b, s.d_bits, s.d_res, 1, s.d_pcm, s.d_res + 5, nullptr) != DDN_OK
        || hipMemcpy(aout_buf, s.d_pcm, sizeof(float) * 160, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(r, s.d_res + 5, sizeof(int32_t) * 5, hipMemcpyDeviceToHost) != hipSuccess
        || ddn_mbe_batch_get_state(b, 0, cur, prev, enh) != DDN_OK) {
        mbe_synthesizeSilencef(aout_buf);
        return MBE_STATUS_NO_DEVICE;
    }
    if ((unsigned)r[0] & DDN_MBE_RESULT_INVALID) {
        return MBE_STATUS_INVALID_BITS;
    }
    unpack_result(r, result);
    return MBE_STATUS_OK;
}

template <int ROWS, int COLS>
void
split_soft(const mbe_soft_bit (*fr)[COLS], uint8_t* bits, uint8_t* rel) {
    for (int r = 0; r < ROWS; r++) {
        for (int c = 0; c < COLS; c++) {
            bits[r * COLS + c] = fr[r][c].bit;
            rel[r * COLS + c] = fr[r][c].reliability;
        }
    }
}
} // namespace

extern "C" int
mbe_decodeImbe7200x4400Frame(const char imbe_fr[8][23], char imbe_d[88], mbe_process_result* result) {
    return decode_one(DDN_MBE_IMBE_7200X4400, (const uint8_t*)imbe_fr, nullptr, 184, imbe_d, 88, result);
}

extern "C" int
mbe_decodeImbe7200x4400SoftFrame(const mbe_soft_bit imbe_fr[8][23], char imbe_d[88], mbe_process_result* result) {
    if (!imbe_fr) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint8_t bits[184], rel[184];
    split_soft<8, 23>(imbe_fr, bits, rel);
    return decode_one(DDN_MBE_IMBE_7200X4400, bits, rel, 184, imbe_d, 88, result);
}

extern "C" int
mbe_decodeAmbe3600x2450Frame(const char ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    return decode_one(DDN_MBE_AMBE_3600X2450, (const uint8_t*)ambe_fr, nullptr, 96, ambe_d, 49, result);
}

extern "C" int
mbe_decodeAmbe3600x2450SoftFrame(const mbe_soft_bit ambe_fr[4][24], char ambe_d[49], mbe_process_result* result) {
    if (!ambe_fr) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    uint8_t bits[96], rel[96];
    split_soft<4, 24>(ambe_fr, bits, rel);
    return decode_one(DDN_MBE_AMBE_3600X2450, bits, rel, 96, ambe_d, 49, result);
}

extern "C" int
mbe_processImbe4400Dataf(float* aout_buf, mbe_process_result* result, const char imbe_d[88], mbe_parms* cur_mp,
                         mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    return process_one(DDN_MBE_IMBE_7200X4400, aout_buf, result, imbe_d, 88, cur_mp, prev_mp, prev_mp_enhanced);
}

extern "C" int
mbe_processAmbe2450Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp,
                         mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    return process_one(DDN_MBE_AMBE_3600X2450, aout_buf, result, ambe_d, 49, cur_mp, prev_mp, prev_mp_enhanced);
}

extern "C" int
mbe_processAmbe3600x2450Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                               mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    mbe_process_result* r = result ? result : &local;
    const int rc = mbe_decodeAmbe3600x2450Frame(ambe_fr, ambe_d, r);
    if (rc < 0) {
        mbe_synthesizeSilencef(aout_buf);
        return rc;
    }
    return mbe_processAmbe2450Dataf(aout_buf, r, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

extern "C" int
mbe_processAmbe3600x2450SoftFramef(float* aout_buf, mbe_process_result* result, const mbe_soft_bit ambe_fr[4][24],
                                   char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    mbe_process_result local;
    mbe_process_result* r = result ? result : &local;
    const int rc = mbe_decodeAmbe3600x2450SoftFrame(ambe_fr, ambe_d, r);
    if (rc < 0) {
        mbe_synthesizeSilencef(aout_buf);
        return rc;
    }
    return mbe_processAmbe2450Dataf(aout_buf, r, ambe_d, cur_mp, prev_mp, prev_mp_enhanced);
}

// ---- the two rates that are not restated (include/ddn_mbe.h): present so that dsd-neo's configure probe links ----------------
extern "C" int
mbe_decodeImbe7100x4400Frame(const char imbe_fr[7][24], char imbe_d[88], mbe_process_result* result) {
    if (!imbe_fr || !imbe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    memset(imbe_d, 0, 88);
    if (result) {
        mbe_initProcessResult(result);
        result->flags = MBE_PROCESS_FLAG_MUTE;
    }
    return MBE_STATUS_UNSUPPORTED;
}

extern "C" int
mbe_processAmbe2400Dataf(float* aout_buf, mbe_process_result* result, const char ambe_d[49], mbe_parms* cur_mp, mbe_parms* prev_mp,
                         mbe_parms* prev_mp_enhanced) {
    (void)cur_mp;
    (void)prev_mp;
    (void)prev_mp_enhanced;
    if (!aout_buf || !ambe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    mbe_synthesizeSilencef(aout_buf);
    if (result) {
        mbe_initProcessResult(result);
        result->flags = MBE_PROCESS_FLAG_MUTE;
    }
    return MBE_STATUS_UNSUPPORTED;
}

// D-STAR's AMBE 3600x2400 frames (src/core/vocoder/dsd_mbe.c:633, mbe_process_dstar): the 2400 rate is not restated (see
// mbe_processAmbe2400Dataf above); the symbol is here so that dsd-neo links, and it says so the same way
extern "C" int
mbe_processAmbe3600x2400Framef(float* aout_buf, mbe_process_result* result, const char ambe_fr[4][24], char ambe_d[49],
                               mbe_parms* cur_mp, mbe_parms* prev_mp, mbe_parms* prev_mp_enhanced) {
    (void)cur_mp;
    (void)prev_mp;
    (void)prev_mp_enhanced;
    if (!aout_buf || !ambe_fr || !ambe_d) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    memset(ambe_d, 0, 49);
    mbe_synthesizeSilencef(aout_buf);
    if (result) {
        mbe_initProcessResult(result);
        result->flags = MBE_PROCESS_FLAG_MUTE;
    }
    return MBE_STATUS_UNSUPPORTED;
}

// src/core/audio/dsd_audio2.c:1376 (soft tones on the short-integer output path): mbelib's float -> short conversion of one
// 160-sample frame - gain 7, clipped to +-32760, truncated toward zero (mbelib 1.3 mbe_floattoshort; host arithmetic on 160 values,
// not a device stage)
extern "C" void
mbe_floattoshort(const float* float_buf, short* aout_buf) {
    if (!float_buf || !aout_buf) {
        return;
    }
    for (int i = 0; i < 160; i++) {
        float a = 7.0f * float_buf[i];
        a = a > 32760.0f ? 32760.0f : (a < -32760.0f ? -32760.0f : a);
        aout_buf[i] = (short)a;
    }
}

// src/runtime/bootstrap/bootstrap.c:695 prints it in the start-up banner
extern "C" const char*
mbe_versionString(void) {
    return "libdsdneo_hip (gfx950) behind the mbelib-neo 2.0 C API";
}
