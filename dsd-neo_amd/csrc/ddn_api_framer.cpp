// ddn_api_framer.cpp — C-ABI of the device-side P25 Phase 1 framer (include/ddn_hip.h, "P25p1 framer"): sync index over
// the receive loop's flags, then field gathers from the capture records into the FEC kernels' input layouts.

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "ddn_device.h"
#include "ddn_internal.h"

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

enum { T_NID = 0, T_BLK0, T_BLK1, T_BLK2, T_LDU1, T_LDU2, T_HDU_HEX, T_HDU_PAR, T_TDULC_DATA, T_TDULC_PAR, T_LSD, T_COUNT };

struct ddn_p25p1_framer {
    int n_channels, max_frames;
    int32_t* d_sync_pos; // [B][F]
    int32_t* d_n_syncs;  // [B]
    int32_t* d_dropped;  // [B] running count of syncs that found no frame slot
    int32_t* d_tab[T_COUNT];
    int n_off[T_COUNT], max_off[T_COUNT];
    int32_t *d_first9, *d_status9;
};

extern "C" int
ddn_p25p1_framer_create(int n_channels, int max_frames_per_channel, ddn_p25p1_framer** out) {
    if (!out || n_channels <= 0 || max_frames_per_channel <= 0) {
        ddn_set_error("ddn_p25p1_framer_create: bad argument");
        return DDN_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ddn_set_error("no HIP device available");
        return DDN_ENODEV;
    }
    ddn_p25p1_framer* f = new (std::nothrow) ddn_p25p1_framer();
    if (!f) {
        return DDN_ENOMEM;
    }
    memset(f, 0, sizeof(*f));
    f->n_channels = n_channels;
    f->max_frames = max_frames_per_channel;
    int32_t tab[T_COUNT][216];
    f->n_off[T_NID] = 32;
    ddn_p25p1_layout_nid(tab[T_NID]);
    for (int b = 0; b < 3; b++) {
        f->n_off[T_BLK0 + b] = 98;
        ddn_p25p1_layout_trellis_block(b, tab[T_BLK0 + b]);
    }
    f->n_off[T_LDU1] = f->n_off[T_LDU2] = 120;
    ddn_p25p1_layout_ldu_words(1, tab[T_LDU1]);
    ddn_p25p1_layout_ldu_words(2, tab[T_LDU2]);
    f->n_off[T_HDU_HEX] = 108;
    f->n_off[T_HDU_PAR] = 216;
    ddn_p25p1_layout_hdu(tab[T_HDU_HEX], tab[T_HDU_PAR]);
    f->n_off[T_TDULC_DATA] = f->n_off[T_TDULC_PAR] = 72;
    ddn_p25p1_layout_tdulc(tab[T_TDULC_DATA], tab[T_TDULC_PAR]);
    f->n_off[T_LSD] = 16;
    ddn_p25p1_layout_ldu_lsd(tab[T_LSD]);
    int32_t first9[9], status9[9];
    ddn_p25p1_layout_ldu_imbe(first9, status9);
    const size_t slots = (size_t)n_channels * (size_t)max_frames_per_channel;
    bool ok = hipMalloc(&f->d_sync_pos, sizeof(int32_t) * slots) == hipSuccess
              && hipMalloc(&f->d_n_syncs, sizeof(int32_t) * (size_t)n_channels) == hipSuccess
              && hipMemset(f->d_n_syncs, 0, sizeof(int32_t) * (size_t)n_channels) == hipSuccess
              && hipMalloc(&f->d_dropped, sizeof(int32_t) * (size_t)n_channels) == hipSuccess
              && hipMemset(f->d_dropped, 0, sizeof(int32_t) * (size_t)n_channels) == hipSuccess
              && hipMalloc(&f->d_first9, sizeof(first9)) == hipSuccess
              && hipMalloc(&f->d_status9, sizeof(status9)) == hipSuccess
              && hipMemcpy(f->d_first9, first9, sizeof(first9), hipMemcpyHostToDevice) == hipSuccess
              && hipMemcpy(f->d_status9, status9, sizeof(status9), hipMemcpyHostToDevice) == hipSuccess;
    for (int t = 0; ok && t < T_COUNT; t++) {
        f->max_off[t] = 0;
        for (int i = 0; i < f->n_off[t]; i++) {
            f->max_off[t] = tab[t][i] > f->max_off[t] ? tab[t][i] : f->max_off[t];
        }
        ok = hipMalloc(&f->d_tab[t], sizeof(int32_t) * (size_t)f->n_off[t]) == hipSuccess
             && hipMemcpy(f->d_tab[t], tab[t], sizeof(int32_t) * (size_t)f->n_off[t], hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        ddn_set_error("ddn_p25p1_framer_create: device allocation failed");
        ddn_p25p1_framer_destroy(f);
        return DDN_ENOMEM;
    }
    *out = f;
    return DDN_OK;
}

extern "C" void
ddn_p25p1_framer_destroy(ddn_p25p1_framer* f) {
    if (!f) {
        return;
    }
    (void)hipFree(f->d_sync_pos);
    (void)hipFree(f->d_n_syncs);
    (void)hipFree(f->d_dropped);
    (void)hipFree(f->d_first9);
    (void)hipFree(f->d_status9);
    for (int t = 0; t < T_COUNT; t++) {
        (void)hipFree(f->d_tab[t]);
    }
    delete f;
}

extern "C" int
ddn_p25p1_framer_index(ddn_p25p1_framer* f, const uint8_t* d_flags, const int32_t* d_counts, size_t max_symbols,
                       void* hip_stream) {
    if (!f || !d_flags || !d_counts) {
        ddn_set_error("ddn_p25p1_framer_index: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_find_syncs(d_flags, d_counts, f->n_channels, max_symbols, f->max_frames, f->d_sync_pos,
                               f->d_n_syncs, f->d_dropped, (hipStream_t)hip_stream));
    return DDN_OK;
}

// device views of the index (valid until the next ddn_p25p1_framer_index on the same object)
extern "C" int
ddn_p25p1_framer_device_syncs(ddn_p25p1_framer* f, const int32_t** d_n_syncs, const int32_t** d_sync_pos) {
    if (!f || !d_n_syncs || !d_sync_pos) {
        return DDN_EINVAL;
    }
    *d_n_syncs = f->d_n_syncs;
    *d_sync_pos = f->d_sync_pos;
    return DDN_OK;
}

// [B] running count (since the object was created) of accepted syncs that found no frame slot in their call: 0 unless
// max_frames_per_channel is too small for the traffic
extern "C" int
ddn_p25p1_framer_device_dropped(ddn_p25p1_framer* f, const int32_t** d_dropped) {
    if (!f || !d_dropped) {
        return DDN_EINVAL;
    }
    *d_dropped = f->d_dropped;
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_get_syncs(ddn_p25p1_framer* f, int32_t* n_syncs, int32_t* sync_pos) {
    if (!f || !n_syncs) {
        return DDN_EINVAL;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(n_syncs, f->d_n_syncs, sizeof(int32_t) * (size_t)f->n_channels, hipMemcpyDeviceToHost));
    if (sync_pos) {
        HIP_TRY(hipMemcpy(sync_pos, f->d_sync_pos, sizeof(int32_t) * (size_t)f->n_channels * (size_t)f->max_frames,
                          hipMemcpyDeviceToHost));
    }
    return DDN_OK;
}

static int
gather(ddn_p25p1_framer* f, int t, const uint8_t* d_rec, const int32_t* d_counts, size_t max_symbols, uint8_t* bits,
       uint8_t* rel, int16_t* llr, int stride, int split, uint8_t* last_bit, uint8_t* last_rel, uint8_t* valid,
       void* hip_stream, uint8_t* dibits = nullptr, uint8_t* dibit_rel = nullptr) {
    if (!f || !d_rec || !d_counts) {
        ddn_set_error("ddn_p25p1_framer_gather_*: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_gather_fields(d_rec, max_symbols, d_counts, f->d_sync_pos, f->d_n_syncs, f->n_channels, f->max_frames,
                                  f->d_tab[t], f->n_off[t], f->max_off[t], bits, rel, llr, stride, split, last_bit,
                                  last_rel, valid, dibits, dibit_rel, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_gather_nid(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                            uint8_t* d_bits63, uint8_t* d_reliab63, uint8_t* d_parity, uint8_t* d_parity_reliab,
                            uint8_t* d_valid, void* hip_stream) {
    if (!d_bits63 || !d_reliab63 || !d_parity || !d_parity_reliab) {
        ddn_set_error("ddn_p25p1_framer_gather_nid: null argument");
        return DDN_EINVAL;
    }
    return gather(f, T_NID, d_records10, d_counts, max_symbols, d_bits63, d_reliab63, nullptr, 63, 1, d_parity,
                  d_parity_reliab, d_valid, hip_stream);
}

extern "C" int
ddn_p25p1_framer_gather_trellis_block(ddn_p25p1_framer* f, int block, const uint8_t* d_records10, const int32_t* d_counts,
                                      size_t max_symbols, int16_t* d_llr196, uint8_t* d_dibit_bits196, uint8_t* d_valid,
                                      void* hip_stream) {
    if (block < 0 || block > 2 || (!d_llr196 && !d_dibit_bits196)) {
        ddn_set_error("ddn_p25p1_framer_gather_trellis_block: bad argument");
        return DDN_EINVAL;
    }
    return gather(f, T_BLK0 + block, d_records10, d_counts, max_symbols, d_dibit_bits196, nullptr, d_llr196, 196, 0,
                  nullptr, nullptr, d_valid, hip_stream);
}

extern "C" int
ddn_p25p1_framer_gather_r34_block(ddn_p25p1_framer* f, int block, const uint8_t* d_records10, const int32_t* d_counts,
                                  size_t max_symbols, uint8_t* d_dibits98, uint8_t* d_reliab98, uint8_t* d_valid,
                                  void* hip_stream) {
    if (block < 0 || block > 2 || !d_dibits98) {
        ddn_set_error("ddn_p25p1_framer_gather_r34_block: bad argument");
        return DDN_EINVAL;
    }
    return gather(f, T_BLK0 + block, d_records10, d_counts, max_symbols, nullptr, nullptr, nullptr, 196, 0, nullptr,
                  nullptr, d_valid, hip_stream, d_dibits98, d_reliab98);
}

extern "C" int
ddn_p25p1_framer_gather_ldu_words(ddn_p25p1_framer* f, int ldu, const uint8_t* d_records10, const int32_t* d_counts,
                                  size_t max_symbols, uint8_t* d_bits240, uint8_t* d_reliab240, uint8_t* d_valid,
                                  void* hip_stream) {
    if ((ldu != 1 && ldu != 2) || !d_bits240) {
        ddn_set_error("ddn_p25p1_framer_gather_ldu_words: bad argument");
        return DDN_EINVAL;
    }
    return gather(f, ldu == 1 ? T_LDU1 : T_LDU2, d_records10, d_counts, max_symbols, d_bits240, d_reliab240, nullptr,
                  240, 0, nullptr, nullptr, d_valid, hip_stream);
}

extern "C" int
ddn_p25p1_framer_imbe_index(ddn_p25p1_framer* f, size_t max_symbols, int64_t* d_first_record, int32_t* d_status_count,
                            void* hip_stream) {
    if (!f || !d_first_record || !d_status_count) {
        ddn_set_error("ddn_p25p1_framer_imbe_index: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_imbe_index(f->d_sync_pos, f->d_n_syncs, f->n_channels, f->max_frames, max_symbols, f->d_first9,
                               f->d_status9, d_first_record, d_status_count, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_voice_index(ddn_p25p1_framer* f, const int32_t* d_nid4, const int32_t* d_counts, int max_ldu_per_channel, size_t max_symbols,
                             int64_t* d_first_record, int32_t* d_status_count, int32_t* d_n_ldu, void* hip_stream) {
    if (!f || !d_nid4 || !d_counts || max_ldu_per_channel <= 0 || !d_first_record || !d_status_count) {
        ddn_set_error("ddn_p25p1_framer_voice_index: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_voice_index(f->d_sync_pos, f->d_n_syncs, d_nid4, d_counts, f->n_channels, f->max_frames, max_ldu_per_channel,
                                max_symbols, f->d_first9, f->d_status9, d_first_record, d_status_count, d_n_ldu,
                                (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_pack_ldu_rs(ddn_p25p1_framer* f, int ldu, const uint8_t* d_words240, uint8_t* d_data_bits,
                             uint8_t* d_parity_bits, void* hip_stream) {
    if (!f || (ldu != 1 && ldu != 2) || !d_words240 || !d_data_bits || !d_parity_bits) {
        ddn_set_error("ddn_p25p1_framer_pack_ldu_rs: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_rs_pack(d_words240, (long)f->n_channels * f->max_frames, 24, 10, ldu == 1 ? 12 : 16, d_data_bits,
                            d_parity_bits, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_gather_hdu(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                            uint8_t* d_hex_bits216, uint8_t* d_parity_bits432, int16_t* d_hex_llr216,
                            int16_t* d_parity_llr432, uint8_t* d_valid, void* hip_stream) {
    if (!d_hex_bits216 || !d_parity_bits432) {
        ddn_set_error("ddn_p25p1_framer_gather_hdu: null argument");
        return DDN_EINVAL;
    }
    int rc = gather(f, T_HDU_HEX, d_records10, d_counts, max_symbols, d_hex_bits216, nullptr, d_hex_llr216, 216, 0,
                    nullptr, nullptr, nullptr, hip_stream);
    if (rc != DDN_OK) {
        return rc;
    }
    // the parity field ends last, so its "fits inside this call's records" flag covers the whole header
    return gather(f, T_HDU_PAR, d_records10, d_counts, max_symbols, d_parity_bits432, nullptr, d_parity_llr432, 432, 0,
                  nullptr, nullptr, d_valid, hip_stream);
}

extern "C" int
ddn_p25p1_framer_pack_hdu_rs(ddn_p25p1_framer* f, const uint8_t* d_hex_bits216, uint8_t* d_data_bits,
                             uint8_t* d_parity_bits, void* hip_stream) {
    if (!f || !d_hex_bits216 || !d_data_bits || !d_parity_bits) {
        ddn_set_error("ddn_p25p1_framer_pack_hdu_rs: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_rs_pack(d_hex_bits216, (long)f->n_channels * f->max_frames, 36, 6, 20, d_data_bits, d_parity_bits,
                            (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_gather_tdulc(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                              uint8_t* d_data_bits144, uint8_t* d_parity_bits144, int16_t* d_data_llr144,
                              int16_t* d_parity_llr144, uint8_t* d_valid, void* hip_stream) {
    if (!d_data_bits144 || !d_parity_bits144) {
        ddn_set_error("ddn_p25p1_framer_gather_tdulc: null argument");
        return DDN_EINVAL;
    }
    int rc = gather(f, T_TDULC_DATA, d_records10, d_counts, max_symbols, d_data_bits144, nullptr, d_data_llr144, 144, 0,
                    nullptr, nullptr, nullptr, hip_stream);
    if (rc != DDN_OK) {
        return rc;
    }
    return gather(f, T_TDULC_PAR, d_records10, d_counts, max_symbols, d_parity_bits144, nullptr, d_parity_llr144, 144, 0,
                  nullptr, nullptr, d_valid, hip_stream);
}

extern "C" int
ddn_p25p1_framer_pack_tdulc_rs(ddn_p25p1_framer* f, const uint8_t* d_data_bits144, uint8_t* d_rs_data_bits,
                               uint8_t* d_rs_parity_bits, void* hip_stream) {
    if (!f || !d_data_bits144 || !d_rs_data_bits || !d_rs_parity_bits) {
        ddn_set_error("ddn_p25p1_framer_pack_tdulc_rs: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_tdulc_rs_pack(d_data_bits144, (long)f->n_channels * f->max_frames, d_rs_data_bits, d_rs_parity_bits,
                                  (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_framer_gather_lsd(ddn_p25p1_framer* f, const uint8_t* d_records10, const int32_t* d_counts, size_t max_symbols,
                            uint8_t* d_bits32, int16_t* d_llr32, uint8_t* d_valid, void* hip_stream) {
    if (!d_bits32) {
        ddn_set_error("ddn_p25p1_framer_gather_lsd: null argument");
        return DDN_EINVAL;
    }
    return gather(f, T_LSD, d_records10, d_counts, max_symbols, d_bits32, nullptr, d_llr32, 32, 0, nullptr, nullptr, d_valid,
                  hip_stream);
}
