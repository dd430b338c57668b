// ddn_api_fec.cpp — C-ABI for the batched trellis / Viterbi decoders (include/ddn_hip.h, "FEC" section) and the
// single-codeword drop-in symbols with the reference's names.
//
// Batched calls take DEVICE pointers and a stream; `_host` variants stage host buffers through the device
// (synchronous).  Nothing here computes on the CPU: without a GPU every call fails with DDN_ENODEV/DDN_EHIP.

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "ddn_device.h"

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

namespace {
struct Dev {
    void* p = nullptr;
    size_t bytes;
    explicit Dev(size_t b) : bytes(b) {
        if (hipMalloc(&p, b ? b : 4) != hipSuccess) {
            p = nullptr;
        }
    }
    ~Dev() { (void)hipFree(p); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    int up(const void* h) { return hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
    int down(void* h) { return hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
};

int
no_dev() {
    ddn_set_error("device allocation/copy failed (no HIP device?)");
    return DDN_ENODEV;
}

int
build_puncture(const uint8_t* punct, int p_len, int in_len, DdnPuncture* pu, int* u_len) {
    memset(pu, 0, sizeof(*pu));
    if (!punct || p_len <= 0) {
        *u_len = in_len;
        return DDN_OK;
    }
    if (p_len > 64) {
        ddn_set_error("puncture pattern longer than 64 entries");
        return DDN_ERANGE;
    }
    pu->p_len = p_len;
    int ones = 0;
    for (int r = 0; r < p_len; r++) {
        pu->keep[r] = punct[r] ? 1 : 0;
        pu->ones_before[r] = (uint8_t)ones;
        ones += pu->keep[r];
    }
    pu->ones_total = ones;
    if (ones == 0) {
        ddn_set_error("puncture pattern keeps nothing");
        return DDN_EINVAL;
    }
    int i = 0, u = 0, p = 0;
    while (i < in_len) { // same walk as viterbi_decode_punctured (reference src/core/util/dsd_misc.c:160-173)
        if (pu->keep[p]) {
            i++;
        }
        u++;
        p = (p + 1) % p_len;
    }
    *u_len = u;
    return DDN_OK;
}
} // namespace

extern "C" int
ddn_fec_p25_12_soft_batch(const int16_t* d_llr196, size_t n, uint8_t* d_out12, int32_t* d_metric, void* hip_stream) {
    if (!d_llr196 || !d_out12) {
        ddn_set_error("ddn_fec_p25_12_soft_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25_half_rate(d_llr196, (int)n, d_out12, d_metric, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_r34_batch(const uint8_t* d_dibits98, const uint8_t* d_reliab98, size_t n, uint8_t* d_out18, void* hip_stream) {
    if (!d_dibits98 || !d_out18) {
        ddn_set_error("ddn_fec_r34_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_r34(d_dibits98, d_reliab98, (int)n, d_out18, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_nxdn_conv_batch(const uint8_t* d_sym, const uint8_t* d_rel, size_t n, int n_steps, int n_bits,
                        uint16_t* d_metrics_io, uint8_t* d_out, int out_stride, void* hip_stream) {
    if (!d_sym || !d_out || n_steps <= 0 || n_steps > 1024 || n_bits < 0 || n_bits > n_steps
        || out_stride < (n_bits + 7) / 8) {
        ddn_set_error("ddn_fec_nxdn_conv_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_k5_nxdn(d_sym, d_rel, (int)n, n_steps, n_bits, d_metrics_io, d_out, out_stride,
                            (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_viterbi_k5_batch(const uint16_t* d_soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* d_out,
                         int out_stride, uint32_t* d_cost, void* hip_stream) {
    if (!d_soft || !d_out || in_len < 2) {
        ddn_set_error("ddn_fec_viterbi_k5_batch: bad argument");
        return DDN_EINVAL;
    }
    DdnPuncture pu;
    int u_len = 0;
    int rc = build_puncture(punct, p_len, in_len, &pu, &u_len);
    if (rc != DDN_OK) {
        return rc;
    }
    if (u_len > 244 * 2) {
        ddn_set_error("viterbi_k5: %d soft bits exceed the decoder's 244-step history", u_len);
        return DDN_ERANGE;
    }
    if (out_stride < (u_len / 2 + 3) / 8 + 1) {
        ddn_set_error("viterbi_k5: out_stride %d too small for %d steps", out_stride, u_len / 2);
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_k5_m17(d_soft, (int)n, in_len, u_len, &pu, d_out, out_stride, d_cost, (hipStream_t)hip_stream));
    return DDN_OK;
}

// the same for callers inside the library that know which code words they want (d_wanted [n], 0 = leave undecoded: the YSF payload's
// dense decode lists are mostly empty for one of the two block lengths)
extern "C" int
ddn_fec_viterbi_k5_batch_wanted(const uint16_t* d_soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* d_out,
                                int out_stride, uint32_t* d_cost, const uint8_t* d_wanted, void* hip_stream) {
    if (!d_soft || !d_out || in_len < 2) {
        ddn_set_error("ddn_fec_viterbi_k5_batch: bad argument");
        return DDN_EINVAL;
    }
    DdnPuncture pu;
    int u_len = 0;
    int rc = build_puncture(punct, p_len, in_len, &pu, &u_len);
    if (rc != DDN_OK) {
        return rc;
    }
    if (u_len > 244 * 2 || out_stride < (u_len / 2 + 3) / 8 + 1) {
        ddn_set_error("viterbi_k5: %d soft bits / out_stride %d out of range", u_len, out_stride);
        return DDN_ERANGE;
    }
    HIP_TRY(ddn_dev_k5_m17_wanted(d_soft, (int)n, in_len, u_len, &pu, d_out, out_stride, d_cost, d_wanted, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_nid_decode_batch(const uint8_t* d_bits63, const uint8_t* d_rel63, const int32_t* d_observed_nac,
                           const uint8_t* d_parity, const uint8_t* d_parity_rel, int erasure_threshold, size_t n,
                           int32_t* d_out4, void* hip_stream) {
    if (!d_bits63 || !d_out4) {
        ddn_set_error("ddn_p25p1_nid_decode_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_nid_decode(d_bits63, d_rel63, d_observed_nac, d_parity, d_parity_rel, erasure_threshold, (int)n,
                               d_out4, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_hamming_10_6_3_batch(uint8_t* d_bits10, size_t n, uint8_t* d_errs, void* hip_stream) {
    if (!d_bits10 || !d_errs) {
        ddn_set_error("ddn_fec_hamming_10_6_3_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_hamming_10_6_3(d_bits10, (int)n, d_errs, (hipStream_t)hip_stream));
    return DDN_OK;
}

// ---- host-buffer variants ------------------------------------------------------------------------------
extern "C" int
ddn_fec_p25_12_soft_host(const int16_t* llr196, size_t n, uint8_t* out12, int32_t* metric) {
    Dev a(n * 196 * 2), b(n * 12), m(n * 4);
    if (!a.p || !b.p || !m.p || a.up(llr196)) {
        return no_dev();
    }
    int rc = ddn_fec_p25_12_soft_batch((const int16_t*)a.p, n, (uint8_t*)b.p, (int32_t*)m.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    if (b.down(out12) || (metric && m.down(metric))) {
        return no_dev();
    }
    return DDN_OK;
}

extern "C" int
ddn_fec_r34_host(const uint8_t* dibits98, const uint8_t* reliab98, size_t n, uint8_t* out18) {
    Dev a(n * 98), r(n * 98), b(n * 18);
    if (!a.p || !r.p || !b.p || a.up(dibits98) || (reliab98 && r.up(reliab98))) {
        return no_dev();
    }
    int rc = ddn_fec_r34_batch((const uint8_t*)a.p, reliab98 ? (const uint8_t*)r.p : nullptr, n, (uint8_t*)b.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return b.down(out18) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_fec_nxdn_conv_host(const uint8_t* sym, const uint8_t* rel, size_t n, int n_steps, int n_bits, uint16_t* metrics_io,
                       uint8_t* out, int out_stride) {
    if (n_steps <= 0 || out_stride <= 0) {
        return DDN_EINVAL;
    }
    Dev a(n * 2 * (size_t)n_steps), r(n * 2 * (size_t)n_steps), m(n * 32), b(n * (size_t)out_stride);
    if (!a.p || !r.p || !m.p || !b.p || a.up(sym) || (rel && r.up(rel)) || (metrics_io && m.up(metrics_io))
        || hipMemset(b.p, 0, b.bytes) != hipSuccess) {
        return no_dev();
    }
    int rc = ddn_fec_nxdn_conv_batch((const uint8_t*)a.p, rel ? (const uint8_t*)r.p : nullptr, n, n_steps, n_bits,
                                     metrics_io ? (uint16_t*)m.p : nullptr, (uint8_t*)b.p, out_stride, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    if (b.down(out) || (metrics_io && m.down(metrics_io))) {
        return no_dev();
    }
    return DDN_OK;
}

extern "C" int
ddn_fec_viterbi_k5_host(const uint16_t* soft, size_t n, int in_len, const uint8_t* punct, int p_len, uint8_t* out,
                        int out_stride, uint32_t* cost) {
    if (in_len < 2 || out_stride <= 0) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)in_len * 2), b(n * (size_t)out_stride), c(n * 4);
    if (!a.p || !b.p || !c.p || a.up(soft)) {
        return no_dev();
    }
    int rc = ddn_fec_viterbi_k5_batch((const uint16_t*)a.p, n, in_len, punct, p_len, (uint8_t*)b.p, out_stride,
                                      (uint32_t*)c.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    if (b.down(out) || (cost && c.down(cost))) {
        return no_dev();
    }
    return DDN_OK;
}

extern "C" int
ddn_p25p1_nid_decode_host(const uint8_t* bits63, const uint8_t* rel63, const int32_t* observed_nac,
                          const uint8_t* parity, const uint8_t* parity_rel, int erasure_threshold, size_t n,
                          int32_t* out4) {
    Dev a(n * 63), r(n * 63), o(n * 4), p(n), q(n), out(n * 16);
    if (!a.p || !r.p || !o.p || !p.p || !q.p || !out.p || a.up(bits63) || (rel63 && r.up(rel63))
        || (observed_nac && o.up(observed_nac)) || (parity && p.up(parity)) || (parity_rel && q.up(parity_rel))) {
        return no_dev();
    }
    int rc = ddn_p25p1_nid_decode_batch((const uint8_t*)a.p, rel63 ? (const uint8_t*)r.p : nullptr,
                                        observed_nac ? (const int32_t*)o.p : nullptr,
                                        parity ? (const uint8_t*)p.p : nullptr,
                                        parity_rel ? (const uint8_t*)q.p : nullptr, erasure_threshold, n,
                                        (int32_t*)out.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return out.down(out4) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_fec_hamming_10_6_3_host(uint8_t* bits10, size_t n, uint8_t* errs) {
    Dev a(n * 10), e(n);
    if (!a.p || !e.p || a.up(bits10)) {
        return no_dev();
    }
    int rc = ddn_fec_hamming_10_6_3_batch((uint8_t*)a.p, n, (uint8_t*)e.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(bits10) || e.down(errs)) ? no_dev() : DDN_OK;
}

// reference: int hamming_10_6_3_decode(char* data, const char* parity)  (include/dsd-neo/fec/block_codes.h)
extern "C" int
hamming_10_6_3_decode(char* data, const char* parity) {
    if (!data || !parity) {
        return 2;
    }
    uint8_t b[10], e = 2;
    for (int i = 0; i < 6; i++) {
        b[i] = (uint8_t)data[i];
    }
    for (int i = 0; i < 4; i++) {
        b[6 + i] = (uint8_t)parity[i];
    }
    if (ddn_fec_hamming_10_6_3_host(b, 1, &e) != DDN_OK) {
        return 2;
    }
    if (e == 1) {
        for (int i = 0; i < 6; i++) {
            data[i] = (char)b[i];
        }
    }
    return e;
}

// C-ABI-safe shape of p25p1_nid_decode() (the reference returns a struct by value,
// include/dsd-neo/protocol/p25/p25p1_check_nid.h:38-39): out4 = {status, nac, duid, error_count}
extern "C" int
ddn_p25p1_nid_decode(const char bch_code[63], const uint8_t* reliab63, int observed_nac, unsigned char parity,
                     uint8_t parity_reliab, int erasure_threshold, int out4[4]) {
    if (!bch_code || !out4) {
        return DDN_EINVAL;
    }
    int32_t obs = observed_nac, o[4];
    int rc = ddn_p25p1_nid_decode_host((const uint8_t*)bch_code, reliab63, &obs, &parity, &parity_reliab,
                                       erasure_threshold, 1, o);
    for (int i = 0; i < 4 && rc == DDN_OK; i++) {
        out4[i] = o[i];
    }
    return rc;
}

// ---- drop-in single-codeword symbols (reference names) ---------------------------------------------------
// include/dsd-neo/protocol/p25/p25_12.h:21, include/dsd-neo/protocol/dmr/r34_viterbi.h:19-26,
// include/dsd-neo/fec/viterbi.h:23-25, include/dsd-neo/protocol/nxdn/nxdn_convolution.h:24-28
extern "C" int
p25_12_soft_llr(const uint8_t* input, const int16_t* bit_llr196, uint8_t treturn[12]) {
    (void)input;
    int32_t metric = 0;
    if (ddn_fec_p25_12_soft_host(bit_llr196, 1, treturn, &metric) != DDN_OK) {
        return -1;
    }
    return metric;
}

extern "C" int
dmr_r34_viterbi_decode(const uint8_t* dibits98, uint8_t out_bytes18[18]) {
    if (!dibits98 || !out_bytes18) {
        return -1;
    }
    return ddn_fec_r34_host(dibits98, nullptr, 1, out_bytes18) == DDN_OK ? 0 : -1;
}

extern "C" int
dmr_r34_viterbi_decode_soft(const uint8_t* dibits98, const uint8_t* reliab98, uint8_t out_bytes18[18]) {
    if (!dibits98 || !reliab98 || !out_bytes18) {
        return -1;
    }
    return ddn_fec_r34_host(dibits98, reliab98, 1, out_bytes18) == DDN_OK ? 0 : -1;
}

extern "C" uint32_t
viterbi_decode(uint8_t* out, const uint16_t* in, const uint16_t len) {
    uint32_t cost = 0;
    const int stride = (len / 2 + 3) / 8 + 1;
    std::vector<uint8_t> tmp((size_t)stride);
    if (ddn_fec_viterbi_k5_host(in, 1, len, nullptr, 0, tmp.data(), stride, &cost) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    memcpy(out, tmp.data(), (size_t)stride);
    return cost;
}

extern "C" uint32_t
viterbi_decode_punctured(uint8_t* out, const uint16_t* in, const uint8_t* punct, const uint16_t in_len,
                         const uint16_t p_len) {
    uint32_t cost = 0;
    DdnPuncture pu;
    int u_len = 0;
    if (build_puncture(punct, p_len, in_len, &pu, &u_len) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    const int stride = (u_len / 2 + 3) / 8 + 1;
    std::vector<uint8_t> tmp((size_t)stride);
    if (ddn_fec_viterbi_k5_host(in, 1, in_len, punct, p_len, tmp.data(), stride, &cost) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    memcpy(out, tmp.data(), (size_t)stride);
    return cost;
}

// Step-wise form of the same decoder (src/core/util/dsd_misc.c:188-283: viterbi_decode_bit updates file-static path
// metrics / history per symbol pair, viterbi_chainback walks them back, viterbi_reset clears them).  The drop-in keeps
// the symbol pairs per thread and runs the accumulated steps on the device at chainback time - the add-compare-select
// of step `pos` depends only on the pairs 0..pos, so the result is the one the incremental form reaches.
namespace {
thread_local std::vector<uint16_t> g_vb_soft;
} // namespace

extern "C" void
viterbi_reset(void) {
    g_vb_soft.clear();
}

extern "C" void
viterbi_decode_bit(uint16_t s0, uint16_t s1, const size_t pos) {
    if (pos >= 244) { // the reference's history array holds 244 steps
        return;
    }
    if (g_vb_soft.size() < 2 * (pos + 1)) {
        g_vb_soft.resize(2 * (pos + 1), 0x7FFF);
    }
    g_vb_soft[2 * pos] = s0;
    g_vb_soft[2 * pos + 1] = s1;
}

extern "C" uint32_t
viterbi_chainback(uint8_t* out, size_t pos, uint16_t len) {
    if (!out || pos == 0 || 2 * pos > g_vb_soft.size() || pos > 244) {
        return 0xFFFFFFFFu;
    }
    // viterbi_decode(out, in, 2 * pos) calls viterbi_chainback(out, pos, pos): bit position = step index + 4 there; a
    // caller's `len` only shifts where the bits land
    uint32_t cost = 0;
    const int stride = (int)((pos + 3) / 8 + 1);
    std::vector<uint8_t> tmp((size_t)stride);
    if (ddn_fec_viterbi_k5_host(g_vb_soft.data(), 1, (int)(2 * pos), nullptr, 0, tmp.data(), stride, &cost) != DDN_OK) {
        return 0xFFFFFFFFu;
    }
    if (len == pos) {
        memcpy(out, tmp.data(), (size_t)((len - 1) / 8 + 1) < (size_t)stride ? (size_t)stride : (size_t)((len - 1) / 8 + 1));
    } else {
        memset(out, 0, (size_t)((len - 1) / 8 + 1));
        for (size_t k = 0; k < pos; k++) { // step k's bit sits at k + 4 in tmp and at len + 4 - pos + k in `out`
            const size_t src = k + 4, dst = (size_t)len + 4 - pos + k;
            if ((tmp[src / 8] >> (7 - (src % 8))) & 1u) {
                out[dst / 8] |= (uint8_t)(1u << (7 - (dst % 8)));
            }
        }
    }
    return cost;
}

// The reference keeps the NXDN decoder's symbols-in-flight and path metrics in file-static storage
// (src/protocol/nxdn/nxdn_convolution.c:48-53); the drop-in keeps the same per-thread streaming contract and runs
// the accumulated steps on the device at chainback time.
namespace {
thread_local std::vector<uint8_t> g_nx_sym, g_nx_rel;
thread_local bool g_nx_soft = false;
thread_local uint16_t g_nx_metrics[16] = {0};
} // namespace

extern "C" void
CNXDNConvolution_init(void) {
    memset(g_nx_metrics, 0, sizeof(g_nx_metrics));
    g_nx_sym.clear();
    g_nx_rel.clear();
    g_nx_soft = false;
}

extern "C" void
CNXDNConvolution_start(void) {
    g_nx_sym.clear();
    g_nx_rel.clear();
    g_nx_soft = false;
}

extern "C" void
CNXDNConvolution_decode(uint8_t s0, uint8_t s1) {
    g_nx_sym.push_back(s0);
    g_nx_sym.push_back(s1);
    g_nx_rel.push_back(0);
    g_nx_rel.push_back(0);
}

extern "C" void
CNXDNConvolution_decode_soft(uint8_t s0, uint8_t s1, uint8_t r0, uint8_t r1) {
    g_nx_sym.push_back(s0);
    g_nx_sym.push_back(s1);
    g_nx_rel.push_back(r0);
    g_nx_rel.push_back(r1);
    g_nx_soft = true;
}

extern "C" void
CNXDNConvolution_chainback(unsigned char* out, unsigned int nBits) {
    const int n_steps = (int)(g_nx_sym.size() / 2);
    if (!out || n_steps <= 0 || (int)nBits > n_steps) {
        return;
    }
    const int stride = ((int)nBits + 7) / 8;
    std::vector<uint8_t> tmp((size_t)(stride > 0 ? stride : 1));
    if (ddn_fec_nxdn_conv_host(g_nx_sym.data(), g_nx_soft ? g_nx_rel.data() : nullptr, 1, n_steps, (int)nBits,
                               g_nx_metrics, tmp.data(), stride > 0 ? stride : 1)
        != DDN_OK) {
        return;
    }
    // the reference writes exactly nBits bits and leaves the rest of the last byte alone
    for (unsigned int i = 0; i < nBits; i++) {
        const uint8_t mask = (uint8_t)(0x80u >> (i & 7));
        if (tmp[i >> 3] & mask) {
            out[i >> 3] |= mask;
        } else {
            out[i >> 3] &= (uint8_t)~mask;
        }
    }
}

// ---- P25p1 Golay(24,12,8)/(18,6,8) and Reed-Solomon GF(64) hard-decision decoders ----------------------------------
static int
rs_code_params(int code, int* n_par, int* n_data, int* t) {
    switch (code) {
        case DDN_RS_24_12_13: *n_par = 12; *n_data = 12; *t = 6; return DDN_OK;
        case DDN_RS_24_16_9: *n_par = 8; *n_data = 16; *t = 4; return DDN_OK;
        case DDN_RS_36_20_17: *n_par = 16; *n_data = 20; *t = 8; return DDN_OK;
        default: ddn_set_error("unknown RS code id %d", code); return DDN_EINVAL;
    }
}

extern "C" int
ddn_fec_golay24_batch(int data_len, uint8_t* d_data_bits, const uint8_t* d_parity12, size_t n, uint8_t* d_status,
                      int32_t* d_fixed, void* hip_stream) {
    if ((data_len != 6 && data_len != 12) || !d_data_bits || !d_parity12 || !d_status) {
        ddn_set_error("ddn_fec_golay24_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_golay24(d_data_bits, d_parity12, data_len, (int)n, d_status, d_fixed, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_golay24_host(int data_len, uint8_t* data_bits, const uint8_t* parity12, size_t n, uint8_t* status,
                     int32_t* fixed) {
    if ((data_len != 6 && data_len != 12) || !data_bits || !parity12 || !status) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)data_len), p(n * 12), s(n), f(n * 4);
    if (!a.p || !p.p || !s.p || !f.p || a.up(data_bits) || p.up(parity12)) {
        return no_dev();
    }
    int rc = ddn_fec_golay24_batch(data_len, (uint8_t*)a.p, (const uint8_t*)p.p, n, (uint8_t*)s.p, (int32_t*)f.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(data_bits) || s.down(status) || (fixed && f.down(fixed))) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_fec_p25_rs_batch(int code, uint8_t* d_data_bits, const uint8_t* d_parity_bits, size_t n, uint8_t* d_status,
                     void* hip_stream) {
    int n_par, n_data, t;
    int rc = rs_code_params(code, &n_par, &n_data, &t);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!d_data_bits || !d_parity_bits || !d_status) {
        ddn_set_error("ddn_fec_p25_rs_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_rs63(d_data_bits, d_parity_bits, n_par, n_data, t, (int)n, d_status, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_rs_host(int code, uint8_t* data_bits, const uint8_t* parity_bits, size_t n, uint8_t* status) {
    int n_par, n_data, t;
    int rc = rs_code_params(code, &n_par, &n_data, &t);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!data_bits || !parity_bits || !status) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)n_data * 6), p(n * (size_t)n_par * 6), s(n);
    if (!a.p || !p.p || !s.p || a.up(data_bits) || p.up(parity_bits)) {
        return no_dev();
    }
    rc = ddn_fec_p25_rs_batch(code, (uint8_t*)a.p, (const uint8_t*)p.p, n, (uint8_t*)s.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(data_bits) || s.down(status)) ? no_dev() : DDN_OK;
}

// reference names, one codeword per call (include/dsd-neo/protocol/p25/p25p1_check_hdu.h, p25p1_check_ldu.h).
// Return 1 ("irrecoverable") when the device path is unavailable, like the reference does when its decoder object
// could not be constructed (src/protocol/p25/phase1/p25p1_check_hdu.cpp:41-44).
static int
golay_one(int len, char* word, const char* parity, int* fixed_errors) {
    if (fixed_errors) {
        *fixed_errors = 0;
    }
    if (!fixed_errors || !word || !parity) {
        return 1;
    }
    uint8_t st = 1;
    int32_t fx = 0;
    if (ddn_fec_golay24_host(len, (uint8_t*)word, (const uint8_t*)parity, 1, &st, &fx) != DDN_OK) {
        return 1;
    }
    *fixed_errors = fx;
    return st;
}

extern "C" int
check_and_fix_golay_24_6(char* hex, const char* parity, int* fixed_errors) {
    return golay_one(6, hex, parity, fixed_errors);
}

extern "C" int
check_and_fix_golay_24_12(char* dodeca, const char* parity, int* fixed_errors) {
    return golay_one(12, dodeca, parity, fixed_errors);
}

static int
rs_one(int code, char* data, const char* parity) {
    uint8_t st = 1;
    if (!data || !parity || ddn_fec_p25_rs_host(code, (uint8_t*)data, (const uint8_t*)parity, 1, &st) != DDN_OK) {
        return 1;
    }
    return st;
}

extern "C" int
check_and_fix_reedsolomon_24_12_13(char* data, const char* parity) {
    return rs_one(DDN_RS_24_12_13, data, parity);
}

extern "C" int
check_and_fix_reedsolomon_24_16_9(char* data, const char* parity) {
    return rs_one(DDN_RS_24_16_9, data, parity);
}

extern "C" int
check_and_fix_redsolomon_36_20_17(char* data, const char* parity) {
    return rs_one(DDN_RS_36_20_17, data, parity);
}

// ---- P25 Phase 2 RS(63,35) sections (ESS / FACCH / SACCH) with caller-given erasures ------------------------------------
static int
rs28_sizes(int kind, int* n_data, int* n_par) {
    static const int nd[3] = {16, 26, 30}, np[3] = {28, 19, 22};
    if (kind < 0 || kind > 2) {
        ddn_set_error("rs28: kind must be DDN_RS28_ESS, _FACCH or _SACCH");
        return DDN_EINVAL;
    }
    *n_data = nd[kind];
    *n_par = np[kind];
    return DDN_OK;
}

extern "C" int
ddn_fec_rs28_batch(int kind, uint8_t* d_payload_bits, const uint8_t* d_parity_bits, const int8_t* d_erasures28,
                   const uint8_t* d_n_erasures, size_t n, int32_t* d_status, void* hip_stream) {
    int nd, np;
    int rc = rs28_sizes(kind, &nd, &np);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!d_payload_bits || !d_parity_bits || !d_status || (d_n_erasures && !d_erasures28)) {
        ddn_set_error("ddn_fec_rs28_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_rs28(kind, d_payload_bits, d_parity_bits, d_erasures28, d_n_erasures, (int)n, d_status,
                         (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_rs28_host(int kind, uint8_t* payload_bits, const uint8_t* parity_bits, const int8_t* erasures28,
                  const uint8_t* n_erasures, size_t n, int32_t* status) {
    int nd, np;
    int rc = rs28_sizes(kind, &nd, &np);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!payload_bits || !parity_bits || !status || (n_erasures && !erasures28)) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)nd * 6), p(n * (size_t)np * 6), e(n * 28), c(n), s(n * sizeof(int32_t));
    if (!a.p || !p.p || !e.p || !c.p || !s.p || a.up(payload_bits) || p.up(parity_bits)) {
        return no_dev();
    }
    if (n_erasures && (e.up(erasures28) || c.up(n_erasures))) {
        return no_dev();
    }
    rc = ddn_fec_rs28_batch(kind, (uint8_t*)a.p, (const uint8_t*)p.p, (const int8_t*)e.p,
                            n_erasures ? (const uint8_t*)c.p : nullptr, n, (int32_t*)s.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(payload_bits) || s.down(status)) ? no_dev() : DDN_OK;
}

// P25 Phase 2 FACCH / SACCH burst stage (include/ddn_hip.h): gather + ranked soft erasures + RS(63,35) with retries
extern "C" int
ddn_p25p2_xcch_batch(int kind, const uint8_t* d_bits360, const int16_t* d_llr360, size_t n, int threshold, uint8_t* d_payload_bits,
                     int32_t* d_ec, uint8_t* d_used_dynamic, void* hip_stream) {
    if ((kind != 0 && kind != 1) || !d_bits360 || !d_llr360 || !d_payload_bits || !d_ec || !d_used_dynamic) {
        ddn_set_error("ddn_p25p2_xcch_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    // scratch (parity bits, erasure lists and their lengths), stream-ordered so concurrent calls never share it
    uint8_t* scratch = nullptr;
    const size_t n_pa = kind == 0 ? 114 : 132;
    HIP_TRY(hipMallocAsync((void**)&scratch, n * (n_pa + 28 + 1) + 64, st));
    uint8_t* parity = scratch;
    int8_t* er = (int8_t*)(scratch + n * n_pa);
    uint8_t* n_total = scratch + n * (n_pa + 28);
    const hipError_t e = ddn_dev_p25p2_xcch(kind, d_bits360, d_llr360, (int)n, threshold, d_payload_bits, parity, er, n_total, d_ec,
                                            d_used_dynamic, st);
    HIP_TRY(hipFreeAsync(scratch, st));
    HIP_TRY(e);
    return DDN_OK;
}

extern "C" int
ddn_p25p2_mac_crc_batch(int kind, const uint8_t* d_payload_bits, size_t n, uint8_t* d_crc12_ok, uint8_t* d_crc16_ok, void* hip_stream) {
    if ((kind != 0 && kind != 1) || !d_payload_bits || !d_crc12_ok) {
        ddn_set_error("ddn_p25p2_mac_crc_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25p2_mac_crc(kind, d_payload_bits, (int)n, d_crc12_ok, d_crc16_ok, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p2_mac_crc_host(int kind, const uint8_t* payload_bits, size_t n, uint8_t* crc12_ok, uint8_t* crc16_ok) {
    if ((kind != 0 && kind != 1) || !payload_bits || !crc12_ok) {
        return DDN_EINVAL;
    }
    const size_t n_pl = kind == 0 ? 156 : 180;
    Dev a(n * n_pl), b(n), c(n);
    if (!a.p || !b.p || !c.p || a.up(payload_bits)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_mac_crc_batch(kind, (const uint8_t*)a.p, n, (uint8_t*)b.p, crc16_ok ? (uint8_t*)c.p : nullptr, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (b.down(crc12_ok) || (crc16_ok && c.down(crc16_ok))) ? no_dev() : DDN_OK;
}

// P25 Phase 2 ESS and voice bursts (include/ddn_hip.h)
extern "C" int
ddn_p25p2_ess_batch(const uint8_t* d_payload_bits96, const int16_t* d_payload_llr96, const uint8_t* d_parity_bits168,
                    const int16_t* d_parity_llr168, size_t n, int threshold, uint8_t* d_payload_out96, int32_t* d_ec, uint8_t* d_used_dynamic,
                    void* hip_stream) {
    if (!d_payload_bits96 || !d_payload_llr96 || !d_parity_bits168 || !d_parity_llr168 || !d_payload_out96 || !d_ec || !d_used_dynamic) {
        ddn_set_error("ddn_p25p2_ess_batch: null argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    uint8_t* scratch = nullptr;
    HIP_TRY(hipMallocAsync((void**)&scratch, n * 29 + 64, st));
    const hipError_t e = ddn_dev_p25p2_ess(d_payload_bits96, d_payload_llr96, d_parity_bits168, d_parity_llr168, (int)n, threshold,
                                           d_payload_out96, (int8_t*)scratch, scratch + n * 28, d_ec, d_used_dynamic, st);
    HIP_TRY(hipFreeAsync(scratch, st));
    HIP_TRY(e);
    return DDN_OK;
}

extern "C" int
ddn_p25p2_ess_host(const uint8_t* payload_bits96, const int16_t* payload_llr96, const uint8_t* parity_bits168, const int16_t* parity_llr168,
                   size_t n, int threshold, uint8_t* payload_out96, int32_t* ec, uint8_t* used_dynamic) {
    if (!payload_bits96 || !payload_llr96 || !parity_bits168 || !parity_llr168 || !payload_out96 || !ec || !used_dynamic) {
        return DDN_EINVAL;
    }
    Dev a(n * 96), al(n * 96 * 2), b(n * 168), bl(n * 168 * 2), o(n * 96), s(n * sizeof(int32_t)), u(n);
    if (!a.p || !al.p || !b.p || !bl.p || !o.p || !s.p || !u.p || a.up(payload_bits96) || al.up(payload_llr96) || b.up(parity_bits168)
        || bl.up(parity_llr168)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_ess_batch((const uint8_t*)a.p, (const int16_t*)al.p, (const uint8_t*)b.p, (const int16_t*)bl.p, n, threshold,
                                       (uint8_t*)o.p, (int32_t*)s.p, (uint8_t*)u.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (o.down(payload_out96) || s.down(ec) || u.down(used_dynamic)) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_p25p2_voice_frames_batch(const uint8_t* d_xbits360, const int16_t* d_xllr360, size_t n, int frame_count, uint8_t* d_ambe_fr,
                             uint8_t* d_ambe_rel, void* hip_stream) {
    if (!d_xbits360 || !d_xllr360 || !d_ambe_fr || !d_ambe_rel || frame_count < 1 || frame_count > 4) {
        ddn_set_error("ddn_p25p2_voice_frames_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25p2_voice_unpack(d_xbits360, d_xllr360, (int)n, frame_count, d_ambe_fr, d_ambe_rel, (hipStream_t)hip_stream));
    return DDN_OK;
}

// P25 Phase 2 frame scrambler (include/ddn_hip.h)
extern "C" int
ddn_p25p2_scramble_bits_batch(const uint64_t* d_seed44, size_t n, size_t bit_count, uint8_t* d_out_bits, void* hip_stream) {
    if (!d_seed44 || !d_out_bits || bit_count > (1u << 20)) {
        ddn_set_error("ddn_p25p2_scramble_bits_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25p2_scramble_bits(d_seed44, (int)n, (int)bit_count, d_out_bits, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p2_descramble_batch(const uint8_t* d_bits, const int16_t* d_llr, const uint8_t* d_scramble4320, const int32_t* d_offset,
                           const int32_t* d_sequence_of, size_t n, int n_bits, int n_llr, uint8_t* d_xbits, int16_t* d_xllr,
                           void* hip_stream) {
    if (!d_bits || !d_scramble4320 || !d_offset || !d_xbits || n_bits <= 0 || n_llr < 0 || n_llr > n_bits || (n_llr > 0 && (!d_llr || !d_xllr))) {
        ddn_set_error("ddn_p25p2_descramble_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25p2_descramble(d_bits, d_llr, d_scramble4320, d_offset, d_sequence_of, (int)n, n_bits, n_llr, d_xbits, d_xllr,
                                     (hipStream_t)hip_stream));
    return DDN_OK;
}

// the reference's name (src/protocol/p25/phase2/p25p2_frame_internal.h:28): one sequence, host pointers, staged through the device
extern "C" void
p25p2_generate_scramble_bits(uint64_t wacn, uint64_t sysid, uint64_t nac, uint8_t* out_bits, size_t bit_count) {
    if (!out_bits || bit_count == 0) {
        return;
    }
    const uint64_t seed = (wacn * 16777216ULL) + (sysid * 4096ULL) + nac;
    Dev a(8), o(bit_count);
    if (!a.p || !o.p || a.up(&seed) || ddn_p25p2_scramble_bits_batch((const uint64_t*)a.p, 1, bit_count, (uint8_t*)o.p, nullptr) != DDN_OK
        || o.down(out_bits)) {
        (void)no_dev(); // (the reference's function cannot fail: the error is in ddn_last_error())
    }
}

extern "C" int
ddn_p25p2_burst_fields_batch(const uint8_t* d_bits360, const int16_t* d_llr360, size_t n, int threshold, int32_t* d_duid, int32_t* d_isch,
                             void* hip_stream) {
    if (!d_bits360 || !d_llr360 || !d_duid || !d_isch) {
        ddn_set_error("ddn_p25p2_burst_fields_batch: null argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    uint8_t* scratch = nullptr; // the I-ISCH words and their reliability rows
    HIP_TRY(hipMallocAsync((void**)&scratch, n * 48 + 64, st));
    uint64_t* words = (uint64_t*)scratch;
    uint8_t* rel = scratch + n * 8;
    hipError_t e = ddn_dev_p25p2_burst_fields(d_bits360, d_llr360, (int)n, threshold, d_duid, words, rel, st);
    if (e == hipSuccess) {
        e = ddn_dev_isch_lookup(words, rel, (int)n, d_isch, st);
    }
    HIP_TRY(hipFreeAsync(scratch, st));
    HIP_TRY(e);
    return DDN_OK;
}

extern "C" int
ddn_p25p2_burst_fields_host(const uint8_t* bits360, const int16_t* llr360, size_t n, int threshold, int32_t* duid, int32_t* isch) {
    if (!bits360 || !llr360 || !duid || !isch) {
        return DDN_EINVAL;
    }
    Dev b(n * 360), l(n * 360 * 2), d(n * sizeof(int32_t)), q(n * sizeof(int32_t));
    if (!b.p || !l.p || !d.p || !q.p || b.up(bits360) || l.up(llr360)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_burst_fields_batch((const uint8_t*)b.p, (const int16_t*)l.p, n, threshold, (int32_t*)d.p, (int32_t*)q.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (d.down(duid) || q.down(isch)) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_p25p2_xcch_host(int kind, const uint8_t* bits360, const int16_t* llr360, size_t n, int threshold, uint8_t* payload_bits, int32_t* ec,
                    uint8_t* used_dynamic) {
    if ((kind != 0 && kind != 1) || !bits360 || !llr360 || !payload_bits || !ec || !used_dynamic) {
        return DDN_EINVAL;
    }
    const size_t n_pl = kind == 0 ? 156 : 180;
    Dev b(n * 360), l(n * 360 * 2), p(n * n_pl), s(n * sizeof(int32_t)), u(n);
    if (!b.p || !l.p || !p.p || !s.p || !u.p || b.up(bits360) || l.up(llr360)) {
        return no_dev();
    }
    const int rc = ddn_p25p2_xcch_batch(kind, (const uint8_t*)b.p, (const int16_t*)l.p, n, threshold, (uint8_t*)p.p, (int32_t*)s.p,
                                        (uint8_t*)u.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (p.down(payload_bits) || s.down(ec) || u.down(used_dynamic)) ? no_dev() : DDN_OK;
}

// reference names (include/dsd-neo/fec/ez.h:29-31): one section per call, int-per-bit arrays.  -2 when the device path is
// unavailable, the value the reference returns when its decoder object could not be constructed (src/fec/ez.cpp:106-117).
static int
rs28_one(int kind, int* payload, const int* parity, const int* erasures, int n_erasures) {
    int nd, np;
    if (!payload || !parity || rs28_sizes(kind, &nd, &np) != DDN_OK) {
        return -2;
    }
    uint8_t pl[180], pa[168];
    int8_t er[28] = {0};
    for (int i = 0; i < nd * 6; i++) {
        pl[i] = (uint8_t)(payload[i] & 1);
    }
    for (int i = 0; i < np * 6; i++) {
        pa[i] = (uint8_t)(parity[i] & 1);
    }
    uint8_t ne = 0;
    for (int i = 0; erasures && i < n_erasures && i < 28; i++) {
        er[ne++] = (int8_t)erasures[i];
    }
    int32_t st = -2;
    if (ddn_fec_rs28_host(kind, pl, pa, er, &ne, 1, &st) != DDN_OK) {
        return -2;
    }
    for (int i = 0; i < nd * 6; i++) {
        payload[i] = pl[i];
    }
    return st;
}

extern "C" int
ez_rs28_ess(int payload[96], int parity[168], const int* erasures, int n_erasures) {
    return rs28_one(0, payload, parity, erasures, n_erasures);
}

extern "C" int
ez_rs28_facch(int payload[156], int parity[114], const int* erasures, int n_erasures) {
    return rs28_one(1, payload, parity, erasures, n_erasures);
}

extern "C" int
ez_rs28_sacch(int payload[180], int parity[132], const int* erasures, int n_erasures) {
    return rs28_one(2, payload, parity, erasures, n_erasures);
}

// ---- P25 Phase 2 I-ISCH lookup ----------------------------------------------------------------------------------------------
extern "C" int
ddn_fec_isch_lookup_batch(const uint64_t* d_words, const uint8_t* d_reliab40, size_t n, int32_t* d_out, void* stream) {
    if (!d_words || !d_out) {
        ddn_set_error("ddn_fec_isch_lookup_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_isch_lookup(d_words, d_reliab40, (int)n, d_out, (hipStream_t)stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_isch_lookup_host(const uint64_t* words, const uint8_t* reliab40, size_t n, int32_t* out) {
    if (!words || !out) {
        return DDN_EINVAL;
    }
    Dev w(n * sizeof(uint64_t)), r(n * 40), o(n * sizeof(int32_t));
    if (!w.p || !r.p || !o.p || w.up(words) || (reliab40 && r.up(reliab40))) {
        return no_dev();
    }
    int rc = ddn_fec_isch_lookup_batch((const uint64_t*)w.p, reliab40 ? (const uint8_t*)r.p : nullptr, n, (int32_t*)o.p,
                                       nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return o.down(out) ? no_dev() : DDN_OK;
}

// reference names (include/dsd-neo/fec/ez.h): one word per call; -2 (the "nothing found" answer) when no device is there
extern "C" int
isch_lookup(uint64_t isch) {
    int32_t v = -2;
    return ddn_fec_isch_lookup_host(&isch, nullptr, 1, &v) == DDN_OK ? v : -2;
}

extern "C" int
isch_lookup_soft(uint64_t isch, const uint8_t reliab40[40]) {
    int32_t v = -2;
    return ddn_fec_isch_lookup_host(&isch, reliab40, 1, &v) == DDN_OK ? v : -2;
}

// ---- P25 1/2-rate list decoder -------------------------------------------------------------------------------------
extern "C" int
ddn_fec_p25_12_soft_list_batch(const int16_t* d_llr196, size_t n, int max_candidates, ddn_p25_12_candidate* d_candidates8,
                               int32_t* d_counts, void* hip_stream) {
    if (!d_llr196 || !d_candidates8 || !d_counts || max_candidates <= 0) {
        ddn_set_error("ddn_fec_p25_12_soft_list_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25_half_rate_list(d_llr196, (int)n, max_candidates, (uint32_t*)d_candidates8, d_counts,
                                       (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_12_soft_list_host(const int16_t* llr196, size_t n, int max_candidates, ddn_p25_12_candidate* candidates8,
                              int32_t* counts) {
    if (!llr196 || !candidates8 || !counts || max_candidates <= 0) {
        return DDN_EINVAL;
    }
    Dev a(n * 196 * 2), c(n * 8 * sizeof(ddn_p25_12_candidate)), k(n * 4);
    if (!a.p || !c.p || !k.p || a.up(llr196)) {
        return no_dev();
    }
    int rc = ddn_fec_p25_12_soft_list_batch((const int16_t*)a.p, n, max_candidates, (ddn_p25_12_candidate*)c.p,
                                            (int32_t*)k.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (c.down(candidates8) || k.down(counts)) ? no_dev() : DDN_OK;
}

// reference: int p25_12_soft_llr_list(const uint8_t* input, const int16_t* bit_llr196, p25_12_candidate_t* candidates,
//                                     int max_candidates)   (include/dsd-neo/protocol/p25/p25_12.h:31-32)
extern "C" int
p25_12_soft_llr_list(const uint8_t* input, const int16_t* bit_llr196, ddn_p25_12_candidate* candidates,
                     int max_candidates) {
    (void)input;
    if (!bit_llr196 || !candidates || max_candidates <= 0) {
        return 0;
    }
    ddn_p25_12_candidate tmp[8];
    int32_t cnt = 0;
    if (ddn_fec_p25_12_soft_list_host(bit_llr196, 1, max_candidates, tmp, &cnt) != DDN_OK) {
        return 0;
    }
    for (int i = 0; i < cnt; i++) {
        candidates[i] = tmp[i];
    }
    return cnt;
}

// ---- P25 confirmed data: rate 3/4 blocks on LLR pairs (p25p1_mbf34.c) --------------------------------------------------------
extern "C" int
ddn_fec_p25_mbf34_list_batch(const int16_t* d_llr196, size_t n, int max_candidates, const uint8_t* d_wanted,
                             ddn_p25_mbf34_candidate* d_candidates8, int32_t* d_counts, void* hip_stream) {
    if (!d_llr196 || !d_candidates8 || !d_counts || max_candidates <= 0) {
        ddn_set_error("ddn_fec_p25_mbf34_list_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    HIP_TRY(ddn_dev_p25_mbf34_list(d_llr196, (int)n, max_candidates, d_wanted, (uint8_t*)d_candidates8, d_counts, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_mbf34_list_host(const int16_t* llr196, size_t n, int max_candidates, ddn_p25_mbf34_candidate* candidates8, int32_t* counts) {
    if (!llr196 || !candidates8 || !counts || max_candidates <= 0) {
        return DDN_EINVAL;
    }
    Dev a(n * 196 * 2), c(n * 8 * sizeof(ddn_p25_mbf34_candidate)), k(n * 4);
    if (!a.p || !c.p || !k.p || a.up(llr196)) {
        return no_dev();
    }
    const int rc = ddn_fec_p25_mbf34_list_batch((const int16_t*)a.p, n, max_candidates, nullptr, (ddn_p25_mbf34_candidate*)c.p, (int32_t*)k.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (c.down(candidates8) || k.down(counts)) ? no_dev() : DDN_OK;
}

// reference: p25_mbf34_decode_soft_list (include/dsd-neo/protocol/p25/p25p1_mbf34.h:28-29); dibits is unused there too
extern "C" int
p25_mbf34_decode_soft_list(const uint8_t dibits[98], const int16_t bit_llr[196], ddn_p25_mbf34_candidate* candidates, int max_candidates) {
    if (!dibits || !bit_llr || !candidates || max_candidates <= 0) {
        return 0;
    }
    ddn_p25_mbf34_candidate tmp[8];
    int32_t cnt = 0;
    if (ddn_fec_p25_mbf34_list_host(bit_llr, 1, max_candidates, tmp, &cnt) != DDN_OK) {
        return 0;
    }
    for (int i = 0; i < cnt; i++) {
        candidates[i] = tmp[i];
    }
    return cnt;
}

// ---- 3/4-rate list decoder -------------------------------------------------------------------------------------------
extern "C" int
ddn_fec_r34_list_batch(const uint8_t* d_dibits98, const uint8_t* d_reliab98, size_t n, int max_candidates,
                       ddn_r34_candidate* d_candidates32, int32_t* d_counts, void* hip_stream) {
    if (!d_dibits98 || !d_candidates32 || !d_counts || max_candidates <= 0) {
        ddn_set_error("ddn_fec_r34_list_batch: bad argument");
        return DDN_EINVAL;
    }
    if (n == 0) {
        return DDN_OK;
    }
    // [n][49][8][32] back-pointer scratch, stream-ordered so concurrent calls never share it
    uint8_t* backs = nullptr;
    HIP_TRY(hipMallocAsync((void**)&backs, n * 49 * 8 * 32, (hipStream_t)hip_stream));
    const hipError_t e = ddn_dev_r34_list(d_dibits98, d_reliab98, (int)n, max_candidates, backs,
                                          (uint32_t*)d_candidates32, d_counts, (hipStream_t)hip_stream);
    HIP_TRY(hipFreeAsync(backs, (hipStream_t)hip_stream));
    HIP_TRY(e);
    return DDN_OK;
}

extern "C" int
ddn_fec_r34_list_host(const uint8_t* dibits98, const uint8_t* reliab98, size_t n, int max_candidates,
                      ddn_r34_candidate* candidates32, int32_t* counts) {
    if (!dibits98 || !candidates32 || !counts || max_candidates <= 0) {
        return DDN_EINVAL;
    }
    Dev a(n * 98), r(n * 98), c(n * 32 * sizeof(ddn_r34_candidate)), k(n * 4);
    if (!a.p || !r.p || !c.p || !k.p || a.up(dibits98) || (reliab98 && r.up(reliab98))) {
        return no_dev();
    }
    int rc = ddn_fec_r34_list_batch((const uint8_t*)a.p, reliab98 ? (const uint8_t*)r.p : nullptr, n, max_candidates,
                                    (ddn_r34_candidate*)c.p, (int32_t*)k.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (c.down(candidates32) || k.down(counts)) ? no_dev() : DDN_OK;
}

// reference: dmr_r34_viterbi_decode_list (include/dsd-neo/protocol/dmr/r34_viterbi.h:51-70)
extern "C" int
dmr_r34_viterbi_decode_list(const uint8_t* dibits98, const uint8_t* reliab98, ddn_r34_candidate* out_candidates,
                            int max_candidates, int* out_count) {
    if (!dibits98 || !out_candidates || !out_count || max_candidates <= 0) {
        return -1;
    }
    ddn_r34_candidate tmp[32];
    int32_t cnt = 0;
    if (ddn_fec_r34_list_host(dibits98, reliab98, 1, max_candidates, tmp, &cnt) != DDN_OK) {
        return -1;
    }
    for (int i = 0; i < cnt; i++) {
        out_candidates[i] = tmp[i];
    }
    *out_count = cnt;
    return 0;
}

// ---- soft (Chase) Golay / Hamming ------------------------------------------------------------------------------------
extern "C" int
ddn_fec_golay24_soft_batch(int data_len, uint8_t* d_data_bits, const uint8_t* d_parity12, const int32_t* d_reliab,
                           size_t n, uint8_t* d_status, int32_t* d_fixed, void* hip_stream) {
    if ((data_len != 6 && data_len != 12) || !d_data_bits || !d_parity12 || !d_reliab || !d_status) {
        ddn_set_error("ddn_fec_golay24_soft_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_golay24_soft(d_data_bits, d_parity12, d_reliab, data_len, (int)n, d_status, d_fixed,
                                 (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_golay24_soft_host(int data_len, uint8_t* data_bits, const uint8_t* parity12, const int32_t* reliab, size_t n,
                          uint8_t* status, int32_t* fixed) {
    if ((data_len != 6 && data_len != 12) || !data_bits || !parity12 || !reliab || !status) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)data_len), p(n * 12), r(n * (size_t)(data_len + 12) * 4), s(n), f(n * 4);
    if (!a.p || !p.p || !r.p || !s.p || !f.p || a.up(data_bits) || p.up(parity12) || r.up(reliab)) {
        return no_dev();
    }
    int rc = ddn_fec_golay24_soft_batch(data_len, (uint8_t*)a.p, (const uint8_t*)p.p, (const int32_t*)r.p, n,
                                        (uint8_t*)s.p, (int32_t*)f.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(data_bits) || s.down(status) || (fixed && f.down(fixed))) ? no_dev() : DDN_OK;
}

extern "C" int
ddn_fec_hamming_10_6_3_soft_batch(const uint8_t* d_bits10, const int32_t* d_reliab10, size_t n, uint8_t* d_out10,
                                  uint8_t* d_status, void* hip_stream) {
    if (!d_bits10 || !d_reliab10 || !d_out10 || !d_status) {
        ddn_set_error("ddn_fec_hamming_10_6_3_soft_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_hamming_10_6_3_soft(d_bits10, d_reliab10, (int)n, d_out10, d_status, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_hamming_10_6_3_soft_host(const uint8_t* bits10, const int32_t* reliab10, size_t n, uint8_t* out10,
                                 uint8_t* status) {
    if (!bits10 || !reliab10 || !out10 || !status) {
        return DDN_EINVAL;
    }
    Dev a(n * 10), r(n * 40), o(n * 10), s(n);
    if (!a.p || !r.p || !o.p || !s.p || a.up(bits10) || r.up(reliab10)) {
        return no_dev();
    }
    int rc = ddn_fec_hamming_10_6_3_soft_batch((const uint8_t*)a.p, (const int32_t*)r.p, n, (uint8_t*)o.p, (uint8_t*)s.p,
                                               nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (o.down(out10) || s.down(status)) ? no_dev() : DDN_OK;
}

// reference names (include/dsd-neo/protocol/p25/p25p1_soft.h): one codeword per call
static int
golay_soft_one(int len, char* data, const char* parity, const int* reliab, int* fixed) {
    if (!fixed) {
        return 1;
    }
    *fixed = 0;
    if (!data || !parity || !reliab) {
        return 1;
    }
    uint8_t st = 1;
    int32_t fx = 0;
    if (ddn_fec_golay24_soft_host(len, (uint8_t*)data, (const uint8_t*)parity, (const int32_t*)reliab, 1, &st, &fx)
        != DDN_OK) {
        return 1;
    }
    *fixed = fx;
    return st;
}

extern "C" int
check_and_fix_golay_24_6_soft(char* data, const char* parity, const int* reliab, int* fixed) {
    return golay_soft_one(6, data, parity, reliab, fixed);
}

extern "C" int
check_and_fix_golay_24_12_soft(char* data, const char* parity, const int* reliab, int* fixed) {
    return golay_soft_one(12, data, parity, reliab, fixed);
}

extern "C" int
hamming_10_6_3_soft(const char* bits, const int* reliab, char* out_bits) {
    if (!bits || !reliab || !out_bits) {
        return 2;
    }
    uint8_t st = 2;
    if (ddn_fec_hamming_10_6_3_soft_host((const uint8_t*)bits, (const int32_t*)reliab, 1, (uint8_t*)out_bits, &st)
        != DDN_OK) {
        memcpy(out_bits, bits, 10);
        return 2;
    }
    return st;
}

// ---- RS with reliability-ranked erasures -------------------------------------------------------------------------------
extern "C" int
ddn_fec_p25_rs_soft_batch(int code, uint8_t* d_data_bits, const uint8_t* d_parity_bits, const uint8_t* d_data_reliab,
                          const uint8_t* d_parity_reliab, size_t n, uint8_t* d_status, void* hip_stream) {
    int n_par, n_data, t;
    int rc = rs_code_params(code, &n_par, &n_data, &t);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!d_data_bits || !d_parity_bits || !d_data_reliab || !d_parity_reliab || !d_status) {
        ddn_set_error("ddn_fec_p25_rs_soft_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_rs63_soft(d_data_bits, d_parity_bits, d_data_reliab, d_parity_reliab, n_par, n_data, t,
                              /*P25P1_SOFT_ERASURE_THRESHOLD*/ 64, (int)n, d_status, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_rs_soft_host(int code, uint8_t* data_bits, const uint8_t* parity_bits, const uint8_t* data_reliab,
                         const uint8_t* parity_reliab, size_t n, uint8_t* status) {
    int n_par, n_data, t;
    int rc = rs_code_params(code, &n_par, &n_data, &t);
    if (rc != DDN_OK) {
        return rc;
    }
    if (!data_bits || !parity_bits || !data_reliab || !parity_reliab || !status) {
        return DDN_EINVAL;
    }
    Dev a(n * (size_t)n_data * 6), p(n * (size_t)n_par * 6), dr(n * (size_t)n_data), pr(n * (size_t)n_par), s(n);
    if (!a.p || !p.p || !dr.p || !pr.p || !s.p || a.up(data_bits) || p.up(parity_bits) || dr.up(data_reliab)
        || pr.up(parity_reliab)) {
        return no_dev();
    }
    rc = ddn_fec_p25_rs_soft_batch(code, (uint8_t*)a.p, (const uint8_t*)p.p, (const uint8_t*)dr.p, (const uint8_t*)pr.p, n,
                                   (uint8_t*)s.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (a.down(data_bits) || s.down(status)) ? no_dev() : DDN_OK;
}

static int
rs_soft_one(int code, char* data, const char* parity, const uint8_t* data_reliab, const uint8_t* parity_reliab) {
    uint8_t st = 1;
    if (!data || !parity || !data_reliab || !parity_reliab
        || ddn_fec_p25_rs_soft_host(code, (uint8_t*)data, (const uint8_t*)parity, data_reliab, parity_reliab, 1, &st)
               != DDN_OK) {
        return 1;
    }
    return st;
}

// reference names (include/dsd-neo/protocol/p25/p25p1_soft.h:90-110)
extern "C" int
p25p1_rs_24_12_13_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                   const uint8_t* parity_reliab) {
    return rs_soft_one(DDN_RS_24_12_13, data, parity, data_reliab, parity_reliab);
}

extern "C" int
p25p1_rs_24_16_9_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                  const uint8_t* parity_reliab) {
    return rs_soft_one(DDN_RS_24_16_9, data, parity, data_reliab, parity_reliab);
}

extern "C" int
p25p1_rs_36_20_17_soft_reliability(char* data, const char* parity, const uint8_t* data_reliab,
                                   const uint8_t* parity_reliab) {
    return rs_soft_one(DDN_RS_36_20_17, data, parity, data_reliab, parity_reliab);
}


// ---- IMBE de-interleave (process_IMBE) ------------------------------------------------------------------------------
extern "C" int
ddn_p25p1_imbe_deinterleave_batch(const uint8_t* d_records10, size_t n_records, const int64_t* d_first_record,
                                  const int32_t* d_status_count, size_t n_frames, uint8_t* d_imbe_fr,
                                  uint8_t* d_imbe_soft, uint8_t* d_flags, int32_t* d_status_count_out,
                                  void* hip_stream) {
    if (!d_records10 || !d_first_record || !d_status_count || !d_imbe_fr || !d_imbe_soft || !d_flags
        || !d_status_count_out) {
        ddn_set_error("ddn_p25p1_imbe_deinterleave_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_imbe_deinterleave(d_records10, (long)n_records, d_first_record, d_status_count, (int)n_frames,
                                      d_imbe_fr, d_imbe_soft, d_flags, d_status_count_out, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_p25p1_imbe_deinterleave_host(const uint8_t* records10, size_t n_records, const int64_t* first_record,
                                 const int32_t* status_count, size_t n_frames, uint8_t* imbe_fr, uint8_t* imbe_soft,
                                 uint8_t* flags, int32_t* status_count_out) {
    if (!records10 || !first_record || !status_count || !imbe_fr || !imbe_soft || !flags || !status_count_out) {
        return DDN_EINVAL;
    }
    Dev r(n_records * 10), f(n_frames * 8), c(n_frames * 4), o(n_frames * 184), so(n_frames * 368), fl(n_frames),
        co(n_frames * 4);
    if (!r.p || !f.p || !c.p || !o.p || !so.p || !fl.p || !co.p || r.up(records10) || f.up(first_record)
        || c.up(status_count)) {
        return no_dev();
    }
    int rc = ddn_p25p1_imbe_deinterleave_batch((const uint8_t*)r.p, n_records, (const int64_t*)f.p, (const int32_t*)c.p,
                                               n_frames, (uint8_t*)o.p, (uint8_t*)so.p, (uint8_t*)fl.p, (int32_t*)co.p,
                                               nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (o.down(imbe_fr) || so.down(imbe_soft) || fl.down(flags) || co.down(status_count_out)) ? no_dev() : DDN_OK;
}


// ---- P25p1 low speed data (16,8) -----------------------------------------------------------------------------------------
extern "C" int
ddn_fec_p25_lsd_batch(uint8_t* d_bits16, const int16_t* d_llr16, size_t n, uint8_t* d_ok, void* hip_stream) {
    if (!d_bits16 || !d_ok) {
        ddn_set_error("ddn_fec_p25_lsd_batch: null argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25_lsd(d_bits16, d_llr16, (int)n, d_ok, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_lsd_host(uint8_t* bits16, const int16_t* llr16, size_t n, uint8_t* ok) {
    if (!bits16 || !ok) {
        return DDN_EINVAL;
    }
    Dev b(n * 16), l(n * 32), o(n);
    if (!b.p || !l.p || !o.p || b.up(bits16) || (llr16 && l.up(llr16))) {
        return no_dev();
    }
    int rc = ddn_fec_p25_lsd_batch((uint8_t*)b.p, llr16 ? (const int16_t*)l.p : nullptr, n, (uint8_t*)o.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return (b.down(bits16) || o.down(ok)) ? no_dev() : DDN_OK;
}

// reference names (include/dsd-neo/protocol/p25/p25_lsd.h): 1 = valid or corrected, 0 = uncorrectable (also on failure)
extern "C" int
p25_lsd_fec_16x8(uint8_t* bits16) {
    uint8_t ok = 0;
    if (!bits16 || ddn_fec_p25_lsd_host(bits16, nullptr, 1, &ok) != DDN_OK) {
        return 0;
    }
    return ok;
}

extern "C" int
p25_lsd_fec_16x8_soft(uint8_t* bits16, const int16_t llr16[16]) {
    uint8_t ok = 0;
    if (!bits16 || ddn_fec_p25_lsd_host(bits16, llr16, 1, &ok) != DDN_OK) {
        return 0;
    }
    return ok;
}


// ---- CRC-CCITT16 of decoded trunking blocks ---------------------------------------------------------------------------------
extern "C" int
ddn_fec_p25_crc16_batch(const uint8_t* d_bytes, int item_bytes, size_t n, uint8_t* d_ok, void* hip_stream) {
    if (!d_bytes || !d_ok || item_bytes < 3) {
        ddn_set_error("ddn_fec_p25_crc16_batch: bad argument");
        return DDN_EINVAL;
    }
    HIP_TRY(ddn_dev_p25_crc16(d_bytes, item_bytes, (int)n, d_ok, (hipStream_t)hip_stream));
    return DDN_OK;
}

extern "C" int
ddn_fec_p25_crc16_host(const uint8_t* bytes, int item_bytes, size_t n, uint8_t* ok) {
    if (!bytes || !ok || item_bytes < 3) {
        return DDN_EINVAL;
    }
    Dev b(n * (size_t)item_bytes), o(n);
    if (!b.p || !o.p || b.up(bytes)) {
        return no_dev();
    }
    int rc = ddn_fec_p25_crc16_batch((const uint8_t*)b.p, item_bytes, n, (uint8_t*)o.p, nullptr);
    if (rc != DDN_OK) {
        return rc;
    }
    return o.down(ok) ? no_dev() : DDN_OK;
}

// reference name (include/dsd-neo/protocol/p25/p25_crc.h): payload = len + 16 ints holding one bit each; 0 good, 65535 bad
// (the reference's helper returns (uint16_t)-1 through an int)
extern "C" int
crc16_lb_bridge(const int* payload, int len) {
    if (!payload || len < 8 || (len & 7) || len + 16 > 190 * 1) {
        return 65535;
    }
    uint8_t bytes[32];
    const int nb = (len + 16) / 8;
    if (nb > (int)sizeof(bytes)) {
        return 65535;
    }
    for (int k = 0; k < nb; k++) {
        unsigned v = 0;
        for (int j = 0; j < 8; j++) {
            v = (v << 1) | (unsigned)(payload[8 * k + j] & 1);
        }
        bytes[k] = (uint8_t)v;
    }
    uint8_t ok = 0;
    if (ddn_fec_p25_crc16_host(bytes, nb, 1, &ok) != DDN_OK) {
        return 65535;
    }
    return ok ? 0 : 65535;
}
