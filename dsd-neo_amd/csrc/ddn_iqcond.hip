// ddn_iqcond.hip — the FSK path with the optional IQ conditioning stages switched on.
//
// reference (src/dsp/demod_pipeline.cpp, per full_demod() block, after the channel LPF):
//   mean_power over the first <= 512 floats + squelch gate   :926-945,1003-1020
//   iq_dc_block: per-sample leaky integrator, alpha 2^-k     :948-978
//   full_demod_apply_iq_balance: block sums in binary64, EMA of the image coefficient, conditional correction :1131-1171
//   dsd_fsk_modem_discriminator_process                       src/dsp/fsk_modem.c:89-164
//
// These stages are off by default in the reference and the fused front-end kernel does not carry them: with either
// switch on, the batch runs channel LPF (k_channel_lpf_c2c, complex result in HBM) -> this kernel.  Everything here
// is a per-sample recurrence or a sequential binary64 sum, so one lane = one channel (as in the symbol-rate kernels):
// wave 1 stages [64 channels][64 samples] complex tiles with coalesced row loads and writes finished discriminator
// tiles back row by row; wave 0 runs the recurrences.  The balance stage needs whole-block sums before the first
// sample of the block can be corrected, so a block is walked twice: pass 1 accumulates the squelch power and the
// balance sums on a throw-away copy of the DC blocker state, pass 2 re-runs the DC blocker for real (same operations,
// same bits), applies the correction and discriminates.  Nothing is written back to the complex buffer.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_atan2f.h"
#include "ddn_device.h"

namespace {
typedef ddn_f2 f2;
constexpr int TS = 64;

struct Step {
    long blk;
    int pass, tile;
};

__global__ __launch_bounds__(128) void
k_iq_cond_disc(const f2* __restrict__ in, long n, size_t stride, int block_len, int n_channels, DdnIqCondConfig cfg,
               DdnFskState* __restrict__ fsk, DdnIqCondState* __restrict__ cond, float* __restrict__ out,
               size_t out_stride) {
    extern __shared__ unsigned char smem_raw[];
    f2(*tile)[64][TS + 1] = reinterpret_cast<f2(*)[64][TS + 1]>(smem_raw);                                  // [2]
    float(*otile)[64][TS + 1] = reinterpret_cast<float(*)[64][TS + 1]>(smem_raw + sizeof(f2) * 2 * 64 * (TS + 1)); // [2]
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;
    const int ch0 = blockIdx.x * 64;
    const int ch = ch0 + lane;
    const bool live = !loader && ch < n_channels;
    const bool two_pass = cfg.squelch_on || cfg.bal_enable;
    const long n_blocks = (n + block_len - 1) / block_len;

    auto blk_len = [&](long b) { return (int)((n - b * block_len) < block_len ? (n - b * block_len) : block_len); };
    auto advance = [&](Step& s) { // next (block, pass, tile); returns false past the end
        const int tiles = (blk_len(s.blk) + TS - 1) / TS;
        if (++s.tile < tiles) {
            return true;
        }
        s.tile = 0;
        if (two_pass && s.pass == 0) {
            s.pass = 1;
            return true;
        }
        s.pass = two_pass ? 0 : 1;
        return ++s.blk < n_blocks;
    };
    auto stage = [&](const Step& s, int buf) {
        const long t0 = s.blk * block_len + (long)s.tile * TS;
        const int bl = blk_len(s.blk);
        const int tn = (bl - s.tile * TS) < TS ? (bl - s.tile * TS) : TS;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            f2 r[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int cc = 16 * h + c;
                const f2 z = {0.0f, 0.0f};
                r[c] = (ch0 + cc < n_channels && lane < tn) ? in[(size_t)(ch0 + cc) * stride + t0 + lane] : z;
            }
#pragma unroll
            for (int c = 0; c < 16; c++) {
                tile[buf][16 * h + c][lane] = r[c];
            }
        }
    };
    auto flush = [&](const Step& s, int buf) { // discriminator tile of a finished pass-2 step -> HBM, row by row
        if (s.pass != 1) {
            return;
        }
        const long t0 = s.blk * block_len + (long)s.tile * TS;
        const int bl = blk_len(s.blk);
        const int tn = (bl - s.tile * TS) < TS ? (bl - s.tile * TS) : TS;
        for (int c = 0; c < 64; c++) {
            if (ch0 + c < n_channels && lane < tn) {
                out[(size_t)(ch0 + c) * out_stride + t0 + lane] = otile[buf][c][lane];
            }
        }
    };

    DdnFskState fs = {0.0f, 0.0f, 0, 0.0f, 0.0f};
    DdnIqCondState cs = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        fs = fsk[ch];
        cs = cond[ch];
    }
    int k = cfg.dc_shift;
    k = k < 6 ? 6 : (k > 15 ? 15 : k);
    const float alpha = 1.0f / (float)(1 << k);
    const float ema = cfg.bal_ema_a > 0.0f ? cfg.bal_ema_a : 0.2f;
    const float thr = cfg.bal_thr > 0.0f ? cfg.bal_thr : 0.02f;

    Step cur = {0, two_pass ? 0 : 1, 0};
    Step prev = cur;
    bool have_prev_step = false;
    if (loader && n > 0) {
        stage(cur, 0);
    }
    __syncthreads();
    // per-block scratch of the recurrence wave
    double pw_t = 0.0, pw_p = 0.0, s2r = 0.0, s2i = 0.0, p2 = 0.0;
    float tdc_r = 0.0f, tdc_i = 0.0f;
    bool squelched = false, apply = false;
    int buf = 0;
    bool more = n > 0;
    while (more) {
        Step nxt = cur;
        const bool has_next = advance(nxt);
        if (loader) {
            if (has_next) {
                stage(nxt, buf ^ 1);
            }
            if (have_prev_step) {
                flush(prev, buf ^ 1);
            }
        } else if (live) {
            const int bl = blk_len(cur.blk);
            const int i0 = cur.tile * TS;
            const int tn = (bl - i0) < TS ? (bl - i0) : TS;
            if (cur.pass == 0) {
                if (cur.tile == 0) {
                    pw_t = pw_p = s2r = s2i = p2 = 0.0;
                    tdc_r = cs.dc_r;
                    tdc_i = cs.dc_i;
                }
                const int pw_pairs = (2 * bl > 512 ? 512 : 2 * bl) >> 1; // complex samples under the power window
                for (int j = 0; j < tn; j++) {
                    const f2 x = tile[buf][lane][j];
                    float I = x.x, Q = x.y;
                    if (cfg.squelch_on && i0 + j < pw_pairs) {
                        const double a = (double)I, b = (double)Q;
                        pw_t += a;
                        pw_p += a * a;
                        pw_t += b;
                        pw_p += b * b;
                    }
                    if (cfg.bal_enable) {
                        if (cfg.dc_enable) {
                            tdc_r += (I - tdc_r) * alpha;
                            tdc_i += (Q - tdc_i) * alpha;
                            I = I - tdc_r;
                            Q = Q - tdc_i;
                        }
                        const double a = (double)I, b = (double)Q;
                        s2r += a * a - b * b;
                        s2i += 2.0 * a * b;
                        p2 += a * a + b * b;
                    }
                }
                if (i0 + tn >= bl) { // block totals known: squelch decision and the balance coefficient
                    squelched = false;
                    if (cfg.squelch_on) {
                        const int len = 2 * pw_pairs;
                        const double dc_corr = len > 0 ? (pw_t * pw_t) / (double)len : 0.0;
                        double energy = pw_p - dc_corr;
                        energy = energy < 0.0 ? 0.0 : energy;
                        const float pwr = (float)(energy / (double)(len > 0 ? len : 1));
                        squelched = pwr < cfg.squelch_level;
                    }
                    apply = false;
                    if (cfg.bal_enable && !squelched) {
                        const double pp = p2 <= 1e-9 ? 1e-9 : p2;
                        const float ar = (float)(s2r / pp), ai = (float)(s2i / pp);
                        cs.er += ema * (ar - cs.er);
                        cs.ei += ema * (ai - cs.ei);
                        apply = !((cs.er * cs.er + cs.ei * cs.ei) < (thr * thr));
                    }
                }
            } else {
                if (cur.tile == 0 && !two_pass) {
                    squelched = false;
                    apply = false;
                }
                if (squelched) {
                    if (cur.tile == 0) {
                        fs.prev_i = 0.0f;
                        fs.prev_q = 0.0f;
                        fs.have_prev = 0;
                        fs.dc_est = 0.0f;
                        fs.peak_est = 0.0f;
                    }
                    for (int j = 0; j < tn; j++) {
                        otile[buf][lane][j] = 0.0f;
                    }
                } else {
                    for (int j = 0; j < tn; j++) {
                        const f2 x = tile[buf][lane][j];
                        float I = x.x, Q = x.y;
                        if (cfg.dc_enable) {
                            cs.dc_r += (I - cs.dc_r) * alpha;
                            cs.dc_i += (Q - cs.dc_i) * alpha;
                            I = I - cs.dc_r;
                            Q = Q - cs.dc_i;
                        }
                        if (apply) {
                            const float tI = cs.er * I + cs.ei * Q;
                            const float tQ = -cs.er * Q + cs.ei * I;
                            I = I - tI;
                            Q = Q - tQ;
                        }
                        float y = 0.0f;
                        if (!fs.have_prev) {
                            fs.have_prev = 1;
                        } else {
                            const float re = I * fs.prev_i + Q * fs.prev_q;
                            const float im = Q * fs.prev_i - I * fs.prev_q;
                            float fr;
                            if (re > 1.0e-7f && fabsf(im) <= (0.35f * re)) {
                                const float q = im / re;
                                const float q2 = q * q;
                                fr = q * (1.0f + q2 * (-0.3333333333333333f + q2 * 0.2f));
                            } else {
                                fr = ddn_atan2f(im, re);
                            }
                            fs.dc_est += 0.00025f * (fr - fs.dc_est);
                            const float c = fr - fs.dc_est;
                            const float mag = fabsf(c);
                            if (mag > 1.0e-7f) {
                                if (fs.peak_est <= 1.0e-7f) {
                                    fs.peak_est = mag;
                                } else if (mag > fs.peak_est) {
                                    fs.peak_est += 0.125f * (mag - fs.peak_est);
                                } else {
                                    fs.peak_est += 0.00005f * (mag - fs.peak_est);
                                }
                            }
                            float pk = fs.peak_est;
                            if (pk <= 1.0e-7f) {
                                pk = 1.0f;
                            }
                            y = c * (30000.0f / pk);
                            y = y > 32767.0f ? 32767.0f : (y < -32768.0f ? -32768.0f : y);
                        }
                        fs.prev_i = I;
                        fs.prev_q = Q;
                        otile[buf][lane][j] = y;
                    }
                }
            }
        }
        __syncthreads();
        prev = cur;
        have_prev_step = true;
        cur = nxt;
        more = has_next;
        buf ^= 1;
    }
    if (loader && have_prev_step) {
        flush(prev, buf ^ 1);
    }
    if (live) {
        fsk[ch] = fs;
        cond[ch] = cs;
    }
}
} // namespace

extern "C" hipError_t
ddn_dev_iq_cond_disc(const void* in, long n, size_t stride, int block_len, int n_channels, const DdnIqCondConfig* cfg,
                     DdnFskState* fsk, DdnIqCondState* cond, float* out, size_t out_stride, hipStream_t st) {
    if (n_channels <= 0 || n <= 0) {
        return hipSuccess;
    }
    const size_t shm = (sizeof(f2) + sizeof(float)) * 2 * 64 * (TS + 1);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_iq_cond_disc),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(k_iq_cond_disc, dim3((unsigned)((n_channels + 63) / 64)), dim3(128), shm, st, (const f2*)in, n,
                       stride, block_len, n_channels, *cfg, fsk, cond, out, out_stride);
    return hipGetLastError();
}
