/* oracle/ddn_oracle_audio.c - TEST INFRASTRUCTURE ONLY: CPU restatement of the float-path audio post-processing applied to
 * every synthesized voice frame before it leaves dsd-neo.
 *   agf()  src/core/audio/gain.c:23-35 (silence test), :47-58 (effective gain), :60-69 (per-slot divisor), :82-95 (one
 *          20-sample block: divide, clip to +-0.9, average, gain x 0.8), :97-117 (gain walk: +-0.5 per block around a 0.075
 *          average, within 1..46), :119-139 (eight blocks per frame).  Kept on purpose: the block average always reads the
 *          FIRST twenty samples of the frame (samp[i], not samp[idx]) - for block 0 before the gain multiply, for blocks
 *          1..7 block 0's finished values.
 * PARITY: pinned against the compiled gain.c (oracle/_ref, refh_agf_run) in tests/test_oracle_audio.py. */
#include <math.h>
#include <stdint.h>

void
orc_agf(float* samp, int n_frames, float audio_gain, int algid_0x21, float* aout_gain_io) {
    float aout = *aout_gain_io;
    float gain = 1.0f;
    if (algid_0x21) {
        gain = 1.75f;
    }
    if (audio_gain != 0) {
        gain = audio_gain / 25.0f;
    }
    for (int f = 0; f < n_frames; f++) {
        float* s = samp + (long)f * 160;
        int silent = 1;
        for (int i = 0; i < 160; i++) {
            if (s[i] > 1e-12f || s[i] < -1e-12f) {
                silent = 0;
                break;
            }
        }
        if (silent) {
            continue;
        }
        for (int j = 0; j < 8; j++) {
            const float df = 384.0f * (50.0f - aout);
            float aavg = 0.0f;
            for (int i = 0; i < 20; i++) {
                const int idx = j * 20 + i;
                s[idx] = s[idx] / df;
                if (s[idx] > 0.90f) {
                    s[idx] = 0.90f;
                } else if (s[idx] < -0.90f) {
                    s[idx] = -0.90f;
                }
                aavg += fabsf(s[i]);
                s[idx] *= gain * 0.8f;
            }
            aavg /= 20.0f;
            if (aavg < 0.075f && aout < 46.0f) {
                aout += 0.5f;
            }
            if (aavg >= 0.075f && aout > 1.0f) {
                aout -= 0.5f;
            }
        }
    }
    *aout_gain_io = aout;
}

/* ---- the short-integer voice path -------------------------------------------------------------------------------------------
 * One talk path, frames in order.  state32 (floats; see include/ddn_hip.h DDN_S16_*): [0] aout_gain, [1] history index,
 * [2] high-pass v_in[0], [3] v_out[0], [4..28] the 25 block peaks.
 *   stage 1  processAudio()  src/core/audio/dsd_audio.c:427-571: block peak, 25-block peak history, gain factor 30000 / peak
 *            (50 when silent), drop at once / rise by at most 5 % per block, ramp over the 160 samples; clamp, truncate to
 *            int16.  audio_gain != 0: no ramp (the caller's aout_gain is applied as is); audio_gain < 0: no multiply at all.
 *            PARITY UNPINNED for this stage: dsd_audio.c includes <sndfile.h>, which this image does not have, and the
 *            reference's tests hold no vector for it - restatement only.
 *   stage 2  hpf_dL()  src/core/util/dsd_misc.c:345-371,516-522 with init_audio_filters' 960 Hz / 8 kHz coefficient (:436-452).
 *   stage 3  agsm()    src/core/audio/gain.c:143-184, one call per 160-sample frame; the last coefficient lands in *gain_a.
 *            Stages 2 and 3 are pinned against the compiled reference in tests/test_oracle_audio.py. */
float
orc_hpf_d_coef(void) {
    float RC = 0.0;
    RC = 1.0 / (2 * 3.141592653 * 960.0f);
    return RC / ((1.0f / 8000.0f) + RC);
}

void
orc_audio_s16(const float* pcm, int n_frames, float audio_gain, int use_hpf_d, int use_agsm, int16_t* out, float* state32,
              float* gain_a) {
    float aout = state32[0];
    int idx = (int)state32[1];
    float vin0 = state32[2], vout0 = state32[3];
    float* hist = state32 + 4;
    const float coef = orc_hpf_d_coef();
    for (int f = 0; f < n_frames; f++) {
        const float* x = pcm + (long)f * 160;
        int16_t* y = out + (long)f * 160;
        float gd = 0.0f;
        if (audio_gain == 0.0f) {
            float max = 0.0f;
            for (int n = 0; n < 160; n++) {
                const float a = fabsf(x[n]);
                if (a > max) {
                    max = a;
                }
            }
            hist[idx] = max;
            idx++;
            if (idx > 24) {
                idx = 0;
            }
            for (int i = 0; i < 25; i++) {
                if (hist[i] > max) {
                    max = hist[i];
                }
            }
            float gf = max > 0.0f ? 30000.0f / max : 50.0f;
            if (gf < aout) {
                aout = gf;
            } else {
                if (gf > 50.0f) {
                    gf = 50.0f;
                }
                gd = gf - aout;
                if (gd > 0.05f * aout) {
                    gd = 0.05f * aout;
                }
            }
            gd = gd / 160.0f;
        }
        for (int n = 0; n < 160; n++) {
            float v = x[n];
            if (!(audio_gain < 0)) {
                v = (aout + ((float)n * gd)) * v;
            }
            v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v);
            y[n] = (int16_t)v;
        }
        if (!(audio_gain < 0)) {
            aout += 160.0f * gd;
        }
        if (use_hpf_d) {
            for (int n = 0; n < 160; n++) {
                const float vin1 = vin0, vout1 = vout0;
                vin0 = (float)y[n];
                vout0 = coef * (vin0 - vin1 + vout1);
                y[n] = vout0 > 32767.0f ? 32767 : (vout0 < -32768.0f ? -32768 : (int16_t)vout0);
            }
        }
        if (use_agsm) {
            float ma = 0.0f;
            for (int n = 0; n < 160; n++) {
                const float a = fabsf((float)y[n]);
                if (a > ma) {
                    ma = a;
                }
            }
            if (ma < 1e-6f) {
                ma = 1e-6f;
            }
            float c = fabsf(4800.0f / ma);
            if (c > 3.0f) {
                c = 3.0f;
            }
            for (int n = 0; n < 160; n++) {
                float s = (float)y[n] * c;
                s = s > 32767.0f ? 32767.0f : (s < -32768.0f ? -32768.0f : s);
                y[n] = (int16_t)s;
            }
            *gain_a = c;
        }
    }
    state32[0] = aout;
    state32[1] = (float)idx;
    state32[2] = vin0;
    state32[3] = vout0;
}
