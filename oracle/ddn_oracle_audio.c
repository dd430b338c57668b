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
