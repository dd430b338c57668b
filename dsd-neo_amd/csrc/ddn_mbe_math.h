// ddn_mbe_math.h - float-only elementary functions and the counter-based random generator of the vocoder kernels.
//
// mbelib calls libm cosf / expf / powf and libc rand(); a libm result is implementation-defined in its last bits and
// rand() is one process-global sequence, so neither can be the contract of a batched device implementation.  This
// library fixes them instead: Cody-Waite reduction + Cephes-style minimax polynomials with explicit fused steps
// (the translation unit is compiled -ffp-contract=off), x^(1/4) as two IEEE square roots, and a hash of (talk path,
// frame number, harmonic, use) where mbelib draws from rand().  Every operation is IEEE binary32 with a defined
// order, so results do not depend on the launch geometry.
#ifndef DDN_MBE_MATH_H
#define DDN_MBE_MATH_H

#include <stdint.h>

#define MBE_PI_F        3.14159265358979323846f
#define MBE_TWO_PI_F    6.28318530717958647692f
#define MBE_UVTHRESHOLD 2.12057504117311f  /* 2700 * pi / 4000 (mbelib.c mbe_synthesizeSpeechf) */
#define MBE_UVSINE      3.69452831983566f  /* 1.3591409 * e */
#define MBE_UVRAND      2.0f
#define MBE_UVSTEP      0.333333333333333f /* 1 / uvquality, uvquality = 3 (dsd-neo's default) */
#define MBE_UVOFFSET    0.333333333333333f /* uvstep * (uvquality - 1) / 2 */
#define MBE_QFACTOR     0.366204096222703f /* log(uvquality) / uvquality */

#define MBE_HD __host__ __device__ __forceinline__

MBE_HD float
mbe_cosf(float x) {
    const float k = __builtin_rintf(x * 0.636619772367581343f); // 2 / pi
    float r = __builtin_fmaf(-k, 1.5703125f, x);                // pi/2 = 1.5703125 + 4.83826794896619e-4
    r = __builtin_fmaf(-k, 4.83826794896619e-4f, r);
    const float z = r * r;
    float cs = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cs = __builtin_fmaf(cs, z, 4.166664568298827e-2f);
    cs = __builtin_fmaf(cs * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    float sn = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sn = __builtin_fmaf(sn, z, -1.6666654611e-1f);
    sn = __builtin_fmaf(sn * z, r, r);
    const int q = (int)k & 3;
    return q == 0 ? cs : (q == 1 ? -sn : (q == 2 ? -cs : sn));
}

MBE_HD float
mbe_expf(float y) {
    y = y > 87.0f ? 87.0f : y;
    y = y < -87.0f ? -87.0f : y;
    const float n = __builtin_rintf(y * 1.44269504088896341f);
    float r = __builtin_fmaf(-n, 0.693359375f, y);
    r = __builtin_fmaf(-n, -2.12194440e-4f, r);
    float p = __builtin_fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p, r * r, r) + 1.0f;
    const uint32_t sc = (uint32_t)((int)n + 127) << 23;
    return p * __builtin_bit_cast(float, sc);
}

MBE_HD uint32_t
mbe_mix(uint32_t h, uint32_t v) {
    h ^= v;
    h ^= h >> 16;
    h *= 0x7feb352dU;
    h ^= h >> 15;
    h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}

MBE_HD float
mbe_u01(uint32_t h) {
    return (float)(h >> 8) * 5.9604644775390625e-8f; // 2^-24
}

MBE_HD float
mbe_rand_phase(uint32_t h) {
    return mbe_u01(h) * MBE_TWO_PI_F - MBE_PI_F;
}

// synthesis window Ws[k], k = 0..320 (time k - 160): 0 beyond +-105, 1 within +-55, linear in between (mbelib Ws table)
MBE_HD float
mbe_ws(int k) {
    int t = k - 160;
    t = t < 0 ? -t : t;
    if (t >= 105) {
        return 0.0f;
    }
    if (t <= 55) {
        return 1.0f;
    }
    return (float)(105 - t) / 50.0f;
}

#endif
