/* oracle/ddn_oracle_mbe_math.h - TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Float-only elementary functions and the counter-based random generator of the vocoder restatement.  mbelib calls
 * libm cosf / expf / powf and libc rand(); neither a libm result nor rand()'s global sequence can be reproduced on a
 * GPU, and the mbelib-neo source is absent anyway (parity unpinned), so the restatement FIXES these: Cody-Waite
 * reduction + Cephes-style minimax polynomials (explicit fmaf where fused, nothing else contracted) and a hash of
 * (talk path, frame number, harmonic, use) instead of rand().  The product's device code evaluates the same operations
 * in the same order, so PCM is compared bit for bit. */
#ifndef DDN_ORACLE_MBE_MATH_H
#define DDN_ORACLE_MBE_MATH_H

#include <math.h>
#include <stdint.h>

#define OM_PI_F        3.14159265358979323846f
#define OM_TWO_PI_F    6.28318530717958647692f
#define OM_UVTHRESHOLD 2.12057504117311f   /* 2700 * pi / 4000 */
#define OM_UVSINE      3.69452831983566f   /* 1.3591409 * e */
#define OM_UVRAND      2.0f
#define OM_UVSTEP      0.333333333333333f  /* 1 / uvquality, uvquality = 3 */
#define OM_UVOFFSET    0.333333333333333f  /* uvstep * (uvquality - 1) / 2 */
#define OM_QFACTOR     0.366204096222703f  /* log(3) / 3 */

static inline void
om_sincos_core(float x, float* s, float* c, int* q) {
    const float k = rintf(x * 0.636619772367581343f); /* 2 / pi */
    float r = fmaf(-k, 1.5703125f, x);                /* pi/2 = 1.5703125 + 4.83826794896619e-4 */
    r = fmaf(-k, 4.83826794896619e-4f, r);
    const float z = r * r;
    float cs = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cs = fmaf(cs, z, 4.166664568298827e-2f);
    cs = fmaf(cs * z, z, fmaf(-0.5f, z, 1.0f));
    float sn = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sn = fmaf(sn, z, -1.6666654611e-1f);
    sn = fmaf(sn * z, r, r);
    *s = sn;
    *c = cs;
    *q = (int)k & 3;
}

static inline float
om_cosf(float x) {
    float s, c;
    int q;
    om_sincos_core(x, &s, &c, &q);
    return q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

/* e^y for |y| < 87 (clamped) */
static inline float
om_expf(float y) {
    if (y > 87.0f) {
        y = 87.0f;
    }
    if (y < -87.0f) {
        y = -87.0f;
    }
    const float n = rintf(y * 1.44269504088896341f);
    float r = fmaf(-n, 0.693359375f, y);
    r = fmaf(-n, -2.12194440e-4f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r * r, r) + 1.0f;
    union {
        uint32_t u;
        float f;
    } sc;
    sc.u = (uint32_t)((int)n + 127) << 23;
    return p * sc.f;
}

static inline uint32_t
om_mix(uint32_t h, uint32_t v) {
    h ^= v;
    h ^= h >> 16;
    h *= 0x7feb352dU;
    h ^= h >> 15;
    h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}

static inline float
om_u01(uint32_t h) {
    return (float)(h >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
}

/* mbe_rand_phase(): uniform in [-pi, pi) */
static inline float
om_rand_phase(uint32_t h) {
    return om_u01(h) * OM_TWO_PI_F - OM_PI_F;
}

/* synthesis window Ws[k], k = 0..320 (time k - 160): 0 beyond +-105, 1 within +-55, linear in between */
static inline float
om_ws(int k) {
    int t = k - 160;
    if (t < 0) {
        t = -t;
    }
    if (t >= 105) {
        return 0.0f;
    }
    if (t <= 55) {
        return 1.0f;
    }
    return (float)(105 - t) / 50.0f;
}

#endif
