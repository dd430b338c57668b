// ddn_atan2f.h — binary32 atan2 for the discriminator's large-angle branch.
//
// The reference calls libm atan2f (src/dsp/fsk_modem.c:34).  On the platforms dsd-neo is built for (glibc
// 2.35 in this image) that is the classic float-only argument-reduction + odd/even polynomial algorithm
// (published as FreeBSD msun / fdlibm e_atan2f.c + s_atanf.c), i.e. a fixed sequence of IEEE binary32
// operations.  Evaluating that same sequence on the GPU (no contraction, correctly-rounded division) gives
// the bit pattern the host libm returns — validated against glibc 2.35 on 6e7 random argument pairs
// (uniform in the unit square, scaled, and raw bit patterns) with zero mismatches — so the discriminator
// stays bit-identical to the CPU path even where the small-angle polynomial is not taken.
#ifndef DDN_ATAN2F_H
#define DDN_ATAN2F_H

#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float
ddn_atanf_core(float x) {
    const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const int32_t hx = __float_as_int(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 (or NaN) */
        if (ix > 0x7f800000) {
            return x + x;
        }
        return (hx > 0) ? (hi[3] + lo[3]) : (-hi[3] - lo[3]);
    }
    if (ix < 0x3ee00000) { /* |x| < 7/16 */
        if (ix < 0x31000000) {
            return x; /* |x| < 2^-29 */
        }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) {
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) {
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 =
        z * (3.3333334327e-01f
             + w * (1.4285714924e-01f
                    + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 =
        w * (-2.0000000298e-01f
             + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    if (id < 0) {
        return x - x * (s1 + s2);
    }
    const float r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
    return (hx < 0) ? -r : r;
}

__device__ __noinline__ float
ddn_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = __float_as_int(x), hy = __float_as_int(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) {
        return x + y;
    }
    if (hx == 0x3f800000) {
        return ddn_atanf_core(y);
    }
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        return (m < 2) ? y : ((m == 2) ? (pi + tiny) : (-pi - tiny));
    }
    if (ix == 0) {
        return (hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
    }
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            return (m == 0) ? (pi_o_4 + tiny) : (m == 1) ? (-pi_o_4 - tiny) : (m == 2) ? (3.0f * pi_o_4 + tiny)
                                                                                           : (-3.0f * pi_o_4 - tiny);
        }
        return (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? (pi + tiny) : (-pi - tiny);
    }
    if (iy == 0x7f800000) {
        return (hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
    }
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) {
        z = pi_o_2 + 0.5f * pi_lo;
    } else if (hx < 0 && k < -60) {
        z = 0.0f;
    } else {
        z = ddn_atanf_core(fabsf(y / x));
    }
    switch (m) {
        case 0: return z;
        case 1: return __int_as_float(__float_as_int(z) ^ (int32_t)0x80000000);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// The same two functions as straight-line code for the arguments the discriminator sees on a noisy channel (every sample of a channel
// without a carrier takes the large-angle branch: the fsk4 captures of the mixed bench are ~90 % noise, and the front end took 3.9 ms
// on them against 2.1 on clean C4FM).  ddn_atanf_core's four argument reductions each end in a division; a wavefront whose lanes fall
// into different ranges ran all four one after the other.  Here every lane forms its range's numerator and denominator (exact or
// single-rounded expressions, the same operations as above), and ONE correctly rounded division serves all of them; |x| < 7/16 divides
// by 1 (exact).  The result selects follow the same way.  Everything rare - NaN, infinities, zeros, x == 1, exponent gaps beyond 2^60,
// |q| >= 2^25 or < 2^-29 - leaves through one branch to the function above, so the common path carries no other branch.
#ifndef DDN_ATAN2_FAST_INLINE
#define DDN_ATAN2_FAST_INLINE __forceinline__
#endif
__device__ DDN_ATAN2_FAST_INLINE float
ddn_atan2f_fast(float y, float x) {
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = __float_as_int(x), hy = __float_as_int(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int k = (iy - ix) >> 23;
    // (unsigned tricks: ix - 1 >= 0x7f7fffff catches 0, infinity and NaN in one compare)
    const bool rare_arg = ((uint32_t)(ix - 1) >= 0x7f7fffffu) | ((uint32_t)(iy - 1) >= 0x7f7fffffu) | (hx == 0x3f800000) | (k > 60) | (k < -60);
    const float q = fabsf(y / x);
    const int32_t iq = __float_as_int(q);
    const bool rare = rare_arg | (iq >= 0x4c000000) | (iq < 0x31000000);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
        return ddn_atan2f(y, x);
    }
    // ddn_atanf_core(q), q > 0: range id -1 (q < 7/16), 0, 1, 2, 3
    const bool r0 = iq < 0x3ee00000, r1 = iq < 0x3f300000, r2 = iq < 0x3f980000, r3 = iq < 0x401c0000;
    const float num = r0 ? q : (r1 ? (2.0f * q - 1.0f) : (r2 ? (q - 1.0f) : (r3 ? (q - 1.5f) : -1.0f)));
    const float den = r0 ? 1.0f : (r1 ? (2.0f + q) : (r2 ? (q + 1.0f) : (r3 ? (1.0f + 1.5f * q) : q)));
    const float hi = r1 ? 4.6364760399e-01f : (r2 ? 7.8539812565e-01f : (r3 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = r1 ? 5.0121582440e-09f : (r2 ? 3.7748947079e-08f : (r3 ? 3.4473217170e-08f : 7.5497894159e-08f));
    const float t = num / den;
    const float z2 = t * t;
    const float w = z2 * z2;
    const float s1 =
        z2 * (3.3333334327e-01f
              + w * (1.4285714924e-01f
                     + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 =
        w * (-2.0000000298e-01f
             + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    const float ts = t * (s1 + s2);
    const float z = r0 ? (t - ts) : (hi - ((ts - lo) - t));
    // quadrant: m = sign(y) | 2 sign(x)
    const float zq = (hx < 0) ? (pi - (z - pi_lo)) : z;                 // m = 2: pi - (z - pi_lo); m = 3: (z - pi_lo) - pi = -(that)
    return __int_as_float(__float_as_int(zq) ^ (hy & (int32_t)0x80000000));
}
#endif
