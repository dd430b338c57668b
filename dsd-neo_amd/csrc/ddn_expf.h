// ddn_expf.h - binary32 exp as the host's libm computes it, for the M17 soft costs (soft_symbol_to_viterbi_cost(),
// src/core/frames/dsd_dibit.c:1189-1242 calls expf on a clamped log-likelihood ratio, |x| < 16).
//
// glibc >= 2.27 evaluates expf in binary64 (the "exp2f_data" algorithm of the ARM optimized routines, sysdeps/ieee754/flt-32/e_expf.c):
// z = x * 32 / ln 2, k = round(z) by the 1.5 * 2^52 shift, r = z - k, 2^(k / 32) from a 32-entry table with the exponent added in the
// integer domain, a cubic in r, one product, one rounding to binary32.  The same sequence of IEEE binary64 operations on the GPU gives
// the same bits (no contraction: the library is built with -ffp-contract=off); tests/test_expf.py compares this header compiled for
// the host with the container's expf over every binary32 |x| <= 17.  (A libm whose expf is the FMA build of the same algorithm can differ
// in the last bit of the binary64 result, which reaches the binary32 result about once in 2^29 arguments.)
#ifndef DDN_EXPF_H
#define DDN_EXPF_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DDN_EXPF_FN __host__ __device__ inline
#else
#define DDN_EXPF_FN static inline
#endif

DDN_EXPF_FN float
ddn_expf(float x) { // valid for |x| < 88 (no overflow / underflow handling: the caller clamps to |x| < 16)
    const uint64_t tab[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
        0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
        0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    const double shift = 0x1.8p+52, inv_ln2_n = 0x1.71547652b82fep+0 * 32.0;
    const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double xd = (double)x;
    double z = inv_ln2_n * xd;
    double kd = z + shift;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= shift;
    const double r = z - kd;
    uint64_t t = tab[ki & 31u];
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}
#endif
