// ddn_slicer_dev.h — device helpers shared by the stand-alone slicer kernel (ddn_slicer.hip) and the P25p1 receive
// loop (ddn_rx.hip): two-smallest / two-largest insertion and the 4-level slice + soft decision of one symbol.
// reference: digitize / compute_dibit_soft_metric, src/core/frames/dsd_dibit.c:456-547,609-721,963-1076.
#ifndef DDN_SLICER_DEV_H
#define DDN_SLICER_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddn_sl {
__device__ __forceinline__ int
clamp255(int v) {
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Two smallest / two largest running pair, select form of `if (x < a1) {a2 = a1; a1 = x;} else if (x < a2) a2 = x;`
// (same result for every input incl. NaN, no branches).
__device__ __forceinline__ void
two_min_insert(float x, float& a1, float& a2) {
    const bool c1 = x < a1, c2 = x < a2;
    const float n2 = c1 ? a1 : (c2 ? x : a2);
    a1 = c1 ? x : a1;
    a2 = n2;
}
__device__ __forceinline__ void
two_max_insert(float x, float& b1, float& b2) {
    const bool c1 = x > b1, c2 = x > b2;
    const float n2 = c1 ? b1 : (c2 ? x : b2);
    b1 = c1 ? x : b1;
    b2 = n2;
}

// compute_bit_magnitude() for both bits of one symbol: squared distance to the nearest ideal level whose bit is 0 / 1,
// scaled by 255 / (smallest level spacing)^2.  Written on scalars (no indexed local array -> no scratch).
__device__ __forceinline__ void
bit_magnitudes(float sym, float i0, float i1, float i2, float i3, int& mag0, int& mag1) {
    const float big = 3.4028234663852886e38f;
    const float d0 = (sym - i0) * (sym - i0), d1 = (sym - i1) * (sym - i1);
    const float d2 = (sym - i2) * (sym - i2), d3 = (sym - i3) * (sym - i3);
    float spacing = big;
    auto sp_min = [&](float a, float b) {
        const float sp = fabsf(a - b);
        if (sp > 1e-6f && sp < spacing) {
            spacing = sp;
        }
    };
    sp_min(i0, i1);
    sp_min(i0, i2);
    sp_min(i0, i3);
    sp_min(i1, i2);
    sp_min(i1, i3);
    sp_min(i2, i3);
    if (spacing == big) {
        spacing = 2.0f;
    }
    const float scale = 255.0f / (spacing * spacing);
    auto lt = [&](float best, float d) { return d < best ? d : best; };
    // bit 0 (high bit of the dibit index): levels 0,1 carry 0 and levels 2,3 carry 1
    float z = lt(lt(big, d0), d1), o = lt(lt(big, d2), d3);
    mag0 = clamp255((int)__float2ll_rn(fabsf(z - o) * scale));
    // bit 1 (low bit): levels 0,2 carry 0 and levels 1,3 carry 1
    z = lt(lt(big, d0), d2);
    o = lt(lt(big, d1), d3);
    mag1 = clamp255((int)__float2ll_rn(fabsf(z - o) * scale));
}

struct Thr {
    float center, umid, lmid, max, min;
};

// dmr_compute_reliability() on a C4FM / GFSK stream (dsd_dibit.c:548-568: c4fm_reliability_from_thresholds + the SNR weight
// with every metrics hook unset, i.e. x 204/256): the reliability stored with a payload dibit while hunting.
__device__ __forceinline__ int
rel_from_thresholds(float x, const Thr& s) {
    const float eps = 1e-6f;
    int rel;
    if (x > s.umid) {
        float span = s.max - s.umid;
        span = span < eps ? eps : span;
        rel = (int)__float2ll_rn(((x - s.umid) * 255.0f) / span);
    } else if (x > s.center) {
        const float d1 = x - s.center, d2 = s.umid - x;
        float span = s.umid - s.center;
        span = span < eps ? eps : span;
        rel = (int)__float2ll_rn(((d1 < d2 ? d1 : d2) * 510.0f) / span);
    } else if (x >= s.lmid) {
        const float d1 = s.center - x, d2 = x - s.lmid;
        float span = s.center - s.lmid;
        span = span < eps ? eps : span;
        rel = (int)__float2ll_rn(((d1 < d2 ? d1 : d2) * 510.0f) / span);
    } else {
        float span = s.lmid - s.min;
        span = span < eps ? eps : span;
        rel = (int)__float2ll_rn(((s.lmid - x) * 255.0f) / span);
    }
    return clamp255((clamp255(rel) * 204) >> 8);
}

// One symbol against fixed thresholds: dibit, reliability, llr0, llr1 (the 10-byte record's first six bytes).
__device__ __forceinline__ void
slice_soft(float x, const Thr& s, int negative, int& dibit, int& reliab, int& l0, int& l1) {
    // ---- slice + soft decision ---------------------------------------------------------------------------
    if (x > s.center) {
        dibit = (x > s.umid) ? (negative ? 3 : 1) : (negative ? 2 : 0);
    } else {
        dibit = (x < s.lmid) ? (negative ? 1 : 3) : (negative ? 0 : 2);
    }
    const float plus_one = 0.5f * (s.center + s.umid), minus_one = 0.5f * (s.lmid + s.center);
    const float ideal0 = negative ? minus_one : plus_one, ideal1 = negative ? s.min : s.max;
    const float ideal2 = negative ? plus_one : minus_one, ideal3 = negative ? s.max : s.min;
    int mag0, mag1;
    bit_magnitudes(x, ideal0, ideal1, ideal2, ideal3, mag0, mag1);
    int rel;
    {
        const float eps = 1e-6f;
        if (x > s.umid) {
            float span = s.max - s.umid;
            span = span < eps ? eps : span;
            rel = (int)__float2ll_rn(((x - s.umid) * 255.0f) / span);
        } else if (x > s.center) {
            const float d1 = x - s.center, d2 = s.umid - x;
            float span = s.umid - s.center;
            span = span < eps ? eps : span;
            rel = (int)__float2ll_rn(((d1 < d2 ? d1 : d2) * 510.0f) / span);
        } else if (x >= s.lmid) {
            const float d1 = s.center - x, d2 = x - s.lmid;
            float span = s.center - s.lmid;
            span = span < eps ? eps : span;
            rel = (int)__float2ll_rn(((d1 < d2 ? d1 : d2) * 510.0f) / span);
        } else {
            float span = s.lmid - s.min;
            span = span < eps ? eps : span;
            rel = (int)__float2ll_rn(((s.lmid - x) * 255.0f) / span);
        }
        rel = clamp255((clamp255(rel) * 204) >> 8);
    }
    const int mn = mag0 < mag1 ? mag0 : mag1;
    if (mn > 0 && rel < mn) {
        mag0 = (mag0 * rel) / mn;
        mag1 = (mag1 * rel) / mn;
    }
    l0 = ((dibit >> 1) & 1) ? clamp255(mag0) : -clamp255(mag0);
    l1 = (dibit & 1) ? clamp255(mag1) : -clamp255(mag1);
    const int a0 = l0 < 0 ? -l0 : l0, a1v = l1 < 0 ? -l1 : l1;
    reliab = clamp255(a1v < a0 ? a1v : a0);
}
} // namespace ddn_sl
#endif
