// ddn_nid_dev.h - device functions of the P25 Phase 1 NID decoder shared by the batched kernels (ddn_block.hip) and the
// receive loop's handler wave (ddn_rx.hip): BCH(63,16,11) over GF(2^6) by Massey's iteration + Chien search, DUID / parity
// validation, the Chase candidates.  reference: include/dsd-neo/fec/BCH_63_16.hpp:47-330,
// src/protocol/p25/phase1/p25p1_check_nid.cpp:200-354.
#ifndef DDN_NID_DEV_H
#define DDN_NID_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddn_nid {

struct Gf {
    const uint8_t* ex; // [128]
    const uint8_t* lg; // [64]
    __device__ __forceinline__ int mul(int a, int b) const { return (a && b) ? ex[lg[a] + lg[b]] : 0; }
    __device__ __forceinline__ int div(int a, int b) const { return a ? ex[lg[a] + 63 - lg[b]] : 0; }
};

__device__ inline void
gf_fill(uint8_t* ex, uint8_t* lg) { // one thread
    int x = 1;
    for (int i = 0; i < 63; i++) {
        ex[i] = (uint8_t)x;
        ex[i + 63] = (uint8_t)x;
        lg[x] = (uint8_t)i;
        x <<= 1;
        if (x & 64) {
            x ^= 0x43; // x^6 = x + 1
        }
    }
    ex[126] = ex[0];
    ex[127] = ex[1];
    lg[0] = 0;
}

// Per-lane working arrays live in LDS, element k of this lane at base[k * 64] (no scratch memory, no bank
// conflicts beyond byte-in-word sharing).  Exponent arithmetic mod 63 uses running sums with a conditional
// subtract instead of integer division.
struct Work {
    uint8_t* S; // [23]
    uint8_t* C; // [24]
    uint8_t* B; // [24]
    uint8_t* T; // [24]
    __device__ __forceinline__ uint8_t& s(int i) const { return S[i * 64]; }
    __device__ __forceinline__ uint8_t& c(int i) const { return C[i * 64]; }
    __device__ __forceinline__ uint8_t& b(int i) const { return B[i * 64]; }
    __device__ __forceinline__ uint8_t& t(int i) const { return T[i * 64]; }
};

// w: bit p = received bit at input position p (0..62; data 0..15 MSB-first, parity 16..62).
// Returns 1 on success with *fixed = corrected word and *nerr = flipped bits.
__device__ inline int
bch_63_16_decode(const Gf& gf, const Work& wk, uint64_t w, uint64_t* fixed, int* nerr) {
    // Odd syndromes S1,S3,..,S21 accumulate in registers (static indices); even ones follow from the binary-code
    // identity S_2i = S_i^2.  Running exponent e = i*j mod 63 advances by 2j per odd step.
    int so[11];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        so[k] = 0;
    }
    {
        uint64_t m = w;
        while (m) {
            const int p = __builtin_ctzll(m);
            m &= m - 1;
            const int j = 62 - p; // r-index of this bit; contributes alpha^(i*j) to S_i
            int j2 = 2 * j;
            if (j2 >= 63) {
                j2 -= 63;
            }
            int e = j;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                so[k] ^= gf.ex[e];
                e += j2;
                if (e >= 63) {
                    e -= 63;
                }
            }
        }
    }
    int any = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        wk.s(2 * k + 1) = (uint8_t)so[k];
        any |= so[k];
    }
    for (int i = 2; i <= 22; i += 2) { // S_i = (S_{i/2})^2, ascending so the source is already final
        const int h = wk.s(i / 2);
        wk.s(i) = (uint8_t)(h ? gf.ex[(2 * gf.lg[h]) % 63] : 0);
    }
    *nerr = 0;
    *fixed = w;
    if (!any) {
        return 1;
    }
    for (int i = 0; i < 24; i++) {
        wk.c(i) = 0;
        wk.b(i) = 0;
    }
    wk.c(0) = 1;
    wk.b(0) = 1;
    int L = 0, m = 1, b = 1;
    for (int n = 0; n < 22; n++) {
        int d = wk.s(n + 1);
        for (int i = 1; i <= L; i++) {
            d ^= gf.mul(wk.c(i), wk.s(n + 1 - i));
        }
        if (d == 0) {
            m++;
            continue;
        }
        const int f = gf.div(d, b);
        if (2 * L <= n) {
            for (int i = 0; i < 24; i++) {
                wk.t(i) = wk.c(i);
            }
            for (int i = 0; i + m < 24; i++) {
                wk.c(i + m) ^= (uint8_t)gf.mul(f, wk.b(i));
            }
            L = n + 1 - L;
            for (int i = 0; i < 24; i++) {
                wk.b(i) = wk.t(i);
            }
            b = d;
            m = 1;
        } else {
            for (int i = 0; i + m < 24; i++) {
                wk.c(i + m) ^= (uint8_t)gf.mul(f, wk.b(i));
            }
            m++;
        }
        if (L > 11) {
            return 0;
        }
    }
    // Chien search: root alpha^i <-> error at r-index 63-i <-> input position i-1.  The <= 12 coefficients move
    // to registers (static indices, terms above L predicated off); te[k] is the running exponent
    // log(C[k]) + i*k (mod 63) of term k.
    int te[12];
    bool on[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int ck = (k <= L) ? wk.c(k) : 0;
        on[k] = ck != 0;
        te[k] = ck ? gf.lg[ck] : 0;
    }
    int count = 0;
    uint64_t flips = 0;
    for (int i = 1; i <= 63; i++) {
        int q = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            int e = te[k] + k;
            if (e >= 63) {
                e -= 63;
            }
            te[k] = e;
            q ^= on[k] ? gf.ex[e] : 0;
        }
        if (q == 0) {
            if (count >= 11) {
                break;
            }
            const int loc = (63 - i) % 63;
            flips |= 1ull << (62 - loc);
            count++;
        }
    }
    if (count != L) {
        return 0;
    }
    *fixed = w ^ flips;
    *nerr = count;
    return 1;
}

struct NidRes {
    int status, nac, duid, errs;
};

// w(x) mod g(x) == 0 for the generator of BCH(63,16,11) over GF(2^6) / x^6 + x + 1 (g = lcm of the minimal polynomials of
// alpha^1 .. alpha^22, degree 47: 0xCD930BDD3B2B with x^j at bit j).  w carries input position p (= x^(62 - p)) at bit p, so
// the division runs from bit 0 up with g's coefficients reversed.
__device__ __forceinline__ bool
bch_63_16_is_codeword(uint64_t w) {
    const uint64_t grev = 0xD4DCBBD0C9B3ull;
#pragma unroll
    for (int p = 0; p < 16; p++) {
        w ^= ((w >> p) & 1ull) ? (grev << p) : 0ull;
    }
    return w == 0;
}

// NAC / DUID / status of a corrected NID word (the tail of decode_nid_codeword(), p25p1_check_nid.cpp:262-301)
__device__ __forceinline__ NidRes
nid_fields(uint64_t fixed, int errs, int parity) {
    NidRes r = {0, 0, 0, 0};
    r.errs = errs;
    int nac = 0, duid = 0;
    for (int i = 0; i < 12; i++) {
        nac = (nac << 1) | (int)((fixed >> i) & 1);
    }
    for (int i = 12; i < 16; i++) {
        duid = (duid << 1) | (int)((fixed >> i) & 1);
    }
    r.nac = nac;
    r.duid = duid;
    // DUIDs defined by TIA-102.BAAA-A table 8-4: 0 HDU, 3 TDU, 5 LDU1, 7 TSDU, A LDU2, C PDU, F TDULC
    const unsigned valid = (1u << 0) | (1u << 3) | (1u << 5) | (1u << 7) | (1u << 10) | (1u << 12) | (1u << 15);
    if (!((valid >> duid) & 1u)) {
        r.errs = 0;
        return r;
    }
    const int want = (duid == 5 || duid == 10) ? 1 : 0;
    r.status = (want == parity) ? 1 : 2;
    return r;
}

__device__ inline NidRes
nid_codeword(const Gf& gf, const Work& wk, uint64_t w, int parity, int* bch_failed) {
    uint64_t fixed;
    int errs;
    if (bch_failed) {
        *bch_failed = 0;
    }
    if (!bch_63_16_decode(gf, wk, w, &fixed, &errs)) {
        if (bch_failed) {
            *bch_failed = 1;
        }
        return NidRes{0, 0, 0, 0};
    }
    return nid_fields(fixed, errs, parity);
}

__device__ __forceinline__ int
rx_nac(uint64_t w) {
    int n = 0;
    for (int i = 0; i < 12; i++) {
        n = (n << 1) | (int)((w >> i) & 1);
    }
    return n;
}

__device__ __forceinline__ uint64_t
put_nac(uint64_t w, int nac) {
    for (int i = 0; i < 12; i++) {
        const uint64_t bit = (uint64_t)((nac >> (11 - i)) & 1);
        w = (w & ~(1ull << i)) | (bit << i);
    }
    return w;
}

// bch_63_16_decode() for ONE word decided by a whole wavefront (every lane calls with the same w; results are wave-uniform): the same
// Massey iteration and Chien search - the same connection polynomial after every step, the same failure tests - with the work
// spread over the lanes instead of ~100 k cycles of dependent LDS table look-ups on one: the odd syndromes as a wave-wide xor of one
// term per received bit, the even ones S_i = S_odd^(2^a) directly, the polynomials C and B one coefficient per lane (a step's
// discrepancy = one product per lane and six ballots, the update = one shifted product per lane), the Chien search one candidate
// root per lane.  Logarithms of C and of the syndromes are carried beside the values (0xFF = zero).
__device__ inline int
bch_63_16_decode_wave(const Gf& gf, uint64_t w, int lane, uint64_t* fixed, int* nerr) {
    const bool bit = lane < 63 && ((w >> lane) & 1ull);
    uint32_t acc[3] = {0u, 0u, 0u};
    {
        const int j = lane < 63 ? 62 - lane : 0; // this bit contributes alpha^(k * j) to S_k
        int j2 = 2 * j;
        j2 -= j2 >= 63 ? 63 : 0;
        int e = j;
#pragma unroll
        for (int k = 0; k < 11; k++) { // S_1, S_3, .., S_21: four to a word
            const uint32_t v = bit ? gf.ex[e] : 0u;
            acc[k >> 2] |= v << (8 * (k & 3));
            e += j2;
            e -= e >= 63 ? 63 : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                acc[q] ^= (uint32_t)__shfl_xor((int)acc[q], off);
            }
        }
    }
    *nerr = 0;
    *fixed = w;
    if ((acc[0] | acc[1] | acc[2]) == 0u) {
        return 1;
    }
    // lane i (1 .. 22) holds S_i: i = odd * 2^a, S_i = S_odd^(2^a) (the serial decoder's S_i = S_{i/2}^2, unrolled)
    int slg = 0xFF;
    if (lane >= 1 && lane <= 22) {
        const int a = __builtin_ctz((unsigned)lane), odd = lane >> a, k = odd >> 1;
        const int h0 = (int)((acc[k >> 2] >> (8 * (k & 3))) & 63u);
        if (h0) {
            int lg = gf.lg[h0];
            for (int r = 0; r < a; r++) {
                lg *= 2;
                lg -= lg >= 63 ? 63 : 0;
            }
            slg = lg;
        }
    }
    int c = lane == 0 ? 1 : 0, b = c;    // coefficient `lane` of the connection polynomial and of its last length change
    int clg = lane == 0 ? 0 : 0xFF;      // log of c (0xFF: c == 0)
    int L = 0, m = 1, bs = 1;
    for (int n = 0; n < 22; n++) {
        const int idx = n + 1 - lane;
        const int sl = __shfl(slg, idx & 63);
        int p = 0;
        if (lane <= L && idx >= 1 && clg != 0xFF && sl != 0xFF) {
            p = gf.ex[clg + sl];
        }
        int d = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            d |= (__popcll(__ballot((p >> q) & 1)) & 1) << q;
        }
        if (d == 0) {
            m++;
            continue;
        }
        const int flg = gf.lg[d] + 63 - gf.lg[bs]; // log of d / b, 1 .. 125
        const int src = lane - m;
        const int bsh = __shfl(b, src & 63);
        int upd = 0;
        if (lane < 24 && src >= 0 && bsh) {
            int e = flg + gf.lg[bsh];
            e -= e >= 126 ? 126 : (e >= 63 ? 63 : 0);
            upd = gf.ex[e];
        }
        const int cold = c;
        c ^= upd;
        if (upd) {
            clg = c ? gf.lg[c] : 0xFF;
        }
        if (2 * L <= n) {
            L = n + 1 - L;
            b = cold;
            bs = d;
            m = 1;
        } else {
            m++;
        }
        if (L > 11) {
            return 0;
        }
    }
    // Chien search: lane r tests alpha^(r + 1); a root there = an error at input position r (i = 63 -> position 62)
    int q = 0;
    {
        const int i = lane + 1;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int lk = __builtin_amdgcn_readlane(clg, k);
            if (k <= L && lk != 0xFF) {
                q ^= gf.ex[(lk + i * k) % 63];
            }
        }
    }
    const uint64_t roots = __ballot(lane < 63 && q == 0);
    const int count = __popcll(roots);
    if (count != L) {
        return 0;
    }
    *fixed = w ^ roots;
    *nerr = count;
    return 1;
}

__device__ inline NidRes
nid_codeword_wave(const Gf& gf, uint64_t w, int parity, int lane, int* bch_failed) {
    uint64_t fixed;
    int errs;
    if (bch_failed) {
        *bch_failed = 0;
    }
    if (!bch_63_16_decode_wave(gf, w, lane, &fixed, &errs)) {
        if (bch_failed) {
            *bch_failed = 1;
        }
        return NidRes{0, 0, 0, 0};
    }
    return nid_fields(fixed, errs, parity);
}

struct ChaseBest {
    int found;
    NidRes dec;
    int score, changes;
};

__device__ inline void
chase_from(const Gf& gf, const Work& wk, uint64_t base, const uint8_t* rel, uint64_t pool, int np, int parity, int parity_rel,
           int threshold, ChaseBest* best) {
    for (int mask = 0; mask < (1 << np); mask++) {
        const int changed = __popc((unsigned)mask);
        if (changed > 3) {
            continue;
        }
        uint64_t cand = base;
        int score = 0;
        for (int b = 0; b < np; b++) {
            if (mask & (1 << b)) {
                const int pos = (int)((pool >> (8 * b)) & 0xFF);
                cand ^= 1ull << pos;
                score += rel[pos];
            }
        }
        if (changed && score > threshold * changed) {
            continue;
        }
        const NidRes dec = nid_codeword(gf, wk, cand, parity, nullptr);
        if (dec.status <= 0) {
            continue;
        }
        const int sc = score + (dec.status == 2 ? parity_rel : 0);
        const bool better = !best->found || sc < best->score
                            || (sc == best->score && dec.status == 1 && best->dec.status != 1)
                            || (sc == best->score && dec.status == best->dec.status && dec.errs < best->dec.errs)
                            || (sc == best->score && dec.status == best->dec.status && dec.errs == best->dec.errs
                                && changed < best->changes);
        if (better) {
            best->found = 1;
            best->dec = dec;
            best->score = sc;
            best->changes = changed;
        }
    }
}


// One NID decided by a whole wavefront (every lane calls this with the same arguments; `wk` is the lane's own work area):
// hard decode + observed-NAC retry run redundantly on all lanes (same instruction count as on one), the Chase search - the
// reference's sequence of <= 186 candidates - one candidate per lane and round, the winner by a wave-wide minimum over the packed
// key (score, status != 1, error count, flips, sequence index) = what the sequential "strictly better" scan keeps.
// rel = 63 per-bit reliabilities (LDS or global), masks = the 93 flip masks of <= 3 bits in increasing order.
__device__ inline NidRes
nid_decode_wave(const Gf& gf, const Work& wk, uint64_t w, const uint8_t* rel, int par, int prel, int obs, int threshold,
                const uint8_t* masks, int lane) {
    const bool obs_ok = obs > 0 && obs < 0xFFF;
    if (bch_63_16_is_codeword(w)) { // the common case: no bit error - the decoder returns the word with an error count of 0
        return nid_fields(w, 0, par);
    }
    int failed = 0;
    NidRes hard = nid_codeword_wave(gf, w, par, lane, &failed);
    if (hard.status == 0 && failed && obs_ok && rx_nac(w) != obs) {
        hard = nid_codeword_wave(gf, put_nac(w, obs), par, lane, nullptr);
    }
    if (hard.status > 0 || rel == nullptr) { // rel == nullptr: hard decision only
        return hard;
    }
    const bool two_bases = obs_ok && rx_nac(w) != obs;
    uint64_t pool = 0, taken = 0;
    int below = 0;
    for (int i = 0; i < 63; i++) {
        below += rel[i] < threshold;
    }
    for (int k = 0; k < 8; k++) {
        int bi = -1, bv = 256;
        for (int i = 0; i < 63; i++) {
            if (!((taken >> i) & 1) && rel[i] < bv) {
                bv = rel[i];
                bi = i;
            }
        }
        pool |= (uint64_t)bi << (8 * k);
        taken |= 1ull << bi;
    }
    int np = below < 8 ? below : 8;
    np = np < 6 ? 6 : np;
    const int per_base = (np == 6) ? 42 : ((np == 7) ? 64 : 93);
    const int total = two_bases ? 2 * per_base : per_base;
    uint32_t best_key = 0xFFFFFFFFu;
    NidRes best_dec = {0, 0, 0, 0};
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        uint32_t key = 0xFFFFFFFFu;
        NidRes dec = {0, 0, 0, 0};
        bool run = t < total;
        int mask = 0, base_idx = 0, changed = 0, score = 0;
        uint64_t cand = w;
        if (run) {
            base_idx = t / per_base;
            mask = masks[t - base_idx * per_base];
            changed = __popc((unsigned)mask);
            cand = base_idx ? put_nac(w, obs) : w;
            for (int b = 0; b < np; b++) {
                if (mask & (1 << b)) {
                    const int pos = (int)((pool >> (8 * b)) & 0xFF);
                    cand ^= 1ull << pos;
                    score += rel[pos];
                }
            }
            run = !(changed && score > threshold * changed);
        }
        if (__any(run)) {
            if (run) {
                dec = nid_codeword(gf, wk, cand, par, nullptr);
            }
            if (run && dec.status > 0) {
                const int sc = score + (dec.status == 2 ? prel : 0);
                key = ((uint32_t)sc << 18) | ((uint32_t)(dec.status != 1) << 17) | ((uint32_t)(dec.errs & 63) << 11)
                      | ((uint32_t)changed << 9) | (uint32_t)(base_idx * 256 + mask);
            }
        }
        if (key < best_key) {
            best_key = key;
            best_dec = dec;
        }
    }
    uint32_t m = best_key;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off);
        m = o < m ? o : m;
    }
    if (m == 0xFFFFFFFFu) {
        return hard;
    }
    const int owner = __ffsll((long long)__ballot(best_key == m)) - 1; // keys are unique: exactly one lane
    NidRes r;
    r.status = __shfl(best_dec.status, owner);
    r.nac = __shfl(best_dec.nac, owner);
    r.duid = __shfl(best_dec.duid, owner);
    r.errs = __shfl(best_dec.errs, owner);
    return r;
}

// the 93 masks of at most three flips among eight positions, increasing (42 of them below 64, 64 below 128)
__device__ inline void
chase_masks_fill(uint8_t* masks) { // one thread
    int k = 0;
    for (int m = 0; m < 256; m++) {
        if (__popc((unsigned)m) <= 3) {
            masks[k++] = (uint8_t)m;
        }
    }
}

} // namespace ddn_nid
#endif
