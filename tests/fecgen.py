"""Test-side generators for the trellis / Viterbi decoders: encoders for valid codewords (from the TIA-102 /
ETSI FSM tables the oracle exposes), channel noise, and ctypes helpers.  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

import orc

VP = C.c_void_p


def _tbl(fn, n):
    fn.restype = C.POINTER(C.c_uint8)
    return np.array([fn()[i] for i in range(n)], np.uint8)


def tables():
    o = orc.oracle()
    il = np.zeros(98, np.uint8)
    o.orc_trellis_interleave_98.argtypes = [VP]
    o.orc_trellis_interleave_98(il.ctypes.data)
    return dict(il=il, half=_tbl(o.orc_tbl_p25_half_rate_nibble, 16), p2n=_tbl(o.orc_tbl_r34_point_to_nibble, 16),
                n2p=_tbl(o.orc_tbl_r34_nibble_to_point, 16), fsm=_tbl(o.orc_tbl_r34_fsm, 64))


def bind_oracle_fec():
    o = orc.oracle()
    o.orc_p25_12_soft_llr.argtypes = [VP, VP]
    o.orc_r34_decode.argtypes = [VP, VP, VP]
    o.orc_nxdn_conv_decode.argtypes = [VP, VP, C.c_int, VP, VP, C.c_int]
    o.orc_m17_viterbi_decode.argtypes = [VP, VP, C.c_int]
    o.orc_m17_viterbi_decode.restype = C.c_uint32
    o.orc_m17_viterbi_decode_punctured.argtypes = [VP, VP, VP, C.c_int, C.c_int]
    o.orc_m17_viterbi_decode_punctured.restype = C.c_uint32
    return o


def gen_p25_half_rate(rng, n, amp=600, sigma=250.0, random_frac=0.2):
    """-> (llr [n,196] int16 in received order, states [n,49])."""
    t = tables()
    st = rng.integers(0, 4, (n, 49)).astype(np.int64)
    prev = np.concatenate([np.zeros((n, 1), np.int64), st[:, :-1]], axis=1)
    nib = t["half"][(prev << 2) | st].astype(np.int64)                # [n,49]
    bits = np.stack([(nib >> 3) & 1, (nib >> 2) & 1, (nib >> 1) & 1, nib & 1], axis=2).reshape(n, 196)
    llr_dei = (2 * bits - 1) * amp + rng.normal(0.0, sigma, (n, 196))
    k = int(n * random_frac)
    if k:
        llr_dei[:k] = rng.integers(-32768, 32768, (k, 196))
    llr_dei = np.clip(np.rint(llr_dei), -32768, 32767).astype(np.int16)
    rx = np.zeros((n, 196), np.int16)
    il = t["il"].astype(np.int64)
    rx[:, 0::2] = llr_dei[:, 2 * il]
    rx[:, 1::2] = llr_dei[:, 2 * il + 1]
    return np.ascontiguousarray(rx), st


def gen_r34(rng, n, p_err=0.03, random_frac=0.2):
    """-> (dibits [n,98] uint8 received order, reliab [n,98] uint8, tribits [n,49])."""
    t = tables()
    tri = rng.integers(0, 8, (n, 49)).astype(np.int64)
    tri[:, 48] = 0
    prev = np.concatenate([np.zeros((n, 1), np.int64), tri[:, :-1]], axis=1)
    point = t["fsm"][prev * 8 + tri].astype(np.int64)
    nib = t["p2n"][point].astype(np.int64)
    dei = np.zeros((n, 98), np.int64)
    dei[:, 0::2] = nib >> 2
    dei[:, 1::2] = nib & 3
    flip = rng.random((n, 98)) < p_err
    dei = np.where(flip, rng.integers(0, 4, (n, 98)), dei)
    k = int(n * random_frac)
    if k:
        dei[:k] = rng.integers(0, 4, (k, 98))
    il = t["il"].astype(np.int64)
    rx = dei[:, il].astype(np.uint8)
    rel = np.where(flip[:, il], rng.integers(0, 80, (n, 98)), rng.integers(100, 256, (n, 98))).astype(np.uint8)
    return np.ascontiguousarray(rx), np.ascontiguousarray(rel), tri


def conv_k5_encode(bits):
    """K=5 R=1/2 encoder, G1 = 0x19, G2 = 0x17 on a 5-bit register (NXDN / M17 / YSF).  bits [n, L] -> [n, 2L]."""
    n, L = bits.shape
    reg = np.zeros(n, np.int64)
    out = np.zeros((n, 2 * L), np.uint8)
    par = np.array([bin(i).count("1") & 1 for i in range(32)], np.uint8)
    for i in range(L):
        reg = ((reg << 1) | bits[:, i]) & 0x1F
        out[:, 2 * i] = par[reg & 0x19]
        out[:, 2 * i + 1] = par[reg & 0x17]
    return out


def gen_nxdn(rng, n, n_steps, p_err=0.04, p_erase=0.05, random_frac=0.2):
    """-> (sym [n, 2*n_steps] uint8 in {0,1,2}, rel [n, 2*n_steps] uint8)."""
    data = rng.integers(0, 2, (n, n_steps)).astype(np.int64)
    data[:, -4:] = 0
    enc = conv_k5_encode(data).astype(np.int64)
    flip = rng.random(enc.shape) < p_err
    enc = np.where(flip, 1 - enc, enc)
    sym = 2 * enc
    er = rng.random(enc.shape) < p_erase
    sym = np.where(er, 1, sym)
    k = int(n * random_frac)
    if k:
        sym[:k] = rng.integers(0, 3, (k, 2 * n_steps))
    rel = np.where(flip | er, rng.integers(0, 90, enc.shape), rng.integers(120, 256, enc.shape)).astype(np.uint8)
    return np.ascontiguousarray(sym.astype(np.uint8)), np.ascontiguousarray(rel)


def gen_m17(rng, n, in_len, sigma=9000.0, random_frac=0.2):
    """-> soft [n, in_len] uint16 (0 = strong 0, 0xFFFF = strong 1)."""
    L = in_len // 2
    data = rng.integers(0, 2, (n, L)).astype(np.int64)
    data[:, -4:] = 0
    enc = conv_k5_encode(data).astype(np.float64)[:, :in_len]
    soft = enc * 65535.0 + rng.normal(0.0, sigma, enc.shape)
    k = int(n * random_frac)
    if k:
        soft[:k] = rng.integers(0, 65536, (k, in_len))
    return np.ascontiguousarray(np.clip(np.rint(soft), 0, 65535).astype(np.uint16))


# oracle batch helpers ------------------------------------------------------------------------------------
def oracle_p25_half_rate(llr):
    o = bind_oracle_fec()
    n = llr.shape[0]
    out = np.zeros((n, 12), np.uint8)
    met = np.zeros(n, np.int32)
    for i in range(n):
        met[i] = o.orc_p25_12_soft_llr(llr[i].ctypes.data, out[i].ctypes.data)
    return out, met


def oracle_r34(dibits, reliab=None):
    o = bind_oracle_fec()
    n = dibits.shape[0]
    out = np.zeros((n, 18), np.uint8)
    for i in range(n):
        o.orc_r34_decode(dibits[i].ctypes.data, reliab[i].ctypes.data if reliab is not None else None,
                         out[i].ctypes.data)
    return out


def oracle_nxdn(sym, rel, n_steps, n_bits, metrics=None):
    o = bind_oracle_fec()
    n = sym.shape[0]
    stride = (n_bits + 7) // 8
    out = np.zeros((n, stride), np.uint8)
    m = np.zeros((n, 16), np.uint16) if metrics is None else metrics.copy()
    for i in range(n):
        o.orc_nxdn_conv_decode(sym[i].ctypes.data, rel[i].ctypes.data if rel is not None else None, n_steps,
                               m[i].ctypes.data, out[i].ctypes.data, n_bits)
    return out, m


def oracle_m17(soft, punct=None):
    o = bind_oracle_fec()
    n, in_len = soft.shape
    if punct is None:
        u_len = in_len
    else:
        i = u = p = 0
        while i < in_len:
            i += int(punct[p] != 0)
            u += 1
            p = (p + 1) % len(punct)
        u_len = u
    stride = (u_len // 2 + 3) // 8 + 1
    out = np.zeros((n, stride), np.uint8)
    cost = np.zeros(n, np.uint32)
    for k in range(n):
        if punct is None:
            cost[k] = o.orc_m17_viterbi_decode(out[k].ctypes.data, soft[k].ctypes.data, in_len)
        else:
            cost[k] = o.orc_m17_viterbi_decode_punctured(out[k].ctypes.data, soft[k].ctypes.data, punct.ctypes.data,
                                                          in_len, len(punct))
    return out, cost, stride


# ---- BCH(63,16,11) / P25 NID test generators ---------------------------------------------------------------
def _gf64():
    exp = [0] * 126
    log = [0] * 64
    x = 1
    for i in range(63):
        exp[i] = exp[i + 63] = x
        log[x] = i
        x <<= 1
        if x & 64:
            x ^= 0x43
    return exp, log


def bch_63_16_generator():
    """g(x) over GF(2) = lcm of the minimal polynomials of alpha^1..alpha^22 (alpha: x^6+x+1).  Bit i = coeff x^i."""
    exp, log = _gf64()
    roots = set()
    for e in range(1, 23):
        c = e
        while c not in roots:
            roots.add(c)
            c = (2 * c) % 63
    poly = [1]  # coefficients in GF(64), low degree first
    for r in sorted(roots):
        a = exp[r]
        nxt = [0] * (len(poly) + 1)
        for i, c in enumerate(poly):
            nxt[i + 1] ^= c
            if c:
                nxt[i] ^= exp[(log[c] + log[a]) % 63]
        poly = nxt
    assert all(c in (0, 1) for c in poly) and len(poly) == 48
    return sum(c << i for i, c in enumerate(poly))


_G63 = None


def bch_63_16_encode(data16):
    """data16: iterable of 16 bits (MSB first, NAC then DUID) -> np.uint8[63] in the decoder's input order."""
    global _G63
    if _G63 is None:
        _G63 = bch_63_16_generator()
    d = 0
    for b in data16:
        d = (d << 1) | int(b)
    rem = d << 47
    for sh in range(62, 46, -1):
        if rem & (1 << sh):
            rem ^= _G63 << (sh - 47)
    cw = (d << 47) | rem  # coefficient of x^j at bit j; input[i] = coeff x^(62-i)
    return np.array([(cw >> (62 - i)) & 1 for i in range(63)], np.uint8)


def gen_nid(rng, n, max_err=14, invalid_duids=True):
    """-> bits [n,63], reliab [n,63], observed_nac [n], parity [n], parity_rel [n] with 0..max_err bit errors."""
    duids = np.array([0, 3, 5, 7, 10, 12, 15])
    bits = np.zeros((n, 63), np.uint8)
    rel = rng.integers(90, 256, (n, 63)).astype(np.uint8)
    obs = np.zeros(n, np.int32)
    par = np.zeros(n, np.uint8)
    prel = rng.integers(0, 256, n).astype(np.uint8)
    for i in range(n):
        nac = int(rng.integers(1, 0xFFF))
        duid = int(duids[rng.integers(0, len(duids))])
        if invalid_duids and i % 11 == 0:
            duid = int(rng.integers(0, 16))  # sometimes an undefined DUID
        data = [(nac >> (11 - k)) & 1 for k in range(12)] + [(duid >> (3 - k)) & 1 for k in range(4)]
        cw = bch_63_16_encode(data)
        ne = int(rng.integers(0, max_err + 1))
        pos = rng.choice(63, ne, replace=False)
        cw[pos] ^= 1
        rel[i, pos] = rng.integers(0, 100, ne)
        bits[i] = cw
        par[i] = (1 if duid in (5, 10) else 0) ^ (1 if i % 7 == 0 else 0)
        obs[i] = nac if i % 3 == 0 else (0 if i % 3 == 1 else int(rng.integers(0, 0x1000)))
    return bits, rel, obs, par, prel


# ---- P25p1 Golay(24,12,8) / RS GF(64) generators for the block-code tests ----------------------------------------
def golay24_encode(data_bits):
    """data_bits: 12 bits, data_bits[k] = bit k of the 12-bit data word (as the reference's char array) ->
    parity[12] in the reference's char-array order (11 check bits then the overall parity bit)."""
    d = 0
    for k, b in enumerate(data_bits):
        d |= int(b) << k
    cw = d
    for _ in range(12):
        if cw & 1:
            cw ^= 0xAE3
        cw >>= 1
    word = (cw << 12) | d
    if bin(word).count("1") & 1:
        word ^= 0x800000
    return np.array([(word >> (12 + k)) & 1 for k in range(12)], np.uint8)


_RS_GEN = {}


def rs63_generator(t):
    if t not in _RS_GEN:
        ex, lg = _gf64()
        g = [1]
        for i in range(1, 2 * t + 1):
            r = int(ex[i])
            ng = [0] * (len(g) + 1)
            for k, c in enumerate(g):          # multiply by (x + alpha^i)
                ng[k + 1] ^= c
                ng[k] ^= int(ex[(lg[c] + lg[r]) % 63]) if c else 0
            g = ng
        _RS_GEN[t] = g
    return _RS_GEN[t]


def rs63_encode(data_syms, t):
    """data_syms: k' <= 63-2t symbols (placed at x^(2t)...), returns parity[2t] (coefficients of x^0..x^(2t-1))."""
    ex, lg = _gf64()
    g = rs63_generator(t)
    n2 = 2 * t
    rem = [0] * n2
    for s in reversed([int(v) for v in data_syms]):
        fb = s ^ rem[n2 - 1]
        for k in range(n2 - 1, 0, -1):
            rem[k] = rem[k - 1] ^ (int(ex[(lg[fb] + lg[g[k]]) % 63]) if fb and g[k] else 0)
        rem[0] = int(ex[(lg[fb] + lg[g[0]]) % 63]) if fb and g[0] else 0
    return rem


def syms_to_bits6(syms):
    s = np.asarray(syms, np.int64)
    return ((s[:, None] >> np.arange(5, -1, -1)[None, :]) & 1).astype(np.uint8)


P25_RS_CODES = {"24_12_13": (12, 12, 6), "24_16_9": (8, 16, 4), "36_20_17": (16, 20, 8)}   # n_par, n_data, t


def gen_p25_rs(rng, code, n, max_extra=3):
    """-> data bits [n, n_data, 6], parity bits [n, n_par, 6] of codewords with 0..t+max_extra symbol errors."""
    n_par, n_data, t = P25_RS_CODES[code]
    data = np.zeros((n, n_data, 6), np.uint8)
    par = np.zeros((n, n_par, 6), np.uint8)
    for i in range(n):
        d = rng.integers(0, 64, n_data)
        p = rs63_encode(d, t)
        w = np.array(list(p) + list(d))
        ne = int(rng.integers(0, t + max_extra + 1))
        pos = rng.choice(n_par + n_data, ne, replace=False)
        w[pos] ^= rng.integers(1, 64, ne)
        par[i] = syms_to_bits6(w[:n_par])
        data[i] = syms_to_bits6(w[n_par:])
    return data, par


def gen_golay24(rng, n, length):
    """-> data bits [n, length], parity [n, 12] with 0..5 bit errors (length 6 or 12: P25 (18,6,8) / (24,12,8))."""
    data = np.zeros((n, length), np.uint8)
    par = np.zeros((n, 12), np.uint8)
    for i in range(n):
        d12 = np.zeros(12, np.uint8)
        d12[12 - length:] = rng.integers(0, 2, length)
        p = golay24_encode(d12)
        w = np.concatenate([d12[12 - length:], p])
        ne = int(rng.integers(0, 6))
        w[rng.choice(length + 12, ne, replace=False)] ^= 1
        data[i] = w[:length]
        par[i] = w[length:]
    return data, par
