"""Synthetic DMR base-station data bursts for the chain tests: CACH (TACT Hamming(7,4)), slot type Golay(20,8), BPTC(196,96) payloads
with the CRC / RS(12,9) the data type names, rate 3/4 and rate 1 payloads, the BS data sync word.  The block codes' parities are not
restated here: they are found by search against the oracle's decoders (a word is a code word when the decoder accepts it unchanged),
so the generator cannot disagree with the decoders it feeds."""
import ctypes as C
import functools

import numpy as np

import fec3
import orc
import p25gen
import rx4

BS_DATA_SYNC = "313333111331131131331131"
CRC_MASK = {0: 0x6969, 1: 0x969696, 2: 0x999999, 3: 0xA5A5, 4: 0xAAAA, 5: 0, 6: 0xCCCC, 7: 0x0F0, 8: 0x1FF, 10: 0x10F, 11: 0x3333}


def _bits(v, n):
    return [(int(v) >> (n - 1 - i)) & 1 for i in range(n)]


@functools.lru_cache(None)
def _parity_matrix(code):
    """P [k][n - k] of the systematic code `code` (fec3.CODES): row i = the parity bits of the data word with only bit i set"""
    n, k, _ = fec3.CODES[code]
    r = n - k
    P = np.zeros((k, r), np.uint8)
    par = ((np.arange(1 << r)[:, None] >> np.arange(r)[None, ::-1]) & 1).astype(np.uint8)
    for i in range(k):
        d = np.zeros(k, np.uint8)
        d[i] = 1
        words = np.concatenate([np.tile(d, (1 << r, 1)), par], axis=1)
        found = None
        if code in (0, 5, 6, 7):                   # in-place decoders: accepted and unchanged = zero syndrome
            got, _, ok = fec3.oracle_decode(code, words)
            good = [j for j in range(1 << r) if ok[j] and np.array_equal(got[j], words[j])]
            assert len(good) == 1, (code, i, len(good))
            found = good[0]
        else:                                       # Hamming(n, k, 3): the code word is the one every single flip decodes back to
            for j in range(1 << r):
                flips = np.tile(words[j], (n + 1, 1))
                for q in range(n):
                    flips[q + 1, q] ^= 1
                _, dec, ok = fec3.oracle_decode(code, flips)
                if ok.all() and (dec == d[None, :]).all():
                    assert found is None
                    found = j
        P[i] = par[found]
    return P


def encode(code, data_bits):
    d = np.asarray(data_bits, np.uint8)
    return np.concatenate([d, (d @ _parity_matrix(code)) & 1]).astype(np.uint8)


def bptc_196x96(bits96, r3=(0, 0, 0)):
    """96 payload bits -> the 196 info bits on the air (BPTC_196x96_Extract_Data's matrix, src/fec/bptc.c:61-126, interleaved)"""
    m = np.zeros((13, 15), np.uint8)
    b = list(map(int, bits96))
    m[0, 0:3] = [r3[2], r3[1], r3[0]]
    m[0, 3:11] = b[:8]
    m[1:9, :11] = np.array(b[8:], np.uint8).reshape(8, 11)
    for i in range(9):
        m[i] = encode(3, m[i, :11])
    for j in range(15):
        m[:, j] = encode(2, m[:9, j])
    d = np.zeros(196, np.uint8)
    d[1:] = m.reshape(-1)
    return np.array([d[(i * 13) % 196] for i in range(196)], np.uint8)        # (BPTCDeInterleavingIndex[i] = 13 i mod 196)


def _gf():
    exp, log = [0] * 256, [0] * 256
    x = 1
    for i in range(255):
        exp[i], log[x] = x, i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    exp[255] = 1
    return exp, log


def rs_12_9_parity(data9):
    """three parity bytes so that the word's syndromes at alpha^1..3 vanish (rs_12_9_calc_syndrome, src/fec/rs-12-9.c)"""
    exp, log = _gf()

    def mul(a, b):
        return 0 if a == 0 or b == 0 else exp[(log[a] + log[b]) % 255]
    g = [1]
    for j in (1, 2, 3):
        g = [a ^ b for a, b in zip(g + [0], [0] + [mul(c, exp[j]) for c in g])]          # (x + alpha^j), highest degree first
    rem = list(map(int, data9)) + [0, 0, 0]
    for i in range(9):
        f = rem[i]
        if f:
            for j in range(1, 4):
                rem[i + j] ^= mul(g[j], f)
    return rem[9:]


def payload_bits(dtype, rng, good_crc=True, confirmed=False, dbsn=0):
    """96 payload bits of a BPTC burst of the given data type, with the CRC / RS parity the reference checks"""
    if dtype in (1, 2):                                         # full link control: 9 bytes + RS(12,9) parity ^ mask
        d = [int(x) for x in rng.integers(0, 256, 9)]
        p = rs_12_9_parity(d)
        m = CRC_MASK[dtype]
        by = d + [p[0] ^ (m >> 16), p[1] ^ ((m >> 8) & 0xFF), p[2] ^ (m & 0xFF)]
        if not good_crc:
            by[2] ^= 0x41
            by[7] ^= 0x80                                       # two symbol errors: beyond RS(12,9)
        return np.unpackbits(np.array(by, np.uint8))
    if dtype == 7 and confirmed:                                # DBSN(7) | CRC9 ^ 0x0F0 | 10 bytes
        pay = [int(x) for x in rng.integers(0, 2, 80)]
        c = p25gen.crc9(pay + _bits(dbsn, 7)) ^ CRC_MASK[7] ^ (0 if good_crc else 0x005)
        return np.array(_bits(dbsn, 7) + _bits(c, 9) + pay, np.uint8)
    b = [int(x) for x in rng.integers(0, 2, 80)]
    c = (rx4.crc_ccitt_bits(b) ^ 0xFFFF ^ CRC_MASK.get(dtype, 0)) & 0xFFFF
    if not good_crc:
        c ^= 0x0100
    return np.array(b + _bits(c, 16), np.uint8)


def r34_bytes(rng, confirmed=False, dbsn=0, good_crc=True):
    if not confirmed:
        return rng.integers(0, 256, 18).astype(np.uint8)
    pay = rng.integers(0, 256, 16).astype(np.uint8)
    c = p25gen.crc9(list(np.unpackbits(pay)) + _bits(dbsn, 7)) ^ 0x1FF ^ (0 if good_crc else 0x011)
    return np.array([((dbsn & 0x7F) << 1) | (c >> 8), c & 0xFF] + [int(x) for x in pay], np.uint8)


def burst(slot, cc, dtype, info196):
    """one 144-dibit BS data burst"""
    tact = encode(0, [1, slot & 1, 0, 0])
    cach = np.zeros(24, np.uint8)
    cach[:7] = tact
    air = [int(cach[rx4.CACH_IL[k]]) for k in range(24)]
    st = encode(5, _bits(cc, 4) + _bits(dtype, 4))
    info = list(map(int, info196))
    bits = air + info[:98] + list(st[:10]) + [b for ch in BS_DATA_SYNC for b in ((1, 1) if ch == "3" else (0, 1))] + list(st[10:]) + info[98:]
    assert len(bits) == 288
    return np.array([2 * bits[2 * i] + bits[2 * i + 1] for i in range(144)], np.int8)


def r34_info(bytes18):
    d = p25gen.encode_three_quarter_rate(bytes18)
    return np.array([b for x in d for b in ((int(x) >> 1) & 1, int(x) & 1)], np.uint8)


def crc9_confirmed_rate1(info196, mask=0x10F):
    return p25gen.crc9(list(info196[16:96]) + list(info196[100:196]) + list(info196[0:7])) ^ mask


BS_VOICE_SYNC = "131111333113313313113313"


def bptc_128x77(lc72, crc5=None):
    """72 link-control bits (+ their 5-bit checksum, ComputeCrc5Bit) -> the 8 x 16 matrix of BPTC_128x77_Extract_Data (src/fec/bptc.c:167-258)"""
    import bptc_small
    b = [int(x) for x in lc72]
    if crc5 is None:
        crc5 = int(np.packbits(np.array(b, np.uint8)).astype(np.int64).sum()) % 31
    c = _bits(crc5, 5)
    m = np.zeros((8, 16), np.uint8)
    m[0] = bptc_small.hamming_row(b[0:11])
    m[1] = bptc_small.hamming_row(b[11:22])
    for r in range(5):
        m[2 + r] = bptc_small.hamming_row(b[22 + 10 * r:32 + 10 * r] + [c[r]])
    m[7] = m[:7].sum(0) & 1
    return m


def voice_burst(slot, cc, index, emb32, rng, lcss=0):
    """one 144-dibit BS voice burst: index 0 = burst A (voice sync word), 1..5 = B..F with EMB (QR(16,7,6) over colour code, PI, LCSS)
    around 32 bits of embedded signalling; the AMBE payload is random"""
    tact = encode(0, [1, slot & 1, 0, 0])
    cach = np.zeros(24, np.uint8)
    cach[:7] = tact
    air = [int(cach[rx4.CACH_IL[k]]) for k in range(24)]
    voice = [int(x) for x in rng.integers(0, 2, 216)]
    if index == 0:
        mid = [b for ch in BS_VOICE_SYNC for b in ((1, 1) if ch == "3" else (0, 1))]
    else:
        emb = encode(7, _bits(cc, 4) + [0] + _bits(lcss, 2))
        mid = list(emb[:8]) + [int(x) for x in emb32] + list(emb[8:])
    bits = air + voice[:108] + mid + voice[108:]
    assert len(bits) == 288
    return np.array([2 * bits[2 * i] + bits[2 * i + 1] for i in range(144)], np.int8)


def voice_superframe(slot, cc, lc72, rng, crc5=None):
    """six voice bursts A..F of one time slot carrying the link control in B..E (column-major 32-bit fragments of the BPTC(128,77) matrix)"""
    m = bptc_128x77(lc72, crc5)
    cm = [int(m[r, c]) for c in range(16) for r in range(8)]
    lcss = [0, 1, 3, 3, 2, 0]
    return [voice_burst(slot, cc, i, cm[32 * (i - 1):32 * i] if 1 <= i <= 4 else [0] * 32, rng, lcss[i]) for i in range(6)]
