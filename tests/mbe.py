"""Test helpers for the vocoder stage: ctypes wrappers around the CPU restatement (oracle/ddn_oracle_mbe.c) and
generators of valid IMBE / AMBE frames.  TEST INFRASTRUCTURE - the product never imports this."""
import ctypes as C
import json
import os

import numpy as np

import ddn
import orc

GOLAY_G = 0xC75
HAMMING_MASKS = (0x7F08, 0x78E4, 0x66D2, 0x55B1)


def tables():
    t = ddn.MbeTables()
    assert ddn.lib().ddn_mbe_default_tables(C.byref(t)) == 0
    return t


def _o():
    o = orc.oracle()
    if not getattr(o, "_mbe_bound", False):
        o.om_imbe7200x4400_decode.restype = C.c_int
        o.om_imbe7200x4400_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        o.om_ambe3600x2450_decode.restype = C.c_int
        o.om_ambe3600x2450_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        o.om_init_parms.restype = None
        o.om_init_parms.argtypes = [C.c_void_p] * 3
        o.om_process_batch.restype = C.c_int
        o.om_process_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        o._mbe_bound = True
    return o


def oracle_frame_decode(codec, frames, soft=False):
    """frames u8 [n][8][23] | [n][4][24] -> (bits u8 [n][88|49], result i32 [n][5], rc [n])"""
    frames = np.ascontiguousarray(frames, np.uint8)
    n = frames.shape[0]
    nb = 88 if codec == ddn.MBE_IMBE else 49
    bits = np.zeros((n, nb), np.uint8)
    res = np.zeros((n, 5), np.int32)
    rc = np.zeros(n, np.int32)
    fn = _o().om_imbe7200x4400_decode if codec == ddn.MBE_IMBE else _o().om_ambe3600x2450_decode
    for i in range(n):
        rc[i] = fn(frames[i].ctypes.data, int(soft), bits[i].ctypes.data, res[i].ctypes.data)
    return bits, res, rc


def oracle_process_stream(v, codec, bits, talk_path, res_in=None):
    """one talk path's frames in order through OracleVocoder `v` (n_streams = 1; its parameters carry on): bits u8 [F][nb],
    res_in i32 [F][5] = the frame decode's results -> pcm f32 [F][160].  talk_path = the path's index in its batch (the noise seed)"""
    bits = np.ascontiguousarray(bits, np.uint8)
    F = bits.shape[0]
    pcm = np.zeros((F, 160), np.float32)
    if res_in is not None:
        res_in = np.ascontiguousarray(res_in, np.int32)
    rc = _o().om_process_batch(codec, C.addressof(v.tab), bits.ctypes.data, res_in.ctypes.data if res_in is not None else None, v.tail,
                               talk_path, 1, F, pcm.ctypes.data, None, C.addressof(v.cur), C.addressof(v.prev), C.addressof(v.enh))
    assert rc == 0
    return pcm


class OracleVocoder:
    """S talk paths of om_process (mbe_processImbe4400Dataf / mbe_processAmbe2450Dataf restated)."""

    def __init__(self, codec, n_streams, tail_rule=0, tab=None):
        self.codec, self.S, self.tail = codec, n_streams, tail_rule
        self.tab = tab if tab is not None else tables()
        self.cur = (ddn.MbeParms * n_streams)()
        self.prev = (ddn.MbeParms * n_streams)()
        self.enh = (ddn.MbeParms * n_streams)()
        sz = C.sizeof(ddn.MbeParms)
        for s in range(n_streams):
            _o().om_init_parms(C.addressof(self.cur) + s * sz, C.addressof(self.prev) + s * sz, C.addressof(self.enh) + s * sz)

    def run(self, bits, res_in=None):
        """bits u8 [S][F][nb] -> (pcm f32 [S][F][160], res_out i32 [S][F][5])"""
        bits = np.ascontiguousarray(bits, np.uint8)
        S, F = bits.shape[0], bits.shape[1]
        assert S == self.S
        pcm = np.zeros((S, F, 160), np.float32)
        res_out = np.zeros((S, F, 5), np.int32)
        if res_in is not None:
            res_in = np.ascontiguousarray(res_in, np.int32)
        rc = _o().om_process_batch(self.codec, C.addressof(self.tab), bits.ctypes.data,
                                   res_in.ctypes.data if res_in is not None else None, self.tail, 0, S, F, pcm.ctypes.data,
                                   res_out.ctypes.data, C.addressof(self.cur), C.addressof(self.prev), C.addressof(self.enh))
        return pcm, res_out, rc


def parms_tuple(p):
    """Every field of one mbe_parms as numpy arrays (for state comparisons)."""
    return (np.float32(p.w0), p.L, p.K, np.array(p.Vl[:]), np.array(p.Ml[:], np.float32), np.array(p.log2Ml[:], np.float32),
            np.array(p.PHIl[:], np.float32), np.array(p.PSIl[:], np.float32), np.float32(p.gamma), p.un, p.repeat)


def parms_equal(a, b):
    for x, y in zip(parms_tuple(a), parms_tuple(b)):
        x, y = np.asarray(x), np.asarray(y)
        if x.dtype == np.float32:
            if not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                return False
        elif not np.array_equal(x, y):
            return False
    return True


# ---- frame builders (encoder side of mbelib's ecc: systematic Golay(23,12) / Hamming(15,11), PN scrambling) ----------
def golay_parity(d12):
    r = d12 << 11
    for i in range(22, 10, -1):
        if (r >> i) & 1:
            r ^= GOLAY_G << (i - 11)
    return r & 0x7FF


def hamming_encode(d11):
    w = d11 << 4
    for i, m in enumerate(HAMMING_MASKS):
        if bin(w & (m & 0x7FF0)).count("1") & 1:
            w |= 1 << (3 - i)
    return w


def pn_masks(u0, lens):
    pr = (16 * u0) & 0xFFFF
    out = []
    for ln in lens:
        m = 0
        for j in range(ln - 1, -1, -1):
            pr = (173 * pr + 13849) & 0xFFFF
            m |= (pr >> 15) << j
        out.append(m)
    return out


def imbe_encode(bits88):
    """88 parameter bits (imbe_d order) -> frame u8 [8][23] (mbelib row order, bit j of row word at [row][j])"""
    b = [int(x) for x in bits88]

    def take(k, n):
        v = 0
        for x in b[k:k + n]:
            v = (v << 1) | x
        return v
    u = [take(0, 12), take(12, 12), take(24, 12), take(36, 12), take(48, 11), take(59, 11), take(70, 11), take(81, 7)]
    rows = [(u[i] << 11) | golay_parity(u[i]) for i in range(4)] + [hamming_encode(u[i]) for i in range(4, 7)] + [u[7]]
    masks = pn_masks(u[0], [23, 23, 23, 15, 15, 15])
    for i in range(1, 7):
        rows[i] ^= masks[i - 1]
    fr = np.zeros((8, 23), np.uint8)
    for r in range(8):
        for j in range(23):
            fr[r, j] = (rows[r] >> j) & 1
    return fr


def ambe_encode(bits49):
    b = [int(x) for x in bits49]

    def take(k, n):
        v = 0
        for x in b[k:k + n]:
            v = (v << 1) | x
        return v
    u0, u1, u2, u3 = take(0, 12), take(12, 12), take(24, 11), take(35, 14)
    c0 = (u0 << 11) | golay_parity(u0)
    c1 = ((u1 << 11) | golay_parity(u1)) ^ pn_masks(u0, [23])[0]
    fr = np.zeros((4, 24), np.uint8)
    for j in range(23):
        fr[0, j + 1] = (c0 >> j) & 1
        fr[1, j] = (c1 >> j) & 1
    fr[0, 0] = bin(c0).count("1") & 1
    for j in range(11):
        fr[2, j] = (u2 >> j) & 1
    for j in range(14):
        fr[3, j] = (u3 >> j) & 1
    return fr


def random_imbe_bits(rng, shape, b0_max=207):
    """Uniform random parameter frames with a VALID fundamental (b0 <= b0_max) - SURVEY §8d C5."""
    bits = rng.integers(0, 2, size=tuple(shape) + (88,), dtype=np.uint8)
    b0 = rng.integers(0, b0_max + 1, size=shape)
    for k in range(6):
        bits[..., k] = (b0 >> (7 - k)) & 1
    bits[..., 85] = (b0 >> 1) & 1
    bits[..., 86] = b0 & 1
    return bits


def random_ambe_bits(rng, shape, b0_max=119):
    bits = rng.integers(0, 2, size=tuple(shape) + (49,), dtype=np.uint8)
    b0 = rng.integers(0, b0_max + 1, size=shape)
    for k, p in enumerate((0, 1, 2, 3, 37, 38, 39)):
        bits[..., p] = (b0 >> (6 - k)) & 1
    return bits


def load_kat():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_mbe_imbe7200_frames.json")) as f:
        kat = json.load(f)["frames"]
    for k in kat:
        rows = [int(r, 16) for r in k["rows"]]
        k["frame"] = np.array([[(rows[r] >> (22 - b)) & 1 for b in range(23)] for r in range(8)], np.uint8)
    return kat


def bits_hex(bits):
    v = 0
    for b in bits:
        v = (v << 1) | int(b)
    return "%0*X" % (len(bits) // 4, v)
