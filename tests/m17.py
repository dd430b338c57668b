"""Test helpers for M17 (the consumers of the K = 5 decoders): wrappers of the restatement
(oracle/ddn_oracle_m17.c), frame construction with the REFERENCE's own encoder (oracle/_ref: m17_algorithms.c compiled where it lies)
and the whole-stream decode the chain is checked against.  TEST INFRASTRUCTURE - the product never imports this."""
import ctypes as C

import numpy as np

import orc
import rx4

SYNC_LSF, SYNC_STR, SYNC_PKT, SYNC_BRT, PREAMBLE, EOT = 0x55F7, 0xFF5D, 0x75FF, 0xDF55, 0x7777, 0x555D      # M17 specification
B40 = " ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/."


def _o():
    o = orc.oracle()
    o.orc_m17_crc16.restype = C.c_uint16
    o.orc_m17_crc16.argtypes = [C.c_void_p, C.c_int]
    o.orc_m17_soft_cost.restype = C.c_uint16
    o.orc_m17_soft_cost.argtypes = [C.c_float, C.c_void_p, C.c_int]
    o.orc_m17_lsf_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_m17_lsf_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_m17_payload_bits.argtypes = [C.c_void_p, C.c_void_p]
    o.orc_m17_str_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_m17_callsign.argtypes = [C.c_uint64, C.c_char_p]
    return o


def crc16(by):
    b = np.ascontiguousarray(by, np.uint8)
    return int(_o().orc_m17_crc16(b.ctypes.data, len(b)))


def lsf_costs(sym184, thr5):
    s, t = np.ascontiguousarray(sym184, np.float32), np.ascontiguousarray(thr5, np.float32)
    out = np.zeros(488, np.uint16)
    _o().orc_m17_lsf_costs(s.ctypes.data, t.ctypes.data, out.ctypes.data)
    return out


def lsf_decode(cost488):
    c = np.ascontiguousarray(cost488, np.uint16)
    lsf, pc = np.zeros(30, np.uint8), C.c_uint32(0)
    ok = _o().orc_m17_lsf_decode(c.ctypes.data, lsf.ctypes.data, C.byref(pc))
    return lsf, int(ok), int(pc.value)


def str_decode(dibits184):
    d = np.ascontiguousarray(dibits184, np.uint8)
    lich, cnt, fp = np.zeros(6, np.uint8), C.c_int(0), np.zeros(18, np.uint8)
    err = _o().orc_m17_str_decode(d.ctypes.data, lich.ctypes.data, C.byref(cnt), fp.ctypes.data)
    return int(err), lich, int(cnt.value), fp


def callsign(addr):
    buf = C.create_string_buffer(10)
    rc = _o().orc_m17_callsign(int(addr), buf)
    return rc, buf.value.decode()


def encode_callsign(text):
    v = 0
    for ch in reversed(text):
        v = v * 40 + B40.index(ch)
    return v


# ---- frames built by the reference's encoder ------------------------------------------------------------------------------------
def _r():
    r = orc.ref()
    r.Golay_24_12_init()                    # (the reference builds its syndrome tables at start-up: InitAllFecFunction())
    r.m17_crc16.restype = C.c_uint16
    r.m17_crc16.argtypes = [C.c_void_p, C.c_uint16]
    r.m17_lsf_encode_type1_bits.restype = C.c_uint16
    r.m17_lsf_encode_type1_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    r.m17_stream_build_type1_bits.argtypes = [C.c_uint16, C.c_void_p, C.c_void_p]
    r.m17_stream_encode_type1_bits.argtypes = [C.c_void_p, C.c_void_p]
    r.m17_lich_build_content.argtypes = [C.c_void_p, C.c_uint8, C.c_void_p]
    r.m17_lich_encode_bits.argtypes = [C.c_void_p, C.c_void_p]
    r.m17_stream_combine_frame_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    r.m17_payload_encode_bits.argtypes = [C.c_void_p, C.c_void_p]
    r.m17_payload_decode_bits.argtypes = [C.c_void_p, C.c_void_p]
    r.m17_frame_build_dibits.argtypes = [C.c_uint16, C.c_void_p, C.c_void_p]
    r.m17_fill_repeating_16bit_dibits.argtypes = [C.c_uint16, C.c_void_p]
    r.m17_lich_decode_bits.argtypes = [C.c_void_p, C.c_void_p]
    r.m17_address_decode_csd.argtypes = [C.c_ulonglong, C.c_char_p]
    return r


def lsf_bits(dst, src, type_word=0x0005, meta=None, good_crc=True):
    """240 LSF bits: DST 48, SRC 48, TYPE 16, META 112, CRC16 (m17_parse_lsf's layout, m17_parse.c:369-420)"""
    by = np.zeros(30, np.uint8)
    by[0:6] = list(int(dst).to_bytes(6, "big"))
    by[6:12] = list(int(src).to_bytes(6, "big"))
    by[12:14] = [type_word >> 8, type_word & 0xFF]
    if meta is not None:
        by[14:28] = meta
    c = int(_r().m17_crc16(by.ctypes.data, 28)) ^ (0 if good_crc else 0x0101)
    by[28:30] = [c >> 8, c & 0xFF]
    return np.unpackbits(by), by


def lsf_frame(bits240):
    """-> 192 dibits: LSF sync word + the reference's m17_lsf_encode_type1_bits (K = 5 encoder, P1, interleave, randomise)"""
    r = _r()
    t1 = np.zeros(244, np.uint8)
    t1[:240] = bits240
    rnd = np.zeros(368, np.uint8)
    r.m17_lsf_encode_type1_bits(t1.ctypes.data, rnd.ctypes.data, None)
    fr = np.zeros(192, np.uint8)
    r.m17_frame_build_dibits(SYNC_LSF, rnd.ctypes.data, fr.ctypes.data)
    return fr


def stream_frame(bits240, lich_cnt, fn, payload16):
    r = _r()
    content, lich = np.zeros(48, np.uint8), np.zeros(96, np.uint8)
    lsf = np.ascontiguousarray(bits240, np.uint8)
    assert r.m17_lich_build_content(lsf.ctypes.data, lich_cnt, content.ctypes.data) == 0
    r.m17_lich_encode_bits(content.ctypes.data, lich.ctypes.data)
    t1, pb = np.zeros(148, np.uint8), np.unpackbits(np.asarray(payload16, np.uint8))
    r.m17_stream_build_type1_bits(fn, pb.ctypes.data, t1.ctypes.data)
    punc = np.zeros(272, np.uint8)
    r.m17_stream_encode_type1_bits(t1.ctypes.data, punc.ctypes.data)
    comb, rnd = np.zeros(368, np.uint8), np.zeros(368, np.uint8)
    r.m17_stream_combine_frame_bits(lich.ctypes.data, punc.ctypes.data, comb.ctypes.data)
    r.m17_payload_encode_bits(comb.ctypes.data, rnd.ctypes.data)
    fr = np.zeros(192, np.uint8)
    r.m17_frame_build_dibits(SYNC_STR, rnd.ctypes.data, fr.ctypes.data)
    return fr


def repeating(word):
    fr = np.zeros(192, np.uint8)
    _r().m17_fill_repeating_16bit_dibits(word, fr.ctypes.data)
    return fr


def transmission(rng, dst, src, n_frames, preamble_syms=192):
    """preamble, LSF, n stream frames (LICH chunks 0..5 round and round, random payloads, end flag on the last), EOT ->
    (dibits int8, LSF bytes, [(fn, payload16)])"""
    bits, by = lsf_bits(dst, src)
    parts = [repeating(PREAMBLE)[:preamble_syms], lsf_frame(bits)]
    sent = []
    for k in range(n_frames):
        pay = rng.integers(0, 256, 16).astype(np.uint8)
        fn = k | (0x8000 if k == n_frames - 1 else 0)
        parts.append(stream_frame(bits, k % 6, fn, pay))
        sent.append((fn, pay))
    parts.append(repeating(EOT))
    return np.concatenate(parts).astype(np.int8), by, sent


# ---- the whole-stream decode the loop's output goes through (what dsd_dispatch_handle_m17 hands to the handlers) ------------------------
def decode_stream(out):
    """out = OracleFsk4Rx.run() of the M17 profile -> list of dicts per accepted sync that starts a frame inside the output:
    {pos, pat, kind: 'lsf' | 'str' | 'pkt' | 'brt' | 'pre' | 'eot', ...}; LSF frames carry lsf30 / crc_ok / cost, stream frames
    lich_err / lich6 / cnt / fn / payload, and - when a chunk counter of 5 completes it - the LSF reassembled from the LICH chunks"""
    frames = []
    asm = np.zeros(240, np.uint8)
    n = len(out["sym"])
    for k, (pos, pat) in enumerate(zip(out["sync_pos"], out["sync_pat"])):
        pos, pat = int(pos), int(pat)
        f = dict(pos=pos, pat=pat)
        if pat in (rx4.M17_PRE_POS, rx4.M17_PRE_NEG):
            f["kind"] = "pre"
        elif pat in (rx4.M17_EOT_POS, rx4.M17_EOT_NEG):
            f["kind"] = "eot"
            asm[:] = 0                                                 # dispatch_m17.c:39: state->m17_lsf cleared
        elif pos + 185 > n:
            f["kind"] = "cut"
        elif pat in (rx4.M17_LSF_POS, rx4.M17_LSF_NEG):
            f["kind"] = "lsf"
            cost = lsf_costs(out["sym"][pos + 1:pos + 185], out["sync_thr"][k])
            f["cost488"] = cost
            f["lsf30"], f["crc_ok"], f["cost"] = lsf_decode(cost)
            asm[:] = np.unpackbits(f["lsf30"])                         # m17_decode_lsf_soft_bits: state->m17_lsf = the decoded LSF
        elif pat in (rx4.M17_STR_POS, rx4.M17_STR_NEG):
            f["kind"] = "str"
            err, lich, cnt, fp = str_decode(out["rec4"][pos + 1:pos + 185, 0])
            f.update(lich_err=err, lich6=lich, cnt=cnt, fn=(int(fp[0]) << 8) | int(fp[1]), payload=fp[2:].copy())
            if err == 0:
                asm[40 * cnt:40 * cnt + 40] = np.unpackbits(lich)[:40]
                if cnt == 5:                                           # M17finalizeLICH: CRC over the reassembled LSF
                    by = np.packbits(asm)
                    f["lich_lsf30"] = by
                    f["lich_crc_ok"] = int(crc16(by[:28]) == ((int(by[28]) << 8) | int(by[29])))
                    asm[:] = 0                                         # (:250)
        else:
            f["kind"] = "pkt" if pat in (rx4.M17_PKT_POS, rx4.M17_PKT_NEG) else "brt"
        frames.append(f)
    return frames
