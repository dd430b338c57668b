"""Test helpers for the profile-driven 4-level FSK receive loop (P25p1 / DMR / NXDN48): profile construction shared by
the oracle wrapper (oracle/ddn_oracle_rx4.c) and the C-ABI tests, burst generators, and DMR burst field extraction.
TEST INFRASTRUCTURE - the product never imports this."""
import ctypes as C
import os
import re

import numpy as np

import orc

MAX_PAT, MAX_TAPS, PRE = 20, 135, 90

# sync words as the reference's dibit strings ('1' = +3, '3' = -3): include/dsd-neo/core/sync_patterns.h:33-34,58-80,
# NXDN FSW variants src/dsp/dsd_frame_sync.c:1508-1512
P25P1_SYNC = "111113113311333313133333"
DMR_BS_DATA, DMR_BS_VOICE = "313333111331131131331131", "131111333113313313113313"
DMR_MS_DATA, DMR_MS_VOICE = "311131133313133331131113", "133313311131311113313331"
DMR_DM1_DATA, DMR_DM1_VOICE = "331333313111313133311111", "113111131333131311133333"
DMR_DM2_DATA, DMR_DM2_VOICE = "311311111333113333133311", "133133333111331111311133"
NXDN_POS = ["3131331131", "3331331131", "3131331111", "3331331111", "3131311131"]
NXDN_NEG = ["1313113313", "1113113313", "1313113333", "1113113333", "1313133313"]

PROTO_P25P1, PROTO_DMR, PROTO_NXDN48, PROTO_NXDN96, PROTO_M17, PROTO_YSF = 0, 1, 2, 3, 4, 5
YSF_SYNC = "31111311313113131131"      # FUSION_SYNC, include/dsd-neo/core/sync_patterns.h:30-31; types = synctype_ids.h:109-110, + 1
T_YSF_POS, T_YSF_NEG = 31, 32
# sync type ids carried in lastsync (any non-zero numbering works; these mirror synctype_ids.h + 1 so 0 stays "none")
T_P25_POS, T_P25_NEG = 1, 2
T_DMR_BS_DATA, T_DMR_BS_VOICE, T_DMR_MS_VOICE, T_DMR_MS_DATA = 11, 13, 33, 34
T_NXDN_POS, T_NXDN_NEG = 29, 30
# M17: the twelve outcomes of frame_sync_try_m17() in the order the loops number them (flags pattern index / sync_pat)
M17_PRE_POS, M17_PRE_NEG, M17_EOT_POS, M17_EOT_NEG, M17_LSF_POS, M17_LSF_NEG = 0, 1, 2, 3, 4, 5
M17_BRT_POS, M17_BRT_NEG, M17_STR_POS, M17_STR_NEG, M17_PKT_POS, M17_PKT_NEG = 6, 7, 8, 9, 10, 11
M17_TYPES = [99, 100, 101, 102, 17, 18, 77, 78, 9, 10, 87, 88]      # synctype_ids.h:52-63, + 1
M17_WORDS = {"LSF": "11113313", "STR": "33331131", "PRE": "31313131", "PIV": "13131313", "BRT": "31331111", "PKT": "13113333",
             "EOT": "11111131", "EOT_INV": "33333313"}                # include/dsd-neo/core/sync_patterns.h:18-28
CLASS_DATA, CLASS_VOICE = 0, 1


def bits_of(pattern):
    v = 0
    for ch in pattern:
        v = (v << 1) | (1 if ch == "1" else 0)
    return v


class Profile(C.Structure):
    _fields_ = [("out_rate", C.c_int), ("sym_rate", C.c_int), ("rf_mod", C.c_int), ("win_len", C.c_int), ("t_max", C.c_int),
                ("warm_len", C.c_int), ("n_pat", C.c_int), ("pat_bits", C.c_uint32 * MAX_PAT), ("pat_type", C.c_uint8 * MAX_PAT),
                ("pat_neg", C.c_uint8 * MAX_PAT), ("pat_class", C.c_uint8 * MAX_PAT), ("confirm", C.c_int),
                ("live_thresholds", C.c_int), ("dmr_window", C.c_int), ("redigitize", C.c_int), ("slow_type", C.c_int),
                ("use_filter", C.c_int), ("nt", C.c_int), ("taps", C.c_uint32 * MAX_TAPS), ("lock_symbols", C.c_int * 4), ("handler", C.c_int), ("proto", C.c_int),
                ("m17", C.c_int)]


def _taps(name):
    """bit patterns of a generated tap table (oracle/ddn_tables_p25.h, oracle/ddn_tables_fsk4.h)"""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
    fn = "ddn_tables_p25.h" if name == "p25" else "ddn_tables_fsk4.h"
    txt = open(os.path.join(here, fn)).read()
    m = re.search(r"ddn_%s_filter_bits\[[A-Z0-9_]+\] = \{(.*?)\};" % name, txt, re.S)
    return [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]+u", m.group(1))]


def profile(proto, rf_mod=0, use_filter=1, lock=None, out_rate=48000, inverted=0, handler=0):
    p = Profile()
    p.handler, p.proto = handler, proto
    p.out_rate, p.rf_mod, p.use_filter = out_rate, rf_mod, use_filter
    pats = []
    if proto == PROTO_P25P1:
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 4800, 24, 24, 24
        p.live_thresholds, p.slow_type = 1, T_P25_NEG
        inv = "".join("1" if c == "3" else "3" for c in P25P1_SYNC)
        pats = [(P25P1_SYNC, T_P25_POS, 0, 0), (inv, T_P25_NEG, 1, 0)]
        taps = _taps("p25")
        lock = lock or [840, 0, 0, 0]
    elif proto == PROTO_DMR:
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 4800, 24, 24, 24
        p.dmr_window, p.redigitize = 1, 1
        pats = [(DMR_BS_DATA, T_DMR_BS_DATA, 0, CLASS_DATA), (DMR_BS_VOICE, T_DMR_BS_VOICE, 0, CLASS_VOICE),
                (DMR_MS_DATA, T_DMR_MS_DATA, 0, CLASS_DATA), (DMR_MS_VOICE, T_DMR_MS_VOICE, 0, CLASS_VOICE),
                (DMR_DM1_DATA, T_DMR_MS_DATA, 0, CLASS_DATA), (DMR_DM2_DATA, T_DMR_MS_DATA, 0, CLASS_DATA),
                (DMR_DM1_VOICE, T_DMR_MS_VOICE, 0, CLASS_VOICE), (DMR_DM2_VOICE, T_DMR_MS_VOICE, 0, CLASS_VOICE)]
        if inverted:   # opts->inverted_dmr (-xr): frame_sync_try_dmr_* swap the roles of the words; BS types become the NEG ones
            T_BS_VOICE_NEG, T_BS_DATA_NEG = 12, 14
            swap = {T_DMR_BS_DATA: (T_BS_VOICE_NEG, 1), T_DMR_BS_VOICE: (T_BS_DATA_NEG, 1), T_DMR_MS_DATA: (T_DMR_MS_VOICE, 0),
                    T_DMR_MS_VOICE: (T_DMR_MS_DATA, 0)}
            pats = [(s_, swap[t][0], swap[t][1], cl ^ 1) for (s_, t, neg, cl) in pats]
        taps = _taps("dmr")
        lock = lock or [120, 54 + 288 * 6, 0, 0]
    elif proto == PROTO_M17:
        # -fz: C4FM lock at 4800 symbols/s, no matched filter (decode_mode_apply_m17: use_cosine_filter = 0), 8-symbol words matched
        # with one error allowed by frame_sync_try_m17() (the pattern table only names the twelve outcomes: type, polarity, class);
        # class 0 = a frame or the EOT marker (184 dibits), class 1 = the preamble (skipDibit(8))
        p.m17 = 1
        p.proto = PROTO_P25P1       # (no handler family: fixed counts)
        p.use_filter = 0
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 4800, 8, 24, 8
        pats = [("11111111", t, k & 1, 1 if k < 2 else 0) for k, t in enumerate(M17_TYPES)]
        taps = _taps("dmr")         # (unused)
        lock = lock or [184, 8, 0, 0]
    elif proto == PROTO_YSF:
        # -fy: C4FM at 4800 symbols/s on the 4800_4 hunt profile, the 20-symbol FUSION_SYNC compared exactly in both polarities
        # (frame_sync_try_ysf(), src/dsp/dsd_frame_sync.c:770-797), 20-symbol warm start, the DMR matched filter once a YSF sync is the last
        # type (symbol_apply_matched_filter(), src/dsp/dsd_symbol.c:306-309); processYSF() reads the 100 FICH dibits and - for every frame
        # type but FI = 3 with DT != 1 - 360 more (src/protocol/ysf/ysf.c:668-686,725-740,836-851,906-918): a fixed 460 here
        p.proto = PROTO_P25P1       # (no handler family: fixed counts)
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 4800, 20, 24, 20
        inv = "".join("1" if c == "3" else "3" for c in YSF_SYNC)
        pats = [(YSF_SYNC, T_YSF_POS, 0, 0), (inv, T_YSF_NEG, 1, 0)]
        taps = _taps("dmr")
        lock = lock or [460, 0, 0, 0]
    elif proto == PROTO_NXDN96:
        # 4800 symbols/s on the 4800_4 hunt profile (level ring 24, src/dsp/dsd_frame_sync.c:1729-1744, matcher :1525-1556), the same
        # frame sync words, LICH gate and 182-symbol frame as NXDN48; the matched filter is the DMR one at every rate but 8 samples
        # per symbol (symbol_apply_matched_filter(), src/dsp/dsd_symbol.c:323-335)
        p.proto = PROTO_NXDN48      # (the handlers are NXDN's: the loop's protocol switch knows P25 / DMR / NXDN)
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 4800, 10, 24, 10
        p.confirm = 1
        pats = [(s, T_NXDN_POS, 0, 0) for s in NXDN_POS] + [(s, T_NXDN_NEG, 1, 0) for s in NXDN_NEG]
        taps = _taps("dmr")
        lock = lock or [182, 0, 0, 0]
    else:
        p.sym_rate, p.win_len, p.t_max, p.warm_len = 2400, 10, 12, 10
        p.confirm = 1
        pats = [(s, T_NXDN_POS, 0, 0) for s in NXDN_POS] + [(s, T_NXDN_NEG, 1, 0) for s in NXDN_NEG]
        # the reference tests positive[i] then negative[i] for i = 0..4; patterns are distinct so the order is immaterial
        taps = _taps("nxdn48")
        lock = lock or [182, 0, 0, 0]
    p.n_pat = len(pats)
    for k, (s, t, neg, cl) in enumerate(pats):
        assert len(s) == p.win_len
        p.pat_bits[k], p.pat_type[k], p.pat_neg[k], p.pat_class[k] = bits_of(s), t, neg, cl
    p.nt = len(taps)
    for k, t in enumerate(taps):
        p.taps[k] = t
    for k in range(4):
        p.lock_symbols[k] = lock[k]
    return p


class OracleFsk4Rx:
    def __init__(self, prof):
        o = orc.oracle()
        o.orc_fsk4rx_sizeof.restype = C.c_size_t
        o.orc_fsk4_profile_sizeof.restype = C.c_size_t
        assert o.orc_fsk4_profile_sizeof() == C.sizeof(Profile)
        o.orc_fsk4rx_init.argtypes = [C.c_void_p, C.c_void_p]
        o.orc_fsk4rx_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long] + [C.c_void_p] * 4 + [C.c_long] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        o.orc_fsk4rx_run.restype = C.c_long
        o.orc_fsk4rx_get_thresholds.argtypes = [C.c_void_p, C.c_void_p]
        self.o, self.prof = o, prof
        self.st = C.create_string_buffer(o.orc_fsk4rx_sizeof())
        o.orc_fsk4rx_init(self.st, C.byref(prof))
        self.events = orc.HEvents()
        o.orc_fsk4rx_set_events.argtypes = [C.c_void_p, C.c_void_p]
        o.orc_fsk4rx_set_events(self.st, C.byref(self.events))

    def run(self, x, max_sync=None):
        """-> dict(sym, rec4, fl, pay [k][2], sync_pos, sync_pat, pre [ns][90], pre_rel)"""
        x = np.ascontiguousarray(x, np.float32)
        cap = x.size // 2 + 8
        ms = max_sync or (x.size // 100 + 4)
        sym, rec, fl = np.zeros(cap, np.float32), np.zeros((cap, 4), np.int32), np.zeros(cap, np.uint8)
        pay = np.zeros((cap, 2), np.uint8)
        spos, spat = np.zeros(ms, np.int32), np.zeros(ms, np.uint8)
        pre, prel = np.zeros((ms, PRE), np.uint8), np.zeros((ms, PRE), np.uint8)
        ns = C.c_int(0)
        thr = np.zeros((ms, 5), np.float32)       # {center, umid, lmid, max, min} as each accepted sync leaves them
        self.o.orc_fsk4rx_set_sync_thresholds.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.o.orc_fsk4rx_set_sync_thresholds(self.st, thr.ctypes.data, ms)
        k = self.o.orc_fsk4rx_run(self.st, x.ctypes.data, x.size, sym.ctypes.data, rec.ctypes.data, fl.ctypes.data, pay.ctypes.data,
                                  cap, spos.ctypes.data, spat.ctypes.data, pre.ctypes.data, prel.ctypes.data, ms, C.byref(ns))
        assert k <= cap and ns.value <= ms
        n = ns.value
        return dict(sym=sym[:k].copy(), rec4=rec[:k].copy(), fl=fl[:k].copy(), pay=pay[:k].copy(), sync_pos=spos[:n].copy(),
                    sync_pat=spat[:n].copy(), pre=pre[:n].copy(), pre_rel=prel[:n].copy(), sync_thr=thr[:n].copy())

    def thresholds(self):
        t = np.zeros(7, np.float32)
        self.o.orc_fsk4rx_get_thresholds(self.st, t.ctypes.data)
        return t


# ---- DMR burst fields from the loop's outputs (what dmr_data_sync() assembles, src/protocol/dmr/dmr_data.c:117-262) ----------
CACH_IL = [0, 7, 8, 9, 1, 10, 11, 12, 2, 13, 14, 15, 3, 16, 4, 17, 18, 19, 5, 20, 21, 22, 6, 23]


def dmr_burst_fields(pre90, live54, inverted):
    """-> (slot_type bits [20], info bits [196], cach bits [24]) from the 90 cached + 54 live payload dibits"""
    d = [int(x) ^ (2 if inverted else 0) for x in pre90] + [int(x) for x in live54]
    hi = [(x >> 1) & 1 for x in d]
    lo = [x & 1 for x in d]
    cach = [0] * 24
    for i in range(12):
        cach[CACH_IL[2 * i]], cach[CACH_IL[2 * i + 1]] = hi[i], lo[i]
    info, st = [], []
    for i in range(12, 61):
        info += [hi[i], lo[i]]
    for i in range(61, 66):
        st += [hi[i], lo[i]]
    for i in range(90, 95):
        st += [hi[i], lo[i]]
    for i in range(95, 144):
        info += [hi[i], lo[i]]
    return np.array(st, np.uint8), np.array(info, np.uint8), np.array(cach, np.uint8)


def crc_ccitt_bits(bits):
    crc = 0
    for b in bits:
        msb = (crc >> 15) & 1
        crc = (crc << 1) & 0xFFFF
        if msb ^ int(b):
            crc ^= 0x1021
    return crc


def bits_int(bits):
    v = 0
    for b in bits:
        v = (v << 1) | int(b)
    return v


def capture_disc(name, lpf_profile):
    """discriminator stream of one of the reference's captures (tests/golden/iq_*.npz) through the pinned front end"""
    from conftest import golden
    g = golden(name)
    return orc.OracleFrontEnd(profile=lpf_profile).run_cu8(np.ascontiguousarray(g["iq"], np.uint8), 8192)


# ---- NXDN frame fields (what nxdn_frame() assembles: src/protocol/nxdn/nxdn_frame.c:181-199,311-331, nxdn_descramble.c,
# ---- nxdn_deperm.c:123-172) and the CRCs of its decoded fields ---------------------------------------------------------------
def nxdn_pn9(n=182, seed=228):
    l, o = seed, []
    for _ in range(n):
        o.append(l & 1)
        b = ((l >> 4) ^ l) & 1
        l = (l >> 1) | (b << 8)
    return np.array(o, np.uint8)


def nxdn_frame_fields(dibits182, rel182):
    """-> (lich7, lich_parity_ok, sacch_sym [36][2], sacch_rel, facch_sym [2][96][2], facch_rel)"""
    d = np.asarray(dibits182, np.uint8) ^ (nxdn_pn9() << 1)
    lich = 0
    for i in range(8):
        lich |= int(d[i] >> 1) << (7 - i)
    par = ((lich >> 7) + (lich >> 6) + (lich >> 5) + (lich >> 4)) & 1
    if (lich >> 1) in (0x08, 0x4A, 0x48, 0x46):
        par = sum((lich >> k) for k in range(1, 8)) & 1
    bits = np.stack([d >> 1, d & 1], 1).reshape(-1)
    rel = np.repeat(np.asarray(rel182, np.uint8), 2)
    dep, depr = np.zeros(60, np.uint8), np.zeros(60, np.uint8)
    p125 = np.array([(i % 5) * 12 + i // 5 for i in range(60)])
    dep[p125], depr[p125] = bits[16:76], rel[16:76]
    ss, sr = [], []
    for p in range(0, 60, 10):
        for m in (0, 1, 2, 3, 4, None, 5, 6, 7, 8, 9, None):
            ss.append(0 if m is None else int(dep[p + m]) << 1)
            sr.append(0 if m is None else int(depr[p + m]))
    p169 = np.array([(i % 9) * 16 + i // 9 for i in range(144)])
    fs, fr = [], []
    for off in (76, 220):
        dep, depr = np.zeros(144, np.uint8), np.zeros(144, np.uint8)
        dep[p169], depr[p169] = bits[off:off + 144], rel[off:off + 144]
        for i in range(0, 144, 3):
            fs += [int(dep[i]) << 1, 0, int(dep[i + 1]) << 1, int(dep[i + 2]) << 1]
            fr += [int(depr[i]), 0, int(depr[i + 1]), int(depr[i + 2])]
    return (lich >> 1, (lich & 1) == par, np.array(ss, np.uint8).reshape(36, 2), np.array(sr, np.uint8).reshape(36, 2),
            np.array(fs, np.uint8).reshape(2, 96, 2), np.array(fr, np.uint8).reshape(2, 96, 2))


def nxdn_crc_ok(bits, kind):
    """kind 0: 26 + CRC6 (x^6+x^5+x^2+x+1), kind 1: 80 + CRC12 (x^12+x^11+x^3+x^2+x+1); registers start all ones"""
    nd, nc, poly = (26, 6, 0x27) if kind == 0 else (80, 12, 0x80F)
    crc, mask, top = (1 << nc) - 1, (1 << nc) - 1, 1 << (nc - 1)
    for b in bits[:nd]:
        fb = (1 if crc & top else 0) ^ int(b)
        crc = (crc << 1) & mask
        if fb:
            crc ^= poly
    return crc == bits_int(bits[nd:nd + nc])


def nxdn_superframes(sacch_rows):
    """sacch_rows: decoded 32-bit SACCH fields (bit arrays) that passed their CRC, in order -> list of (ran, 72-bit message)"""
    out, cur = [], []
    for t in sacch_rows:
        sf = int(t[0]) * 2 + int(t[1])
        if sf == 3:
            cur = [t]
        elif cur and len(cur) == 3 - sf:
            cur.append(t)
        else:
            cur = []
        if len(cur) == 4:
            out.append((bits_int(cur[0][2:8]), np.concatenate([c[8:26] for c in cur])))
            cur = []
    return out


def oracle_trellis_decode(source_bits, result_len):
    """rows of 0/1 bits (>= 2 * result_len + 6 each) -> [n][result_len] bits (oracle/ddn_oracle_fec3.c)"""
    o = orc.oracle()
    o.orc_trellis_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    o.orc_trellis_decode.restype = None
    src = np.ascontiguousarray(source_bits, np.uint8)
    assert src.shape[1] >= 2 * result_len + 6
    out = np.zeros((src.shape[0], result_len), np.uint8)
    for i in range(src.shape[0]):
        row = np.ascontiguousarray(src[i])
        o.orc_trellis_decode(out[i].ctypes.data, row.ctypes.data, result_len)
    return out


# ---- AMBE 3600x2450 voice frames (include/dsd-neo/core/ambe_interleave.h; dmr_bs.c:137-160, nxdn_voice.c:57-74) -------------
def ambe2450_map():
    """the generated schedule (oracle/ddn_tables_ambe.h, measured from the compiled reference): [36][4] = high row, high col,
    low row, low col"""
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "ddn_tables_ambe.h")).read()
    rows = re.findall(r"\{(\d+), (\d+), (\d+), (\d+)\}", txt)
    assert len(rows) == 36
    return np.array(rows, np.int32)


def ambe2450_deinterleave(dibits36, rel36=None):
    """-> (ambe_fr uint8 [4][24], reliabilities uint8 [4][24])"""
    m = ambe2450_map()
    fr, rl = np.zeros((4, 24), np.uint8), np.zeros((4, 24), np.uint8)
    for i in range(36):
        d = int(dibits36[i]) & 3
        r = 0 if rel36 is None else int(rel36[i])
        fr[m[i, 0], m[i, 1]], fr[m[i, 2], m[i, 3]] = d >> 1, d & 1
        rl[m[i, 0], m[i, 1]] = rl[m[i, 2], m[i, 3]] = r
    return fr, rl


def nxdn_voice_frames(dibits182, rel182):
    """the four voice frames behind LICH + SACCH of one NXDN frame (de-scrambled) -> ([4][4][24], [4][4][24])"""
    d = np.asarray(dibits182, np.uint8) ^ (nxdn_pn9() << 1)
    out = [ambe2450_deinterleave(d[38 + 36 * v:74 + 36 * v], np.asarray(rel182)[38 + 36 * v:74 + 36 * v]) for v in range(4)]
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])


def nxdn_lich_voice(lich7):
    """which voice frames a LICH announces (nxdn_frame.c:117-150): 3 = all four, 1 = the first two, 2 = the last two, 0 none"""
    if lich7 in (0x36, 0x37, 0x56, 0x57, 0x46, 0x76, 0x77):
        return 3
    if lich7 in (0x34, 0x35, 0x54, 0x55, 0x75):
        return 1
    if lich7 in (0x32, 0x33, 0x52, 0x53, 0x72, 0x73):
        return 2
    return 0


def dmr_voice_burst_fields(dibits144, rel144, inverted):
    """-> (ambe_fr [3][4][24], reliabilities [3][4][24], sync / EMB bits [48], cach bits [24]) of one voice burst"""
    d = np.array([(int(x) ^ (2 if inverted else 0)) & 3 for x in dibits144], np.uint8)
    r = np.asarray(rel144, np.uint8)
    f1 = ambe2450_deinterleave(d[12:48], r[12:48])
    f2 = ambe2450_deinterleave(np.concatenate([d[48:66], d[90:108]]), np.concatenate([r[48:66], r[90:108]]))
    f3 = ambe2450_deinterleave(d[108:144], r[108:144])
    sync = np.stack([d[66:90] >> 1, d[66:90] & 1], 1).reshape(-1)
    cach = np.zeros(24, np.uint8)
    for i in range(12):
        cach[CACH_IL[2 * i]], cach[CACH_IL[2 * i + 1]] = d[i] >> 1, d[i] & 1
    return np.stack([f1[0], f2[0], f3[0]]), np.stack([f1[1], f2[1], f3[1]]), sync, cach
