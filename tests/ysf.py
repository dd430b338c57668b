"""Yaesu System Fusion helpers of the tests: the frame information channel through the CPU restatement (oracle/ddn_oracle_ysf.c)."""
import ctypes as C

import numpy as np

import orc


def fich(dibits100):
    """-> (err, 32 FICH bits, Viterbi path cost)"""
    o = orc.oracle()
    o.orc_ysf_fich.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_ysf_fich.restype = C.c_int
    d = np.ascontiguousarray(dibits100, np.uint8)
    out = np.zeros(32, np.uint8)
    ve = C.c_uint32()
    err = o.orc_ysf_fich(d.ctypes.data, out.ctypes.data, C.byref(ve))
    return int(err), out, int(ve.value)


def fields(bits32):
    """ysf_parse_fich(), src/protocol/ysf/ysf.c:534-546"""
    v = lambda a, n: int("".join(str(int(b)) for b in bits32[a:a + n]), 2)
    return dict(fi=v(0, 2), cm=v(4, 2), bn=v(6, 2), bt=v(8, 2), fn=v(10, 3), ft=v(13, 3), mr=v(18, 3), vp=int(bits32[21]), dt=v(22, 2),
                st=int(bits32[24]), sc=v(25, 7))


def summary(f):
    """ysf_print_fich_type / _call_mode / _path_and_frame (:562-619) for a frame without errors"""
    return "%s %s %s %s" % (["V/D1", "DATA", "V/D2", "VWFR"][f["dt"]], ["Group/CQ", "RID Mode", "Res: 2", "Private"][f["cm"]],
                            ["-Simplex", "Repeater"][f["vp"]], ["HC", "CC", "TC", "XX"][f["fi"]])


def decode_frames(out):
    """out = the loop's output (oracle or device arrays: rec4, sync_pos) -> per sync with its 100 FICH dibits inside: dict(pos, err,
    bits, cost, fields)"""
    n = len(out["rec4"])
    fr = []
    for pos in out["sync_pos"]:
        pos = int(pos)
        if pos + 101 > n:
            continue
        err, bits, cost = fich(out["rec4"][pos + 1:pos + 101, 0])
        fr.append(dict(pos=pos, err=err, bits=bits, cost=cost, fields=fields(bits)))
    return fr


def payload(dibits360, fi, dt, csd3=False):
    """orc_ysf_payload + orc_ysf_voice_frames -> dict(kind, dch[2][20], dch_status[2], dch_cost[2], ambe_d[5][49], errs2[5], frames[5][184],
    n_frames); csd3: the frame's own FICH is good and says FT = 1, FN = 0 (only looked at for full-rate voice)"""
    o = orc.oracle()
    o.orc_ysf_payload.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    o.orc_ysf_payload.restype = C.c_int
    p = np.ascontiguousarray(dibits360, np.uint8)
    dch, st, cost = np.zeros((2, 20), np.uint8), np.zeros(2, np.uint8), np.zeros(2, np.uint32)
    ambe, errs = np.zeros((5, 49), np.uint8), np.zeros(5, np.uint8)
    kind = o.orc_ysf_payload(p.ctypes.data, int(fi), int(dt), dch.ctypes.data, st.ctypes.data, cost.ctypes.data, ambe.ctypes.data,
                             errs.ctypes.data)
    o.orc_ysf_voice_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_ysf_voice_frames.restype = C.c_int
    fr = np.zeros((5, 184), np.uint8)
    d20, st1, c1 = np.zeros(20, np.uint8), C.c_uint8(0), C.c_uint32(0)
    nf = o.orc_ysf_voice_frames(p.ctypes.data, int(kind), int(bool(csd3) and kind == 4), fr.ctypes.data, d20.ctypes.data, C.byref(st1), C.byref(c1))
    if kind == 4 and csd3:
        dch[0], st[0], cost[0] = d20, st1.value, c1.value
    return dict(kind=int(kind), dch=dch, dch_status=st, dch_cost=cost, ambe_d=ambe, errs2=errs, frames=fr, n_frames=int(nf),
                csd3=bool(csd3) and kind == 4)


def decode_payloads(out, last=(0, 0), n_rec=None):
    """processYSF() over a stream (ysf.c:924-938): every sync whose FICH lies inside the records is a frame; a frame whose FICH failed is
    read as the last good frame's DT / FI (ysf_parse_fich's statics, :553-555); the payload is decoded when its 360 dibits lie inside
    the records.  -> (list of dict(pos, err, fi, dt, payload | None), last)"""
    n = len(out["rec4"]) if n_rec is None else n_rec
    last_dt, last_fi = last
    fr = []
    for pos in out["sync_pos"]:
        pos = int(pos)
        if pos + 101 > n:
            continue
        err, bits, cost = fich(out["rec4"][pos + 1:pos + 101, 0])
        csd3 = False
        if err == 0:
            f = fields(bits)
            last_dt, last_fi = f["dt"], f["fi"]
            csd3 = f["ft"] == 1 and f["fn"] == 0
        dt, fi = last_dt, last_fi
        pl = payload(out["rec4"][pos + 101:pos + 461, 0], fi, dt, csd3) if pos + 461 <= n else None
        fr.append(dict(pos=pos, err=err, fi=fi, dt=dt, payload=pl))
    return fr, (last_dt, last_fi)
