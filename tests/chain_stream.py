"""The P25 Phase 1 chain over a whole stream on the CPU, with the reference's handlers inside the receive loop: one channel of cu8
I/Q, handed over in calls of a fixed size, through the oracles (front end per call -> receive loop with per-DUID in-frame lengths
-> NID BCH + Chase -> TSDU blocks 0..2 list decode + CRC16 selection -> LDU voice frames -> IMBE frame decode -> synthesis).
Frames are keyed by the stream position of their sync's last symbol, so a chain that decodes them in any split of the stream into
calls can be compared.  TEST INFRASTRUCTURE: the checker of tests/test_chain_gpu.py and of bench.py's parity gate."""
import ctypes as C

import numpy as np

import ddn
import mbe
import orc
import p25gen

_NID_KEEP = [k for k in range(33) if k != 11]
_L = None


def layout():
    """offsets from the sync's LAST symbol (+1 = first NID dibit)"""
    global _L
    if _L is None:
        l = ddn.lib()
        first9, st9 = np.zeros(9, np.int32), np.zeros(9, np.int32)
        l.ddn_p25p1_layout_ldu_imbe(first9.ctypes.data, st9.ctypes.data)
        blocks = []
        for b in range(3):
            o = np.zeros(98, np.int32)
            l.ddn_p25p1_layout_trellis_block(b, o.ctypes.data)
            blocks.append(o - 23)
        _L = (first9, st9, blocks)
    return _L


def oracle_nid(bits, rel, obs, par, prel):
    from test_oracle_block import oracle_nid as f
    return f(bits, rel, obs, par, prel)


def ref_nid(bits, rel, obs, par, prel):
    """the compiled reference's p25p1_nid_decode (oracle/_ref), same calling convention as oracle_nid"""
    r = orc.ref()
    r.refh_nid_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros((len(bits), 4), np.int32)
    for i in range(len(bits)):
        r.refh_nid_decode(bits[i].ctypes.data, rel[i].ctypes.data, int(obs[i]), int(par[i]), int(prel[i]), out[i].ctypes.data)
    return out


def _select(ob, no):
    for k in range(no):
        if p25gen.crc16_ccitt(ob[k][:10]) == ((int(ob[k][10]) << 8) | int(ob[k][11])):
            return ob[k].copy(), 1, k
    return (ob[0].copy() if no > 0 else np.zeros(12, np.uint8)), 0, 0


def oracle_tsbk(llr196):
    """tsbk_decode_repetition_bytes(): list of 8, first CRC16-clean candidate, else the best -> (bytes12, crc_ok, sel)"""
    o = orc.oracle()
    o.orc_p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    llr = np.ascontiguousarray(llr196, np.int16)
    ob = np.zeros((8, 12), np.uint8)
    om = np.zeros(8, np.uint32)
    no = o.orc_p25_12_soft_llr_list(llr.ctypes.data, ob.ctypes.data, om.ctypes.data, 8)
    return _select(ob, no)


def ref_tsbk(llr196):
    """the same through the compiled reference's p25_12_soft_llr_list"""
    r = orc.ref()
    r.p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    llr = np.ascontiguousarray(llr196, np.int16)
    cand = np.zeros((8, 16), np.uint8)
    no = r.p25_12_soft_llr_list(None, llr.ctypes.data, cand.ctypes.data, 8)
    return _select(cand[:, :12], no)


def run_stream(iq_c, samples_per_call, seed, vocoder=True, timers=None, use_ref=False):
    """iq_c u8 [n_total][2] -> dict(sym, rec4, fl, events [(pos, kind, a, b|c<<16)], frames {pos: {...}}, voice [(pos, v, bits,
    res, skip, pcm)]).  timers (dict): seconds inside the C calls of each stage are added there (Python glue excluded).  use_ref:
    the stages the compiled reference (oracle/_ref) holds - front end (single call only), NID decode, half-rate list decode -
    run through it instead of the restatement (bench.py's cpu_baseline)"""
    import time
    first9, st9, blocks = layout()
    T = timers if timers is not None else {}

    def timed(key, fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        T[key] = T.get(key, 0.0) + (time.perf_counter() - t0)
        return r

    use_ref = use_ref and orc.have_ref()
    if use_ref and len(iq_c) <= samples_per_call:
        disc = timed("front_end", orc.ref_front_end_cu8, iq_c, 8192)[0]
    else:
        fe = orc.OracleFrontEnd()
        disc = np.concatenate([timed("front_end", fe.run_cu8, iq_c[a:a + samples_per_call], 8192)
                               for a in range(0, len(iq_c), samples_per_call)])
    deint = orc.oracle_imbe_deinterleave
    rx = orc.OracleP25Rx(lock_symbols=-1, use_filter=1)
    sym, rec4, fl = timed("rx", rx.run, disc)
    cnt = len(sym)
    rows, evd = rx.events.rows(), rx.events.data()
    events = np.array([[e[0], e[1], e[2], (e[3] & 0xFFFF) | ((e[4] & 0xFFFF) << 16)] for e in rows], np.int64).reshape(-1, 4)
    acc = np.flatnonzero(fl[:cnt] & 2)
    # the NID and the TSDU blocks of a frame are what the loop's handlers decoded (orc_hevent.data), filed by the frame's sync
    frames = {int(a): {"complete": int(a) + 842 <= cnt} for a in acc}
    for e, d in zip(rows, evd):
        if e[1] == orc.HEV_P25_NID and e[0] - 33 in frames:
            frames[e[0] - 33]["nid"] = d.copy()
        elif e[1] == orc.HEV_P25_TSBK:
            k = (int(d[3]) >> 16) & 0xFF
            a = e[0] - (33 + 101 * (k + 1))     # the handler reads 101 symbols per block (the third ends on a status symbol)
            if a in frames:
                frames[a]["tsbk%d" % k] = (d[:3].copy().view(np.uint8), int(d[3]) & 1, (int(d[3]) >> 8) & 0xFF)
    if timers is not None:
        # cpu_baseline bookkeeping: the handlers' NID / half-rate decodes sit inside "rx" (restatement).  Their share is measured by
        # repeating them outside the loop - restatement, and compiled reference where present - so that bench.py can report the
        # loop without them and the two FEC stages by the reference's own code: rx_fec_port is SUBTRACTED from rx there.
        for a in acc:
            a = int(a)
            if a + 34 <= cnt:
                nd = rec4[a + 1:a + 34][_NID_KEEP]
                bb = np.stack([(nd[:, 0] >> 1) & 1, nd[:, 0] & 1], axis=1).reshape(64).astype(np.uint8)
                rr = np.minimum(np.abs(np.stack([nd[:, 2], nd[:, 3]], axis=1)), 255).reshape(64).astype(np.uint8)
                args = (bb[None, :63].copy(), rr[None, :63].copy(), np.zeros(1, np.int32), bb[63:64].copy(), rr[63:64].copy())
                timed("rx_fec_port", oracle_nid, *args)
                if use_ref:
                    timed("nid", ref_nid, *args)
            for k, off in enumerate(blocks):
                if "tsbk%d" % k in frames[a]:
                    llr = np.stack([rec4[a + off, 2], rec4[a + off, 3]], axis=1).reshape(196)
                    timed("rx_fec_port", oracle_tsbk, llr)
                    if use_ref:
                        got = timed("trellis", ref_tsbk, llr)
                        assert np.array_equal(got[0], frames[a]["tsbk%d" % k][0]) and got[1] == frames[a]["tsbk%d" % k][1]
    voice = []
    if vocoder:
        voc = mbe.OracleVocoder(ddn.MBE_IMBE, 1, tail_rule=1)
        for a in acc:
            a = int(a)
            n = frames[a].get("nid")
            if n is None or n[0] <= 0 or n[2] not in (5, 10):     # NID_OK and NID_PARITY_OVERRIDE both reach processLDU1 / 2 (dispatch_p25p1.c:214-218)
                continue
            for v in range(9):
                s0 = a - 23 + int(first9[v])
                if s0 + 76 > cnt:       # runs past the stream's end: the device leaves such a frame out
                    voice.append((a, v, None, None, True, np.zeros(160, np.float32)))
                    continue
                d = rec4[s0:s0 + 76]
                fr, _, flag, _, _ = timed("imbe_deint", deint, d[:, 0].astype(np.uint8), d[:, 2].astype(np.int16),
                                          d[:, 3].astype(np.int16), int(st9[v]))
                bits, res, rc = timed("imbe_fec", mbe.oracle_frame_decode, ddn.MBE_IMBE, fr[None])
                assert rc[0] == 0
                pcm = np.zeros((1, 160), np.float32)
                if flag == 0:
                    lb, lr = np.ascontiguousarray(bits), np.ascontiguousarray(res)
                    rc = timed("mbe_synth", mbe._o().om_process_batch, ddn.MBE_IMBE, C.addressof(voc.tab), lb.ctypes.data,
                               lr.ctypes.data, 1, int(seed), 1, 1, pcm.ctypes.data, None, C.addressof(voc.cur), C.addressof(voc.prev),
                               C.addressof(voc.enh))
                    assert rc == 0
                voice.append((a, v, bits[0], res[0], flag != 0, pcm[0]))
    return dict(sym=sym, rec4=rec4, fl=fl, events=events, event_data=evd, frames=frames, voice=voice)


class Collector:
    """gathers what a ddn.P25ChainC produced call by call (and in its flush), keyed like run_stream(); `channels` = the channels
    to follow (default all), only their rows are copied back"""

    def __init__(self, chain, everything=False, channels=None):
        self.ch = chain
        self.chans = list(range(chain.B)) if channels is None else list(channels)
        n = len(self.chans)
        self.base = np.zeros(n, np.int64)          # stream index of the first new record of the current call
        self.rec = [[] for _ in range(n)]
        self.fl = [[] for _ in range(n)]
        self.events = [[] for _ in range(n)]
        self.event_data = [[] for _ in range(n)]
        self.frames = [dict() for _ in range(n)]
        self.voice = [[] for _ in range(n)]
        self.everything = everything

    def _rows(self, ptr, dtype, row_shape, lead=1):
        """array [lead][B][row_shape] on the device -> [lead][len(chans)][row_shape] (lead = 1 is squeezed)"""
        ch = self.ch
        row = int(np.prod(row_shape)) * np.dtype(dtype).itemsize
        out = np.zeros((lead, len(self.chans)) + tuple(row_shape), dtype)
        for b in range(lead):
            for i, c in enumerate(self.chans):
                out[b, i] = ch.fetch(int(ptr) + (b * ch.B + c) * row, dtype, row_shape)
        return out[0] if lead == 1 else out

    def take(self):
        ch = self.ch
        F, Fv, T, E, st = ch.F, ch.Fv, ch.T, ch.E, ch.stride
        r = ch.results()
        f = self._rows
        rec = f(r.d_records10, np.uint8, (st, 10))
        fl = f(r.d_flags, np.uint8, (st,))
        new = f(r.d_new, np.int32, (1,))[:, 0]
        ev = f(r.d_events, np.int32, (E, 4))
        nev = f(r.d_n_events, np.int32, (1,))[:, 0]
        evd = f(r.d_event_data, np.int32, (E, 4))
        ns = f(r.d_n_syncs, np.int32, (1,))[:, 0]
        pos = f(r.d_sync_pos, np.int32, (F,))
        nid = f(r.d_nid4, np.int32, (F, 4))
        tsbk = f(r.d_tsbk, np.uint8, (F, 12), lead=3)
        tcrc = f(r.d_tsbk_crc, np.uint8, (F,), lead=3)
        nldu = f(r.d_n_ldu, np.int32, (1,))[:, 0]
        bits = f(r.d_imbe_bits, np.uint8, (Fv * 9, 88))
        res = f(r.d_imbe_result, np.int32, (Fv * 9, 5))
        pcm = f(r.d_pcm, np.float32, (Fv * 9, 160))
        extra = {}
        if self.everything:
            extra = dict(words1=f(r.d_ldu_words[0], np.uint8, (F, 240)), words2=f(r.d_ldu_words[1], np.uint8, (F, 240)),
                         rs1=f(r.d_ldu_rs_data[0], np.uint8, (F, 72)), rs2=f(r.d_ldu_rs_data[1], np.uint8, (F, 96)),
                         rs1s=f(r.d_ldu_rs_status[0], np.uint8, (F,)), rs2s=f(r.d_ldu_rs_status[1], np.uint8, (F,)),
                         lsd=f(r.d_lsd_bits, np.uint8, (F, 32)), lsd_ok=f(r.d_lsd_ok, np.uint8, (F, 2)),
                         hdu=f(r.d_hdu_rs_data, np.uint8, (F, 120)), hdus=f(r.d_hdu_rs_status, np.uint8, (F,)),
                         tdulc=f(r.d_tdulc_rs_data, np.uint8, (F, 72)), tdulcs=f(r.d_tdulc_rs_status, np.uint8, (F,)))
        for c in range(len(self.chans)):
            k = int(new[c])
            self.rec[c].append(rec[c, T:T + k])
            self.fl[c].append(fl[c, T:T + k])
            e = ev[c, :nev[c]].astype(np.int64)
            e[:, 0] += self.base[c]
            self.events[c].append(e)
            self.event_data[c].append(evd[c, :nev[c]])
            assert ns[c] <= F, "frame slots exhausted"
            assert nldu[c] <= Fv
            kv = 0
            for s in range(int(ns[c])):
                g = int(self.base[c]) + int(pos[c, s]) - T
                assert g not in self.frames[c], ("sync decoded twice", c, g)
                d = dict(nid=nid[c, s].copy(), tsbk=[(tsbk[b, c, s].copy(), int(tcrc[b, c, s])) for b in range(3)])
                for name, a in extra.items():
                    d[name] = a[c, s].copy()
                self.frames[c][g] = d
                if nid[c, s, 0] > 0 and nid[c, s, 2] in (5, 10) and kv < Fv:
                    for v in range(9):
                        i = kv * 9 + v
                        self.voice[c].append((g, v, bits[c, i].copy(), res[c, i].copy(), pcm[c, i].copy()))
                    kv += 1
            assert kv == nldu[c], (c, kv, nldu[c])
            self.base[c] += k


def check_channel(col, i, want):
    """Collector channel slot i against run_stream()'s answer; raises AssertionError on the first difference -> (NIDs, TSBK blocks,
    voice frames compared)"""
    cnt = len(want["sym"])
    rec = np.concatenate(col.rec[i])
    fl = np.concatenate(col.fl[i])
    assert len(rec) == cnt, (i, len(rec), cnt)
    r4, sy = orc.unpack_records10(rec)
    assert np.array_equal(fl, want["fl"]) and np.array_equal(r4, want["rec4"]), i
    assert np.array_equal(sy.view(np.uint32), want["sym"].view(np.uint32)), i
    mask = np.array([-1, -1, -1, 0xFFFFFFFF])
    ev = np.concatenate(col.events[i]) & mask
    assert np.array_equal(ev, want["events"] & mask), i
    assert np.array_equal(np.concatenate(col.event_data[i]), want["event_data"]), i
    # every sync of the stream decoded exactly once
    assert sorted(col.frames[i]) == sorted(want["frames"]), (i, sorted(set(col.frames[i]) ^ set(want["frames"]))[:6])
    n_nid = n_tsbk = 0
    for g, w in want["frames"].items():
        d = col.frames[i][g]
        if "nid" in w:
            assert tuple(d["nid"]) == tuple(int(x) for x in w["nid"]), (i, g, d["nid"], w["nid"])
            n_nid += 1
        for b in range(3):
            if "tsbk%d" % b in w:
                by, ok, _ = w["tsbk%d" % b]
                assert np.array_equal(d["tsbk"][b][0], by) and d["tsbk"][b][1] == ok, (i, g, b)
                n_tsbk += 1
    # voice: the same frames in the same order, bits, result words and PCM
    wv = [v for v in want["voice"] if want["frames"][v[0]]["complete"]]
    gv = [v for v in col.voice[i] if want["frames"][v[0]]["complete"]]
    assert [(v[0], v[1]) for v in gv] == [(v[0], v[1]) for v in wv], i
    for g_, w_ in zip(gv, wv):
        if w_[4]:
            assert not g_[4].any(), (i, g_[0], g_[1])
            continue
        assert np.array_equal(g_[2], w_[2]) and np.array_equal(g_[3], w_[3]), (i, g_[0], g_[1])
        assert np.array_equal(g_[4].view(np.uint32), w_[5].view(np.uint32)), (i, g_[0], g_[1], np.abs(g_[4] - w_[5]).max())
    return n_nid, n_tsbk, len(wv)
