"""The P25 Phase 1 chain object (include/ddn_chain.h: ddn_p25_chain_*) against the whole-stream oracle (tests/chain_stream.py):
a stream handed over in several calls + a flush gives, frame for frame, what the oracle gives for the stream in one piece -
including the frames that straddle call boundaries (the carried tail) - and the three ways of running a call (one stream,
pipelined over two, from pinned host memory) agree bit for bit."""
import ctypes as C

import numpy as np
import pytest

import chain_stream
import ddn
import mbe
import orc
import p25gen
from conftest import golden

pytestmark = pytest.mark.gpu


def _stream(B, n_total):
    """B channels of cu8 I/Q: the reference's two P25 captures, then synthetic voice, control and mixed traffic"""
    iq = np.full((B, n_total, 2), 127, np.uint8)
    rng = np.random.default_rng(4242)
    for c in range(B):
        if c == 0:
            g = golden("iq_p25p1_c4fm_vc.npz")["iq"]
            iq[c, :min(n_total, len(g))] = g[:n_total]
        elif c == 1:
            g = golden("iq_p25p1_c4fm_cc.npz")["iq"]
            iq[c, :min(n_total, len(g))] = g[:n_total]
        elif c % 3 == 2:
            n_ldus = n_total // 8640 + 1
            bits = mbe.random_imbe_bits(rng, (n_ldus * 9,))
            dib, _ = p25gen.make_ldus(rng, n_ldus, 0x293, np.stack([mbe.imbe_encode(b) for b in bits]))
            iq[c] = p25gen.modulate_cu8(dib, n_total, lead=200 + 37 * c, seed=c, noise=0.02 + 0.04 * (c % 2))
        elif c % 3 == 0:
            parts = []
            while sum(len(p) for p in parts) * 10 < n_total:
                parts.append(p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=int(rng.integers(1, 4)))[0])
                parts.append(np.zeros(int(rng.integers(0, 40)), np.int8))
            iq[c] = p25gen.modulate_cu8(np.concatenate(parts), n_total, lead=300 + 11 * c, seed=c, noise=0.03)
        else:
            parts = []
            while sum(len(p) for p in parts) * 10 < n_total:
                k = int(rng.integers(0, 5))
                if k == 0:
                    bits = mbe.random_imbe_bits(rng, (18,))
                    parts.append(p25gen.make_ldus(rng, 2, 0x293, np.stack([mbe.imbe_encode(b) for b in bits]))[0])
                elif k == 1:
                    parts.append(p25gen.make_frames(rng, 2, 0x293, crc=True, blocks=3)[0])
                elif k == 2:
                    parts.append(p25gen.make_hdu(rng, 0x293)[0] if rng.random() < 0.7 else p25gen.frame_with_duid(rng, 0x293, 0x0, 339 + 5))
                elif k == 3:
                    parts.append(p25gen.make_tdulc(rng, 0x293)[0] if rng.random() < 0.7 else p25gen.frame_with_duid(rng, 0x293, 0xF, 159 + 5))
                else:
                    parts.append(p25gen.make_pdu(rng, 0x293, int(rng.integers(0, 4))))
            iq[c] = p25gen.modulate_cu8(np.concatenate(parts), n_total, lead=250 + 7 * c, seed=c, noise=0.03)
    return iq


def _upload(a):
    p = C.c_void_p()
    assert ddn.lib().ddn_device_alloc(a.nbytes, C.byref(p)) == 0
    assert ddn.lib().ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
    return p


def _run(iq, n_call, how="run", everything=False, max_ldu=0):
    B, n_total = iq.shape[0], iq.shape[1]
    ch = ddn.P25ChainC(B, n_call, max_ldu=max_ldu)
    col = chain_stream.Collector(ch, everything)
    pinned = []
    for a in range(0, n_total, n_call):
        part = np.ascontiguousarray(iq[:, a:a + n_call])
        if how == "host":
            p = C.c_void_p()
            assert ddn.lib().ddn_host_alloc_pinned(part.nbytes, C.byref(p)) == 0
            C.memmove(p, part.ctypes.data, part.nbytes)
            pinned.append(p)
            ch.run_host(p, None)
            ch.wait()
        else:
            d = _upload(part)
            if how == "run":
                ch.run(d)
            else:
                ch.run_pipelined(d)
                ch.wait()
            ddn.lib().ddn_device_free(d)
        col.take()
    ch.flush()
    col.take()
    for p in pinned:
        ddn.lib().ddn_host_free_pinned(p)
    ch.close()
    return col


_check_against_oracle = chain_stream.check_channel


def test_stream_in_calls_equals_the_oracle(built):
    B, n_call, calls = 6, 24000, 6
    iq = _stream(B, n_call * calls)
    col = _run(iq, n_call)
    tot = np.zeros(3, np.int64)
    for c in range(B):
        want = chain_stream.run_stream(iq[c], n_call, seed=c)
        tot += _check_against_oracle(col, c, want)
    assert tot[0] > 150 and tot[1] > 150 and tot[2] > 250, tot


def test_short_calls_and_frames_across_boundaries(built):
    """calls shorter than a voice frame: every LDU straddles one or more call boundaries"""
    B, n_call, calls = 3, 6000, 16
    iq = _stream(B, n_call * calls)
    col = _run(iq, n_call, max_ldu=4)
    for c in range(B):
        want = chain_stream.run_stream(iq[c], n_call, seed=c)
        _check_against_oracle(col, c, want)


def test_parity_override_ldu_still_gives_voice(built):
    """an LDU whose 64th NID bit (the parity bit, outside the BCH word) arrives flipped decodes as NID_PARITY_OVERRIDE (status 2,
    src/protocol/p25/phase1/p25p1_check_nid.cpp:293-303) and is dispatched like any accepted frame (dispatch_p25p1.c:214-218:
    status > 0): Hamming / RS run AND its nine IMBE frames are synthesized"""
    rng = np.random.default_rng(77)
    n_call, calls = 15000, 3
    bits = mbe.random_imbe_bits(rng, (45,))
    dib, _ = p25gen.make_ldus(rng, 5, 0x293, np.stack([mbe.imbe_encode(b) for b in bits]))
    dib = dib.copy()
    for f in (2, 3):
        dib[f * 864 + 56] ^= 1            # frame dibit 56 = BCH bit 62 | parity bit
    iq = p25gen.modulate_cu8(dib, n_call * calls, lead=260, seed=5, noise=0.02)[None]
    col = _run(iq, n_call, everything=True)
    want = chain_stream.run_stream(iq[0], n_call, seed=0)
    _check_against_oracle(col, 0, want)
    st = sorted((g, int(d["nid"][0]), int(d["nid"][2])) for g, d in col.frames[0].items())
    assert len(st) == 5
    # (the stream's very first NID is read before the slicer has settled and may fail; the flipped ones are frames 2 and 3)
    assert [s[1] for s in st[1:]] == [1, 2, 2, 1] and [s[2] for s in st[1:]] == [10, 5, 10, 5], st
    for g, status, duid in st[1:]:
        d = col.frames[0][g]
        assert (d["rs1s"] if duid == 5 else d["rs2s"]) == 0
    ok = {g for g, status, _ in st if status > 0}
    voiced = [v for v in col.voice[0] if v[0] in ok]
    assert len(voiced) == 9 * len(ok) and {v[0] for v in voiced} == ok
    by_frame = {g: np.stack([v[4] for v in voiced if v[0] == g]) for g in ok}
    for g, status, _ in st[1:]:
        assert (np.abs(by_frame[g]).sum(axis=1) > 0).all(), (g, status)      # PCM for the parity-override frames too


_BY_TYPE = {"words1": 5, "rs1": 5, "rs1s": 5, "words2": 10, "rs2": 10, "rs2s": 10, "hdu": 0, "hdus": 0, "tdulc": 15, "tdulcs": 15}


def _same(a, b):
    assert sorted(a) == sorted(b)
    for g in a:
        ok, duid = a[g]["nid"][0] > 0, a[g]["nid"][2]
        for k, v in a[g].items():
            if k in _BY_TYPE and not (ok and duid == _BY_TYPE[k]):
                continue             # per-type outputs are only written for frames of that type
            if k in ("lsd", "lsd_ok") and not (ok and duid in (5, 10)):
                continue
            w = b[g][k]
            if k == "tsbk":
                assert all(np.array_equal(x[0], y[0]) and x[1] == y[1] for x, y in zip(v, w)), (g, k)
            else:
                assert np.array_equal(v, w), (g, k)


def test_split_invariance_and_the_three_run_forms(built):
    """every output array (LDU words, Reed-Solomon data and status, low speed data, HDU, TDULC too), frame by frame: one long call
    == many short ones == pipelined == from pinned host memory"""
    B, n_total = 6, 98304        # whole front-end blocks per call in every split: the block partition is part of the result
    iq = _stream(B, n_total)
    one = _run(iq, n_total, everything=True)
    for how, n_call in (("run", 16384), ("pipelined", 8192), ("host", 24576)):
        got = _run(iq, n_call, how=how, everything=True)
        for c in range(B):
            assert np.array_equal(np.concatenate(got.rec[c]), np.concatenate(one.rec[c])), (how, c)
            # frames near the end of the stream see a shorter tail in the flush than inside the long call
            cnt = len(np.concatenate(one.rec[c]))
            a = {g: v for g, v in one.frames[c].items() if g + 900 < cnt}
            b = {g: v for g, v in got.frames[c].items() if g + 900 < cnt}
            _same(a, b)
            va = [v for v in one.voice[c] if v[0] + 900 < cnt]
            vb = [v for v in got.voice[c] if v[0] + 900 < cnt]
            assert len(va) == len(vb) and all(x[0] == y[0] and np.array_equal(x[2], y[2]) and np.array_equal(x[4].view(np.uint32), y[4].view(np.uint32))
                                              for x, y in zip(va, vb)), (how, c)


def test_run_host_copies_back_what_the_device_holds(built):
    """ddn_p25_chain_run_host with output buffers: the pinned-host copies of a call (records, flags, counts, decisions + payloads,
    NIDs, TSDU blocks, PCM) equal the device arrays of that call, call after call (the D2H copies run on their own stream behind
    the decode, double buffered against the next call)"""
    l = ddn.lib()
    B, n_call, calls = 4, 16384, 4
    iq = _stream(B, n_call * calls)
    ch = ddn.P25ChainC(B, n_call)
    S, V, st, E = B * ch.F, B * ch.Fv * 9, ch.stride, ch.E
    shapes = {"records10": (np.uint8, (B, st, 10)), "flags": (np.uint8, (B, st)), "counts": (np.int32, (B,)),
              "events": (np.int32, (B, E, 4)), "n_events": (np.int32, (B,)), "event_data": (np.int32, (B, E, 4)),
              "nid4": (np.int32, (S, 4)), "tsbk": (np.uint8, (3, S, 12)), "pcm": (np.float32, (V, 160)),
              "records2": (np.uint8, (B, st, 2)), "pcm_dense": (np.float32, (V, 160)), "pcm_slot": (np.int32, (V,)),
              "pcm_count": (np.int32, (1,))}
    pinned, outs, views = [], [], []
    for _ in range(2):                                   # two sets: call k's copies finish while call k + 1 runs
        o, v = ddn.P25ChainHostOut(), {}
        for name, (dt, shp) in shapes.items():
            nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
            p = C.c_void_p()
            assert l.ddn_host_alloc_pinned(nbytes, C.byref(p)) == 0
            pinned.append(p)
            setattr(o, name, p.value)
            v[name] = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dt).reshape(shp)
        o.pcm_dense_frames = V
        outs.append(o)
        views.append(v)
    h_iq = []
    for k in range(calls):
        part = np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call])
        p = C.c_void_p()
        assert l.ddn_host_alloc_pinned(part.nbytes, C.byref(p)) == 0
        C.memmove(p, part.ctypes.data, part.nbytes)
        h_iq.append(p)
    names = {"records10": "d_records10", "flags": "d_flags", "counts": "d_counts", "events": "d_events", "n_events": "d_n_events",
             "event_data": "d_event_data", "nid4": "d_nid4", "tsbk": "d_tsbk", "pcm": "d_pcm"}
    seen_pcm = 0.0
    for k in range(calls):
        ch.run_host(h_iq[k], outs[k & 1])
        ch.wait()
        r = ch.results()
        for name, (dt, shp) in shapes.items():
            host = views[k & 1][name]
            if name in ("pcm_slot", "pcm_count"):
                continue
            if name == "pcm_dense":                       # the synthesized frames only, dense, in slot order
                used = np.flatnonzero(ch.fetch(r.d_imbe_result, np.int32, (V, 5))[:, 0] >= 0)
                cnt = int(views[k & 1]["pcm_count"][0])
                assert cnt == len(used) and np.array_equal(views[k & 1]["pcm_slot"][:cnt], used), (k, cnt, len(used))
                assert np.array_equal(host[:cnt].view(np.uint32), views[k & 1]["pcm"][used].view(np.uint32)), k
                continue
            if name == "records2":                        # the host form: {dibit | flags << 2, reliability} of every record
                rec, fl = views[k & 1]["records10"], views[k & 1]["flags"]
                assert np.array_equal(host[..., 0], (rec[..., 0] & 3) | ((fl & 0x3F) << 2)), k
                assert np.array_equal(host[..., 1], rec[..., 1]), k
                continue
            dev = ch.fetch(getattr(r, names[name]), dt, shp)
            if name in ("events", "event_data"):          # rows beyond n_events are not written by the loop
                ne = ch.fetch(r.d_n_events, np.int32, (B,))
                assert all(np.array_equal(dev[c, :ne[c]], host[c, :ne[c]]) for c in range(B)), (k, name)
            else:
                assert np.array_equal(dev.view(np.uint8), host.view(np.uint8)), (k, name)
        seen_pcm += float(np.abs(views[k & 1]["pcm"]).sum())
    assert seen_pcm > 0
    ch.close()
    for p in pinned + h_iq:
        l.ddn_host_free_pinned(p)


def test_run_host_then_device_form_keeps_the_host_results(built):
    """A _run_host call defers its result copies; a device-form call right behind it (no _wait in between) must not decode over the
    single decode buffers before those copies have left: host buffers = what a run of the same stream with a _wait after every
    call delivers for that call (NIDs, TSDU blocks, counts, PCM, records)"""
    l = ddn.lib()
    B, n_call = 4, 16384
    iq = _stream(B, n_call * 3)
    parts = [np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call]) for k in range(3)]

    def host_bufs(ch):
        S, V, st = B * ch.F, B * ch.Fv * 9, ch.stride
        shapes = {"records10": (np.uint8, (B, st, 10)), "counts": (np.int32, (B,)), "nid4": (np.int32, (S, 4)),
                  "tsbk": (np.uint8, (3, S, 12)), "pcm": (np.float32, (V, 160))}
        o, v, pins = ddn.P25ChainHostOut(), {}, []
        for name, (dt, shp) in shapes.items():
            nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
            p = C.c_void_p()
            assert l.ddn_host_alloc_pinned(nbytes, C.byref(p)) == 0
            pins.append(p)
            setattr(o, name, p.value)
            v[name] = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dt).reshape(shp)
        return o, v, pins

    def pin_iq(part):
        p = C.c_void_p()
        assert l.ddn_host_alloc_pinned(part.nbytes, C.byref(p)) == 0
        C.memmove(p, part.ctypes.data, part.nbytes)
        return p

    results = {}
    for how in ("waited", "pipelined", "run", "stage"):
        ch = ddn.P25ChainC(B, n_call)
        o, v, pins = host_bufs(ch)
        h0, h1 = pin_iq(parts[0]), pin_iq(parts[1])
        ch.run_host(h0, None)
        ch.run_host(h1, o)                       # call 1: the results this test is about
        d2 = _upload(parts[2])
        if how == "waited":
            ch.wait()
        if how in ("waited", "pipelined"):
            ch.run_pipelined(d2)
        elif how == "run":
            ch.run(d2)
        else:
            for stage in range(3):
                assert l.ddn_p25_chain_stage(ch.h, stage, d2 if stage == 0 else None, None) == 0
        ch.wait()
        results[how] = {k: a.copy() for k, a in v.items()}
        ch.close()
        l.ddn_device_free(d2)
        for p in pins + [h0, h1]:
            l.ddn_host_free_pinned(p)
    assert float(np.abs(results["waited"]["pcm"]).sum()) > 0
    for how in ("pipelined", "run", "stage"):
        for k, a in results["waited"].items():
            assert np.array_equal(a.view(np.uint8), results[how][k].view(np.uint8)), (how, k)


def test_data_units_header_blocks_and_crc32(built):
    """P25 data units (DUID 0xC, processMPDU): the chain files every unit of a call - header as the loop decoded it, the data blocks
    behind it through the half-rate trellis (best path, p25p1_mdpu.c:263-273), CRC32 over the data (crc32mbf) - and follows the
    reference's first fallback when the header fails its CRC16 (blocks 1 / 2 read as header repetitions).  Stream in calls + flush
    against the restatement on the whole stream (the handlers' header decisions from the loop oracle, the blocks from the oracle's
    half-rate list decoder, candidate 0), units that straddle a call boundary included; confirmed (rate 3/4) units are flagged"""
    rng = np.random.default_rng(11)
    nac = 0x293
    units = []
    parts = [p25gen.make_frames(rng, 1, nac, crc=True, blocks=1)[0], np.zeros(160, np.int8)]      # (the slicer settles on this one)
    plan = [dict(blks=2), dict(blks=0), dict(blks=5), dict(blks=1, good_crc32=False), dict(blks=3, confirmed=True),
            dict(blks=4, confirmed=True, bad_crc9_at=(1,)), dict(blks=2, confirmed=True, good_crc32=False),
            dict(blks=0, good_crc16=False, header_reps=2), dict(blks=7), dict(blks=4), dict(blks=2, good_crc16=False), dict(blks=6),
            dict(blks=0, combined=True), dict(blks=1), dict(blks=0, combined=True),
            dict(blks=8), dict(blks=8, confirmed=True), dict(blks=9), dict(blks=8, confirmed=True), dict(blks=8)]   # (eight blocks: the default carry's most)
    for kw in plan:
        fr, hdr, data = p25gen.make_pdu_coded(rng, nac, **kw)
        units.append((kw, hdr, data, sum(len(q) for q in parts)))
        parts += [fr, np.zeros(int(rng.integers(150, 220)), np.int8)]      # (at most pdu_per_channel = 2 units end inside one call)
    dib = np.concatenate(parts)
    n_call = 9000
    calls = -(-(len(dib) * 10 + 600) // n_call)
    iq = p25gen.modulate_cu8(dib, n_call * calls, lead=260, seed=9, noise=0.02)[None]
    ch = ddn.P25ChainC(1, n_call)
    PFn = PBn = None
    got = {}
    base = 0
    for k in range(calls + 1):
        if k < calls:
            d = _upload(np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call]))
            ch.run(d)
            ch.wait()
            ddn.lib().ddn_device_free(d)
        else:
            ch.flush()
        r = ch.results()
        PFn, PBn = r.pdu_per_channel, r.pdu_blocks
        npdu = int(ch.fetch(r.d_n_pdu, np.int32, (1,))[0])
        slot = ch.fetch(r.d_pdu_slot, np.int32, (PFn,))
        hdr = ch.fetch(r.d_pdu_header, np.uint8, (PFn, 12))
        info = ch.fetch(r.d_pdu_info, np.int32, (PFn, 4))
        blk = ch.fetch(r.d_pdu_blocks, np.uint8, (PFn, PBn, 12))
        vld = ch.fetch(r.d_pdu_block_valid, np.uint8, (PFn, PBn))
        b18 = ch.fetch(r.d_pdu_blocks18, np.uint8, (PFn, PBn, 18))
        c9 = ch.fetch(r.d_pdu_crc9_ok, np.uint8, (PFn, PBn))
        pos = ch.fetch(r.d_sync_pos, np.int32, (ch.F,))
        assert npdu <= PFn
        for e in range(PFn):
            if slot[e] >= 0:
                g = base + int(pos[slot[e] % ch.F]) - ch.T
                assert g not in got
                got[g] = (hdr[e].copy(), info[e].copy(), blk[e].copy(), vld[e].copy(), b18[e].copy(), c9[e].copy())
        base += int(ch.fetch(r.d_new, np.int32, (1,))[0])
    ch.close()
    want = chain_stream.run_stream(iq[0], n_call, seed=0, vocoder=False)
    rec4 = want["rec4"]
    cnt = len(want["sym"])
    evs = {int(e[0]): (e, d) for e, d in zip(want["events"], want["event_data"]) if e[1] == 3}
    pdus = [a for a, f in sorted(want["frames"].items()) if "nid" in f and f["nid"][0] > 0 and f["nid"][2] == 0xC and a + 33 + 101 in evs]
    assert sorted(got) == pdus and len(pdus) >= len(plan) - 1, (sorted(got), pdus)
    # which unit a decoded sync belongs to: the stream positions differ by a constant (lead, filter delays)
    shift = int(np.median([a - u[3] for a, u in zip(pdus, units)])) if len(pdus) == len(units) else None
    if shift is None:
        shift = min((pdus[0] - units[0][3], pdus[-1] - units[-1][3]), key=abs)
    by_pos = {u[3]: u for u in units}
    seen_flags = set()
    for a in pdus:
        near = min(by_pos, key=lambda q: abs(a - shift - q))
        assert abs(a - shift - near) <= 12, (a, shift, near)      # (the symbol clock drifts a few symbols over the stream)
        kw, hdr_sent, data_sent, _ = by_pos[near]
        e, d = evs[a + 33 + 101]
        w_hdr = d[:3].copy().view(np.uint8)
        w_crc, w_end = int(d[3]) & 1, int(e[3]) & 0xFFFF
        hdr, info, blk, vld, blk18, crc9ok = got[a]
        blocks, llrs = [], []
        for b in range(1, w_end):
            idx = [a - 23 + n + n // 35 for n in range(56 + 98 * b, 56 + 98 * b + 98)]
            if idx[-1] >= cnt or b > PBn:
                blocks.append(None)
                llrs.append(None)
                continue
            llr = np.stack([rec4[idx, 2], rec4[idx, 3]], axis=1).reshape(196)
            o = orc.oracle()
            o.orc_p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            ob, om = np.zeros((8, 12), np.uint8), np.zeros(8, np.uint32)
            assert o.orc_p25_12_soft_llr_list(np.ascontiguousarray(llr, np.int16).ctypes.data, ob.ctypes.data, om.ctypes.data, 8) >= 1
            blocks.append(ob[0].copy())
            llrs.append(np.ascontiguousarray(llr, np.int16))
        flags = 0
        ok = w_crc
        if not ok:
            for rep in (1, 2):
                bb = blocks[rep - 1] if rep < w_end and rep - 1 < len(blocks) else None
                if bb is not None and p25gen.crc16_ccitt(bb[:10]) == ((int(bb[10]) << 8) | int(bb[11])):
                    w_hdr, ok, flags = bb, 1, flags | rep
                    break
            if not ok and w_end >= 3 and len(blocks) >= 2 and blocks[0] is not None and blocks[1] is not None:
                # every repetition failed: the summed LLRs through the list decoder, then the bitwise majority (:336-410)
                idx0 = [a - 23 + n + n // 35 for n in range(56, 56 + 98)]
                l0 = np.stack([rec4[idx0, 2], rec4[idx0, 3]], axis=1).reshape(196).astype(np.int16)
                o.orc_p25_mpdu_finalize_header.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
                rb = np.ascontiguousarray(np.stack([w_hdr, blocks[0], blocks[1]]).astype(np.uint8))
                ll = np.ascontiguousarray(np.stack([l0, llrs[0], llrs[1]]))
                out = np.zeros(12, np.uint8)
                how = o.orc_p25_mpdu_finalize_header(rb.ctypes.data, ll.ctypes.data, 3, out.ctypes.data)
                assert how in (32, 64, 64 | 16), how
                w_hdr, flags = out, flags | how
                ok = 0 if how & 16 else 1
            elif not ok:
                flags |= 16
        r34 = bool(w_crc) and bool((w_hdr[0] >> 6) & 1) and (w_hdr[0] & 0x1F) == 0x16          # (the FIRST header decides)
        if r34:
            flags |= 4
        if any(b is None for b in blocks):
            flags |= 8
        blks = int(w_hdr[6]) & 0x7F
        crc32 = 0
        if ok and not (flags & (15 | 32 | 64)):
            if blks == 0:
                crc32 = 1
            elif blks == w_end - 1:
                flat = np.concatenate(blocks)
                crc32 = int(p25gen.crc32mbf(flat, 96 * blks - 32) == int.from_bytes(bytes(flat[-4:].tolist()), "big"))
        w18 = []
        if r34:                 # the rate 3/4 list decoder, first candidate with a good CRC9, CRC32 over the 16-byte payloads
            o.orc_p25_mbf34_list.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            for l in llrs:
                if l is None:
                    w18.append(None)
                    continue
                cb, cm = np.zeros((8, 18), np.uint8), np.zeros(8, np.uint32)
                nc = o.orc_p25_mbf34_list(l.ctypes.data, 8, cb.ctypes.data, cm.ctypes.data)
                assert nc >= 1
                good = [k for k in range(nc) if p25gen.crc9([(int(cb[k, 0]) >> (7 - i)) & 1 for i in range(7)]
                                                           + list(np.unpackbits(cb[k, 2:]))) == (((int(cb[k, 0]) & 1) << 8) | int(cb[k, 1]))]
                w18.append((cb[good[0] if good else 0].copy(), 1 if good else 0))
            if not (flags & 8):
                if blks == 0:
                    crc32 = 1
                elif blks == w_end - 1:
                    flat = np.concatenate([w[0][2:] for w in w18])
                    crc32 = int(p25gen.crc32mbf(flat, 128 * blks - 32) == int.from_bytes(bytes(flat[-4:].tolist()), "big"))
        assert np.array_equal(hdr, w_hdr) and tuple(info) == (ok, w_end, flags, crc32), (a, kw, hdr, w_hdr, info, (ok, w_end, flags, crc32))
        for b, bb in enumerate(blocks):
            if bb is not None:
                assert vld[b] == 1 and np.array_equal(blk[b], bb), (a, b)
            elif b < PBn:
                assert vld[b] == 0
        for b, w in enumerate(w18):
            if w is not None:
                assert np.array_equal(blk18[b], w[0]) and crc9ok[b] == w[1], (a, b, blk18[b], w)
        if kw.get("confirmed") and kw.get("good_crc16", True):
            assert r34 and w_end == kw["blks"] + 1 and crc32 == (1 if kw.get("good_crc32", True) else 0), (a, kw, crc32)
            for b in range(kw["blks"]):
                assert np.array_equal(blk18[b], data_sent[b]) and crc9ok[b] == (0 if b in kw.get("bad_crc9_at", ()) else 1), (a, kw, b)
        seen_flags.add(flags)
        # the clean units decode to what was sent
        if kw.get("combined"):
            assert flags == 32 and ok == 1 and np.array_equal(hdr, hdr_sent), (a, kw, flags, hdr, hdr_sent)
        elif kw.get("good_crc16", True) and not kw.get("confirmed"):
            assert ok == 1 and np.array_equal(hdr, hdr_sent) and w_end == kw["blks"] + 1
            if kw["blks"] > PBn:      # longer than the chain reads: flagged, no CRC32 verdict, the blocks it did read are right
                assert (flags & 8) and crc32 == 0, (a, kw, flags)
            else:
                assert crc32 == (1 if kw.get("good_crc32", True) else 0), (a, kw)
            assert all(np.array_equal(blk[b], data_sent[b]) for b in range(min(kw["blks"], PBn)))
    assert PBn == 8 and {0, 1, 4, 8, 32, 64 | 16} <= seen_flags, seen_flags


def test_syncs_beyond_the_frame_slots_are_counted_not_lost_silently(built):
    """max_frames too small for the traffic: the chain decodes the first max_frames syncs of a call and reports the rest in
    d_dropped_syncs (a running count per channel) - and with the default slots the same traffic drops nothing"""
    rng = np.random.default_rng(5)
    n_call, calls = 24000, 2
    parts = [p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=1)[0] for _ in range(26)]     # 13 single-block TSDUs per call
    iq = p25gen.modulate_cu8(np.concatenate(parts), n_call * calls, lead=240, seed=3, noise=0.02)[None]
    want = chain_stream.run_stream(iq[0], n_call, seed=0)
    found = len(want["frames"])
    assert found >= 24
    for max_frames, expect_drop in ((4, True), (0, False)):
        ch = ddn.P25ChainC(1, n_call, max_frames=max_frames)
        used = 0
        for k in range(calls):
            d = _upload(np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call]))
            ch.run(d)
            ch.wait()
            r = ch.results()
            ns = int(ch.fetch(r.d_n_syncs, np.int32, (1,))[0])
            assert ns <= ch.F
            used += ns
            ddn.lib().ddn_device_free(d)
        ch.flush()
        r = ch.results()
        used += int(ch.fetch(r.d_n_syncs, np.int32, (1,))[0])
        dropped = int(ch.fetch(r.d_dropped_syncs, np.int32, (1,))[0])
        ch.close()
        if expect_drop:
            assert dropped > 0 and used <= 3 * 4 and used + dropped >= found, (used, dropped, found)
        else:
            assert dropped == 0 and used == found, (used, dropped, found)


def test_run_host_streaming_host_never_waits(built):
    """the contract of ddn_p25_chain_run_host as include/ddn_chain.h words it, used the way a streaming host would: two pinned input
    buffers refilled in turn as soon as the NEXT call has returned, three output sets, call k's read as soon as call k + 3 has returned
    (and before call k + 4 is made), no ddn_p25_chain_wait() until the end.  Every call's outputs equal those of a run that waits
    after every call"""
    l = ddn.lib()
    B, n_call, calls = 4, 16384, 7
    iq = _stream(B, n_call * calls)
    fields = {"records2": np.uint8, "counts": np.int32, "nid4": np.int32, "tsbk": np.uint8, "pcm_dense": np.float32,
              "pcm_slot": np.int32, "pcm_count": np.int32, "n_events": np.int32}

    def pin(nbytes):
        p = C.c_void_p()
        assert l.ddn_host_alloc_pinned(nbytes, C.byref(p)) == 0
        return p

    def go(streaming):
        ch = ddn.P25ChainC(B, n_call)
        S, V, st = B * ch.F, B * ch.Fv * 9, ch.stride
        sizes = {"records2": B * st * 2, "counts": B * 4, "nid4": S * 16, "tsbk": 3 * S * 12, "pcm_dense": V * 640, "pcm_slot": V * 4,
                 "pcm_count": 4, "n_events": B * 4}
        pinned, outs, views = [], [], []
        for _ in range(3):
            o, v = ddn.P25ChainHostOut(), {}
            for name, nb in sizes.items():
                p = pin(nb)
                pinned.append(p)
                setattr(o, name, p.value)
                v[name] = np.frombuffer((C.c_uint8 * nb).from_address(p.value), dtype=fields[name])
            o.pcm_dense_frames = V
            outs.append(o)
            views.append(v)
        h_in = [pin(B * n_call * 2) for _ in range(2)]
        got = []

        def read(k):
            v = views[k % 3]
            cnt = int(v["pcm_count"][0])
            got.append({name: (a[:cnt * 160].copy() if name == "pcm_dense" else a[:cnt].copy() if name == "pcm_slot" else a.copy())
                        for name, a in v.items()})
            for a in v.values():
                a[...] = 0xA5 if a.dtype == np.uint8 else 0      # a stale set must not pass for a result

        for k in range(calls):
            part = np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call])
            C.memmove(h_in[k & 1], part.ctypes.data, part.nbytes)     # legal: call k - 1 (the last user of k - 2's buffer) has returned
            ch.run_host(h_in[k & 1], outs[k % 3])
            if streaming:
                if k >= 3:
                    read(k - 3)
            else:
                ch.wait()
                read(k)
        if streaming:
            ch.wait()
            read(calls - 3)
            read(calls - 2)
            read(calls - 1)
        ch.close()
        for p in pinned + h_in:
            l.ddn_host_free_pinned(p)
        return got

    want, got = go(False), go(True)
    assert len(want) == len(got) == calls
    frames = 0
    for k in range(calls):
        for name in fields:
            assert np.array_equal(want[k][name].view(np.uint8), got[k][name].view(np.uint8)), (k, name)
        frames += int(want[k]["pcm_count"][0])
    assert frames > 100


@pytest.mark.parametrize("case", range(4))
def test_odd_call_sizes_and_noise_against_the_oracle(built, case):
    """the chain at call sizes that cut the front end's blocks, the receive loop's tiles and the frames at odd places, on noisy mixed
    traffic (NIDs through the Chase search, TSDU blocks through the list decoder, failed NIDs): a stream in calls + flush equals the
    whole-stream oracle; DDN_FUZZ_BASE shifts the seeds"""
    import os
    base = int(os.environ.get("DDN_FUZZ_BASE", "0"))
    rng = np.random.default_rng(9000 + 17 * base + case)
    n_call = int(rng.choice([4999, 8192, 12345, 20011, 33333]))
    calls = int(rng.integers(3, 7))
    B = 4
    n_total = n_call * calls
    iq = np.full((B, n_total, 2), 127, np.uint8)
    for c in range(B):
        parts = []
        while sum(len(p) for p in parts) * 10 < n_total:
            k = int(rng.integers(0, 7))
            if k == 0:
                bits = mbe.random_imbe_bits(rng, (18,), 215)
                parts.append(p25gen.make_ldus(rng, 2, 0x293, np.stack([mbe.imbe_encode(b) for b in bits]))[0])
            elif k == 1:
                parts.append(p25gen.make_frames(rng, 1, 0x293, crc=bool(rng.integers(0, 4)), blocks=int(rng.integers(1, 4)))[0])
            elif k == 2:
                parts.append(p25gen.make_hdu(rng, 0x293)[0])
            elif k == 3:
                parts.append(p25gen.make_tdulc(rng, 0x293)[0])
            elif k == 4:
                parts.append(p25gen.make_pdu(rng, 0x293, int(rng.integers(0, 4)), good_crc=bool(rng.integers(0, 2))))
            elif k == 5:
                parts.append(p25gen.frame_with_duid(rng, 0x293, int(rng.choice([0x1, 0x9, 0x3])), int(rng.integers(10, 60))))
            else:
                parts.append(np.zeros(int(rng.integers(5, 120)), np.int8))
        noise = float(rng.choice([0.02, 0.08, 0.16, 0.24]))
        iq[c] = p25gen.modulate_cu8(np.concatenate(parts), n_total, lead=int(rng.integers(50, 600)), seed=int(rng.integers(0, 1 << 30)),
                                    noise=noise)
    col = _run(iq, n_call, how=["run", "pipelined", "host"][case % 3])
    tot = np.zeros(3, np.int64)
    for c in range(B):
        tot += _check_against_oracle(col, c, chain_stream.run_stream(iq[c], n_call, seed=c))
    assert tot[0] > 3
