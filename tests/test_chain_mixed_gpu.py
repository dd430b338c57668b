"""The DMR / NXDN48 chain object and the mixed-protocol object (include/ddn_chain.h: ddn_fsk4_chain_*, ddn_mixed_chain_*) on the
reference's own captures: receive-loop outputs with the handlers inside the loop equal the CPU restatement bit for bit over two
streamed calls, the frame FEC behind it gives the captures' known answers, and the three groups of a mixed batch run side by side."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import rx4
from conftest import golden

pytestmark = pytest.mark.gpu
N = 48000


def _upload(a):
    p = C.c_void_p()
    assert ddn.lib().ddn_device_alloc(a.nbytes, C.byref(p)) == 0
    assert ddn.lib().ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
    return p


def _tiles(name, lo, B, calls, n=N):
    iq = np.ascontiguousarray(golden(name)["iq"], np.uint8)
    offs = [lo + 371 * c for c in range(B)]
    return [np.stack([iq[o + k * n:o + (k + 1) * n] for o in offs]) for k in range(calls)], offs, iq


def _check_rx(ch, Bc, calls_iq, proto, lpf, rf):
    """streamed calls + the flush against the restatement fed the same pieces: this call's records / flags / payload dibits bit for
    bit, and every accepted sync of the stream decoded exactly once (in the call that brings the records behind it, or in the
    flush), with its position and 90-dibit hand-over"""
    fes = [orc.OracleFrontEnd(profile=lpf) for _ in range(Bc)]
    rxs = [rx4.OracleFsk4Rx(rx4.profile(proto, rf_mod=rf, handler=1)) for _ in range(Bc)]
    base = np.zeros(Bc, np.int64)
    got = [[] for _ in range(Bc)]
    want = [[] for _ in range(Bc)]
    out = []

    def take():
        r = ch.results()
        st, T, my = r.stride_symbols, r.carry_symbols, r.max_syncs
        f = ch.fetch
        ns = f(r.d_n_sync, np.int32, (Bc,))
        spos, pre = f(r.d_sync_pos, np.int32, (Bc, my)), f(r.d_pre, np.uint8, (Bc, my, 90))
        for c in range(Bc):
            got[c] += [(int(base[c]) + int(spos[c, k]) - int(T), pre[c, k].copy()) for k in range(int(ns[c]))]
        out.append((r, ns))
        return r, st, T

    for part in calls_iq:
        d = _upload(part)
        ch.run(d)
        r, st, T = take()
        f = ch.fetch
        rec, fl, pay = f(r.d_records10, np.uint8, (Bc, st, 10)), f(r.d_flags, np.uint8, (Bc, st)), f(r.d_payload2, np.uint8, (Bc, st, 2))
        new, cnt = f(r.d_new, np.int32, (Bc,)), f(r.d_counts, np.int32, (Bc,))
        ddn.lib().ddn_device_free(d)
        for c in range(Bc):
            disc = fes[c].run_cu8(np.ascontiguousarray(part[c]), 8192)
            w = rxs[c].run(disc, max_sync=512)
            k = int(new[c])
            assert k == len(w["sym"]) and cnt[c] == k + T, (c, k, len(w["sym"]))
            rr = rec[c, T:T + k]
            assert np.array_equal(rr[:, 6:10].copy().view(np.uint32).reshape(-1), w["sym"].view(np.uint32)), c
            assert np.array_equal(rr[:, 0].astype(np.int32), w["rec4"][:, 0]) and np.array_equal(fl[c, T:T + k], w["fl"]), c
            assert np.array_equal(pay[c, T:T + k], w["pay"]), c
            want[c] += [(int(base[c]) + int(p), q) for p, q in zip(w["sync_pos"], w["pre"])]
            base[c] += k
    ch.flush()
    r, _, _ = take()
    assert not ch.fetch(r.d_dropped_syncs, np.int32, (Bc,)).any()
    for c in range(Bc):
        assert [g for g, _ in got[c]] == [g for g, _ in want[c]], (c, len(got[c]), len(want[c]))
        assert all(np.array_equal(a[1], b[1]) for a, b in zip(got[c], want[c])), c
    return out


def test_dmr_chain_two_calls_and_known_answer(built):
    B = 4
    n = 44000                                  # the capture holds 96000 samples
    calls, _, _ = _tiles("iq_dmr_t3_ras_cc.npz", 0, B, 2, n)
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_DMR, rf_mod=2)
    _check_rx(ch, B, calls, rx4.PROTO_DMR, 2, 2)
    ch.close()
    # known answer on a fresh object, the decode outputs of its second call (the bursts that straddle the first boundary included)
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_DMR, rf_mod=2)
    for part in calls:
        d = _upload(part)
        ch.run(d)
        ddn.lib().ddn_device_free(d)
    r = ch.results()
    my = r.max_syncs
    S = B * my
    ns = ch.fetch(r.d_n_sync, np.int32, (B,))
    valid, st_ok = ch.fetch(r.d_valid, np.uint8, (S,)), ch.fetch(r.d_dmr_slot_type_ok, np.uint8, (S,))
    stb, errs = ch.fetch(r.d_dmr_slot_type, np.uint8, (S, 20)), ch.fetch(r.d_dmr_bptc_errs, np.uint32, (S,))
    used = (np.arange(my)[None, :] < ns[:, None]).reshape(S)
    rows = np.flatnonzero(used)
    # every burst the call decodes is complete (the carry), Tier III control channel: colour code 0 in every slot type, BPTC clean
    assert len(rows) > 20 and np.all(valid[rows] == 1) and np.all(st_ok[rows] == 1) and np.all(stb[rows][:, :4] == 0)
    assert np.mean(errs[rows] == 0) > 0.99
    ch.close()


def test_nxdn48_chain_two_calls_voice(built):
    B = 3
    calls, _, _ = _tiles("iq_nxdn48.npz", 60000, B, 2)
    ch = ddn.Fsk4ChainC(B, N, ddn.FSK4_NXDN48, rf_mod=0)
    _check_rx(ch, B, calls, rx4.PROTO_NXDN48, 1, 0)
    ch.close()
    ch = ddn.Fsk4ChainC(B, N, ddn.FSK4_NXDN48, rf_mod=0)
    for part in calls:
        d = _upload(part)
        ch.run(d)
        ddn.lib().ddn_device_free(d)
    r = ch.results()
    my, vf = r.max_syncs, r.voice_slots
    S = B * my
    ns = ch.fetch(r.d_n_sync, np.int32, (B,))
    nv, nl = ch.fetch(r.d_valid, np.uint8, (S,)), ch.fetch(r.d_nxdn_lich, np.uint8, (S,))
    s1, s2 = ch.fetch(r.d_nxdn_sacch_ok, np.uint8, (S,)), ch.fetch(r.d_nxdn_sacch_hard_ok, np.uint8, (S,))
    used = (np.arange(my)[None, :] < ns[:, None]).reshape(S)
    assert np.all(nv[used] == 1)                                  # the carry: every frame decoded in a call is complete
    rows = np.flatnonzero(used)
    assert len(rows) >= 10 and np.mean((nl[rows] & 0x80) != 0) > 0.9 and np.mean((s1[rows] | s2[rows]) != 0) > 0.5
    skip = ch.fetch(r.d_nxdn_voice_skip, np.uint8, (B, vf, 4))
    pcm = ch.fetch(r.d_nxdn_pcm, np.float32, (B, vf * 4, 160))
    assert np.all(np.isfinite(pcm)) and (skip == 0).sum() >= 8            # the capture is a voice call: frames were synthesized
    assert np.all(pcm.reshape(B, vf, 4, 160)[skip != 0] == 0)
    ch.close()


def test_mixed_chain_groups_side_by_side(built):
    """the three groups through ddn_mixed_chain give what each group's own chain object gives alone"""
    import p25gen
    rng = np.random.default_rng(5)
    Bp, Bd, Bn = 3, 2, 2
    p25 = np.stack([p25gen.modulate_cu8(np.concatenate([p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=1 + (c + k) % 3)[0] for k in range(20)]),
                                        N, lead=250 + 31 * c, seed=c) for c in range(Bp)])
    dmr, _, _ = _tiles("iq_dmr_t3_ras_cc.npz", 0, Bd, 1)
    nx, _, _ = _tiles("iq_nxdn48.npz", 60000, Bn, 1)
    m = ddn.MixedChainC(Bp, Bd, Bn, N)
    dp, dd, dn = _upload(p25), _upload(dmr[0]), _upload(nx[0])
    m.run(dp, dd, dn)
    m.wait()
    l = ddn.lib()
    # P25 group: the same call through a chain object of its own
    own = ddn.P25ChainC(Bp, N)
    own.run(dp)
    ro = own.results()
    rm = ddn.P25ChainResults()
    assert l.ddn_p25_chain_get_results(l.ddn_mixed_chain_part(m.h, 0), C.byref(rm)) == 0
    for name, dt, shape in (("d_new", np.int32, (Bp,)), ("d_nid4", np.int32, (Bp * own.F, 4)), ("d_tsbk", np.uint8, (3, Bp * own.F, 12)),
                            ("d_records10", np.uint8, (Bp, own.stride, 10))):
        assert np.array_equal(own.fetch(getattr(ro, name), dt, shape), own.fetch(getattr(rm, name), dt, shape)), name
    assert own.fetch(ro.d_tsbk_crc, np.uint8, (3, Bp * own.F)).sum() >= 30
    for which, Bc, proto, iq in ((1, Bd, ddn.FSK4_DMR, dd), (2, Bn, ddn.FSK4_NXDN48, dn)):
        a = m.part(which)
        b = ddn.Fsk4ChainC(Bc, N, proto, rf_mod=2 if which == 1 else 0)
        b.run(iq)
        ra, rb = a.results(), b.results()
        for name, dt, shape in (("d_counts", np.int32, (Bc,)), ("d_n_sync", np.int32, (Bc,)), ("d_records10", np.uint8, (Bc, ra.stride_symbols, 10)),
                                ("d_valid", np.uint8, (Bc * ra.max_syncs,))):
            assert np.array_equal(a.fetch(getattr(ra, name), dt, shape), b.fetch(getattr(rb, name), dt, shape)), (which, name)
        b.close()
    own.close()
    m.close()
    for p in (dp, dd, dn):
        l.ddn_device_free(p)


def test_mixed_chain_overlapped_schedule_equals_default(built, monkeypatch):
    """ddn_mixed_chain_config.overlap = 1 (front ends on streams of their own into two discriminator buffers per group, the next call's front ends
    beside this call's loops, fsk4 loops one channel per wavefront): five calls issued back to back without a wait give, array for
    array, what the default schedule gives for the same five calls - every group, the carried tails included"""
    import p25gen
    rng = np.random.default_rng(11)
    Bp, Bd, Bn, calls = 4, 3, 3, 4
    dib = [np.concatenate([p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=1 + (c + k) % 3)[0] for k in range(20 * calls)]) for c in range(Bp)]
    p25 = np.stack([p25gen.modulate_cu8(dib[c], N * calls, lead=250 + 31 * c, seed=c) for c in range(Bp)])
    dmr, _, _ = _tiles("iq_dmr_t3_ras_cc.npz", 0, Bd, 1)      # (the DMR capture is two seconds long: the same second in every call)
    nx, _, _ = _tiles("iq_nxdn48.npz", 60000, Bn, calls)

    def piece(k):
        return np.ascontiguousarray(p25[:, k * N:(k + 1) * N]), dmr[0], nx[k]

    l = ddn.lib()
    got = {}
    for mode in ("0", "1"):
        m = ddn.MixedChainC(Bp, Bd, Bn, N, overlap=int(mode))
        ptrs = []
        for k in range(calls):
            ps = [_upload(x) for x in piece(k)]
            ptrs += ps
            m.run(*ps)  # no wait: the host runs ahead, which is what lets the overlapped schedule overlap
        m.wait()
        out = {}
        rm = ddn.P25ChainResults()
        assert l.ddn_p25_chain_get_results(l.ddn_mixed_chain_part(m.h, 0), C.byref(rm)) == 0
        own = ddn.P25ChainC(Bp, N)  # (for its shapes only)
        for name, dt, shape in (("d_new", np.int32, (Bp,)), ("d_nid4", np.int32, (Bp * own.F, 4)), ("d_tsbk", np.uint8, (3, Bp * own.F, 12)),
                                ("d_records10", np.uint8, (Bp, own.stride, 10)), ("d_tsbk_crc", np.uint8, (3, Bp * own.F))):
            out["p25." + name] = own.fetch(getattr(rm, name), dt, shape)
        own.close()
        for which, Bc in ((1, Bd), (2, Bn)):
            a = m.part(which)
            ra = a.results()
            for name, dt, shape in (("d_counts", np.int32, (Bc,)), ("d_new", np.int32, (Bc,)), ("d_n_sync", np.int32, (Bc,)),
                                    ("d_records10", np.uint8, (Bc, ra.stride_symbols, 10)), ("d_flags", np.uint8, (Bc, ra.stride_symbols)),
                                    ("d_valid", np.uint8, (Bc * ra.max_syncs,)), ("d_sync_pos", np.int32, (Bc, ra.max_syncs))):
                out["%d.%s" % (which, name)] = a.fetch(getattr(ra, name), dt, shape)
        got[mode] = out
        m.close()
        for q in ptrs:
            l.ddn_device_free(q)
    assert got["0"].keys() == got["1"].keys()
    for name in got["0"]:
        a, b = got["0"][name], got["1"][name]
        if name.endswith("d_sync_pos") or name.endswith("d_records10") or name.endswith("d_flags"):
            continue  # compared below up to the counts (what lies beyond them is scratch)
        assert np.array_equal(a, b), name
    assert got["0"]["p25.d_tsbk_crc"].sum() >= 30 and got["0"]["1.d_n_sync"].sum() > 0
    for which, Bc in ((1, Bd), (2, Bn)):
        for c in range(Bc):
            k = int(got["0"]["%d.d_counts" % which][c])
            assert np.array_equal(got["0"]["%d.d_records10" % which][c, :k], got["1"]["%d.d_records10" % which][c, :k]), (which, c)
            assert np.array_equal(got["0"]["%d.d_flags" % which][c, :k], got["1"]["%d.d_flags" % which][c, :k]), (which, c)
            ns = int(got["0"]["%d.d_n_sync" % which][c])
            assert np.array_equal(got["0"]["%d.d_sync_pos" % which][c, :ns], got["1"]["%d.d_sync_pos" % which][c, :ns]), (which, c)
    assert np.array_equal(got["0"]["p25.d_records10"], got["1"]["p25.d_records10"])


def test_dmr_chain_voice_bursts_to_pcm(built):
    """DMR voice inside the chain object: the bursts the reference's BS voice handlers hand to the vocoder (dmrBSBootstrap / dmrBS,
    src/protocol/dmr/dmr_bs.c:585-640,697-760 - decided inside the receive loop, event kind 6 with VC >= 1) -> three AMBE 3600x2450
    frames each -> frame FEC -> synthesis, one talk path per time slot.  The reference's two DMR voice captures streamed in three
    calls + the flush against the CPU restatement of the whole stream: the same bursts in the same order on the same talk paths, the
    same AMBE frames, parameter bits and PCM bit for bit - bursts that straddle a call boundary included"""
    import mbe
    caps = ("iq_dmr_voice.npz", "iq_dmr_t3_cc.npz")
    B, n, calls = len(caps), 32000, 3
    iq = np.stack([np.ascontiguousarray(golden(c)["iq"], np.uint8)[:n * calls] for c in caps])
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_DMR, rf_mod=2)
    got = [[] for _ in range(2 * B)]           # per talk path: (stream position of the burst's last symbol, frames, bits, pcm)
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        vb, T = r.dmr_voice_bursts, int(r.carry_symbols)
        f = ch.fetch
        nv, vs = f(r.d_dmr_n_voice, np.int32, (2 * B,)), f(r.d_dmr_voice_start, np.int32, (2 * B, vb))
        fr = f(r.d_dmr_ambe_frames, np.uint8, (2 * B, vb, 3, 4, 24))
        bits, pcm = f(r.d_dmr_ambe_bits, np.uint8, (2 * B, vb * 3, 49)), f(r.d_dmr_pcm, np.float32, (2 * B, vb * 3, 160))
        res = f(r.d_dmr_ambe_result, np.int32, (2 * B, vb * 3, 5))
        skip = f(r.d_dmr_voice_skip, np.uint8, (2 * B, vb, 3))
        for tp in range(2 * B):
            assert nv[tp] <= vb and np.all(skip[tp, :nv[tp]] == 0) and np.all(skip[tp, nv[tp]:] == 0xFF)
            assert not pcm[tp, 3 * nv[tp]:].any()
            for k in range(int(nv[tp])):
                got[tp].append((int(base[tp // 2]) + int(vs[tp, k]) - T + 143, fr[tp, k].copy(), bits[tp, 3 * k:3 * k + 3].copy(),
                                pcm[tp, 3 * k:3 * k + 3].copy(), res[tp, 3 * k:3 * k + 3].copy()))
        return f(r.d_new, np.int32, (B,))

    for k in range(calls):
        d = _upload(np.ascontiguousarray(iq[:, k * n:(k + 1) * n]))
        ch.run(d)
        base += take() * 0
        new = ch.fetch(ch.results().d_new, np.int32, (B,))
        base += new
        ddn.lib().ddn_device_free(d)
    ch.flush()
    # (the flush decodes no voice: a burst is decoded in the call that holds its last symbol)
    r = ch.results()
    assert not ch.fetch(r.d_dmr_n_voice, np.int32, (2 * B,)).any()
    ch.close()
    total = 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(iq[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        o = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=2, handler=1))
        w = o.run(disc, max_sync=512)
        ev = [e for e in o.events.rows() if e[1] == 6 and e[3] >= 1]
        syncs = {int(p): i for i, p in enumerate(w["sync_pos"])}
        for slot in range(2):
            tp = 2 * c + slot
            mine = [e for e in ev if e[4] == slot]
            assert [g[0] for g in got[tp]] == [int(e[0]) for e in mine], (tp, [g[0] for g in got[tp]], [int(e[0]) for e in mine])
            voc = mbe.OracleVocoder(ddn.MBE_AMBE, 1)
            for g, e in zip(got[tp], mine):
                pos = int(e[0])
                dib = (w["rec4"][pos - 143:pos + 1, 0] & 3).astype(np.uint8)
                if pos - 54 in syncs:                  # the burst the sync search found: its first 90 dibits are the hand-over
                    dib[:90] = w["pre"][syncs[pos - 54]] & 3
                frames = rx4.dmr_voice_burst_fields(dib, np.zeros(144, np.uint8), 0)[0]
                assert np.array_equal(g[1], frames), (tp, pos)
                bits, res, rc = mbe.oracle_frame_decode(ddn.MBE_AMBE, frames)
                assert np.array_equal(g[2], bits), (tp, pos)
                pcm = np.zeros((1, 3, 160), np.float32)
                ro = np.zeros((1, 3, 5), np.int32)
                lb, lr = np.ascontiguousarray(bits[None]), np.ascontiguousarray(res[None])
                assert mbe._o().om_process_batch(ddn.MBE_AMBE, C.addressof(voc.tab), lb.ctypes.data, lr.ctypes.data, 0, tp, 1, 3,
                                                 pcm.ctypes.data, ro.ctypes.data, C.addressof(voc.cur), C.addressof(voc.prev),
                                                 C.addressof(voc.enh)) == 0
                assert np.array_equal(g[3].view(np.uint32), pcm[0].view(np.uint32)), (tp, pos, float(np.abs(g[3] - pcm[0]).max()))
                assert np.array_equal(g[4], ro[0]), (tp, pos)
                total += 1
    assert total >= 18, total


def _dmr_data_stream(rng):
    """a BS stream of data bursts (both time slots alternating): CSBKs until the colour-code gate locks, then every data type the
    handler treats differently -> (dibits, [(type, kwargs, what was sent)])"""
    import dmrgen
    plan = [(3, {})] * 8 + [(6, {}), (8, {}), (8, {}), (8, dict(confirmed=True, dbsn=0)), (8, dict(confirmed=True, dbsn=1)),
                            (8, dict(confirmed=True, dbsn=2, good_crc=False)), (7, {}), (7, dict(confirmed=True, dbsn=3)),
                            (10, dict(confirmed=True)), (1, {}), (2, {}), (1, dict(hurt=True)), (3, dict(good_crc=False)), (0, {}), (11, {}),
                            (4, {}), (5, {}), (9, {}), (6, dict(good_crc=False)), (3, {})]
    plan = plan + plan[8:]
    out, sent = [], []
    for k, (ty, kw) in enumerate(plan):
        kw = dict(kw)
        hurt = kw.pop("hurt", False)
        if ty == 8:
            s = dmrgen.r34_bytes(rng, **kw)
            info = dmrgen.r34_info(s)
        elif ty == 10:
            info = rng.integers(0, 2, 196).astype(np.uint8)
            info[96:100] = 0
            c = dmrgen.crc9_confirmed_rate1(info)
            info[7:16] = [(c >> (8 - i)) & 1 for i in range(9)]
            s = info.copy()
        else:
            s = dmrgen.payload_bits(ty, rng, **kw)
            t = s.copy()
            if hurt:
                t[16:24] ^= np.unpackbits(np.array([0xA5], np.uint8))          # one wrong byte: RS(12,9) repairs it
            info = dmrgen.bptc_196x96(t)
        out.append(dmrgen.burst(k & 1, 7, ty, info))
        sent.append((ty, kw, hurt, s))
    return np.concatenate(out), sent


def _dmr_voice_stream(rng, n_superframes=4):
    """CSBKs on both slots until the colour-code gate locks, then voice superframes on slot 1 (link control embedded in bursts B..E,
    the third one with a wrong checksum) beside idle data bursts on slot 2 -> (dibits, the link controls sent)"""
    import dmrgen
    out = [dmrgen.burst(k & 1, 7, 3, dmrgen.bptc_196x96(dmrgen.payload_bits(3, rng))) for k in range(8)]
    lcs = []
    for q in range(n_superframes):
        lc = rng.integers(0, 2, 72).astype(np.uint8)
        good = q != 2
        crc5 = None if good else (int(np.packbits(lc).astype(np.int64).sum()) % 31) ^ 0x0A
        lcs.append((lc, good))
        for b in dmrgen.voice_superframe(0, 7, lc, rng, crc5):
            out += [b, dmrgen.burst(1, 7, 9, dmrgen.bptc_196x96(rng.integers(0, 2, 96)))]
    return np.concatenate(out), lcs


def test_dmr_chain_data_bursts_link_control_rate34_and_embedded_lc(built):
    """The DMR chain's data-burst and embedded-signalling stages (include/ddn_chain.h: d_dmr_data_*, d_dmr_r34_*, d_dmr_emb_*): three of
    the reference's DMR captures and a synthetic stream with every data type, streamed in three calls + the flush, against the CPU
    restatement of the whole stream (tests/dmr_data.py): the same dispatched bursts in the same order with the same type, BPTC bits,
    RS(12,9)-repaired link control, CRC flags, rate 3/4 picks and candidate pools, and the same embedded link controls per talk path -
    bursts that straddle a call boundary included."""
    import dmr_data
    import p25gen
    rng = np.random.default_rng(23)
    caps = ("iq_dmr_t3_ras_cc.npz", "iq_dmr_voice.npz", "iq_dmr_t3_cc.npz")
    n, calls = 32000, 3
    dib, sent = _dmr_data_stream(rng)
    syn = p25gen.modulate_cu8(dib, n * calls, lead=300, seed=4, noise=0.02)
    vdib, lcs_sent = _dmr_voice_stream(rng)
    vsyn = p25gen.modulate_cu8(vdib, n * calls, lead=300, seed=5, noise=0.02)
    iq = np.stack([np.ascontiguousarray(golden(c)["iq"], np.uint8)[:n * calls] for c in caps] + [vsyn, syn])
    B = iq.shape[0]
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_DMR, rf_mod=2, vocoder=0)
    got = [[] for _ in range(B)]
    got_lc = [[] for _ in range(2 * B)]
    base = np.zeros(B, np.int64)
    for k in range(calls + 1):
        if k < calls:
            d = _upload(np.ascontiguousarray(iq[:, k * n:(k + 1) * n]))
            ch.run(d)
        else:
            ch.flush()
        r = ch.results()
        f = ch.fetch
        db, lb, T = r.dmr_data_bursts, r.dmr_emb_lcs, int(r.carry_symbols)
        nd, st = f(r.d_dmr_n_data, np.int32, (B,)), f(r.d_dmr_data_start, np.int32, (B, db))
        slot, ty = f(r.d_dmr_data_slot, np.uint8, (B, db)), f(r.d_dmr_data_type, np.uint8, (B, db))
        bits, by = f(r.d_dmr_data_bits96, np.uint8, (B, db, 96)), f(r.d_dmr_data_bytes12, np.uint8, (B, db, 12))
        info, errs = f(r.d_dmr_data_info196, np.uint8, (B, db, 196)), f(r.d_dmr_data_errs, np.uint32, (B, db))
        crc = f(r.d_dmr_data_crc, np.uint8, (B, db))
        un, co = f(r.d_dmr_r34_unconfirmed, np.uint8, (B, db, 18)), f(r.d_dmr_r34_confirmed, np.uint8, (B, db, 18))
        cc, pn = f(r.d_dmr_r34_confirmed_crc, np.uint8, (B, db)), f(r.d_dmr_r34_pool_n, np.int32, (B, db))
        pool = f(r.d_dmr_r34_pool, np.uint8, (B, db, 34, 24))
        ne, ep = f(r.d_dmr_n_emb, np.int32, (2 * B,)), f(r.d_dmr_emb_pos, np.int32, (2 * B, lb))
        lc, le, lo = f(r.d_dmr_emb_lc77, np.uint8, (2 * B, lb, 77)), f(r.d_dmr_emb_errs, np.uint32, (2 * B, lb)), f(r.d_dmr_emb_ok, np.uint8, (2 * B, lb))
        for c in range(B):
            assert nd[c] <= db and np.all(st[c, nd[c]:] == -1)
            for j in range(int(nd[c])):
                got[c].append(dict(pos=int(base[c]) + int(st[c, j]) - T + 143, slot=int(slot[c, j]), type=int(ty[c, j]), bits96=bits[c, j].copy(),
                                   bytes12=by[c, j].copy(), info=info[c, j].copy(), errs=int(errs[c, j]), crc=int(crc[c, j]),
                                   unconfirmed=un[c, j].copy(), confirmed=co[c, j].copy(), confirmed_crc=int(cc[c, j]),
                                   pool=pool[c, j, :int(pn[c, j])].copy()))
        for tp in range(2 * B):
            assert ne[tp] <= lb and np.all(ep[tp, ne[tp]:] == -1)
            for j in range(int(ne[tp])):
                got_lc[tp].append((int(base[tp // 2]) + int(ep[tp, j]) - T, lc[tp, j].copy(), int(le[tp, j]), int(lo[tp, j])))
        if k < calls:
            base += f(r.d_new, np.int32, (B,))
            ddn.lib().ddn_device_free(d)
        else:
            assert not nd.any() and not ne.any()         # (a burst is decoded in the call that holds its last symbol)
    ch.close()
    types_seen, n_lc, n_r34 = set(), 0, 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(iq[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        o = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_DMR, rf_mod=2, handler=1))
        w = o.run(disc, max_sync=512)
        want, want_lc = dmr_data.stream_expectation(w, o.events.rows())
        assert [g["pos"] for g in got[c]] == [p for p, _, _ in want], (c, len(got[c]), len(want))
        for g, (pos, slot, x) in zip(got[c], want):
            assert g["slot"] == slot and g["type"] == x["type"] and np.array_equal(g["info"], x["info"]), (c, pos)
            assert g["errs"] == x["errs"] and g["crc"] == x["crc"], (c, pos, g["type"], g["crc"], x["crc"])
            if not x["undefined"]:
                assert np.array_equal(g["bits96"], x["bits96"]) and np.array_equal(g["bytes12"], x["bytes12"]), (c, pos)
            types_seen.add(x["type"])
            if x["type"] == 8:
                n_r34 += 1
                assert np.array_equal(g["unconfirmed"], x["unconfirmed"]) and np.array_equal(g["confirmed"], x["confirmed"]), (c, pos)
                assert g["confirmed_crc"] == x["confirmed_crc"] and len(g["pool"]) == len(x["pool"]), (c, pos)
                for e, (b18, metric, ok9, dbsn) in zip(g["pool"], x["pool"]):
                    assert int(e[:4].copy().view(np.int32)[0]) == metric and np.array_equal(e[4:22], b18) and (e[22], e[23]) == (ok9, dbsn), (c, pos)
            else:
                assert len(g["pool"]) == 0
        for slot in range(2):
            tp = 2 * c + slot
            assert [g[0] for g in got_lc[tp]] == [x[0] for x in want_lc[slot]], (tp, got_lc[tp], want_lc[slot])
            for g, (pos, lc77, e, ok, undefined) in zip(got_lc[tp], want_lc[slot]):
                assert g[2] == e and (undefined or (np.array_equal(g[1], lc77) and g[3] == ok)), (tp, pos)
                n_lc += 1
        if c == B - 2:
            # the synthetic voice stream: the link controls sent come back on talk path 0, the one with the wrong checksum flagged;
            # the idle bursts of the other slot were read inside dmrBS() (no hand-over: all 144 dibits live)
            back = [(x[1][:72], x[3]) for x in want_lc[0]]
            assert len(back) >= len(lcs_sent) - 1 and not want_lc[1], (len(back), len(lcs_sent))
            for (lc72, ok), (s72, good) in zip(back, lcs_sent[len(lcs_sent) - len(back):]):
                assert np.array_equal(lc72, s72) and ok == (1 if good else 0)
            assert sum(1 for _, _, x in want if x["type"] == 9) >= 20
        if c == B - 1:
            # the synthetic stream: what the clean bursts carried comes back (the first bursts feed the colour-code gate)
            assert len(want) >= len(sent) - 8, (len(want), len(sent))
            tail = sent[len(sent) - len(want):]
            for (pos, slot, x), (ty, kw, hurt, s) in zip(want, tail):
                assert x["type"] == ty, (pos, x["type"], ty)
                if ty == 8:
                    key = "confirmed" if kw.get("confirmed") and kw.get("good_crc", True) else "unconfirmed"
                    assert np.array_equal(x[key], s), (pos, kw)
                elif ty == 10:
                    assert np.array_equal(x["info"], s) and x["crc"] == 3
                elif ty in (1, 2):
                    assert np.array_equal(np.unpackbits(x["bytes12"]), s) and x["crc"] == (5 if hurt else 1), (pos, ty, x["crc"])
                elif ty != 9:
                    assert np.array_equal(x["bits96"], s), (pos, ty)
                    if ty != 5:          # (an MBC continuation carries no CRC: the handler's compare against 0 says nothing)
                        assert (x["crc"] & 1) == (1 if kw.get("good_crc", True) or ty == 7 else 0), (pos, ty, kw)
    assert {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11} <= types_seen and n_r34 >= 10 and n_lc >= 4, (types_seen, n_r34, n_lc)
