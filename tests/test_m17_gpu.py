"""M17 on the device past the loop: the thresholds every sync leaves (ddn_fsk4_rx_set_sync_thresholds) and the link setup frame
decode (ddn_m17_lsf_decode_batch: soft costs with the host libm's expf -> de-randomise -> de-interleave -> P1 -> the K = 5 decoder of
SURVEY row a17 -> CRC16) against the CPU restatement, on transmissions built by the reference's own encoder."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import rx4

pytestmark = pytest.mark.gpu


def _device_loop(x):
    """x [B][n] float32 -> the DDN_FSK4_M17 loop's device outputs (torch tensors) + the per-sync thresholds"""
    import torch
    l = ddn.lib()
    B, n = x.shape
    d = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rx = ddn.Fsk4Rx(B, ddn.FSK4_M17)
    ms, my = l.ddn_fsk4_rx_max_symbols(rx.h, n), l.ddn_fsk4_rx_max_syncs(rx.h, n)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    rec, fl, pay = z((B, ms, 10), torch.uint8), z((B, ms), torch.uint8), z((B, ms, 2), torch.uint8)
    cnt, ns, spos = z((B,), torch.int32), z((B,), torch.int32), z((B, my), torch.int32)
    spat, pre, prel = z((B, my), torch.uint8), z((B, my, 90), torch.uint8), z((B, my, 90), torch.uint8)
    thr = z((B, my, 5), torch.float32)
    p = lambda t: t.data_ptr()
    assert l.ddn_fsk4_rx_set_sync_thresholds(rx.h, p(thr)) == 0
    assert l.ddn_fsk4_rx_run(rx.h, p(d), n, p(rec), p(fl), p(pay), p(cnt), ms, p(spos), p(spat), p(pre), p(prel), p(ns), my, None) == 0
    torch.cuda.synchronize()
    return dict(rx=rx, rec=rec, cnt=cnt, ns=ns, spos=spos, spat=spat, thr=thr, ms=ms, my=my, keep=(d, fl, pay, pre, prel))


def test_sync_thresholds_and_lsf_decode_on_the_device(built):
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    import torch
    import m17
    import p25gen
    from test_oracle_m17 import two_transmissions
    l = ddn.lib()
    xs, sent_lsf = [], []
    for gap, seed, noise in (([1, 3, 1], 5, 0.02), ([1, 3, 1], 9, 0.12), ([], 1, 0.02)):
        d, by, _ = two_transmissions(gap, seed)
        iq = p25gen.modulate_cu8(d, len(d) * 10 + 1200, lead=20, seed=seed, noise=noise)
        xs.append(orc.OracleFrontEnd(profile=2).run_cu8(iq, 8192))
        sent_lsf.append(by)
    n = min(len(v) for v in xs)
    x = np.stack([v[:n] for v in xs] + [-xs[0][:n]])
    B = x.shape[0]
    g = _device_loop(x)
    my = g["my"]
    wants = [rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(x[c], max_sync=my) for c in range(B)]
    ns = g["ns"].cpu().numpy()
    thr = g["thr"].cpu().numpy()
    for c in range(B):
        assert int(ns[c]) == len(wants[c]["sync_pos"]) and np.array_equal(g["spos"].cpu().numpy()[c, :ns[c]], wants[c]["sync_pos"])
        assert np.array_equal(thr[c, :ns[c]].view(np.uint32), wants[c]["sync_thr"].view(np.uint32)), c   # the thresholds every sync left
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    lsf, st, pc = z((B, my, 30), torch.uint8), z((B, my), torch.uint8), z((B, my), torch.int32)
    p = lambda t: t.data_ptr()
    assert l.ddn_m17_lsf_decode_batch(p(g["rec"]), g["ms"], p(g["cnt"]), p(g["spos"]), p(g["spat"]), p(g["ns"]), p(g["thr"]), B, my, p(lsf),
                                      p(st), p(pc), None) == 0, l.ddn_last_error()
    torch.cuda.synchronize()
    lsf, st, pc = lsf.cpu().numpy(), st.cpu().numpy(), pc.cpu().numpy().view(np.uint32)
    good = 0
    for c in range(B):
        fr = m17.decode_stream(wants[c])
        by_pos = {f["pos"]: f for f in fr if f["kind"] == "lsf"}
        for k in range(int(ns[c])):
            pos = int(wants[c]["sync_pos"][k])
            if pos in by_pos:      # an LSF sync whose frame is complete: bytes, CRC verdict and path cost of the restatement
                f = by_pos[pos]
                assert st[c, k] == (2 if f["crc_ok"] else 1), (c, k)
                assert np.array_equal(lsf[c, k], f["lsf30"]) and int(pc[c, k]) == int(f["cost"]), (c, k)
                good += int(f["crc_ok"])
            else:
                assert st[c, k] == 0, (c, k)
    assert good >= 3
    named = 0
    for c in range(3):
        for k in np.flatnonzero(st[c] == 2):
            assert np.array_equal(lsf[c, k], sent_lsf[c])                  # what the reference's encoder sent
            assert m17.callsign(int.from_bytes(bytes(lsf[c, k, 6:12].tolist()), "big"))[1] == "N0CALL"
            named += 1
    assert named >= 3


def _decode_all(g, B):
    """LSF + stream frames + LICH reassembly of one loop call on the device -> numpy arrays per sync slot"""
    import torch
    l = ddn.lib()
    my = g["my"]
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    p = lambda t: t.data_ptr()
    o = dict(lsf=z((B, my, 30), torch.uint8), lsf_st=z((B, my), torch.uint8), l6=z((B, my, 6), torch.uint8), cnt=z((B, my), torch.uint8),
             fp=z((B, my, 18), torch.uint8), st=z((B, my), torch.uint8), asm=z((B, 32), torch.uint8), ll=z((B, my, 30), torch.uint8),
             ll_st=z((B, my), torch.uint8))
    assert l.ddn_m17_lsf_decode_batch(p(g["rec"]), g["ms"], p(g["cnt"]), p(g["spos"]), p(g["spat"]), p(g["ns"]), p(g["thr"]), B, my, p(o["lsf"]),
                                      p(o["lsf_st"]), None, None) == 0, l.ddn_last_error()
    assert l.ddn_m17_str_decode_batch(p(g["rec"]), g["ms"], p(g["cnt"]), p(g["spos"]), p(g["spat"]), p(g["ns"]), B, my, p(o["l6"]), p(o["cnt"]),
                                      p(o["fp"]), p(o["st"]), None) == 0, l.ddn_last_error()
    assert l.ddn_m17_lich_assemble_batch(p(g["spat"]), p(g["ns"]), B, my, p(o["lsf"]), p(o["lsf_st"]), p(o["l6"]), p(o["cnt"]), p(o["st"]),
                                         p(o["asm"]), p(o["ll"]), p(o["ll_st"]), None) == 0, l.ddn_last_error()
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}


def _check_against_python_decode(o, c, want, ns):
    import m17
    fr = {f["pos"]: f for f in m17.decode_stream(want)}
    n_str = n_fin = 0
    for k in range(ns):
        f = fr[int(want["sync_pos"][k])]
        if f["kind"] == "str":
            assert o["st"][c, k] == (2 if f["lich_err"] == 0 else 1), (c, k)
            assert np.array_equal(o["l6"][c, k], f["lich6"]) and int(o["cnt"][c, k]) == f["cnt"], (c, k)
            fp = o["fp"][c, k]
            if f["lich_err"] == 0:
                assert ((int(fp[0]) << 8) | int(fp[1])) == f["fn"] and np.array_equal(fp[2:], f["payload"]), (c, k)
                n_str += 1
            else:
                assert not fp.any()
            if "lich_lsf30" in f:
                assert o["ll_st"][c, k] == (2 if f["lich_crc_ok"] else 1) and np.array_equal(o["ll"][c, k], f["lich_lsf30"]), (c, k)
                n_fin += 1
            else:
                assert o["ll_st"][c, k] == 0
        else:
            assert o["st"][c, k] == 0 and o["ll_st"][c, k] == 0, (c, k)
    return n_str, n_fin


def test_stream_frames_and_lich_reassembly_on_the_device(built):
    """the reference's M17 capture and transmissions of its encoder: every stream frame's LICH (Golay words, chunk counter), frame number
    and payload (P2 + the NXDN-style K = 5 decoder), and the LSF reassembled from six chunks with its CRC verdict, equal the Python
    decode of the restatement's loop output slot for slot; the capture's known answer - SRC N0CALL - from device arrays alone"""
    import m17
    import p25gen
    disc = rx4.capture_disc("iq_m17.npz", 2)
    xs = [disc, -disc, np.roll(disc, 5)]
    if orc.ref() is not None:
        from test_oracle_m17 import two_transmissions
        d, by, sent = two_transmissions([1, 3, 1], 5)
        iq = p25gen.modulate_cu8(d, len(d) * 10 + 1200, lead=20, seed=5, noise=0.02)
        syn = orc.OracleFrontEnd(profile=2).run_cu8(iq, 8192)
        pad = np.zeros(len(disc), np.float32)
        pad[:min(len(syn), len(disc))] = syn[:len(disc)]
        xs.append(pad)
    x = np.stack(xs)
    B = x.shape[0]
    g = _device_loop(x)
    o = _decode_all(g, B)
    ns = g["ns"].cpu().numpy()
    tot_str = tot_fin = 0
    for c in range(B):
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(x[c], max_sync=g["my"])
        assert int(ns[c]) == len(want["sync_pos"])
        a, b = _check_against_python_decode(o, c, want, int(ns[c]))
        tot_str += a
        tot_fin += b
    assert tot_str >= 70 and tot_fin >= 10
    # the capture's known answer from the device arrays: every reassembled LSF that passes its CRC names N0CALL
    good = np.flatnonzero(o["ll_st"][0] == 2)
    assert len(good) >= 5
    assert {m17.callsign(int.from_bytes(bytes(o["ll"][0, k, 6:12].tolist()), "big"))[1] for k in good} == {"N0CALL"}


def test_m17_capture_through_the_chain_object_src_n0call(built):
    """ddn_fsk4_chain with protocol M17: cu8 I/Q of the reference's capture in three calls + the flush -> front end (12.5 kHz filter) ->
    loop -> the frames behind every sync, decoded in the call that holds their last symbol (frames that cross a call boundary wait in
    the carried tail; the LICH assembly buffer and the syncs' thresholds are carried with them) = the Python decode of the CPU
    pipeline's whole-stream output, frame for frame; SRC N0CALL"""
    import m17
    from conftest import golden
    iq = np.ascontiguousarray(golden("iq_m17.npz")["iq"], np.uint8)
    n = 40000
    calls = len(iq) // n
    assert calls >= 3
    B = 2
    x = np.stack([iq[:calls * n], np.roll(iq[:calls * n], 2 * 77)])     # channel 1: the capture 77 samples later
    ch = ddn.Fsk4ChainC(B, n, ddn.FSK4_M17, rf_mod=0, handlers=0, vocoder=0)
    l = ddn.lib()
    got = [[] for _ in range(B)]
    base = np.zeros(B, np.int64)

    def take():
        r = ch.results()
        S, T = r.max_syncs, r.carry_symbols
        f = ch.fetch
        ns, pos, pat = f(r.d_n_sync, np.int32, (B,)), f(r.d_sync_pos, np.int32, (B, S)), f(r.d_sync_pat, np.uint8, (B, S))
        lsf, lst = f(r.d_m17_lsf30, np.uint8, (B, S, 30)), f(r.d_m17_lsf_status, np.uint8, (B, S))
        l6, cnt = f(r.d_m17_lich6, np.uint8, (B, S, 6)), f(r.d_m17_lich_cnt, np.uint8, (B, S))
        fp, st = f(r.d_m17_fn_payload18, np.uint8, (B, S, 18)), f(r.d_m17_str_status, np.uint8, (B, S))
        ll, lls = f(r.d_m17_lich_lsf30, np.uint8, (B, S, 30)), f(r.d_m17_lich_status, np.uint8, (B, S))
        new = f(r.d_new, np.int32, (B,))
        for c in range(B):
            for k in range(int(ns[c])):
                got[c].append(dict(pos=int(base[c]) + int(pos[c, k]) - int(T), pat=int(pat[c, k]), lsf=lsf[c, k].copy(), lst=int(lst[c, k]),
                                   l6=l6[c, k].copy(), cnt=int(cnt[c, k]), fp=fp[c, k].copy(), st=int(st[c, k]), ll=ll[c, k].copy(),
                                   lls=int(lls[c, k])))
            base[c] += int(new[c])

    for k in range(calls):
        part = np.ascontiguousarray(x[:, k * n:(k + 1) * n])
        p = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, part.ctypes.data, part.nbytes) == 0
        ch.run(p)
        take()
        l.ddn_device_free(p)
    ch.flush()   # (no new records: the row is the last call's carried tail re-based, positions count on from the same base)
    take()
    ch.close()
    n_named = 0
    for c in range(B):
        fe = orc.OracleFrontEnd(profile=2)
        disc = np.concatenate([fe.run_cu8(np.ascontiguousarray(x[c, k * n:(k + 1) * n]), 8192) for k in range(calls)])
        want = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(disc, max_sync=4096)
        fr = [f for f in m17.decode_stream(want) if f["kind"] != "cut"]
        # every sync of the stream is decoded exactly once, in order (the flush also hands out a frame sync whose frame the stream's end
        # cut short - all statuses 0 -, which the Python decode files as "cut")
        total = len(want["sym"])
        mine = [g for g in got[c] if g["pat"] < 4 or g["pos"] + 185 <= total]
        assert all(g["lst"] == 0 and g["st"] == 0 for g in got[c] if not (g["pat"] < 4 or g["pos"] + 185 <= total))
        assert [g["pos"] for g in mine] == [f["pos"] for f in fr], (c, len(mine), len(fr))
        for g, f in zip(mine, fr):
            assert g["pat"] == f["pat"]
            if f["kind"] == "lsf":
                assert g["lst"] == (2 if f["crc_ok"] else 1) and np.array_equal(g["lsf"], f["lsf30"])
            elif f["kind"] == "str":
                assert g["st"] == (2 if f["lich_err"] == 0 else 1) and np.array_equal(g["l6"], f["lich6"]) and g["cnt"] == f["cnt"]
                if f["lich_err"] == 0:
                    assert ((int(g["fp"][0]) << 8) | int(g["fp"][1])) == f["fn"] and np.array_equal(g["fp"][2:], f["payload"])
                if "lich_lsf30" in f:
                    assert g["lls"] == (2 if f["lich_crc_ok"] else 1) and np.array_equal(g["ll"], f["lich_lsf30"])
                    if f["lich_crc_ok"]:
                        assert m17.callsign(int.from_bytes(bytes(g["ll"][6:12].tolist()), "big"))[1] == "N0CALL"
                        n_named += 1
                else:
                    assert g["lls"] == 0
            else:
                assert g["lst"] == 0 and g["st"] == 0 and g["lls"] == 0
    assert n_named >= 8
