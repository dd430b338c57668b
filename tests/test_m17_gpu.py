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
