"""BASELINE configs[2] with the vocoder on: mixed voice (LDU1 / LDU2) and control (TSDU) P25 Phase 1 channels, cu8 I/Q ->
PCM on the device (dsd-neo_amd/bindings/ddn_chain.py), against the chain of CPU oracles on the same bytes and against
what was transmitted."""
import ctypes as C

import numpy as np
import pytest

import chain_oracle
import ddn
import fecgen
import mbe
import orc
import p25gen
from test_oracle_block import oracle_nid

B, N, NAC = 6, 31000, 0x293
VOICE = [0, 1, 3, 4]


def _traffic():
    rng = np.random.default_rng(77)
    iq = np.zeros((B, N, 2), np.uint8)
    sent = {}
    for c in range(B):
        if c in VOICE:
            n_ldus = 4
            bits = mbe.random_imbe_bits(rng, (n_ldus * 9,))
            frames = np.stack([mbe.imbe_encode(b) for b in bits])
            dib, words = p25gen.make_ldus(rng, n_ldus, NAC, frames)
            sent[c] = dict(bits=bits, words=words)
        else:
            dib, _ = p25gen.make_frames(rng, N // 1800 + 1, NAC, crc=True)
        iq[c] = p25gen.modulate_cu8(dib, N, lead=300 + 37 * c, seed=c)
    lock = np.array([840 if c in VOICE else p25gen.FRAME - 24 for c in range(B)], np.int32)
    return iq, lock, sent


def _oracle_chain(iq, lock, Fv):
    """tests/chain_oracle.py per channel; talk path c is seeded with its channel index like the device batch"""
    out = []
    for c in range(B):
        w = chain_oracle.run_channel(iq[c], lock[c], Fv, seed=c)
        w["imbe_res"][w["skip"], 0] |= np.int32(-2147483648)
        out.append(w)
    return out


def _oracle_pcm(want, Fv):
    return np.stack([w["pcm"] for w in want])


def _sent_offset(w, sent):
    """index of the transmitted LDU that the first detected LDU is (frames before it were lost to the start-up)"""
    m = [j for j in range(0, len(sent["bits"]), 9) if np.array_equal(w["imbe_d"][9], sent["bits"][j])]
    assert len(m) == 1 and m[0] >= 9
    return m[0] // 9 - 1


def test_ldu_generator_decodes_on_the_oracle_chain(built):
    """The synthetic LDU traffic is valid P25: the CPU chain finds every LDU's NID, and the voice frames it extracts carry
    the parameter bits that were sent with zero corrections."""
    iq, lock, sent = _traffic()
    want = _oracle_chain(iq, lock, 5)
    for c in VOICE:
        good = [n for n in want[c]["nids"] if n is not None and n[0] == 1]
        assert len(good) >= 3 and all(n[1] == NAC and n[2] in (5, 10) for n in good)
        k = int((~want[c]["skip"]).sum())
        assert k >= 18
        off = _sent_offset(want[c], sent[c])
        # the channel's very first frame is lost to the matched filter's turn-on (no sync yet -> no filter) and the first
        # LDU that is found still sees the thresholds settle; from the next one on the frames are clean
        # (the smoothed-FM test modulator leaves a raw dibit error rate of a few 1e-3: most are corrected, a few land in
        # the seven unprotected bits of c7)
        diff = want[c]["imbe_d"][9:k] != sent[c]["bits"][off * 9 + 9:off * 9 + k]
        assert diff.mean() < 0.004 and (diff.sum(axis=1) == 0).mean() > 0.7, (c, diff.sum(axis=1))
        assert np.all(want[c]["imbe_res"][9:k, 3] <= 6)
    for c in set(range(B)) - set(VOICE):
        assert want[c]["n_ldu"] == 0


@pytest.mark.gpu
def test_voice_chain_on_device_equals_oracle_chain(built):
    import torch
    import ddn_chain
    iq, lock, sent = _traffic()
    ch = ddn_chain.P25Chain(torch, B, N, lock)
    want = _oracle_chain(iq, lock, ch.Fv)
    d_iq = torch.from_numpy(iq).cuda()
    ch.run(d_iq)
    torch.cuda.synchronize()
    nid = ch.nid.cpu().numpy().reshape(B, ch.F, 4)
    n_ldu = ch.n_ldu.cpu().numpy()
    imbe_d = ch.imbe_d.cpu().numpy().reshape(B, ch.Fv * 9, 88)
    res = ch.imbe_res.cpu().numpy().reshape(B, ch.Fv * 9, 5)
    pcm = ch.pcm.cpu().numpy()
    werrs = [w.cpu().numpy().reshape(B, ch.F, 24) for w in ch.werrs]
    rs_st = [r.cpu().numpy().reshape(B, ch.F) for r in ch.rs_st]
    rs_d = [r.cpu().numpy().reshape(B, ch.F, -1, 6) for r in ch.rs_d]
    crc_ok = ch.crc_ok.cpu().numpy().reshape(B, ch.F)
    v_ldu = ch.v_ldu.cpu().numpy().reshape(B, ch.F)          # 1 = the slot's whole LDU lies inside this call's records
    want_pcm = _oracle_pcm(want, ch.Fv)
    for c in range(B):
        w = want[c]
        for k, n in enumerate(w["nids"]):
            if n is not None:
                assert np.array_equal(nid[c, k], n), (c, k)
        assert n_ldu[c] == w["n_ldu"]
        live = ~w["skip"]
        assert np.array_equal(imbe_d[c][live], w["imbe_d"][live]) and np.array_equal(res[c][live], w["imbe_res"][live]), c
        assert np.all(res[c][~live, 0].view(np.uint32) & 0x80000000)
        assert np.array_equal(pcm[c].view(np.uint32), want_pcm[c].view(np.uint32)), c
        if c in VOICE:
            assert np.abs(pcm[c]).max() > 0
            # link control / encryption sync words: Hamming corrects what the channel flipped, Reed-Solomon accepts, data symbols = what was sent
            seen, seen_any = 0, False                        # the first LDU found is still settling: skip it
            ldu_no = _sent_offset(w, sent[c])
            for k, n in enumerate(w["nids"]):
                if n is None or n[0] != 1 or n[2] not in (5, 10):
                    continue
                i = 0 if n[2] == 5 else 1
                if seen_any and v_ldu[c, k] and ldu_no < len(sent[c]["words"]) and (ldu_no & 1) == i:
                    nd = 12 if i == 0 else 16
                    assert rs_st[i][c, k] == 0 and werrs[i][c, k].max() <= 1, (c, k)   # Hamming fixes the odd channel error
                    got = (rs_d[i][c, k] * (1 << np.arange(5, -1, -1))[None, :]).sum(axis=1)
                    assert np.array_equal(got, sent[c]["words"][ldu_no][:nd]), (c, k)
                    seen += 1
                seen_any = True
                ldu_no += 1
            assert seen >= 1
        else:
            assert np.all(pcm[c] == 0) and crc_ok[c].sum() >= 10
    ch.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams", [2, 3, -2])
def test_pipelined_steps_equal_one_stream_steps(built, n_streams):
    """P25Chain.run_pipelined (frame FEC + vocoder of batch k on a second stream beside the next batch's front end + receive
    loop, double-buffered loop outputs) and run_pipelined3 (the front end on a third stream as well, double-buffered
    discriminator) produce, batch after batch, exactly what run() produces on one stream - records, counts, NIDs, TSBKs,
    voice parameter bits, result flags and PCM, with the state carried across four batch intervals."""
    import torch
    import ddn_chain
    iq, lock, _ = _traffic()
    # four consecutive intervals of one stream per channel: the traffic cut in four pieces
    n = N // 4
    d_iq = [torch.from_numpy(np.ascontiguousarray(iq[:, k * n:(k + 1) * n])).cuda() for k in range(4)]
    a = ddn_chain.P25Chain(torch, B, n, lock, block_len=4096)
    b = ddn_chain.P25Chain(torch, B, n, lock, block_len=4096)
    s1, s2, s0 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    want = []
    for k in range(4):
        a.run(d_iq[k])
        torch.cuda.synchronize()
        want.append([t.clone() for t in (a.rec, a.fl, a.cnt, a.nid, a.tsbk, a.crc_ok, a.imbe_d, a.res_out, a.pcm)])
    got = []
    for k in range(4):          # no host synchronisation between the calls: the overlap is real
        if n_streams == 3:
            b.run_pipelined3(d_iq[k], s0, s1, s2)
        elif n_streams == -2:      # decode of batch k - 1 is queued by this call, after this batch's front end
            b.run_pipelined_deferred(d_iq[k], s1, s2)
            if k >= 1:
                with torch.cuda.stream(s2):
                    rec, fl, cnt = b.sets[(k - 1) & 1]
                    got.append([t.clone() for t in (rec, fl, cnt, b.nid, b.tsbk, b.crc_ok, b.imbe_d, b.res_out, b.pcm)])
            if k == 3:
                b.flush(s1, s2)
                with torch.cuda.stream(s2):
                    got.append([t.clone() for t in (b.rec, b.fl, b.cnt, b.nid, b.tsbk, b.crc_ok, b.imbe_d, b.res_out, b.pcm)])
            continue
        else:
            b.run_pipelined(d_iq[k], s1, s2)
        # snapshot on the consumer stream, ordered after this batch's last stage
        with torch.cuda.stream(s2):
            got.append([t.clone() for t in (b.rec, b.fl, b.cnt, b.nid, b.tsbk, b.crc_ok, b.imbe_d, b.res_out, b.pcm)])
    torch.cuda.synchronize()
    names = ("rec", "fl", "cnt", "nid", "tsbk", "crc_ok", "imbe_d", "res_out", "pcm")
    for k in range(4):
        cnt = want[k][2].cpu().numpy()
        for name, x, y in zip(names, want[k], got[k]):
            if name in ("rec", "fl"):      # entries past a channel's count are leftovers of whichever batch used the buffer before
                for c in range(B):
                    assert torch.equal(x[c, :cnt[c]], y[c, :cnt[c]]), (k, name, c)
            else:
                assert torch.equal(x, y), (k, name, int((x != y).sum()))
    assert int(want[3][2].sum()) > 0 and float(want[2][8].abs().sum()) > 0
