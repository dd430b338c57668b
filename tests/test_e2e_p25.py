"""End to end on synthetic P25p1 TSDU traffic: cu8 IQ -> front end -> receive loop -> NID (BCH) -> 1/2-rate trellis.

CPU test: the whole chain on the oracles recovers NAC / DUID / every payload block that was sent.
GPU test: the same chain through the C-ABI gives, stage by stage, exactly the oracle's bytes.
"""
import numpy as np
import pytest

import fecgen
import orc
import p25gen
from test_oracle_block import oracle_nid

B, N = 3, 40000
NACS = [0x293, 0x5A1, 0x0F7]


def _traffic():
    rng = np.random.default_rng(77)
    iq = np.zeros((B, N, 2), np.uint8)
    states = []
    for c in range(B):
        nfr = N // (10 * p25gen.FRAME) + 1
        dib, st = p25gen.make_frames(rng, nfr, NACS[c])
        iq[c] = p25gen.modulate_cu8(dib, N, lead=200 + 37 * c, seed=c)
        states.append(st)
    return iq, states


def _check_decoded(c, acc, nid_out, blocks, states):
    """Every frame after the matched filter's turn-on transient decodes to what was sent."""
    want = p25gen.expected_half_rate_output(states)
    ok = 0
    for k in range(len(nid_out)):
        if k < 2:
            continue
        assert nid_out[k][0] == 1 and nid_out[k][1] == NACS[c] and nid_out[k][2] == p25gen.DUID_TSBK, (c, k, nid_out[k])
        hits = [f for f in range(want.shape[0]) if np.array_equal(want[f], blocks[k])]
        assert len(hits) == 1, (c, k)
        ok += 1
    assert ok >= 15


def _oracle_chain(iq):
    res = []
    for c in range(B):
        fe = orc.OracleFrontEnd()
        disc = fe.run_cu8(iq[c], 8192)
        rx = orc.OracleP25Rx(lock_symbols=p25gen.FRAME - 24, use_filter=1)
        sym, rec4, fl = rx.run(disc)
        acc, frames = p25gen.extract_frames(rec4, fl, len(sym))
        bits = np.stack([f[0] for f in frames])
        rel = np.stack([f[1] for f in frames])
        par = np.array([f[2] for f in frames], np.uint8)
        prel = np.array([f[3] for f in frames], np.uint8)
        llr = np.stack([f[4] for f in frames])
        nid = oracle_nid(bits, rel, np.zeros(len(frames), np.int32), par, prel)
        blocks, met = fecgen.oracle_p25_half_rate(np.ascontiguousarray(llr))
        res.append(dict(disc=disc, sym=sym, rec4=rec4, fl=fl, acc=acc, bits=bits, rel=rel, par=par, prel=prel, llr=llr,
                        nid=nid, blocks=blocks, met=met))
    return res


def test_e2e_oracle_chain_decodes_traffic(built):
    iq, states = _traffic()
    res = _oracle_chain(iq)
    for c in range(B):
        _check_decoded(c, res[c]["acc"], res[c]["nid"], res[c]["blocks"], states[c])


@pytest.mark.gpu
def test_e2e_gpu_chain_matches_oracle(built):
    import ctypes as C

    import ddn
    iq, states = _traffic()
    want = _oracle_chain(iq)
    fe = ddn.Batch(B, block_len=8192)
    disc = fe.run_host(iq, N)
    rx = ddn.P25Rx(B, lock_symbols=p25gen.FRAME - 24, use_matched_filter=1)
    rec, fl, cnt = rx.run(disc)
    for c in range(B):
        w = want[c]
        assert np.array_equal(disc[c].view(np.uint32), w["disc"].view(np.uint32)), c
        k = int(cnt[c])
        r4, sy = orc.unpack_records10(rec[c, :k])
        assert k == len(w["sym"]) and np.array_equal(sy.view(np.uint32), w["sym"].view(np.uint32)), c
        assert np.array_equal(r4, w["rec4"]) and np.array_equal(fl[c, :k], w["fl"]), c
        acc, frames = p25gen.extract_frames(r4, fl[c], k)
        n = len(frames)
        bits = np.ascontiguousarray(np.stack([f[0] for f in frames]))
        rel = np.ascontiguousarray(np.stack([f[1] for f in frames]))
        par = np.array([f[2] for f in frames], np.uint8)
        prel = np.array([f[3] for f in frames], np.uint8)
        llr = np.ascontiguousarray(np.stack([f[4] for f in frames]))
        obs = np.zeros(n, np.int32)
        nid = np.zeros((n, 4), np.int32)
        assert ddn.lib().ddn_p25p1_nid_decode_host(bits.ctypes.data, rel.ctypes.data, obs.ctypes.data, par.ctypes.data,
                                                   prel.ctypes.data, 64, n, nid.ctypes.data) == 0
        out = np.zeros((n, 12), np.uint8)
        met = np.zeros(n, np.int32)
        assert ddn.lib().ddn_fec_p25_12_soft_host(llr.ctypes.data, n, out.ctypes.data, met.ctypes.data) == 0
        assert np.array_equal(nid, w["nid"]) and np.array_equal(out, w["blocks"]) and np.array_equal(met, w["met"]), c
        _check_decoded(c, acc, nid, out, states[c])


def test_crc_valid_tsbk_generator(built):
    """p25gen.make_frames(crc=True): the decoded block is ten payload bytes + their CRC16 (checked by the oracle's CRC)."""
    import ctypes as C
    rng = np.random.default_rng(5)
    _, st = p25gen.make_frames(rng, 6, 0x123, crc=True)
    out = p25gen.expected_half_rate_output(st)
    o = orc.oracle()
    o.orc_p25_crc16_ok.argtypes = [C.c_void_p, C.c_int]
    for b in out:
        assert o.orc_p25_crc16_ok(np.ascontiguousarray(b).ctypes.data, 10) == 0
    _, st2 = p25gen.make_frames(rng, 6, 0x123)
    bad = [o.orc_p25_crc16_ok(np.ascontiguousarray(b).ctypes.data, 10) for b in p25gen.expected_half_rate_output(st2)]
    assert any(v != 0 for v in bad)
