"""GPU: the P25 Phase 2 chain object (ddn_p25p2_chain): cu8 I/Q of the reference's own Phase 2 capture in, MAC PDUs out, device kernels
only - CQPSK demodulator at 6000 symbols/s -> symbol-rate loop -> groups behind every sync -> processP2().  Checked against the CPU
pipeline of the same stages (oracle CQPSK front end + oracle/ddn_oracle_cqrx.c + tests/p2seq.py) group for group across call
boundaries, and against the known answer: "P25p2 SACCH" (DECODE_IQ_P25P2_CC, tests/CMakeLists.txt:8923) = the SACCH MAC PDUs this
capture is known to carry (tests/test_oracle_p25p2_capture.py).  Voice: a synthetic channel (two logical channels of 4V / 2V bursts)
through the same object to PCM, against the CPU vocoder restatement fed the same frames."""
import ctypes as C

import numpy as np
import pytest

import ddn
import orc
import p2capture
import p2seq
from conftest import golden
from test_oracle_p25p2_capture import SACCH_OCTETS

pytestmark = pytest.mark.gpu
SEED = p2capture.WACN * 16777216 + p2capture.SYSID * 4096 + p2capture.NAC


def _upload(a):
    p = C.c_void_p()
    assert ddn.lib().ddn_device_alloc(a.nbytes, C.byref(p)) == 0
    assert ddn.lib().ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
    return p


def oracle_groups(iq, n_call):
    """-> (symbols, rec4, flags, sync positions, bits [G][1400], llr [G][1400]) of the whole stream, front end block structure per call"""
    x = ((iq.astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    fe = orc.OracleCqpskFe(rate=48000, sym_rate=6000)
    sym = np.concatenate([fe.run(x[k:k + n_call], 8192) for k in range(0, len(x), n_call)])
    rec, fl = orc.OracleCqRx(orc.CQ_P25P2).run(sym)
    pos = [int(p) for p in np.flatnonzero(fl & 2) if p + 700 < len(sym)]
    bits = np.zeros((len(pos), 1400), np.uint8)
    llr = np.zeros((len(pos), 1400), np.int16)
    for k, p in enumerate(pos):
        d = rec[p + 1:p + 701]
        bits[k, 0::2], bits[k, 1::2] = (d[:, 0] >> 1) & 1, d[:, 0] & 1
        llr[k, 0::2], llr[k, 1::2] = d[:, 2], d[:, 3]
    return sym, rec, fl, pos, bits, llr


@pytest.mark.parametrize("n_call", [48000, 32000])
def test_capture_from_iq_to_sacch_pdus(built, n_call):
    iq = np.ascontiguousarray(golden("iq_p25p2_cc.npz")["iq"])
    n_total = (len(iq) // n_call) * n_call
    iq = iq[:n_total]
    sym, rec, fl, pos, wb, wl = oracle_groups(iq, n_call)
    assert len(pos) >= 10
    want = p2seq.run_groups(wb, wl, p2capture.WACN, p2capture.SYSID, p2capture.NAC, p2seq.new_state())
    ch = ddn.P25P2ChainC([SEED, SEED, 0], n_call, vocoder=0)
    got_pos, rows, base = [], [], 0
    for k in range(n_total // n_call + 1):
        if k < n_total // n_call:
            d = _upload(np.ascontiguousarray(np.broadcast_to(iq[k * n_call:(k + 1) * n_call], (3, n_call, 2))))
            ch.run(d)
            ddn.lib().ddn_device_free(d)
        else:
            ch.flush()
        r = ch.results()
        G, T = r.max_groups, r.carry_symbols
        ng = ch.fetch(r.d_n_groups, np.int32, (3,))
        gp = ch.fetch(r.d_group_pos, np.int32, (3, G))
        info = ch.fetch(r.d_info, np.int32, (3, G, 4, 8))
        pay = ch.fetch(r.d_payload, np.uint8, (3, G, 4, 180))
        new = ch.fetch(r.d_new, np.int32, (3,))
        assert ng[0] == ng[1] == ng[2]
        for g in range(int(ng[0])):
            got_pos.append(base + int(gp[0, g]) - T)
            rows.append((info[0, g].copy(), pay[0, g].copy(), info[1, g].copy(), pay[1, g].copy(), info[2, g].copy()))
        base += int(new[0])
    ch.close()
    assert base == len(sym)
    assert got_pos == pos, (got_pos, pos)
    octets = []
    for g, (i0, p0, i1, p1, i2) in enumerate(rows):
        for ts in range(4):
            w = want[4 * g + ts]
            assert (i0[ts, 0], i0[ts, 1], i0[ts, 2], i0[ts, 3], i0[ts, 4]) == (w["duid"], w["isch"], w["offset"], w["slot"], w["action"]), (g, ts)
            assert np.array_equal(i0[ts], i1[ts]) and np.array_equal(p0[ts], p1[ts])
            if w["action"] in (p2seq.A_SACCH_S, p2seq.A_SACCH_C, p2seq.A_FACCH_C, p2seq.A_FACCH_S, p2seq.A_LCCH_C, p2seq.A_LCCH_S):
                assert i0[ts, 5] == w["ec"] and np.array_equal(p0[ts], w["payload"]), (g, ts)
            if i0[ts, 4] == p2seq.A_SACCH_S:
                assert i0[ts, 7] & 2                                     # CRC-12 good
                octets.append(bytes(np.packbits(p0[ts])[:12]).hex())
                assert i2[ts, 4] == p2seq.A_NOSITE                       # the channel without a valid site skips the scrambled burst
    assert len(octets) >= 6 and all(o in SACCH_OCTETS for o in octets)  # "P25p2 SACCH": the capture's known PDUs
    # in order: a subsequence of the known list
    it = iter(SACCH_OCTETS)
    assert all(any(o == k for k in it) for o in octets), octets
