"""GPU: the P25 Phase 1 chain object with modulation = CQPSK (ddn_p25_chain_config.modulation): cu8 I/Q of the reference's CQPSK / LSM
captures in, decoded frames out, device kernels only - CQPSK demodulator -> symbol-rate loop with the handlers inside -> framer ->
frame FEC -> IMBE -> PCM.  Checked (a) record for record, flag for flag, decision for decision against the whole-stream oracle
(oracle/ddn_oracle_cqpsk.c + ddn_oracle_cqrx.c) with the stream handed over in several calls, and (b) against the known answers the
reference's own full-chain tests expect of these captures (tests/CMakeLists.txt:8901-8918)."""
import ctypes as C

import numpy as np
import pytest

import chain_stream
import ddn
import orc
from conftest import golden

pytestmark = pytest.mark.gpu


def _upload(a):
    p = C.c_void_p()
    assert ddn.lib().ddn_device_alloc(a.nbytes, C.byref(p)) == 0
    assert ddn.lib().ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
    return p


def run_chain(iq, n_call):
    """iq cu8 [n][2] of one channel (tiled to 3 channels: the rows must not see each other) -> Collector after calls + flush"""
    B = 3
    n_total = (len(iq) // n_call) * n_call
    ch = ddn.P25ChainC(B, n_call, block_len=8192, modulation=1)
    col = chain_stream.Collector(ch, everything=True)
    for k in range(n_total // n_call):
        part = np.ascontiguousarray(np.broadcast_to(iq[k * n_call:(k + 1) * n_call], (B, n_call, 2)))
        d = _upload(part)
        ch.run_pipelined(d)
        ch.wait()
        col.take()
        ddn.lib().ddn_device_free(d)
    ch.flush()
    col.take()
    ch.close()
    return col, n_total


def oracle_stream(iq, n_total, n_call):
    x = ((iq[:n_total].astype(np.float32) - 127.5) * np.float32(1.0 / 127.5)).astype(np.float32)
    fe = orc.OracleCqpskFe(rate=48000)          # a call = consecutive full_demod() blocks of 8192 + a shorter last one, like the chain's
    sym = np.concatenate([fe.run(x[k:k + n_call], 8192) for k in range(0, n_total, n_call)])
    rx = orc.OracleCqRx(orc.CQ_P25P1)
    rec, fl = rx.run(sym)
    return sym, rec, fl, rx.events.rows(), rx.events.data()


def check_stream(col, sym, rec, fl, rows, data):
    for c in range(3):
        got = np.concatenate(col.rec[c])
        gfl = np.concatenate(col.fl[c])
        assert len(got) == len(sym), (c, len(got), len(sym))
        r4, sy = orc.unpack_records10(got)
        assert np.array_equal(sy.view(np.uint32), sym.view(np.uint32)), c
        assert np.array_equal(r4, rec), (c, np.flatnonzero((r4 != rec).any(axis=1))[:5])
        assert np.array_equal(gfl & 0x7F, fl), c
        ev = np.concatenate(col.events[c])
        evd = np.concatenate(col.event_data[c])
        assert len(ev) == len(rows), (c, len(ev), len(rows))
        assert np.array_equal(ev[:, 0], np.array([r[0] for r in rows])) and np.array_equal(ev[:, 1], np.array([r[1] for r in rows])), c
        assert np.array_equal(evd, data), c
        # every sync decoded exactly once by the decode stage
        syncs = set(int(i) for i in np.flatnonzero(fl & 2))
        assert set(col.frames[c]) == syncs, (c, sorted(set(col.frames[c]) ^ syncs)[:6])


@pytest.mark.parametrize("n_call", [32000, 24000])
def test_control_channel_capture_to_tsbk_payloads(built, n_call):
    iq = np.ascontiguousarray(golden("iq_p25p1_cqpsk_cc.npz")["iq"])
    col, n_total = run_chain(iq, n_call)
    check_stream(col, *oracle_stream(iq, n_total, n_call))
    blocks = []
    for g, f in sorted(col.frames[0].items()):
        if f["nid"][0] > 0 and f["nid"][2] == 7:
            for b in range(3):
                by, ok = f["tsbk"][b]
                if ok:
                    blocks.append(by)
                if by[0] & 0x80:
                    break
    assert len(blocks) >= 30
    net = [b for b in blocks if (int(b[0]) & 0x3F) == 0x3B]
    assert len(net) >= 2       # "WACN: 92065; SYS: 0D5"
    for b in net:
        assert ((int(b[3]) << 12) | (int(b[4]) << 4) | (int(b[5]) >> 4), ((int(b[5]) & 0xF) << 8) | int(b[6])) == (0x92065, 0x0D5)


def test_simulcast_capture_grant_update(built):
    iq = np.ascontiguousarray(golden("iq_p25p1_cqpsk_cc_simulcast.npz")["iq"])
    col, n_total = run_chain(iq, 32000)
    check_stream(col, *oracle_stream(iq, n_total, 32000))
    ops = [int(f["tsbk"][b][0][0]) & 0x3F for _, f in sorted(col.frames[0].items()) if f["nid"][0] > 0 and f["nid"][2] == 7
           for b in range(3) if f["tsbk"][b][1]]
    assert len(ops) >= 15 and 0x02 in ops       # "Group Voice Channel Grant Update - Implicit"


def test_voice_capture_link_control_and_pcm(built):
    """"Group Voice Channel User": LDU1 link control through Hamming(10,6,3) + RS(24,12,13) clean, LCF 0x00 (group voice channel user);
    the LDUs' IMBE frames reach the vocoder and the PCM is not silence"""
    iq = np.ascontiguousarray(golden("iq_p25p1_cqpsk_vc.npz")["iq"])
    col, n_total = run_chain(iq, 48000)
    check_stream(col, *oracle_stream(iq, n_total, 48000))
    ldu1 = [f for _, f in sorted(col.frames[0].items()) if f["nid"][0] > 0 and f["nid"][2] == 5]
    ldu2 = [f for _, f in sorted(col.frames[0].items()) if f["nid"][0] > 0 and f["nid"][2] == 10]
    assert len(ldu1) >= 3 and len(ldu2) >= 2 and all(int(f["nid"][1]) == 0x106 for f in ldu1 + ldu2)
    for f in ldu1:
        assert f["rs1s"] == 0                                   # RS(24,12,13) clean
        hexw = f["rs1"].reshape(12, 6)                          # 12 data hex words, 6 bits each; the last one is first on the air
        lcf = int("".join(str(int(b)) for b in hexw[11]) + "".join(str(int(b)) for b in hexw[10][:2]), 2)
        assert lcf in (0x00, 0x42), hex(lcf)
    assert all(f["rs2s"] == 0 for f in ldu2)
    pcm = np.concatenate([v[4] for v in col.voice[0]])
    assert len(col.voice[0]) >= 45 and float(np.abs(pcm).sum()) > 0
    for c in (1, 2):                                            # the three rows carry the same stream
        assert len(col.voice[c]) == len(col.voice[0])
        # (the voice parameter bits; the PCM differs by design: every talk path draws its own unvoiced-band noise)
        assert all(np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) for a, b in zip(col.voice[0], col.voice[c]))
