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


def test_synthetic_voice_channels_to_pcm(built):
    """two TDMA channels whose logical channel 0 carries a voice call (4V 4V 4V 4V 2V with valid AMBE 3600x2450 frames planted through
    the burst interleave), modulated to cu8 I/Q, through the chain object in three calls + flush: groups and timeslot decisions equal the
    CPU pipeline, and the PCM of every talk path equals the CPU vocoder restatement fed the frames in air order - the talk path's
    parameters carried across the calls - and is not silence"""
    import mbe
    rng = np.random.default_rng(5)
    sites = [(0x12345, 0x2A1, 0x3C1), (0xBEE00, 0x164, 0x161)]
    n_groups, n_call = 22, 48000
    streams, planted = [], []
    for c, (w, s, nac) in enumerate(sites):
        frames = []

        def voice():
            b = mbe.random_ambe_bits(rng, ())
            frames.append(b)
            return mbe.ambe_encode(b)
        gb, gl = p2seq.make_stream(rng, n_groups, w, s, nac, start_sf=4 * c, plan=None, voice=voice)
        dib = []
        for g in range(n_groups):
            dib.append(p2seq.SYNC20)
            dib.append((gb[g, 0::2] << 1) | gb[g, 1::2])
        streams.append(orc.modulate_dqpsk_cu8(np.concatenate(dib), 8, seed=c))
        planted.append(np.stack(frames))
    n = min(len(s) for s in streams)
    n_total = (n // n_call) * n_call
    iq = np.stack([s[:n_total] for s in streams])
    seeds = [w * 16777216 + s * 4096 + nac for (w, s, nac) in sites]
    ch = ddn.P25P2ChainC(seeds, n_call, vocoder=1)
    per_path = [[] for _ in range(4)]
    total_groups = 0
    for k in range(n_total // n_call + 1):
        if k < n_total // n_call:
            d = _upload(np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call]))
            ch.run(d)
            ddn.lib().ddn_device_free(d)
        else:
            ch.flush()
        r = ch.results()
        G, cap = r.max_groups, r.voice_frames
        ng = ch.fetch(r.d_n_groups, np.int32, (2,))
        info = ch.fetch(r.d_info, np.int32, (2, G, 4, 8))
        afr = ch.fetch(r.d_ambe_fr, np.uint8, (2 * G * 4 * 4, 4, 24))
        vsrc = ch.fetch(r.d_voice_src, np.int32, (4, cap))
        vcnt = ch.fetch(r.d_voice_count, np.int32, (4,))
        vbits = ch.fetch(r.d_voice_bits, np.uint8, (4, cap, 49))
        vres = ch.fetch(r.d_voice_result, np.int32, (4, cap, 5))
        pcm = ch.fetch(r.d_pcm, np.float32, (4, cap, 160))
        total_groups += int(ng.sum())
        for tp in range(4):
            c, slot = tp >> 1, tp & 1
            # air order: the 4V / 2V timeslots the sequencing filed under this slot
            want_src = []
            for g in range(int(ng[c])):
                for ts in range(4):
                    i8 = info[c, g, ts]
                    if i8[3] == slot and i8[4] in (p2seq.A_4V, p2seq.A_2V):
                        want_src += [((c * G + g) * 4 + ts) * 4 + f for f in range(4 if i8[4] == p2seq.A_4V else 2)]
            assert vcnt[tp] == len(want_src) and vsrc[tp, :len(want_src)].tolist() == want_src, (k, tp)
            if not want_src:
                continue
            # frame FEC + synthesis against the CPU restatement on the same frames, state carried
            fr = afr[want_src]
            bits, res, _ = mbe.oracle_frame_decode(ddn.MBE_AMBE, fr, soft=True)
            assert np.array_equal(vbits[tp, :len(want_src)], bits), (k, tp)
            per_path[tp].append((bits, res, pcm[tp, :len(want_src)].copy()))
    ch.close()
    assert total_groups >= 2 * (n_total // (8 * 720)) - 2
    n_voice = 0
    for tp in range(4):
        if not per_path[tp]:
            continue
        bits = np.concatenate([p[0] for p in per_path[tp]])
        got_pcm = np.concatenate([p[2] for p in per_path[tp]])
        v = mbe.OracleVocoder(ddn.MBE_AMBE, 1)
        want_pcm = mbe.oracle_process_stream(v, ddn.MBE_AMBE, bits, tp, np.concatenate([p[1] for p in per_path[tp]]))
        assert np.array_equal(got_pcm.view(np.uint32), want_pcm.view(np.uint32)), tp
        assert float(np.abs(got_pcm).sum()) > 0
        n_voice += len(bits)
    assert n_voice >= 100
    # the frames that went in come out: logical channel 0 of both channels carried the planted parameter bits
    for c in (0, 1):
        got = np.concatenate([p[0] for p in per_path[2 * c]])
        sent = planted[c]
        # (the first group or two fall before the loop's first sync; the decoded run is a contiguous part of what was sent)
        k0 = next(k for k in range(len(sent) - 8) if np.array_equal(sent[k:k + 8], got[:8]))
        assert np.array_equal(got, sent[k0:k0 + len(got)]), c
