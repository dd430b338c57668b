"""M17 on the CPU side: the restatement of the receive-side handlers (oracle/ddn_oracle_m17.c) pinned to the reference's own
m17_algorithms.c / m17_parse.c compiled into oracle/_ref, frames built by the reference's encoder decoded by the restatement, the
loop's M17 matcher on synthetic transmissions, and the reference's known answer for its own capture: DECODE_IQ_M17 (-fz) expects
"SRC: N0CALL" (tests/CMakeLists.txt:8964)."""
import ctypes as C

import numpy as np
import pytest

import m17
import orc
import p25gen
import rx4


def test_bit_shuffles_crc_lich_and_callsign_equal_the_compiled_reference(built):
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    r, o = m17._r(), m17._o()
    rng = np.random.default_rng(5)
    for _ in range(50):
        by = rng.integers(0, 256, 28).astype(np.uint8)
        assert m17.crc16(by) == int(r.m17_crc16(by.ctypes.data, 28))
        bits = rng.integers(0, 2, 368).astype(np.uint8)
        dib = ((bits[0::2] << 1) | bits[1::2]).astype(np.uint8)
        want, got = np.zeros(368, np.uint8), np.zeros(368, np.uint8)
        r.m17_payload_decode_bits(bits.ctypes.data, want.ctypes.data)
        o.orc_m17_payload_bits(dib.ctypes.data, got.ctypes.data)
        assert np.array_equal(want, got)
        a = int(rng.integers(1, 40 ** 9))
        buf = C.create_string_buffer(10)
        assert r.m17_address_decode_csd(a, buf) == 0 and m17.callsign(a) == (0, buf.value.decode())
    assert m17.crc16(np.frombuffer(b"123456789", np.uint8)) == 0x772B          # M17 specification's CRC check value
    assert m17.callsign(m17.encode_callsign("N0CALL")) == (0, "N0CALL")
    assert m17.callsign(0)[0] == -2 and m17.callsign(40 ** 9)[0] == -2 and m17.callsign(0xFFFFFFFFFFFF)[0] == -2
    # LICH words through Golay(24,12): up to three bit errors per word come back, four do not go unnoticed
    for trial in range(40):
        content = rng.integers(0, 2, 48).astype(np.uint8)
        cnt = int(rng.integers(0, 6))
        content[40:43] = [(cnt >> 2) & 1, (cnt >> 1) & 1, cnt & 1]
        enc = np.zeros(96, np.uint8)
        r.m17_lich_encode_bits(content.ctypes.data, enc.ctypes.data)
        for w in range(4):
            flip = rng.choice(24, int(rng.integers(0, 4)), replace=False)
            enc[24 * w + flip] ^= 1
        ref_out = np.zeros(48, np.uint8)
        assert r.m17_lich_decode_bits(enc.ctypes.data, ref_out.ctypes.data) == 0 and np.array_equal(ref_out, content)
        # as a frame: LICH + zero payload, randomised and interleaved by the reference
        comb, rnd = np.zeros(368, np.uint8), np.zeros(368, np.uint8)
        comb[:96] = enc
        r.m17_payload_encode_bits(comb.ctypes.data, rnd.ctypes.data)
        err, lich, c, _ = m17.str_decode(((rnd[0::2] << 1) | rnd[1::2]).astype(np.uint8))
        assert err == 0 and c == cnt and np.array_equal(np.unpackbits(lich), content)


def _ideal(dibits, amp=3.0, noise=0.0, rng=None):
    lv = np.array([1.0, 3.0, -1.0, -3.0], np.float32)[np.asarray(dibits, np.int64)] * (amp / 3.0)
    if noise:
        lv = lv + rng.normal(0, noise, lv.shape).astype(np.float32)
    return lv.astype(np.float32)


def test_frames_of_the_reference_encoder_decode(built):
    """LSF frames (soft costs -> P1 de-puncture -> the libM17-style decoder -> CRC) and stream frames (LICH + P2 + the NXDN-style
    decoder) built by m17_lsf_encode_type1_bits / m17_stream_encode_type1_bits come back, clean and through noise"""
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(17)
    thr = np.array([0.0, 2.0, -2.0, 3.0, -3.0], np.float32)          # centre, umid, lmid, max, min of levels +-1 / +-3
    for trial in range(12):
        dst, src = m17.encode_callsign("ALL"), m17.encode_callsign("N0CALL" if trial == 0 else "W%dXYZ" % trial)
        bits, by = m17.lsf_bits(dst, src, type_word=0x0005 | (trial << 7))
        fr = m17.lsf_frame(bits)
        sym = _ideal(fr[8:], noise=0.25 if trial % 2 else 0.0, rng=rng)
        lsf, ok, cost = m17.lsf_decode(m17.lsf_costs(sym, thr))
        assert ok == 1 and np.array_equal(lsf, by), trial
        assert m17.callsign(int.from_bytes(bytes(lsf[6:12].tolist()), "big"))[1] == ("N0CALL" if trial == 0 else "W%dXYZ" % trial)
        pay = rng.integers(0, 256, 16).astype(np.uint8)
        sf = m17.stream_frame(bits, trial % 6, 0x1230 + trial, pay)
        d = sf[8:].copy()
        hit = rng.choice(184, 3, replace=False)                       # a few symbol errors: Golay and the K = 5 code absorb them
        d[hit[0]] ^= 1
        if trial % 2:
            d[96 + int(hit[1]) % 80] ^= 2
        err, lich, cnt, fp = m17.str_decode(d)
        assert err == 0 and cnt == trial % 6
        assert np.array_equal(np.unpackbits(lich)[:40], bits[40 * cnt:40 * cnt + 40])
        assert (int(fp[0]) << 8 | int(fp[1])) == 0x1230 + trial and np.array_equal(fp[2:], pay), trial
    # degenerate thresholds: the symmetric set is synthesised (dsd_dibit.c:1201-1212)
    c0 = m17.lsf_costs(_ideal(fr[8:]), np.array([0, 0, 0, 3, -3], np.float32))
    assert np.array_equal(m17.lsf_decode(c0)[0], by)


def test_cost_function_edges():
    thr = np.array([0.0, 2.0, -2.0, 3.0, -3.0], np.float32)
    o = m17._o()
    cost = lambda x, b: int(o.orc_m17_soft_cost(C.c_float(x), thr.ctypes.data, b))
    assert cost(3.0, 0) < 100 and cost(-3.0, 0) > 65435 and cost(30.0, 0) == 0 and cost(-30.0, 0) == 65535   # sign bit: + is 0
    assert cost(3.0, 1) > 50000 and cost(0.5, 1) < 15000             # magnitude bit: outer is 1
    assert cost(0.0, 0) in (32767, 32768) and abs(cost(1.75, 1) - 32768) < 1000
    assert [cost(x, 0) for x in np.linspace(-4, 4, 33)] == sorted([cost(x, 0) for x in np.linspace(-4, 4, 33)], reverse=True)


def _run(dibits, noise=0.02, lead=20, seed=1):
    iq = p25gen.modulate_cu8(dibits, len(dibits) * 10 + 1200, lead=lead, seed=seed, noise=noise)
    disc = orc.OracleFrontEnd(profile=2).run_cu8(iq, 8192)
    return rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(disc)


def two_transmissions(gap, seed):
    """-> (dibits, LSF bytes, [(fn, payload)]) of one transmission sent twice with `gap` dibits in between"""
    rng = np.random.default_rng(3)
    tx, by, sent = m17.transmission(rng, m17.encode_callsign("ALL"), m17.encode_callsign("N0CALL"), 14)
    return np.concatenate([tx, np.array(gap, np.int8), tx]), by, sent


def test_synthetic_transmission_through_the_loop(built):
    """preamble -> LSF -> stream frames -> EOT, built by the reference's encoder, through modulator, front end and the loop's M17
    matcher: every frame is found 192 symbols after the one before, the LSF decodes (soft costs -> a17), the LICH chunks reassemble
    it, the payloads come back (a18), EOT clears the sync type so that the second transmission needs its own preamble.
    The reference consumes a preamble 16 symbols at a time (8 matched, 8 skipped) and reads the polarity off the phase of the first
    match, so whether a transmission is taken the right way up depends on where its preamble starts in that rhythm: both cases here."""
    if orc.ref() is None:
        pytest.skip("oracle/_ref not built")
    d, by, sent = two_transmissions([1, 3, 1], 5)
    fr = m17.decode_stream(_run(d, seed=5))
    kinds = [f["kind"] for f in fr]
    lsf = [f for f in fr if f["kind"] == "lsf"]
    assert len(lsf) == 2 and all(f["crc_ok"] and np.array_equal(f["lsf30"], by) for f in lsf)
    assert m17.callsign(int.from_bytes(bytes(lsf[0]["lsf30"][6:12].tolist()), "big"))[1] == "N0CALL"
    st = [f for f in fr if f["kind"] == "str"]
    assert len(st) == 28 and all(f["lich_err"] == 0 for f in st)
    assert [(f["fn"], bytes(f["payload"].tolist())) for f in st] == [(fn, bytes(p.tolist())) for fn, p in sent] * 2
    assert all(b["pos"] - a["pos"] == 192 for a, b in zip(st[:14], st[1:14]))
    fin = [f for f in st if "lich_lsf30" in f]
    assert len(fin) == 4 and all(f["lich_crc_ok"] and np.array_equal(f["lich_lsf30"], by) for f in fin)
    assert kinds.count("eot") == 2 and all(f["pat"] == rx4.M17_PRE_POS for f in fr if f["kind"] == "pre")
    e = kinds.index("eot")
    assert "pre" not in kinds[kinds.index("lsf"):e] and kinds[e + 1] == "pre" and "lsf" in kinds[e:]
    # the same pair without the gap: the second preamble is matched on the other phase -> -M17, nothing of it passes a CRC
    d, by, sent = two_transmissions([], 1)
    fr = m17.decode_stream(_run(d, seed=1))
    kinds = [f["kind"] for f in fr]
    e = kinds.index("eot")
    assert sum(1 for f in fr[:e] if f["kind"] == "str" and f["lich_err"] == 0) == 14 and fr[e + 1]["pat"] == rx4.M17_PRE_NEG
    assert not any(f.get("crc_ok") or f.get("lich_crc_ok") for f in fr[e:])


def test_m17_capture_known_answer_src_n0call(built):
    """the reference's own capture: the LSF (from its frame, or reassembled from the LICH chunks) passes its CRC16 and names N0CALL"""
    disc = rx4.capture_disc("iq_m17.npz", 2)
    out = rx4.OracleFsk4Rx(rx4.profile(rx4.PROTO_M17)).run(disc)
    fr = m17.decode_stream(out)
    kinds = [f["kind"] for f in fr]
    srcs = []
    for f in fr:
        for key, okk in (("lsf30", "crc_ok"), ("lich_lsf30", "lich_crc_ok")):
            if key in f and f[okk]:
                srcs.append(m17.callsign(int.from_bytes(bytes(f[key][6:12].tolist()), "big"))[1])
    assert srcs and set(srcs) == {"N0CALL"}, (kinds, srcs)
    # The capture starts in mid-transmission (no preamble / LSF of its own: the loop gets in through chance matches in what comes
    # first); from there on every stream frame is found 192 symbols after the one before, frame numbers count up by one to the frame
    # with the end-of-stream flag, and the EOT marker follows
    st = [f for f in fr if f["kind"] == "str" and f["lich_err"] == 0]
    run = [st[-1]]
    for f in reversed(st[:-1]):
        if run[0]["pos"] - f["pos"] == 192 and (run[0]["fn"] & 0x7FFF) - (f["fn"] & 0x7FFF) == 1:
            run.insert(0, f)
        else:
            break
    assert len(run) >= 35 and run[-1]["fn"] & 0x8000 and not any(f["fn"] & 0x8000 for f in run[:-1])
    assert [f["cnt"] for f in run] == [(run[0]["cnt"] + k) % 6 for k in range(len(run))]
    assert sum(1 for f in run if f.get("lich_crc_ok")) >= 5
    after = [f for f in fr if f["pos"] > run[-1]["pos"]]
    assert after and after[0]["kind"] == "eot" and after[0]["pos"] - run[-1]["pos"] == 192
