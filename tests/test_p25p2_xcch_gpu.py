"""GPU: the P25 Phase 2 FACCH / SACCH burst stage (ddn_p25p2_xcch_batch / _host: burst gather, ranked soft erasures, RS(63,35) with
the fixed erasures and the retries) against the CPU restatement, which tests/test_oracle_p25p2_xcch.py pins to the reference's own
ez.cpp + p25p2_soft.c compiled in place: ec, the used-dynamic flag and the payload bit for bit, every outcome class covered."""
import os

import numpy as np
import pytest

import ddn
import rs28
from test_oracle_p25p2_xcch import N_PL, oracle_xcch

pytestmark = pytest.mark.gpu
FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))


def test_xcch_batch_equals_oracle(built):
    rng = np.random.default_rng(47 + FZ)
    for kind in (0, 1):
        n = 1500
        bits, llr, want = np.zeros((n, 360), np.uint8), np.zeros((n, 360), np.int16), []
        for i in range(n):
            n_err = int(rng.integers(0, 15))
            b, l, _ = rs28.make_xcch_burst(rng, kind, n_err, int(rng.integers(0, n_err + 1)), int(rng.integers(0, 8)))
            if i % 97 == 0:
                l[:] = rng.integers(-300, 300, 360)         # all-weak metrics: the minimum / maximum list lengths
            bits[i], llr[i] = b, l
            want.append(oracle_xcch(kind, b, l))
        pl, ec, used = np.zeros((n, N_PL[kind]), np.uint8), np.zeros(n, np.int32), np.zeros(n, np.uint8)
        assert ddn.lib().ddn_p25p2_xcch_host(kind, bits.ctypes.data, llr.ctypes.data, n, 64, pl.ctypes.data, ec.ctypes.data, used.ctypes.data) == 0
        classes = set()
        for i, (wec, wpl, wused) in enumerate(want):
            assert ec[i] == wec and used[i] == wused and np.array_equal(pl[i], wpl), (kind, i, ec[i], wec, used[i], wused)
            classes.add((wec >= 0, wused))
        assert classes == {(True, 0), (True, 1), (False, 0)}, classes
        # another threshold moves the list lengths, not the rule
        thr = 20
        want2 = [oracle_xcch(kind, bits[i], llr[i], thr) for i in range(0, n, 7)]
        assert ddn.lib().ddn_p25p2_xcch_host(kind, bits.ctypes.data, llr.ctypes.data, n, thr, pl.ctypes.data, ec.ctypes.data, used.ctypes.data) == 0
        for j, i in enumerate(range(0, n, 7)):
            assert ec[i] == want2[j][0] and used[i] == want2[j][2] and np.array_equal(pl[i], want2[j][1]), (kind, i)


def test_xcch_large_batches_take_the_in_thread_retries_and_agree(built):
    """the retries run side by side for small batches and inside the thread for large ones of short lists (rs28_retries(),
    ddn_rs.hip): the same bursts through both routes give the same answers, which are the oracle's"""
    rng = np.random.default_rng(59 + FZ)
    for kind, tiles in ((0, 128), (1, 80)):                     # 65 536 x 10 and 40 960 x 16 (section, attempt) pairs: beyond 600 k
        u = 512
        bits, llr = np.zeros((u, 360), np.uint8), np.zeros((u, 360), np.int16)
        for i in range(u):
            n_err = int(rng.integers(0, 14))
            bits[i], llr[i], _ = rs28.make_xcch_burst(rng, kind, n_err, int(rng.integers(0, n_err + 1)), int(rng.integers(0, 6)))
        outs = []
        for reps in (1, tiles):
            n = u * reps
            b, l = np.tile(bits, (reps, 1)), np.tile(llr, (reps, 1))
            pl, ec, used = np.zeros((n, N_PL[kind]), np.uint8), np.zeros(n, np.int32), np.zeros(n, np.uint8)
            assert ddn.lib().ddn_p25p2_xcch_host(kind, b.ctypes.data, l.ctypes.data, n, 64, pl.ctypes.data, ec.ctypes.data, used.ctypes.data) == 0
            outs.append((pl.reshape(reps, u, -1), ec.reshape(reps, u), used.reshape(reps, u)))
        small, large = outs
        for t in range(tiles):
            assert np.array_equal(large[0][t], small[0][0]) and np.array_equal(large[1][t], small[1][0]) and np.array_equal(large[2][t], small[2][0]), (kind, t)
        for i in range(0, u, 5):
            wec, wpl, wused = oracle_xcch(kind, bits[i], llr[i])
            assert small[1][0][i] == wec and small[2][0][i] == wused and np.array_equal(small[0][0][i], wpl), (kind, i)
        assert (small[2][0] == 1).sum() >= 10 and (small[1][0] < 0).sum() >= 10


def test_burst_fields_duid_and_isch_equal_oracle(built):
    import ctypes as C
    import orc
    from test_oracle_p25p2_xcch import oracle_duid
    rng = np.random.default_rng(53 + FZ)
    n = 4000
    canon = [0x00, 0x17, 0x2E, 0x39, 0x4B, 0x5C, 0x65, 0x72, 0x8D, 0x9A, 0xA3, 0xB4, 0xC6, 0xD1, 0xE8, 0xFF]
    o = orc.oracle()
    o.orc_isch_lookup_soft.argtypes = [C.c_uint64, C.c_void_p]
    bits = rng.integers(0, 2, (n, 360)).astype(np.uint8)
    llr = (rng.integers(0, 300, (n, 360)) * rng.choice([-1, 1], (n, 360))).astype(np.int16)
    off = [0, 1, 74, 75, 244, 245, 318, 319]
    for i in range(n):                       # DUID words near the code: canonical, one or two flips, the 0x80 guard; weak bits here and there
        w = canon[int(rng.integers(0, 16))] if i % 5 else 0x80
        for _ in range(int(rng.integers(0, 3))):
            w ^= 1 << int(rng.integers(0, 8))
        bits[i, off] = [(w >> (7 - k)) & 1 for k in range(8)]
        for k in rng.choice(8, int(rng.integers(0, 4)), replace=False):
            llr[i, off[k]] = int(rng.integers(0, 70))
        if i % 3 == 0:                       # an I-ISCH word a few bits from the one another burst carries, some of them weak
            src = bits[(i * 7) % n, 320:360].copy()
            for k in rng.choice(40, int(rng.integers(0, 9)), replace=False):
                src[k] ^= 1
                llr[i, 320 + k] = int(rng.integers(0, 40))
            bits[i, 320:360] = src
    duid, isch = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert ddn.lib().ddn_p25p2_burst_fields_host(bits.ctypes.data, llr.ctypes.data, n, 64, duid.ctypes.data, isch.ctypes.data) == 0
    seen = set()
    for i in range(n):
        rel = np.minimum(np.abs(llr[i].astype(np.int32)), 255).astype(np.uint8)
        w = 0
        for k in range(8):
            w = (w << 1) | int(bits[i, off[k]])
        want = oracle_duid(w, rel[off])
        assert duid[i] == want, (i, hex(w), duid[i], want)
        seen.add(want)
        word = 0
        for k in range(40):
            word = (word << 1) | int(bits[i, 320 + k])
        r40 = np.ascontiguousarray(rel[320:360])
        assert isch[i] == o.orc_isch_lookup_soft(C.c_uint64(word), r40.ctypes.data), i
    assert seen >= set(range(16)) | {-1}, seen


def test_scramble_sequence_and_descramble(built):
    """ddn_p25p2_scramble_bits_batch / _descramble_batch / the reference-named p25p2_generate_scramble_bits: the LFSR sequence bit for bit
    (the Python restatement is pinned to the compiled p25p2_scramble.c in tests/test_oracle_p25p2_xcch.py), de-scrambled bits and
    sign-flipped metrics against process_Frame_Scramble()'s two loops (p25p2_frame.c:381-392), offsets that wrap included"""
    import ctypes as C
    import torch
    from test_oracle_p25p2_xcch import scramble_bits
    rng = np.random.default_rng(61 + FZ)
    l = ddn.lib()
    ns = 5
    ids = [(int(rng.integers(1, 1 << 20)), int(rng.integers(1, 1 << 12)), int(rng.integers(1, 1 << 12))) for _ in range(ns)]
    seeds = torch.tensor([w * 16777216 + s * 4096 + n for w, s, n in ids], dtype=torch.int64, device="cuda")
    seq = torch.zeros((ns, 4320), dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p2_scramble_bits_batch(seeds.data_ptr(), ns, 4320, seq.data_ptr(), None) == 0
    torch.cuda.synchronize()
    want = np.stack([scramble_bits(w, s, n, 4320) for w, s, n in ids])
    assert np.array_equal(seq.cpu().numpy(), want)
    one = np.zeros(1000, np.uint8)
    l.p25p2_generate_scramble_bits(ids[0][0], ids[0][1], ids[0][2], one.ctypes.data, 1000)
    assert np.array_equal(one, want[0, :1000])
    n, nb, nl = 40, 4300, 1400
    bits = rng.integers(0, 2, (n, nb)).astype(np.uint8)
    llr = rng.integers(-300, 300, (n, nl)).astype(np.int16)
    off = rng.integers(0, 12, n).astype(np.int32)
    which = rng.integers(0, ns, n).astype(np.int32)
    tb, tl = torch.from_numpy(bits).cuda(), torch.from_numpy(llr).cuda()
    to, tw = torch.from_numpy(off).cuda(), torch.from_numpy(which).cuda()
    xb, xl = torch.zeros_like(tb), torch.zeros_like(tl)
    assert l.ddn_p25p2_descramble_batch(tb.data_ptr(), tl.data_ptr(), seq.data_ptr(), to.data_ptr(), tw.data_ptr(), n, nb, nl, xb.data_ptr(),
                                        xl.data_ptr(), None) == 0
    torch.cuda.synchronize()
    for i in range(n):
        lb = np.concatenate([want[which[i]], want[which[i]]])          # "doubling up for offset roll-over"
        s = lb[20 + 360 * int(off[i]):20 + 360 * int(off[i]) + nb]
        assert np.array_equal(xb[i].cpu().numpy(), bits[i] ^ s), i
        assert np.array_equal(xl[i].cpu().numpy(), np.where(s[:nl] == 1, -llr[i], llr[i]).astype(np.int16)), i


def test_ess_batch_equals_oracle(built):
    from test_oracle_p25p2_xcch import oracle_ess
    rng = np.random.default_rng(71 + FZ)
    n = 1200
    pl, pll = np.zeros((n, 96), np.uint8), np.zeros((n, 96), np.int16)
    pa, pal = np.zeros((n, 168), np.uint8), np.zeros((n, 168), np.int16)
    want = []
    for i in range(n):
        n_err = int(rng.integers(0, 26))
        a, b, c, d, _ = rs28.make_ess_case(rng, n_err, int(rng.integers(0, n_err + 1)), int(rng.integers(0, 8)))
        if i % 89 == 0:
            b[:], d[:] = rng.integers(-300, 300, 96), rng.integers(-300, 300, 168)
        pl[i], pll[i], pa[i], pal[i] = a, b, c, d
        want.append(oracle_ess(a, b, c, d))
    out, ec, used = np.zeros((n, 96), np.uint8), np.zeros(n, np.int32), np.zeros(n, np.uint8)
    assert ddn.lib().ddn_p25p2_ess_host(pl.ctypes.data, pll.ctypes.data, pa.ctypes.data, pal.ctypes.data, n, 64, out.ctypes.data, ec.ctypes.data,
                                        used.ctypes.data) == 0
    classes = set()
    for i, (acc, wec, wout) in enumerate(want):
        assert (ec[i] >= 0) == bool(acc) and ec[i] == wec and np.array_equal(out[i], wout), (i, ec[i], acc, wec)
        classes.add((acc, int(used[i])))
    assert classes == {(1, 0), (1, 1), (0, 0)}, classes


def test_voice_frames_unpack(built):
    import torch
    import rx4
    rng = np.random.default_rng(73 + FZ)
    m = np.asarray(rx4.ambe2450_map())
    n = 50
    bits = rng.integers(0, 2, (n, 360)).astype(np.uint8)
    llr = rng.integers(-400, 400, (n, 360)).astype(np.int16)
    tb, tl = torch.from_numpy(bits).cuda(), torch.from_numpy(llr).cuda()
    for fc in (4, 2):
        fr = torch.full((n, fc, 4, 24), 9, dtype=torch.uint8, device="cuda")
        rl = torch.full((n, fc, 4, 24), 9, dtype=torch.uint8, device="cuda")
        assert ddn.lib().ddn_p25p2_voice_frames_batch(tb.data_ptr(), tl.data_ptr(), n, fc, fr.data_ptr(), rl.data_ptr(), None) == 0
        torch.cuda.synchronize()
        wf, wr = np.zeros((n, fc, 4, 24), np.uint8), np.zeros((n, fc, 4, 24), np.uint8)
        for f, off in enumerate((2, 76, 172, 246)[:fc]):
            for x in range(72):
                row, col = (m[x // 2][0], m[x // 2][1]) if x % 2 == 0 else (m[x // 2][2], m[x // 2][3])
                wf[:, f, row, col] = bits[:, off + x]
                wr[:, f, row, col] = np.minimum(np.abs(llr[:, off + x].astype(np.int32)), 255)
        assert np.array_equal(fr.cpu().numpy(), wf) and np.array_equal(rl.cpu().numpy(), wr), fc


def test_mac_crc_batch(built):
    from test_oracle_p25p2_xcch import crc12_ok, crc16_ok, with_crc12
    rng = np.random.default_rng(83 + FZ)
    for kind, n_pl in ((0, 156), (1, 180)):
        n = 300
        rows = np.zeros((n, n_pl), np.uint8)
        for i in range(n):
            rows[i] = with_crc12(rng, n_pl) if i % 3 else rng.integers(0, 2, n_pl)
            if i % 5 == 2:
                rows[i, int(rng.integers(0, n_pl))] ^= 1
            if kind == 1 and i % 4 == 1:          # a good CRC16 over the first 164 bits (LCCH)
                c = 0
                for k in range(164):
                    c = (((c << 1) ^ 0x1021) if (((c >> 15) & 1) ^ int(rows[i, k])) else (c << 1)) & 0xFFFF
                c ^= 0xFFFF
                rows[i, 164:180] = [(c >> (15 - k)) & 1 for k in range(16)]
        a, b = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        assert ddn.lib().ddn_p25p2_mac_crc_host(kind, rows.ctypes.data, n, a.ctypes.data, b.ctypes.data) == 0
        w12 = [crc12_ok(rows[i], n_pl - 12) for i in range(n)]
        assert a.tolist() == w12 and 0 < sum(w12) < n
        if kind == 1:
            w16 = [crc16_ok(rows[i]) for i in range(n)]
            assert b.tolist() == w16 and sum(w16) >= 50
        else:
            assert not b.any()


def test_the_reference_capture_through_the_burst_layer(built):
    """the reference's own Phase 2 capture (tests/p2capture.py) through the C-ABI: DUID and I-ISCH of all 66 timeslots, the scrambler
    sequence of the system, de-scrambling at the superframe slot the I-ISCH names, SACCH RS(63,35) and the MAC CRC-12 - equal to the
    CPU restatement on every timeslot, and the ten SACCH bursts give the known MAC PDUs (tests/test_oracle_p25p2_capture.py)"""
    import torch
    import p2capture
    from test_oracle_p25p2_capture import SACCH_OCTETS, expected_isch
    from test_oracle_p25p2_xcch import crc12_ok
    l = ddn.lib()
    bits, llr, sf, _ = p2capture.timeslots()
    n = len(bits)
    duid, isch = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert l.ddn_p25p2_burst_fields_host(bits.ctypes.data, llr.ctypes.data, n, 64, duid.ctypes.data, isch.ctypes.data) == 0
    assert [int(v) for v in isch] == [expected_isch(i) for i in range(n)]
    assert [int(v) for v in duid] == [3 if s >= 10 else 10 for s in sf]
    seed = torch.tensor([p2capture.WACN * 16777216 + p2capture.SYSID * 4096 + p2capture.NAC], dtype=torch.int64, device="cuda")
    seq = torch.zeros((1, 4320), dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p2_scramble_bits_batch(seed.data_ptr(), 1, 4320, seq.data_ptr(), None) == 0
    tb, tl = torch.from_numpy(bits).cuda(), torch.from_numpy(llr).cuda()
    to, tw = torch.from_numpy(sf).cuda(), torch.zeros(n, dtype=torch.int32, device="cuda")
    xb, xl = torch.zeros_like(tb), torch.zeros_like(tl)
    assert l.ddn_p25p2_descramble_batch(tb.data_ptr(), tl.data_ptr(), seq.data_ptr(), to.data_ptr(), tw.data_ptr(), n, 360, 360, xb.data_ptr(),
                                        xl.data_ptr(), None) == 0
    pl = torch.zeros((n, 180), dtype=torch.uint8, device="cuda")
    ec = torch.zeros(n, dtype=torch.int32, device="cuda")
    used = torch.zeros(n, dtype=torch.uint8, device="cuda")
    c12 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert l.ddn_p25p2_xcch_batch(1, xb.data_ptr(), xl.data_ptr(), n, 64, pl.data_ptr(), ec.data_ptr(), used.data_ptr(), None) == 0
    assert l.ddn_p25p2_mac_crc_batch(1, pl.data_ptr(), n, c12.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    xbn, xln, pln, ecn, usedn, c12n = xb.cpu().numpy(), xl.cpu().numpy(), pl.cpu().numpy(), ec.cpu().numpy(), used.cpu().numpy(), c12.cpu().numpy()
    got = []
    for i in range(n):
        wec, wpl, wused = oracle_xcch(1, xbn[i], xln[i])
        assert ecn[i] == wec and usedn[i] == wused and np.array_equal(pln[i], wpl), (i, ecn[i], wec)
        assert c12n[i] == crc12_ok(wpl, 168), i
        if sf[i] >= 10:
            assert ecn[i] == 11 and c12n[i] == 1, (i, ecn[i], c12n[i])
            got.append(bytes(np.packbits(pln[i])[:12]).hex())
    assert got == SACCH_OCTETS
