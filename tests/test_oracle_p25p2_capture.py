"""The Phase 2 burst layer's CPU restatement on a real signal: the reference's own P25 Phase 2 capture (tests/p2capture.py).  Known answers
that do not come from this repository's generators: the S-ISCH cadence, the I-ISCH words (exact (40,9,16) code words whose channel /
location / ultraframe fields count the way TIA-102.BBAC lays a superframe out), the DUID words (exact (8,4) code words), and - the one that
ties bit positions, the scrambler, its offset rule, RS(63,35) with the SACCH erasures and the CRC-12 together - ten scrambled SACCH
bursts (DUID 3, slots 10 and 11 of each superframe) that decode with nothing to correct and a good MAC checksum."""
import ctypes as C

import numpy as np
import pytest

import orc
import p2capture
from test_oracle_p25p2_xcch import crc12_ok, needs_ref, oracle_duid, oracle_xcch, ref_xcch, scramble_bits

# MAC PDUs of the ten SACCH bursts (first 12 of 21 octets + CRC): MAC_IDLE, three distinct broadcasts repeating
SACCH_OCTETS = ["7c8790038b900fff3070ff40", "7c87900381900b0191019101", "7c87900381900b0191019101", "7c8790038b900fff3070ff40",
                "7c87900381900b0191019101", "7c87900381900b0191019101", "7c8790034280fe4e2230184e", "7c87900354004048ffff031c",
                "7c87900381900b0191019101", "7c8790038b900fff3070ff40"]


def isch_word(bits):
    w = 0
    for k in range(40):
        w = (w << 1) | int(bits[320 + k])
    return w


def expected_isch(idx):
    """TIA-102.BBAC superframe as this capture shows it: S-ISCH behind slots 1, 2 (mod 4), I-ISCH behind slots 3 (channel 0) and 0
    (channel 1) carrying the location 0..2 of the four-slot group and the ultraframe count"""
    sf = (idx + 2) % 12
    if sf % 4 in (1, 2):
        return -2
    chan = 0 if sf % 4 == 3 else 1
    loc = ((sf + 1) // 4) % 3
    uf = ((idx + 3) // 12) & 3
    return (chan << 5) | (loc << 3) | uf


def descrambled(bits, sf):
    seq = scramble_bits(p2capture.WACN, p2capture.SYSID, p2capture.NAC, 4320)
    two = np.concatenate([seq, seq])
    return np.stack([bits[i] ^ two[20 + 360 * int(sf[i]):380 + 360 * int(sf[i])] for i in range(len(bits))])


def test_sync_cadence_isch_and_duid_words_of_the_capture():
    bits, llr, sf, hits = p2capture.timeslots()
    assert len(bits) == 66
    # two S-ISCH 180 dibits apart in every 720
    assert hits[:6] == [65, 245, 785, 965, 1505, 1685]
    o = orc.oracle()
    o.orc_isch_lookup.argtypes = [C.c_uint64]
    o.orc_isch_lookup_soft.argtypes = [C.c_uint64, C.c_void_p]
    n_exact = 0
    for i in range(len(bits)):
        w = isch_word(bits[i])
        r40 = np.minimum(np.abs(llr[i, 320:360]), 255).astype(np.uint8)
        want = expected_isch(i)
        assert o.orc_isch_lookup_soft(C.c_uint64(w), r40.ctypes.data) == want, (i, hex(w))
        n_exact += int(o.orc_isch_lookup(C.c_uint64(w)) == want)
        word = 0
        for k in p2capture.DUID_OFFSETS:
            word = (word << 1) | int(bits[i, k])
        r8 = np.minimum(np.abs(llr[i, p2capture.DUID_OFFSETS]), 255).astype(np.uint8)
        assert oracle_duid(word, r8) == (3 if sf[i] >= 10 else 10), (i, hex(word))
    assert n_exact >= 64


def test_scrambled_sacch_bursts_of_the_capture_decode_with_a_good_crc12():
    bits, llr, sf, _ = p2capture.timeslots()
    x = descrambled(bits, sf)
    got = []
    for i in range(len(bits)):
        ec, pl, used = oracle_xcch(1, x[i], llr[i])
        if sf[i] >= 10:
            assert ec == 11 and used == 0 and crc12_ok(pl, 168) == 1, (i, ec, used)        # 11 = the fixed erasures, nothing else
            got.append(bytes(np.packbits(pl)[:12]).hex())
            # and not without the scrambler
            ec2, pl2, _ = oracle_xcch(1, bits[i], llr[i])
            assert ec2 < 0 or crc12_ok(pl2, 168) == 0
        else:
            assert ec < 0 or crc12_ok(pl, 168) == 0, i
    assert got == SACCH_OCTETS


@needs_ref
def test_the_compiled_reference_pieces_agree_on_the_capture():
    bits, llr, sf, _ = p2capture.timeslots()
    x = descrambled(bits, sf)
    for kind in (0, 1):
        for i in range(len(bits)):
            for b in (bits[i], x[i]):
                want = ref_xcch(kind, b, llr[i], i % 4)
                ec, pl, used = oracle_xcch(kind, b, llr[i])
                assert (ec, used) == (want[0], want[2]) and np.array_equal(pl, want[1]), (kind, i)
