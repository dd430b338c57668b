"""CPU: the test-side pieces of the DMR data-burst stage - the synthetic burst generator (tests/dmrgen.py) against the oracle's block
decoders, the stage's restatement (tests/dmr_data.py) on clean bursts of every data type, and the helpers it leans on against the
reference's own functions compiled in place (oracle/_ref: dmr_r34_candidate_metric, ComputeCrcCCITT, ComputeCrc5Bit,
ComputeAndCorrectFullLinkControlCrc)."""
import ctypes as C
import os

import numpy as np
import pytest

import dmr_data
import dmrgen
import fec3
import orc
import p25gen
import rx4

FZ = 7919 * int(os.environ.get("DDN_FUZZ_BASE", "0"))
needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="compiled reference (oracle/_ref) not present")


def test_generator_code_words_decode_clean():
    rng = np.random.default_rng(3 + FZ)
    for _ in range(6):
        bits = rng.integers(0, 2, 96).astype(np.uint8)
        r3 = tuple(int(x) for x in rng.integers(0, 2, 3))
        out, r, errs = fec3.oracle_bptc(dmrgen.bptc_196x96(bits, r3)[None], 1)
        assert errs[0] == 0 and np.array_equal(out[0], bits) and tuple(r[0]) == r3
        x = dmrgen.bptc_196x96(bits, r3)
        x[int(rng.integers(0, 196))] ^= 1                       # a single error is corrected
        out, r, errs = fec3.oracle_bptc(x[None], 1)
        assert errs[0] == 0 and np.array_equal(out[0], bits)
    o = orc.oracle()
    o.orc_rs_12_9.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    for _ in range(20):
        d = [int(x) for x in rng.integers(0, 256, 9)]
        cw = np.array(d + dmrgen.rs_12_9_parity(d), np.uint8)
        syn, found = np.zeros(3, np.uint8), C.c_uint8(0)
        assert o.orc_rs_12_9(cw.ctypes.data, syn.ctypes.data, C.byref(found)) == 0 and not syn.any()
        bad = cw.copy()
        bad[int(rng.integers(0, 12))] ^= int(rng.integers(1, 256))
        assert o.orc_rs_12_9(bad.ctypes.data, syn.ctypes.data, C.byref(found)) == 1 and np.array_equal(bad, cw)


def _clean(dtype, rng, **kw):
    """a clean burst of the type through the restatement -> (result, what was sent)"""
    if dtype == 8:
        sent = dmrgen.r34_bytes(rng, **kw)
        info = dmrgen.r34_info(sent)
    elif dtype == 10:
        info = rng.integers(0, 2, 196).astype(np.uint8)
        info[96:100] = 0
        if kw.get("confirmed"):
            c = dmrgen.crc9_confirmed_rate1(info)
            info[7:16] = [(c >> (8 - i)) & 1 for i in range(9)]
        sent = info.copy()
    else:
        sent = dmrgen.payload_bits(dtype, rng, **kw)
        info = dmrgen.bptc_196x96(sent)
    dib = dmrgen.burst(int(rng.integers(0, 2)), 5, dtype, info)
    return dmr_data.data_burst(dib.astype(np.uint8), np.full(144, 200, np.uint8)), sent


def test_restatement_on_clean_bursts_of_every_type():
    rng = np.random.default_rng(11 + FZ)
    for dtype in (0, 1, 2, 3, 4, 6, 11):
        r, sent = _clean(dtype, rng)
        assert r["type"] == dtype and r["errs"] == 0 and np.array_equal(r["bits96"], sent) and r["crc"] & 1, dtype
        if dtype not in (1, 2):                  # (what RS(12,9) makes of two wrong bytes is the decoder's business: pinned below)
            r, _ = _clean(dtype, rng, good_crc=False)
            assert r["type"] == dtype and not (r["crc"] & 1), dtype
    r, sent = _clean(7, rng)
    assert r["crc"] == 1
    r, sent = _clean(7, rng, confirmed=True, dbsn=9)
    assert r["crc"] == 3 and np.array_equal(r["bits96"], sent)
    r, sent = _clean(7, rng, confirmed=True, dbsn=9, good_crc=False)
    assert r["crc"] == 1
    r, sent = _clean(8, rng)
    assert r["type"] == 8 and np.array_equal(r["unconfirmed"], sent) and r["pool"][0][1] == 0
    r, sent = _clean(8, rng, confirmed=True, dbsn=33)
    assert np.array_equal(r["confirmed"], sent) and r["confirmed_crc"] == 1 and r["pool"][0][3] == 33
    r, sent = _clean(8, rng, confirmed=True, dbsn=33, good_crc=False)
    # (a wrong CRC9 sends the pick to whichever of the 32 list candidates happens to pass, if one does - the reference's rule)
    assert np.array_equal(r["pool"][0][0], sent) and r["pool"][0][1:3] == (0, 0)
    assert (np.array_equal(r["confirmed"], sent) and r["confirmed_crc"] == 0) or r["confirmed_crc"] == 1
    r, sent = _clean(10, rng, confirmed=True)
    assert r["type"] == 10 and r["crc"] == 3 and np.array_equal(r["info"], sent)
    r, _ = _clean(9, rng)
    assert r["type"] == 9 and r["crc"] == 0
    # a full link control with one wrong byte is repaired by RS(12,9), the repaired bytes replace the received ones
    sent = dmrgen.payload_bits(1, rng)
    hurt = sent.copy()
    hurt[8:16] ^= np.unpackbits(np.array([0x5A], np.uint8))
    dib = dmrgen.burst(0, 5, 1, dmrgen.bptc_196x96(hurt))
    r = dmr_data.data_burst(dib.astype(np.uint8), np.full(144, 200, np.uint8))
    assert r["crc"] == 5 and np.array_equal(np.unpackbits(r["bytes12"]), sent) and np.array_equal(r["bits96"], hurt)


@needs_ref
def test_helpers_equal_the_reference():
    r = orc.ref()
    rng = np.random.default_rng(19 + FZ)
    r.dmr_r34_candidate_metric.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for _ in range(40):
        td, rel = rng.integers(0, 4, 98).astype(np.uint8), rng.integers(0, 256, 98).astype(np.uint8)
        b18 = rng.integers(0, 256, 18).astype(np.uint8)
        m = C.c_int(0)
        assert r.dmr_r34_candidate_metric(td.ctypes.data, rel.ctypes.data, b18.ctypes.data, C.byref(m)) == 0
        assert dmr_data.candidate_metric(td, rel, b18) == m.value
    r.ComputeCrcCCITT.restype = C.c_uint16
    r.ComputeCrcCCITT.argtypes = [C.c_void_p]
    r.ComputeCrc5Bit.restype = C.c_uint8
    r.ComputeCrc5Bit.argtypes = [C.c_void_p]
    for _ in range(20):
        bits = rng.integers(0, 2, 96).astype(np.uint8)
        assert (rx4.crc_ccitt_bits(bits[:80]) ^ 0xFFFF) == r.ComputeCrcCCITT(bits.ctypes.data)
        assert int(np.packbits(bits[:72]).astype(np.int64).sum()) % 31 == r.ComputeCrc5Bit(bits.ctypes.data)
    r.ComputeAndCorrectFullLinkControlCrc.restype = C.c_uint32
    r.ComputeAndCorrectFullLinkControlCrc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    for k in range(30):
        dtype = 1 + (k & 1)
        sent = np.packbits(dmrgen.payload_bits(dtype, rng, good_crc=(k % 5 != 4)))
        hurt = sent.copy()
        if k % 3 == 1:
            hurt[int(rng.integers(0, 12))] ^= int(rng.integers(1, 256))
        dib = dmrgen.burst(0, 1, dtype, dmrgen.bptc_196x96(np.unpackbits(hurt)))
        mine = dmr_data.data_burst(dib.astype(np.uint8), np.full(144, 200, np.uint8))
        by = hurt.copy()
        comp = C.c_uint32(0)
        ok = r.ComputeAndCorrectFullLinkControlCrc(by.ctypes.data, C.byref(comp), dmrgen.CRC_MASK[dtype])
        assert (mine["crc"] & 1) == ok and np.array_equal(mine["bytes12"], by), k
