"""CPU: DMR / NXDN block codes - the restatement (oracle/ddn_oracle_fec3.c) against the reference's own objects compiled in
place (oracle/_ref: src/fec/fec.c, bptc.c, rs-12-9.c), exhaustively where the code is small, and on the reference-held KAT
of tests/fec/test_fec_bptc_rs.c:19-63."""
import ctypes as C

import numpy as np
import pytest

import fec3
import orc

needs_ref = pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("code", list(fec3.CODES))
def test_block_codes_oracle_vs_reference(built, code):
    rng = np.random.default_rng(code)
    words = fec3.words_for(code, rng)
    if code in (5, 6):                                   # around two real code words, not only around zero
        _, _, ok = fec3.ref_decode(code, words[:1])
        cwords = []
        for seed in range(2):
            r = rng.integers(0, 2, size=fec3.CODES[code][0], dtype=np.uint8)
            fixed, _, okk = fec3.ref_decode(code, r[None])
            if okk[0]:
                cwords.append(fixed[0])
        words = np.concatenate([words] + [words[:3000] ^ c[None, :] for c in cwords])
    a = fec3.oracle_decode(code, words)
    b = fec3.ref_decode(code, words)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@needs_ref
@pytest.mark.parametrize("code", [1, 2, 3, 4])
def test_hamming_multi_codeword_quirks(built, code):
    """nbCodewords > 1: the reference applies the correction inside the FIRST code word and (except (12,8)) stops at an
    uncorrectable word - identical here"""
    rng = np.random.default_rng(40 + code)
    n = fec3.CODES[code][0]
    words = rng.integers(0, 2, size=(600, 3 * n), dtype=np.uint8)
    words[::2] = 0
    for i in range(0, 600, 2):
        words[i, rng.integers(0, 3 * n)] = 1             # a single error somewhere in a three-word item
    a = fec3.oracle_decode(code, words, nb=3)
    b = fec3.ref_decode(code, words, nb=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@needs_ref
def test_bptc_196x96_kat_and_reference(built):
    r = orc.ref()
    r.InitAllFecFunction()
    r.BPTC_196x96_Extract_Data.restype = C.c_uint32
    rng = np.random.default_rng(9)
    x = fec3.bptc_inputs(rng, 400)
    out, r3, errs = fec3.oracle_bptc(x, 0)
    ub = fec3.oracle_bptc.undefined.copy()
    # the reference-held answers (test_fec_bptc_rs.c:31-47): payload, R bits, no irrecoverable rows
    assert errs[0] == 0 and list(r3[0]) == [1, 0, 1] and list(out[0]) == [((i * 17) + (i // 5)) & 1 for i in range(96)]
    one = x[0].copy()
    one[1 + 4 * 15 + 5] ^= 1                              # :49-56
    o1, _, e1 = fec3.oracle_bptc(one[None], 0)
    assert e1[0] == 0 and np.array_equal(o1[0], out[0])
    assert ub.sum() < x.shape[0] // 4
    for i in range(x.shape[0]):
        if ub[i]:
            continue   # the reference's col_corrected[] is read uninitialised when the first column fails (bptc.c:95-111)
        w = x[i].copy()
        ro, rr = np.zeros(96, np.uint8), np.zeros(3, np.uint8)
        re = r.BPTC_196x96_Extract_Data(C.c_void_p(w.ctypes.data), C.c_void_p(ro.ctypes.data), C.c_void_p(rr.ctypes.data))
        assert re == errs[i] and np.array_equal(ro, out[i]) and np.array_equal(rr, r3[i]), i
    # de-interleave: BPTCDeInterleaveDMRData then extract == the fused form
    inter = np.zeros_like(x)
    idx = (np.arange(196) * 13) % 196
    inter[:, np.arange(196)] = x[:, idx]                  # air order such that Output[(13 i) % 196] = Input[i]
    out2, r32, errs2 = fec3.oracle_bptc(inter, 1)
    assert np.array_equal(out2, out) and np.array_equal(errs2, errs)
    d = np.zeros(196, np.uint8)
    r.BPTCDeInterleaveDMRData(C.c_void_p(np.ascontiguousarray(inter[5]).ctypes.data), C.c_void_p(d.ctypes.data))
    assert np.array_equal(d, x[5])


@needs_ref
def test_rs_12_9_oracle_vs_reference(built):
    r = orc.ref()
    o = orc.oracle()
    r.rs_12_9_correct_errors.restype = C.c_uint8
    r.rs_12_9_check_syndrome.restype = C.c_uint8
    rng = np.random.default_rng(12)
    # code words: start from zero (valid), add what the reference itself corrects back - and plain random words
    words = []
    for i in range(1500):
        w = np.zeros(12, np.uint8)
        k = int(rng.integers(0, 4))
        w[rng.choice(12, k, replace=False)] = rng.integers(1, 256, k)
        words.append(w)
    words += [rng.integers(0, 256, 12).astype(np.uint8) for _ in range(1500)]
    for w in words:
        a = w.copy()
        syn = np.zeros(6, np.uint8)
        r.rs_12_9_calc_syndrome(C.c_void_p(a.ctypes.data), C.c_void_p(syn.ctypes.data))
        found = C.c_uint8(0)
        if r.rs_12_9_check_syndrome(C.c_void_p(syn.ctypes.data)):
            res = r.rs_12_9_correct_errors(C.c_void_p(a.ctypes.data), C.c_void_p(syn.ctypes.data), C.byref(found))
        else:
            res = 0
        b = w.copy()
        s3 = np.zeros(3, np.uint8)
        f2 = C.c_uint8(0)
        res2 = o.orc_rs_12_9(C.c_void_p(b.ctypes.data), C.c_void_p(s3.ctypes.data), C.byref(f2))
        assert np.array_equal(s3, syn[:3]) and res2 == res and f2.value == found.value and np.array_equal(a, b), w
