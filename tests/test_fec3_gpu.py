"""GPU: DMR / NXDN block codes through the C-ABI (batched device calls, host variants, drop-ins with the reference's names)
against the restatement pinned to the compiled reference (tests/test_oracle_fec3.py).  Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import ddn
import fec3

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("code", list(fec3.CODES))
def test_block_code_batch_bit_exact(built, code):
    import torch
    rng = np.random.default_rng(70 + code)
    n = fec3.CODES[code][0]
    words = fec3.words_for(code, rng)
    if n > 16:
        words = np.concatenate([words, words[:5000] ^ rng.integers(0, 2, size=(1, n), dtype=np.uint8)])
    want_w, _, want_ok = fec3.oracle_decode(code, words)
    d = torch.from_numpy(np.ascontiguousarray(words)).cuda()
    ok = torch.zeros(len(words), dtype=torch.uint8, device="cuda")
    dec = torch.zeros((len(words), fec3.CODES[code][1]), dtype=torch.uint8, device="cuda")
    assert ddn.lib().ddn_fec_block_code_batch(code, d.data_ptr(), len(words), 1, dec.data_ptr(), ok.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), want_w) and np.array_equal(ok.cpu().numpy(), want_ok)


@pytest.mark.parametrize("code", [1, 2, 3, 4])
def test_hamming_multi_codeword_items(built, code):
    rng = np.random.default_rng(80 + code)
    n, k, _ = fec3.CODES[code]
    words = rng.integers(0, 2, size=(2000, 3 * n), dtype=np.uint8)
    words[::2] = 0
    for i in range(0, 2000, 2):
        words[i, rng.integers(0, 3 * n)] = 1
    want_w, want_dec, want_ok = fec3.oracle_decode(code, words, nb=3)
    w = words.copy()
    dec = np.zeros((2000, 3 * k), np.uint8)
    ok = np.zeros(2000, np.uint8)
    assert ddn.lib().ddn_fec_block_code_host(code, w.ctypes.data, 2000, 3, dec.ctypes.data, ok.ctypes.data) == 0
    assert np.array_equal(w, want_w) and np.array_equal(ok, want_ok)
    done = want_ok.astype(bool) | (code == 1)              # an item that stopped early leaves later copies unwritten
    assert np.array_equal(dec[done], want_dec[done])


def test_bptc_and_rs_batches(built):
    rng = np.random.default_rng(5)
    x = fec3.bptc_inputs(rng, 3000)
    want_out, want_r3, want_e = fec3.oracle_bptc(x, 0)
    out, r3, e = np.zeros((3000, 96), np.uint8), np.zeros((3000, 3), np.uint8), np.zeros(3000, np.uint32)
    assert ddn.lib().ddn_fec_bptc_196x96_host(x.ctypes.data, 0, 3000, out.ctypes.data, r3.ctypes.data, e.ctypes.data) == 0
    assert np.array_equal(out, want_out) and np.array_equal(r3, want_r3) and np.array_equal(e, want_e)
    inter = np.ascontiguousarray(x[:, (np.arange(196) * 13) % 196])
    out2 = np.zeros((3000, 96), np.uint8)
    assert ddn.lib().ddn_fec_bptc_196x96_host(inter.ctypes.data, 1, 3000, out2.ctypes.data, None, None) == 0
    assert np.array_equal(out2, want_out)
    # Reed-Solomon (12,9)
    o = fec3.orc.oracle()
    cw = np.zeros((4000, 12), np.uint8)
    for i in range(2000):
        k = int(rng.integers(0, 4))
        cw[i, rng.choice(12, k, replace=False)] = rng.integers(1, 256, k)
    cw[2000:] = rng.integers(0, 256, size=(2000, 12))
    want = cw.copy()
    wres, wf, wsyn = np.zeros(4000, np.uint8), np.zeros(4000, np.uint8), np.zeros((4000, 3), np.uint8)
    for i in range(4000):
        f = C.c_uint8(0)
        wres[i] = o.orc_rs_12_9(C.c_void_p(want[i].ctypes.data), C.c_void_p(wsyn[i].ctypes.data), C.byref(f))
        wf[i] = f.value
    got = cw.copy()
    res, fnd, syn = np.zeros(4000, np.uint8), np.zeros(4000, np.uint8), np.zeros((4000, 3), np.uint8)
    assert ddn.lib().ddn_fec_rs_12_9_host(got.ctypes.data, 4000, res.ctypes.data, fnd.ctypes.data, syn.ctypes.data) == 0
    assert np.array_equal(got, want) and np.array_equal(res, wres) and np.array_equal(fnd, wf) and np.array_equal(syn, wsyn)
    assert (wres == 1).sum() > 500 and (wres == 2).sum() > 500


def test_dropins_with_reference_names(built):
    """tests/fec/test_fec_block_codes.c-style calls: init, decode in place, bool result; BPTC KAT of test_fec_bptc_rs.c"""
    l = ddn.lib()
    l.InitAllFecFunction()
    rng = np.random.default_rng(1)
    for code, (n, k, name) in fec3.CODES.items():
        for _ in range(3):
            w = rng.integers(0, 2, n).astype(np.uint8)
            ww, dd, okk = fec3.oracle_decode(code, w[None])
            g = w.copy()
            dec = np.zeros(k, np.uint8)
            fn = getattr(l, name + "_decode")
            got = fn(g.ctypes.data, dec.ctypes.data, 1) if 1 <= code <= 4 else fn(g.ctypes.data)
            assert bool(got) == bool(okk[0]) and np.array_equal(g, ww[0])
    x = np.array(fec3.BPTC_KAT, np.uint8)
    out, r3 = np.zeros(96, np.uint8), np.zeros(3, np.uint8)
    assert l.BPTC_196x96_Extract_Data(x.ctypes.data, out.ctypes.data, r3.ctypes.data) == 0
    assert list(r3) == [1, 0, 1] and list(out) == [((i * 17) + (i // 5)) & 1 for i in range(96)]
    bad = x.copy()
    for row in range(5):
        for col in range(5):
            bad[1 + row * 15 + col] ^= 1
    assert l.BPTC_196x96_Extract_Data(bad.ctypes.data, out.ctypes.data, r3.ctypes.data) > 0     # test_fec_bptc_rs.c:58-63
    cw = np.zeros(12, np.uint8)
    cw[4] = 0x5A
    syn = np.zeros(6, np.uint8)
    l.rs_12_9_calc_syndrome(cw.ctypes.data, syn.ctypes.data)
    assert l.rs_12_9_check_syndrome(syn.ctypes.data) == 1
    f = C.c_uint8(0)
    assert l.rs_12_9_correct_errors(cw.ctypes.data, syn.ctypes.data, C.byref(f)) == 1 and f.value == 1 and not cw.any()
