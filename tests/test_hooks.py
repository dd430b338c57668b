"""The consumer-side seam: ddn_stream_set + ddn_hooks_read / ddn_hooks_return_pwr serve the
reference's stream-read hook contract (include/dsd-neo/runtime/rtl_stream_io_hooks.h:25-32; callers read 512 floats or 1,
block until >= 1 is available, < 0 on end of stream).  Host-only code: runs without a GPU.  The GPU test feeds it from the
batched front end and checks a decoder-style reader sees exactly the oracle's discriminator stream."""
import ctypes as C
import threading

import numpy as np
import pytest

import ddn
import orc


def _reader(ctx, counts, out, rc):
    l = ddn.lib()
    buf = np.zeros(512, np.float32)
    got = C.c_int(0)
    i = 0
    while True:
        r = l.ddn_hooks_read(ctx, buf.ctypes.data, counts[i % len(counts)], C.byref(got))
        i += 1
        if r < 0:
            rc.append(r)
            return
        assert 1 <= got.value <= counts[(i - 1) % len(counts)]
        out.append(buf[:got.value].copy())


def test_stream_set_contract(built):
    l = ddn.lib()
    B, cap = 5, 1000                       # queue shorter than what is pushed: the producer has to wait for readers
    h = C.c_void_p()
    assert l.ddn_stream_set_create(B, cap, 48000, 0, 4800, 4, 4, C.byref(h)) == 0
    assert l.ddn_stream_set_create(0, cap, 48000, 0, 4800, 4, 4, C.byref(C.c_void_p())) != 0
    assert l.ddn_stream_set_ctx(h, B) is None and l.ddn_stream_set_ctx(h, -1) is None
    ctx = [l.ddn_stream_set_ctx(h, c) for c in range(B)]
    rng = np.random.default_rng(1)
    data = rng.normal(0, 1, (B, 7000)).astype(np.float32)
    outs, rcs = [[] for _ in range(B)], [[] for _ in range(B)]
    th = [threading.Thread(target=_reader, args=(ctx[c], [512, 1, 1, 512, 7], outs[c], rcs[c])) for c in range(B)]
    for t in th:
        t.start()
    for a, b in [(0, 1), (1, 300), (300, 3000), (3000, 7000)]:
        chunk = np.ascontiguousarray(data[:, a:b])
        assert l.ddn_stream_set_push(h, chunk.ctypes.data, b - a, b - a, None) == 0
    for c in range(B):
        assert l.ddn_stream_set_set_power(h, c, 0.5 + c) == 0
    l.ddn_stream_set_close(h)
    for t in th:
        t.join(30)
        assert not t.is_alive()
    for c in range(B):
        assert np.array_equal(np.concatenate(outs[c]), data[c]) and rcs[c] == [-1]
        assert l.ddn_hooks_return_pwr(ctx[c]) == 0.5 + c
        assert l.ddn_hooks_output_rate_hz(ctx[c]) == 48000 and l.ddn_hooks_output_kind(ctx[c]) == 0
        r, lv, pr = C.c_int(), C.c_int(), C.c_int()
        assert l.ddn_hooks_symbol_profile(ctx[c], C.byref(r), C.byref(lv), C.byref(pr)) == 0
        assert (r.value, lv.value, pr.value) == (4800, 4, 4) and l.ddn_hooks_stream_generation(ctx[c]) == 1
    # after close: reads fail at once, pushes are refused
    got = C.c_int(7)
    buf = np.zeros(4, np.float32)
    assert l.ddn_hooks_read(ctx[0], buf.ctypes.data, 4, C.byref(got)) < 0 and got.value == 0
    assert l.ddn_stream_set_push(h, data.ctypes.data, 10, 7000, None) != 0
    assert l.ddn_hooks_read(None, buf.ctypes.data, 4, C.byref(got)) < 0
    l.ddn_stream_set_destroy(h)


def test_stream_set_counts_and_generation(built):
    l = ddn.lib()
    h = C.c_void_p()
    assert l.ddn_stream_set_create(3, 64, 4800, 1, 4800, 4, 5, C.byref(h)) == 0
    rows = np.arange(3 * 10, dtype=np.float32).reshape(3, 10)
    cnt = np.array([10, 4, 0], np.int32)                       # ragged per-channel symbol counts (CQPSK output)
    assert l.ddn_stream_set_push(h, rows.ctypes.data, 10, 10, cnt.ctypes.data) == 0
    buf = np.zeros(16, np.float32)
    got = C.c_int()
    assert l.ddn_hooks_read(l.ddn_stream_set_ctx(h, 1), buf.ctypes.data, 16, C.byref(got)) == 0 and got.value == 4
    assert np.array_equal(buf[:4], rows[1, :4])
    assert l.ddn_stream_set_bump_generation(h) == 0            # retune: what was queued for channel 0 is dropped
    assert l.ddn_hooks_stream_generation(l.ddn_stream_set_ctx(h, 0)) == 2
    assert l.ddn_stream_set_push(h, rows.ctypes.data, 2, 10, None) == 0
    assert l.ddn_hooks_read(l.ddn_stream_set_ctx(h, 0), buf.ctypes.data, 16, C.byref(got)) == 0 and got.value == 2
    assert np.array_equal(buf[:2], rows[0, :2]) and l.ddn_hooks_output_kind(l.ddn_stream_set_ctx(h, 0)) == 1
    l.ddn_stream_set_destroy(h)


@pytest.mark.gpu
def test_hook_reader_sees_the_oracle_stream(built):
    l = ddn.lib()
    B, n = 6, 20000
    iq = orc.synth_c4fm_cu8(3, B, n)
    fe = ddn.Batch(B, block_len=4096)
    h = C.c_void_p()
    assert l.ddn_stream_set_create(B, 8192, 48000, 0, 4800, 4, 4, C.byref(h)) == 0
    outs, rcs = [[] for _ in range(B)], [[] for _ in range(B)]
    th = [threading.Thread(target=_reader, args=(l.ddn_stream_set_ctx(h, c), [512, 1], outs[c], rcs[c])) for c in range(B)]
    for t in th:
        t.start()
    for a in range(0, n, 4096):                                 # one batch interval per reference block
        b = min(n, a + 4096)
        disc = fe.run_host(np.ascontiguousarray(iq[:, a:b]), b - a)
        assert l.ddn_stream_set_push(h, disc.ctypes.data, b - a, b - a, None) == 0
    l.ddn_stream_set_close(h)
    for t in th:
        t.join(60)
    for c in range(B):
        want = orc.OracleFrontEnd().run_cu8(iq[c], 4096)
        got = np.concatenate(outs[c])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), c
    l.ddn_stream_set_destroy(h)
