"""ddn_node (include/ddn_node.h): the C-side driver of all the devices of a node.  On a one-GPU box the device list wraps, so three
chain objects with a host thread each share the device: every channel's results must equal those of one chain object over all the
channels (channels are independent streams - SURVEY.md 8e - so how they are partitioned cannot show in any result)."""
import ctypes as C

import numpy as np
import pytest

import ddn
from test_chain_gpu import _stream

pytestmark = pytest.mark.gpu


def _pinned(l, nbytes, keep):
    p = C.c_void_p()
    assert l.ddn_host_alloc_pinned(nbytes, C.byref(p)) == 0
    keep.append(p)
    return p


def _out_set(l, B, F, Fv, st, E, keep):
    S, V = B * F, B * Fv * 9
    shapes = {"records10": (np.uint8, (B, st, 10)), "flags": (np.uint8, (B, st)), "counts": (np.int32, (B,)),
              "events": (np.int32, (B, E, 4)), "n_events": (np.int32, (B,)), "event_data": (np.int32, (B, E, 4)),
              "nid4": (np.int32, (S, 4)), "tsbk": (np.uint8, (3, S, 12)), "pcm": (np.float32, (V, 160))}
    o, v = ddn.P25ChainHostOut(), {}
    for name, (dt, shp) in shapes.items():
        nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
        p = _pinned(l, nbytes, keep)
        C.memset(p, 0, nbytes)
        setattr(o, name, p.value)
        v[name] = np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dt).reshape(shp)
    return o, v


@pytest.mark.parametrize("parts", [1, 3])
def test_node_of_parts_equals_one_chain_over_all_channels(built, parts):
    l = ddn.lib()
    B, n_call, calls = 7, 16384, 4
    iq = _stream(B, n_call * calls)
    keep = []
    h_iq = []
    for k in range(calls):
        part = np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call])
        p = _pinned(l, part.nbytes, keep)
        C.memmove(p, part.ctypes.data, part.nbytes)
        h_iq.append(p)

    one = ddn.P25ChainC(B, n_call)
    F, Fv, st, E = one.F, one.Fv, one.stride, one.E
    want = []
    for k in range(calls):
        o, v = _out_set(l, B, F, Fv, st, E, keep)
        one.run_host(h_iq[k], o)
        one.wait()
        want.append({a: b.copy() for a, b in v.items()})
    one.flush()
    one.close()

    node = ddn.NodeC(B, n_call, n_devices=parts)
    assert node.parts == parts
    assert [(f, n) for _, f, n in node.info] == [ddn.node_partition(B, r, parts) for r in range(parts)]
    voiced = 0
    for k in range(calls):
        sets = [_out_set(l, n, F, Fv, st, E, keep) for _, _, n in node.info]
        node.run_host(h_iq[k], [o for o, _ in sets])
        node.wait()
        w = want[k]
        for (_, f, n), (_, v) in zip(node.info, sets):
            ch = slice(f, f + n)
            for name in ("records10", "flags", "counts", "n_events"):
                assert np.array_equal(v[name], w[name][ch]), (k, f, name)
            for c in range(n):
                ne = int(v["n_events"][c])
                assert np.array_equal(v["events"][c, :ne], w["events"][f + c, :ne]), (k, f, c)
                assert np.array_equal(v["event_data"][c, :ne], w["event_data"][f + c, :ne]), (k, f, c)
            assert np.array_equal(v["nid4"], w["nid4"][f * F:(f + n) * F]), (k, f)
            assert np.array_equal(v["tsbk"], w["tsbk"][:, f * F:(f + n) * F]), (k, f)
            a, b = v["pcm"].view(np.uint32), w["pcm"][f * Fv * 9:(f + n) * Fv * 9].view(np.uint32)
            assert np.array_equal(a, b), (k, f)
            voiced += int(np.count_nonzero(a))
    assert voiced > 0
    node.flush()
    node.close()
    for p in keep:
        l.ddn_host_free_pinned(p)


def test_node_device_form_and_errors(built):
    l = ddn.lib()
    B, n_call = 5, 12288
    iq = _stream(B, n_call * 2)
    node = ddn.NodeC(B, n_call, n_devices=2)
    one = ddn.P25ChainC(B, n_call)
    for k in range(2):
        part = np.ascontiguousarray(iq[:, k * n_call:(k + 1) * n_call])
        ptrs = []
        for p, (_, f, n) in enumerate(node.info):
            sub = np.ascontiguousarray(part[f:f + n])
            d = C.c_void_p()
            assert l.ddn_node_device_alloc(node.h, p, sub.nbytes, C.byref(d)) == 0
            assert l.ddn_node_device_upload(node.h, p, d, sub.ctypes.data, sub.nbytes) == 0
            ptrs.append(d)
        node.run_device(ptrs)
        node.wait()
        d = C.c_void_p()
        assert l.ddn_device_alloc(part.nbytes, C.byref(d)) == 0 and l.ddn_device_upload(d, part.ctypes.data, part.nbytes) == 0
        one.run(d)
        r1 = one.results()
        want = one.fetch(r1.d_counts, np.int32, (B,)), one.fetch(r1.d_records10, np.uint8, (B, one.stride, 10))
        for p, (_, f, n) in enumerate(node.info):
            r = ddn.P25ChainResults()
            assert l.ddn_p25_chain_get_results(node.chain(p), C.byref(r)) == 0
            cnt = np.zeros(n, np.int32)
            rec = np.zeros((n, one.stride, 10), np.uint8)
            assert l.ddn_node_device_download(node.h, p, cnt.ctypes.data, r.d_counts, cnt.nbytes) == 0
            assert l.ddn_node_device_download(node.h, p, rec.ctypes.data, r.d_records10, rec.nbytes) == 0
            assert np.array_equal(cnt, want[0][f:f + n]) and cnt.max() > 0
            for c in range(n):
                assert np.array_equal(rec[c, :cnt[c]], want[1][f + c, :cnt[c]]), (k, f, c)
            l.ddn_node_device_free(node.h, p, ptrs[p])
        l.ddn_device_free(d)
    node.close()
    one.close()
    with pytest.raises(ddn.DdnError):
        ddn.NodeC(0, n_call)
    assert l.ddn_node_run_host(None, None, None) == -1 and l.ddn_node_parts(None) == 0


def test_c_example_runs_all_devices(built, tmp_path):
    """examples/p25_node_host.c: compiled with plain gcc, run here over the visible devices and with the device list wrapped to 2"""
    import os
    import subprocess
    exe = str(tmp_path / "p25_node_host")
    lib_dir = os.path.dirname(ddn.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ddn.ROOT, "include"),
                           os.path.join(ddn.ROOT, "examples", "p25_node_host.c"), "-L", lib_dir, "-ldsdneo_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    for parts in ("0", "2"):
        p = subprocess.run([exe, "256", "4", parts], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert "Gsamples/s" in p.stdout and "part 0: device 0, channels 0.." in p.stdout, p.stdout
        if parts == "2":
            assert "part 1: device" in p.stdout and "channels 128..255" in p.stdout
