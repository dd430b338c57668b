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
    exe = str(tmp_path / "mixed_node_host")      # examples/mixed_node_host.c: the mixed batch of BASELINE configs[3], same way
    subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ddn.ROOT, "include"),
                           os.path.join(ddn.ROOT, "examples", "mixed_node_host.c"), "-L", lib_dir, "-ldsdneo_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    for parts in ("0", "3"):
        p = subprocess.run([exe, "96", "3", parts], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        assert "Gsamples/s" in p.stdout and "part 0: device 0, P25 0.." in p.stdout and "DMR records held" in p.stdout, p.stdout
        if parts == "3":
            assert "part 2: device" in p.stdout and "NXDN48 0..31" in p.stdout, p.stdout


def _mixed_outputs(mixed_handle, counts, n_call):
    """what a ddn_mixed_chain's three groups hold after a call: {name: array}, the rows of every group in channel order"""
    l = ddn.lib()
    Bp, Bd, Bn = counts
    out = {}
    if Bp:
        rm = ddn.P25ChainResults()
        assert l.ddn_p25_chain_get_results(l.ddn_mixed_chain_part(mixed_handle, 0), C.byref(rm)) == 0
        shp = ddn.P25ChainC(1, n_call)     # (for its shapes: they follow from samples_per_call alone)
        for name, dt, shape in (("d_new", np.int32, (Bp,)), ("d_counts", np.int32, (Bp,)), ("d_nid4", np.int32, (Bp, shp.F, 4)),
                                ("d_records10", np.uint8, (Bp, shp.stride, 10)), ("d_flags", np.uint8, (Bp, shp.stride)),
                                ("d_n_syncs", np.int32, (Bp,))):
            out["p25." + name] = shp.fetch(getattr(rm, name), dt, shape)
        tsbk = shp.fetch(rm.d_tsbk, np.uint8, (3, Bp, shp.F, 12))
        out["p25.d_tsbk"] = np.ascontiguousarray(tsbk.transpose(1, 0, 2, 3))          # channel first, like every other array here
        shp.close()
    for which, Bc in ((1, Bd), (2, Bn)):
        if not Bc:
            continue
        a = ddn.Fsk4ChainC(Bc, n_call, 0, handle=l.ddn_mixed_chain_part(mixed_handle, which))
        ra = a.results()
        for name, dt, shape in (("d_counts", np.int32, (Bc,)), ("d_new", np.int32, (Bc,)), ("d_n_sync", np.int32, (Bc,)),
                                ("d_records10", np.uint8, (Bc, ra.stride_symbols, 10)), ("d_flags", np.uint8, (Bc, ra.stride_symbols)),
                                ("d_valid", np.uint8, (Bc, ra.max_syncs)), ("d_sync_pos", np.int32, (Bc, ra.max_syncs))):
            out["%d.%s" % (which, name)] = a.fetch(getattr(ra, name), dt, shape)
    return out


@pytest.mark.parametrize("parts", [2, 3])
def test_mixed_node_of_parts_equals_one_mixed_chain(built, parts):
    """kind = DDN_NODE_MIXED (BASELINE configs[3] from C): ddn_mixed_partition gives every part its block of the P25, DMR and NXDN48
    groups, a part owns one ddn_mixed_chain and copies its rows out of the caller's pinned I/Q.  Wrapped onto one device, three calls
    and the flush give every channel what ONE ddn_mixed_chain over all channels gives it - records, flags, counts, syncs, NIDs, TSDU
    blocks, call by call."""
    import p25gen
    from conftest import golden
    l = ddn.lib()
    n_call, calls = 24000, 3
    Bp, Bd, Bn = 5, 4, 3
    rng = np.random.default_rng(21)
    dib = [np.concatenate([p25gen.make_frames(rng, 1, 0x293, crc=True, blocks=1 + (c + k) % 3)[0] for k in range(12 * calls)]) for c in range(Bp)]
    p25 = np.stack([p25gen.modulate_cu8(dib[c], n_call * calls, lead=250 + 31 * c, seed=c) for c in range(Bp)])

    def tiles(name, lo, B):
        iq = np.ascontiguousarray(golden(name)["iq"], np.uint8)
        return np.stack([iq[lo + 371 * c:lo + 371 * c + n_call * calls] for c in range(B)])

    dmr, nx = tiles("iq_dmr_t3_ras_cc.npz", 0, Bd), tiles("iq_nxdn48.npz", 60000, Bn)
    rows = np.concatenate([p25, dmr, nx])                       # the global channel index: [P25 | DMR | NXDN48]
    keep = []

    def upload(a):
        p = C.c_void_p()
        assert l.ddn_device_alloc(a.nbytes, C.byref(p)) == 0 and l.ddn_device_upload(p, a.ctypes.data, a.nbytes) == 0
        return p

    one = ddn.MixedChainC(Bp, Bd, Bn, n_call)
    want = []
    for k in range(calls):
        ps = [upload(np.ascontiguousarray(x[:, k * n_call:(k + 1) * n_call])) for x in (p25, dmr, nx)]
        one.run(*ps)
        one.wait()
        want.append(_mixed_outputs(one.h, (Bp, Bd, Bn), n_call))
        for p in ps:
            l.ddn_device_free(p)
    one.close()

    node = ddn.NodeC(Bp, n_call, n_devices=parts, kind=ddn.NODE_MIXED, n_dmr=Bd, n_nxdn48=Bn)
    assert node.parts == parts and l.ddn_node_kind_of(node.h) == ddn.NODE_MIXED and not node.chain(0)
    groups = [node.groups(p) for p in range(parts)]
    for p in range(parts):                                      # ddn_mixed_partition: a contiguous block of the global channel index
        f3, n3 = (C.c_int32 * 3)(), (C.c_int32 * 3)()
        assert l.ddn_mixed_partition(Bp, Bd, Bn, p, parts, f3, n3) == 0
        assert groups[p] == [(f3[g], n3[g]) for g in range(3)]
        assert node.info[p][1:] == ddn.node_partition(Bp + Bd + Bn, p, parts)
    assert [sum(groups[p][g][1] for p in range(parts)) for g in range(3)] == [Bp, Bd, Bn]
    decoded = 0
    for k in range(calls):
        piece = np.ascontiguousarray(rows[:, k * n_call:(k + 1) * n_call])
        h = _pinned(l, piece.nbytes, keep)
        C.memmove(h, piece.ctypes.data, piece.nbytes)
        node.run_host(h)
        node.wait()
        for p in range(parts):
            counts = tuple(n for _, n in groups[p])
            got = _mixed_outputs(node.chain_object(p), counts, n_call)
            for name, a in got.items():
                g = 0 if name.startswith("p25.") else int(name[0])
                f, n = groups[p][g]
                w = want[k][name][f:f + n]
                if name.endswith(("d_records10", "d_flags")):       # (beyond a row's count lies scratch)
                    cnt = got[name.rsplit(".", 1)[0] + ".d_counts"]
                    for c in range(n):
                        assert np.array_equal(a[c, :cnt[c]], w[c, :cnt[c]]), (k, p, name, c)
                elif name.endswith("d_sync_pos"):
                    ns = got["%d.d_n_sync" % g]
                    for c in range(n):
                        assert np.array_equal(a[c, :ns[c]], w[c, :ns[c]]), (k, p, name, c)
                elif name in ("p25.d_nid4", "p25.d_tsbk"):
                    ns = got["p25.d_n_syncs"]
                    for c in range(n):
                        assert np.array_equal(a[c, :ns[c]], w[c, :ns[c]]), (k, p, name, c)
                        decoded += int(ns[c])
                else:
                    assert np.array_equal(a, w), (k, p, name)
    assert decoded > 20 and sum(int(w["1.d_n_sync"].sum()) for w in want) > 0
    # a result getter on the part's own thread (what a host with several devices uses), and the flush of every group
    seen = []
    assert node.on_part(1, lambda chain, arg: (seen.append(int(chain)), 0)[1]) == 0 and seen == [node.chain_object(1)]
    node.flush()
    node.close()
    for p in keep:
        l.ddn_host_free_pinned(p)


def test_fsk4_node_equals_one_chain(built):
    """kind = DDN_NODE_FSK4: one DMR chain object per part, device-resident and host input"""
    from conftest import golden
    l = ddn.lib()
    n_call, B = 24000, 5
    iq = np.ascontiguousarray(golden("iq_dmr_t3_ras_cc.npz")["iq"], np.uint8)
    x = np.stack([iq[371 * c:371 * c + 2 * n_call] for c in range(B)])
    one = ddn.Fsk4ChainC(B, n_call, ddn.FSK4_DMR, rf_mod=2)
    cfg = ddn.Fsk4ChainConfig(0, 0, 0, 0, ddn.FSK4_DMR, 2, 0, 1, 1)
    node = ddn.NodeC(B, n_call, n_devices=2, kind=ddn.NODE_FSK4, chain_cfg=cfg)
    keep = []
    for k in range(2):
        piece = np.ascontiguousarray(x[:, k * n_call:(k + 1) * n_call])
        d = C.c_void_p()
        assert l.ddn_device_alloc(piece.nbytes, C.byref(d)) == 0 and l.ddn_device_upload(d, piece.ctypes.data, piece.nbytes) == 0
        one.run(d)
        r1 = one.results()
        want_cnt = one.fetch(r1.d_counts, np.int32, (B,))
        want_rec = one.fetch(r1.d_records10, np.uint8, (B, r1.stride_symbols, 10))
        want_ns = one.fetch(r1.d_n_sync, np.int32, (B,))
        l.ddn_device_free(d)
        if k == 0:      # host input: the node copies each part's block itself
            h = _pinned(l, piece.nbytes, keep)
            C.memmove(h, piece.ctypes.data, piece.nbytes)
            node.run_host(h)
        else:           # device input: one pointer per part
            ptrs = []
            for p, (_, f, n) in enumerate(node.info):
                sub = np.ascontiguousarray(piece[f:f + n])
                dp = C.c_void_p()
                assert l.ddn_node_device_alloc(node.h, p, sub.nbytes, C.byref(dp)) == 0
                assert l.ddn_node_device_upload(node.h, p, dp, sub.ctypes.data, sub.nbytes) == 0
                ptrs.append(dp)
            node.run_device(ptrs)
        node.wait()
        for p, (_, f, n) in enumerate(node.info):
            a = ddn.Fsk4ChainC(n, n_call, 0, handle=node.chain_object(p))
            ra = a.results()
            cnt = a.fetch(ra.d_counts, np.int32, (n,))
            rec = a.fetch(ra.d_records10, np.uint8, (n, ra.stride_symbols, 10))
            assert np.array_equal(cnt, want_cnt[f:f + n]) and np.array_equal(a.fetch(ra.d_n_sync, np.int32, (n,)), want_ns[f:f + n])
            for c in range(n):
                assert np.array_equal(rec[c, :cnt[c]], want_rec[f + c, :cnt[c]]), (k, p, c)
        if k == 1:
            for p, dp in enumerate(ptrs):
                l.ddn_node_device_free(node.h, p, dp)
    assert want_ns.sum() > 0
    node.flush()
    node.close()
    one.close()
    for p in keep:
        l.ddn_host_free_pinned(p)
