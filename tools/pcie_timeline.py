"""Workload + summariser for the PCIe-inclusive step's timeline (tools/pcie_timeline.sh runs it under
rocprofv3 --kernel-trace --memory-copy-trace):
    python tools/pcie_timeline.py run <mode>        mode: h2d | compact | all | resident
    python tools/pcie_timeline.py summarise <dir>   per step of the steady state: when each copy and each stage's kernels ran"""
import csv
import ctypes as C
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mode, steps=12):
    import bench
    import numpy as np
    import torch
    import ddn
    B, n = bench.B_PER_GPU, bench.N_SAMPLES
    voice, ctrl = bench.make_base_traffic(n)
    dev = torch.device("cuda:0")
    kinds = [bench.channel_source(c) for c in range(B)]
    d_iq = torch.empty((B, n, 2), dtype=torch.uint8, device=dev)
    dv, dc = torch.from_numpy(voice).to(dev), torch.from_numpy(ctrl).to(dev)
    is_v = torch.tensor([k == "voice" for k, _ in kinds], device=dev)
    d_iq[is_v] = dv[torch.tensor([b for k, b in kinds if k == "voice"], device=dev)]
    d_iq[~is_v] = dc[torch.tensor([b for k, b in kinds if k == "ctrl"], device=dev)]
    torch.cuda.synchronize()
    chain = ddn.P25ChainC(B, n, block_len=bench.BLOCK)
    l = ddn.lib()
    S, V, st, E = B * chain.F, B * chain.Fv * 9, chain.stride, chain.E

    def pin(nb):
        p = C.c_void_p()
        assert l.ddn_host_alloc_pinned(nb, C.byref(p)) == 0
        return p

    if mode == "resident":
        for _ in range(3):
            chain.run_pipelined(d_iq.data_ptr())
        chain.wait()
        t0 = time.perf_counter()
        for _ in range(steps):
            chain.run_pipelined(d_iq.data_ptr())
        chain.wait()
        print("resident %.3f ms per step" % ((time.perf_counter() - t0) / steps * 1e3))
        return
    h_iq = [pin(B * n * 2) for _ in range(2)]
    for p in h_iq:
        assert l.ddn_device_download(p, d_iq.data_ptr(), B * n * 2) == 0
    sizes = {"h2d": {}, "compact": {"counts": B * 4, "events": B * E * 16, "n_events": B * 4, "event_data": B * E * 16, "nid4": S * 16,
                                   "tsbk": 3 * S * 12, "records2": B * st * 2, "pcm_dense": (V // 3) * 640, "pcm_slot": (V // 3) * 4,
                                   "pcm_count": 4},
             "all": {"records10": B * st * 10, "flags": B * st, "counts": B * 4, "events": B * E * 16, "n_events": B * 4,
                     "event_data": B * E * 16, "nid4": S * 16, "tsbk": 3 * S * 12, "pcm": V * 640}}[mode]
    outs = []
    for _ in range(3):
        o = ddn.P25ChainHostOut()
        for k, nb in sizes.items():
            setattr(o, k, pin(nb).value)
        if "pcm_dense" in sizes:
            o.pcm_dense_frames = V // 3
        outs.append(o)
    for k in range(3):
        chain.run_host(h_iq[k & 1], outs[k % 3] if sizes else None)
    chain.wait()
    t0 = time.perf_counter()
    for k in range(steps):
        chain.run_host(h_iq[(k + 1) & 1], outs[k % 3] if sizes else None)
    chain.wait()
    print("%s %.3f ms per step, %.1f MB out, D2H route 0x%x" % (mode, (time.perf_counter() - t0) / steps * 1e3, sum(sizes.values()) / 1e6,
                                                                l.ddn_p25_chain_d2h_route(chain.h)))


def summarise(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[:24] ))
    rows.sort()
    # steps are delimited by the front-end kernel
    fe = [i for i, r in enumerate(rows) if r[2].startswith("k_front_end")]
    if len(fe) < 5:
        print("too few steps", len(rows))
        return
    t_ref = rows[fe[-4]][0]
    lo, hi = fe[-4], fe[-1]
    print("timeline of three steady-state steps (ms from the first one's front end); kernels < 0.15 ms are summed per gap")
    small_t, small_n, small_s, small_e = 0.0, 0, None, None
    for s, e, name in rows[lo - 40 if lo >= 40 else 0:hi]:
        if e < t_ref - 3e6:
            continue
        dur = (e - s) / 1e6
        if dur < 0.15 and not name.startswith("COPY"):
            small_t += dur
            small_n += 1
            small_s = s if small_s is None else small_s
            small_e = e
            continue
        if small_n:
            print("   %8.3f .. %8.3f   (%d small kernels, %.3f ms busy)" % ((small_s - t_ref) / 1e6, (small_e - t_ref) / 1e6, small_n, small_t))
            small_t, small_n, small_s = 0.0, 0, None
        print("   %8.3f .. %8.3f  %7.3f  %s" % ((s - t_ref) / 1e6, (e - t_ref) / 1e6, dur, name))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarise(sys.argv[2])
