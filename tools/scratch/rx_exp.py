"""rx loop timing on the bench traffic: voice-only / control-only / mixed x channels-per-wave."""
import sys, os, json, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", p) for p in ("tests", "dsd-neo_amd/bindings", ".")]
import numpy as np, torch
import bench, ddn, ddn_chain
n, B = 48000, 4096
voice, ctrl = bench.make_base_traffic(n)
fe = ddn.Batch(B, block_len=8192)
def disc_of(kind):
    if kind == "voice": src = np.tile(voice, (B // 64, 1, 1))
    elif kind == "ctrl": src = np.tile(ctrl, (B // 64, 1, 1))
    else:
        src = np.empty((B, n, 2), np.uint8); src[0::2] = np.tile(voice, (B // 128, 1, 1)); src[1::2] = np.tile(ctrl, (B // 128, 1, 1))
    d = torch.from_numpy(src).cuda(); out = torch.zeros((B, n), dtype=torch.float32, device="cuda")
    fe.reset(); fe.run_device(d.data_ptr(), n, out.data_ptr()); torch.cuda.synchronize(); return out
l = ddn.lib()
for kind in ("voice", "ctrl", "mixed"):
    x = disc_of(kind)
    lock = np.full(B, 840 if kind == "voice" else 156, np.int32)
    if kind == "mixed": lock[1::2] = 156; lock[0::2] = 840
    for cpw in (8, 16, 32):
        rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1, channels_per_wave=cpw)
        l.ddn_p25_rx_set_lock_symbols(rx.h, lock.ctypes.data)
        l.ddn_p25_rx_set_timing(rx.h, 1)
        ms = l.ddn_p25_rx_max_symbols(rx.h, n)
        rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda"); fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
        ts = []
        for it in range(4):
            assert l.ddn_p25_rx_run(rx.h, x.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
            t2 = np.zeros(2, np.float32); l.ddn_p25_rx_get_timing(rx.h, t2.ctypes.data); ts.append(float(t2[1]))
        inframe = int((fl & 1).ne(0).sum().item()); tot = int(cnt.sum().item())
        print(json.dumps({"kind": kind, "cpw": cpw, "rx_ms": round(min(ts[1:]), 3), "first_ms": round(ts[0], 3), "symbols": tot, "in_frame_frac": round(inframe / tot, 3)}), flush=True)
