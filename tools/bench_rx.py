"""Time the batched P25p1 receive loop (ddn_p25_rx_run) on device-resident discriminator samples.
usage: python tools/bench_rx.py [B] [n] [cpw...]   -> JSON lines (also appended to gpurun_out/bench_rx.jsonl)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dsd-neo_amd", "bindings"))
import torch  # noqa: E402

import ddn  # noqa: E402
import orc  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
    cpws = [int(a) for a in sys.argv[3:]] or [16, 32, 64]
    base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=864)
    x = np.tile(base, (B // 64 + 1, 1))[:B].copy()
    if os.environ.get("DDN_BENCH_SPREAD"):  # frame phases spread over the whole frame period instead of within ~40 symbols
        for c in range(B):
            x[c] = np.roll(x[c], (c * 977) % 8640)
    d = torch.from_numpy(x).cuda()
    os.makedirs("gpurun_out", exist_ok=True)
    for filt in (1, 0):
        for cpw in cpws:
            rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=filt, channels_per_wave=cpw)
            ms = ddn.lib().ddn_p25_rx_max_symbols(rx.h, n)
            rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
            fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
            cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
            ts = []
            for it in range(4):
                torch.cuda.synchronize()
                t = time.perf_counter()
                rc = ddn.lib().ddn_p25_rx_run(rx.h, d.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
                assert rc == 0
            line = {"stage": "p25_rx", "B": B, "n": n, "cpw": cpw, "matched_filter": filt, "ms": min(ts[1:]) * 1e3,
                    "Msamples_per_s": B * n / min(ts[1:]) / 1e6, "symbols": int(cnt.sum().item()),
                    "syncs": int((fl & 2).ne(0).sum().item())}
            print(json.dumps(line), flush=True)
            with open("gpurun_out/bench_rx.jsonl", "a") as f:
                f.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
