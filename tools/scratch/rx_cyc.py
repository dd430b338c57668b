import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
os.environ["DDN_RX_DBG"] = sys.argv[2] if len(sys.argv) > 2 else "8192"
import torch, ddn, orc
B, n = 4096, 48000
cpw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=864)
x = np.tile(base, (B // 64, 1))
if os.environ.get('DDN_BENCH_SPREAD'):  # frame phases spread over the whole frame period
    for c in range(B):
        x[c] = np.roll(x[c], (c * 977) % 8640)
d = torch.from_numpy(x).cuda()
rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1, channels_per_wave=cpw)
ms = ddn.lib().ddn_p25_rx_max_symbols(rx.h, n)
rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda")
fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(2):
    assert ddn.lib().ddn_p25_rx_run(rx.h, d.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
torch.cuda.synchronize()
r = rec.cpu().numpy().reshape(B, -1)
tails = np.stack([r[c, -192:] for c in range(0, B, cpw)]).copy().view(np.int64).reshape(-1, 3, 8)
names = ["recurrence", "loader", "winprep"]
for w in range(3):
    print("cpw", cpw, names[w], "busy/tile %.0f  wait/tile %.0f cycles (mean over %d workgroups)" % (tails[:, w, 0].mean() / 750, tails[:, w, 1].mean() / 750, tails.shape[0]))
t = tails[:, 0]
for k, nm in enumerate(["std", "general", "lean"]):
    print("cpw", cpw, nm, "trips/tile %.2f at %.0f cycles" % (t[:, 5 + k].mean() / 750, t[:, 2 + k].sum() / max(1, t[:, 5 + k].sum())))
