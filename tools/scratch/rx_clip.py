import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
os.environ["DDN_RX_DBG"] = "8192"
import torch, ddn, orc
B, n, cpw = 4096, 48000, 16
for noise in (100.0, 400.0, 1500.0):
    base, _, _ = orc.synth_p25_disc(5, 64, n, frame_dibits=864, noise=noise)
    x = np.tile(base, (B // 64, 1))
    d = torch.from_numpy(x).cuda()
    rx = ddn.P25Rx(B, lock_symbols=840, use_matched_filter=1, channels_per_wave=cpw)
    ms = ddn.lib().ddn_p25_rx_max_symbols(rx.h, n)
    rec = torch.zeros((B, ms, 10), dtype=torch.uint8, device="cuda"); fl = torch.zeros((B, ms), dtype=torch.uint8, device="cuda"); cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    assert ddn.lib().ddn_p25_rx_run(rx.h, d.data_ptr(), n, rec.data_ptr(), fl.data_ptr(), cnt.data_ptr(), ms, None) == 0
    torch.cuda.synchronize()
    r = rec.cpu().numpy().reshape(B, -1)
    t = np.stack([r[c, -192:] for c in range(0, B, cpw)]).copy().view(np.int64).reshape(-1, 3, 8)[:, 0]
    # lane 0 of each workgroup counted its own symbols: n[0] = clipped, n[1] = lean symbols (kinds are not separated in this build)
    print("noise %.0f: lean symbols with a clipped window sample: %.2f %% (%d of %d, lane 0 of each workgroup)" % (noise, 100.0 * t[:, 0].sum() / max(1, t[:, 1].sum()), t[:, 0].sum(), t[:, 1].sum()))
