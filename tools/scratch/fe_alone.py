import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ddn
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1365, 48000
x = torch.randint(0, 256, (B, n, 2), dtype=torch.uint8, device="cuda")
l = ddn.lib()
for name, mk in (("p25", lambda: ddn.P25ChainC(B, n)), ("dmr", lambda: ddn.Fsk4ChainC(B, n, ddn.FSK4_DMR, rf_mod=2)), ("nxdn", lambda: ddn.Fsk4ChainC(B, n, ddn.FSK4_NXDN48))):
    ch = mk()
    fn = l.ddn_p25_chain_stage if name == "p25" else l.ddn_fsk4_chain_stage
    for st in (0, 1, 2):
        ts = []
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert fn(ch.h, st, x.data_ptr(), None) == 0
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(name, "stage", st, "%.3f ms" % np.median(ts[1:]))
    ch.close()
