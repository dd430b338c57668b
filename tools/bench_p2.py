#!/usr/bin/env python3
"""The P25 Phase 2 stages alone (for rocprofv3 --kernel-trace --stats): ddn_p25p2_sync_cut_batch and ddn_p25p2_groups_batch over
4096 channels x 6 groups of synthetic TDMA traffic (tests/p2seq.make_stream), `reps` calls each.  One JSON line per stage."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsd-neo_amd", "bindings"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import ddn
    import p2seq
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    l = ddn.lib()
    rng = np.random.default_rng(1)
    st = torch.cuda.current_stream().cuda_stream
    Cn, G = 4096, 6
    base = [p2seq.make_stream(rng, G, 0xBEE00 + k, 0x164, 0x161, start_sf=int(rng.integers(0, 12)), noise=0.002) for k in range(16)]
    gb = np.stack([base[c % 16][0] for c in range(Cn)])
    gl = np.stack([base[c % 16][1] for c in range(Cn)])
    obj = ddn.P25P2Groups([((0xBEE00 + c % 16) << 24) | (0x164 << 12) | 0x161 for c in range(Cn)])
    tb, tl = torch.from_numpy(gb).cuda(), torch.from_numpy(gl).cuda()
    nr = Cn * G * 4
    o_info, o_pay = torch.zeros((nr, 8), dtype=torch.int32, device="cuda"), torch.zeros((nr, 180), dtype=torch.uint8, device="cuda")
    o_fr, o_rel = torch.zeros((nr, 384), dtype=torch.uint8, device="cuda"), torch.zeros((nr, 384), dtype=torch.uint8, device="cuda")
    o_ess = torch.zeros((nr, 96), dtype=torch.uint8, device="cuda")

    def run():
        rc = l.ddn_p25p2_groups_batch(tb.data_ptr(), tl.data_ptr(), Cn, G, None, obj.seed.data_ptr(), obj.state.data_ptr(), 64, o_info.data_ptr(),
                                      o_pay.data_ptr(), o_fr.data_ptr(), o_rel.data_ptr(), o_ess.data_ptr(), st)
        assert rc == 0, rc

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    info = o_info.cpu().numpy()
    acts = np.bincount(info[:, 4], minlength=11).tolist()
    print(json.dumps({"stage": "p25p2_groups", "channels": Cn, "groups": G, "timeslots": nr, "ms": round(ms, 3),
                      "timeslots_per_s": round(nr / ms * 1e3, 1), "actions": acts}))


if __name__ == "__main__":
    main()
