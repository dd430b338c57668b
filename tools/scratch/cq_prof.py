import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "dsd-neo_amd/bindings")
import torch, ddn, orc
B, sps = 4096, 5
q1 = orc.synth_dqpsk_f32(12, 8, 4808, sps)
nq = q1.shape[1]
d_q = torch.from_numpy(np.tile(q1, (B // 8, 1, 1))).cuda()
cq = ddn.CqpskBatch(B, rate=24000, block_len=4096)
l = ddn.lib()
strd = l.ddn_cqpsk_max_symbols(cq.h, nq)
d_s = torch.zeros((B, strd), dtype=torch.float32, device="cuda")
d_c = torch.zeros(B, dtype=torch.int32, device="cuda")
for _ in range(3):
    l.ddn_cqpsk_run(cq.h, d_q.data_ptr(), nq, d_s.data_ptr(), strd, d_c.data_ptr(), None)
torch.cuda.synchronize()
