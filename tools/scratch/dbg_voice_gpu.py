import sys; sys.path[:0]=['tests','dsd-neo_amd/bindings']
import numpy as np, torch
import test_e2e_voice as t, ddn_chain
iq,lock,sent=t._traffic()
ch=ddn_chain.P25Chain(torch,t.B,t.N,lock)
ch.run(torch.from_numpy(iq).cuda()); torch.cuda.synchronize()
F=ch.F
print("cnt",ch.cnt.cpu().numpy())
print("nid0",ch.nid.cpu().numpy().reshape(t.B,F,4)[0,:6])
print("v_ldu",ch.v_ldu.cpu().numpy().reshape(t.B,F)[0,:6])
for i in (0,1):
    print("rs_st",i,ch.rs_st[i].cpu().numpy().reshape(t.B,F)[0,:6])
    d=ch.rs_d[i].cpu().numpy().reshape(t.B,F,-1,6)[0,:5]
    print((d*(1<<np.arange(5,-1,-1))[None,None,:]).sum(axis=2))
ns=np.zeros(t.B,np.int32); pos=np.zeros((t.B,F),np.int32)
ch.l.ddn_p25p1_framer_get_syncs(ch.fr, ns.ctypes.data, pos.ctypes.data); print(ns, pos[0,:6])
print([w for w in sent[0]['words']])
