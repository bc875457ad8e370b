import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd.fused_policy import ACTOR_HEAD, PackedNetworks
for (W,H) in ((5,4),(9,5)):
    P=(W-2)*(H-2)
    g=torch.Generator().manual_seed(0); r=lambda *s: (torch.randn(*s,generator=g)*0.1).cuda()
    sd={"base.cnn.cnn.0.weight": r(32,20,3,3),"base.cnn.cnn.0.bias": r(32),"base.cnn.cnn.3.weight": r(64,32*P),"base.cnn.cnn.3.bias": r(64),
        "base.cnn.cnn.5.weight": r(64,64),"base.cnn.cnn.5.bias": r(64),ACTOR_HEAD+".weight": r(6,64),ACTOR_HEAD+".bias": r(6)}
    nets=PackedNetworks((W,H,20),16,"cuda")
    for i in range(16): nets.pack(i,sd)
    torch.cuda.synchronize()
    for chk in (True, False):
        t0=time.perf_counter()
        for rep in range(5):
            for i in range(16): nets.pack(i,sd,check_finite=chk)
        torch.cuda.synchronize()
        print(W,H,"check_finite",chk,"us per pack",(time.perf_counter()-t0)/80*1e6)
