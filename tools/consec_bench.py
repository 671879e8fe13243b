#!/usr/bin/env python3
"""Video order vs independent pairs: with PAIRS_CONSECUTIVE (DenseOF.py:525, prev_gray = gray) every frame's level images
and polynomial expansion are computed once and serve two pairs.  Device resident, 1920x1080, levels=5."""
import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import hackathonopticalflow_amd as ofa
W,H,B=1920,1080,512
g=torch.Generator(device='cuda'); g.manual_seed(1)
frames=torch.randint(0,256,(B+1,H,W),dtype=torch.uint8,device='cuda',generator=g)
flow=torch.empty((B,H,W,2),dtype=torch.float32,device='cuda')
P=len(ofa.grid_points(W,H,30)); mask=torch.zeros((B,P),dtype=torch.uint8,device='cuda'); v=torch.zeros_like(mask)
eng=ofa.FarnebackEngine(W,H,256,0,levels=5)
st=torch.cuda.current_stream().cuda_stream
for mode,nf in ((ofa.PAIRS_CONSECUTIVE,B+1),(ofa.PAIRS_INDEPENDENT,B)):
    npairs = nf-1 if mode==ofa.PAIRS_CONSECUTIVE else nf//2
    run=lambda: eng.calc_batch_device(frames,nf,W,H,mode,flow,mask,v,stream=st)
    run(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
    print("consecutive" if mode else "independent", npairs, "pairs", round(dt*1e3,2), "ms ->", round(npairs/dt,1), "pairs/s")
