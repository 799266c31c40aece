import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from efgh_amd import synthetic as syn, lattice
from efgh_amd.nets import EFGHBackbone
raw=(768,2560)
m=EFGHBackbone(syn.default_args(raw,'cuda')).cuda().eval()
b=syn.make_batch(raw,131072,4)
pc=torch.from_numpy(b['pc']).cuda()
SC=(1.0,0.75,0.5,0.25,0.125)
with torch.no_grad():
    for _ in range(3): m.E(pc)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): m.E(pc)
    torch.cuda.synchronize(); print('E forward wall %.3f ms'%((time.perf_counter()-t0)/20*1e3))
    for _ in range(3): lattice.build_pyramid_batched(pc,SC)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): lattice.build_pyramid_batched(pc,SC)
    torch.cuda.synchronize(); print('pyramid wall %.3f ms'%((time.perf_counter()-t0)/20*1e3))
    # host time only: time to return from the call
    t0=time.perf_counter()
    for _ in range(20): lv=lattice.build_pyramid_batched(pc,SC)
    print('pyramid host-return %.3f ms'%((time.perf_counter()-t0)/20*1e3))
