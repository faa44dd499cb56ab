import sys, time
sys.path.insert(0,'/root/repo')
import torch
import m4ri_rust_amd
from m4ri_rust_amd import device as dev
for (r,c) in [(65536,65536),(16384,16384),(65536,64),(64,65536),(20000,30001),(4096,4096),(1048576,256)]:
    S = dev.DMat.random(r,c,1); D = dev.DMat(c,r)
    for _ in range(5): dev.transpose(S, D)
    torch.cuda.synchronize(); t0=time.perf_counter()
    reps=20
    for _ in range(reps): dev.transpose(S, D)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    print(r,c,'%.3f ms'%(dt*1e3),'%.0f GB/s'%(2*r*c/8/dt/1e9), flush=True)
