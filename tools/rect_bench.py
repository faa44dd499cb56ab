import sys, time, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
ge.build()
import torch
import m4ri_rust_amd as pkg
from m4ri_rust_amd import device as dev
def t(m,l,n,acc=True,algo="m4rm",reps=5):
    A=dev.DMat.random(m,l,1); B=dev.DMat.random(l,n,2); C=dev.DMat.random(m,n,3)
    dev.mul(A,B,C,accumulate=acc,algo=algo); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(reps): dev.mul(A,B,C,accumulate=acc,algo=algo)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/reps
    print("m=%d l=%d n=%d acc=%d %s: %.3f ms  %.3e bitops/s"%(m,l,n,acc,algo,dt*1e3,2.0*m*l*n/dt),flush=True)
t(65536,2048,63488)
t(65536,2048,65536)
t(65536,2048,65536,acc=False)
t(65536,4096,65536)
t(65536,512,65536)
t(65536,65536,65536,acc=False,reps=2)
t(65536,2048,63488,algo="auto")
t(32768,2048,32768)
