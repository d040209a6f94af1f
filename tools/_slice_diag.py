import sys, os, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch.nn.functional as F
from test_gpu_kernels import run_conv, rnd
B=6
def chk(name, fn):
    full = fn(slice(0,B)); parts = torch.cat([fn(slice(0,4)), fn(slice(4,6))])
    print(f"{name:40s} {float((full-parts).abs().max()):.3e}")
for prec in (96, 32, 0):
  for (Cin,Cout,S) in ((128,128,10),(64,128,20),(128,128,20),(256,128,10),(128,64,20)):
    x = rnd(1,B,Cin,S,S); w = rnd(2,Cout,Cin,3,3)/30; b = rnd(3,Cout)
    chk(f"3x3 plain {Cin}->{Cout}@{S} prec{prec}", lambda s: run_conv(x[s],None,w,b,3,prec))
    A, Bv = rnd(4,B,Cin)*0.5+1, rnd(5,B,Cin)
    chk(f"3x3 affine {Cin}->{Cout}@{S} prec{prec}", lambda s: run_conv(x[s],None,w,b,3,8|prec,A=A[s],Bv=Bv[s]))
    w1 = rnd(6,Cout,Cin,1,1)/8
    chk(f"1x1 plain {Cin}->{Cout}@{S} prec{prec}", lambda s: run_conv(x[s],None,w1,b,1,prec))
    g = rnd(7,Cin)*0.2+1
    chk(f"1x1 ln {Cin}->{Cout}@{S} prec{prec}", lambda s: run_conv(x[s],None,w1,None,1,16|prec,A=g))
