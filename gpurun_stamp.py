import sys, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_kernels import run_conv, rnd
from hicdiff_amd import _lib as L
lib = L.load()
lib.hd_debug_stamp.argtypes = [C.c_int, C.c_ulonglong * 8]
def measure(name, fn):
    fn()
    out = (C.c_ulonglong * 8)()
    lib.hd_debug_stamp(1, out)
    fn()
    lib.hd_debug_stamp(2, out)
    w, b, m, s, ep, tot, waves, nit = [float(v) for v in out]
    per = lambda v: v / waves
    print(f"{name:40s} waves={int(waves)} iters/wave={nit/waves:.0f} | per wave cycles: total={per(tot):9.0f} wstage={per(w):8.0f} barrier={per(b):8.0f} read+mfma={per(m):8.0f} slice_stage={per(s):8.0f} epilogue={per(ep):8.0f} other={per(tot-w-b-m-s-ep):8.0f} | per-iter: wstage={w/nit:6.0f} barrier={b/nit:6.0f} mfma={m/nit:6.0f}", flush=True)
prec = 96
for (B,S,Cin,Cout) in [(64,64,256,256),(64,64,64,64),(64,32,128,128),(64,8,512,512)]:
    x,w,b=rnd(1,B,Cin,S,S),rnd(2,Cout,Cin,3,3)/(3*Cin**0.5),rnd(3,Cout)
    measure(f"3x3 B{B} S{S} {Cin}->{Cout}", lambda: run_conv(x,None,w,b,3,0|prec))
