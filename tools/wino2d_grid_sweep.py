"""2-D Winograd kernel vs grid size: time and issued fraction of one 384 -> 128 layer over launches of 0.9 .. 7.5 block rounds
(2 blocks per CU x 256 CUs).  Build the MFMA-only body with tools/build_variant.sh ... -DDV_W2_ABL=63 for the ceiling of the tile shape.
NOTE: the first figure of a process is low (clocks ramp up) -- the shapes of interest are not first.  python tools/wino2d_grid_sweep.py"""
import sys; sys.path.insert(0,'/root/repo')
import torch
from diffuvolume_amd import submodule as S
dev='cuda:0'
def timeit(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
cin,cout=384,128
wt=torch.randn(cout,cin,3,3,device=dev)*0.02
p=S.Conv2dPlan(wt,None,act=S.ACT_RELU)
for b,h,w in ((4,64,256),(4,80,256),(4,96,256),(4,112,256),(4,96,320),(4,128,256),(4,144,256),(4,160,256),(4,96,304),(4,96,312),(4,192,256),(4,240,256),(1,96,312),(2,96,312),(3,96,312),(5,96,312),(6,96,312),(8,96,312)):
    x=torch.randn(b,cin,h,w,device=dev)
    t=timeit(lambda:p(x)); fl=2.0*b*cout*h*w*cin*9
    blocks=b*(-(-h//16))*(-(-w//16))*(cout//32)
    print(f"B{b} {h}x{w}: {t:.3f} ms  issued {fl/t/1e9/2.25/157.3:.3f}  blocks {blocks} rounds {blocks/512:.2f}  per-block {t*1e3/max(1,blocks/512):.1f} us/round")
