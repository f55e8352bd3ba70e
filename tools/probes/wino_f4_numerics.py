"""fp32 error of the F(4x4, 3x3) Winograd form against direct and F(2x2, 3x3) on one 32 -> 32 plane (CPU, torch emulation with
the same order of operations as the kernels: weights transformed in fp64 and rounded once, data and output transforms in
fp32).  DESIGN 4b quotes its output.     python tools/probes/wino_f4_numerics.py"""
import torch, numpy as np
torch.manual_seed(0)
def mats(m):
    if m==2:
        BT=[[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
        G=[[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        AT=[[1,1,1,0],[0,1,-1,-1]]
    else:
        BT=[[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
        G=[[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
        AT=[[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
    return [torch.tensor(x,dtype=torch.float64) for x in (BT,G,AT)]
def wino2d(x,w,m,dt):
    # x [C,H,W] (padded so H-2, W-2 multiple of m), w [K,C,3,3]; returns [K,H-2,W-2]
    BT,G,AT=[t.to(dt) for t in mats(m)]
    a=m+2
    C,H,W=x.shape; K=w.shape[0]
    U=torch.einsum('ij,kcjl,ml->kcim',G.double(),w.double(),G.double()).to(dt)   # weights transformed in f64 then rounded (host pack)
    th,tw=(H-2)//m,(W-2)//m
    p=x.unfold(1,a,m).unfold(2,a,m)            # [C,th,tw,a,a]
    V=torch.einsum('ij,ctwjl,ml->ctwim',BT,p,BT)
    M=torch.einsum('kcim,ctwim->ktwim',U,V)
    Y=torch.einsum('ij,ktwjl,ml->ktwim',AT,M,AT)   # [K,th,tw,m,m]
    return Y.permute(0,1,3,2,4).reshape(K,th*m,tw*m)
C=32;K=32;H=48;W=96
x=torch.randn(C,H+2,W+2)
# activations after relu typical: make non-negative mean
x=torch.relu(x)
w=torch.randn(K,C,3,3)*(2/(9*C))**.5
ref=torch.nn.functional.conv2d(x.double()[None],w.double())[0]
d32=torch.nn.functional.conv2d(x[None],w)[0]
s=ref.abs().max()
for name,y in (('direct f32',d32),('F(2x2) f32',wino2d(x,w,2,torch.float32)),('F(4x4) f32',wino2d(x,w,4,torch.float32)),('F(4x4) f64',wino2d(x.double(),w.double(),4,torch.float64))):
    e=(y.double()-ref).abs()
    print(f'{name:12s} max {e.max()/s:.2e} rms {e.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt():.2e}')
