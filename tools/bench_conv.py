"""Micro-benchmark of single aggregation layers at the bench.py sizes (B pairs, 960x512 -> 48x128x240).
usage: python tools/bench_conv.py [layer ...]   layers: c32 c64in c1 s2_64 c64 s2_128 c128 dc128 dc64 k1_32 k1_64 attn"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diffuvolume_amd import submodule as S

B = 8
LAYERS = {  # name: (kind, cin, cout, k, stride, (D,H,W))
    "c32": ("conv", 32, 32, 3, 1, (48, 128, 240)),
    "c64in": ("conv", 64, 32, 3, 1, (48, 128, 240)),
    "c1": ("conv", 32, 1, 3, 1, (48, 128, 240)),
    "s2_64": ("conv", 32, 64, 3, 2, (48, 128, 240)),
    "c64": ("conv", 64, 64, 3, 1, (24, 64, 120)),
    "s2_128": ("conv", 64, 128, 3, 2, (24, 64, 120)),
    "c128": ("conv", 128, 128, 3, 1, (12, 32, 60)),
    "dc128": ("deconv", 128, 64, 3, 2, (12, 32, 60)),
    "dc64": ("deconv", 64, 32, 3, 2, (24, 64, 120)),
    "dc128r": ("deconv_res", 128, 64, 3, 2, (12, 32, 60)),
    "dc64r": ("deconv_res", 64, 32, 3, 2, (24, 64, 120)),
    "dc128x": ("deconv_redir", 128, 64, 3, 2, (12, 32, 60)),
    "dc64x": ("deconv_redir", 64, 32, 3, 2, (24, 64, 120)),
    "k1_32": ("conv", 32, 32, 1, 1, (48, 128, 240)),
    "k1_64": ("conv", 64, 64, 1, 1, (24, 64, 120)),
}
names = sys.argv[1:] or list(LAYERS)
iters = 20
dev = "cuda:0"
for n in names:
    kind, cin, cout, k, s, dims = LAYERS[n]
    x = torch.randn(B, cin, *dims, device=dev)
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    if kind == "conv":
        w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
        plan = S.Conv3dPlan(w, bn, stride=s, act=S.ACT_RELU)
        flops = 2.0 * B * cout * cin * k ** 3 * (dims[0] // s) * (dims[1] // s) * (dims[2] // s)
    elif kind == "deconv_redir":
        w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
        rw = torch.randn(cout, cout, 1, 1, 1, device=dev) * 0.1
        plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU, redir=(rw, tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))))
        flops = 2.0 * B * cout * cin * 27 * dims[0] * dims[1] * dims[2] + 2.0 * B * cout * cout * 8 * dims[0] * dims[1] * dims[2]
    else:
        w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
        plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU)
        flops = 2.0 * B * cout * cin * 27 * dims[0] * dims[1] * dims[2]
    if kind == "deconv_res":
        res = torch.randn(B, cout, *(2 * d for d in dims), device=dev)
        run = lambda: plan(x, residual=res)
    elif kind == "deconv_redir":
        res = torch.randn(B, cout, *(2 * d for d in dims), device=dev)
        run = lambda: plan(x, skip=res)
    else:
        run = lambda: plan(x)
    y = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{n:7s} {ms:8.3f} ms  {flops / ms / 1e9:7.2f} TFLOP/s", flush=True)
    del x, y, plan
