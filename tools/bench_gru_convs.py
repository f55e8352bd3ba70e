"""Time the ConvGRU launches of IGEV's update block at the config-5 sizes (batch 4, 1248x384): the fused z / r gate launch and
the candidate launch, with the sources of the virtual concatenation split as the update block hands them over.
python tools/bench_gru_convs.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from diffuvolume_amd import submodule as S

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


_w = torch.randn(4096, 4096, device=dev)
for _ in range(50):                       # clocks up before the first figure
    _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
for name, b, split, h, w in (("gru04 [128|127|1|128]", 4, (128, 127, 1, 128), 96, 312), ("gru04 [128|128|128]", 4, (128, 128, 128), 96, 312), ("gru04 [128|127|1|128]", 4, (128, 127, 1, 128), 96, 312),
                             ("gru08 [128|128|128]", 4, (128, 128, 128), 48, 156), ("gru16 [128|128]", 4, (128, 128), 24, 78)):
    cin = sum(split)
    xs = [torch.randn(b, c, h, w, device=dev) for c in split]
    wz, wr, wq = (torch.randn(128, cin, 3, 3, device=dev) * 0.02 for _ in range(3))
    bz, br, bq = (torch.zeros(128, device=dev) for _ in range(3))
    cz, cr, cq = (torch.randn(b, 128, h, w, device=dev) for _ in range(3))
    pzr = S.Conv2dPairPlan((wz, bz), (wr, br), S.ACT_SIGMOID)
    pq = S.Conv2dPlan(wq, None, dilation=1, act=S.ACT_TANH, bias=bq)
    hh = xs[0]
    z, rh = pzr(xs, residual=(cz, cr), mul=(None, hh))
    tzr = timeit(lambda: pzr(xs, residual=(cz, cr), mul=(None, hh)))
    tq = timeit(lambda: pq([rh, *xs[1:]], residual=cq, blend=(z, hh)))
    fl = 2.0 * b * 128 * h * w * cin * 9
    print(f"{name:24s} z|r {tzr:6.3f} ms ({2 * fl / tzr / 1e9 / 2.25 / 157.3:5.3f} issued)   q {tq:6.3f} ms ({fl / tq / 1e9 / 2.25 / 157.3:5.3f} issued)", flush=True)
