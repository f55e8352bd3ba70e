"""Time the 3-D Winograd layers of the bench step (with and without a residual tensor) in several builds of the library
(gpurun_scratch/lib_<name>.so, tools/build_variant.sh): python tools/ab_wino_variants.py name [name ...]"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for name in sys.argv[1:]:
        env = dict(os.environ, DV_LIB_PATH=str(ROOT / "gpurun_scratch" / f"lib_{name}.so"), DV_VARIANT=name)
        subprocess.run([sys.executable, __file__, "--child"], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, str(ROOT))
import torch
from diffuvolume_amd import submodule as S
dev = "cuda:0"


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for name, cin, cout, dims in (("c32", 32, 32, (48, 128, 240)), ("c64", 64, 64, (24, 64, 120)), ("c128", 128, 128, (12, 32, 60))):
    x = torch.randn(8, cin, *dims, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    plan = S.Conv3dPlan(w, bn, act=S.ACT_RELU, precision="f32")
    r = torch.randn(8, cout, *dims, device=dev)
    a = min(timeit(lambda: plan(x)) for _ in range(2))
    b = min(timeit(lambda: plan(x, residual=r)) for _ in range(2))
    out.append(f"{name} {a:.3f} res {b:.3f}")
    del x, r, plan
print(f"{os.environ.get('DV_VARIANT', '?'):12s} " + "   ".join(out), flush=True)
