"""Time the 3x3x3 stride-1 layers of the bench step on the F(2x2x2,3x3x3) kernel in several builds
(gpurun_scratch/lib_<name>.so; timing-only ablations give wrong results):  python tools/ab_wino3_variants.py name [name ...]"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for name in sys.argv[1:]:
        env = dict(os.environ, DV_VARIANT=name)
        if name != "shipped":
            env["DV_LIB_PATH"] = str(ROOT / "gpurun_scratch" / f"lib_{name}.so")
        subprocess.run([sys.executable, __file__, "--child"], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, str(ROOT))
import torch
from diffuvolume_amd import submodule as S
dev = "cuda:0"
S.Conv3dPlan.WINO3_MIN_CIN = 1      # every layer on the kernel under test


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for name, c, dims in (("c32", 32, (48, 128, 240)), ("c64", 64, (24, 64, 120)), ("c128", 128, (12, 32, 60))):
    x = torch.randn(8, c, *dims, device=dev)
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(c, device=dev) + 0.5 for _ in range(4))
    plan = S.Conv3dPlan(w, bn, act=S.ACT_RELU, precision="f32")
    ts = [timeit(lambda: plan(x)) for _ in range(3)]
    out.append(f"{name} " + " / ".join(f"{t:.3f}" for t in ts))
    del x, plan
print(f"{os.environ.get('DV_VARIANT', '?'):14s} " + "   ".join(out), flush=True)
