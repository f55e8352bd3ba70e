"""Time the persistent 3-D Winograd kernel in several builds (gpurun_scratch/lib_wp_*.so): python tools/ab_wino_pl_variants.py name ..."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for name in sys.argv[1:]:
        env = dict(os.environ, DV_LIB_PATH=str(ROOT / "gpurun_scratch" / f"lib_wp_{name}.so"), DV_VARIANT=name)
        subprocess.run([sys.executable, __file__, "--child"], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, str(ROOT))
import torch
from diffuvolume_amd import submodule as S, _lib
dev = "cuda:0"
lib = _lib.load()


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
    lib.dv_conv3d_wino_set_impl(2)
    a = min(timeit(lambda: plan(x)) for _ in range(2))
    lib.dv_conv3d_wino_set_impl(1)
    b = min(timeit(lambda: plan(x)) for _ in range(2))
    out.append(f"{name} {a:.3f} (one-tile {b:.3f})")
    del x, plan
print(f"{os.environ.get('DV_VARIANT', '?'):10s} " + "   ".join(out), flush=True)
