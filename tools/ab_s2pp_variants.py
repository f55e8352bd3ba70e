"""Time the polyphase stride-2 layers of the bench step in several builds (gpurun_scratch/lib_<name>.so):
python tools/ab_s2pp_variants.py name [name ...]"""
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
for name, cin, cout, dims in (("s2_64", 32, 64, (48, 128, 240)), ("s2_128", 64, 128, (24, 64, 120))):
    x = torch.randn(8, cin, *dims, device=dev)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    plan = S.Conv3dPlan(w, bn, stride=2, act=S.ACT_RELU, precision="f32")
    assert plan.s2pp
    a = [timeit(lambda: plan(x)) for _ in range(3)]
    out.append(f"{name} " + " / ".join(f"{t:.3f}" for t in a))
    del x, plan
print(f"{os.environ.get('DV_VARIANT', '?'):12s} " + "   ".join(out), flush=True)
