"""Time the 32 -> 1 classification head (conv3d_c1z_kernel) of the bench step in several builds and hash its outputs:
python tools/ab_c1z_variants.py name [name ...]      (name = gpurun_scratch/lib_<name>.so, or `shipped`)"""
import hashlib, os, subprocess, sys
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


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]


g = torch.Generator(device="cpu").manual_seed(7)
hashes = []
for (b, c, d, h, w) in ((2, 32, 48, 128, 240), (1, 32, 13, 37, 70), (3, 7, 5, 16, 64), (1, 64, 24, 94, 310), (1, 128, 12, 40, 100)):
    x = torch.randn(b, c, d, h, w, generator=g).to(dev)
    wt = (torch.randn(1, c, 3, 3, 3, generator=g) * 0.05).to(dev)
    plan = S.Conv3dPlan(wt, None, act=S.ACT_NONE, precision="f32")
    hashes.append(digest(plan(x)))
x = torch.randn(8, 32, 48, 128, 240, device=dev)
wt = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.05
plan = S.Conv3dPlan(wt, None, act=S.ACT_NONE, precision="f32")
ts = [timeit(lambda: plan(x)) for _ in range(4)]
print(f"{os.environ.get('DV_VARIANT', '?'):12s} head 32->1 @ 8x48x128x240: " + " / ".join(f"{t:.4f}" for t in ts) + "  ms   hashes " + " ".join(hashes), flush=True)
