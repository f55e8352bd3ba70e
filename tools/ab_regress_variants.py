"""Time the x4-upsample + softmax + regression tail (upsample_softmax_regress_kernel) of the bench step in several builds
and hash its outputs:  python tools/ab_regress_variants.py name [name ...]   (gpurun_scratch/lib_<name>.so, or `shipped`)"""
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
from diffuvolume_amd import _lib
dev = "cuda:0"
lib = _lib.load()


def tail(cost, align, want_unc=True):
    b, d, h, w = cost.shape
    disp = torch.empty(b, 4 * h, 4 * w, device=dev)
    unc = torch.empty_like(disp) if want_unc else None
    _lib.check(lib.dv_upsample_softmax_regress_f32(cost.data_ptr(), disp.data_ptr(), _lib.ptr(unc), b, d, h, w, int(align),
                                                   _lib.stream_ptr()), "tail")
    return disp, unc


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def digest(*ts):
    hsh = hashlib.sha256()
    for t in ts:
        if t is not None:
            hsh.update(t.detach().cpu().numpy().tobytes())
    return hsh.hexdigest()[:12]


g = torch.Generator(device="cpu").manual_seed(11)
hashes = []
for (b, h, w, scale, align) in ((2, 64, 120, 3.0, False), (1, 37, 53, 30.0, False), (1, 94, 310, 8.0, True), (2, 20, 33, 0.5, True)):
    cost = (torch.randn(b, 48, h, w, generator=g) * scale).to(dev)
    hashes.append(digest(*tail(cost, align)))
cost = torch.randn(1, 48, 16, 16, generator=g).to(dev)
cost[0, 5, 3, 3] = float("nan"); cost[0, 7, 9, 9] = float("-inf")
hashes.append(digest(*tail(cost, False)))
cost = torch.randn(8, 48, 128, 240, device=dev) * 3
ts = [timeit(lambda: tail(cost, False)) for _ in range(4)]
costk = torch.randn(2, 48, 94, 310, device=dev) * 3
tk = [timeit(lambda: tail(costk, True)) for _ in range(2)]
print(f"{os.environ.get('DV_VARIANT', '?'):12s} tail @ 8x48x128x240: " + " / ".join(f"{t:.4f}" for t in ts) +
      "  align_corners 2x48x94x310: " + " / ".join(f"{t:.4f}" for t in tk) + "  ms   hashes " + " ".join(hashes), flush=True)
