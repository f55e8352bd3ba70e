"""Time the persistent transposed convolution in several builds (gpurun_scratch/lib_pl_*.so, tools/build_variant.sh) on the
bench shapes; a build with -DDVPL_STAMP also dumps the per-step stamps of block 0.
python tools/ab_deconv_variants.py name [name ...]      (run as a child process per variant: DV_LIB_PATH is read at import)"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] != "--child"):
    for name in sys.argv[1:]:
        env = dict(os.environ, DV_LIB_PATH=str(ROOT / "gpurun_scratch" / f"lib_pl_{name}.so"), DV_VARIANT=name)
        subprocess.run([sys.executable, __file__, "--child"], env=env, check=False)
    sys.exit(0)

sys.path.insert(0, str(ROOT))
import ctypes
import torch
from diffuvolume_amd import submodule as S, _lib
dev = "cuda:0"
lib = _lib.load()
name = os.environ.get("DV_VARIANT", "?")


def timeit(run, n=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B = 8
res = []
for lname, cin, cout, dims in (("dc64x", 64, 32, (24, 64, 120)), ("dc128x", 128, 64, (12, 32, 60))):
    x = torch.randn(B, cin, *dims, device=dev)
    w = torch.randn(cin, cout, 3, 3, 3, device=dev) * 0.05
    bn = tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))
    rw = torch.randn(cout, cout, 1, 1, 1, device=dev) * 0.1
    plan = S.Deconv3dPlan(w, bn, act=S.ACT_RELU, redir=(rw, tuple(torch.rand(cout, device=dev) + 0.5 for _ in range(4))))
    t = torch.randn(B, cout, *(2 * d for d in dims), device=dev)
    lib.dv_deconv3d_set_impl(2)
    ms = [timeit(lambda: plan(x, skip=t)) for _ in range(2)]
    res.append(f"{lname} {ms[0]:.3f} / {ms[1]:.3f}")
    if name.startswith("stamp") and lname == "dc64x":
        raw = ctypes.CDLL(os.environ["DV_LIB_PATH"])
        buf = (ctypes.c_ulonglong * (3 * 2 * 256))()
        plan(x, skip=t); torch.cuda.synchronize()
        raw.dv_deconv3d_pl_read_stamps.argtypes = [ctypes.c_void_p]
        rc = raw.dv_deconv3d_pl_read_stamps(buf)
        v = list(buf)
        t0 = v[0]
        print("stamps rc", rc, "(cycles since step 0 start of wave 0; per step: group0 start/end | group1 start/end | loader0 start/end(after vmcnt0))")
        for s in range(12, 34):
            row = []
            for slot in range(3):
                a, b = v[(slot * 256 + s) * 2], v[(slot * 256 + s) * 2 + 1]
                row.append(f"{a - t0:8d} {b - a:6d}")
            print(f"step {s:3d}: " + " | ".join(row))
    del x, plan, t
print(f"{name:10s} " + "   ".join(res), flush=True)
