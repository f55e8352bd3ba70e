"""Config 5: how much of one `IGEVStereo_ddim.forward` is the once-per-pair front (feature pyramid, stems, context encoder,
cost volume), and which kernels it runs.     python tools/igev_front_time.py [--steps 20] [--iters 32]"""
import argparse
import json
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_flavours import DEV, _gen, synth_state_dict, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    from diffuvolume_amd.igev_stereo_ddim import Feature, IGEVStereo_ddim
    from diffuvolume_amd.synth import StubMobileNetV2
    b, h, w = a.batch, 384, 1248
    args = types.SimpleNamespace(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
                                 slow_fast_gru=False, max_disp=192, mixed_precision=False)
    cof = [0.5] + [0.0] * (a.steps - 1) + [0.5] if a.steps != 2 else None
    m = IGEVStereo_ddim(args, feature=Feature(StubMobileNetV2()), sampling_timesteps=a.steps, ensemble_cof=cof)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=7, scale={"update_block.disp_head.conv2.weight": 0.05,
                                                                      "update_block.disp_head.conv2.bias": 0.0,
                                                                      "classifier.weight": 20.0}), strict=True)
    m = m.to(DEV).eval()
    g = _gen(77, "cfg5")
    img1 = (torch.rand(b, 3, h, w, generator=g) * 255).to(DEV)
    img2 = torch.roll(img1, -9, dims=-1)
    flow_full = (9 + torch.randn(b, 1, h, w, generator=g)).clamp(0.5, 47).to(DEV)
    flow_gt = F.interpolate(flow_full, size=(h // 4, w // 4), mode="bilinear") / 4
    out = {}
    with torch.no_grad():
        out["front_ms"] = timeit(lambda: m._front(img1, img2), warmup=2, steps=3)
        out["forward_ms"] = timeit(lambda: m(img1, img2, flow_full, flow_gt, iters=a.iters, test_mode=True), warmup=1, steps=1)
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            m._front(img1, img2)
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:25]
        out["front_kernels"] = [{"name": e.key[:90], "calls": e.count, "ms": round(e.device_time_total / 1e3, 3)} for e in rows
                                if e.device_time_total > 0]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
