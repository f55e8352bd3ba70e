"""KITTI15 at the reference's OWN default: `sampling_timesteps = 2` is hard-coded in igev_stereo_ddim.py:124 (BASELINE config 5
names 20 steps), 32 GRU iterations (evaluate_stereo.py), the origin IGEVStereo forward first (evaluate_stereo.py:88-98).
Batch 4 at 1248x384, stub backbone.     python tools/igev_default_config.py"""
import sys, types, time, torch
import torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from bench_flavours import DEV, _gen, synth_state_dict, timeit
from diffuvolume_amd.igev_stereo_ddim import Feature, IGEVStereo_ddim
from diffuvolume_amd.igev_stereo import IGEVStereo
from diffuvolume_amd.synth import StubMobileNetV2
b, h, w = 4, 384, 1248
args = types.SimpleNamespace(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
                             slow_fast_gru=False, max_disp=192, mixed_precision=False)
m = IGEVStereo_ddim(args, feature=Feature(StubMobileNetV2()))
m.load_state_dict(synth_state_dict(m.state_dict(), seed=7, scale={"update_block.disp_head.conv2.weight": 0.05, "update_block.disp_head.conv2.bias": 0.0, "classifier.weight": 20.0}), strict=True)
m = m.to(DEV).eval()
o = IGEVStereo(args, feature=Feature(StubMobileNetV2()))
o.load_state_dict(synth_state_dict(o.state_dict(), seed=8, scale={"update_block.disp_head.conv2.weight": 0.05, "update_block.disp_head.conv2.bias": 0.0, "classifier.weight": 20.0}), strict=True)
o = o.to(DEV).eval()
g = _gen(77, "cfg5")
img1 = (torch.rand(b, 3, h, w, generator=g) * 255).to(DEV)
img2 = torch.roll(img1, -9, dims=-1)
def both():
    with torch.no_grad():
        flow_pr = o(img1, img2, iters=32, test_mode=True)
        flow_4 = F.interpolate(torch.clamp(flow_pr, 0, w - 1), size=(h // 4, w // 4), mode="bilinear") / 4
        return m(img1, img2, flow_pr, flow_4, iters=32, test_mode=True)
with torch.no_grad():
    flow_pr = o(img1, img2, iters=32, test_mode=True)
    flow_4 = F.interpolate(torch.clamp(flow_pr, 0, w - 1), size=(h // 4, w // 4), mode="bilinear") / 4
    t_o = timeit(lambda: o(img1, img2, iters=32, test_mode=True), warmup=1, steps=3)
    t_m = timeit(lambda: m(img1, img2, flow_pr, flow_4, iters=32, test_mode=True), warmup=1, steps=3)
    t_b = timeit(both, warmup=1, steps=3)
print(f"B=4 1248x384, 32 iterations: origin IGEVStereo {t_o:.1f} ms, IGEVStereo_ddim at the reference's default 2 steps {t_m:.1f} ms, "
      f"validate_kitti's pair of forwards {t_b:.1f} ms = {t_b / b:.1f} ms per pair = {b / t_b * 1e3:.2f} pairs/s")
