"""Time the PyTorch-side pieces of ACVNet_DDIM.attention_concat_volume at the bench size."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import diffuvolume_amd as dv
from diffuvolume_amd.synth import synth_state_dict
dev = "cuda:0"
m = dv.ACVNet_DDIM(192, False, False)
m.load_state_dict(synth_state_dict(m.state_dict(), seed=0, logit_gain=8.0))
m = m.to(dev).eval()
B = 8
fl, fr = torch.randn(B, 320, 128, 240, device=dev), torch.randn(B, 320, 128, 240, device=dev)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    gwc = dv.build_gwc_volume(fl, fr, 48, 40)
    print("patch            %.2f ms" % t(lambda: m.patch(gwc)))
    g2 = m.patch(gwc)
    print("patch_l1..3+cat  %.2f ms" % t(lambda: torch.cat((m.patch_l1(g2[:, :8]), m.patch_l2(g2[:, 8:24]), m.patch_l3(g2[:, 24:40])), dim=1)))
    print("concatconv x2    %.2f ms" % t(lambda: (m.concatconv(fl), m.concatconv(fr))))
    print("whole attention_concat_volume %.2f ms" % t(lambda: m.attention_concat_volume(fl, fr)))
