"""Golden vectors for the IGEV DDIM loop, produced by the REFERENCE methods
IGEVStereo_ddim.model_predictions / ddim_sample (KITTI15/core/igev_stereo_ddim.py:226-359) bound to a light
object: the class itself cannot be constructed here (timm pretrained backbone), but its two methods only need
the time head, the schedule buffers and the update / upsample callables.  Build container only."""
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, toy_update_block, toy_upsample_disp  # noqa: E402

warnings.filterwarnings("ignore")
torch.Tensor.cuda = lambda self, *a, **k: self
sys.modules.setdefault("timm", types.ModuleType("timm"))
oe = types.ModuleType("opt_einsum")
oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", oe)
sys.path.insert(0, "/root/reference/KITTI15")
import core.igev_stereo_ddim as R  # noqa: E402
from core.geometry_ddim import Combined_Geo_Encoding_Volume  # noqa: E402
from core.head import DynamicHead  # noqa: E402
from diffuvolume_amd.igev_stereo_ddim import DynamicHead180  # noqa: E402


class Light:
    """`self` for the two reference methods."""
    model_predictions = R.IGEVStereo_ddim.model_predictions
    ddim_sample = R.IGEVStereo_ddim.ddim_sample
    q_sample = R.IGEVStereo_ddim.q_sample
    predict_noise_from_start = R.IGEVStereo_ddim.predict_noise_from_start


def main():
    obj = Light()
    obj.args = types.SimpleNamespace(n_gru_layers=3, slow_fast_gru=False, mixed_precision=False)
    obj.scale, obj.num_timesteps, obj.sampling_timesteps, obj.ddim_sampling_eta = 1.0, 1000, 2, 1
    obj.renewal, obj.use_ensemble = True, True
    betas = R.cosine_beta_schedule(1000)
    ac = torch.cumprod(1.0 - betas, dim=0)
    obj.alphas_cumprod, obj.sqrt_alphas_cumprod = ac, torch.sqrt(ac)
    obj.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - ac)
    obj.sqrt_recip_alphas_cumprod, obj.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)
    head = DynamicHead(d_model=180).eval()
    head.load_state_dict(synth_state_dict(DynamicHead180().state_dict(), seed=81), strict=True)
    obj.time_embedding = head
    obj.update_block = toy_update_block
    obj.upsample_disp = toy_upsample_disp

    b, c, d, h, w = 1, 8, 48, 8, 24
    geo = torch.randn(b, c, d, h, w, generator=_gen(82, "geo"))
    f1, f2 = torch.randn(b, 16, h, w, generator=_gen(82, "f1")), torch.randn(b, 16, h, w, generator=_gen(82, "f2"))
    init = torch.rand(b, 1, h, w, generator=_gen(82, "init")) * 40
    used = F.interpolate(init * 4, scale_factor=4, mode="bilinear") + torch.randn(b, 1, 4 * h, 4 * w, generator=_gen(82, "u")) * 3
    asd = torch.rand(b, 48, h, w, generator=_gen(82, "asd")) * 2 - 1
    geo_fn = Combined_Geo_Encoding_Volume(f1, f2, geo, radius=4, num_levels=2)
    tsh = torch.tensor([999, 499, 3])
    with torch.no_grad():
        shifts = torch.stack([head(torch.zeros(1, 48, 1, 1), tsh[i:i + 1]).reshape(48) for i in range(3)])
        x_t = torch.randn(b, 48, h, w, generator=_gen(82, "xt"))
        t = torch.full((b,), 999, dtype=torch.long)
        pn, xs, pred, c1 = obj.model_predictions(init, init, None, 3, [None], [None], geo_fn, x_t, t, None)
        tape = NoiseTape(83)
        calls = {"n": 0}
        real = torch.randn_like

        def fake_randn_like(x, *a, **k):
            calls["n"] += 1
            if calls["n"] == 1:
                return tape("x_T", tuple(x.shape), x.dtype)          # img = randn_like(asd) :303
            return tape("eps" if calls["n"] % 2 == 0 else "q", tuple(x.shape), x.dtype)

        torch.randn_like = fake_randn_like
        try:
            final = obj.ddim_sample(init, init, None, 3, [None], [None], geo_fn, used, asd, None)
        finally:
            torch.randn_like = real
    np.savez_compressed(REPO / "tests/golden/igev_loop.npz", seed=82, head_seed=81, shift_t=tsh.numpy(), shifts=shifts.numpy(),
                        x_t=x_t.numpy(), pred_noise=pn.numpy(), x_start=xs.numpy(), pred=pred.numpy(), coords1=c1.numpy(),
                        used=used.numpy(), asd=asd.numpy(), tape_seed=83, final=final.numpy())
    print("igev_loop.npz", tuple(final.shape), float(final.min()), float(final.max()))


if __name__ == "__main__":
    main()
