"""Golden vectors for the KITTI12 flavour (PCWNet + DiffuVolume) from the imported reference
(KITTI12/models).  Build container only:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_pcw.py"""
import os
import sys
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, synth_stereo_batch  # noqa: E402

warnings.filterwarnings("ignore")
OUT = REPO / "tests" / "golden"


def save(name, **arrays):
    np.savez_compressed(OUT / f"{name}.npz", **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                for k, v in arrays.items()})
    print(f"  {name}.npz  {(OUT / f'{name}.npz').stat().st_size / 1024:.1f} KB")


def main():
    torch.set_num_threads(8)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device                 # KITTI12 warp() (submodule.py:146)
    sys.path.insert(0, "/root/reference/KITTI12")
    os.chdir("/root/reference/KITTI12")
    from models import __models__ as REF_MODELS
    from models.pwcnet_ddim import hourglass as ref_hourglass, hourglassup as ref_hourglassup
    from diffuvolume_amd.pwcnet_ddim import Hourglass, HourglassUp, PWCNet_ddim

    # ---- layers ----
    hg = ref_hourglass(32).eval()
    hg.load_state_dict(synth_state_dict(Hourglass(32).state_dict(), seed=71), strict=True)
    x = torch.randn(1, 32, 8, 8, 12, generator=_gen(71, "x"))
    hu = ref_hourglassup(32).eval()
    hu.load_state_dict(synth_state_dict(HourglassUp(32).state_dict(), seed=72), strict=True)
    xu = torch.randn(1, 32, 16, 8, 8, generator=_gen(72, "x"))
    f4 = torch.randn(1, 64, 8, 4, 4, generator=_gen(72, "f4"))
    f5 = torch.randn(1, 64, 4, 2, 2, generator=_gen(72, "f5"))
    f6 = torch.randn(1, 64, 2, 1, 1, generator=_gen(72, "f6"))
    with torch.no_grad():
        save("pcw_layers", hg_x=x, hg_y=hg(x), hu_x=xu, hu_f4=f4, hu_f5=f5, hu_f6=f6, hu_y=hu(xu, f4, f5, f6))

    # ---- model ----
    ref = REF_MODELS["pwc_ddimgc"](192).eval()
    sd = synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=8.0,
                            scale={"refinenet3.conv8.weight": 0.002})
    ref.load_state_dict(sd, strict=True)
    b, h, w = 1, 16, 32
    vol = torch.rand(b, 32, 48, h, w, generator=_gen(73, "vol"))
    fl = {"finetune_feature": torch.randn(b, 32, h, w, generator=_gen(73, "fl"))}
    fr = {"finetune_feature": torch.randn(b, 32, h, w, generator=_gen(73, "fr"))}
    x_t = torch.randn(b, 48, h, w, generator=_gen(73, "xt"))
    t = torch.full((b,), 999, dtype=torch.long)
    real_randn, real_randn_like = torch.randn, torch.randn_like
    with torch.no_grad():
        pn, xs, disp, pv = ref.model_predictions(vol, x_t, t, fl, fr)
        kk = torch.arange(0, 192, dtype=disp.dtype).view(1, -1, 1, 1)
        unc = torch.sum(torch.abs(disp.unsqueeze(1) - kk) * pv, dim=1)
        save("pcw_model_predictions", seed=73, x_t=x_t, t=t, pred_noise=pn, x_start=xs, disp=disp, unc=unc)
        used = disp + (torch.rand(disp.shape, generator=_gen(73, "jit")) * 4 - 2)
        asd = torch.rand(b, 48, h, w, generator=_gen(73, "asd")) * 2 - 1
        tape = NoiseTape(79)
        calls = {"n": 0, "first": True}
        steps = []

        def fake_randn(*a, **k):
            if calls["first"]:                                        # img = torch.randn(shape) (:541)
                calls["first"] = False
                return tape("x_T", tuple(a[0]) if isinstance(a[0], (tuple, list, torch.Size)) else tuple(a), torch.float32)
            return real_randn(*a, **k)

        def fake_randn_like(x, *a, **k):
            calls["n"] += 1
            return tape("eps" if calls["n"] % 2 == 1 else "q", tuple(x.shape), x.dtype)

        orig = ref.model_predictions

        def rec(volume, img, tc, a, bb):
            out = orig(volume, img, tc, a, bb)
            steps.append(out[2].clone())
            return out

        ref.model_predictions = rec
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        try:
            final, _ = ref.ddim_sample(vol, used, asd, fl, fr)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        save("pcw_ddim_sample", seed=73, used=used, asd=asd, tape_seed=79, final=final, stack=torch.stack([used] + steps))

        # Untrained residual stacks blow the 2-D features up to 1e6..1e9; calibrate the last 1x1 conv of
        # every feature head to unit output scale so the end-to-end vector sits in a sane range.  The
        # factors are stored with the fixture and re-applied by the tests.
        batch = synth_stereo_batch(1, 64, 128, seed=74, shifts=(8,))
        ref = REF_MODELS["pwc_ddimgc"](192).eval()
        ref.load_state_dict(sd, strict=True)
        feats = ref.feature_extraction(batch["left"])
        heads = {"gw1": "layer11.2.weight", "gw2": "gw2.2.weight", "gw3": "gw3.2.weight", "gw4": "gw4.2.weight",
                 "concat_feature1": "lastconv.2.weight", "concat_feature2": "concat2.2.weight",
                 "concat_feature3": "concat3.2.weight", "concat_feature4": "concat4.2.weight"}
        fscale = {"feature_extraction." + k: 1.0 / float(feats[f].std()) for f, k in heads.items()}
        fscale["feature_extraction.layer_refine.0.0.weight"] = 1.0 / float(feats["finetune_feature"].abs().mean() + 1)
        fscale["refinenet3.conv8.weight"] = 0.002
        sd = synth_state_dict(PWCNet_ddim(192, True).state_dict(), seed=2, logit_gain=8.0, scale=fscale)
        ref.load_state_dict(sd, strict=True)
        tape = NoiseTape(80)
        calls.update(n=0, first=True)
        torch.randn, torch.randn_like = fake_randn, fake_randn_like
        try:
            out, _ = ref(batch["left"], batch["right"], batch["used"], batch["disp"], None)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        print("    forward pred range", float(out[0].min()), float(out[0].max()))
        save("pcw_forward_eval", stereo_seed=74, tape_seed=80, pred=out[0], scale_keys=np.array(list(fscale.keys())),
             scale_vals=np.array(list(fscale.values()), dtype=np.float64))
    print("done")


if __name__ == "__main__":
    main()
