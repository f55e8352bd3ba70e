"""Generate tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference, which exists only in the build container) on seeded inputs.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Fixtures are data only: inputs (or the seed that regenerates them through
diffuvolume_amd.synth) and the reference's outputs.  Weights are never stored: they are
regenerated from (seed, key, shape) by ``synth_state_dict`` on both sides, and this script
checks with ``load_state_dict(strict=True)`` that the build's key set equals the
reference's.  Noise draws are injected by patching torch.randn_like / torch.rand_like while
the reference's ``ddim_sample`` runs; the same ``NoiseTape`` feeds the
oracle and the HIP path in the tests.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import NoiseTape, _gen, synth_state_dict, synth_stereo_batch  # noqa: E402

REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"
warnings.filterwarnings("ignore")


def _load_file(path: Path, name: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v))
                                                for k, v in arrays.items()})
    size = (OUT / f"{name}.npz").stat().st_size
    print(f"  {name}.npz  {size / 1024:.1f} KB")


def rnd(seed, key, *shape):
    return torch.randn(*shape, generator=_gen(seed, key))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    torch.Tensor.cuda = lambda self, *a, **k: self            # hard-coded .cuda() in the reference
    sf = _load_file(REF / "SceneFlow/models/submodule.py", "ref_sf_submodule")
    k12 = _load_file(REF / "KITTI12/models/submodule.py", "ref_k12_submodule")
    k15 = _load_file(REF / "KITTI15/core/submodule.py", "ref_k15_submodule")

    # ---- 1. builders ---------------------------------------------------------------------
    print("builders")
    for tag, (b, c, h, w, d, g) in {"small": (2, 16, 3, 12, 4, 4), "cpg8": (1, 80, 4, 32, 12, 10),
                                    "cpg12": (2, 24, 3, 20, 6, 2), "ragged": (1, 8, 2, 7, 9, 2)}.items():
        L, R = rnd(11, f"L{tag}", b, c, h, w), rnd(11, f"R{tag}", b, c, h, w)
        v = sf.build_gwc_volume(L, R, d, g)
        assert torch.equal(v, k12.build_gwc_volume(L, R, d, g)) and torch.equal(v, k15.build_gwc_volume(L, R, d, g))
        cc = sf.build_concat_volume(L, R, d)
        assert torch.equal(cc, k15.build_concat_volume(L, R, d))
        save(f"builders_{tag}", L=L, R=R, maxdisp=d, groups=g, gwc=v, concat=cc,
             concat_k12=k12.build_concat_volume(L, R, d))
    # attention-weighted concat volume (acv_ddim.py:390)
    L, R = rnd(12, "Lc", 2, 8, 4, 16), rnd(12, "Rc", 2, 8, 4, 16)
    att = rnd(12, "att", 2, 1, 8, 4, 16) * 3
    save("concat_attention", L=L, R=R, att=att, maxdisp=8,
         out=F.softmax(att, dim=2) * sf.build_concat_volume(L, R, 8))

    # ---- 2. regression tail -----------------------------------------------------------------
    print("regression")
    prob = F.softmax(rnd(13, "p", 2, 12, 5, 7), dim=1)
    save("disparity_regression", prob=prob, flat=sf.disparity_regression(prob, 12),
         keepdim=k15.disparity_regression(prob, 12))
    for tag, (d, h, w) in {"d12": (12, 8, 16), "d48": (48, 6, 8)}.items():
        cost = rnd(13, f"cost{tag}", 2, 1, d, h, w) * 4
        res = {}
        for ac in (False, True):
            up = F.interpolate(cost, [4 * d, 4 * h, 4 * w], mode="trilinear", align_corners=ac) if ac \
                else F.upsample(cost, [4 * d, 4 * h, 4 * w], mode="trilinear")
            pv = F.softmax(up.squeeze(1), dim=1)
            disp = sf.disparity_regression(pv, 4 * d)
            kk = torch.arange(0, 4 * d, dtype=disp.dtype).view(1, -1, 1, 1)      # acv_ddim.py:325-329
            unc = torch.sum(torch.abs(disp.unsqueeze(1) - kk) * pv, dim=1)
            sfx = "_ac" if ac else ""
            res["disp" + sfx], res["unc" + sfx] = disp, unc
        save(f"regress_{tag}", cost=cost, **res)

    # ---- reference model ----------------------------------------------------------------------
    sys.path.insert(0, str(REF / "SceneFlow"))
    os.chdir(REF / "SceneFlow")
    from models import __models__ as REF_MODELS
    from models.acv_ddim import hourglass as ref_hourglass
    from models.submodule import attention_block as ref_attention, convbn_3d as ref_convbn_3d
    from diffuvolume_amd.acv_ddim import ACVNet_DDIM as OursModel

    ref = REF_MODELS["acvnet_ddim"](192, False, False).eval()
    ours_keys = OursModel(192, False, False).state_dict()
    sd = synth_state_dict(ours_keys, seed=1, logit_gain=8.0)
    ref.load_state_dict(sd, strict=True)                        # key set / shapes identical

    # ---- 3/4. two-hot encoder, schedule ----------------------------------------------------------
    print("encoder / schedule")
    enc = {}

    def grab_xT(volume, used, x_T):
        enc["x_T"] = x_T.clone()
        raise StopIteration

    def reference_x_T(model, disp_q):
        """x_T exactly as ACVNet_DDIM.forward builds it (acv_ddim.py:403-419): run the reference
        forward on a 64x128 pair and stop at the ddim_sample call."""
        keep = model.ddim_sample
        model.ddim_sample = grab_xT
        try:
            with torch.no_grad():
                model(torch.zeros(1, 3, 64, 128), torch.zeros(1, 3, 64, 128), torch.zeros(1, 64, 128), disp_q, None)
        except StopIteration:
            pass
        model.ddim_sample = keep
        return enc["x_T"]

    dq = torch.rand(1, 1, 16, 32, generator=_gen(15, "dq")) * 47.75
    dq.view(-1)[:8] = torch.tensor([0.0, 3.25, 46.9, 47.0, 47.75, 12.5, 46.0, 0.999])
    x_T_enc = reference_x_T(ref, dq)
    times = {}
    for s in (2, 3, 5, 20):
        tt = torch.linspace(-1, 999, steps=s + 1)
        times[f"times_{s}"] = np.array(list(reversed(tt.int().tolist())))
    save("encoder_schedule", disp_q=dq, x_T=x_T_enc, alphas_cumprod=ref.alphas_cumprod,
         sqrt_recip=ref.sqrt_recip_alphas_cumprod, sqrt_recipm1=ref.sqrt_recipm1_alphas_cumprod, **times)
    shift_t = torch.tensor([999, 799, 599, 399, 199, 0, 17])
    save("time_shift", t=shift_t, shift=ref.time_embedding.block_time_mlp(ref.time_embedding.time_mlp(shift_t)),
         noisy=rnd(14, "noisy", 7, 48, 2, 3),
         out=ref.time_embedding(rnd(14, "noisy", 7, 48, 2, 3), shift_t))

    # ---- 5. layers --------------------------------------------------------------------------------
    print("layers")
    from diffuvolume_amd.acv_ddim import Hourglass as OursHourglass, _WindowAttention, _cb3
    for tag, (cin, cout, k, s, dims) in {"c3s1": (8, 16, 3, 1, (1, 6, 9, 20)), "c3s2": (16, 32, 3, 2, (2, 8, 10, 12)),
                                         "c1s1": (16, 16, 1, 1, (1, 4, 5, 8)), "c3s1_wide": (12, 40, 3, 1, (1, 5, 6, 48)),
                                         "c3s1_one": (8, 1, 3, 1, (1, 4, 8, 16))}.items():
        layer = ref_convbn_3d(cin, cout, k, s, (k - 1) // 2).eval()
        layer.load_state_dict(synth_state_dict(_cb3(cin, cout, k, s, (k - 1) // 2).state_dict(), seed=21))
        x = rnd(21, f"x{tag}", dims[0], cin, *dims[1:])
        with torch.no_grad():
            y = layer(x)
        save(f"layer_{tag}", x=x, y=y, y_relu=F.relu(y), cin=cin, cout=cout, k=k, stride=s, seed=21)
    # transposed conv + BN
    dc = torch.nn.Sequential(torch.nn.ConvTranspose3d(16, 8, 3, padding=1, output_padding=1, stride=2, bias=False),
                             torch.nn.BatchNorm3d(8)).eval()
    dc.load_state_dict(synth_state_dict(dc.state_dict(), seed=22))
    x = rnd(22, "xdc", 2, 16, 3, 5, 8)
    with torch.no_grad():
        save("layer_deconv", x=x, y=dc(x), cin=16, cout=8, seed=22)
    # window attention (no padding / both padded / only W padded)
    at = ref_attention(channels_3d=128, num_heads=16, block=(4, 4, 4)).eval()
    at.load_state_dict(synth_state_dict(_WindowAttention(128, 16).state_dict(), seed=23))
    for tag, dims in {"nopad": (2, 4, 8, 8), "pad": (1, 4, 6, 7), "padw": (1, 4, 4, 6)}.items():
        x = rnd(23, f"xat{tag}", dims[0], 128, *dims[1:])
        with torch.no_grad():
            save(f"layer_attention_{tag}", x=x, y=at(x), seed=23)
    # one full hourglass
    hg = ref_hourglass(32).eval()
    hg.load_state_dict(synth_state_dict(OursHourglass(32).state_dict(), seed=24), strict=True)
    x = rnd(24, "xhg", 1, 32, 16, 16, 16).relu()
    with torch.no_grad():
        save("layer_hourglass", x=x, y=hg(x), seed=24)

    # ---- 6. model_predictions and ddim_sample at 64x128 -------------------------------------------
    print("diffusion loop")
    b, h, w = 1, 16, 32
    vol = torch.rand(b, 64, 48, h, w, generator=_gen(31, "vol"))
    used0 = torch.rand(b, 4 * h, 4 * w, generator=_gen(31, "used")) * 150 + 10
    dq = F.interpolate(used0.clamp(0, 191).unsqueeze(1), size=(h, w), mode="bilinear") / 4
    with torch.no_grad():
        x_T = reference_x_T(ref, dq)
        t = torch.full((b,), 999, dtype=torch.long)
        pn, xs, pred, pv = ref.model_predictions(vol, x_T, t)
        kk = torch.arange(0, 192, dtype=pred.dtype).view(1, -1, 1, 1)
        unc = torch.sum(torch.abs(pred.unsqueeze(1) - kk) * pv, dim=1)
        save("model_predictions", vol_seed=31, used0=used0, x_T=x_T, t=t, pred_noise=pn, x_start=xs, pred=pred,
             unc=unc, prob_mean=pv.mean(dim=(2, 3)))
        # `used` close to the first-step prediction so both renewal-mask branches are alive
        used = pred + (torch.rand(pred.shape, generator=_gen(31, "jit")) * 4 - 2)
        tape = NoiseTape(seed=77)
        calls = {"n": 0}
        steps = []
        real_randn_like, real_rand_like = torch.randn_like, torch.rand_like

        def fake_randn_like(x, *a, **k):
            calls["n"] += 1
            if calls["n"] % 2 == 1:                              # :354 eps; the even call is q_sample's (:359)
                return tape("eps", tuple(x.shape), x.dtype)
            return torch.zeros_like(x)

        def fake_rand_like(x, *a, **k):
            return tape("fill", tuple(x.shape), x.dtype)

        orig_mp = ref.model_predictions

        def recording_mp(volume, img, tc):
            out = orig_mp(volume, img, tc)
            steps.append((img.clone(), out[0].clone(), out[1].clone(), out[2].clone()))
            return out

        ref.model_predictions = recording_mp
        torch.randn_like, torch.rand_like = fake_randn_like, fake_rand_like
        try:
            final, stack = ref.ddim_sample(vol, used, x_T)
        finally:
            torch.randn_like, torch.rand_like = real_randn_like, real_rand_like
        mask_frac = float(((pred - used).abs() < 1).float().mean())
        print(f"    |disp-used|<1 on {mask_frac:.2%} of pixels, unc<3 on {float((unc < 3).float().mean()):.2%}")
        save("ddim_sample", vol_seed=31, used=used, x_T=x_T, tape_seed=77, final=final, stack=stack,
             state_dtypes=np.array([str(s[0].dtype) for s in steps]),
             x_t_step2=steps[1][0], x_start_step1=steps[0][2], pred_noise_step1=steps[0][1],
             x_t_step5=steps[4][0])

        # ---- 7. whole eval forward (features + attention branch + DDIM), 64x128 ------------------
        print("full forward")
        ref = REF_MODELS["acvnet_ddim"](192, False, False).eval()
        ref.load_state_dict(sd, strict=True)
        batch = synth_stereo_batch(1, 64, 128, seed=41, shifts=(8,))
        tape = NoiseTape(seed=78)
        calls["n"] = 0
        torch.randn_like, torch.rand_like = fake_randn_like, fake_rand_like
        try:
            out = ref(batch["left"], batch["right"], batch["used"], batch["disp"], None)
        finally:
            torch.randn_like, torch.rand_like = real_randn_like, real_rand_like
        save("forward_eval", stereo_seed=41, tape_seed=78, pred=out[0])

    # ---- 8. metrics ----------------------------------------------------------------------------------
    print("metrics")
    stub = types.ModuleType("utils.experiment")
    stub.make_nograd_func = lambda f: f
    pkg = types.ModuleType("utils")
    sys.modules["utils"], sys.modules["utils.experiment"] = pkg, stub
    met = _load_file(REF / "SceneFlow/utils/metrics.py", "ref_metrics")
    gt = torch.rand(3, 16, 24, generator=_gen(51, "gt")) * 200 - 5
    est = gt + rnd(51, "err", 3, 16, 24) * 3
    mask = (gt < 192) & (gt > 0)
    mask[2] = False
    mask[2, 0, :3] = gt[2, 0, :3] > 0                              # ratio < 0.1 -> image skipped
    vals = {"EPE": met.EPE_metric(est, gt, mask), "D1": met.D1_metric(est, gt, mask)}
    for th in (1.0, 2.0, 3.0):
        vals[f"Thres{int(th)}"] = met.Thres_metric(est, gt, mask, th)
    save("metrics", est=est, gt=gt, mask=mask, **vals)
    none = torch.zeros_like(mask)
    none[:, 0, 0] = gt[:, 0, 0] > 0
    save("metrics_all_skipped", est=est, gt=gt, mask=none, EPE=met.EPE_metric(est, gt, none))
    print("done")


if __name__ == "__main__":
    main()
