"""Full-size timing of the KITTI12 (config 4) and KITTI15 (config 5, volume front + lookup) flavours on one GPU.
Not a bench line (BASELINE.json: the other configs are parity cases); the numbers go to DESIGN.md section 5c.
    python tools/bench_flavours.py [--pcw] [--igev]"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from diffuvolume_amd.profiling import KernelTimer  # noqa: E402
from diffuvolume_amd.synth import _gen, synth_state_dict, synth_stereo_batch  # noqa: E402

DEV = "cuda:0"


def timeit(fn, warmup=1, steps=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def kernels(fn):
    kt = KernelTimer()
    KernelTimer.active = kt
    try:
        fn()
    finally:
        KernelTimer.active = None
    # tag -> [launches, ms, algorithmic TFLOP/s, TFLOP/s issued on the matrix pipe (Winograd: algorithmic / 2.25)]
    return {k: [v["launches"], round(v["total_ms"], 3), round(v["flops"] / max(v["total_ms"], 1e-9) / 1e9, 1),
                round(v["issued_flops"] / max(v["total_ms"], 1e-9) / 1e9, 1)]
            for k, v in kt.summary().items()}


def pcw(b=4, h=384, w=1248):
    from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
    m = PWCNet_ddim(192, True)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=2, logit_gain=8.0, scale={"refinenet3.conv8.weight": 0.002}))
    m = m.to(DEV).eval()
    batch = {k: v.to(DEV) for k, v in synth_stereo_batch(b, h, w, seed=0).items()}
    with torch.no_grad():
        fl = m.feature_extraction(batch["left"] * 0.05)
        fr = m.feature_extraction(batch["right"] * 0.05)
        combine = m.fused_volume(fl, fr)
        x_t = m.encode_disparity(batch["disp"])
    res = {"config": f"KITTI12 PCWNet+DiffuVolume B={b} {w}x{h}, 3 DDIM steps"}
    with torch.no_grad():
        res["volume_build_ms"] = timeit(lambda: m.fused_volume(fl, fr))
        res["ddim_sample_ms"] = timeit(lambda: m.ddim_sample(combine, batch["used"], x_t, fl, fr))
        res["forward_ms"] = timeit(lambda: m(batch["left"] * 0.05, batch["right"] * 0.05, batch["used"], batch["disp"]))
        res["ddim_sample_kernels_ms"] = kernels(lambda: m.ddim_sample(combine, batch["used"], x_t, fl, fr))
    res["pairs_per_s_hot"] = b / ((res["volume_build_ms"] + res["ddim_sample_ms"]) / 1e3)
    res["pairs_per_s_forward"] = b / (res["forward_ms"] / 1e3)
    return res


def igev(b=4, h=96, w=312):
    from diffuvolume_amd.geometry_ddim import Combined_Geo_Encoding_Volume
    from diffuvolume_amd.igev_stereo_ddim import IGEVCostVolume
    m = IGEVCostVolume()
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=91, logit_gain=60.0))
    m = m.to(DEV).eval()
    ml = torch.randn(b, 96, h, w, generator=_gen(5, "ml"))
    mr = torch.roll(ml, -5, dims=-1) + 0.1 * torch.randn(b, 96, h, w, generator=_gen(5, "mr"))
    feats = [torch.randn(b, c, h // s, w // s, generator=_gen(5, f"f{i}")).to(DEV)
             for i, (c, s) in enumerate(((96, 1), (64, 2), (192, 4), (160, 8)))]
    ml, mr = ml.to(DEV), mr.to(DEV)
    res = {"config": f"KITTI15 IGEV cost-volume front B={b} 1/4 res {w}x{h}, D/4=48"}
    with torch.no_grad():
        res["front_ms"] = timeit(lambda: m(ml, mr, feats))
        geo, init = m(ml, mr, feats)
        fn = Combined_Geo_Encoding_Volume(ml, mr, geo, radius=4, num_levels=2)
        coords = torch.arange(w, dtype=torch.float32, device=DEV).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
        noisy = torch.rand(b, 48, h, w, device=DEV)
        res["geo_lookup_ms"] = timeit(lambda: fn(init, coords, noisy), warmup=2, steps=20)
        res["front_kernels_ms"] = kernels(lambda: m(ml, mr, feats))
    res["lookups_per_pair_cfg5"] = "20 steps x 32 iters = 640 (reference default 2 x 32)"
    # one GRU iteration of the update block (KITTI15/core/update.py) at the same size: HIP convs with fused gates
    # vs the same arithmetic written with torch ops (MIOpen convolutions + ATen elementwise)
    import types
    from diffuvolume_amd.update import BasicMultiUpdateBlock
    ub = BasicMultiUpdateBlock(types.SimpleNamespace(corr_levels=2, corr_radius=4, n_gru_layers=3, n_downsample=2),
                               hidden_dims=[128, 128, 128])
    ub.load_state_dict(synth_state_dict(ub.state_dict(), seed=101))
    ub = ub.to(DEV).eval()
    dims = [(h, w), (h // 2, w // 2), (h // 4, w // 4)]
    net = [torch.tanh(torch.randn(b, 128, hh, ww, device=DEV)) for hh, ww in dims]
    inp = [[torch.randn(b, 128, hh, ww, device=DEV) * 0.5 for _ in range(3)] for hh, ww in dims]
    corr = torch.randn(b, 162, h, w, device=DEV)
    dsp = torch.rand(b, 1, h, w, device=DEV) * 40

    def torch_gru(g, hh, cz, cr, cq, *xs):
        x = torch.cat(xs, 1)
        hx = torch.cat([hh, x], 1)
        z = torch.sigmoid(g.convz(hx) + cz)
        r = torch.sigmoid(g.convr(hx) + cr)
        q = torch.tanh(g.convq(torch.cat([r * hh, x], 1)) + cq)
        return (1 - z) * hh + z * q

    def torch_update():
        import torch.nn.functional as F
        from diffuvolume_amd.update import interp, pool2x
        n = list(net)
        n[2] = torch_gru(ub.gru16, n[2], *inp[2], pool2x(n[1]))
        n[1] = torch_gru(ub.gru08, n[1], *inp[1], pool2x(n[0]), interp(n[2], n[1]))
        e = ub.encoder
        cor = F.relu(e.convc2(F.relu(e.convc1(corr))))
        d_ = F.relu(e.convd2(F.relu(e.convd1(dsp))))
        mf = torch.cat([F.relu(e.conv(torch.cat([cor, d_], 1))), dsp], 1)
        n[0] = torch_gru(ub.gru04, n[0], *inp[0], mf, interp(n[1], n[0]))
        dh = ub.disp_head
        return n, F.relu(ub.mask_feat_4[0](n[0])), dh.conv2(F.relu(dh.conv1(n[0])))

    with torch.no_grad():
        res["update_block_hip_ms"] = timeit(lambda: ub(list(net), inp, corr, dsp), warmup=2, steps=10)
        res["update_block_torch_ms"] = timeit(torch_update, warmup=2, steps=10)
        res["update_block_kernels_ms"] = kernels(lambda: ub(list(net), inp, corr, dsp))
        # the whole per-batch DDIM loop at the reference's defaults (2 DDIM steps) and 32 GRU iterations per step
        from diffuvolume_amd.igev_stereo_ddim import DynamicHead180, IGEVDiffusionLoop
        from diffuvolume_amd.synth import toy_upsample_disp
        head = DynamicHead180()
        head.load_state_dict(synth_state_dict(head.state_dict(), seed=81))
        loop = IGEVDiffusionLoop(head.to(DEV).eval(), ub, toy_upsample_disp, n_gru_layers=3, slow_fast_gru=False)
        used = torch.nn.functional.interpolate(init * 4, scale_factor=4, mode="bilinear")
        asd = torch.rand(b, 48, h, w, device=DEV) * 2 - 1
        run = lambda: loop.ddim_sample(init, init, None, 32, list(net), inp, fn, used, asd, None)
        res["ddim_loop_2steps_32iters_ms"] = timeit(run, warmup=1, steps=2)
        res["ddim_loop_pairs_per_s_2steps"] = b / (res["ddim_loop_2steps_32iters_ms"] / 1e3)
        res["ddim_loop_pairs_per_s_20steps_extrapolated"] = b / (10 * res["ddim_loop_2steps_32iters_ms"] / 1e3)
    return res


def igev_model(b=4, h=384, w=1248, steps=20, iters=32, quick=False):
    """BASELINE config 5 on one GPU: the IGEVStereo_ddim drop-in module end to end (stub MobileNetV2 backbone: timm's
    pretrained one does not exist offline), 1248x384, `steps` DDIM steps x `iters` GRU iterations, batch b."""
    import types
    import torch.nn.functional as F
    from diffuvolume_amd.igev_stereo_ddim import Feature, IGEVStereo_ddim
    from diffuvolume_amd.synth import StubMobileNetV2
    args = types.SimpleNamespace(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
                                 slow_fast_gru=False, max_disp=192, mixed_precision=False)
    cof = [0.5] + [0.0] * (steps - 1) + [0.5] if steps != 2 else None
    m = IGEVStereo_ddim(args, feature=Feature(StubMobileNetV2()), sampling_timesteps=steps, ensemble_cof=cof)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=7, scale={"update_block.disp_head.conv2.weight": 0.05,
                                                                      "update_block.disp_head.conv2.bias": 0.0,
                                                                      "classifier.weight": 20.0}), strict=True)
    m = m.to(DEV).eval()
    g = _gen(77, "cfg5")
    img1 = (torch.rand(b, 3, h, w, generator=g) * 255).to(DEV)
    img2 = torch.roll(img1, -9, dims=-1)
    flow_full = (9 + torch.randn(b, 1, h, w, generator=g)).clamp(0.5, 47).to(DEV)
    flow_gt = F.interpolate(flow_full, size=(h // 4, w // 4), mode="bilinear") / 4
    with torch.no_grad():
        if quick:       # bench.py extras: one short warm-up pass (2 GRU iterations per step), one timed full pass
            m(img1, img2, flow_full, flow_gt, iters=2, test_mode=True)
            ms = timeit(lambda: m(img1, img2, flow_full, flow_gt, iters=iters, test_mode=True), warmup=0, steps=1)
            ks = kernels(lambda: m(img1, img2, flow_full, flow_gt, iters=2, test_mode=True))
        else:
            ms = timeit(lambda: m(img1, img2, flow_full, flow_gt, iters=iters, test_mode=True), warmup=1, steps=2)
            ks = None
    if ks is not None:
        return {"config": f"KITTI15 IGEVStereo_ddim B={b} {w}x{h}, {steps} DDIM steps x {iters} GRU iterations (stub backbone)",
                "forward_ms": ms, "pairs_per_s": b / (ms / 1e3), "ms_per_gru_iteration": ms / (steps * iters),
                "kernels_of_a_2_iteration_pass": ks}
    return {"config": f"KITTI15 IGEVStereo_ddim B={b} {w}x{h}, {steps} DDIM steps x {iters} GRU iterations (stub backbone)",
            "forward_ms": ms, "pairs_per_s": b / (ms / 1e3), "ms_per_gru_iteration": ms / (steps * iters)}


def igev_reference_default(b=1, h=384, w=1248, iters=32):
    """KITTI15 the way the reference evaluates it (KITTI15/evaluate_stereo.py:88-129): the origin IGEVStereo forward, its
    disparity down-sampled to 1/4, then IGEVStereo_ddim at the reference's own hard-coded 2 DDIM steps
    (core/igev_stereo_ddim.py:124), `iters` GRU iterations each, batch b (the reference: 1).  Stub MobileNetV2 backbone."""
    import types
    import torch.nn.functional as F
    from diffuvolume_amd.igev_stereo import IGEVStereo
    from diffuvolume_amd.igev_stereo_ddim import Feature, IGEVStereo_ddim
    from diffuvolume_amd.synth import StubMobileNetV2
    args = types.SimpleNamespace(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
                                 slow_fast_gru=False, max_disp=192, mixed_precision=False)
    sc = {"update_block.disp_head.conv2.weight": 0.05, "update_block.disp_head.conv2.bias": 0.0, "classifier.weight": 20.0}
    m = IGEVStereo_ddim(args, feature=Feature(StubMobileNetV2()))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=7, scale=sc), strict=True)
    m = m.to(DEV).eval()
    o = IGEVStereo(args, feature=Feature(StubMobileNetV2()))
    o.load_state_dict(synth_state_dict(o.state_dict(), seed=8, scale=sc), strict=True)
    o = o.to(DEV).eval()
    g = _gen(77, "cfg5")
    img1 = (torch.rand(b, 3, h, w, generator=g) * 255).to(DEV)
    img2 = torch.roll(img1, -9, dims=-1)

    def both():
        flow_pr = o(img1, img2, iters=iters, test_mode=True)
        flow_4 = F.interpolate(torch.clamp(flow_pr, 0, w - 1), size=(h // 4, w // 4), mode="bilinear") / 4
        return m(img1, img2, flow_pr, flow_4, iters=iters, test_mode=True)

    with torch.no_grad():
        ms = timeit(both, warmup=1, steps=3)
    return {"config": f"KITTI15 origin IGEVStereo + IGEVStereo_ddim (2 DDIM steps x {iters} GRU iterations) B={b} {w}x{h} (stub backbone)",
            "ms_per_pair": ms / b, "ms_per_batch": ms, "batch": b}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pcw", action="store_true")
    ap.add_argument("--igev", action="store_true")
    ap.add_argument("--igev-model", action="store_true", help="config 5: the whole IGEVStereo_ddim forward")
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-ksplit", action="store_true", help="A/B: one block per output tile for the small launches too")
    a = ap.parse_args()
    if a.no_ksplit:
        from diffuvolume_amd import submodule as S
        S.Conv2dPlan.KSPLIT = False
    out = {}
    if a.igev_model:
        out["igev_model"] = igev_model(b=a.batch, steps=a.ddim_steps)
        print(json.dumps(out, indent=1))
        return
    if a.igev or not (a.pcw or a.igev):
        out["igev"] = igev(b=a.batch)
    if a.pcw or not (a.pcw or a.igev):
        out["pcw"] = pcw()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
