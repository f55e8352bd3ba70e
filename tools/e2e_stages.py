"""Per-stage times of the end-to-end `test_sample` equivalent (SceneFlow/test_sceneflow_ddim.py:89-122: origin ACVNet ->
used / disp -> ACVNet_DDIM.forward -> metrics) at the bench size, batch 8, 960x512: HIP events around every stage of the
two forwards, host gaps = wall time of the whole call minus the sum of its stages.  Writes gpurun_out/e2e_stages.json.

  python tools/e2e_stages.py            stage table (3 repetitions)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e2e_prof -- python tools/e2e_stages.py --once
                                        kernel statistics of ONE end-to-end pass (profiles/r04_e2e_kernel_stats.csv)"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import diffuvolume_amd as dv  # noqa: E402
from diffuvolume_amd import metrics as M  # noqa: E402
from diffuvolume_amd.submodule import (build_concat_attention_volume, build_gwc_volume, patch_volume,  # noqa: E402
                                       upsample_softmax_regress)
from diffuvolume_amd.synth import _gen, synth_state_dict  # noqa: E402

DEV = "cuda:0"
B, H, W = 8, 512, 960


class Stages:
    """``roof=True``: every stage runs under its own KernelTimer (a HIP-event pair around every kernel launch of the library),
    so that a stage carries the algorithmic flops / bytes and the matrix-pipe flops its kernels issue -- its own roofline."""

    def __init__(self, roof=False):
        self.ev, self.order, self.roof, self.kt = {}, [], roof, {}

    def run(self, name, fn):
        if self.roof:
            from diffuvolume_amd.profiling import KernelTimer
            KernelTimer.active = self.kt.setdefault(name, KernelTimer())
            try:
                return fn()
            finally:
                KernelTimer.active = None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        if name not in self.ev:
            self.ev[name] = []
            self.order.append(name)
        self.ev[name].append((a, b))
        return out

    def table(self, skip=1):
        torch.cuda.synchronize()
        return {n: sum(a.elapsed_time(b) for a, b in self.ev[n][skip:]) / max(1, len(self.ev[n]) - skip) for n in self.order}


def attention_concat(model, fl, fr, st, tag):
    """ACVNet_DDIM.attention_concat_volume (acv_ddim.py:375-390) stage by stage."""
    p = model.prepare()
    gwc = st.run(tag + "gwc volume", lambda: build_gwc_volume(fl, fr, 48, 40))
    pv = st.run(tag + "patch convs", lambda: patch_volume(gwc, p.patch_w1, p.patch_w2, p.patch_dil))
    att = st.run(tag + "attention aggregation (dres1_att, hourglass, classif_att)",
                 lambda: p.classif_att(p.dres2_att(p.dres1_att(pv))))
    cl = st.run(tag + "concat convs (left, right)", lambda: (p.concat_b(p.concat_a(fl)), p.concat_b(p.concat_a(fr))))
    return st.run(tag + "softmax(att) + volume factors", lambda: build_concat_attention_volume(cl[0], cl[1], att, 48, lazy=True))


def main():
    once = "--once" in sys.argv
    g = _gen(7, "e2e")
    left = torch.randn(B, 3, H, W, generator=g).to(DEV)
    right = torch.roll(left, -8, dims=-1)
    gt = (8 + torch.randn(B, H, W, generator=g)).clamp(0.5, 191).to(DEV)
    mask = (gt < 192) & (gt > 0)
    origin = dv.ACVNet(192, False, False)
    origin.load_state_dict(synth_state_dict(origin.state_dict(), seed=3, logit_gain=8.0), strict=True)
    origin = origin.to(DEV).eval()
    ddim = dv.ACVNet_DDIM(192, False, False)
    ddim.load_state_dict(synth_state_dict(ddim.state_dict(), seed=1, logit_gain=8.0), strict=True)
    ddim = ddim.to(DEV).eval()
    st = Stages()

    def test_sample():
        nonlocal st
        with torch.no_grad():
            origin.prepare(check_weights=True)
            fl = st.run("origin: feature CNN (left)", lambda: origin.feature_extraction(left)["gwc_feature"])
            fr = st.run("origin: feature CNN (right)", lambda: origin.feature_extraction(right)["gwc_feature"])
            vol = attention_concat(origin, fl, fr, st, "origin: ")
            cost = st.run("origin: aggregation (dres0..classif2)", lambda: origin._aggregate(vol, None))
            used = st.run("origin: upsample + softmax + regression", lambda: upsample_softmax_regress(cost, want_uncertainty=False)[0])
            dn = st.run("glue: clamp + bilinear /4 of `used`",
                        lambda: torch.nn.functional.interpolate(torch.clamp(used, 0, 191).unsqueeze(1), size=(H // 4, W // 4), mode="bilinear") / 4)
            ddim.prepare(check_weights=True)
            fl2 = st.run("ddim: feature CNN (left)", lambda: ddim.feature_extraction(left)["gwc_feature"])
            fr2 = st.run("ddim: feature CNN (right)", lambda: ddim.feature_extraction(right)["gwc_feature"])
            vol2 = attention_concat(ddim, fl2, fr2, st, "ddim: ")
            x_T = st.run("ddim: two-hot x_T", lambda: ddim.encode_disparity(dn))
            pred = st.run("ddim: 5-step ddim_sample (the hot path)", lambda: ddim.ddim_sample(vol2, used, x_T)[0])
            return st.run("metrics", lambda: M.batch_metrics(pred, gt, mask))

    test_sample()                                   # warm-up: plans, allocator pools
    torch.cuda.synchronize()
    if once:
        test_sample()
        torch.cuda.synchronize()
        return
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        test_sample()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    tab = st.table(skip=1)
    total = sum(tab.values())
    # and the two public forwards as the caller runs them (no stage events)
    def plain():
        with torch.no_grad():
            used = origin(left, right)[-1]
            dn = torch.nn.functional.interpolate(torch.clamp(used, 0, 191).unsqueeze(1), size=(H // 4, W // 4), mode="bilinear") / 4
            return M.batch_metrics(ddim(left, right, used, dn, None)[0], gt, mask)
    plain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plain()
    torch.cuda.synchronize()
    wall_plain = (time.perf_counter() - t0) / reps * 1e3
    # per-stage rooflines: one more pass with a HIP-event pair around every kernel (kernel times only; slower than the
    # stage times above by the event overhead, so the fractions are computed from THIS pass's kernel times)
    PEAK_TF, PEAK_GBS = 157.3, 8000.0
    rs = Stages(roof=True)
    st_saved = st
    st = rs
    test_sample()
    st = st_saved
    roof = {}
    for name, kt in rs.kt.items():
        ks = kt.summary()
        ms = sum(v["total_ms"] for v in ks.values())
        if ms <= 0:
            continue
        fl, iss, by = (sum(v[k] for v in ks.values()) for k in ("flops", "issued_flops", "bytes"))
        top = max(ks, key=lambda k: ks[k]["total_ms"])
        roof[name] = {"kernel_ms": round(ms, 3), "launches": sum(v["launches"] for v in ks.values()),
                      "algorithmic_tflops": round(fl / ms / 1e9, 1), "issued_frac_of_mfma_f32_peak": round(iss / ms / 1e9 / PEAK_TF, 3),
                      "algorithmic_gb_s": round(by / ms / 1e6, 0), "frac_of_hbm_peak": round(by / ms / 1e6 / PEAK_GBS, 3),
                      "bound": "mfma" if iss / ms / 1e9 / PEAK_TF > by / ms / 1e6 / PEAK_GBS else "hbm",
                      "largest_kernel": top, "largest_kernel_share": round(ks[top]["total_ms"] / ms, 2)}
    out = {"batch": B, "size": [H, W], "stages_ms": {k: round(v, 3) for k, v in tab.items()}, "sum_of_stages_ms": round(total, 2),
           "stage_rooflines": roof,
           "stage_rooflines_note": "per stage: sum over its kernel launches of algorithmic flops / bytes and of the flops the matrix "
                                   "pipe really issues (Winograd / polyphase forms issue 2.25x / 1.44x fewer), over the HIP-event "
                                   "time of those launches; peaks 157.3 TFLOP/s (fp32 MFMA) and 8 TB/s",
           "wall_ms_with_stage_events": round(wall, 2), "wall_ms_public_forwards": round(wall_plain, 2),
           "host_gap_ms": round(wall - total, 2), "pairs_per_s": round(B / (wall_plain * 1e-3), 2)}
    os.makedirs(ROOT / "gpurun_out", exist_ok=True)
    with open(ROOT / "gpurun_out" / "e2e_stages.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
