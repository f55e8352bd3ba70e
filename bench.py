#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the DiffuVolume hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sceneflow|kitti12|kitti15] [--batch B] [--height H] [--width W]

One "step" = one pass of the hot path (SURVEY 8d "HOT") over one batch of synthetic stereo
pairs per GPU, inputs already resident in HBM:
    build_gwc_volume (K1)  +  softmax(att) * build_concat_volume (K2)
    + S DDIM steps x [ time-shift filter (K3) -> dres0/dres1/hourglass x2/classif2 (K4-K6)
                       -> trilinear/softmax/regression (K7) -> two-hot + DDIM update (K8) ]
    + masked EPE/D1 sums (K9);  one all-reduce of the metric sums after the K steps (RCCL).
Workload = BASELINE.json configs[1]: SceneFlow ACVNet+DiffuVolume, 960x540 frames cropped to
960x512 as the reference's loader does (sceneflow_dataset.py:60-65), maxdisp 192, batch 8 per
GPU, 5 DDIM steps, fp32.  N>1: one process per GPU over RCCL, weak scaling -- either launched by
torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when those are absent, by this
script itself: `python bench.py --gpus N` starts N workers (before anything touches the GPU), and
fails if fewer than N devices are visible.  It never silently measures fewer GPUs than asked.

Prints ONE JSON line (rank 0): the driver contract plus `roofline` (dominant kernel: the
fp32-MFMA Winograd conv of the 32-channel layers, HIP-event timed inside the timed region) and `cpu_baseline`
(the CPU oracle on a bounded sample of the same workload, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0
ALGO_BYTES_PER_PAIR = 23.2e9      # ideal-fusion fp32 HBM bytes of HOT per pair (5 steps)
ALGO_FLOP_PER_PAIR = 3.76e12


# Dominant kernel = one rocprofv3 kernel name: the Winograd instantiation the 32->32 layers run (no filter prologue,
# one plane per wave, 2 x 8 tiles per wave): 20 launches per DDIM step, a third of the step.
DOMINANT_KERNEL = "conv3d_wino_kernel<false, 1, 0>"
DOMINANT_TAGS = ("conv3d_k3s1_co32",)
WINO_MULT_REDUCTION = 2.25        # F(2x2,3x3) in-plane: 16 multiplies per 2x2 outputs and depth tap instead of 36
S2PP_MULT_REDUCTION = 1.44        # polyphase F(2,2) stride-2 form: 25 instead of 36
WINO3_MULT_REDUCTION = 3.375      # F(2x2x2,3x3x3): 64 multiplies per 2x2x2 outputs instead of 216
# the next kernels by time: (KernelTimer tags, rocprofv3 kernel name, issued-flop divisor).  The 64- and 128-channel
# layers run the F(2x2x2,3x3x3) kernel (round 6, csrc/conv3d_wino3.hip): 4 x 4 tiles per wave on the 120-wide planes, 2 x 8
# on the 60-wide ones (padded to 64: narrow tiles stage more surplus columns than the padding costs).
SIDE_KERNELS = [
    # transposed convolution + fused redir (round 6): persistent, 8 MFMA waves in two groups + 4 loader waves per CU
    (("deconv3d_k3s2_redir",), "deconv3d_pl_kernel<true, false, 0>", 1.0),
    # stride 2 (round 5): the polyphase minimal-filtering kernel, 8 x 8-output patches, persistent with loader waves; it
    # issues 1.44 x fewer multiplies than the direct count (25 instead of 36 per 2 x 2 outputs and depth tap)
    (("conv3d_k3s2_co64", "conv3d_k3s2_co128"), "conv3d_s2pp_kernel<1>", S2PP_MULT_REDUCTION),
    (("conv3d_k3s1_co64",), "conv3d_wino3_kernel<1>", WINO3_MULT_REDUCTION),
    (("conv3d_k3s1_co128",), "conv3d_wino3_kernel<0>", WINO3_MULT_REDUCTION),
]


# BASELINE.json configs by workload name: per-GPU batch, frame size and DDIM steps the config names
WORKLOADS = {
    "sceneflow": {"defaults": dict(batch=8, height=512, width=960, ddim_steps=5),
                  "metric": "stereo pairs/sec, SceneFlow 960x540 (cropped 960x512) maxdisp=192, hot path"},
    "kitti12": {"defaults": dict(batch=4, height=384, width=1248, ddim_steps=3),
                "metric": "stereo pairs/sec, KITTI12 PCWNet+DiffuVolume 1248x384 maxdisp=192, hot path"},
    "kitti15": {"defaults": dict(batch=4, height=384, width=1248, ddim_steps=20),
                "metric": "stereo pairs/sec, KITTI15 IGEV-Stereo+DiffuVolume 1248x384, whole forward"},
}


def env_overrides():
    """The package's environment switches that are set for this run (diffuvolume_amd/_env.py: the one list of them): a
    measured number carries the switches it was measured under.  {} on a default run."""
    from diffuvolume_amd._env import overrides
    return overrides()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command (profiles/*_pmc_traffic.json, gfx950 x2 read correction applied).  The profile carries the fingerprint of
    the kernel sources it was collected on (`csrc_sha16`, diffuvolume_amd/_build.py): a profile of other sources is
    refused (None) -- counter figures of an older kernel are never quoted for the current one."""
    files = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))
    if not files:
        return None
    try:
        from diffuvolume_amd._build import csrc_sha16
        doc = json.loads(files[-1].read_text())
        if doc.get("csrc_sha16") != csrc_sha16():
            return None
        for k in doc["kernels"]:
            if k["kernel"] == kernel:
                return k["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def pmc_traffic_stamp():
    files = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))
    if not files:
        return None
    try:
        from diffuvolume_amd._build import csrc_sha16
        doc = json.loads(files[-1].read_text())
        return {"file": f"profiles/{files[-1].name}", "git_head": doc.get("git_head"), "csrc_sha16": doc.get("csrc_sha16"),
                "current_csrc_sha16": csrc_sha16(), "stale": doc.get("csrc_sha16") != csrc_sha16()}
    except (OSError, ValueError):
        return None


def visible_gpu_count():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the launcher must not initialise the GPU
    before it starts its workers): KFD topology nodes with SIMDs, cut down by the *_VISIBLE_DEVICES lists."""
    nodes = Path("/sys/class/kfd/kfd/topology/nodes")
    n = None
    try:
        n = 0
        for d in nodes.iterdir():
            props = dict(l.split(None, 1) for l in (d / "properties").read_text().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    if n is None:
        n = torch.cuda.device_count()       # no KFD sysfs: fall back to the library call
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def under_profiler():
    """rocprofv3 preloads its tool library into the process it starts, and that library initialises the GPU: starting
    worker processes from such a parent is the exec-after-init hop this pool forbids."""
    keys = ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH", "ROCPROF_OUTPUT_FILE_NAME")
    return any(os.environ.get(k) for k in keys) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="sceneflow",
                    help="sceneflow = BASELINE configs 2 / 3 (the headline; default), kitti12 = config 4, kitti15 = config 5")
    ap.add_argument("--batch", type=int, default=None, help="stereo pairs per GPU (default: 8 / 4 / 4)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--ddim-steps", type=int, default=None, help="default: 5 / 3 / 20 (BASELINE configs 2, 4, 5)")
    ap.add_argument("--gru-iters", type=int, default=32, help="kitti15: GRU iterations per DDIM step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the hipGraph, end-to-end and config 4 / 5 side measurements")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check of the N-process launch on CPU (gloo): rendezvous + one all-reduce, no GPU work")
    a = ap.parse_args()
    for k, v in WORKLOADS[a.workload]["defaults"].items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def launch_workers(a, argv):
    """`python bench.py --gpus N` without a launcher: start N worker processes (one per GPU, the same command line)
    with the torchrun environment and wait for them.  Runs BEFORE this process touches the GPU and never re-execs
    itself; the devices are counted from the KFD topology in sysfs, not through the HIP runtime."""
    import socket
    import subprocess
    if under_profiler() and not a.dry_run:
        print("bench.py: refusing to start worker processes from a profiled parent (the profiler's preloaded library "
              "has already initialised the GPU here); launch the workers first -- torchrun ... bench.py --gpus N -- and "
              "profile a worker", file=sys.stderr)
        return 2
    oversub = os.environ.get("DV_BENCH_OVERSUBSCRIBE") == "1"     # plumbing test of the N-rank path on a smaller box:
    if not a.dry_run and not oversub:                              # ranks share devices, rendezvous over gloo
        n_vis = visible_gpu_count()
        if n_vis < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {n_vis} GPU(s) visible; refusing to measure fewer", file=sys.stderr)
            return 2
    from diffuvolume_amd.distributed import free_port
    port = free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", DV_BENCH_SELF_LAUNCHED="1")
        if oversub:
            env["DV_DIST_BACKEND"] = "gloo"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or (code if code > 0 else 1)
                for q in alive:               # a dead rank would leave the others waiting in a collective
                    q.terminate()
        time.sleep(0.05)
    return rc


def dry_run(a):
    """Worker side of --dry-run: rendezvous over gloo, one all-reduce, rank 0 prints what it saw."""
    import torch.distributed as dist
    from diffuvolume_amd import distributed as D
    rank, world, _ = D.init_from_env(backend="gloo")
    seen = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    lo, hi = D.shard_range(a.batch * world, rank, world)
    shard = torch.tensor([float(hi - lo)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(shard)
    if rank == 0:
        print(json.dumps({"dry_run": True, "workload": a.workload, "n_gpus": world, "ranks": int(seen.item()), "backend": "gloo",
                          "global_batch": int(shard.item()), "launcher": "self" if os.environ.get("DV_BENCH_SELF_LAUNCHED") else "torchrun"}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def under_launcher() -> bool:
    """A real launcher environment: RANK, LOCAL_RANK, WORLD_SIZE and MASTER_PORT all set (torchrun, or this script
    launching its own ranks).  A bare WORLD_SIZE=1 exported by a scheduler does not start a process group."""
    return all(k in os.environ for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"))


def launcher_label() -> str:
    if os.environ.get("DV_BENCH_SELF_LAUNCHED"):
        return "self"
    if os.environ.get("TORCHELASTIC_RUN_ID"):
        return "torchrun"
    return "env" if under_launcher() else "single"


def make_inputs(batch, h, w, seed, device):
    """Synthetic quarter-resolution features with a real correlation ridge."""
    from diffuvolume_amd.synth import synth_hot_inputs
    host = synth_hot_inputs(batch, h, w, seed)
    return host, {k: v.to(device) for k, v in host.items()}


def hot_path(model, x, tape=None):
    """One pass of the hot path; returns (final disparity, per-step stack, gwc volume)."""
    import diffuvolume_amd as dv
    gwc = dv.build_gwc_volume(x["fl"], x["fr"], 48, 40)
    # the attention-concat volume as its factors (what ACVNet_DDIM.forward passes): its one consumer, dres0[0], reads
    # the factors, so the [B,64,48,h,w] tensor of acv_ddim.py:390 is not written (config: "attention_concat_volume")
    vol = dv.build_concat_attention_volume(x["cl"], x["cr"], x["att"], 48, lazy=True)
    x_T = model.encode_disparity(x["dq"])
    final, stack = model.ddim_sample(vol, x["used"], x_T, noise=tape)
    return final, stack, gwc


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return min(int(n), len(os.sched_getaffinity(0)))
    except Exception:                         # noqa: BLE001
        pass
    return len(os.sched_getaffinity(0))


def cpu_baseline(sd, host, ddim_steps, cof, seed=1):
    """The CPU oracle on a bounded sample of the same workload: pair 0 at full size, both builders and the WHOLE
    S-step DDIM loop (nothing extrapolated), noise from a NoiseTape so that the GPU can be compared on the same draws.
    Returns (baseline dict, everything the parity leg needs)."""
    from oracle import acv_oracle as O
    from oracle import loop_parity as LP
    one = {k: v[:1].clone() for k, v in host.items()}
    orc = O.ACVDiffusionOracle(sd, sampling_timesteps=ddim_steps, cof=cof)
    cores = physical_cores()
    torch.set_num_threads(cores)          # one thread per PHYSICAL core, count stated
    t0 = time.perf_counter()
    gwc = O.build_gwc_volume(one["fl"], one["fr"], 48, 40)
    vol = O.attention_concat_volume(one["att"], O.build_concat_volume(one["cl"], one["cr"], 48))
    x_T = orc.encode_x_T(one["dq"])
    t1 = time.perf_counter()
    final, stack, trace = LP.oracle_trajectory(orc, vol, one["used"], x_T, seed)
    t2 = time.perf_counter()
    del gwc
    tb, tl = t1 - t0, t2 - t1
    base = {"value": 1.0 / (tb + tl), "unit": "pairs/s", "cores": cores, "logical_cpus": os.cpu_count(), "kind": "port",
            "extrapolated": False,
            "sample": f"oracle/acv_oracle.py, 1 pair 960x512, the whole hot path: builders {tb:.2f} s + "
                      f"{ddim_steps} DDIM steps {tl:.2f} s",
            "builders_s": tb, "ddim_loop_s": tl}
    return base, dict(one=one, x_T=x_T, final=final, stack=stack, trace=trace, seed=seed, sd=sd, orc=orc, cof=cof)


def parity_vs_oracle(model, x, ref):
    """All S steps of pair 0 against the oracle run of `cpu_baseline` (oracle/loop_parity.py): every step from the
    oracle's state (teacher forced: the contract's bars apply), and the free run with the count of renewal
    decisions that came out differently."""
    from oracle import loop_parity as LP
    import diffuvolume_amd as dv
    one = {k: v[:1] for k, v in x.items()}
    host = ref["one"]
    with torch.no_grad():
        vol_d = dv.build_concat_attention_volume(one["cl"], one["cr"], one["att"], 48)
        tf = LP.teacher_forced(model, ref["trace"], vol_d, one["used"], host["used"], host["gt"])
        fr = LP.free_run(model, ref["trace"], ref["stack"], ref["final"], vol_d, one["used"], ref["x_T"], host["gt"],
                         ref["seed"])
    keys = ("step", "mean_abs_px", "frac_gt_1e-3", "max_px", "share_unc_lt_3", "frac_gt_1e-3_where_unc_lt_3",
            "frac_gt_bar", "unc_mean_px", "epe_delta", "flips_mask_zero")
    # float64 triangulation of the first two steps (step 1 enters with the float32 state, step 2 with the float64 state
    # every later step has): HIP and the fp32 oracle each against the oracle evaluated in float64, from the same state.
    # Two fp32 evaluations that are each ~1e-4 px from the truth are ~sqrt(2) x that from one another: this is why
    # `within_raw_bars` (HIP vs fp32 oracle, all pixels) can be false while HIP is as close to float64 as the oracle is.
    tri = None
    if ref.get("sd") is not None:
        from oracle import acv_oracle as O
        sd64 = {k: (v.double() if v.is_floating_point() and not k.startswith("time_embedding") else v) for k, v in ref["sd"].items()}
        with torch.no_grad():
            vol_h = O.attention_concat_volume(host["att"], O.build_concat_volume(host["cl"], host["cr"], 48))
            t3 = LP.teacher_forced_vs_fp64(model, ref["orc"], O.ACVDiffusionOracle(sd64, sampling_timesteps=len(ref["trace"]), cof=ref["cof"]),
                                           ref["trace"][:2], vol_h, vol_d, one["used"])
        pick = lambda d: {k: d.get(k) for k in ("mean_abs_px", "frac_gt_1e-3", "max_px")}
        tri = [{"step": s["step"], "hip_vs_fp64": pick(s["hip_vs_fp64"]), "oracle32_vs_fp64": pick(s["oracle32_vs_fp64"]),
                "hip_vs_oracle32": pick(s["hip_vs_oracle32"])} for s in t3]
    return {"bars": {"px": LP.BAR_PX, "frac": LP.BAR_FRAC, "epe": LP.BAR_EPE,
                     "note": "frac_gt_1e-3 = RAW share of all pixels beyond 1e-3 px (the contract's figure); "
                             "share_unc_lt_3 = share of pixels the reference itself calls confident (uncertainty < 3 px, "
                             "acv_ddim.py:330) and frac_gt_1e-3_where_unc_lt_3 = the raw bar on those pixels only; "
                             "frac_gt_bar = a builder-defined bar that grows as unc/3 on the other pixels (soft-argmax "
                             "sensitivity; these untrained weights give unc ~ 30-50 px) -- NOT the contract bar"},
            "teacher_forced": [{k: s.get(k) for k in keys} for s in tf],
            "fp64_triangulation_steps_1_2": tri,
            "hip_frac_within_bar_vs_fp64_steps_1_2": None if tri is None else all(s["hip_vs_fp64"]["frac_gt_1e-3"] <= LP.BAR_FRAC for s in tri),
            "px_bar_note": "north_star's `within 1e-3 px` holds per step as a 99.9 % quantile over all pixels (max 1.2-2.1e-3 px) "
                           "and as a maximum only on the final ensemble output (BASELINE.md section 5)",
            "within_raw_bars": all(s["frac_gt_1e-3"] <= LP.BAR_FRAC and s["epe_delta"] < LP.BAR_EPE for s in tf),
            "within_raw_bars_where_unc_lt_3": all(s["frac_gt_1e-3_where_unc_lt_3"] <= LP.BAR_FRAC and s["epe_delta"] < LP.BAR_EPE for s in tf),
            "within_spread_scaled_bars": all(s["frac_gt_bar"] <= LP.BAR_FRAC and s["epe_delta"] < LP.BAR_EPE for s in tf),
            "free_run": [{k: s.get(k) for k in keys} for s in fr["steps"]], "free_run_final": fr["final"],
            "final_within_raw_bars": fr["final"]["frac_gt_1e-3"] <= LP.BAR_FRAC and fr["final"]["epe_delta"] < LP.BAR_EPE}


def parity_calibrated(a, host, x, device):
    """The same comparison on the CALIBRATED network (oracle/calibrate.py): these synthetic weights with the BatchNorm
    buffers of the loop layers holding the statistics of the data (tests/golden/acv_calibrated_fullsize.npz, written by
    oracle/make_golden_acv_calibrated.py; classifier gain 1), as a trained checkpoint's buffers would.  On it the fp32
    oracle is within 1e-3 px of its own float64 evaluation on every pixel, so `within_raw_bars` -- the contract as
    written, all pixels, every step -- is a statement about the implementation and not about the conditioning of an
    untrained network.  Not timed; the oracle is the checker."""
    import numpy as np
    import diffuvolume_amd as dv
    from diffuvolume_amd.synth import synth_state_dict
    from oracle import calibrate as C
    f = ROOT / "tests" / "golden" / "acv_calibrated_fullsize.npz"
    if not f.exists():
        return {"error": "tests/golden/acv_calibrated_fullsize.npz is missing"}
    with np.load(f, allow_pickle=False) as z:
        gain, stats = float(z["gain"]), C.unpack(z["bn_keys"], z["bn_vals"], z["bn_lens"])
    m = dv.ACVNet_DDIM(192, False, False)
    sd = synth_state_dict(m.state_dict(), seed=1, logit_gain=gain)
    sd.update(stats)
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    _, ref = cpu_baseline(sd, host, a.ddim_steps, m.ensemble_cof)
    rep = parity_vs_oracle(m, x, ref)
    rep["network"] = ("BatchNorm buffers of dres0..classif2 = statistics of the synthetic data, classifier gain "
                      f"{gain:g} (logits within +-25; the default bench network: random buffers, gain 8)")
    return rep


def extras(a, sd, x, mask, device):
    """Side measurements (not `value`): (1) the timed step replayed from a hipGraph; (2) the end-to-end `test_sample`
    equivalent of SceneFlow/test_sceneflow_ddim.py:89-122 -- origin ACVNet -> used/disp -> ACVNet_DDIM.forward ->
    metrics, all on the HIP kernels -- with its per-stage times; (3) BASELINE configs 4 and 5 on one GPU.
    (The opt-in split-fp16 convolution path, DV_CONV_PRECISION=f16x3, is no longer measured here: it is not the
    arithmetic the contract names and gains 1-2 % at this point; tools/bench_flavours.py still times it.)"""
    import diffuvolume_amd as dv
    from diffuvolume_amd import metrics as M
    from diffuvolume_amd import submodule as S
    from diffuvolume_amd.synth import synth_state_dict
    res = {}

    def timed_loop(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        # (0) the same timed step replayed from a hipGraph: the whole hot path of a batch (builders, 5 DDIM steps with
        # their device-generator draws, metric sums) captured once on a side stream, inputs static in HBM
        try:
            model = dv.ACVNet_DDIM(192, False, False, sampling_timesteps=a.ddim_steps,
                                   ensemble_cof=None if a.ddim_steps == 5 else tuple([0.5] + [0.0] * (a.ddim_steps - 1) + [0.5]))
            model.load_state_dict(sd, strict=True)
            model = model.to(device).eval()
            model.prepare()

            def one():
                return M.image_sums(hot_path(model, x)[0], x["gt"], mask)

            eager = timed_loop(one, a.steps)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one()                                    # warm the allocator pools on the capture stream
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                sums = one()
            graphed = timed_loop(graph.replay, a.steps)
            res["hipgraph"] = {"eager_ms_per_step": 1e3 * eager, "graph_ms_per_step": 1e3 * graphed,
                               "value": a.batch / graphed, "unit": "pairs/s", "finite": bool(torch.isfinite(sums).all()),
                               "note": "one hipGraph launch per batch: ~330 kernel launches (HIP kernels through the C "
                                       "ABI on the capture stream + the device-generator draws); eager is already "
                                       "GPU-bound at batch 8, so the gain is the launch gaps only"}
            del graph, model
        except Exception as e:                           # noqa: BLE001 -- a side measurement must not sink the bench line
            res["hipgraph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # end to end
        from diffuvolume_amd.synth import _gen
        g = _gen(7, "e2e")
        left = torch.randn(a.batch, 3, a.height, a.width, generator=g).to(device)
        right = torch.roll(left, -8, dims=-1)
        origin = dv.ACVNet(192, False, False)
        origin.load_state_dict(synth_state_dict(origin.state_dict(), seed=3, logit_gain=8.0), strict=True)
        origin = origin.to(device).eval()
        ddim = dv.ACVNet_DDIM(192, False, False)
        ddim.load_state_dict(sd, strict=True)
        ddim = ddim.to(device).eval()

        def test_sample():
            used = origin(left, right)[-1]
            dn = torch.clamp(used, 0, 191).unsqueeze(1)
            dn = torch.nn.functional.interpolate(dn, size=(a.height // 4, a.width // 4), mode="bilinear") / 4
            pred = ddim(left, right, used, dn, None)[0]
            return M.batch_metrics(pred, x["gt"], mask)

        dt = timed_loop(test_sample, 2)
        res["end_to_end_test_sample"] = {"value": a.batch / dt, "unit": "pairs/s", "ms_per_batch": 1e3 * dt,
                                         "note": "origin ACVNet + feature CNNs + attention branch + hot path + metrics, all on the HIP kernels"}
        # the same at batch 1: per-pair LATENCY, the way the reference evaluates (test_sceneflow_ddim.py:39,:48,:73-82) and
        # the only kind of number it publishes
        left, right = left[:1].contiguous(), right[:1].contiguous()
        gt1, mask1 = x["gt"][:1].contiguous(), mask[:1].contiguous()

        def test_sample_b1():
            used = origin(left, right)[-1]
            dn = torch.clamp(used, 0, 191).unsqueeze(1)
            dn = torch.nn.functional.interpolate(dn, size=(a.height // 4, a.width // 4), mode="bilinear") / 4
            pred = ddim(left, right, used, dn, None)[0]
            return M.batch_metrics(pred, gt1, mask1)

        dt1 = timed_loop(test_sample_b1, 3)
        res["latency_b1_test_sample"] = {
            "ms_per_pair": 1e3 * dt1, "batch": 1, "frame": f"{a.width}x{a.height}",
            "what": "origin ACVNet + ACVNet_DDIM.forward (features, attention branch, 5 DDIM steps) + metrics, one pair",
            "reference_readme_s_per_pair": 1.11,
            "reference_note": "README.md:108 (SceneFlow table, Runtime) quotes 1.11 s per pair for DiffuVolume on unspecified hardware: context, "
                              "not a baseline (never vs_baseline)"}
        del origin, ddim
    # (3) BASELINE configs 4 and 5 on one GPU (parity cases, not the headline): hot-path speed, dominant kernel and its
    # issued fraction of the fp32 matrix pipe -- a few seconds each
    torch.cuda.empty_cache()
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import bench_flavours as BF

        def dominant(ks):
            tag = max(ks, key=lambda k: ks[k][1])
            n, ms, algo, issued = ks[tag]
            return {"tag": tag, "launches": n, "ms": ms, "share_of_kernel_time": ms / sum(v[1] for v in ks.values()),
                    "algorithmic_tflops": algo, "issued_tflops": issued, "issued_frac_of_mfma_f32_peak": issued / PEAK_MFMA_F32_TFLOPS}

        r4 = BF.pcw()
        res["config4"] = {"workload": r4["config"], "value": r4["pairs_per_s_hot"], "unit": "pairs/s (hot path: fused volume + "
                          "3-step ddim_sample incl. the 2-D refinement)", "forward_pairs_per_s": r4["pairs_per_s_forward"],
                          "ddim_sample_ms": r4["ddim_sample_ms"], "dominant_kernel": dominant(r4["ddim_sample_kernels_ms"])}
        torch.cuda.empty_cache()
        r5 = BF.igev_model(quick=True)
        res["config5"] = {"workload": r5["config"], "value": r5["pairs_per_s"], "unit": "pairs/s (whole IGEVStereo_ddim "
                          "forward, batch 4 on one GPU)", "forward_ms": r5["forward_ms"],
                          "ms_per_gru_iteration": r5["ms_per_gru_iteration"],
                          "dominant_kernel": dominant(r5["kernels_of_a_2_iteration_pass"])}
        torch.cuda.empty_cache()
        rd = BF.igev_reference_default(b=1)
        res["latency_b1_kitti15_reference_default"] = {
            "ms_per_pair": rd["ms_per_pair"], "batch": 1, "workload": rd["config"],
            "reference_readme_s_per_pair": 0.18,
            "reference_note": "README.md:98 (KITTI 2015 leaderboard, Runtime) quotes 0.18 s per pair for IGEV+DiffuVolume on unspecified hardware; the reference "
                              "hard-codes 2 DDIM steps (core/igev_stereo_ddim.py:124), BASELINE config 5 names 20: context, "
                              "not a baseline"}
    except Exception as e:                               # noqa: BLE001 -- a side measurement must not sink the bench line
        res["config4_5_error"] = f"{type(e).__name__}: {e}"[:300]
    return res


def flavour_workload(a, rank, device):
    """BASELINE configs 4 / 5 as N-rank workloads: (model, step() -> (prediction [B,H,W], gt [B,H,W]), description).
    Inputs are synthetic, per-rank (seed 100 + rank), resident in HBM before the timed region; random-init weights."""
    import types
    import torch.nn.functional as F
    from diffuvolume_amd.synth import _gen, synth_state_dict, synth_stereo_batch
    b, h, w = a.batch, a.height, a.width
    if a.workload == "kitti12":
        from diffuvolume_amd.pwcnet_ddim import PWCNet_ddim
        cof = None if a.ddim_steps == 3 else tuple([0.9] + [0.0] * (a.ddim_steps - 1) + [0.1])
        m = PWCNet_ddim(192, True, sampling_timesteps=a.ddim_steps, ensemble_cof=cof)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=2, logit_gain=8.0, scale={"refinenet3.conv8.weight": 0.002}))
        m = m.to(device).eval()
        batch = {k: v.to(device) for k, v in synth_stereo_batch(b, h, w, seed=100 + rank).items()}
        with torch.no_grad():                      # the hot path's inputs: the two feature pyramids (like fl / fr of config 2)
            fl = m.feature_extraction(batch["left"] * 0.05)
            fr = m.feature_extraction(batch["right"] * 0.05)

        def step():
            combine = m.fused_volume(fl, fr)
            pred, _ = m.ddim_sample(combine, batch["used"], m.encode_disparity(batch["disp"]), fl, fr)
            return pred, batch["gt"]

        what = (f"KITTI12 PCWNet+DiffuVolume hot path (fused gwc+concat volume -> hourglassup -> {a.ddim_steps} DDIM steps "
                f"incl. the 2-D refinement + EPE), {w}x{h}, maxdisp=192, batch={b}/GPU, random-init weights")
        return m, step, what
    from diffuvolume_amd.igev_stereo_ddim import Feature, IGEVStereo_ddim
    from diffuvolume_amd.synth import StubMobileNetV2
    args = types.SimpleNamespace(hidden_dims=[128, 128, 128], n_gru_layers=3, n_downsample=2, corr_levels=2, corr_radius=4,
                                 slow_fast_gru=False, max_disp=192, mixed_precision=False)
    cof = [0.5] + [0.0] * (a.ddim_steps - 1) + [0.5] if a.ddim_steps != 2 else None
    m = IGEVStereo_ddim(args, feature=Feature(StubMobileNetV2()), sampling_timesteps=a.ddim_steps, ensemble_cof=cof)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=7, scale={"update_block.disp_head.conv2.weight": 0.05,
                                                                      "update_block.disp_head.conv2.bias": 0.0,
                                                                      "classifier.weight": 20.0}), strict=True)
    m = m.to(device).eval()
    g = _gen(177 + rank, "cfg5")
    img1 = (torch.rand(b, 3, h, w, generator=g) * 255).to(device)
    img2 = torch.roll(img1, -9, dims=-1)
    flow_full = (9 + torch.randn(b, 1, h, w, generator=g)).clamp(0.5, 47).to(device)
    flow_gt = F.interpolate(flow_full, size=(h // 4, w // 4), mode="bilinear") / 4
    gt = flow_full[:, 0].contiguous()

    def step():
        pred, _ = m(img1, img2, flow_full, flow_gt, iters=a.gru_iters, test_mode=True)
        return pred.reshape(b, h, w), gt

    what = (f"KITTI15 IGEV-Stereo+DiffuVolume, whole IGEVStereo_ddim forward (stub MobileNetV2 backbone: timm's pretrained "
            f"one does not exist offline), {w}x{h}, {a.ddim_steps} DDIM steps x {a.gru_iters} GRU iterations, "
            f"batch={b}/GPU, random-init weights")
    return m, step, what


def main_flavour(a, rank, world, device):
    """`--workload kitti12 | kitti15`: the same contract line as the headline for BASELINE configs 4 / 5 -- one process per
    GPU, per-rank batches (weak scaling), W warm-up steps, K timed steps between barriers + synchronize, MAX over ranks,
    the metric table all-reduced once.  `roofline` = the workload's dominant kernel from a second, HIP-event-timed pass."""
    from diffuvolume_amd import distributed as D
    from diffuvolume_amd import metrics as M
    from diffuvolume_amd.profiling import KernelTimer
    model, step_fn, what = flavour_workload(a, rank, device)
    acc = M.MetricAccumulator(device)

    def step():
        pred, gt = step_fn()
        acc.update_sums(M.image_sums(pred, gt, (gt < 192) & (gt > 0)))

    def timed_region(timer):
        KernelTimer.active = timer
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        e = acc.reduce()
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        d = time.perf_counter() - t0
        KernelTimer.active = None
        return D.barrier_and_max(d, device), e

    with torch.no_grad():
        for _ in range(a.warmup):
            step()
        acc = M.MetricAccumulator(device)
        dt, epe = timed_region(None)
        timer = None
        if not a.no_kernel_timer:
            acc = M.MetricAccumulator(device)
            timer = KernelTimer()
            # the per-kernel pass runs SERIAL: one stream (the update block's side stream would make the HIP-event times of
            # the encoder and the gru16 / gru08 kernels overlap each other) and eager launches (timing events cannot be
            # recorded inside a stream capture).  `value` above was measured with the defaults.
            restore = None
            if a.workload == "kitti15":
                from diffuvolume_amd import update as U
                from diffuvolume_amd import igev_stereo_ddim as I
                restore = (U.BasicMultiUpdateBlock.OVERLAP, I.IGEVDiffusionLoop.use_graph)
                U.BasicMultiUpdateBlock.OVERLAP, I.IGEVDiffusionLoop.use_graph = False, False
            try:
                timed_region(timer)
            finally:
                if restore is not None:
                    U.BasicMultiUpdateBlock.OVERLAP, I.IGEVDiffusionLoop.use_graph = restore
    value = a.batch * world * a.steps / dt
    out = {"metric": WORKLOADS[a.workload]["metric"], "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": what, "global_batch": a.batch * world, "ddim_steps": a.ddim_steps,
                      "parallelism": f"dp{world}"},
           "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
           "launcher": launcher_label(), "env_overrides": env_overrides(), "epe_px": epe["EPE"],
           "epe_note": "random-init weights and synthetic pairs: the number only shows the metric path runs",
           "cpu_baseline": None,
           "cpu_baseline_note": "timed for the headline workload only (python bench.py); the oracles of configs 4 / 5 are "
                                "the checkers of tests/, minutes per pair at these sizes"}
    if rank == 0 and timer is not None:
        ks = timer.summary()
        mf = {k: v for k, v in ks.items() if v["issued_flops"] > 0}
        if mf:
            tag = max(mf, key=lambda k: mf[k]["total_ms"])
            v = mf[tag]
            issued = v["issued_flops"] / v["total_ms"] / 1e9
            out["roofline"] = {"bound": "mfma", "achieved": issued, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                               "frac": issued / PEAK_MFMA_F32_TFLOPS, "traffic": None, "kernel": tag,
                               "launches": v["launches"], "avg_ms": v["total_ms"] / v["launches"],
                               "algorithmic_tflops": v["flops"] / v["total_ms"] / 1e9,
                               "share_of_kernel_time": v["total_ms"] / sum(x["total_ms"] for x in ks.values()),
                               "note": "dominant matrix-pipe kernel of this workload by HIP-event time (KernelTimer tag); "
                                       "issued flops / time over the fp32 MFMA peak"}
        out["kernels_ms_per_step"] = {k: round(v["total_ms"] / a.steps, 3) for k, v in sorted(ks.items())}
        out["kernels_issued_frac_of_mfma_f32_peak"] = {k: round(v["issued_flops"] / v["total_ms"] / 1e9 / PEAK_MFMA_F32_TFLOPS, 3)
                                                       for k, v in sorted(mf.items()) if v["total_ms"] > 0}
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    if rank == 0:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # ONE notion of "under a launcher" everywhere (RANK + LOCAL_RANK + WORLD_SIZE + MASTER_PORT): a bare WORLD_SIZE exported
    # by a scheduler neither suppresses the self-launch nor starts a process group
    if a.gpus > 1 and not under_launcher():
        # no launcher: be the launcher (nothing has touched the GPU yet; children are started, never exec'd into)
        raise SystemExit(launch_workers(a, sys.argv[1:]))
    env_world = int(os.environ["WORLD_SIZE"]) if under_launcher() else 1
    if env_world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher environment has WORLD_SIZE={env_world}: refusing to report a "
                         "different GPU count")
    if a.dry_run:
        raise SystemExit(dry_run(a))
    from diffuvolume_amd import distributed as D
    # under a launcher (WORLD_SIZE set, even to 1) the process group is always initialised: `torchrun --nproc-per-node 1`
    # goes through RCCL exactly like N = 8
    if under_launcher():
        rank, world, local = D.init_from_env(force=True)
    else:
        rank, world, local = 0, 1, 0
    assert world == a.gpus
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    torch.cuda.set_device(D.device_index(local))
    device = torch.device("cuda", D.device_index(local))
    if a.workload != "sceneflow":
        return main_flavour(a, rank, world, device)

    import diffuvolume_amd as dv
    from diffuvolume_amd import metrics as M
    from diffuvolume_amd.profiling import KernelTimer
    from diffuvolume_amd.synth import NoiseTape, synth_state_dict

    h, w = a.height // 4, a.width // 4
    cof = None if a.ddim_steps == 5 else tuple([0.5] + [0.0] * (a.ddim_steps - 1) + [0.5])
    model = dv.ACVNet_DDIM(192, False, False, sampling_timesteps=a.ddim_steps, ensemble_cof=cof)
    sd = synth_state_dict(model.state_dict(), seed=1, logit_gain=8.0)
    model.load_state_dict(sd, strict=True)
    model = model.to(device).eval()
    model.prepare()
    host, x = make_inputs(a.batch, h, w, seed=100 + rank, device=device)
    mask = (x["gt"] < 192) & (x["gt"] > 0)
    acc = M.MetricAccumulator(device)

    def step(tape=None):
        final, stack, _ = hot_path(model, x, tape)
        acc.update_sums(M.image_sums(final, x["gt"], mask))      # this rank's shard of the step's global batch
        return final, stack

    def timed_region(timer):
        """K steps between barriers + synchronize; returns the MAX over ranks of the wall time."""
        KernelTimer.active = timer
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        e = acc.reduce()                 # the one collective of the path (48 bytes per batch, SUM over RCCL)
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        d = time.perf_counter() - t0
        KernelTimer.active = None
        return D.barrier_and_max(d, device), e

    with torch.no_grad():
        for _ in range(a.warmup):
            step()
        # pass 1: `value` -- nothing but the hot path in the timed region (no per-kernel events)
        acc = M.MetricAccumulator(device)
        dt, epe = timed_region(None)
        # pass 2 (rank 0's figures are the ones reported): the same K steps again with a HIP-event pair around every
        # kernel for the roofline / per-kernel numbers; its wall time is reported beside `value`, never as `value`
        timer, dt_events = None, None
        if not a.no_kernel_timer:
            acc = M.MetricAccumulator(device)
            timer = KernelTimer()
            dt_events, _ = timed_region(timer)

    pairs = a.batch * world * a.steps
    value = pairs / dt
    rccl_ranks = 1
    if torch.distributed.is_initialized():     # read the group size back through the collective itself
        ones = torch.ones(1, device="cpu" if torch.distributed.get_backend() == "gloo" else device)
        torch.distributed.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == a.gpus, (rccl_ranks, a.gpus)
    out = {
        "metric": WORKLOADS["sceneflow"]["metric"],
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SceneFlow ACVNet+DiffuVolume hot path (gwc + concat*softmax(att) + "
                               f"{a.ddim_steps} DDIM steps + EPE), {a.width}x{a.height}, maxdisp=192, "
                               f"batch={a.batch}/GPU, random-init weights",
                   "global_batch": a.batch * world, "ddim_steps": a.ddim_steps, "parallelism": f"dp{world}",
                   "attention_concat_volume": "factors only (softmax(att) + the two feature maps; its one consumer, "
                                              "dres0[0], runs on them): the [B,64,48,h,w] tensor is not materialised"},
        "rccl_ranks": rccl_ranks,
        "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
        "launcher": launcher_label(),
        "env_overrides": env_overrides(),
        "epe_px": epe["EPE"],
        "epe_note": "random-init weights and synthetic pairs: the number only shows the metric path runs; dataset EPE "
                    "(0.46 px, README) is unpinned -- no checkpoint or data ship with the reference",
        "timing": {"value_pass": "K steps, no per-kernel events in the timed region",
                   "ms_per_step_with_kernel_events": None if dt_events is None else 1e3 * dt_events / a.steps},
        "hbm_roofline_frac_whole_path": value / world * ALGO_BYTES_PER_PAIR / (PEAK_HBM_GBS * 1e9),
        # SURVEY 8(d)'s direct-convolution flop count / time / peak: NOT a hardware fraction (the Winograd layers issue
        # 2.25x fewer multiplies); the hardware figure is `mfma_f32_issued_frac_whole_path` below
        "algorithmic_flop_rate_over_mfma_f32_peak_whole_path": value / world * ALGO_FLOP_PER_PAIR / (PEAK_MFMA_F32_TFLOPS * 1e12),
    }
    if rank == 0 and timer is not None:
        ks = timer.summary()
        # dominant kernel: `avg_ms` is the average rocprofv3 --stats reports for that kernel name in the same command
        fam = {k: v for k, v in ks.items() if k in DOMINANT_TAGS}
        allw = {k: v for k, v in ks.items() if k in ("conv3d_k3s1_co32", "conv3d_k3s1_co64", "conv3d_k3s1_co128")}
        if fam:
            flops = sum(v["flops"] for v in fam.values())
            ms = sum(v["total_ms"] for v in fam.values())
            launches = sum(v["launches"] for v in fam.values())
            algo = flops / (ms * 1e-3) / 1e12
            issued = algo / WINO_MULT_REDUCTION
            algo_bytes = sum(v["bytes"] for v in fam.values()) / launches
            traffic = pmc_traffic(DOMINANT_KERNEL)
            out["traffic_profile"] = pmc_traffic_stamp()
            # `achieved` / `frac` price the matrix pipe with the flops the kernel ISSUES: the Winograd F(2x2,3x3) form
            # executes 2.25x fewer multiplies than the direct-convolution (algorithmic, SURVEY 8d: 2*27*Cin*Cout per
            # output voxel) count, which is reported beside it as `algorithmic_tflops`.
            out["roofline"] = {"bound": "mfma", "achieved": issued, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                               "frac": issued / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
                               "algorithmic_tflops": algo, "multiply_reduction": WINO_MULT_REDUCTION,
                               "algorithmic_bytes_per_launch": algo_bytes,
                               "traffic_over_algorithmic": None if traffic is None else traffic / algo_bytes,
                               "kernel": DOMINANT_KERNEL + " (the 32->32 3x3x3 stride-1 layers of dres0/dres1/hourglass/"
                                         "classif2 at 48x128x240; Winograd F(2x2,3x3) in-plane, depth taps direct, "
                                         "v_mfma_f32_16x16x4_f32)",
                               "launches": launches, "avg_ms": ms / launches,
                               "algorithmic_gflop_per_launch": flops / launches / 1e9,
                               # every instantiation of the template without the filter prologue (32 / 64 / 128 channels)
                               "all_shapes_issued_frac": sum(v["flops"] for v in allw.values())
                                                         / sum(v["total_ms"] for v in allw.values()) / 1e9
                                                         / WINO_MULT_REDUCTION / PEAK_MFMA_F32_TFLOPS}
        side = []
        for tags, kname, div in SIDE_KERNELS:
            sel = [ks[t] for t in tags if t in ks]
            if not sel:
                continue
            fl, ms = sum(v["flops"] for v in sel), sum(v["total_ms"] for v in sel)
            n = sum(v["launches"] for v in sel)
            ab = sum(v["bytes"] for v in sel) / n
            tr = pmc_traffic(kname)
            side.append({"kernel": kname, "tags": list(tags), "bound": "mfma", "launches": n, "avg_ms": ms / n,
                         "ms_per_step": ms / a.steps, "achieved": fl / ms / 1e9 / div, "unit": "TFLOP/s",
                         "peak": PEAK_MFMA_F32_TFLOPS, "frac": fl / ms / 1e9 / div / PEAK_MFMA_F32_TFLOPS,
                         "algorithmic_tflops": fl / ms / 1e9, "multiply_reduction": div,
                         "algorithmic_bytes_per_launch": ab, "traffic": tr,
                         "traffic_over_algorithmic": None if tr is None else tr / ab})
        # the first layer of every step on the factors of its input (csrc/rank1_filter.hip): vector-ALU / LDS bound
        r1 = ks.get("conv3d_k3s1_co32_filter_rank1")
        if r1:
            side.append({"kernel": "rank1_filter_split_kernel", "tags": ["conv3d_k3s1_co32_filter_rank1"], "bound": "valu",
                         "launches": r1["launches"], "avg_ms": r1["total_ms"] / r1["launches"],
                         "ms_per_step": r1["total_ms"] / a.steps, "achieved": r1["valu_flops"] / r1["total_ms"] / 1e9,
                         "unit": "TFLOP/s", "peak": PEAK_MFMA_F32_TFLOPS, "frac": r1["valu_flops"] / r1["total_ms"] / 1e9 / PEAK_MFMA_F32_TFLOPS,
                         "algorithmic_tflops": r1["flops"] / r1["total_ms"] / 1e9,
                         "note": "dres0[0] of acv_ddim.py:200-203 on s * [L ; R(x-d)]: 108 vector flops per output instead of "
                                 "3456 matrix flops; algorithmic_tflops counts the layer it replaces (SURVEY 8d), peak = "
                                 "packed-fp32 vector rate (the same pipe as the fp32 MFMA)",
                         "traffic": pmc_traffic("rank1_filter_split_kernel")})
        out["roofline_kernels"] = side
        # whole path against the matrix pipe: flops the kernels ISSUE (Winograd layers: algorithmic / 2.25; vector-ALU
        # kernels: 0) over the wall time of the `value` pass
        issued_per_step = sum(v["issued_flops"] for v in ks.values()) / a.steps
        out["mfma_f32_issued_frac_whole_path"] = issued_per_step / (dt / a.steps) / (PEAK_MFMA_F32_TFLOPS * 1e12)
        out["issued_tflop_per_step"] = issued_per_step / 1e12
        out["kernels_ms_per_step"] = {k: round(v["total_ms"] / a.steps, 3) for k, v in sorted(ks.items())}
        out["kernels_issued_tflops_or_gbs"] = {
            k: (round(v["issued_flops"] / v["total_ms"] / 1e9, 2) if v["issued_flops"] > 0
                else round(v["bytes"] / v["total_ms"] / 1e6, 1)) for k, v in sorted(ks.items())}
        out["kernels_algorithmic_tflops"] = {k: round(v["flops"] / v["total_ms"] / 1e9, 2) for k, v in sorted(ks.items())
                                             if v["issued_flops"] > 0}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base, ref = cpu_baseline(sd, host, a.ddim_steps, model.ensemble_cof)
        out["cpu_baseline"] = base
        out["parity_vs_oracle"] = parity_vs_oracle(model, x, ref)
        del ref
        if a.ddim_steps == 5:
            out["parity_vs_oracle_calibrated"] = parity_calibrated(a, host, x, device)
    if rank == 0 and world == 1 and not a.no_extras:
        out["extras"] = extras(a, sd, x, mask, device)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    if rank == 0:
        # RCCL prints its version banner with printf (this image exports NCCL_DEBUG=VERSION): flush the C stdio buffer
        # first so that the JSON line is the LAST line on stdout, whatever else the libraries wrote there
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
