"""The ONE list of environment switches this package reads (INTEGRATION.md section 5 is generated from it).

Nothing here changes results silently on the default path: every switch is either an opt-in alternative that gives the same
bits (side stream, hipGraph replay), a documented numerics alternative (`DV_CONV_PRECISION`), or plumbing.  `overrides()`
returns the ones that are set; bench.py prints that dict in its JSON line as `env_overrides`, so a measured number carries
the switches it was measured under.  The native library reads NO environment variable (tests pin kernel choices through
`dv_*_set_*` hooks of the C ABI)."""
from __future__ import annotations

import os

KNOBS = {
    # name: (default, where it is read, what it does)
    "DV_LIB_PATH": ("<package>/libdiffuvolume_hip.so", "_lib.py at import",
                    "load another build of the same C ABI (A/B of kernel variants, tools/build_variant.sh)"),
    "DV_CONV_PRECISION": ("f32", "submodule.default_conv_precision() when a plan is built",
                          "f32 = Winograd / polyphase fp32 MFMA kernels; f32_direct = direct implicit GEMM everywhere; "
                          "f16x3 = split-fp16 products (opt-in, not the contract's arithmetic)"),
    "DV_S2PP": ("1", "submodule.Conv3dPlan when a stride-2 plan is built",
                "0 = the stride-2 3-D layers on the direct kernel instead of the polyphase one (tests)"),
    "DV_IGEV_OVERLAP": ("1", "update.BasicMultiUpdateBlock.OVERLAP at import",
                        "0 = the motion encoder on the main stream instead of a side stream (same bits)"),
    "DV_IGEV_GRAPH": ("0", "igev_stereo_ddim.IGEVDiffusionLoop.use_graph at import",
                      "1 = replay the GRU iterations of a DDIM step as a hipGraph (same bits, no gain measured)"),
    "DV_DIST_BACKEND": ("nccl on GPUs, gloo on CPUs", "distributed.init_from_env()", "torch.distributed backend of the metric reduce"),
    "DV_BENCH_OVERSUBSCRIBE": ("unset", "bench.py", "1 = let N ranks share fewer GPUs (plumbing test of the N-rank path only)"),
    "DV_BENCH_SELF_LAUNCHED": ("unset", "bench.py (set by bench.py for the ranks it starts itself)", "internal marker"),
    "DV_FULL_PARITY": ("unset", "tests/", "1 = the long parity runs (all steps against float64, second pair, diagnostic networks)"),
}


def overrides() -> dict:
    """The switches of `KNOBS` that are set in this process's environment, with their values."""
    return {k: os.environ[k] for k in KNOBS if os.environ.get(k) not in (None, "")}
