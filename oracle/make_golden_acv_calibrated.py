"""BatchNorm statistics of the CALIBRATED SceneFlow network (oracle/calibrate.py) at the BASELINE size: the synthetic
ACVNet_DDIM weights (seed 1, classifier gain 1) with the BatchNorm buffers of the DDIM-loop layers set to the statistics
of pair 0 of the bench workload (960x512), pooled over the five DDIM steps of the oracle's own trajectory.  On this
network the fp32 oracle is within 1e-3 px of its float64 evaluation on every pixel at every step, so
tests/test_gpu_fullsize.py::test_fullsize_oracle_5step_calibrated asserts the contract's RAW bars on ALL pixels.
No reference run is involved (the oracle is pinned to the reference by tests/test_oracle_golden.py).

  python oracle/make_golden_acv_calibrated.py          ~3 min on 8 cores"""
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from diffuvolume_amd.synth import synth_hot_inputs, synth_state_dict  # noqa: E402
from oracle import acv_oracle as O  # noqa: E402
from oracle import calibrate as C  # noqa: E402
from oracle import loop_parity as LP  # noqa: E402

warnings.filterwarnings("ignore")
GAIN = 1.0
LOOP_PREFIXES = ("dres0.", "dres1.", "dres2.", "dres3.", "classif2.")


def main():
    from diffuvolume_amd.acv_ddim import ACVNet_DDIM
    torch.set_num_threads(8)
    t0 = time.time()
    sd = synth_state_dict(ACVNet_DDIM(192, False, False).state_dict(), seed=1, logit_gain=GAIN)
    x = synth_hot_inputs(1, 128, 240, seed=100)
    vol = O.attention_concat_volume(x["att"], O.build_concat_volume(x["cl"], x["cr"], 48))
    x_T = O.ACVDiffusionOracle(sd).encode_x_T(x["dq"])
    with C.calibrating_bn(), torch.no_grad():
        O.ACVDiffusionOracle(sd).model_predictions(vol, x_T, torch.full((1,), 999, dtype=torch.long))
    print(f"  step-1 calibration {time.time() - t0:.0f} s")
    _, _, trace = LP.oracle_trajectory(O.ACVDiffusionOracle(sd), vol, x["used"], x_T, 1)
    print(f"  trajectory {time.time() - t0:.0f} s; uncertainty mean per step {[round(float(r['unc'].mean()), 1) for r in trace]}")
    # pooled over the steps: running sums of the per-step batch statistics (one step at a time: a batch of five volumes
    # would need 10 GB) -- mean of means and mean of second moments
    acc = {}
    for r in trace:
        sdi = {k: v.clone() for k, v in sd.items()}
        with C.calibrating_bn(), torch.no_grad():
            O.ACVDiffusionOracle(sdi).model_predictions(vol, r["img"].float(), torch.full((1,), r["time"], dtype=torch.long))
        for k, v in C.bn_buffers(sdi, LOOP_PREFIXES).items():
            stem = k.rsplit(".", 1)[0]
            m, var = sdi[stem + ".running_mean"].double(), sdi[stem + ".running_var"].double()
            a = acc.setdefault(stem, [torch.zeros_like(m), torch.zeros_like(m)])
            if k.endswith("running_mean"):
                a[0] += m / len(trace)
                a[1] += (var + m * m) / len(trace)
    stats = {}
    for stem, (m, m2) in acc.items():
        stats[stem + ".running_mean"] = m.float()
        stats[stem + ".running_var"] = (m2 - m * m).clamp(min=1e-6).float()
    print(f"  pooled statistics {time.time() - t0:.0f} s  (per-step statistics taken with the step-1 buffers upstream)")
    keys, vals, lens = C.pack(stats)
    np.savez_compressed(REPO / "tests" / "golden" / "acv_calibrated_fullsize.npz", gain=GAIN, bn_keys=keys, bn_vals=vals,
                        bn_lens=lens)
    print("  acv_calibrated_fullsize.npz", (REPO / "tests" / "golden" / "acv_calibrated_fullsize.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
