"""Worker of tests/test_gpu_shard.py: one rank of the multi-GPU path on whatever GPU `device_index` maps it to (on a
1-GPU box both ranks share cuda:0), gloo rendezvous.  Runs the HIP hot path on its contiguous shard of a batch of 8 and
all-reduces the metric sums exactly like bench.py.  Prints one JSON line."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def shard_epe(model, x, gt, lo, hi, dev, acc, batch=8, seed=5):
    """Hot path on pairs [lo, hi) with the draws those pairs get inside the full batch; updates `acc`."""
    import diffuvolume_amd as dv
    from diffuvolume_amd import metrics as M
    from diffuvolume_amd.synth import NoiseTape
    tape = NoiseTape(seed)

    def draw(kind, shape, dtype):
        return tape(kind, (batch,) + tuple(shape[1:]), dtype)[lo:hi]

    s = {k: v[lo:hi].to(dev) for k, v in x.items()}
    with torch.no_grad():
        gwc = dv.build_gwc_volume(s["fl"], s["fr"], 48, 40)
        vol = dv.build_concat_attention_volume(s["cl"], s["cr"], s["att"], 48)
        final, _ = model.ddim_sample(vol, s["used"], model.encode_disparity(s["dq"]), noise=draw)
        g = gt[lo:hi].to(dev)
        sums = M.image_sums(final, g, (g < 192) & (g > 0))
        acc.update_sums(sums)                              # this rank's shard of the one global batch
    return final, float(gwc.double().sum()), sums


def build(dev):
    import diffuvolume_amd as dv
    from diffuvolume_amd.synth import synth_hot_inputs, synth_state_dict
    model = dv.ACVNet_DDIM(192, False, False)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1, logit_gain=8.0), strict=True)
    model = model.to(dev).eval()
    x = synth_hot_inputs(8, 16, 40, seed=100)
    return model, x, x["gt"]


def main():
    from diffuvolume_amd import distributed as D
    from diffuvolume_amd import metrics as M
    rank, world, local = D.init_from_env(backend="gloo")
    dev = torch.device("cuda", D.device_index(local))
    torch.cuda.set_device(dev)
    model, x, gt = build(dev)
    lo, hi = D.shard_range(8, rank, world)
    acc = M.MetricAccumulator(dev)
    final, gsum, _ = shard_epe(model, x, gt, lo, hi, dev, acc)
    out = acc.reduce()                                  # the one collective of the path
    print(json.dumps({"rank": rank, "lo": lo, "hi": hi, "metrics": out, "final_sum": float(final.double().sum()),
                      "final_hex": final.double().sum().item().hex(), "gwc_sum": gsum}), flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
