"""Probe: the SceneFlow hot path of a batch of 8 as two half-batches on two HIP streams against one pass on one stream.
Every launch pays ~20 us of ramp-up / tail (tools/wino2d_grid_sweep.py) and a dependent chain cannot hide them; two
independent half-batches can run one's tail under the other's body.  python tools/two_stream_probe.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import diffuvolume_amd as dv
from diffuvolume_amd.synth import synth_state_dict

dev = torch.device("cuda:0")
model = dv.ACVNet_DDIM(192, False, False, sampling_timesteps=5)
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1, logit_gain=8.0), strict=True)
model = model.to(dev).eval()
model.prepare()
host, x = bench.make_inputs(8, 128, 240, seed=100, device=dev)
halves = [{k: v[i * 4:(i + 1) * 4].contiguous() for k, v in x.items()} for i in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def one():
    return bench.hot_path(model, x)[0]


def two():
    outs = []
    main = torch.cuda.current_stream(dev)
    for s, xh in zip(streams, halves):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(bench.hot_path(model, xh)[0])
    for s in streams:
        main.wait_stream(s)
    return torch.cat(outs)


def timeit(fn, n=10):
    with torch.no_grad():
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    print(f"one stream, batch 8: {timeit(one):7.2f} ms      two streams, 2 x batch 4: {timeit(two):7.2f} ms", flush=True)
