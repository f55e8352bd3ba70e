"""Fixed-seed subsets of the random-shape fuzzers (tools/fuzz_kernels.py, fuzz_round3.py .. fuzz_round6.py) inside the driver's
`-m gpu` run: conv2d (direct / Winograd / dilated / stride 2 / cat / K-split), conv3d (both stride-1 tilings, stride 2 with
and without the filter prologue and both tilings, single-channel head), transposed conv + redir, rank-1 layer incl. odd
Cout, table build, patch stencils, attention-concat volume, the 2-D Winograd source modes / gate pair / K-split, the fused
geometry lookup, `interp` -- each against float64 / MIOpen PyTorch statements on the GPU; round 6: the F(2x2x2,3x3x3) kernel
against the in-plane Winograd kernel and the persistent transposed convolution against the one-tile kernel (+ grid caps).
The scripts seed their generators themselves, so a case count selects a reproducible prefix of their sequence; the full
runs (40-80 cases) stay a tool."""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from diffuvolume_amd import submodule as S

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("script,cases", [("fuzz_kernels.py", 8), ("fuzz_round3.py", 8), ("fuzz_round4.py", 8), ("fuzz_round5.py", 12), ("fuzz_round6.py", 30)])
def test_fuzzer_prefix(script, cases):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / script), str(cases)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "FAIL" not in r.stdout and "BAD" not in r.stdout, tail


def test_stride2_polyphase_random_shapes():
    """csrc/conv3d_s2pp.hip on 16 random shapes (fixed seed): every width a multiple of 4 (the kernel's domain), odd depths /
    heights, channel tails, 64 / 128 / 192 output channels, with BatchNorm + residual + ReLU, against float64."""
    import random
    rnd = random.Random(505)
    g = torch.Generator().manual_seed(505)
    for i in range(16):
        cin, cout = rnd.choice([1, 3, 4, 8, 13, 32, 64]), rnd.choice([64, 128, 192])
        b, d, h, w = rnd.choice([1, 2, 3]), rnd.randint(1, 13), rnd.randint(1, 40), 4 * rnd.randint(1, 34)
        x = torch.randn(b, cin, d, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
        bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
              torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
        ref = torch.nn.functional.batch_norm(torch.nn.functional.conv3d(x.double(), wt.double(), None, 2, 1), bn[2].double(),
                                             bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5)
        res = torch.randn(ref.shape, generator=g)
        ref = torch.relu(ref + res.double())
        plan = S.Conv3dPlan(wt.cuda(), tuple(t.cuda() for t in bn), stride=2, act=S.ACT_RELU)
        assert plan.s2pp
        out = plan(x.cuda(), residual=res.cuda())
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max().clamp(min=1e-30))
        assert out.shape == ref.shape and err < 1e-5, (i, cin, cout, b, d, h, w, err)
