"""BatchNorm buffers consistent with the data -- the "conditioned" synthetic networks of the parity tests.

TEST INFRASTRUCTURE ONLY (imported by tests/ and the oracle/make_golden_*conditioned.py generators).

Why.  `synth_state_dict` draws BatchNorm running statistics at random, so an untrained stack of 3-D hourglasses lets
the activation scale drift by orders of magnitude (KITTI12 at 1248x384: classifier logits up to +-2600).  The last
bits of an fp32 logit of that size are 3e-4 wide, and a soft-argmax moves by `uncertainty x |d cost|`: two CORRECT fp32
evaluations of such a network differ by more than 1e-3 px on 2-6 % of the pixels (measured: fp32 oracle vs float64
oracle), so the contract's bar cannot be asserted on it.  A trained checkpoint never looks like that: its BatchNorm
buffers hold the statistics of the data the layer actually sees, and every layer's output is O(1).  This module puts a
synthetic network into that state: one forward pass of the ORACLE in which every BatchNorm writes the batch statistics
of its input into its `running_mean` / `running_var` before normalising with them (what `momentum = 1` training-mode
BatchNorm leaves behind).  The statistics are stored with the golden fixtures (a few thousand floats), so the GPU box
needs no calibration pass.  Nothing else about the weights changes; the tests state the classifier gain they use.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, Iterable

import torch
import torch.nn.functional as F
from torch import nn

from . import acv_oracle as A
from . import pcw_oracle as P


def _calibrating_bn(x, sd, p):
    dims = [0] + list(range(2, x.dim()))
    sd[p + ".running_mean"] = x.mean(dims).to(sd[p + ".running_mean"].dtype)
    sd[p + ".running_var"] = x.var(dims, unbiased=False).clamp(min=1e-6).to(sd[p + ".running_var"].dtype)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


@contextmanager
def calibrating_bn():
    """Inside the block every BatchNorm of the functional oracles (acv_oracle._bn, pcw_oracle._bn2) overwrites its
    buffers in the state dict it is given with the statistics of its input."""
    a_bn, p_bn = A._bn, P._bn2
    A._bn, P._bn2 = _calibrating_bn, _calibrating_bn
    try:
        yield
    finally:
        A._bn, P._bn2 = a_bn, p_bn


@torch.no_grad()
def calibrate_modules(module: nn.Module, run) -> None:
    """The same for a PyTorch module graph (the 2-D feature CNNs): BatchNorm in training mode with momentum 1 for one
    call of ``run()``; the buffers then hold the batch statistics of that call, layer by layer."""
    bns = [m for m in module.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    saved = [(m.momentum, m.training) for m in bns]
    for m in bns:
        m.momentum, m.training = 1.0, True
    try:
        run()
    finally:
        for m, (mom, tr) in zip(bns, saved):
            m.momentum, m.training = mom, tr


def bn_buffers(sd: Dict[str, torch.Tensor], prefixes: Iterable[str] = ("",)) -> Dict[str, torch.Tensor]:
    """The running_mean / running_var entries of a state dict (optionally only under some prefixes)."""
    pre = tuple(prefixes)
    return {k: v for k, v in sd.items() if k.rsplit(".", 1)[-1] in ("running_mean", "running_var") and k.startswith(pre)}


def pack(buffers: Dict[str, torch.Tensor]):
    """dict -> (keys array, flat float64 values, lengths) for an .npz fixture."""
    import numpy as np
    keys = sorted(buffers)
    return (np.array(keys), np.concatenate([buffers[k].double().numpy().ravel() for k in keys]),
            np.array([buffers[k].numel() for k in keys], dtype=np.int64))


def unpack(keys, values, lengths) -> Dict[str, torch.Tensor]:
    out, o = {}, 0
    values = torch.as_tensor(values)
    for k, n in zip([str(k) for k in keys.tolist()], torch.as_tensor(lengths).tolist()):
        out[k] = values[o:o + n].clone().float()
        o += n
    return out
