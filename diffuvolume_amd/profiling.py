"""HIP-event timing of individual kernel launches on the stream they run on (bench.py's
`roofline` figures).  Disabled unless ``KernelTimer.active`` is set: the hot path pays nothing."""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Dict, Optional

import torch


class KernelTimer:
    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.records = defaultdict(list)     # tag -> [(start, end, flops, bytes, issued matrix-pipe flops)]

    def launch(self, tag: str, flops: float, nbytes: float, fn: Callable[[], None], issued: float = 0.0,
               valu: float = 0.0) -> None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records[tag].append((s, e, flops, nbytes, issued, valu))

    def summary(self) -> Dict[str, dict]:
        torch.cuda.synchronize()
        out = {}
        for tag, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[tag] = {"launches": len(recs), "total_ms": ms, "avg_ms": ms / len(recs),
                        "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs),
                        "issued_flops": sum(r[4] for r in recs), "valu_flops": sum(r[5] for r in recs)}
        return out


S2PP_MULT_REDUCTION = 1.44     # polyphase F(2,2) stride-2 form: 25 multiplies per 2x2 outputs and (depth tap, channel) instead of 36
WINO_MULT_REDUCTION = 2.25     # F(2x2,3x3): 16 multiplies per 2x2 outputs and (depth tap, channel) instead of 36
WINO3_MULT_REDUCTION = 3.375   # F(2x2x2,3x3x3): 64 multiplies per 2x2x2 outputs and channel instead of 216


def timed(tag: str, flops: float, nbytes: float, fn: Callable[[], None], issued: float = 0.0, valu: float = 0.0) -> None:
    """``flops`` = algorithmic (direct-convolution) count; ``issued`` = flops the kernel puts through the MATRIX pipe
    (0 for vector-ALU kernels; algorithmic / 2.25 for the Winograd forms); ``valu`` = flops a vector-ALU-bound kernel
    issues on the packed-fp32 vector pipe (same 157.3 TFLOP/s peak as the fp32 matrix instructions: one pipe)."""
    t = KernelTimer.active
    if t is None:
        fn()
    else:
        t.launch(tag, flops, nbytes, fn, issued, valu)
