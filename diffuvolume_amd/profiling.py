"""HIP-event timing of individual kernel launches on the stream they run on (bench.py's
`roofline` figures).  Disabled unless ``KernelTimer.active`` is set: the hot path pays nothing."""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Dict, Optional

import torch


class KernelTimer:
    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.records = defaultdict(list)     # tag -> [(start, end, flops, bytes)]

    def launch(self, tag: str, flops: float, nbytes: float, fn: Callable[[], None]) -> None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records[tag].append((s, e, flops, nbytes))

    def summary(self) -> Dict[str, dict]:
        torch.cuda.synchronize()
        out = {}
        for tag, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            out[tag] = {"launches": len(recs), "total_ms": ms, "avg_ms": ms / len(recs),
                        "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs)}
        return out


def timed(tag: str, flops: float, nbytes: float, fn: Callable[[], None]) -> None:
    t = KernelTimer.active
    if t is None:
        fn()
    else:
        t.launch(tag, flops, nbytes, fn)
