"""HIP-event timing of individual launches on torch's current stream (the
stream every dn_* call is enqueued on).  Used by bench.py for the roofline
numbers; inactive (zero overhead beyond one `is None` test) otherwise."""
import contextlib

import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.records = []          # (tag, kernel, flops, bytes, start_event, end_event)

    def summary(self):
        """-> {tag: dict(kernel, calls, ms_total, flops, bytes)} after a device sync."""
        torch.cuda.synchronize()
        out = {}
        for tag, kernel, flops, nbytes, e0, e1 in self.records:
            r = out.setdefault(tag, dict(kernel=kernel, calls=0, ms_total=0.0, flops=0.0, bytes=0.0))
            r["calls"] += 1
            r["ms_total"] += e0.elapsed_time(e1)
            r["flops"] += flops
            r["bytes"] += nbytes
        return out


@contextlib.contextmanager
def timing(timer):
    global _active
    prev, _active = _active, timer
    try:
        yield timer
    finally:
        _active = prev


@contextlib.contextmanager
def region(tag, kernel, flops=0.0, nbytes=0.0):
    if _active is None:
        yield
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    try:
        yield
    finally:
        e1.record()
        _active.records.append((tag, kernel, flops, nbytes, e0, e1))
