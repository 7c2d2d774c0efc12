"""HIP-event timing of individual launches on torch's current stream (the
stream every dn_* call is enqueued on).  Used by bench.py for the roofline
numbers; inactive (zero overhead beyond one `is None` test) otherwise."""
import contextlib

import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.records = []          # (tag, kernel, flops, bytes, start_event, end_event, executed MFMA flops or None)

    def summary(self):
        """-> {tag: dict(kernel, calls, ms_total, ms_median, flops, bytes)} after a
        device sync.  ms_total = calls x median launch duration: the median drops
        the rare launch whose event pair straddles a host-side stall (GC, allocator)
        instead of charging that stall to the kernel."""
        torch.cuda.synchronize()
        out, samples = {}, {}
        for tag, kernel, flops, nbytes, e0, e1, xflops in self.records:
            r = out.setdefault(tag, dict(kernel=kernel, calls=0, ms_total=0.0, ms_mean=0.0,
                                         flops=0.0, bytes=0.0, exec_flops=0.0, exec_known=True))
            if xflops is None:
                r["exec_known"] = False
            else:
                r["exec_flops"] += xflops
            r["calls"] += 1
            samples.setdefault(tag, []).append(e0.elapsed_time(e1))
            r["flops"] += flops
            r["bytes"] += nbytes
        for tag, r in out.items():
            xs = sorted(samples[tag])
            med = xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])
            r["ms_median"] = med
            r["ms_total"] = med * r["calls"]
            r["ms_mean"] = sum(xs) / len(xs)
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
def region(tag, kernel, flops=0.0, nbytes=0.0, exec_flops=None):
    """exec_flops: the MFMA work the launch EXECUTES for `flops` of algorithmic work (split-f16: 3 MFMAs per product, 2 on
    a hi-only operand; the tap-merged up-convs run 4 of 9 taps on the upsampled source's chunks).  None: the engine's
    uniform factor (bench.py :: EXECUTED_FLOP_FACTOR)."""
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
        _active.records.append((tag, kernel, flops, nbytes, e0, e1, exec_flops))
