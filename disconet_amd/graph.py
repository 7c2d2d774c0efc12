"""hipGraph capture of a whole forward step.

At batch 4 the step is ~30 short launches (2.6 ms on the GPU); replaying them
from one captured graph removes the per-launch host work (ctypes call, tensor
allocation, Python) from the critical path -- it matters most when 8 ranks
share one host's cores.  Every dn_* call is enqueued on torch's current stream
and allocates nothing itself, so torch.cuda.CUDAGraph (a hipGraph on ROCm)
captures the step as is; buffers come from the graph's private pool.

The captured step reads its inputs from static tensors and overwrites its
outputs in place on every replay: copy what you need to keep.
"""
import torch


class GraphedStep:
    def __init__(self, fn, warmup=3, range_guard=True):
        """fn: zero-argument callable that runs the step on the current stream and
        returns a tensor / dict / tuple of tensors (its inputs must be tensors that
        stay alive and are updated in place between replays).

        range_guard (default on): the split-f16 range guard INSIDE the captured step.  ops.check_sp_range skips itself
        under capture, so a replayed step -- the production mode -- would otherwise never look at the sticky clamp / NaN
        flags.  The capture ends with dn_sp_range_flags_async (three 1-thread launches OR-ing the flags into a device word that
        a kernel zeroes first; the flags stay sticky) and a copy of the word to pinned host memory; every __call__ looks at the
        copy of the PREVIOUS replay once its event has completed (an event query, no synchronisation) and raises DnError --
        the outputs of a flagged replay are invalid.  drain() waits for the outstanding replay and checks it.  ~3 us per
        replay; range_guard=False leaves the guard to the caller (ops.sp_range_flags() / drain_sp_range() between replays):
        bench.py times its step that way and reads the flags once, blocking, after the timed regions."""
        self.fn = fn
        self._guard = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the default stream: one-time inits
            for _ in range(warmup):        # (LDS attributes, weight packing) stay out of the graph
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if range_guard:
            word = torch.zeros(1, dtype=torch.int32, device="cuda")
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._guard = {"word": word, "host": host, "event": None}
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (e.g. the RCCL watchdog of an
        # initialised process group) may keep issuing HIP calls during the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = fn()
            if range_guard:
                from . import ops
                ops.sp_range_flags_into(self._guard["word"], zero_first=True)
                self._guard["host"].copy_(self._guard["word"], non_blocking=True)

    def _look(self, wait):
        g = self._guard
        if g is None or g["event"] is None:
            return
        if wait:
            g["event"].synchronize()
        elif not g["event"].query():
            return
        g["event"] = None
        flags = int(g["host"][0]) & 0xffffffff
        if flags & 7:
            from . import ops
            ops.sp_range_flags(reset=True)          # sticky: cleared where they are reported (blocking, error path only)
            ops._raise_on_range_flags(flags, "GraphedStep (a replay of the captured step)")

    def __call__(self):
        self._look(wait=False)
        self.graph.replay()
        if self._guard is not None:
            ev = torch.cuda.Event()
            ev.record()
            self._guard["event"] = ev
        return self.outputs

    def drain(self):
        """wait for the outstanding replay's range-guard word and raise if it carried a flag"""
        self._look(wait=True)
