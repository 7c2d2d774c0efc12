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
    def __init__(self, fn, warmup=3):
        """fn: zero-argument callable that runs the step on the current stream and
        returns a tensor / dict / tuple of tensors (its inputs must be tensors that
        stay alive and are updated in place between replays)."""
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the default stream: one-time inits
            for _ in range(warmup):        # (LDS attributes, weight packing) stay out of the graph
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (e.g. the RCCL watchdog of an
        # initialised process group) may keep issuing HIP calls during the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs
