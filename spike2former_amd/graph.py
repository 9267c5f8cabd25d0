"""hipGraph capture of the whole training step.

One C2 step issues ~5 000 kernel launches from Python; at ~20 us of host time per launch the step is host-bound long
before the GPU is busy.  Shapes are static (fixed crop, fixed T), so the step -- membrane reset, gradient-buffer clear,
forward, loss, backward -- is captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed: one host call
per step.  The kernels launched through the C ABI take the capture stream like any other launch; the ABI allocates
nothing and never synchronises, so it is capture-safe by construction (s2f_* use hipMemsetAsync only).
"""
import torch

from . import ops
from .neuron import reset_net


class GraphedStep:
    def __init__(self, model, loss_fn, example_input, grad_buffer=None, warmup=3):
        self.model, self.loss_fn = model, loss_fn
        self.static_in = example_input.clone()
        self.grad_buffer = grad_buffer
        self.graph = torch.cuda.CUDAGraph()
        self.static_loss = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up on a side stream (allocator pools, lazy inits, caches)
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # With a process group alive, RCCL's watchdog thread polls its events while this thread captures: "thread_local"
        # keeps its (legal, uncaptured) calls from invalidating the capture; single-process runs keep the strict default.
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.static_loss = self._eager_step()
        torch.cuda.synchronize()

    def _eager_step(self):
        reset_net(self.model)
        if self.grad_buffer is not None:
            self.grad_buffer.zero()
        else:
            for p in self.model.parameters():
                p.grad = None
        out = self.model(self.static_in)
        loss = self.loss_fn(*out)
        loss.backward()
        ops.wgrad_join()                  # side-stream weight gradients (ops.WGRAD_STREAM) rejoin before packing
        if self.grad_buffer is not None:
            self.grad_buffer.gather()
        return loss.detach()

    def __call__(self, x=None):
        if x is not None:
            self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_loss
