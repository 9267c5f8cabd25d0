"""hipGraph capture of the whole training step.

One C2 step issues ~5 000 kernel launches from Python; at ~20 us of host time per launch the step is host-bound long
before the GPU is busy.  Shapes are static (fixed crop, fixed T), so the step -- membrane reset, gradient-buffer clear,
forward, loss, backward -- is captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed: one host call
per step.  The kernels launched through the C ABI take the capture stream like any other launch; the ABI allocates
nothing and never synchronises, so it is capture-safe by construction (clears are fill KERNELS: no hipMemsetAsync nodes).

Weights: the spike GEMMs read bf16 hi / mid / lo terms of every weight that `ops.split_weight*` caches per weight version.
A captured step does not rely on that cache: `reset_net -> ops.begin_step` re-splits EVERY registered weight inside the
graph (ops.resplit_all, one launch over a pointer table), so each replay multiplies by the live fp32 weights -- an optimiser
step or load_state_dict between replays is honoured, forward and backward stay consistent
(tests/test_gpu_full_size.py::test_graph_replay_follows_weight_updates).  Weights that enter the model only after the
capture (none on this path) would need a new capture.
"""
import torch

from . import ops
from .neuron import reset_net


def _check_settings(captured):
    """A captured hipGraph bakes in the launch structure the op-layer switches selected at capture time (ops.cfg): replaying it
    under different settings would mix two configurations silently -- e.g. eager steps of a comparison running with a switch
    flipped while the graph still replays the old kernels."""
    now = ops.cfg.snapshot()
    if now != captured:
        diff = {k: (captured[k], now[k]) for k in now if now[k] != captured.get(k)}
        raise RuntimeError(f"this hipGraph was captured under different op-layer settings (captured, now): {diff}; re-capture")


class GraphedStep:
    """`optimizer` (train.FlatAdamW over `grad_buffer`): the parameter update -- clip + AdamW, three launches -- is captured behind
    the gradient packing, so one replay is one training ITERATION (single process; with N > 1 the all-reduce sits between the step
    and the update, which the caller then runs eagerly: `step(); grad_buffer.reduce(); optimizer.step()`).  Its per-parameter
    (lr, weight_decay) table is uploaded before every replay (a scheduler may have rewritten it)."""

    def __init__(self, model, loss_fn, example_input, grad_buffer=None, warmup=3, optimizer=None):
        self.model, self.loss_fn = model, loss_fn
        self.static_in = example_input.clone()
        self.grad_buffer = grad_buffer
        self.optimizer = optimizer
        if optimizer is not None:
            optimizer.sync_hyper()
        self.graph = torch.cuda.CUDAGraph()
        self.static_loss = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up on a side stream (allocator pools, lazy inits, caches)
            for _ in range(warmup):
                self._eager_step()
        if ops.GLUE_MODE:                              # one more warm-up on the launch structure that is captured (ops.GlueMode)
            with torch.cuda.stream(side), ops.glue_mode():
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        ops.resplit_all(self.static_in.device)          # builds the weight-split job table the captured step replays
        torch.cuda.synchronize()
        # With a process group alive, RCCL's watchdog thread polls its events while this thread captures: "thread_local"
        # keeps its (legal, uncaptured) calls from invalidating the capture; single-process runs keep the strict default.
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        # ops.GLUE_MODE: the residual aten calls of the step (autograd's gradient accumulation, scalar multiples, copies, fills, small
        # sums) are routed to csrc/glue.hip while the step is RECORDED -- the replay then consists of this package's kernels only
        with torch.cuda.graph(self.graph, capture_error_mode=mode), ops.glue_mode():
            self.static_loss = self._eager_step()
        torch.cuda.synchronize()
        # the graph bakes in the addresses of the conversion job tables and of every cached split / pack buffer: hold them (an
        # eager forward after an optimiser step re-converts INTO the same buffers, ops._cache_buffer, and replaces tables)
        self._converted = ops.conversion_state()
        self._settings = ops.cfg.snapshot()

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
        if self.optimizer is not None:
            self.optimizer.step(sync_hyper=False)
        return loss.detach()

    def __call__(self, x=None):
        _check_settings(self._settings)
        if x is not None:
            self.static_in.copy_(x, non_blocking=True)
        if self.optimizer is not None:
            self.optimizer.sync_hyper()
        self.graph.replay()
        if self.optimizer is not None:
            # the replayed update changed every weight after this replay's own re-split ran (at its START): the version-keyed
            # conversion caches now hold the terms of the weights BEFORE the update under unchanged version counters -- an eager
            # forward (validation, predict(), export) would multiply by one-step-stale bf16 terms.  Bump the versions.
            self.optimizer.mark_updated()
        return self.static_loss


class GraphedSplitStep:
    """The training step with a loss that needs the host in the middle (the Hungarian assignment):
    TWO hipGraphs around an eager loss --
        graph A : membrane reset + gradient-buffer clear + model forward           (autograd recorded once, at capture)
        eager   : loss(outputs.detach()) and its backward -> d loss / d outputs
        graph B : backward of graph A's recorded autograd graph from those gradients + packing into the flat buffer
    The same construction as torch.cuda.make_graphed_callables, but the parameter gradients never leave the graph as
    per-parameter tensors (that path clones 1 200 gradients per step: 79 ms/step, no better than eager launches).
    No autograd graph of an earlier eager step may be alive when this is built (drop the old outputs / losses first): its
    gradient accumulators belong to the default stream, the capture would have to wait on it, and ROCm 7.2 crashes in
    hipStreamEndCapture instead of reporting the illegal dependency."""

    def __init__(self, model, example_input, grad_buffer, warmup=3):
        self.model, self.red = model, grad_buffer
        self.static_in = example_input.clone()
        params = [p for p in grad_buffer.params]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                outs = self._forward()
                torch.autograd.grad(outs, params, [torch.ones_like(o) for o in outs], allow_unused=True)
        torch.cuda.current_stream().wait_stream(side)
        ops.resplit_all(self.static_in.device)          # builds the weight-split job table graph A replays
        torch.cuda.synchronize()
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.graph_a, pool=pool, capture_error_mode=mode):
            self.outs = self._forward()
        self.grad_outs = [torch.zeros_like(o) for o in self.outs]
        with torch.cuda.graph(self.graph_b, pool=pool, capture_error_mode=mode):
            grads = torch.autograd.grad(self.outs, params, self.grad_outs, allow_unused=True)
            ops.wgrad_join()
            self.red.pack(grads)
        torch.cuda.synchronize()
        self._converted = ops.conversion_state()
        self._settings = ops.cfg.snapshot()          # see GraphedStep

    def _forward(self):
        reset_net(self.model)
        self.red.zero()
        return tuple(self.model(self.static_in))

    def forward(self, x=None):
        """-> the model outputs (static tensors, valid until the next forward), detached leaves that require grad."""
        _check_settings(self._settings)
        if x is not None:
            self.static_in.copy_(x, non_blocking=True)
        self.graph_a.replay()
        return [o.detach().requires_grad_(True) for o in self.outs]

    def backward(self, leaves):
        """`leaves`: what forward() returned, after loss.backward() has filled their .grad."""
        for buf, leaf in zip(self.grad_outs, leaves):
            if leaf.grad is None:
                buf.zero_()
            else:
                buf.copy_(leaf.grad)
        self.graph_b.replay()


class GraphedHungarianStep:
    """The REAL training step (Hungarian-matched loss on semantic maps) as two hipGraphs around the one
    thing that has to happen on the host, the assignment:
        graph A : membrane reset + gradient-buffer clear + model forward + matching costs against every class id
                  (loss.MaskFormerLoss.costs_all_classes) + their copy into pinned host memory
        host    : scipy linear_sum_assignment per (layer, image) on the columns of the classes present -> three small tables
                  (loss.MaskFormerLoss.match_tables), uploaded into static device buffers
        graph B : the losses from the tables (loss_from_tables: every shape is independent of the matching) + backward of the
                  whole model + packing of the gradients into the flat buffer
    Between the graphs the GPU idles for the copy, the assignment and one upload -- not for ~450 eager launches of the loss and
    its backward as with GraphedSplitStep.  `__call__` returns the loss dictionary (static tensors, valid until the next call)."""

    def __init__(self, model, example_input, example_seg, grad_buffer, warmup=3, ignore_index=None, optimizer=None):
        head = model.decode_head
        self.optimizer = optimizer          # train.FlatAdamW: captured at the end of graph B (single process; see GraphedStep)
        if optimizer is not None:
            optimizer.sync_hyper()
        self.model, self.red, self.crit = model, grad_buffer, head.criterion
        self.ignore_index = head.ignore_index if ignore_index is None else ignore_index
        self.static_in = example_input.clone()
        self.static_seg = self.crit.seg_as_u8(example_seg, self.ignore_index).clone()
        params = [p for p in grad_buffer.params]
        dev = example_input.device
        with torch.no_grad():                                     # shapes of the outputs (and a first warm-up)
            cls, masks = self._forward()
        if not self.crit.semantic_ok(masks, self.static_seg):
            raise RuntimeError("GraphedHungarianStep needs semantic maps at twice the mask predictions' resolution")
        L, B, Q = cls.shape[:3]
        self.tgt_labels = torch.full((L, B, Q), self.crit.num_classes, dtype=torch.int64, device=dev)
        self.row_class = torch.full((B, L * Q), -1, dtype=torch.int32, device=dev)
        self.num_masks = torch.ones(L, dtype=torch.float32, device=dev)
        self.host_cost = torch.empty(L, B, Q, self.crit.num_classes, dtype=torch.float32).pin_memory()
        self.host_count = torch.empty(B, 256, dtype=torch.float32).pin_memory()
        self.host_tgt, self.host_rows, self.host_avg = (torch.empty(t.shape, dtype=t.dtype).pin_memory()
                                                        for t in (self.tgt_labels, self.row_class, self.num_masks))
        del cls, masks
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                outs = self._forward()
                self._costs(outs)
                side.synchronize()
                self._match()
                total = sum(self._losses(outs).values())
                torch.autograd.grad(total, params, allow_unused=True)
                ops.wgrad_join()
                del outs, total
        torch.cuda.current_stream().wait_stream(side)
        ops.wgrad_drop()                                 # warm-up gradients are not packed: drop their deferred launches
        ops.resplit_all(dev)
        torch.cuda.synchronize()
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.graph_a, pool=pool, capture_error_mode=mode):
            self.outs = self._forward()
            self._costs(self.outs)
        with torch.cuda.graph(self.graph_b, pool=pool, capture_error_mode=mode):
            losses = self._losses(self.outs)
            grads = torch.autograd.grad(sum(losses.values()), params, allow_unused=True)
            ops.wgrad_join()
            self.red.pack(grads)
            if self.optimizer is not None:
                self.optimizer.step(sync_hyper=False)
            self.losses = {k: v.detach() for k, v in losses.items()}
        torch.cuda.synchronize()
        self._converted = ops.conversion_state()
        self._settings = ops.cfg.snapshot()          # see GraphedStep

    def _forward(self):
        reset_net(self.model)
        self.red.zero()
        return tuple(self.model(self.static_in))

    def _costs(self, outs):
        with torch.no_grad():
            cost, count = self.crit.costs_all_classes(outs[0], outs[1], self.static_seg)
            self.host_cost.copy_(cost, non_blocking=True)
            self.host_count.copy_(count, non_blocking=True)

    def _match(self):
        """host_cost / host_count (complete: the stream was synchronised) -> the three device tables."""
        tgt, rows, avg = self.crit.match_tables(self.host_cost.numpy(), self.host_count.numpy())
        self.host_tgt.copy_(torch.from_numpy(tgt))
        self.host_rows.copy_(torch.from_numpy(rows))
        self.host_avg.copy_(torch.from_numpy(avg))
        self.tgt_labels.copy_(self.host_tgt, non_blocking=True)
        self.row_class.copy_(self.host_rows, non_blocking=True)
        self.num_masks.copy_(self.host_avg, non_blocking=True)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.num_masks.div_(dist.get_world_size()))          # reduce_mean (maskformer_head.py:459)

    def _losses(self, outs):
        return self.crit.loss_from_tables(outs[0], outs[1], self.static_seg, self.tgt_labels, self.row_class, self.num_masks)

    def __call__(self, x=None, seg=None):
        _check_settings(self._settings)
        if x is not None:
            self.static_in.copy_(x, non_blocking=True)
        if seg is not None:
            self.static_seg.copy_(self.crit.seg_as_u8(seg, self.ignore_index), non_blocking=True)
        if self.optimizer is not None:
            self.optimizer.sync_hyper()
        self.graph_a.replay()
        torch.cuda.current_stream().synchronize()
        self._match()
        self.graph_b.replay()
        if self.optimizer is not None:
            self.optimizer.mark_updated()          # see GraphedStep.__call__
        return self.losses


class GraphedOverlapStep:
    """BENCHMARK-ONLY (no weight update between steps): forward(k+1) replays before all-reduce(k) has finished, so an optimiser
    could not apply the averaged gradients of step k before step k+1 reads (and re-splits) the weights -- with an optimiser in the
    loop this would be one-step-stale data parallelism, not the reference's synchronous MMDistributedDataParallel step.  The
    synchronous form is GraphedStep + FlatGradAllReduce.reduce().
    The data-parallel step with its gradient all-reduce OVERLAPPED ("launched as backward finishes"):
    the step is two hipGraphs,
        graph F : membrane reset + weight re-split + forward + loss                    (does not touch the gradient buffer)
        graph B : gradient-buffer clear + backward + packing
    and the averaging all-reduce of step k runs on a side stream while graph F of step k + 1 replays:
        F(k+1) | wait for all-reduce(k) | B(k+1) | all-reduce(k+1) async | F(k+2) ...
    At C2 the forward is ~1/3 of a 50 ms step, the collective moves 137 MB (0.2-1.6 ms over xGMI): it disappears behind the
    forward.  `finish()` joins the last collective (the gradients of the last step are then averaged in `grad_buffer.flat`)."""

    def __init__(self, model, loss_fn, example_input, grad_buffer, warmup=3, buckets=1):
        self.model, self.loss_fn, self.red, self.buckets = model, loss_fn, grad_buffer, buckets
        self.static_in = example_input.clone()
        params = [p for p in grad_buffer.params]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                loss = self._forward()
                grad_buffer.zero()
                torch.autograd.grad([loss], params, allow_unused=True)
                ops.wgrad_join()
        torch.cuda.current_stream().wait_stream(side)
        ops.wgrad_drop()                                 # warm-up gradients are not packed: drop their deferred launches
        ops.resplit_all(self.static_in.device)
        torch.cuda.synchronize()
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        self.graph_f, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.graph_f, pool=pool, capture_error_mode=mode):
            self.loss = self._forward()
        with torch.cuda.graph(self.graph_b, pool=pool, capture_error_mode=mode):
            self.red.zero()
            grads = torch.autograd.grad([self.loss], params, allow_unused=True)
            ops.wgrad_join()
            self.red.pack(grads)
        torch.cuda.synchronize()
        self._converted = ops.conversion_state()
        self._settings = ops.cfg.snapshot()          # see GraphedStep

    def _forward(self):
        reset_net(self.model)
        return self.loss_fn(*self.model(self.static_in))

    def __call__(self, x=None):
        _check_settings(self._settings)
        if x is not None:
            self.static_in.copy_(x, non_blocking=True)
        self.graph_f.replay()                 # runs under the previous step's all-reduce
        self.red.wait()                       # ... which must be done before the buffer is cleared
        self.graph_b.replay()
        self.red.reduce_async(self.buckets)
        return self.loss

    def finish(self):
        self.red.wait()
