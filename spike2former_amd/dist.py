"""Data parallelism: one process per GPU, the image batch sharded across ranks, parameters replicated, and ONE
all-reduce of a flat fp32 gradient buffer per step over RCCL/xGMI (the reference wraps the model in
MMDistributedDataParallel, tools/train.py:40-44, configs/_base_/default_runtime.py:5).  BatchNorm statistics stay
per-GPU, as in the reference (plain nn.BatchNorm, SyncBN ignored -- SURVEY 2.3)."""
import os

import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        if os.environ.get("S2F_FORCE_DIST") and torch.cuda.is_available() and not dist.is_initialized():
            # single-GPU rehearsal of the N > 1 path on the real RCCL: communicator, watchdog thread, hipGraph capture with a
            # process group alive, the side-stream all-reduce of the flat gradient buffer (a one-rank all-reduce)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            torch.cuda.set_device(0)
            dist.init_process_group(backend=backend or os.environ.get("S2F_DIST_BACKEND") or "nccl", rank=0, world_size=1)
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # "nccl" is RCCL on ROCm; S2F_DIST_BACKEND=gloo lets two ranks share one GPU when testing the N > 1 code path
        backend = os.environ.get("S2F_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    elif torch.cuda.is_available():
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradAllReduce:
    """One contiguous fp32 buffer holds every parameter gradient (137 MB at C2).

    `gather()` packs the gradients autograd produced into the buffer with a handful of batched-copy launches
    (torch.cat into `out=`), `reduce()` issues a single averaging all-reduce (ncclAvg on RCCL; SUM + division on gloo),
    `wait()` joins it when it ran on a side stream.  Gradients are *assigned* by autograd (`p.grad` is None before backward) rather than accumulated
    into pre-existing views: accumulating costs one tiny add kernel per parameter (1 021 launches, 4.7 ms per C2 step).
    `views[i]` is parameter i's slice of the flat buffer (what an optimiser would consume)."""

    def __init__(self, params, world_size=None):
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # every parameter's slot starts on a 16-byte boundary (slots padded to a multiple of 4 floats; the pads stay zero): the
        # kernels that add into a slot store 16-byte rows, and whether a weight gradient goes through its sink must not depend on
        # where compact() happens to place it
        n = sum(self._slot(p) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._lay_out()
        self.stream = None                     # side stream, only for gloo on device memory (see reduce())
        self.work = None
        self.sinks = False
        self._zero = None
        self._n_dense = self._dense_elems = None
        self.missing = []                      # see without_gradient(); refreshed by gather()

    @staticmethod
    def _slot(p):
        return (p.numel() + 3) // 4 * 4

    def _lay_out(self):
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            self.offsets.append(off)
            off += self._slot(p)

    def _pieces(self, params):
        """flat pieces (gradient or zeros, then the slot's zero pad) of consecutive parameters, for one batched copy"""
        out = []
        for p in params:
            out.append((p.grad if p.grad is not None else self._zero[:p.numel()]).reshape(-1))
            pad = self._slot(p) - p.numel()
            if pad:
                out.append(self._zero[:pad])
        return out

    def _need_zero(self):
        if self._zero is None:
            self._zero = torch.zeros(max(p.numel() for p in self.params), dtype=torch.float32, device=self.flat.device)

    def install_sinks(self):
        """Let the split-K weight-gradient kernels accumulate straight into this buffer (ops.GRAD_SINKS): `zero()` then
        clears the whole buffer with ONE memset per step instead of one per weight-gradient launch (231 at C2)."""
        from . import ops
        import weakref
        # keyed by address for the lookup, but an address is not an identity (a freed model's weight address is reused by the
        # next one): each entry carries a weak reference to the parameter it belongs to, checked by ops._sink_for
        self._sink_dict = ops.GRAD_SINKS = {p.data_ptr(): (weakref.ref(p), v) for p, v in zip(self.params, self.views)}
        self.sinks = True

    def close(self):
        """Detach the gradient sinks (call before dropping this object: ops.GRAD_SINKS is process-global)."""
        from . import ops
        ops.wgrad_drop()
        if self.sinks:
            if ops.GRAD_SINKS is getattr(self, "_sink_dict", None):          # not if a newer buffer has installed its own
                ops.GRAD_SINKS = None
            self.sinks = False

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown
            pass

    def zero(self):
        """Before backward: drop the old gradients so that autograd assigns instead of accumulating."""
        from . import ops
        ops.wgrad_drop()                       # deferred weight gradients of an abandoned step must not land in this one
        for p in self.params:
            p.grad = None
        if self.sinks:
            self.flat.zero_()

    def gather(self):
        """After backward: pack p.grad into the flat buffer (parameters without a gradient keep zeros)."""
        from . import ops
        ops.wgrad_flush()                      # deferred weight gradients (ops.DEFER_DW) go into their sinks now
        self.missing = self.without_gradient()
        # No gradient tensor: with sinks the parameter's slice already holds the sum (or the zeros of `zero()`) and is left
        # alone -- the packing runs over the maximal runs of parameters that do have a tensor; without sinks it is a zero.
        self._need_zero()
        if not self.sinks:
            torch.cat(self._pieces(self.params), out=self.flat)
            return
        dense = self.params[:self._n_dense] if self._n_dense is not None else []
        if (self._n_dense is not None and all(p.grad is not None for p in dense)
                and all(p.grad is None for p in self.params[self._n_dense:])):
            # compacted and this step's gradients arrived as in the discovery step: ONE batched copy over the leading region
            torch.cat(self._pieces(dense), out=self.flat[:self._dense_elems])
            return
        # not compacted, or a parameter changed sides since compact() (unused on this iteration: no tensor although it lies in the
        # dense region; a gradient tensor for a parameter placed in the sink region): one batched copy per run of gradient tensors.
        # A slot without a tensor is left alone -- it holds the zeros of zero() or the sum a kernel added through the sink.  A gradient
        # TENSOR for a parameter of the sink region is ADDED to its slot: a kernel may have added a share through the sink in the same
        # step (a weight used twice, once by a sinking kernel and once by an ATen op), and a copy would overwrite that share.
        run, start, off = [], 0, 0
        for i, p in enumerate(self.params):
            sunk = self._n_dense is not None and i >= self._n_dense
            if p.grad is not None and not sunk:
                if not run:
                    start = off
                run += self._pieces([p])
            else:
                if run:
                    torch.cat(run, out=self.flat[start:off])
                    run = []
                if p.grad is not None:
                    self.views[i].add_(p.grad)
            off += self._slot(p)
        if run:
            torch.cat(run, out=self.flat[start:off])

    def without_gradient(self):
        """Indices of the parameters that certainly received NO gradient in the step just packed: no tensor from autograd and no
        sink a kernel could have added through (train.FlatAdamW skips them, as torch.optim.AdamW skips a parameter whose .grad is
        None).  A parameter WITH a sink cannot be told apart from one whose gradient is all zeros without reading the device: it
        is treated as having a (zero) gradient -- decayed and moment-updated, where torch would skip it if its layer never ran.
        Every sunk weight of this path is used by every step, so the deviation is latent."""
        from . import ops
        sinks = ops.GRAD_SINKS if self.sinks else None
        return [i for i, p in enumerate(self.params) if p.grad is None and not (sinks and p.data_ptr() in sinks)]

    def pack(self, grads):
        """Like gather(), from an explicit gradient list aligned with `self.params` (torch.autograd.grad output; None =
        no tensor: sunk or unused)."""
        saved = [p.grad for p in self.params]
        try:
            for p, g in zip(self.params, grads):
                p.grad = g
            self.gather()
        finally:
            for p, g in zip(self.params, saved):
                p.grad = g

    def compact(self):
        """After one step with sinks: move the parameters whose gradient arrived through a sink (p.grad is None) behind the
        others in the flat layout, so that packing stays ONE batched copy over the leading region."""
        from . import ops
        ops.wgrad_drop()                       # pending jobs hold views of the OLD layout
        have = [p for p in self.params if p.grad is not None]
        rest = [p for p in self.params if p.grad is None]
        self.params = have + rest
        self._n_dense, self._dense_elems = len(have), sum(self._slot(p) for p in have)
        self._lay_out()
        if self.sinks:
            self.install_sinks()

    def reduce_async(self, buckets=1):
        """The averaging all-reduce issued WITHOUT blocking the caller's stream: it runs on a side stream behind everything
        enqueued so far, as `buckets` back-to-back collectives over consecutive slices of the flat buffer (137 MB at C2: one
        bucket ~ 0.2-1.6 ms over xGMI; more buckets let a consumer start on the first slice while the rest is in flight).
        The caller's stream goes on -- graph.GraphedOverlapStep replays the NEXT step's forward under it -- and `wait()`
        orders the caller behind the collectives before the buffer is touched again."""
        if self.world == 1 and not (os.environ.get("S2F_FORCE_DIST") and dist.is_initialized()):
            return
        if not self.flat.is_cuda:                 # host tensors (the gloo tests): the same bucketing, blocking
            n = self.flat.numel()
            step = (n + buckets - 1) // buckets
            for lo in range(0, n, step):
                piece = self.flat[lo:lo + step]
                dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                piece.div_(self.world)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.flat.device)
        self.stream.wait_stream(torch.cuda.current_stream())
        n = self.flat.numel()
        step = (n + buckets - 1) // buckets
        avg = dist.get_backend() == "nccl"
        with torch.cuda.stream(self.stream):
            for lo in range(0, n, step):
                piece = self.flat[lo:lo + step]
                w = dist.all_reduce(piece, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=not avg)
                if not avg:
                    w.wait()
                    piece.div_(self.world)

    def reduce(self, async_op=True):
        if self.world == 1 and not (os.environ.get("S2F_FORCE_DIST") and dist.is_initialized()):
            return
        if self.flat.is_cuda and dist.get_backend() == "nccl":
            # RCCL: the blocking form queues the collective on the CALLER's stream, right behind the replayed step, and averages
            # inside the collective (ncclAvg: no separate pass over the 137 MB buffer).  The asynchronous form runs on RCCL's own
            # stream; its two event hand-overs around the graph launch cost ~1 ms each (55.6 vs 53.5 ms/step in the one-rank
            # rehearsal), and nothing in the step overlaps with the collective anyway.
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, async_op=False)
            return
        if self.flat.is_cuda:
            # gloo on device memory (two ranks rehearsing on one GPU): on a side stream -- queued on the step's own stream the
            # staging copies took 21 s per call
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=self.flat.device)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
                self.work.wait()
                self.flat.div_(self.world)
            return
        self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=False)
        self.flat.div_(self.world)

    def wait(self):
        if self.world == 1 and not (os.environ.get("S2F_FORCE_DIST") and dist.is_initialized()):
            return
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.work = None


def broadcast_params(module, src=0):
    """Rank `src`'s parameters and buffers to every rank: ONE broadcast per dtype of a flat packed buffer (three collectives
    instead of ~1 860 at C2), copied back IN PLACE under no_grad -- not through `.data`, whose separate version counter would
    leave the cached bf16 weight splits (ops.split_weight) looking valid on the receiving ranks."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    groups = {}
    for t in list(module.parameters()) + list(module.buffers()):
        groups.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for ts in groups.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def shard_batch(global_batch, rank, world):
    """Even split of the image batch (C3: 16 -> 2 per GPU; C5: 8 -> 1 per GPU)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, per
