"""The thin training driver around the hot path -- SURVEY section 8 row f2: what mmengine's runner does per iteration for
the Spike2Former configs (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-167), without the runner:

    ResetModelHook -> data_preprocessor(training=True) -> model(mode='loss') -> parse_losses -> backward ->
    [all-reduce] -> clip_grad(max_norm=0.01) -> AdamW(paramwise lr / decay multipliers) -> LinearLR -> PolyLR

mmengine itself is not part of the reference tree (third-party, mmengine 0.8.4 per Seg/README.md:25): `parse_losses`,
the `custom_keys` rule of DefaultOptimWrapperConstructor and the two schedulers are restated from its published behaviour;
parity is pinned by the config's own values only.  The reference's optimiser is torch.optim.AdamW behind clip_grad_norm_;
`OptimWrapper` runs exactly that (eager, one group per parameter), `FlatAdamW` is the same update as three HIP launches over
the flat gradient buffer of the data-parallel step (csrc/optim.hip), pinned against the former in tests/test_gpu_round5.py."""
import torch

from .neuron import reset_net


def parse_losses(losses):
    """mmengine BaseModel.parse_losses: tensors -> mean, lists -> sum of means; the step's loss = sum of the entries whose
    key contains 'loss'.  -> (loss, log_vars)"""
    log_vars = {}
    for k, v in losses.items():
        if torch.is_tensor(v):
            log_vars[k] = v.mean()
        elif isinstance(v, (list, tuple)):
            log_vars[k] = sum(t.mean() for t in v)
        else:
            raise TypeError(f"{k} is not a tensor or list of tensors")
    loss = sum(v for k, v in log_vars.items() if "loss" in k)
    log_vars = {"loss": loss, **log_vars}
    return loss, log_vars


def param_groups(model, lr, weight_decay, paramwise_cfg=None):
    """One group per parameter, DefaultOptimWrapperConstructor's `custom_keys` rule: the longest key (ties: alphabetical)
    that is a substring of the parameter's full name sets `lr_mult` / `decay_mult`."""
    custom = dict((paramwise_cfg or {}).get("custom_keys", {}))
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    groups = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g = {"params": [p], "lr": lr, "weight_decay": weight_decay, "name": name}
        for k in keys:
            if k in name:
                g["lr"] = lr * custom[k].get("lr_mult", 1.0)
                g["weight_decay"] = weight_decay * custom[k].get("decay_mult", 1.0)
                break
        groups.append(g)
    return groups


def _multipliers(names, paramwise_cfg):
    """DefaultOptimWrapperConstructor's `custom_keys` rule per parameter name -> [(lr_mult, decay_mult)]"""
    custom = dict((paramwise_cfg or {}).get("custom_keys", {}))
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    out = []
    for name in names:
        lm, dm = 1.0, 1.0
        for k in keys:
            if k in name:
                lm, dm = custom[k].get("lr_mult", 1.0), custom[k].get("decay_mult", 1.0)
                break
        out.append((lm, dm))
    return out


class FlatAdamW:
    """clip_grad_norm_(max_norm) + AdamW over `dist.FlatGradAllReduce`'s flat gradient buffer: three HIP launches per iteration
    (s2f_grad_sqnorm -> s2f_adamw_prepare -> s2f_adamw_step), no host synchronisation -- `step()` can be captured behind the
    step's hipGraph (graph.GraphedStep(optimizer=...)) or called eagerly after the all-reduce when N > 1.

    What mmengine's OptimWrapper does for the config (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-155) with
    ~1 000 single-parameter groups: gradients are read where the step packed (and the all-reduce averaged) them, both moments
    are flat buffers of the same layout, parameters are addressed through a pointer table (they stay where their modules own
    them).  `param_groups` is a list of {'lr', 'weight_decay', 'name'} dictionaries, one per parameter, that a scheduler
    (LinearThenPoly) may rewrite between iterations; the table goes to the device at the start of `step()`.
    Build it AFTER `grad_buffer.compact()`: the tables follow the buffer's layout (checked at every step)."""

    def __init__(self, model, grad_buffer, lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.005, paramwise_cfg=None,
                 clip_grad=None):
        from ._lib import lib
        self.red, self.betas, self.eps = grad_buffer, (float(betas[0]), float(betas[1])), float(eps)
        self.max_norm = float((clip_grad or {}).get("max_norm", 0.0)) if clip_grad else 0.0
        if clip_grad and clip_grad.get("norm_type", 2) != 2:
            raise NotImplementedError("FlatAdamW clips by the 2-norm (the Spike2Former configs' norm_type)")
        names = {id(p): n for n, p in model.named_parameters()}
        self.params = list(grad_buffer.params)
        mult = _multipliers([names.get(id(p), "") for p in self.params], paramwise_cfg)
        self.param_groups = [{"params": [p], "lr": lr * lm, "weight_decay": weight_decay * dm, "name": names.get(id(p), "")}
                             for p, (lm, dm) in zip(self.params, mult)]
        dev = grad_buffer.flat.device
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(grad_buffer.flat), torch.zeros_like(grad_buffer.flat)
        n = grad_buffer.flat.numel()
        self.partials = torch.zeros(int(lib.s2f_grad_sqnorm_parts(n)), dtype=torch.float64, device=dev)
        self.state = torch.zeros(8, dtype=torch.float32, device=dev)
        self._hyper_host = [torch.empty(len(self.params), 2, dtype=torch.float32) for _ in range(2)]
        if dev.type == "cuda":
            self._hyper_host = [h.pin_memory() for h in self._hyper_host]
        self._hyper_events, self._hyper_turn, self._hyper_last = [None, None], 0, None
        self.hyper = torch.zeros(len(self.params), 2, dtype=torch.float32, device=dev)
        self._build_tables()

    def _build_tables(self):
        from ._lib import lib
        chunk = int(lib.s2f_adamw_chunk_elems())
        dev = self.red.flat.device
        self._layout = (tuple(self.red.offsets), tuple(p.data_ptr() for p in self.params))
        slots, chunks = [], []
        for i, (p, off) in enumerate(zip(self.params, self.red.offsets)):
            if not p.is_contiguous():
                raise RuntimeError("FlatAdamW: parameters must be contiguous")
            slots.append((p.data_ptr(), off, p.numel()))
            chunks += [(i, s) for s in range(0, p.numel(), chunk)]
        self.slots = torch.tensor(slots, dtype=torch.int64).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)

    def sync_hyper(self):
        """the per-parameter (lr, weight_decay) table of this iteration -> device.  Uploaded only when a scheduler changed a value,
        from one of TWO pinned tables in turn: the copy reads host memory when it EXECUTES, and the host may run several replays
        ahead -- the table a queued copy still has to read is never the one being rewritten (its event is waited for first)."""
        import numpy as np
        vals = np.array([(g["lr"], g["weight_decay"]) for g in self.param_groups], dtype=np.float32).reshape(-1, 2)
        for i in getattr(self.red, "missing", ()):
            vals[i, 0] = -1.0          # no gradient in the step just packed (p.grad is None): the kernel skips the parameter (optim.hip)
        if self._hyper_last is not None and np.array_equal(vals, self._hyper_last):
            return
        self._hyper_turn ^= 1
        host, ev = self._hyper_host[self._hyper_turn], self._hyper_events[self._hyper_turn]
        if ev is not None:
            ev.synchronize()
        host.copy_(torch.from_numpy(vals))
        self.hyper.copy_(host, non_blocking=True)
        if self.hyper.is_cuda:
            ev = torch.cuda.Event() if ev is None else ev
            ev.record()
            self._hyper_events[self._hyper_turn] = ev
        self._hyper_last = vals

    def mark_updated(self):
        """The kernels wrote the parameters behind autograd's back: bump their version counters so that every cache keyed on a
        weight's version (the bf16 term splits / packs of ops.gemm, the composed eval-mode BatchNorm affines of fused.py) re-converts
        at its next eager use.  step() does it itself when it runs eagerly; a REPLAY of a hipGraph that holds the update cannot
        (nothing of Python runs inside it): graph.GraphedStep / GraphedHungarianStep call this after every replay."""
        torch.autograd.graph.increment_version(self.params)

    def step(self, sync_hyper=True):
        """Gradients: `grad_buffer.flat` as the step left it (packed, averaged).  -> the gradient norm (device scalar)."""
        from ._lib import check, lib
        from .ops.core import _stream
        if ((tuple(self.red.offsets), tuple(p.data_ptr() for p in self.params)) != self._layout
                or [id(p) for p in self.red.params] != [id(p) for p in self.params]):
            raise RuntimeError("FlatAdamW: the gradient buffer's layout (compact()) or a parameter's storage changed since the tables "
                               "were built; call rebuild()")
        if sync_hyper:
            self.sync_hyper()
        flat, b1, b2 = self.red.flat, self.betas[0], self.betas[1]
        s = _stream()
        check(lib.s2f_grad_sqnorm(flat.data_ptr(), flat.numel(), self.partials.data_ptr(), s), "s2f_grad_sqnorm")
        check(lib.s2f_adamw_prepare(self.partials.data_ptr(), self.partials.numel(), self.max_norm, b1, b2, self.state.data_ptr(), s),
              "s2f_adamw_prepare")
        check(lib.s2f_adamw_step(self.slots.data_ptr(), self.hyper.data_ptr(), self.chunks.data_ptr(), self.chunks.shape[0],
                                 flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.state.data_ptr(),
                                 b1, b2, self.eps, s), "s2f_adamw_step")
        # the kernels wrote the parameters behind autograd's back: bump their version counters so that every cache keyed on a
        # weight's version (the bf16 term splits / packs of ops.gemm) re-converts in eager use; a captured step re-converts inside
        # the graph anyway (ops.resplit_all)
        if not torch.cuda.is_current_stream_capturing():
            self.mark_updated()
        return self.state[1]

    def rebuild(self):
        """after grad_buffer.compact(): re-lay the moments out with the buffer (values kept per parameter)"""
        old_off = dict(zip((id(p) for p in self.params), self._layout[0]))
        m, v = torch.zeros_like(self.red.flat), torch.zeros_like(self.red.flat)
        groups = {id(g["params"][0]): g for g in self.param_groups}
        for p, off in zip(self.red.params, self.red.offsets):
            o = old_off[id(p)]
            m[off:off + p.numel()] = self.exp_avg[o:o + p.numel()]
            v[off:off + p.numel()] = self.exp_avg_sq[o:o + p.numel()]
        self.exp_avg, self.exp_avg_sq = m, v
        self.params = list(self.red.params)
        self.param_groups = [groups[id(p)] for p in self.params]
        self._hyper_last = None                      # the table's row order changed with the buffer's layout: upload again
        self._build_tables()

    @property
    def grad_norm(self):
        return self.state[1]

    @property
    def clip_coef(self):
        return self.state[0]

    def state_dict(self):
        """per-parameter moments by name + the step count (what torch.optim.AdamW.state_dict carries, keyed by name)"""
        out = {"step": int(self.state[4].item()), "state": {}}
        for g, p, off in zip(self.param_groups, self.params, self.red.offsets):
            out["state"][g["name"]] = {"exp_avg": self.exp_avg[off:off + p.numel()].view_as(p).clone(),
                                       "exp_avg_sq": self.exp_avg_sq[off:off + p.numel()].view_as(p).clone()}
        return out

    def load_state_dict(self, sd):
        self.state[4] = float(sd["step"])
        for g, p, off in zip(self.param_groups, self.params, self.red.offsets):
            st = sd["state"][g["name"]]
            self.exp_avg[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))


class LinearThenPoly:
    """LinearLR(start_factor, begin=0, end=warmup) followed by PolyLR(eta_min, power, begin=warmup, end=total), by iteration:
    factor(t) multiplies every group's base lr."""

    def __init__(self, optimizer, warmup=1500, total=160000, start_factor=1e-6, eta_min=0.0, power=1.0):
        self.opt, self.warmup, self.total = optimizer, warmup, total
        self.start_factor, self.eta_min, self.power = start_factor, eta_min, power
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.t = 0
        self._apply()

    def factor(self, t):
        if t < self.warmup:
            return self.start_factor + (1.0 - self.start_factor) * t / max(self.warmup - 1, 1) if self.warmup > 1 else 1.0
        # mmengine 0.8.4 PolyLR.__init__: total_iters = end - begin - 1; its recursive rule
        #   lr_t = (lr_{t-1} - eta_min) * (1 - 1 / (total_iters - s + 1))^power + eta_min,  s = t - begin = 1 .. total_iters
        # telescopes to (1 - s / total_iters)^power: the floor eta_min is reached at t = end - 1 and held
        span = max(self.total - self.warmup - 1, 1)
        return max(1.0 - min(t - self.warmup, span) / span, 0.0) ** self.power

    def _apply(self):
        f = self.factor(self.t)
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self.eta_min + (b - self.eta_min) * f if self.t >= self.warmup else b * f

    def step(self):
        self.t += 1
        self._apply()


class OptimWrapper:
    """optimizer + clip_grad, as the config's `optim_wrapper` (:150-155)."""

    def __init__(self, model, optimizer=None, clip_grad=None, paramwise_cfg=None):
        cfg = dict(optimizer or dict(type="AdamW", lr=0.001, betas=(0.9, 0.999), weight_decay=0.005))
        kind = cfg.pop("type", "AdamW")
        if kind != "AdamW":
            raise NotImplementedError(f"optimizer {kind}: the Spike2Former configs use AdamW")
        lr, wd = cfg.pop("lr"), cfg.pop("weight_decay", 0.0)
        self.optimizer = torch.optim.AdamW(param_groups(model, lr, wd, paramwise_cfg), lr=lr, weight_decay=wd, **cfg)
        self.clip = dict(clip_grad) if clip_grad else None
        self.params = [p for g in self.optimizer.param_groups for p in g["params"]]

    def update_params(self, loss):
        loss.backward()
        return self.step()

    def step(self):
        norm = None
        if self.clip:
            norm = torch.nn.utils.clip_grad_norm_(self.params, self.clip.get("max_norm", 0.01), self.clip.get("norm_type", 2))
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        return norm


def train_step(model, data, optim_wrapper, scheduler=None):
    """One iteration: reset (ResetModelHook.before_train_iter) -> preprocess -> loss -> backward -> clip -> AdamW -> lr.
    `model` is an EncoderDecoder carrying a `data_preprocessor` (or data is already {'inputs': tensor, 'data_samples': ...}).
    -> log_vars (python floats) with 'grad_norm'."""
    reset_net(model)
    pre = getattr(model, "data_preprocessor", None)
    if pre is not None:
        data = pre(data, True)
    losses = model(data["inputs"], data["data_samples"], mode="loss")
    loss, log_vars = parse_losses(losses)
    norm = optim_wrapper.update_params(loss)
    if scheduler is not None:
        scheduler.step()
    out = {k: float(v) for k, v in log_vars.items()}
    if norm is not None:
        out["grad_norm"] = float(norm)
    return out
