"""The thin training driver around the hot path -- SURVEY section 8 row f2: what mmengine's runner does per iteration for
the Spike2Former configs (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-167), without the runner:

    ResetModelHook -> data_preprocessor(training=True) -> model(mode='loss') -> parse_losses -> backward ->
    [all-reduce] -> clip_grad(max_norm=0.01) -> AdamW(paramwise lr / decay multipliers) -> LinearLR -> PolyLR

mmengine itself is not part of the reference tree (third-party, mmengine 0.8.4 per Seg/README.md:25): `parse_losses`,
the `custom_keys` rule of DefaultOptimWrapperConstructor and the two schedulers are restated from its published behaviour;
parity is pinned by the config's own values only.  The optimiser is torch.optim.AdamW, which is what the reference runs."""
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
        span = max(self.total - self.warmup, 1)
        return max(1.0 - (t - self.warmup) / span, 0.0) ** self.power

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
