"""Host-side mirror of `Q_IFNode` / `Quant` / `reset_net` (Qtrick_architecture/clock_driven/neuron.py:395-550,
surrogate.py:644-701, functional.py:9-33, base.py:25-161).  The arithmetic runs in libs2f_hip.so (ops.lif)."""
import torch
import torch.nn as nn

from . import ops


class Quant(nn.Module):
    """Surrogate marker object (`Q_IFNode(surrogate_function=Quant())`): forward round(clamp(.,0,D)), straight-through
    backward inside [0, D] (surrogate.py:522-538).  Carries only the level count."""

    def __init__(self, alpha=4.0, spiking=True, D=8):
        super().__init__()
        self.alpha, self.spiking, self.D = alpha, spiking, D


class Q_IFNode(nn.Module):
    """Quantised integrate-and-fire node: h = v + x; s = rint(clamp(h,0,D)); v <- h - s*vth; returns s/D.

    `v` is a non-parameter memory that is NOT in the state_dict (base.py:25-69): python float 0. after `reset()`,
    afterwards a tensor shaped like the input, carried to the next call.
    `keep_membrane=False` skips writing `v` -- output-identical whenever `reset()` precedes every forward (what
    ResetModelHook guarantees during training, resetmodel_hook.py:17-37); it saves 4 B/element of HBM traffic.
    """

    def __init__(self, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False,
                 cupy_fp32_inference=False):
        super().__init__()
        assert isinstance(v_threshold, float) and isinstance(detach_reset, bool)
        if detach_reset:
            raise NotImplementedError("detach_reset=True is not instantiated anywhere on the Spike2Former path")
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = surrogate_function if surrogate_function is not None else Quant()
        self.D = getattr(self.surrogate_function, "D", 8)
        self.v = 0.0
        self.keep_membrane = True
        self.stats = None          # ops.new_stats() counters {sum of counts, non-zero counts} when firing is recorded
        self.stats_elems = 0
        self._prefired = None      # (u, version, y): this call's output, already produced by the kernel that made u

    def reset(self):
        self.v = 0.0
        self._prefired = None

    def prefire(self, u, y):
        """The producer of `u` (fused.bn_act with next_lif=self) has already applied this neuron to it -- same update, same
        membrane / mask / firing counters as a call would have done; the next forward(u) just hands `y` out."""
        self._prefired = (u, u._version, y)          # holding u keeps its address from being reused while pending
        if (PURE_MEMO and isinstance(self.v, float) and not self.keep_membrane and self.stats is None and u.is_cuda
                and u.is_contiguous()):
            # ... and any other pure neuron applied to the same tensor (the pixel decoder's lateral neurons read the
            # backbone taps that the next backbone stage's first neuron was just applied to) gets it from the memo
            _PURE_MEMO[(u.data_ptr(), u._version, u.numel(), u.requires_grad, self.D, self.v_threshold)] = (u, y)

    def extra_repr(self):
        return f"v_threshold={self.v_threshold}, v_reset={self.v_reset}, detach_reset={self.detach_reset}, D={self.D}"

    def _apply(self, fn, *a, **k):
        if isinstance(self.v, torch.Tensor):
            self.v = fn(self.v)
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        """The reference's interface: fp32 in, fp32 spikes out."""
        return self.fire(x, as_float=True)

    def fire(self, x, as_float=False, skip=False):
        """One call of the neuron -> ops.Spikes: the spike map as the kernels of this package pass it on (bf16 data + fp32
        autograd handle, ops.SPIKES_BF16).  `as_float`: the fp32 tensor instead (what `forward` returns).
        `skip`: -> (spikes, x') where x' is x for the residual branch of `x + f(neuron(x))`: the gradient that branch sends back is
        summed inside this neuron's backward kernel instead of by an add of the autograd engine (ops.lif; x' is x itself whenever
        that does not apply)."""
        if skip:
            return self._fire_skip(x)
        return self._fire(x, as_float)

    def _fire_skip(self, x):
        plain = (self._forward_hooks or self._forward_pre_hooks or self._prefired is not None or not isinstance(self.v, float)
                 or not x.is_cuda or not x.is_contiguous())
        if not plain and PURE_MEMO and not self.keep_membrane and self.stats is None:
            plain = (x.data_ptr(), x._version, x.numel(), x.requires_grad, self.D, self.v_threshold) in _PURE_MEMO
        if plain:
            return self._fire(x, False), x
        if self.stats is not None:
            self.stats_elems += x.numel()
        y, v_out, through = ops.lif(x, None, self.D, self.v_threshold, self.keep_membrane, self.stats, spikes=True, skip=True)
        if PURE_MEMO and not self.keep_membrane and self.stats is None:
            _PURE_MEMO[(x.data_ptr(), x._version, x.numel(), x.requires_grad, self.D, self.v_threshold)] = (x, y)
        self.v = v_out if self.keep_membrane else 0.0
        return y, through

    def _fire(self, x, as_float=False):
        if not as_float and (self._forward_hooks or self._forward_pre_hooks):
            # somebody watches this neuron through nn.Module hooks (the reference's tools do): take the module call, whose
            # hooks see the fp32 spikes of the reference's interface
            return ops.as_spikes(self(x))
        pf, self._prefired = self._prefired, None
        if (pf is not None and pf[0].data_ptr() == x.data_ptr() and pf[0].numel() == x.numel() and x.is_contiguous()
                and x._version == pf[1]):
            y = pf[2].view(x.shape)
            return y.float() if as_float else y
        v_in = None if isinstance(self.v, float) else self.v
        if self.stats is not None:
            self.stats_elems += x.numel()
        # A neuron that starts from a reset membrane, keeps none and records nothing is a pure function of its input: two
        # such neurons applied to the SAME tensor (the decoder's key / value neurons of the two layers that share a feature
        # level) share one kernel launch forward and one backward.  Cleared by reset_net().
        pure = (PURE_MEMO and v_in is None and not self.keep_membrane and self.stats is None and x.is_cuda
                and x.is_contiguous() and not as_float)
        if pure:
            key = (x.data_ptr(), x._version, x.numel(), x.requires_grad, self.D, self.v_threshold)
            hit = _PURE_MEMO.get(key)
            if hit is not None:
                return hit[1].view(x.shape).second()          # both tensors are contiguous: same address + size = same layout; the
                #                                               second consumer's gradient arrives on the neuron's spare handle
        y, v_out = ops.lif(x, v_in, self.D, self.v_threshold, self.keep_membrane, self.stats, spikes=not as_float)
        if pure:
            _PURE_MEMO[key] = (x, y)                      # holding x keeps its address from being reused
        if self.keep_membrane:
            self.v = v_out
        else:
            self.v = 0.0
        return y


class LIFNode(nn.Module):
    """Leaky integrate-and-fire node (neuron.py:694-814) under this fork's `BaseNode.forward` (:166-197: quantised multi-level
    firing, soft reset, output s / D) -- the neuron file's second node type.  No Spike2Former module instantiates it (SURVEY fact 3:
    the live path's Q_IFNode has no leak); it is here so that a model which swaps the node type still runs on the HIP kernels
    (csrc/lif.hip `lif_leaky_*`).  Charge: decay_input -> h = v + (x - v) / tau, otherwise h = v (1 - 1/tau) + x.
    Deliberately NOT a subclass of Q_IFNode: the fused producer kernels (fused.bn_act(lif=...), prefire) implement the leak-free
    charge and recognise their neuron by that type."""

    def __init__(self, tau=2.0, decay_input=True, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False,
                 cupy_fp32_inference=False):
        super().__init__()
        assert isinstance(tau, float) and tau > 1.0                       # neuron.py:792
        assert isinstance(v_threshold, float) and isinstance(detach_reset, bool)
        if detach_reset:
            raise NotImplementedError("detach_reset=True is not instantiated anywhere on the Spike2Former path")
        if not (v_reset is None or v_reset == 0.0):
            raise NotImplementedError("LIFNode: a non-zero v_reset changes the charge expression (neuron.py:808, 814); unused by the fork")
        self.tau, self.decay_input = tau, decay_input
        self.v_threshold, self.v_reset, self.detach_reset = v_threshold, v_reset, detach_reset
        self.surrogate_function = surrogate_function if surrogate_function is not None else Quant()
        self.D = getattr(self.surrogate_function, "D", 8)
        self.v = 0.0
        self.keep_membrane = True
        self.stats = None
        self.stats_elems = 0

    def reset(self):
        self.v = 0.0

    def extra_repr(self):
        return (f"v_threshold={self.v_threshold}, v_reset={self.v_reset}, detach_reset={self.detach_reset}, D={self.D}, "
                f"tau={self.tau}")

    def _apply(self, fn, *a, **k):
        if isinstance(self.v, torch.Tensor):
            self.v = fn(self.v)
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        v_in = None if isinstance(self.v, float) else self.v
        if self.stats is not None:
            self.stats_elems += x.numel()
        y, v_out = ops.lif_leaky(x, v_in, self.D, self.v_threshold, self.tau, self.decay_input, self.keep_membrane, self.stats)
        self.v = v_out if self.keep_membrane else 0.0
        return y

    def fire(self, x, as_float=False):
        y = self(x)
        return y if as_float else ops.as_spikes(y)


_PURE_MEMO = {}
PURE_MEMO = True          # share the launch of pure neurons applied to the same tensor (57.3 vs 57.7 ms/step at C2)


def reset_net(net: nn.Module):
    """functional.reset_net (functional.py:9-33): call `reset()` on every module that has one."""
    _PURE_MEMO.clear()
    for m in net.modules():
        if hasattr(m, "reset"):
            m.reset()
    p = next(net.parameters(), None)
    if p is not None and p.is_cuda:
        ops.begin_step(p.device)        # a reset opens a new step: rewind + clear the reduction-workspace arena


def set_keep_membrane(net: nn.Module, keep: bool):
    for m in net.modules():
        if isinstance(m, Q_IFNode):
            m.keep_membrane = keep
