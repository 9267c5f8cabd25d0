"""Module-level glue for the fused BatchNorm(+bias)(+residual)(+Q_IFNode) op.  The nn.BatchNorm / Conv / Q_IFNode
modules stay in the module tree as parameter and state holders (the checkpoint ABI); their arithmetic is done here."""
import torch
import torch.nn.functional as F

from . import ops
from .neuron import Q_IFNode


def bn_act(z, conv_bias, bn, residual=None, lif: Q_IFNode = None, want_pre=None, next_lif: Q_IFNode = None,
           want_border=False, scale=None):
    """z: conv output WITHOUT its bias, [N, C, *].  Returns (u, y): u = BN(z + bias) [+ residual] (None unless wanted),
    y = lif(u) as an ops.Spikes pair (None without lif).  Shapes follow z.  want_border: also return BN(0) from the running statistics as
    updated by this call (BNAndPadLayer's padding value, sdtv2.py:68-78) -- written by the same kernel.
    next_lif: the neuron that the caller's consumer will apply to `u` next (the first Q_IFNode of the following block on
    the residual stream).  Its update is done by this kernel as well and handed over with Q_IFNode.prefire: the reference's
    separate neuron pass over u (one more read of u forward; a neuron backward + a gradient add backward) disappears.
    scale: per-channel factor on the BatchNorm output, u = scale * BN(z + bias) [+ residual] (the layer-scale `gamma` of the
    pixel decoder's encoder layers, detr_layers.py:331-337), folded into the affine pair: two [C] products, no pass over u."""
    if next_lif is not None and lif is None:
        u, y = bn_act(z, conv_bias, bn, residual=residual, lif=next_lif, want_pre=True, scale=scale)
        next_lif.prefire(u, y)
        return u, None
    if want_pre is None:
        want_pre = lif is None
    shape = z.shape
    N, C = shape[0], shape[1]
    L = z.numel() // max(N * C, 1)
    training = bn.training or (bn.running_mean is None)
    if L % 4 != 0 or z.numel() == 0:
        # odd row length: unfused ATen BatchNorm (native HIP kernels, MIOpen is disabled) + the stand-alone neuron kernel
        t = z if conv_bias is None else z + conv_bias.view(1, -1, *([1] * (z.dim() - 2)))
        u = F.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
        if scale is not None:
            u = u * scale.view(1, -1, *([1] * (z.dim() - 2)))
        if residual is not None:
            u = u + residual.reshape(shape)
        out = (u if want_pre else None), (lif.fire(u) if lif is not None else None)
        if want_border:
            out += ((bn.bias.detach() - bn.running_mean * bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)),)
        return out
    v_in = None
    if lif is not None and not isinstance(lif.v, float):
        # the carried membrane enters as a constant: the fused kernel has no gradient path into it (ResetModelHook zeroes it
        # before every training iteration, resetmodel_hook.py:17-37; the stand-alone neuron op does back-propagate through it)
        v_in = lif.v.detach()
    if lif is not None and lif.stats is not None:
        lif.stats_elems += z.numel()
    weight, bias = (bn.weight, bn.bias) if scale is None else ops.scale_affine(bn.weight, bn.bias, scale)
    u, y, v_out, border = ops.bn_act(
        z, conv_bias, weight, bias, bn.running_mean, bn.running_var,
        bn.num_batches_tracked if training else None, training, bn.momentum, bn.eps,
        residual=residual, lif=lif is not None, want_pre=want_pre, v_in=v_in,
        keep_v=(lif is not None and lif.keep_membrane), D=(lif.D if lif is not None else 8),
        vth=(lif.v_threshold if lif is not None else 1.0), stats=(lif.stats if lif is not None else None), want_border=True)
    if lif is not None:
        lif.v = v_out if lif.keep_membrane else 0.0
        if lif._forward_hooks:
            # the neuron ran inside the BatchNorm kernel, not through its module call: anyone watching it with nn.Module
            # forward hooks (the reference's cal_firing_num.py does) still sees (module, (input,), fp32 spikes)
            yf = y.float().detach()
            for hook in list(lif._forward_hooks.values()):
                hook(lif, (u,), yf)
    return (u, y, border) if want_border else (u, y)
