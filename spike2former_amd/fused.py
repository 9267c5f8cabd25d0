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
    wanted_pre = want_pre
    if lif is not None and lif._forward_hooks:
        want_pre = True          # a forward hook receives the neuron's fp32 input: the kernel must write it out
    shape = z.shape
    N, C = shape[0], shape[1]
    L = z.numel() // max(N * C, 1)
    training = bn.training or (bn.running_mean is None)
    if z.numel() == 0:
        # nothing to normalise: ATen's own handling of the empty tensor (train mode raises, as in the reference)
        t = z if conv_bias is None else z + conv_bias.view(1, -1, *([1] * (z.dim() - 2)))
        u = F.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
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
    if not wanted_pre:
        u = None
    return (u, y, border) if want_border else (u, y)


class _EvalBN:
    """what bn_act reads of an eval-mode BatchNorm"""
    training, affine, momentum, num_batches_tracked = False, True, 0.0, None


def _composed_eval_pair(bn1, bn2):
    ts = (bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var)
    key = tuple((t.data_ptr(), t._version) for t in ts) + (bn1.eps, bn2.eps)
    hit = getattr(bn2, "_s2f_eval_pair", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        r2g2 = bn2.weight / torch.sqrt(bn2.running_var + bn2.eps)
        out = _EvalBN()
        out.weight = (bn1.weight * r2g2).contiguous()
        out.bias = ((bn1.bias - bn2.running_mean) * r2g2 + bn2.bias).contiguous()
        out.running_mean, out.running_var, out.eps = bn1.running_mean, bn1.running_var, bn1.eps
    try:
        bn2._s2f_eval_pair = (key, out)
    except AttributeError:          # (an object without a __dict__)
        pass
    return out


def _scaled_eval_bn(bn, scale):
    """scale * BN(.) of an eval-mode BatchNorm as ONE affine map: its running statistics with (scale gamma, scale beta) -- cached on
    the module by parameter version (inference: the parameters do not change between calls)."""
    ts = (bn.weight, bn.bias, scale)
    key = tuple((t.data_ptr(), t._version) for t in ts)
    hit = getattr(bn, "_s2f_eval_scaled", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        out = _EvalBN()
        out.weight = (bn.weight * scale).contiguous()
        out.bias = (bn.bias * scale).contiguous()
        out.running_mean, out.running_var, out.eps = bn.running_mean, bn.running_var, bn.eps
    try:
        bn._s2f_eval_scaled = (key, out)
    except AttributeError:
        pass
    return out


def bn_bn_act(z, bn1, bn2, residual=None, lif: Q_IFNode = None, want_pre=None, next_lif: Q_IFNode = None):
    """bn2(bn1(z)) [+ residual] [-> neuron]: the BatchNorm pair that closes a RepConv chain (Sequential(RepConv(.., BN), BN),
    sdtv2.py:280-296).  In training mode on the single-pass shapes (the 32x32-stage maps, where these chains live) the pair is ONE
    kernel forward and one backward (ops.bn2_act: the second BatchNorm's batch statistics follow from the first's); otherwise two
    bn_act calls.  Returns (u, y) as bn_act."""
    # (real modules only: the composed pair is cached on them by parameter version; the batched q / k / v chain hands in transient
    # concatenations of its twins' parameters, whose version counters say nothing)
    eval_both = (isinstance(bn1, torch.nn.Module) and isinstance(bn2, torch.nn.Module) and not bn1.training and not bn2.training
                 and bn1.running_mean is not None and bn2.running_mean is not None and bn1.weight is not None
                 and bn2.weight is not None and not torch.is_grad_enabled())
    if eval_both:
        # inference: two affine maps are one.  BN2(BN1(z)) = (z - m1) r1 [g1 r2 g2] + [(b1 - m2) r2 g2 + b2]: BN1's running statistics
        # with a composed (gamma, beta) -- one BatchNorm launch (the fold the reference ships for inference,
        # clock_driven/functional.py:574-692).  The composed pair is cached per module pair and parameter version.
        return bn_act(z, None, _composed_eval_pair(bn1, bn2), residual=residual, lif=lif, want_pre=want_pre, next_lif=next_lif)
    training = (bn1.training or bn1.running_mean is None) and (bn2.training or bn2.running_mean is None)
    momentum_ok = bn1.momentum is not None and bn2.momentum is not None
    if not (training and momentum_ok and ops.bn2_act_ok(z) and bn1.weight is not None and bn2.weight is not None
            and bn1.running_mean is not None and bn2.running_mean is not None):
        x, _ = bn_act(z, None, bn1)
        return bn_act(x, None, bn2, residual=residual, lif=lif, want_pre=want_pre, next_lif=next_lif)
    if next_lif is not None and lif is None:
        u, y = bn_bn_act(z, bn1, bn2, residual=residual, lif=next_lif, want_pre=True)
        next_lif.prefire(u, y)
        return u, None
    if want_pre is None:
        want_pre = lif is None
    wanted_pre = want_pre
    if lif is not None and lif._forward_hooks:
        want_pre = True
    v_in = None
    if lif is not None and not isinstance(lif.v, float):
        v_in = lif.v.detach()
    if lif is not None and lif.stats is not None:
        lif.stats_elems += z.numel()
    u, y, v_out = ops.bn2_act(z, bn1, bn2, residual=residual, lif=lif is not None, want_pre=want_pre, v_in=v_in,
                              keep_v=(lif is not None and lif.keep_membrane), D=(lif.D if lif is not None else 8),
                              vth=(lif.v_threshold if lif is not None else 1.0), stats=(lif.stats if lif is not None else None))
    if lif is not None:
        lif.v = v_out if lif.keep_membrane else 0.0
        if lif._forward_hooks:
            yf = y.float().detach()
            for hook in list(lif._forward_hooks.values()):
                hook(lif, (u,), yf)
    return (u if wanted_pre else None), y


def conv_bn_act(conv, x, bn, residual=None, lif: Q_IFNode = None, want_pre=None, next_lif: Q_IFNode = None, scale=None):
    """1x1 convolution (Conv2d / Conv1d of this package, fed by a neuron) -> BatchNorm [+ residual] [-> neuron]; x [N, K, *].
    In eval mode on bf16 spikes this is ONE launch -- the packed-weight GEMM with the BatchNorm (running statistics), the
    residual add and the neuron in its epilogue (ops.gemm_bn_lif_eval; the inference-time fold of SURVEY section 8 row f4,
    reference helpers clock_driven/functional.py:574-692): the fp32 convolution output never reaches HBM.  Everything else
    (training, fp32 inputs, shapes the kernel does not take, someone recording gradients) is conv.forward_nobias + bn_act.
    `scale`: per-channel factor on the BatchNorm output as in bn_act (the pixel decoder's layer scale): in eval mode folded into the
    affine pair once (cached), so that the layer stays one launch.  Returns (u, y) as bn_act."""
    shape_in = x.shape
    L = 1
    for d in shape_in[2:]:
        L *= d
    if scale is not None:
        if ((not bn.training) and bn.running_mean is not None and bn.affine and EVAL_FUSION
                and _no_grad_needed(x, conv, bn, residual) and not (torch.is_grad_enabled() and scale.requires_grad)):
            bn = _scaled_eval_bn(bn, scale)
        else:
            z = conv.forward_nobias(x, stats=bool(bn.training or bn.running_mean is None))
            return bn_act(z, conv.bias, bn, residual=residual, lif=lif, want_pre=want_pre, next_lif=next_lif, scale=scale)
    fire = next_lif if (next_lif is not None and lif is None) else lif
    if want_pre is None:
        want_pre = lif is None
    # eval mode, depthwise stencil -> BatchNorm -> [neuron]: the BatchNorm and the neuron run in the stencil's store (row f4)
    if (x.dim() == 4 and getattr(conv, "groups", 1) == conv.in_channels == conv.out_channels and conv.groups > 1
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] in (3, 5, 7) and conv.bias is None
            and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.padding[0] == conv.padding[1]
            and (not bn.training) and bn.running_mean is not None and bn.affine and EVAL_FUSION and residual is None and x.is_cuda
            and _no_grad_needed(x, conv, bn)
            and (fire is None or (ops.spikes_bf16_ok(fire.D) and isinstance(fire.v, float) and not fire.keep_membrane
                                  and not fire._forward_pre_hooks and (L % 4 == 0)))):
        if fire is not None and fire.stats is not None:
            fire.stats_elems += shape_in[0] * conv.out_channels * L
        pre = bool(want_pre or (next_lif is not None and lif is None) or (fire is not None and bool(fire._forward_hooks)))
        u, y = ops.dwconv_bn_lif_eval(x, conv.weight, conv.padding[0], bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps,
                                      want_pre=pre or fire is None, lif=fire is not None, D=(fire.D if fire is not None else 8),
                                      vth=(fire.v_threshold if fire is not None else 1.0), stats=(fire.stats if fire is not None else None))
        if fire is not None:
            fire.v = 0.0
            if fire._forward_hooks:
                yf = y.float().detach()
                for hook in list(fire._forward_hooks.values()):
                    hook(fire, (u,), yf)
        if next_lif is not None and lif is None:
            next_lif.prefire(u, y)
            return u, None
        return (u if (want_pre or fire is None) else None), y
    pure_conv = (conv.kernel_size in ((1, 1), (1,)) and conv.groups == 1 and tuple(conv.stride) in ((1, 1), (1,))
                 and tuple(conv.padding) in ((0, 0), (0,)))
    # 3x3 / stride 1 / padding 1 without a bias on a map the implicit kernel takes (MS_ConvBlock's convolutions)
    conv3 = (conv.kernel_size == (3, 3) and conv.groups == 1 and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
             and tuple(conv.dilation) == (1, 1) and conv.bias is None and x.dim() == 4 and conv.in_channels % 32 == 0
             and x.shape[-1] % 4 == 0 and ops.cfg.PGEMM_CONV)
    eval_bn = (not bn.training) and bn.running_mean is not None and bn.affine
    # a 1x1 convolution of a DENSE fp32 map (SepConv.pwconv2 behind the depthwise stencil, the stem's column matrix): the 6-pass
    # product with the same epilogue (ops.dense_gemm_bn_lif_eval); reset neurons only
    dense = (pure_conv and eval_bn and EVAL_FUSION and ops.dense_gemm_bn_lif_eval_ok(x, L) and _no_grad_needed(x, conv, bn, residual)
             and (fire is None or (ops.spikes_bf16_ok(fire.D) and not fire._forward_pre_hooks and isinstance(fire.v, float)
                                   and not fire.keep_membrane)))
    if not dense and not ((pure_conv or conv3) and eval_bn and EVAL_FUSION and ops.gemm_bn_lif_eval_ok(x, L)
                          and _no_grad_needed(x, conv, bn, residual)
                          and (fire is None or (ops.spikes_bf16_ok(fire.D) and not fire._forward_pre_hooks))):
        z = conv.forward_nobias(x, stats=bool(bn.training or bn.running_mean is None))
        return bn_act(z, conv.bias, bn, residual=residual, lif=lif, want_pre=want_pre, next_lif=next_lif)
    M = conv.out_channels
    v_in = None
    if fire is not None and not isinstance(fire.v, float):
        v_in = fire.v.detach()
    if fire is not None and fire.stats is not None:
        fire.stats_elems += shape_in[0] * M * L
    kw = dict(want_pre=bool(want_pre or (next_lif is not None and lif is None) or (fire is not None and bool(fire._forward_hooks))),
              lif=fire is not None, v_in=v_in, keep_v=(fire is not None and fire.keep_membrane), D=(fire.D if fire is not None else 8),
              vth=(fire.v_threshold if fire is not None else 1.0), stats=(fire.stats if fire is not None else None))
    if dense:
        u, y = ops.dense_gemm_bn_lif_eval(
            x.reshape(shape_in[0], shape_in[1], L), conv.weight.view(M, -1), conv.bias, bn.running_mean, bn.running_var, bn.weight,
            bn.bias, bn.eps, residual=None if residual is None else residual.reshape(shape_in[0], M, L), want_pre=kw["want_pre"],
            lif=kw["lif"], D=kw["D"], vth=kw["vth"], stats=kw["stats"])
        v_out = None
    elif conv3:
        u, y, v_out = ops.conv3x3_bn_lif_eval(x, conv.weight, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps,
                                              residual=None if residual is None else residual.reshape(shape_in[0], M, *shape_in[2:]), **kw)
    else:
        u, y, v_out = ops.gemm_bn_lif_eval(
            x.reshape(shape_in[0], shape_in[1], L), conv.weight.view(M, -1), conv.bias, bn.running_mean, bn.running_var, bn.weight,
            bn.bias, bn.eps, residual=None if residual is None else residual.reshape(shape_in[0], M, L), **kw)
    out_shape = (shape_in[0], M) + tuple(shape_in[2:])
    if u is not None:
        u = u.view(out_shape)
    if y is not None:
        y = y.view(*out_shape)
    if fire is not None:
        fire.v = v_out.view(out_shape) if fire.keep_membrane else 0.0
        if fire._forward_hooks:
            yf = y.float().detach()
            for hook in list(fire._forward_hooks.values()):
                hook(fire, (u,), yf)
    if next_lif is not None and lif is None:
        next_lif.prefire(u, y)
        return u, None
    return (u if want_pre else None), y


def _no_grad_needed(x, conv, bn, residual=None):
    """The eval-mode fusions build no autograd graph: they are only legal when nothing that feeds them wants a gradient -- the
    input, the residual, AND the trainable parameters (a frozen backbone feeding a trainable head under bn.eval(), norm_eval
    fine-tuning: the convolution's weight and the BatchNorm affine must still receive gradients)."""
    if not torch.is_grad_enabled():
        return True
    if x.requires_grad:                      # a tensor or ops.Spikes (its token carries the flag)
        return False
    return not any(t is not None and t.requires_grad for t in (residual, conv.weight, conv.bias, bn.weight, bn.bias))


EVAL_FUSION = True          # eval-mode conv + BatchNorm + neuron as one GEMM launch (False: the two-kernel path, for A/B and tests)
