"""Inference-time re-parameterisation helpers, mirroring the reference's conv + BatchNorm fold
(Qtrick_architecture/clock_driven/functional.py:574-692: fused_conv2d_weight_of_convbn2d, fused_conv2d_bias_of_convbn2d,
fuse_convbn2d -- same names, same arguments, same results).

On this path the fold is not a separate model-surgery step: in eval mode every  conv1x1 -> BatchNorm [+ residual] -> Q_IFNode
chain fed by a neuron runs as ONE kernel (fused.conv_bn_act -> s2f_gemm_bn_lif_fwd), which applies the BatchNorm's
per-channel affine pair -- exactly the (weight scale, bias) these helpers compute -- in the GEMM epilogue, with the
arithmetic of the unfused kernels (bit-identical spikes).  The helpers below produce the folded tensors for callers that
want them explicitly (export, checks against the reference)."""
import torch
import torch.nn as nn


def _scale(bn):
    return bn.weight / (bn.running_var + bn.eps).sqrt()


def fused_conv2d_weight_of_convbn2d(conv2d: nn.Conv2d, bn2d: nn.BatchNorm2d):
    """Weight of the Conv2d that equals {Conv2d (no bias) -> BatchNorm2d (running statistics)}: w[o] * gamma[o] / sqrt(var[o] + eps)."""
    assert conv2d.bias is None
    return conv2d.weight * _scale(bn2d).view(-1, 1, 1, 1)


def fused_conv2d_bias_of_convbn2d(conv2d: nn.Conv2d, bn2d: nn.BatchNorm2d):
    """Bias of that Conv2d: beta - running_mean * gamma / sqrt(var + eps)."""
    assert conv2d.bias is None
    return bn2d.bias - bn2d.running_mean * _scale(bn2d)


@torch.no_grad()
def fuse_convbn2d(conv2d: nn.Conv2d, bn2d: nn.BatchNorm2d, k=None, b=None):
    """-> the fused Conv2d (a module of the same class as `conv2d`, with a bias)."""
    fused = type(conv2d)(conv2d.in_channels, conv2d.out_channels, conv2d.kernel_size, conv2d.stride, conv2d.padding,
                         conv2d.dilation, conv2d.groups, bias=True, padding_mode=conv2d.padding_mode)
    fused = fused.to(conv2d.weight.device)
    fused.weight.data = fused_conv2d_weight_of_convbn2d(conv2d, bn2d)
    fused.bias.data = fused_conv2d_bias_of_convbn2d(conv2d, bn2d)
    if getattr(conv2d, "spike_input", False):
        fused.spike_input = True
    return fused
