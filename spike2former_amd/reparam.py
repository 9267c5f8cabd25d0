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


@torch.no_grad()
def merge_repconv(rep, outer_bn=None):
    """The dense 3x3 convolution (zero padding 1) + bias that equals an eval-mode RepConv [+ the BatchNorm behind it]
    (sdtv2.py:112-132 with 48-89: conv1x1 -> BNAndPad -> depthwise 3x3 -> conv1x1 -> BatchNorm [-> BatchNorm]).  Everything between
    the two neurons is linear, and BNAndPadLayer pads with BN_1(0), i.e. with what BN_1 gives where the ZERO-padded input's 1x1
    image is zero -- so the chain is a zero-padded 3x3 convolution:
        K[m, c, ky, kx] = sum_j W2'[m, j] Wd[j, ky, kx] W1'[j, c],      bias[m] = sum_j W2'[m, j] b1[j] sum_taps Wd[j] + b2[m]
    with W1' = diag(s1) W1, b1 = BN_1's shift, W2' / b2 = the second 1x1 with the closing BatchNorm(s) folded in.  Formed in fp64.
    -> (weight [M, C, 3, 3] fp32, bias [M] fp32).  NINE times the multiply-adds of a 1x1: measured 4x slower than the three-launch
    chain on the C2 maps (tools/probe_repconv_merge.py) -- an export helper, not used on the product path."""
    c1, bnp, (dw, c2, bn2) = rep.body[0], rep.body[1].bn, rep.body[2]
    d = torch.float64
    s1 = (bnp.weight.to(d) / (bnp.running_var.to(d) + bnp.eps).sqrt())
    b1 = bnp.bias.to(d) - bnp.running_mean.to(d) * s1
    w1 = c1.weight.to(d).flatten(1) * s1.view(-1, 1)                                     # [J, C]
    s2 = bn2.weight.to(d) / (bn2.running_var.to(d) + bn2.eps).sqrt()
    b2 = bn2.bias.to(d) - bn2.running_mean.to(d) * s2
    if outer_bn is not None:
        s3 = outer_bn.weight.to(d) / (outer_bn.running_var.to(d) + outer_bn.eps).sqrt()
        b2 = (b2 - outer_bn.running_mean.to(d)) * s3 + outer_bn.bias.to(d)
        s2 = s2 * s3
    w2 = c2.weight.to(d).flatten(1) * s2.view(-1, 1)                                     # [M, J]
    wd = dw.weight.to(d).view(-1, 3, 3)                                                  # [J, 3, 3]
    k = torch.einsum("mj,jyx,jc->mcyx", w2, wd, w1)
    bias = w2 @ (b1 * wd.sum((1, 2))) + b2
    return k.float().contiguous(), bias.float().contiguous()
