"""Convolution plumbing without MIOpen.

This image ships no gfx950 MIOpen find-db / kernel cache, so every MIOpen convolution JIT-compiles (and, in benchmark
mode, exhaustively searches) its kernels on a fresh box: the first C2 step took > 15 minutes.  The path therefore never
touches MIOpen (`torch.backends.cudnn.enabled = False`, set when the package is imported):

  * 1x1 Conv2d / Conv1d(k=1) (the bulk of the path's FLOPs: q/k/v, MLP, pixel-decoder and decoder projections) are
    plain GEMMs  Y[n] = W @ X[n]  on the channel-major activations -> this package's packed-weight kernels (csrc/pgemm.hip;
    rounds 1-2: rocBLAS);
  * dense kxk convolutions (stem 7x7, MS_ConvBlock 3x3, downsampling 3x3/s2) are lowered to im2col + the same GEMM;
  * depthwise convolutions and BatchNorm are this package's own kernels (csrc/dwconv.hip, csrc/bn_lif.hip).

A convolution fed by a neuron takes the spike map as an `ops.Spikes` pair (bf16 data + autograd handle) and runs on the
bf16 matrix cores (csrc/gemm_bf16.hip); a plain fp32 tensor of spikes is accepted as well.

The classes subclass nn.Conv2d / nn.Conv1d, so parameter names and shapes -- the checkpoint ABI -- are unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

torch.backends.cudnn.enabled = False        # = MIOpen on ROCm; see module docstring


def spikes_in(*convs):
    """Mark convolutions whose input is produced by a Q_IFNode (exact in bf16 -> eligible for the spike GEMM)."""
    for c in convs:
        c.spike_input = True


def _gemm_nc(weight2d, x3, bias, spike_input=False, stats=False):
    """x3 [N, K, L] (tensor or ops.Spikes), weight2d [M, K] -> [N, M, L].  stats: the module is in training mode -- a BatchNorm
    that follows can take its statistics from the product's epilogue (ops.BN_PARTIALS)."""
    if (spike_input or isinstance(x3, ops.Spikes)) and ops.SPIKE_GEMM_ENABLED and x3.shape[2] % 4 == 0:
        return ops.spike_gemm(x3, weight2d, bias, stats=stats)       # bf16 matrix cores, exact for spike activations
    # general fp32 input: ops.dense_gemm (6-pass packed-weight kernel; the library GEMM only for shapes it does not take)
    y = ops.dense_gemm(ops.spikes_float(x3), weight2d, stats=stats and bias is None)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y


class Conv2d(nn.Conv2d):
    spike_input = False       # set by the owning module when the input is a Q_IFNode output (multiples of 1/D)

    def forward(self, x, border=None):
        return self._conv(x, self.bias, border)

    def forward_nobias(self, x, stats=None):
        """The convolution without its bias (the fused BatchNorm kernels add the bias on the fly).  `stats`: whether the consumer is a
        train-mode BatchNorm that wants the per-tile statistics out of the GEMM's epilogue (None: this module's own training flag;
        fused.conv_bn_act passes the BatchNorm's, so that conv.train() + bn.eval() does not pay for partials nobody reads)."""
        return self._conv(x, None, stats=stats)

    def _conv(self, x, bias, border=None, stats=None):
        if x.device.type != "cuda":
            raise RuntimeError("spike2former_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
        if self.groups != 1:
            kh, kw = self.kernel_size
            if (self.groups == self.in_channels == self.out_channels and kh == kw and kh in (3, 5, 7) and bias is None
                    and self.stride == (1, 1) and self.dilation == (1, 1) and self.padding[0] == self.padding[1]):
                return ops.dwconv(x, self.weight, self.padding[0], border)      # hand-written depthwise stencil
            assert border is None
            ops.fallback("grouped Conv2d", f"groups={self.groups} k={self.kernel_size}")
            return F.conv2d(ops.spikes_float(x), self.weight, bias, self.stride, self.padding, self.dilation, self.groups)
        N, C, H, W = x.shape
        M = self.out_channels
        kh, kw = self.kernel_size
        stats = (self.training if stats is None else bool(stats)) and bias is None
        if kh == 1 and kw == 1 and self.stride == (1, 1) and self.padding == (0, 0):
            y = _gemm_nc(self.weight.view(M, C), x.reshape(N, C, H * W), bias, self.spike_input, stats)
            return ops.carry_stats(y, y.view(N, M, H, W))
        if (self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and self.dilation == (1, 1)):
            return ops.conv_dense(x, self.weight, bias, self.stride[0], self.padding[0], self.spike_input, stats=stats)
        Ho = (H + 2 * self.padding[0] - self.dilation[0] * (kh - 1) - 1) // self.stride[0] + 1
        Wo = (W + 2 * self.padding[1] - self.dilation[1] * (kw - 1) - 1) // self.stride[1] + 1
        cols = F.unfold(ops.spikes_float(x), (kh, kw), self.dilation, self.padding, self.stride)          # [N, C*kh*kw, Ho*Wo]
        return _gemm_nc(self.weight.view(M, -1), cols, bias, self.spike_input).view(N, M, Ho, Wo)


class Conv1d(nn.Conv1d):
    spike_input = False

    def forward(self, x):
        return self._conv(x, self.bias)

    def forward_nobias(self, x, stats=None):
        return self._conv(x, None, stats=stats)

    def _conv(self, x, bias, stats=None):
        if x.device.type != "cuda":
            raise RuntimeError("spike2former_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
        assert self.kernel_size == (1,) and self.stride == (1,) and self.groups == 1, "only k=1 Conv1d is on the path"
        return _gemm_nc(self.weight.view(self.out_channels, -1), x, bias, self.spike_input,
                        (self.training if stats is None else bool(stats)) and bias is None)
