"""Meta-SpikeFormer (SDT-v2) backbone on the MI355X kernels.

Mirrors the registry surface of the reference backbone -- class names, constructor kwargs and state_dict keys of
mmseg/models/backbones/sdtv2.py:48-655 -- so reference configs and checkpoints load unchanged.  Every Q_IFNode, BatchNorm, depthwise
stencil, spike GEMM and the softmax-free attention core run in libs2f_hip.so (csrc/*.hip); spike maps travel between them
as bf16 (`ops.Spikes`); the input gradients of the 1x1 and 3x3 convolutions are this package's 6-pass kernels too (csrc/pgemm.hip).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, ops
from .conv import Conv1d, Conv2d, spikes_in
from .fused import bn_act, bn_bn_act, conv_bn_act
from .neuron import Q_IFNode, Quant
from .registry import MODELS


# the q / k / v projection chains of an attention block as one 3C-channel chain (pure neurons only)
QKV_BATCHED = True


def _lif():
    return Q_IFNode(surrogate_function=Quant())


class BNAndPadLayer(nn.Module):
    """BatchNorm2d, then a 1-pixel border filled with BN(0) computed from the *running* statistics
    (sdtv2.py:48-89) -- also in training, where the interior is normalised with batch statistics."""

    def __init__(self, pad_pixels, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm2d(num_features, eps, momentum, affine, track_running_stats)
        self.pad_pixels = pad_pixels

    def border_values(self):
        """BN(0) from the running statistics, detached (sdtv2.py:68-78)."""
        bn = self.bn
        std = torch.sqrt(bn.running_var + bn.eps)
        if bn.affine:
            return bn.bias.detach() - bn.running_mean * bn.weight.detach() / std
        return -bn.running_mean / std

    def pad(self, out, border):
        p = self.pad_pixels
        N, C, H, W = out.shape
        full = border.view(1, C, 1, 1).expand(N, C, H + 2 * p, W + 2 * p).clone()
        full[:, :, p:-p, p:-p] = out
        return full

    def forward(self, x, return_border=False):
        # the kernel updates the running statistics first and derives the border from them, as the reference does
        out, _, border = bn_act(x, None, self.bn, want_border=True)
        if self.pad_pixels == 0:
            return (out, None) if return_border else out
        if self.bn.running_mean is None or not self.bn.affine:
            border = self.border_values()
        if return_border:                            # RepConv feeds the border to the depthwise kernel directly
            return out, border
        return self.pad(out, border)

    weight = property(lambda self: self.bn.weight)
    bias = property(lambda self: self.bn.bias)
    running_mean = property(lambda self: self.bn.running_mean)
    running_var = property(lambda self: self.bn.running_var)
    eps = property(lambda self: self.bn.eps)


class RepConv(nn.Module):
    """conv1x1 -> BNAndPad -> depthwise 3x3 (no padding) -> conv1x1 -> BN  (sdtv2.py:112-132); not re-parameterised
    at train time, as in the reference."""

    def __init__(self, in_channel, out_channel, bias=False):
        super().__init__()
        conv1x1 = Conv2d(in_channel, in_channel, 1, 1, 0, bias=False, groups=1)
        bn = BNAndPadLayer(pad_pixels=1, num_features=in_channel)
        conv3x3 = nn.Sequential(
            Conv2d(in_channel, in_channel, 3, 1, 0, groups=in_channel, bias=False),
            Conv2d(in_channel, out_channel, 1, 1, 0, groups=1, bias=False),
            nn.BatchNorm2d(out_channel))
        self.body = nn.Sequential(conv1x1, bn, conv3x3)
        spikes_in(conv1x1)                      # RepConv is always fed by a neuron (head_spike / attn_spike)

    def forward(self, x, outer_bn=None, lif=None, residual=None, next_lif=None):
        """conv1x1 -> BN+pad -> dw3x3 -> conv1x1 -> BN [-> outer BN [+ residual] [-> neuron]].
        Returns (pre-activation or None, spikes or None) when `outer_bn` is given, else the tensor."""
        dw = self.body[2][0]
        bn1 = self.body[1].bn
        if (outer_bn is not None and dw.kernel_size == (3, 3) and self.body[1].pad_pixels == 1 and not bn1.training
                and bn1.running_mean is not None and bn1.affine and isinstance(outer_bn, nn.Module) and not outer_bn.training
                and not self.body[2][2].training and outer_bn.running_mean is not None and self.body[2][2].running_mean is not None
                and fused.EVAL_FUSION and not torch.is_grad_enabled() and x.dim() == 4
                and ops.gemm_bn_lif_eval_ok(x, x.shape[2] * x.shape[3])):
            # inference (row f4): BatchNorm_1 rides in the first GEMM's epilogue, its border value BN_1(0) is cached, and the
            # BatchNorm pair that closes the chain [+ residual] [-> neuron] rides in the second (dense) GEMM's: six launches -> three
            N_, C_, H_, W_ = x.shape
            z = ops.gemm_bn_lif_eval(x.reshape(N_, C_, H_ * W_), self.body[0].weight.view(self.body[0].out_channels, -1), None,
                                     bn1.running_mean, bn1.running_var, bn1.weight, bn1.bias, bn1.eps, want_pre=True, lif=False)[0]
            z = ops.dwconv(z.view(N_, -1, H_, W_), dw.weight, 1, self._eval_border())
            return conv_bn_act(self.body[2][1], z, fused._composed_eval_pair(self.body[2][2], outer_bn), residual=residual, lif=lif,
                               next_lif=next_lif)
        x, border = self.body[1](self.body[0](x), return_border=True)
        # un-padded depthwise 3x3 over the constant-bordered map == pad-1 stencil reading `border` outside the plane
        x = ops.dwconv(x, dw.weight, dw.kernel_size[0] // 2, border) if dw.kernel_size == (3, 3) else dw(self.body[1].pad(x, border))
        x = self.body[2][1](x)
        if outer_bn is None:
            return bn_act(x, None, self.body[2][2])[0]
        return bn_bn_act(x, self.body[2][2], outer_bn, residual=residual, lif=lif, next_lif=next_lif)

    def _eval_border(self):
        """BN_1(0) from the running statistics (the value the depthwise stencil reads outside the plane), cached by parameter version"""
        bn = self.body[1].bn
        key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        hit = getattr(self, "_s2f_eval_border", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            border = (bn.bias - bn.running_mean * bn.weight / torch.sqrt(bn.running_var + bn.eps)).contiguous()
        self._s2f_eval_border = (key, border)
        return border


class SepConv(nn.Module):
    """LIF -> pw 1x1 (C->2C) -> BN -> LIF -> dw 7x7 -> pw 1x1 (2C->C) -> BN   (sdtv2.py:135-180)."""

    def __init__(self, dim, expansion_ratio=2, act2_layer=nn.Identity, bias=False, kernel_size=7, padding=3, T=None):
        super().__init__()
        med = int(expansion_ratio * dim)
        self.spike1 = _lif()
        self.pwconv1 = Conv2d(dim, med, kernel_size=1, stride=1, bias=bias)
        self.bn1 = nn.BatchNorm2d(med)
        self.spike2 = _lif()
        self.dwconv = Conv2d(med, med, kernel_size=kernel_size, padding=padding, groups=med, bias=bias)
        self.pwconv2 = Conv2d(med, dim, kernel_size=1, stride=1, bias=bias)
        self.bn2 = nn.BatchNorm2d(dim)
        spikes_in(self.pwconv1)                 # pwconv2 reads the depthwise output (not spikes)

    def forward(self, x, residual=None, next_lif=None):
        """Returns SepConv(x) [+ residual] (the residual add is fused into the last BatchNorm kernel; `next_lif`, the
        neuron that reads the result next, is applied there as well -- fused.bn_act)."""
        T, B, C, H, W = x.shape
        s = self.spike1.fire(x)
        _, s = conv_bn_act(self.pwconv1, s.flatten(0, 1), self.bn1, lif=self.spike2)
        # (conv_bn_act: in eval mode the dense 1x1 + BatchNorm + residual + next neuron are one launch; training: forward_nobias + bn_act)
        u, _ = conv_bn_act(self.pwconv2, self.dwconv(s), self.bn2, residual=None if residual is None else residual.flatten(0, 1),
                           next_lif=next_lif)
        return u.reshape(T, B, C, H, W)


class MS_ConvBlock(nn.Module):
    """x += SepConv(x);  x += BN(conv3x3(LIF(BN(conv3x3(LIF(x))))))   (sdtv2.py:183-219)."""

    def __init__(self, dim, mlp_ratio=4.0, T=4):
        super().__init__()
        self.T = T
        self.Conv = SepConv(dim=dim)
        self.mlp_ratio = mlp_ratio
        self.spike1 = _lif()
        self.conv1 = Conv2d(dim, dim * mlp_ratio, kernel_size=3, padding=1, groups=1, bias=False)
        self.bn1 = nn.BatchNorm2d(dim * mlp_ratio)
        self.spike2 = _lif()
        self.conv2 = Conv2d(dim * mlp_ratio, dim, kernel_size=3, padding=1, groups=1, bias=False)
        self.bn2 = nn.BatchNorm2d(dim)
        spikes_in(self.conv1, self.conv2)

    @property
    def first_lif(self):
        return self.Conv.spike1

    def forward(self, x, next_lif=None):
        T, B, C, H, W = x.shape
        feat = self.Conv(x, residual=x, next_lif=self.spike1)            # x + SepConv(x)
        s = self.spike1.fire(feat)
        # (conv_bn_act: in eval mode conv3x3 + BatchNorm + neuron are one launch; in training conv.forward_nobias + bn_act)
        _, s = conv_bn_act(self.conv1, s.flatten(0, 1), self.bn1, lif=self.spike2)
        u, _ = conv_bn_act(self.conv2, s, self.bn2, residual=feat.flatten(0, 1), next_lif=next_lif)   # feat + BN(conv2(.))
        return u.reshape(T, B, C, H, W)


class MS_MLP(nn.Module):
    """LIF -> Conv1d(C->4C) -> BN1d -> LIF -> Conv1d(4C->C) -> BN1d on [T,B,C,N]   (sdtv2.py:222-255)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0, layer=0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1_conv = Conv1d(in_features, hidden_features, kernel_size=1, stride=1)
        self.fc1_bn = nn.BatchNorm1d(hidden_features)
        self.fc1_spike = _lif()
        self.fc2_conv = Conv1d(hidden_features, out_features, kernel_size=1, stride=1)
        self.fc2_bn = nn.BatchNorm1d(out_features)
        self.fc2_spike = _lif()
        self.c_hidden = hidden_features
        self.c_output = out_features
        spikes_in(self.fc1_conv, self.fc2_conv)

    def forward(self, x, residual=None, next_lif=None):
        T, B, C, H, W = x.shape
        s = self.fc1_spike.fire(x.flatten(3)).flatten(0, 1)
        _, s = conv_bn_act(self.fc1_conv, s, self.fc1_bn, lif=self.fc2_spike)
        res = None if residual is None else residual.reshape(T * B, C, H * W)
        u, _ = conv_bn_act(self.fc2_conv, s, self.fc2_bn, residual=res, next_lif=next_lif)
        return u.reshape(T, B, C, H, W)


class MS_Attention_RepConv_qkv_id(nn.Module):
    """Spike-driven self-attention (sdtv2.py:258-344): q,k,v = LIF(BN(RepConv(LIF(x)))); o = scale * q (k^T v);
    proj(LIF(o)).  The core runs as ops.sdsa directly on the channel-major spikes (no head permutes)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, sr_ratio=1,
                 T=None):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim = dim
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.head_spike = _lif()
        self.q_conv = nn.Sequential(RepConv(dim, dim, bias=False), nn.BatchNorm2d(dim))
        self.k_conv = nn.Sequential(RepConv(dim, dim, bias=False), nn.BatchNorm2d(dim))
        self.v_conv = nn.Sequential(RepConv(dim, dim, bias=False), nn.BatchNorm2d(dim))
        self.q_spike = _lif()
        self.k_spike = _lif()
        self.v_spike = _lif()
        self.attn_spike = _lif()
        self.proj_conv = nn.Sequential(RepConv(dim, dim, bias=False), nn.BatchNorm2d(dim))
        self._flatten_qkv()

    # ---- q / k / v as ONE chain of 3C channels ---------------------------------------------------------------------------
    # The three projection chains read the same spikes and have the same shapes; run separately each of their 18 forward
    # (and ~32 backward) kernels is a 128-256-workgroup launch on a 32x32 map.  BatchNorm, the depthwise stencil and the
    # neuron are per-channel, so on the channel-concatenated tensor they ARE the three separate ops; the first 1x1 conv is
    # one GEMM with the stacked weight, the second a 3-group batched GEMM.  The parameters stay the reference's separate
    # tensors (state_dict keys unchanged); `_flatten_qkv` lays the twins out back to back so that the concatenated views
    # exist without copies (ops.cat_params falls back to torch.cat when something re-allocated them).
    def _twins(self):
        convs = (self.q_conv, self.k_conv, self.v_conv)
        g = lambda f: [f(c) for c in convs]                               # noqa: E731
        bn1, bn2, bn3 = g(lambda c: c[0].body[1].bn), g(lambda c: c[0].body[2][2]), g(lambda c: c[1])
        return dict(w1=g(lambda c: c[0].body[0].weight), dw=g(lambda c: c[0].body[2][0].weight),
                    w2=g(lambda c: c[0].body[2][1].weight), bns=(bn1, bn2, bn3))

    def _flatten_qkv(self):
        t = self._twins()
        for ws in (t["w1"], t["dw"], t["w2"]):
            ops.flatten_together(ws)
        for bns in t["bns"]:
            for name in ("weight", "bias", "running_mean", "running_var"):
                ops.flatten_together([getattr(b, name) for b in bns])

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)            # .to() / .cuda() re-allocate every tensor: lay the twins out again
        self._flatten_qkv()
        return self

    def _qkv_batched(self, s, T, B, C, H, W):
        t = self._twins()
        N = H * W
        training = self.q_conv[1].training
        out = []

        class CatBN:                            # what fused.bn_act reads of a BatchNorm module
            def __init__(bn, mods):
                bn.weight, bn.bias = ops.cat_params([m.weight for m in mods]), ops.cat_params([m.bias for m in mods])
                bn.running_mean, wb1 = ops.cat_buffers([m.running_mean for m in mods])
                bn.running_var, wb2 = ops.cat_buffers([m.running_var for m in mods])
                bn.num_batches_tracked = None   # counted below, once for all nine
                bn.training, bn.momentum, bn.eps = mods[0].training, mods[0].momentum, mods[0].eps
                out.append(wb1); out.append(wb2)

        bn1, bn2, bn3 = (CatBN(m) for m in t["bns"])
        w1 = ops.cat_params(t["w1"]).view(3 * C, C)
        w1._s2f_version = sum(p._version for p in t["w1"])              # for the cached bf16 split (ops.split_weight)
        w1._s2f_owner = t["w1"][0]
        sv = s.view(T * B, C, N)
        if (not training) and fused.EVAL_FUSION and not torch.is_grad_enabled() and ops.gemm_bn_lif_eval_ok(sv, N):
            # inference: BatchNorm_1 (running statistics) rides in the first GEMM's epilogue, and the
            # BatchNorm pair that closes the chain is ONE affine map (composed on the host, cached per parameter version):
            # five launches -> three  (GEMM+BN | stencil | grouped GEMM+BN+neuron)
            ev = self._eval_affines(t, bn1, bn2, bn3)
            z = ops.gemm_bn_lif_eval(sv, w1, None, bn1.running_mean, bn1.running_var, bn1.weight, bn1.bias, bn1.eps, want_pre=True,
                                     lif=False)[0].view(T * B, 3 * C, H, W)
            z = ops.dwconv(z, ops.cat_params(t["dw"]), 1, ev["border"])
            pair, n = ev["pair"], self.q_spike
            if ops.dense_gemm_bn_lif_eval_ok(z, N) and ops.spikes_bf16_ok(n.D):
                _, y = ops.dense_gemm_bn_lif_eval(z.view(T * B, 3 * C, N), [p.view(C, C) for p in t["w2"]], None, pair.running_mean,
                                                  pair.running_var, pair.weight, pair.bias, pair.eps, lif=True, D=n.D, vth=n.v_threshold)
                n.v = 0.0
                return y.view(T * B, 3 * C, N)
            z = ops.dense_gemm(z.view(T * B, 3 * C, N), [p.view(C, C) for p in t["w2"]], stats=False).view(T * B, 3 * C, H, W)
            _, y = bn_act(z, None, pair, lif=self.q_spike)
            return y.view(T * B, 3 * C, N)
        z = ops.spike_gemm(sv, w1, stats=training)                    # the three first 1x1 convs: one GEMM
        z = ops.carry_stats(z, z.view(T * B, 3 * C, H, W))
        z, _, border = bn_act(z, None, bn1, want_border=True)
        z = ops.dwconv(z, ops.cat_params(t["dw"]), 1, border)
        # second 1x1: three products on the channel groups of z, each with its own parameter (cached pack, gradient sink)
        z = ops.dense_gemm(z.view(T * B, 3 * C, N), [p.view(C, C) for p in t["w2"]], stats=training)
        z = ops.carry_stats(z, z.view(T * B, 3 * C, H, W))
        _, y = bn_bn_act(z, bn2, bn3, lif=self.q_spike)   # q / k / v neurons: pure and identical here (checked by the caller)
        for wb in out:
            wb()
        if training:
            torch._foreach_add_([b.num_batches_tracked for bns in t["bns"] for b in bns], 1)
        return y.view(T * B, 3 * C, N)          # q | k | v spikes, channel-stacked: the attention core reads the ranges in place

    def _eval_affines(self, t, bn1, bn2, bn3):
        """eval mode: BN_1(0) (the stencil's border value) and BN_3 o BN_2 as one affine pair on BN_2's running statistics, cached on
        the versions of the 27 parameter / buffer tensors involved"""
        key = tuple((x.data_ptr(), x._version) for bns in t["bns"] for b in bns for x in (b.weight, b.bias, b.running_mean, b.running_var))
        hit = getattr(self, "_s2f_eval_affines", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            border = (bn1.bias - bn1.running_mean * bn1.weight / torch.sqrt(bn1.running_var + bn1.eps)).contiguous()
            r3g3 = bn3.weight / torch.sqrt(bn3.running_var + bn3.eps)
            pair = fused._EvalBN()
            pair.weight = (bn2.weight * r3g3).contiguous()
            pair.bias = ((bn2.bias - bn3.running_mean) * r3g3 + bn3.bias).contiguous()
            pair.running_mean, pair.running_var, pair.eps = bn2.running_mean.clone(), bn2.running_var.clone(), bn2.eps
        out = dict(border=border, pair=pair)
        self._s2f_eval_affines = (key, out)
        return out

    def _can_batch(self):
        lifs = (self.q_spike, self.k_spike, self.v_spike)
        bns = [b for group in self._twins()["bns"] for b in group]
        return (QKV_BATCHED and all(isinstance(n.v, float) and not n.keep_membrane and n.stats is None and not n._forward_hooks
                                    for n in lifs)
                and len({(n.D, n.v_threshold) for n in lifs}) == 1
                and all(c[0].body[2][0].kernel_size == (3, 3) for c in (self.q_conv, self.k_conv, self.v_conv))
                and len({(b.training, b.momentum, b.eps, b.running_mean is None) for b in bns}) == 1
                and bns[0].momentum is not None and bns[0].running_mean is not None)

    def forward(self, x, residual=None, next_lif=None):
        T, B, C, H, W = x.shape
        N = H * W
        s = self.head_spike.fire(x).flatten(0, 1)
        # attention core + attn_spike: one fused kernel on the bf16 spikes when the neuron is stateless (ops.sdsa)
        if s.is_cuda and N % 4 == 0 and self._can_batch():
            o = ops.sdsa_packed(self._qkv_batched(s, T, B, C, H, W), self.num_heads, self.scale, lif=self.attn_spike)
        else:
            q, k, v = ops.branches([
                lambda: self.q_conv[0](s, outer_bn=self.q_conv[1], lif=self.q_spike)[1].view(T * B, C, N),
                lambda: self.k_conv[0](s, outer_bn=self.k_conv[1], lif=self.k_spike)[1].view(T * B, C, N),
                lambda: self.v_conv[0](s, outer_bn=self.v_conv[1], lif=self.v_spike)[1].view(T * B, C, N)], inputs=(s,))
            o = ops.sdsa(q, k, v, self.num_heads, self.scale, lif=self.attn_spike)           # [TB, C, N], c = head*d + j
        o = o.view(T * B, C, H, W)
        res = None if residual is None else residual.flatten(0, 1)
        return self.proj_conv[0](o, outer_bn=self.proj_conv[1], residual=res, next_lif=next_lif)[0].reshape(T, B, C, H, W)


class MS_Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, norm_layer=nn.LayerNorm, sr_ratio=1, T=None):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 is not used by any Spike2Former config")
        self.attn = MS_Attention_RepConv_qkv_id(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                                attn_drop=attn_drop, proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_path = nn.Identity()
        self.mlp = MS_MLP(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)

    @property
    def first_lif(self):
        return self.attn.head_spike

    def forward(self, x, next_lif=None):
        x = self.attn(x, residual=x, next_lif=self.mlp.fc1_spike)   # x + attn(x), residual fused into the last BatchNorm kernel
        return self.mlp(x, residual=x, next_lif=next_lif)           # x + mlp(x)


class MS_DownSampling(nn.Module):
    """[LIF ->] conv(k, s) -> BN   (sdtv2.py:386-421)."""

    def __init__(self, in_channels=2, embed_dims=256, kernel_size=3, stride=2, padding=1, first_layer=True, T=None):
        super().__init__()
        self.encode_conv = Conv2d(in_channels, embed_dims, kernel_size=kernel_size, stride=stride, padding=padding)
        self.encode_bn = nn.BatchNorm2d(embed_dims)
        self.first_layer = first_layer
        if not first_layer:
            self.encode_spike = _lif()
            spikes_in(self.encode_conv)

    @property
    def first_lif(self):
        return getattr(self, "encode_spike", None)

    def forward(self, x, next_lif=None):
        T, B = x.shape[:2]
        if hasattr(self, "encode_spike"):
            x = self.encode_spike.fire(x)
        x = x.flatten(0, 1)
        u = self._eval_fused(x, next_lif)
        if u is None:
            u, _ = bn_act(self.encode_conv.forward_nobias(x), self.encode_conv.bias, self.encode_bn, next_lif=next_lif)
        return u.reshape(T, B, *u.shape[1:])

    def _eval_fused(self, x, next_lif):
        """Inference (row f4): the convolution as its column matrix times the weight with the BatchNorm (running statistics) and the
        next block's first neuron in the GEMM's epilogue -- bf16 spike columns on the 3-pass kernel, the stem's fp32 image columns on the
        6-pass one.  None when the fusion does not apply (training, gradients, a stateful / observed neuron, odd shapes)."""
        conv, bn, n = self.encode_conv, self.encode_bn, next_lif
        if (bn.training or bn.running_mean is None or not bn.affine or torch.is_grad_enabled() or not fused.EVAL_FUSION
                or not x.is_cuda or conv.groups != 1 or conv.dilation != (1, 1) or conv.stride[0] != conv.stride[1]
                or conv.padding[0] != conv.padding[1] or conv.kernel_size[0] != conv.kernel_size[1]):
            return None
        if n is not None and not (ops.spikes_bf16_ok(n.D) and isinstance(n.v, float) and not n.keep_membrane and n.stats is None
                                  and not n._forward_hooks and not n._forward_pre_hooks):
            return None
        k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        N_, C_, H_, W_ = x.shape
        Ho, Wo = (H_ + 2 * pd - k) // st + 1, (W_ + 2 * pd - k) // st + 1
        L, M = Ho * Wo, conv.out_channels
        spikes = isinstance(x, ops.Spikes)
        if spikes and (x.tok is None or x.data.dtype != torch.bfloat16):
            return None
        if L % 4 or L < 128:
            return None
        kw = dict(want_pre=True, lif=n is not None, D=(n.D if n is not None else 8), vth=(n.v_threshold if n is not None else 1.0))
        bnargs = (conv.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps)
        w2d = conv.weight.view(M, -1)
        if spikes:
            cols = ops.im2col(x.data, k, k, st, pd)
            u, y, _ = ops.gemm_bn_lif_eval(ops.Spikes(cols, ops.core._new_tok(cols)), w2d, *bnargs, **kw)
        else:
            u, y = ops.dense_gemm_bn_lif_eval(ops.im2col(x, k, k, st, pd), w2d, *bnargs, **kw)
        u = u.view(N_, M, Ho, Wo)
        if n is not None:
            n.v = 0.0
            n.prefire(u, y.view(N_, M, Ho, Wo))
        return u


@MODELS.register_module()
class Spiking_vit_MetaFormer(nn.Module):
    """Registry type 'Spiking_vit_MetaFormer' (sdtv2.py:424-655).  forward(img [B,3,H,W]) -> 4 maps [T,B,C_i,H_i,W_i]
    at strides 2, 4, 8, 16 for decode_mode='Qsnn'.  Block counts 6 and 2 are fixed as in the reference (:536,565)."""

    def __init__(self, img_size_h=128, img_size_w=128, patch_size=16, in_channels=2, num_classes=11,
                 embed_dim=(64, 128, 256), num_heads=(1, 2, 4), mlp_ratios=(4, 4, 4), qkv_bias=False, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm, depths=(6, 8, 6),
                 sr_ratios=(8, 4, 2), T=1, decode_mode="snn", init_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, pretrained=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.num_classes = num_classes
        self.depths = depths
        self.T = T
        self.decode_mode = decode_mode
        self.freeze_bn_ = norm_eval          # accepted but unused, as in the reference (:448,457)
        self.norm_cfg = norm_cfg
        if drop_path_rate != 0.0:
            raise NotImplementedError("drop_path_rate != 0 is not used by any Spike2Former config")
        e = list(embed_dim)
        blk = dict(num_heads=num_heads, mlp_ratio=mlp_ratios, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                   attn_drop=attn_drop_rate, drop_path=0.0, norm_layer=norm_layer, sr_ratio=sr_ratios)
        self.downsample1_1 = MS_DownSampling(in_channels, e[0] // 2, kernel_size=7, stride=2, padding=3,
                                             first_layer=True)
        self.ConvBlock1_1 = nn.ModuleList([MS_ConvBlock(dim=e[0] // 2, mlp_ratio=mlp_ratios)])
        self.downsample1_2 = MS_DownSampling(e[0] // 2, e[0], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.ConvBlock1_2 = nn.ModuleList([MS_ConvBlock(dim=e[0], mlp_ratio=mlp_ratios)])
        self.downsample2 = MS_DownSampling(e[0], e[1], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.ConvBlock2_1 = nn.ModuleList([MS_ConvBlock(dim=e[1], mlp_ratio=mlp_ratios)])
        self.ConvBlock2_2 = nn.ModuleList([MS_ConvBlock(dim=e[1], mlp_ratio=mlp_ratios)])
        self.downsample3 = MS_DownSampling(e[1], e[2], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.block3 = nn.ModuleList([MS_Block(dim=e[2], **blk) for _ in range(6)])
        self.downsample4 = MS_DownSampling(e[2], e[3], kernel_size=3, stride=1, padding=1, first_layer=False)
        self.block4 = nn.ModuleList([MS_Block(dim=e[3], **blk) for _ in range(2)])

    def init_weights(self):
        """Load `init_cfg['checkpoint']`, stripping a leading 'backbone.' from the keys, strict=False (:577-612)."""
        if self.init_cfg is None:
            return
        assert "checkpoint" in self.init_cfg, f"Only support specify `Pretrained` in `init_cfg` in {type(self).__name__}"
        ckpt = torch.load(self.init_cfg["checkpoint"], map_location="cpu")
        sd = ckpt.get("state_dict", ckpt.get("model", ckpt))
        sd = OrderedDict((k[9:] if k.startswith("backbone.") else k, v) for k, v in sd.items())
        return self.load_state_dict(sd, strict=False)

    def forward_features(self, x):
        # (sdtv2.py:617 `x.unsqueeze(0).repeat(T, 1, 1, 1, 1)`: the same T copies, written by one broadcast copy kernel instead of
        # ATen's concatenation -- 25 MB at C2: 141 -> ~20 us)
        x = x.unsqueeze(0).expand(self.T, *x.shape).contiguous()
        # The stages as one chain: every module is told which neuron reads its output next, so that neuron's update runs
        # inside the kernel that produces the output (fused.bn_act `next_lif`).  x1..x4 are taps of the same stream.
        chain = [self.downsample1_1, *self.ConvBlock1_1, self.downsample1_2, *self.ConvBlock1_2, self.downsample2,
                 *self.ConvBlock2_1, *self.ConvBlock2_2, self.downsample3, *self.block3, self.downsample4, *self.block4]
        taps = {id(self.ConvBlock1_1[-1]): 0, id(self.ConvBlock1_2[-1]): 1, id(self.ConvBlock2_2[-1]): 2,
                id(self.block4[-1]): 3}
        outs = [None] * 4
        for i, m in enumerate(chain):
            x = m(x, next_lif=chain[i + 1].first_lif if i + 1 < len(chain) else None)
            if id(m) in taps:
                outs[taps[id(m)]] = x
        x1, x2, x3, x4 = outs
        if self.decode_mode == "snn":
            return [t.mean(0, keepdim=True) for t in (x1, x2, x3, x4)]
        if self.decode_mode == "Qsnn":
            return [x1, x2, x3, x4]
        return [t.flatten(0, 1) for t in (x1, x2, x3, x4)]

    def forward(self, x):
        return self.forward_features(x)
