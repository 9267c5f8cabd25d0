"""Spiking layers of the MaskFormer head on the MI355X kernels: SepConv_Spike / MLP (mmcv_spike/SNN_core.py:11-123),
DCNv3 module (ops_dcnv3/modules/dcnv3.py:96-233), pixel-decoder MS_MLP and decoder attention / FFN
(mmcv_spike/transformer.py:196-361, 505-638, 710-831), encoder / decoder layers (detr_layers.py:19-60, 112-185,
263-339, 417-559) and the sine positional encoding (positional_encoding.py:59-98).  Class names, kwargs and state_dict
keys follow the reference; the bug-compatible `reshape`-instead-of-permute quirks are reproduced and flagged."""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .conv import Conv1d, Conv2d, spikes_in
from .fused import bn_act, conv_bn_act
from .neuron import Q_IFNode, Quant
from .registry import ConfigDict


def _lif():
    return Q_IFNode(surrogate_function=Quant())


FUSED_QUERY_NEURONS = os.environ.get("S2F_FUSED_QN", "1") != "0"          # decoder: `query + query_pos` and the neurons that read it as one launch (False: add + neurons, for A/B)


class SepConv_Spike(nn.Module):
    """NHWC in/out: LIF -> pw(C->rC)+BN -> LIF -> dw kxk+BN -> LIF -> pw(rC->C)+BN   (SNN_core.py:11-63)."""

    def __init__(self, dim, expansion_ratio=2, T=4, act2_layer=nn.Identity, bias=False, kernel_size=7, padding=3):
        super().__init__()
        med = int(expansion_ratio * dim)
        self.T = T
        self.expansion_ratio = expansion_ratio
        self.spike1 = _lif()
        self.pwconv1 = nn.Sequential(Conv2d(dim, med, kernel_size=1, stride=1, bias=bias), nn.BatchNorm2d(med))
        self.spike2 = _lif()
        self.dwconv = nn.Sequential(
            Conv2d(med, med, kernel_size=kernel_size, padding=padding, groups=med, bias=bias), nn.BatchNorm2d(med))
        self.spike3 = _lif()
        self.pwconv2 = nn.Sequential(Conv2d(med, dim, kernel_size=1, stride=1, bias=bias), nn.BatchNorm2d(dim))
        spikes_in(self.pwconv1[0], self.pwconv2[0])       # both follow a neuron (spike1, spike3)

    def forward_nchw(self, x, scale=None, residual=None, next_lif=None):
        """Channel-major form: x, residual, result [T,B,C,H,W].  -> scale * SepConv(x) [+ residual]; the layer scale, the
        residual add and the next neuron on the stream are folded into the last BatchNorm kernel (fused.bn_act)."""
        T, B, C, H, W = x.shape
        if residual is x:          # x + scale * SepConv(x): the residual's gradient is summed inside spike1's backward kernel
            x, residual = self.spike1.fire(x, skip=True)
            x = x.flatten(0, 1)
        else:
            x = self.spike1.fire(x).flatten(0, 1)
        _, x = conv_bn_act(self.pwconv1[0], x, self.pwconv1[1], lif=self.spike2)
        _, x = conv_bn_act(self.dwconv[0], x, self.dwconv[1], lif=self.spike3)          # (eval: stencil + BatchNorm + neuron, one launch)
        x, _ = conv_bn_act(self.pwconv2[0], x, self.pwconv2[1], scale=scale, next_lif=next_lif,          # (eval: one launch)
                           residual=None if residual is None else residual.flatten(0, 1))
        return x.reshape(T, B, C, H, W)

    def forward(self, x):
        """The reference's interface: NHWC in, NHWC out (SNN_core.py:46-63)."""
        return self.forward_nchw(x.permute(0, 1, 4, 2, 3).contiguous()).permute(0, 1, 3, 4, 2).contiguous()


class MLP(nn.Module):
    """Mask-embedding MLP: fc1 -> 4*LIF -> fc2 -> 4*LIF -> fc_out   (SNN_core.py:95-123)."""

    def __init__(self, in_dim, out_dim, layer, quant_const=4, T=4):
        super().__init__()
        self.T = T
        self.fc1 = nn.Linear(in_dim, in_dim, bias=False)
        self.spike1 = _lif()
        self.fc2 = nn.Linear(in_dim, in_dim, bias=False)
        self.spike2 = _lif()
        self.fc_out = nn.Linear(in_dim, out_dim)
        self.quant_const = quant_const
        nn.init.constant_(self.fc_out.bias, 0)
        nn.init.trunc_normal_(self.fc_out.weight, std=0.02)

    def forward(self, x):
        x = self.spike1(ops.linear_tm(x, self.fc1.weight)) * self.quant_const
        x = self.spike2(ops.linear_tm(x, self.fc2.weight)) * self.quant_const
        return ops.linear_tm(x, self.fc_out.weight, self.fc_out.bias)


class DCNv3_pytorch(nn.Module):
    """DCNv3 module of the pixel decoder (dcnv3.py:96-233); registry-visible name kept.  The sampling core is the HIP
    kernel `s2f_dcnv3_fwd/bwd`, not grid_sample."""

    def __init__(self, channels=64, kernel_size=3, dw_kernel_size=None, stride=1, pad=1, dilation=1, group=4,
                 offset_scale=1.0, expension_ratio=4, T=4, act_layer="GELU", norm_layer="LN",
                 center_feature_scale=False):
        super().__init__()
        if channels % group != 0:
            raise ValueError(f"channels must be divisible by group, but got {channels} and {group}")
        if center_feature_scale:
            raise NotImplementedError("center_feature_scale is False in every Spike2Former config")
        dw_kernel_size = dw_kernel_size if dw_kernel_size is not None else kernel_size
        self.offset_scale = offset_scale
        self.channels = channels
        self.kernel_size = kernel_size
        self.dw_kernel_size = dw_kernel_size
        self.stride = stride
        self.dilation = dilation
        self.pad = pad
        self.group = group
        self.group_channels = channels // group
        self.center_feature_scale = center_feature_scale
        self.T = T
        self.dw_spike = _lif()
        self.offset_spike = _lif()
        self.mask_spike = _lif()
        kk = kernel_size * kernel_size
        self.dw_conv = nn.Sequential(
            Conv2d(channels, channels, kernel_size=dw_kernel_size, padding=(dw_kernel_size - 1) // 2,
                      groups=channels, bias=False), nn.BatchNorm2d(channels))
        self.offset = nn.Sequential(Conv2d(channels, group * kk * 2, kernel_size=1, stride=1),
                                    nn.BatchNorm2d(group * kk * 2))
        self.mask = nn.Sequential(Conv2d(channels, group * kk, kernel_size=1, stride=1), nn.BatchNorm2d(group * kk))
        self.input_proj = SepConv_Spike(dim=channels, kernel_size=dw_kernel_size, padding=(dw_kernel_size - 1) // 2,
                                        expansion_ratio=expension_ratio)
        self.output_proj = SepConv_Spike(dim=channels, kernel_size=dw_kernel_size, padding=(dw_kernel_size - 1) // 2,
                                         expansion_ratio=expension_ratio)
        spikes_in(self.offset[0], self.mask[0])            # both read offset_spike's output
        for m in (self.offset[0], self.mask[0]):           # zero init as in the reference (:192-196)
            nn.init.constant_(m.weight, 0.0)
            nn.init.constant_(m.bias, 0.0)

    def forward_nchw(self, inp, scale=None, residual=None, next_lif=None):
        """Channel-major form: inp, residual, result [T,N,C,H,W].  Only the sampling core works on NHWC (its gather reads
        the Cg channels of a group as one vector): one transposition in, one out, instead of the reference's permute pair
        around every sub-module."""
        T, N, C, H, W = inp.shape
        through = [residual]

        def sampled_input():
            x = self.input_proj.forward_nchw(inp)
            return ops.transpose_last2(x.reshape(T * N, C, H * W)).view(T * N, H, W, C)   # NCHW -> [T*N, H, W, C]

        def offset_and_mask():
            if residual is inp:          # inp + scale * DCN(inp): the residual's gradient joins dw_spike's backward kernel
                x1, through[0] = self.dw_spike.fire(inp, skip=True)
                x1 = x1.flatten(0, 1)
            else:
                x1 = self.dw_spike.fire(inp).flatten(0, 1)
            ops.use_here(x1)
            _, x1 = conv_bn_act(self.dw_conv[0], x1, self.dw_conv[1], lif=self.offset_spike)
            # bug-compatible: the NCHW conv outputs are *reinterpreted* as [T*N, H, W, C'] (dcnv3.py:213-214)
            offset, _ = conv_bn_act(self.offset[0], x1, self.offset[1])
            # (the second reader of offset_spike's map: its gradient travels on the spare handle, ops.Spikes.second)
            _, mask = conv_bn_act(self.mask[0], x1.second() if isinstance(x1, ops.Spikes) else x1, self.mask[1], lif=self.mask_spike)
            return offset.reshape(T * N, H, W, -1), mask.reshape(T * N, H, W, -1)

        # the two chains share only `inp`: launched side by side when ops.BRANCH_STREAMS is set.  (Forking offset / mask once
        # more costs more than it hides: two 2-kernel chains, 56.9 vs 54.4 ms/step -- a fork / join pair in the replayed
        # hipGraph is worth several short kernels.)
        x, (offset, mask) = ops.branches([sampled_input, offset_and_mask], inputs=(inp,))
        k = self.kernel_size
        y = ops.dcnv3_core(x, offset, mask, k, k, self.stride, self.stride, self.pad, self.pad,
                           self.dilation, self.dilation, self.group, self.group_channels, self.offset_scale)
        y = ops.transpose_last2(y.reshape(T * N, H * W, C)).view(T, N, C, H, W)
        return self.output_proj.forward_nchw(y, scale=scale, residual=through[0], next_lif=next_lif)

    def forward(self, inp):
        """The reference's interface: NHWC in, NHWC out (dcnv3.py:198-233)."""
        return self.forward_nchw(inp.permute(0, 1, 4, 2, 3).contiguous()).permute(0, 1, 3, 4, 2).contiguous()


class MS_MLP(nn.Module):
    """Pixel-decoder FFN on NHWC input (mmcv_spike/transformer.py:787-831)."""

    def __init__(self, embed_dims=256, feedforward_channels=2048, num_fcs=2, act_cfg=None, ffn_drop=0.0, T=4,
                 dropout_layer=None, add_identity=True, init_cfg=None, layer_scale_init_value=0.0):
        super().__init__()
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        self.T = T
        self.fc1_spike = _lif()
        self.fc1_conv = Conv1d(embed_dims, feedforward_channels, kernel_size=1, stride=1)
        self.fc1_bn = nn.BatchNorm1d(feedforward_channels)
        self.fc2_spike = _lif()
        self.fc2_conv = Conv1d(feedforward_channels, embed_dims, kernel_size=1, stride=1)
        self.fc2_bn = nn.BatchNorm1d(embed_dims)
        spikes_in(self.fc1_conv, self.fc2_conv)

    def forward_nchw(self, x):
        """x [T,B,C,H,W] -> the FFN output as it lies in memory, [T*B, C, H*W] (the caller applies the reference's
        reinterpretation of that buffer as [T,B,H,W,C], :829)."""
        x = self.fc1_spike.fire(x.flatten(3)).flatten(0, 1)
        _, x = conv_bn_act(self.fc1_conv, x, self.fc1_bn, lif=self.fc2_spike)
        x, _ = conv_bn_act(self.fc2_conv, x, self.fc2_bn)
        return x

    def forward(self, x):
        T, B, H, W, C = x.shape
        # bug-compatible: [T*B, C, N] reinterpreted as [T, B, H, W, C] (:829)
        return self.forward_nchw(x.permute(0, 1, 4, 2, 3).contiguous()).reshape(T, B, H, W, C)


class DCNDetrTransformerEncoderLayer(nn.Module):
    """q += g1*SepConv(q); q += g2*DCN(q); q += g3*MLP(q) on NHWC [T,B,H,W,C]   (detr_layers.py:263-339)."""

    def __init__(self, self_attn_cfg=None, ffn_cfg=None, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.self_attn_cfg = ConfigDict(self_attn_cfg or dict(embed_dims=256, num_heads=8, dropout=0.0))
        if "batch_first" not in self.self_attn_cfg:
            self.self_attn_cfg["batch_first"] = True
        else:
            assert self.self_attn_cfg["batch_first"] is True
        self.ffn_cfg = ConfigDict(ffn_cfg or dict(embed_dims=256, feedforward_channels=1024, num_fcs=2))
        self.layer_scale = 1e-6
        self.embed_dims = self.self_attn_cfg.embed_dims
        self.Conv = SepConv_Spike(dim=self.embed_dims, kernel_size=3, padding=1, expansion_ratio=2)
        self.dcn = DCNv3_pytorch(channels=self.embed_dims, kernel_size=3, stride=1, pad=1, dilation=1,
                                 group=self.self_attn_cfg.group, offset_scale=1.0, expension_ratio=2, act_layer="GELU",
                                 norm_layer="BN", dw_kernel_size=self.self_attn_cfg.dw_kernel_size,
                                 center_feature_scale=False)
        self.ffn = MS_MLP(**self.ffn_cfg)
        self.gamma1 = nn.Parameter(self.layer_scale * torch.ones(self.embed_dims))
        self.gamma2 = nn.Parameter(self.layer_scale * torch.ones(self.embed_dims))
        self.gamma3 = nn.Parameter(self.layer_scale * torch.ones(self.embed_dims))

    def forward_nchw(self, q, next_lif=None):
        """The layer on the channel-major stream q [T,B,C,H,W] (value-identical to `forward` on q.permute(0,1,3,4,2)): the
        reference permutes to NCHW and back around each of the six sub-modules; here the stream stays NCHW, every
        `q + gamma * f(q)` is folded into f's last BatchNorm kernel, and only the DCN core and the FFN's memory
        reinterpretation transpose anything."""
        T, B, C, H, W = q.shape
        q = self.Conv.forward_nchw(q, scale=self.gamma1, residual=q, next_lif=self.dcn.input_proj.spike1)
        q = self.dcn.forward_nchw(q, scale=self.gamma2, residual=q, next_lif=self.ffn.fc1_spike)
        m = self.ffn.forward_nchw(q)                                     # [T*B, C, H*W] in memory == [T,B,H,W,C] semantically
        # q + gamma3 * m^T: transposition, layer scale and residual add in one pass (forward and backward)
        return ops.transpose_scale_add(m.view(T * B, H * W, C), q.reshape(T * B, C, H * W), self.gamma3).view(T, B, C, H, W)

    def forward(self, query):
        """The reference's interface: NHWC in, NHWC out (detr_layers.py:331-337)."""
        return self.forward_nchw(query.permute(0, 1, 4, 2, 3).contiguous()).permute(0, 1, 3, 4, 2).contiguous()


class DCNDetrTransformerEncoder(nn.Module):
    def __init__(self, num_layers, layer_cfg, init_cfg=None):
        super().__init__()
        self.num_layers = num_layers
        self.layer_cfg = ConfigDict(layer_cfg)
        self.layers = nn.ModuleList([DCNDetrTransformerEncoderLayer(**self.layer_cfg) for _ in range(num_layers)])
        self.embed_dims = self.layers[0].embed_dims

    def forward_nchw(self, q):
        for layer in self.layers:
            q = layer.forward_nchw(q)
        return q

    def forward(self, query):
        """NHWC in, NHWC out (the reference's interface); the layers run on the channel-major stream."""
        return self.forward_nchw(query.permute(0, 1, 4, 2, 3).contiguous()).permute(0, 1, 3, 4, 2).contiguous()


class MultiHeadAttentionBlock(nn.Module):
    """Decoder attention block (identical code for SA and CA in the reference, transformer.py:196-361):
    q,k,v = LIF(BN1d(Conv1d(LIF(.)))); out = BN1d(Conv1d(LIF( (q k^T / sqrt(C)) v ))).  No softmax, so the core is
    evaluated as q (k^T v) / sqrt(C) by ops.sdsa on channel-major spikes -- exact for spike operands."""

    def __init__(self, embed_dims, num_heads=8, attn_drop=0.0, dropout=0.0, proj_drop=0.0, batch_first=True,
                 dropout_layer=None):
        super().__init__()
        self.num_heads = num_heads
        self.embed_dim = embed_dims
        self.scale = (embed_dims // num_heads) ** -0.5

        def proj():
            return nn.Sequential(Conv1d(embed_dims, embed_dims, kernel_size=1, stride=1), nn.BatchNorm1d(embed_dims))

        self.q_conv_spike = _lif(); self.q_conv = proj()
        self.k_conv_spike = _lif(); self.k_conv = proj()
        self.v_conv_spike = _lif(); self.v_conv = proj()
        self.q_spike = _lif()
        self.k_spike = _lif()
        self.v_spike = _lif()
        self.attn_spike = _lif()
        self.out_conv = proj()
        spikes_in(self.q_conv[0], self.k_conv[0], self.v_conv[0], self.out_conv[0])

    @staticmethod
    def _proj(spike_in, conv, spike_out, x, channel_major=False, fired=None):
        """neuron -> Conv1d -> BN1d -> neuron on [t,b,L,dim] (or channel-major [t,b,dim,L]) -> channel-major spikes [t*b, dim, L]"""
        x = spike_in.fire(x) if fired is None else fired
        x = x.flatten(0, 1) if channel_major else x.permute(0, 1, 3, 2).flatten(0, 1)
        return conv_bn_act(conv[0], x, conv[1], lif=spike_out)[1]

    def project_kv(self, key, value, kv_channel_major=False, kv_spikes=None):
        """The key and value chains alone (they do not depend on the query) -> (k, v) channel-major spikes."""
        cm = kv_channel_major or kv_spikes is not None
        k = self._proj(self.k_conv_spike, self.k_conv, self.k_spike, key, cm, None if kv_spikes is None else kv_spikes[0])
        v = self._proj(self.v_conv_spike, self.v_conv, self.v_spike, value, cm, None if kv_spikes is None else kv_spikes[1])
        return k, v

    def fused_neurons_ok(self, which, x):
        """The input neurons `which` (of q_conv_spike / k_conv_spike / v_conv_spike) are pure functions of their input with the same
        parameters, and x [t, b, dim, n] is a map ops.sum2_lif takes: the position add and the neurons can run as one launch."""
        ns = [getattr(self, w + "_conv_spike") for w in which]
        return (FUSED_QUERY_NEURONS and x.is_cuda and x.dim() == 4 and x.shape[-1] % 4 == 0 and x.dtype == torch.float32
                and all(isinstance(n.v, float) and not n.keep_membrane and n.stats is None and not n._forward_hooks
                        and not n._forward_pre_hooks for n in ns)
                and len({(n.D, n.v_threshold) for n in ns}) == 1 and ops.spikes_bf16_ok(ns[0].D))

    def fire_with_pos(self, x, pos, skip=False):
        """x [t, b, dim, n], pos [b, dim, n] -> (Q_IFNode(x + pos), Q_IFNode(x)) as ops.Spikes of x's shape: ONE launch forward, one
        backward, the sum never materialised (ops.sum2_lif with a zero level embedding; fused_neurons_ok says when this is legal).
        `skip`: -> (.., .., x') with x' = x for the layer's `query + attention(query)`: that branch's gradient is summed in the
        neurons' backward kernel (ops.sum2_lif)."""
        z = getattr(self, "_zero_e", None)
        if z is None or z.device != x.device or z.shape[0] != x.shape[2]:
            z = self._zero_e = torch.zeros(x.shape[2], dtype=torch.float32, device=x.device)
        n0 = self.q_conv_spike
        out = ops.sum2_lif(x.flatten(0, 1), z, pos, x.shape[1], n0.D, n0.v_threshold, skip=skip)
        for m in (self.q_conv_spike, self.k_conv_spike, self.v_conv_spike):
            m.v = 0.0
        return tuple(o.view(x.shape) for o in out)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None, kv_channel_major=False, kv_spikes=None,
                kv_projected=None, query_channel_major=False, residual_cm=None, q_spikes=None):
        """`residual_cm` (channel-major streams only): a [t,b,dim,nq] tensor added to the result inside the output BatchNorm kernel
        (the decoder layer's `query + attention(query)`, detr_layers.py:523-537) -- the add launch disappears.
        query [t,b,nq,dim]; key/value [t,b,nk,dim] as in the reference, or -- `kv_channel_major` -- [t,b,dim,nk], the
        layout the pixel decoder produces them in (saves two 33 M-element transposes per projection at the 128x128 level;
        the neuron is elementwise, so the values are the same).  `kv_spikes` = (k_conv_spike(key), v_conv_spike(value))
        already formed by the caller (channel-major; the head's fused add + neuron kernel) -- key / value are then unused.
        `query_channel_major`: query is [t,b,dim,nq] and so is the result (the decoder's channel-major query stream: the
        projections and the attention core work on channel-major maps anyway, so nothing is transposed)."""
        if query_channel_major:
            t, b, dim, nq = query.shape
        else:
            t, b, nq, dim = query.shape
        qcm = query_channel_major

        if kv_projected is not None:
            # keys / values do not depend on the query: the head projected them for every layer ahead of the query chain
            q = self._proj(self.q_conv_spike, self.q_conv, self.q_spike, query, qcm, q_spikes)
            k, v, handle = kv_projected
            ops.join(handle, (k, v))
        else:
            cm = kv_channel_major or kv_spikes is not None
            fk, fv = (None, None) if kv_spikes is None else kv_spikes
            k, v, q = ops.branches([   # independent chains (the long ones first: keys / values are the 1 024 - 16 384-token maps)
                lambda: self._proj(self.k_conv_spike, self.k_conv, self.k_spike, key, cm, fk),
                lambda: self._proj(self.v_conv_spike, self.v_conv, self.v_spike, value, cm, fv),
                lambda: self._proj(self.q_conv_spike, self.q_conv, self.q_spike, query, qcm, q_spikes)],
                inputs=(query, key, value, fk, fv))
        if attn_mask is not None:
            # scores.masked_fill(mask, 0) (transformer.py:266-269, 349-352): the explicit O(nq nk d) form (ops.sdsa_masked); never
            # taken by the MaskFormerHead path, whose cross_attn_mask is None (dense_heads/maskformer_head.py:554-564)
            o = self.attn_spike.fire(ops.sdsa_masked(q, k, v, attn_mask, self.num_heads, 1.0 / (self.embed_dim ** 0.5), b))
        else:
            o = ops.sdsa(q, k, v, self.num_heads, 1.0 / (self.embed_dim ** 0.5), lif=self.attn_spike)  # embed_dim**0.5, not head dim
        res = residual_cm.reshape(t * b, dim, nq) if (residual_cm is not None and qcm) else None
        o, _ = conv_bn_act(self.out_conv[0], o, self.out_conv[1], residual=res)
        if qcm:
            return o.view(t, b, dim, nq), None
        return o.permute(0, 2, 1).reshape(t, b, nq, dim), None


CrossMultiHeadAttentionBlock = MultiHeadAttentionBlock


class MultiheadAttention(nn.Module):
    """Wrapper adding the positional encodings (transformer.py:505-638); returns the block output WITHOUT identity."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, attn_type="SA", dropout_layer=None,
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        if attn_type in ("LinearCA", "LinearSA"):
            raise NotImplementedError(f"attn_type={attn_type} is not used by any Spike2Former config")
        self.attn = MultiHeadAttentionBlock(embed_dims, num_heads, attn_drop, **kwargs)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, kv_channel_major=False, kv_spikes=None, kv_projected=None, **kwargs):
        qcm = bool(kwargs.get("query_channel_major", False))
        if kv_spikes is not None or kv_projected is not None:
            q_spikes = None
            residual_cm = kwargs.get("residual_cm") if qcm else None
            if qcm and query_pos is not None and self.attn.fused_neurons_ok(("q",), query):
                # query + query_pos only feeds the query neuron: add and neuron as one launch (ops.sum2_lif), the sum never written
                if residual_cm is query:
                    q_spikes, _, residual_cm = self.attn.fire_with_pos(query, query_pos, skip=True)
                else:
                    q_spikes = self.attn.fire_with_pos(query, query_pos)[0]
            elif query_pos is not None:
                query = query + query_pos
            return self.attn(query=query, key=None, value=None,
                             attn_mask=attn_mask, key_padding_mask=key_padding_mask, kv_spikes=kv_spikes,
                             kv_projected=kv_projected, query_channel_major=qcm, q_spikes=q_spikes,
                             residual_cm=residual_cm)[0]
        if key is None:
            key = query
        if value is None:
            value = key
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        return self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask,
                         kv_channel_major=kv_channel_major, query_channel_major=qcm)[0]


class MSDA_FFN(nn.Module):
    """Decoder FFN (transformer.py:710-784); both `reshape`s are reinterpretations, not transposes (:777,:781)."""

    def __init__(self, embed_dims=256, feedforward_channels=2048, num_fcs=2, act_cfg=None, ffn_drop=0.0, T=4,
                 dropout_layer=None, add_identity=True, init_cfg=None, layer_scale_init_value=0.0):
        super().__init__()
        assert num_fcs >= 2
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        self.T = T
        self.fc1_spike = _lif()
        self.fc1 = Conv1d(embed_dims, feedforward_channels, kernel_size=1, stride=1)
        self.bn1 = nn.BatchNorm1d(feedforward_channels)
        self.fc2_spike = _lif()
        self.fc2 = Conv1d(feedforward_channels, embed_dims, kernel_size=1, stride=1)
        self.bn2 = nn.BatchNorm1d(embed_dims)
        spikes_in(self.fc1, self.fc2)

    def forward(self, x, identity=None):
        """`identity` [t,bs,N,C]: added to the result -- elementwise in memory order, so inside bn2's kernel on the reinterpreted
        buffer (detr_layers.py:556: query + ffn(query))."""
        t, bs, N, C = x.shape
        if identity is x:          # x + ffn(x): the identity's gradient is summed inside fc1_spike's backward kernel
            a, identity = self.fc1_spike.fire(x, skip=True)
            a = a.reshape(t * bs, C, N)
        else:
            a = self.fc1_spike.fire(x).reshape(t * bs, C, N)
        _, a = conv_bn_act(self.fc1, a, self.bn1, lif=self.fc2_spike)
        res = identity.reshape(t * bs, C, N) if identity is not None else None
        a, _ = conv_bn_act(self.fc2, a, self.bn2, residual=res)
        return a.reshape(t, bs, N, C)


class DetrTransformerDecoderLayer(nn.Module):
    """query += CA(query, memory); query += SA(query); query += FFN(query)   (detr_layers.py:417-559)."""

    def __init__(self, self_attn_cfg=None, cross_attn_cfg=None, ffn_cfg=None, T=4, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.self_attn_cfg = ConfigDict(self_attn_cfg or dict(embed_dims=256, num_heads=8, batch_first=True))
        self.cross_attn_cfg = ConfigDict(cross_attn_cfg or dict(embed_dims=256, num_heads=8, batch_first=True))
        for c in (self.self_attn_cfg, self.cross_attn_cfg):
            if "batch_first" not in c:
                c["batch_first"] = True
            else:
                assert c["batch_first"] is True
        self.ffn_cfg = ConfigDict(ffn_cfg or dict(embed_dims=256, feedforward_channels=1024, num_fcs=2))
        self.self_attn = MultiheadAttention(**self.self_attn_cfg)
        self.cross_attn = MultiheadAttention(**self.cross_attn_cfg)
        self.embed_dims = self.self_attn.embed_dims
        self.ffn = MSDA_FFN(**self.ffn_cfg)

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, self_attn_mask=None,
                cross_attn_mask=None, key_padding_mask=None, kv_channel_major=False, kv_spikes=None, kv_projected=None,
                **kwargs):
        query = query + self.cross_attn(query=query, key=key, value=value, query_pos=query_pos, key_pos=key_pos,
                                        attn_mask=cross_attn_mask, key_padding_mask=key_padding_mask,
                                        kv_channel_major=kv_channel_major, kv_spikes=kv_spikes, kv_projected=kv_projected)
        query = query + self.self_attn(query=query, key=query, value=query, query_pos=query_pos, key_pos=query_pos,
                                       attn_mask=self_attn_mask)
        return query + self.ffn(query)

    def forward_stream(self, q_cm, query_pos_cm, key=None, value=None, kv_spikes=None, kv_projected=None, last=False):
        """The layer on the channel-major query stream q_cm [t,b,dim,nq] (query_pos_cm [b,dim,nq]; keys / values
        channel-major as with `kv_channel_major`) -> (query token-major [t,b,nq,dim], query channel-major or None when
        `last`).  Value-identical to `forward`: the reference keeps the queries token-major and transposes around each of
        the seven projections of a layer (12 copies per layer and step, forward + backward); the projections and the
        attention core are channel-major, so only the FFN's bug-compatible reinterpretation of the token-major buffer
        (transformer.py:777,:781) needs the other layout -- one transposition in, one out."""
        # the three residual adds of the layer run inside the last BatchNorm kernel of each branch (residual=)
        fused = kv_spikes is not None or kv_projected is not None
        # (query_pos_cm may come as a pair: one alias per reader -- cross-attention, self-attention -- of ops.fan_out, whose backward
        # sums the position embedding's twelve gradients of a step in one launch)
        pos_ca, query_pos_cm = query_pos_cm if isinstance(query_pos_cm, (tuple, list)) else (query_pos_cm, query_pos_cm)
        ca = self.cross_attn(query=q_cm, key=key, value=value, query_pos=pos_ca, kv_channel_major=True,
                             kv_spikes=kv_spikes, kv_projected=kv_projected, query_channel_major=True,
                             residual_cm=q_cm if fused else None)
        q_cm = ca if fused else q_cm + ca
        sa = self.self_attn.attn
        if sa.fused_neurons_ok(("q", "k", "v"), q_cm):
            # query + query_pos == key + key_pos feeds the query and key neurons, the query itself the value neuron: the add and
            # the three neurons as ONE launch (ops.sum2_lif: Q_IFNode(q + pos), Q_IFNode(q)); q and k share the first map
            yk, yv, through = sa.fire_with_pos(q_cm, query_pos_cm, skip=True)
            # (q and k read the same spikes: the second reader's gradient arrives on the neurons' spare handle, ops.Spikes.second)
            q_cm = sa(query=q_cm, key=None, value=None, kv_spikes=(yk, yv), q_spikes=yk.second(), query_channel_major=True,
                      residual_cm=through)[0]
        else:
            qp = q_cm + query_pos_cm                       # query + query_pos == key + key_pos: formed once
            q_cm = sa(query=qp, key=qp, value=q_cm, kv_channel_major=True, query_channel_major=True, residual_cm=q_cm)[0]
        q_tm = ops.transpose_last2(q_cm)
        out = self.ffn(q_tm, identity=q_tm)
        if last:
            return out, None
        # `out` goes to the prediction stack AND, transposed, to the next layer: the stack reads the pass-through, whose gradient the
        # transposition's adjoint sums (ops.transpose_last2 skip=True)
        q_next, out = ops.transpose_last2(out, skip=True)
        return out, q_next


class DetrTransformerDecoder(nn.Module):
    def __init__(self, num_layers, layer_cfg, post_norm_cfg=None, return_intermediate=True, init_cfg=None):
        super().__init__()
        self.layer_cfg = ConfigDict(layer_cfg)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.layers = nn.ModuleList([DetrTransformerDecoderLayer(**self.layer_cfg) for _ in range(num_layers)])
        self.embed_dims = self.layers[0].embed_dims

    def forward(self, query, key, value, query_pos, key_pos, key_padding_mask=None, **kwargs):
        inter = []
        for layer in self.layers:
            query = layer(query, key=key, value=value, query_pos=query_pos, key_pos=key_pos,
                          key_padding_mask=key_padding_mask, **kwargs)
            if self.return_intermediate:
                inter.append(query)
        return torch.stack(inter) if self.return_intermediate else query.unsqueeze(0)


class SinePositionalEncoding(nn.Module):
    """positional_encoding.py:59-98 (normalised sine/cosine embedding of the cumulative valid-pixel count)."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0.0,
                 init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize = num_feats, temperature, normalize
        self.scale, self.eps, self.offset = scale, eps, offset

    def forward(self, mask):
        not_mask = 1 - mask.to(torch.int)
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)
