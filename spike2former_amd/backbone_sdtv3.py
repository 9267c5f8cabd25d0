"""E-SpikeFormer (SDT-v3) backbone on the MI355X kernels -- SURVEY section 8 row f3, BASELINE config 5.
Mirrors mmseg/models/backbones/sdtv3.py:99-616 (class names, constructor kwargs, state_dict keys) and the stateless
four-level neuron `Multispike_norm` (mmseg/models/utils/Qtrick.py:4-38).

What differs from SDT-v2 (backbone_sdtv2.py):
  * the neuron keeps no membrane: y = round(clamp(x, 0, 4)) / 4 with the same straight-through gradient -- the neuron
    kernels in stateless mode with D = 4 (they already take D), fused into the producing BatchNorm kernel as before;
  * SepConv_Spike normalises after every convolution; the transformer blocks start with a 3x3 SepConv_Spike;
  * attention (`MS_Attention_linear`, :228-318): plain 1x1 conv + BN projections, values 4x wider than queries / keys
    (lamda_ratio = 4), out = (q k^T) v * 2 scale, no softmax.  Evaluated as q (k^T v) -- exact for spike operands (multiples
    of 1/4, sums far below 2^24), O(N d^2) instead of the O(N^2 d) the reference spends (N = 4 200 tokens at 800x1344).
    The 4d-wide value heads run as four d-wide problems of the existing attention kernels."""
import torch
import torch.nn as nn

from . import ops
from .backbone_sdtv2 import MS_DownSampling as _DownSamplingV2
from .backbone_sdtv2 import MS_MLP as _MLPV2
from .conv import Conv2d, spikes_in
from .fused import bn_act
from .neuron import Q_IFNode, Quant
from .registry import MODELS


class Multispike_norm(Q_IFNode):
    """Qtrick.py:26-38: stateless, D = T = 4 levels.  A Q_IFNode that never keeps a membrane, so every fused path
    (BatchNorm + neuron kernels, prefire) applies unchanged."""

    def __init__(self, T=4):
        super().__init__(surrogate_function=Quant(D=T))
        self.T = T
        object.__setattr__(self, "_keep", False)

    keep_membrane = property(lambda self: False, lambda self, v: None)      # nothing to carry: there is no state


class SepConv_Spike(nn.Module):
    """neuron -> pw 1x1 + BN -> neuron -> dw kxk + BN -> neuron -> pw 1x1 + BN   (sdtv3.py:99-150)."""

    def __init__(self, dim, expansion_ratio=2, act2_layer=nn.Identity, bias=False, kernel_size=7, padding=3, T=4):
        super().__init__()
        med = int(expansion_ratio * dim)
        self.T = T
        self.spike1 = Multispike_norm()
        self.pwconv1 = nn.Sequential(Conv2d(dim, med, kernel_size=1, stride=1, bias=bias), nn.BatchNorm2d(med))
        self.spike2 = Multispike_norm()
        self.dwconv = nn.Sequential(Conv2d(med, med, kernel_size=kernel_size, padding=padding, groups=med, bias=bias),
                                    nn.BatchNorm2d(med))
        self.spike3 = Multispike_norm()
        self.pwconv2 = nn.Sequential(Conv2d(med, dim, kernel_size=1, stride=1, bias=bias), nn.BatchNorm2d(dim))
        spikes_in(self.pwconv1[0], self.pwconv2[0])

    def forward(self, x, residual=None, next_lif=None):
        T, B, C, H, W = x.shape
        s = self.spike1.fire(x).flatten(0, 1)
        _, s = bn_act(self.pwconv1[0].forward_nobias(s), self.pwconv1[0].bias, self.pwconv1[1], lif=self.spike2)
        _, s = bn_act(self.dwconv[0].forward_nobias(s), self.dwconv[0].bias, self.dwconv[1], lif=self.spike3)
        u, _ = bn_act(self.pwconv2[0].forward_nobias(s), self.pwconv2[0].bias, self.pwconv2[1],
                      residual=None if residual is None else residual.flatten(0, 1), next_lif=next_lif)
        return u.reshape(T, B, C, H, W)


class MS_ConvBlock_spike_SepConv(nn.Module):
    """x += SepConv_Spike(x);  x += BN(conv3x3(neuron(BN(conv3x3(neuron(x))))))   (sdtv3.py:153-189)."""

    def __init__(self, dim, mlp_ratio=4.0, T=4):
        super().__init__()
        self.T = T
        self.Conv = SepConv_Spike(dim=dim)
        self.mlp_ratio = mlp_ratio
        self.spike1 = Multispike_norm()
        self.conv1 = Conv2d(dim, dim * mlp_ratio, kernel_size=3, padding=1, groups=1, bias=False)
        self.bn1 = nn.BatchNorm2d(dim * mlp_ratio)
        self.spike2 = Multispike_norm()
        self.conv2 = Conv2d(dim * mlp_ratio, dim, kernel_size=3, padding=1, groups=1, bias=False)
        self.bn2 = nn.BatchNorm2d(dim)
        spikes_in(self.conv1, self.conv2)

    @property
    def first_lif(self):
        return self.Conv.spike1

    def forward(self, x, next_lif=None):
        T, B, C, H, W = x.shape
        feat = self.Conv(x, residual=x, next_lif=self.spike1)
        s = self.spike1.fire(feat)
        _, s = bn_act(self.conv1(s.flatten(0, 1)), None, self.bn1, lif=self.spike2)
        u, _ = bn_act(self.conv2(s), None, self.bn2, residual=feat.flatten(0, 1), next_lif=next_lif)
        return u.reshape(T, B, C, H, W)


class MS_MLP(_MLPV2):
    """sdtv3.py:192-225 -- the SDT-v2 MLP with the stateless neuron."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0, layer=0):
        super().__init__(in_features, hidden_features, out_features, drop, layer)
        self.fc1_spike, self.fc2_spike = Multispike_norm(), Multispike_norm()


class MS_DownSampling(_DownSamplingV2):
    """sdtv3.py:363-399."""

    def __init__(self, in_channels=2, embed_dims=256, kernel_size=3, stride=2, padding=1, first_layer=True):
        super().__init__(in_channels, embed_dims, kernel_size, stride, padding, first_layer)
        if not first_layer:
            self.encode_spike = Multispike_norm()


class MS_Attention_linear(nn.Module):
    """sdtv3.py:228-318."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, sr_ratio=1,
                 lamda_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.lamda_ratio = lamda_ratio
        cv = int(dim * lamda_ratio)
        self.head_spike = Multispike_norm()
        self.q_conv = nn.Sequential(Conv2d(dim, dim, 1, 1, bias=False), nn.BatchNorm2d(dim))
        self.q_spike = Multispike_norm()
        self.k_conv = nn.Sequential(Conv2d(dim, dim, 1, 1, bias=False), nn.BatchNorm2d(dim))
        self.k_spike = Multispike_norm()
        self.v_conv = nn.Sequential(Conv2d(dim, cv, 1, 1, bias=False), nn.BatchNorm2d(cv))
        self.v_spike = Multispike_norm()
        self.attn_spike = Multispike_norm()
        self.proj_conv = nn.Sequential(Conv2d(cv, dim, 1, 1, bias=False), nn.BatchNorm2d(dim))
        spikes_in(self.q_conv[0], self.k_conv[0], self.v_conv[0], self.proj_conv[0])

    def forward(self, x, residual=None, next_lif=None):
        T, B, C, H, W = x.shape
        N, h = H * W, self.num_heads
        d = C // h
        r = int(self.lamda_ratio)
        assert r * C == int(C * self.lamda_ratio), "lamda_ratio must be an integer (4 in every config)"
        s = self.head_spike.fire(x).flatten(0, 1)
        q = bn_act(self.q_conv[0](s), None, self.q_conv[1], lif=self.q_spike)[1].view(T * B, C, N)
        k = bn_act(self.k_conv[0](s), None, self.k_conv[1], lif=self.k_spike)[1].view(T * B, C, N)
        v = bn_act(self.v_conv[0](s), None, self.v_conv[1], lif=self.v_spike)[1]                 # [TB, r*C, H, W]
        # value channel c_v = head * (r d) + j * d + jj: the j-th d-wide slice of every head is one ordinary attention problem
        if isinstance(v, ops.Spikes) and v.tok is not None:
            # bf16 spikes stay bf16 (the regrouping copy moves half the bytes, the attention core runs its bf16 kernels)
            vj = v.view(T * B, h, r, d, N).permute(2, 0, 1, 3, 4).contiguous()                  # [r, TB, h, d, N]
            vs = [ops.Spikes(vj.data[j].view(T * B, C, N), vj.tok[j].reshape(T * B, C, N)) for j in range(r)]
        else:
            vj = ops.spikes_float(v).view(T * B, h, r, d, N).permute(2, 0, 1, 3, 4).contiguous()
            vs = [vj[j].reshape(T * B, C, N) for j in range(r)]
        o = torch.stack([ops.sdsa(q, k, vs[j], h, self.scale * 2) for j in range(r)], 0)
        o = o.view(r, T * B, h, d, N).permute(1, 2, 0, 3, 4).reshape(T * B, r * C, H, W)        # back to c_v order
        o = self.attn_spike.fire(o)
        res = None if residual is None else residual.flatten(0, 1)
        u, _ = bn_act(self.proj_conv[0](o), None, self.proj_conv[1], residual=res, next_lif=next_lif)
        return u.reshape(T, B, C, H, W)


class MS_Block_Spike_SepConv(nn.Module):
    """x += SepConv_Spike3x3(x); x += attn(x); x += mlp(x)   (sdtv3.py:321-360)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, norm_layer=nn.LayerNorm, sr_ratio=1, init_values=1e-6):
        super().__init__()
        if drop_path > 0.0:
            raise NotImplementedError("drop_path > 0 is not used by any Spike2Former config")
        self.conv = SepConv_Spike(dim=dim, kernel_size=3, padding=1)
        self.attn = MS_Attention_linear(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                        proj_drop=drop, sr_ratio=sr_ratio, lamda_ratio=4)
        self.drop_path = nn.Identity()
        self.mlp = MS_MLP(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)

    @property
    def first_lif(self):
        return self.conv.spike1

    def forward(self, x, next_lif=None):
        x = self.conv(x, residual=x, next_lif=self.attn.head_spike)
        x = self.attn(x, residual=x, next_lif=self.mlp.fc1_spike)
        return self.mlp(x, residual=x, next_lif=next_lif)


@MODELS.register_module()
class Spiking_vit_MetaFormerv2(nn.Module):
    """Registry type 'Spiking_vit_MetaFormerv2' (sdtv3.py:401-616).  forward(img [B,3,H,W]) -> 4 maps [T,B,C_i,H_i,W_i] at
    strides 2, 4, 8, 16 for decode_mode='QTrick'.  Block counts 6 and 2 are fixed as in the reference (:500,526)."""

    def __init__(self, img_size_h=128, img_size_w=128, patch_size=16, in_channels=2, num_classes=11,
                 embed_dim=(64, 128, 256), num_heads=(1, 2, 4), mlp_ratios=(4, 4, 4), qkv_bias=False, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm, depths=(6, 8, 6),
                 sr_ratios=(8, 4, 2), T=1, decode_mode="QTrick", init_cfg=None, norm_cfg=dict(type="BN", requires_grad=True),
                 pretrained=None, norm_eval=False):
        super().__init__()
        if drop_path_rate != 0.0:
            raise NotImplementedError("drop_path_rate > 0 is not used by any Spike2Former config")
        self.num_classes, self.depths, self.T, self.decode_mode, self.init_cfg = num_classes, depths, T, decode_mode, init_cfg
        e = list(embed_dim)
        blk = dict(num_heads=num_heads, mlp_ratio=mlp_ratios, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                   attn_drop=attn_drop_rate, drop_path=0.0, norm_layer=norm_layer, sr_ratio=sr_ratios)
        self.downsample1_1 = MS_DownSampling(in_channels, e[0] // 2, kernel_size=7, stride=2, padding=3, first_layer=True)
        self.ConvBlock1_1 = nn.ModuleList([MS_ConvBlock_spike_SepConv(dim=e[0] // 2, mlp_ratio=mlp_ratios)])
        self.downsample1_2 = MS_DownSampling(e[0] // 2, e[0], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.ConvBlock1_2 = nn.ModuleList([MS_ConvBlock_spike_SepConv(dim=e[0], mlp_ratio=mlp_ratios)])
        self.downsample2 = MS_DownSampling(e[0], e[1], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.ConvBlock2_1 = nn.ModuleList([MS_ConvBlock_spike_SepConv(dim=e[1], mlp_ratio=mlp_ratios)])
        self.ConvBlock2_2 = nn.ModuleList([MS_ConvBlock_spike_SepConv(dim=e[1], mlp_ratio=mlp_ratios)])
        self.downsample3 = MS_DownSampling(e[1], e[2], kernel_size=3, stride=2, padding=1, first_layer=False)
        self.block3 = nn.ModuleList([MS_Block_Spike_SepConv(dim=e[2], **blk) for _ in range(6)])
        self.downsample4 = MS_DownSampling(e[2], e[3], kernel_size=3, stride=1, padding=1, first_layer=False)
        self.block4 = nn.ModuleList([MS_Block_Spike_SepConv(dim=e[3], **blk) for _ in range(2)])

    def init_weights(self):
        if self.init_cfg is None:
            return
        assert "checkpoint" in self.init_cfg, f"Only support specify `Pretrained` in `init_cfg` in {self.__class__.__name__} "
        ckpt = torch.load(self.init_cfg["checkpoint"], map_location="cpu")
        sd = ckpt.get("state_dict", ckpt.get("model", ckpt))
        self.load_state_dict({(k[9:] if k.startswith("backbone.") else k): v for k, v in sd.items()}, strict=False)

    def forward_features(self, x):
        x = x.unsqueeze(0).repeat(self.T, 1, 1, 1, 1)
        chain = [self.downsample1_1, *self.ConvBlock1_1, self.downsample1_2, *self.ConvBlock1_2, self.downsample2,
                 *self.ConvBlock2_1, *self.ConvBlock2_2, self.downsample3, *self.block3, self.downsample4, *self.block4]
        taps = {id(self.ConvBlock1_1[-1]): 0, id(self.ConvBlock1_2[-1]): 1, id(self.ConvBlock2_2[-1]): 2, id(self.block4[-1]): 3}
        outs = [None] * 4
        for i, m in enumerate(chain):
            x = m(x, next_lif=chain[i + 1].first_lif if i + 1 < len(chain) else None)
            if id(m) in taps:
                outs[taps[id(m)]] = x
        return outs if self.decode_mode == "QTrick" else x

    def forward(self, x):
        return self.forward_features(x)
