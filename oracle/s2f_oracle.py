"""TEST INFRASTRUCTURE -- CPU oracle for the Spike2Former hot path.

This file is the *checker*, not the product: a plain-PyTorch (CPU, fp32) functional
restatement of the reference's algorithm, written from SURVEY.md Appendix C and from
reading the reference files cited on every function below.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package `spike2former_amd` never does and has no CPU fallback.

Parity status: PINNED.  `oracle/gen_golden.py` (run in the build container, where
`/root/reference` is mounted) imports the reference's own Python files through
`oracle/ref_shells.py`, runs them on seeded inputs, and (1) asserts this restatement
reproduces the reference to fp32 round-off, (2) writes the reference's outputs to
`tests/golden/*.npz`.  `tests/test_oracle_golden.py` re-checks this file against
those committed vectors everywhere (no reference needed).  The reference's own test
suite holds no vectors for this path.

Everything is a function of a flat `params` dict whose keys are the reference's
state_dict keys (`backbone.*`, `decode_head.*` -- SURVEY.md Appendix A), so the same
dict loads into the reference modules (`load_state_dict`), into this oracle, and into
the product modules.
"""
import math
import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- configs
@dataclass
class ModelCfg:
    """Shapes of one BASELINE.json config."""
    H: int = 512
    W: int = 512
    T: int = 4
    B: int = 2
    num_classes: int = 150
    embed_dim: Tuple[int, int, int, int] = (64, 128, 256, 360)
    num_heads: int = 8
    feat_channels: int = 256
    num_queries: int = 100
    pd_layers: int = 6
    pd_ffn: int = 1024
    dec_layers: int = 6
    dec_ffn: int = 2048
    group: int = 32
    dw_kernel_size: int = 5
    num_feats: int = 128
    D: int = 8            # quantisation levels of Q_IFNode (surrogate.py:525 `max_value=8`, neuron.py:197 `/ 8`)

    @property
    def in_channels(self):
        e = self.embed_dim
        return [e[0] // 2, e[0], e[1], e[3]]


CONFIGS = {
    # BASELINE.json configs[0]: plumbing config, shrunken widths (SURVEY.md 8d "C1")
    "C1": ModelCfg(H=128, W=128, T=1, B=1, num_classes=20, embed_dim=(16, 32, 64, 72), feat_channels=64,
                   num_queries=10, pd_layers=2, pd_ffn=256, dec_layers=2, dec_ffn=512, group=8, num_feats=32),
    # 64x64 form of C1 used for the committed end-to-end fixture (keeps the .npz small)
    "C1_64": ModelCfg(H=64, W=64, T=2, B=2, num_classes=20, embed_dim=(16, 32, 64, 72), feat_channels=64,
                      num_queries=10, pd_layers=2, pd_ffn=256, dec_layers=2, dec_ffn=512, group=8, num_feats=32),
    # configs[1]: ADE20K-150 512x512 T=4 B=2 (configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:23-93)
    "C2": ModelCfg(),
    # configs[2]: Cityscapes-19 1024x512, pixel-decoder ffn 2048, 2 images per GPU
    "C3": ModelCfg(H=512, W=1024, num_classes=19, pd_ffn=2048),
    # configs[3]: C2 with T=8
    "C4": ModelCfg(T=8),
}


# ----------------------------------------------------------------------------- name-seeded weights
def _gen(name: str) -> torch.Generator:
    return torch.Generator().manual_seed(zlib.crc32(name.encode()))


def seeded_tensor(name: str, shape, keys) -> torch.Tensor:
    """Deterministic, non-degenerate value for one state_dict entry.

    The kind is derived from the key: BN tensors are recognised by a sibling
    `running_mean`.  Defeats the reference's degenerate default init (zero DCN offset/mask
    convs, `gamma1..3 = 1e-6` -- dcnv3.py:192-196, detr_layers.py:301).
    """
    g = _gen(name)
    shape = tuple(shape)
    prefix, _, leaf = name.rpartition(".")
    is_bn = (prefix + ".running_mean") in keys
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if is_bn and leaf == "weight":
        return torch.rand(shape, generator=g) + 0.5
    if is_bn and leaf == "bias":
        return torch.randn(shape, generator=g) * 0.1
    if leaf in ("gamma1", "gamma2", "gamma3"):
        return torch.ones(shape)
    if name.endswith("decode_head.w") or name == "w":
        return torch.ones(shape)
    if "query_embed" in name or "query_feat" in name or "level_embed" in name:
        return torch.randn(shape, generator=g)
    if leaf == "bias":
        return torch.randn(shape, generator=g) * 0.1
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * fan_in ** -0.5
    return torch.randn(shape, generator=g) * 0.1


def seeded_state(shapes: Dict[str, tuple]) -> "OrderedDict[str, torch.Tensor]":
    keys = set(shapes)
    return OrderedDict((k, seeded_tensor(k, s, keys)) for k, s in shapes.items())


def param_shapes(cfg: ModelCfg) -> "OrderedDict[str, tuple]":
    """state_dict key -> shape for backbone + decode_head, built without the reference."""
    S = OrderedDict()

    def bn(p, c):
        S[p + ".weight"] = (c,); S[p + ".bias"] = (c,)
        S[p + ".running_mean"] = (c,); S[p + ".running_var"] = (c,); S[p + ".num_batches_tracked"] = ()

    def conv(p, co, ci, k, bias, one_d=False, kw=None):
        S[p + ".weight"] = (co, ci, k) if one_d else (co, ci, k, kw or k)
        if bias:
            S[p + ".bias"] = (co,)

    e = cfg.embed_dim
    b = "backbone."

    def down(n, ci, co, k):
        conv(b + n + ".encode_conv", co, ci, k, True); bn(b + n + ".encode_bn", co)

    def convblock(n, c):
        p = b + n + ".0."
        conv(p + "Conv.pwconv1", 2 * c, c, 1, False); bn(p + "Conv.bn1", 2 * c)
        conv(p + "Conv.dwconv", 2 * c, 1, 7, False)
        conv(p + "Conv.pwconv2", c, 2 * c, 1, False); bn(p + "Conv.bn2", c)
        conv(p + "conv1", 4 * c, c, 3, False); bn(p + "bn1", 4 * c)
        conv(p + "conv2", c, 4 * c, 3, False); bn(p + "bn2", c)

    def repconv(p, c):
        conv(p + ".0.body.0", c, c, 1, False); bn(p + ".0.body.1.bn", c)
        conv(p + ".0.body.2.0", c, 1, 3, False); conv(p + ".0.body.2.1", c, c, 1, False); bn(p + ".0.body.2.2", c)
        bn(p + ".1", c)

    def block(n, c):
        p = b + n + "."
        for q in ("q_conv", "k_conv", "v_conv", "proj_conv"):
            repconv(p + "attn." + q, c)
        conv(p + "mlp.fc1_conv", 4 * c, c, 1, True, one_d=True); bn(p + "mlp.fc1_bn", 4 * c)
        conv(p + "mlp.fc2_conv", c, 4 * c, 1, True, one_d=True); bn(p + "mlp.fc2_bn", c)

    down("downsample1_1", 3, e[0] // 2, 7); convblock("ConvBlock1_1", e[0] // 2)
    down("downsample1_2", e[0] // 2, e[0], 3); convblock("ConvBlock1_2", e[0])
    down("downsample2", e[0], e[1], 3); convblock("ConvBlock2_1", e[1]); convblock("ConvBlock2_2", e[1])
    down("downsample3", e[1], e[2], 3)
    for i in range(6):
        block(f"block3.{i}", e[2])
    down("downsample4", e[2], e[3], 3)
    for i in range(2):
        block(f"block4.{i}", e[3])

    h = "decode_head."
    Fc, Q, K = cfg.feat_channels, cfg.num_queries, cfg.num_classes
    S[h + "w"] = (1,)
    pd = h + "pixel_decoder."
    ic = cfg.in_channels
    for i in range(3):
        conv(pd + f"lateral_convs.{i}.0", Fc, ic[i], 1, True); bn(pd + f"lateral_convs.{i}.1", Fc)
    for i in range(3):
        conv(pd + f"output_convs.{i}.0", Fc, 1, 3, False); bn(pd + f"output_convs.{i}.1", Fc)
    conv(pd + "mask_feature", Fc, Fc, 1, True)

    def sepconv_spike(p, c, k):
        conv(p + ".pwconv1.0", 2 * c, c, 1, False); bn(p + ".pwconv1.1", 2 * c)
        conv(p + ".dwconv.0", 2 * c, 1, k, False); bn(p + ".dwconv.1", 2 * c)
        conv(p + ".pwconv2.0", c, 2 * c, 1, False); bn(p + ".pwconv2.1", c)

    G, KK = cfg.group, 9
    for i in range(cfg.pd_layers):
        p = pd + f"encoder.layers.{i}."
        for gname in ("gamma1", "gamma2", "gamma3"):
            S[p + gname] = (Fc,)
        sepconv_spike(p + "Conv", Fc, 3)
        conv(p + "dcn.dw_conv.0", Fc, 1, cfg.dw_kernel_size, False); bn(p + "dcn.dw_conv.1", Fc)
        conv(p + "dcn.offset.0", G * KK * 2, Fc, 1, True); bn(p + "dcn.offset.1", G * KK * 2)
        conv(p + "dcn.mask.0", G * KK, Fc, 1, True); bn(p + "dcn.mask.1", G * KK)
        sepconv_spike(p + "dcn.input_proj", Fc, cfg.dw_kernel_size)
        sepconv_spike(p + "dcn.output_proj", Fc, cfg.dw_kernel_size)
        conv(p + "ffn.fc1_conv", cfg.pd_ffn, Fc, 1, True, one_d=True); bn(p + "ffn.fc1_bn", cfg.pd_ffn)
        conv(p + "ffn.fc2_conv", Fc, cfg.pd_ffn, 1, True, one_d=True); bn(p + "ffn.fc2_bn", Fc)
    conv(pd + "encoder_in_proj.0", Fc, ic[3], 1, True); bn(pd + "encoder_in_proj.1", Fc)
    conv(pd + "encoder_out_proj.0", Fc, Fc, 1, True); bn(pd + "encoder_out_proj.1", Fc)
    for i in range(cfg.dec_layers):
        p = h + f"transformer_decoder.layers.{i}."
        for a in ("self_attn", "cross_attn"):
            for c_ in ("q_conv", "k_conv", "v_conv", "out_conv"):
                conv(p + f"{a}.attn.{c_}.0", Fc, Fc, 1, True, one_d=True); bn(p + f"{a}.attn.{c_}.1", Fc)
        conv(p + "ffn.fc1", cfg.dec_ffn, Fc, 1, True, one_d=True); bn(p + "ffn.bn1", cfg.dec_ffn)
        conv(p + "ffn.fc2", Fc, cfg.dec_ffn, 1, True, one_d=True); bn(p + "ffn.bn2", Fc)
    S[h + "query_embed.weight"] = (Q, Fc)
    S[h + "query_feat.weight"] = (Q, Fc)
    S[h + "level_embed.weight"] = (3, Fc)
    S[h + "cls_embed.weight"] = (K + 1, Fc); S[h + "cls_embed.bias"] = (K + 1,)
    S[h + "mask_embed.fc1.weight"] = (Fc, Fc)
    S[h + "mask_embed.fc2.weight"] = (Fc, Fc)
    S[h + "mask_embed.fc_out.weight"] = (Fc, Fc); S[h + "mask_embed.fc_out.bias"] = (Fc,)
    S[h + "shortcut_conv.0.weight"] = (Q, Q, 1)
    bn(h + "shortcut_conv.1", Q)
    return S


# ----------------------------------------------------------------------------- a1/a2: the neuron
class _QuantSTE(torch.autograd.Function):
    """`quant` (Qtrick_architecture/clock_driven/surrogate.py:522-538): fwd round(clamp(i, 0, D)) with
    torch.round = round-half-to-even; bwd passes the gradient where 0 <= i <= D (masks are i<0 and i>D)."""

    @staticmethod
    def forward(ctx, h, D):
        ctx.save_for_backward(h)
        ctx.D = D
        return torch.round(torch.clamp(h, min=0, max=D))

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        return g * ((h >= 0) & (h <= ctx.D)).to(g.dtype), None


def lif_step(x: torch.Tensor, v: Optional[torch.Tensor], D: int = 8, vth: float = 1.0):
    """One `Q_IFNode.forward` call (neuron.py:166-197, charge :459-460, soft reset :153).

    h = v + x ; s = rint(clamp(h, 0, D)) ; v' = h - s*vth ; y = s / D.   `v is None` == the python
    float 0. left by `reset_net` (base.py:25-69, functional.py:9-33).  Returns (y, v', s).
    """
    h = x if v is None else v + x
    s = _QuantSTE.apply(h, D)
    v_new = h - s * vth
    return s / D, v_new, s


def lif_seq_numpy(x_seq, v0=None, D=8):
    """numpy restatement of T successive stateful calls on the same neuron (what cal_firing_num.py does
    across images, tools/cal_firing_num.py:203-225).  x_seq [T, ...] float32 -> (counts u8 [T,...], v_T)."""
    import numpy as np
    x_seq = np.asarray(x_seq, dtype=np.float32)
    v = np.zeros_like(x_seq[0]) if v0 is None else np.asarray(v0, dtype=np.float32).copy()
    out = np.empty(x_seq.shape, dtype=np.uint8)
    for t in range(x_seq.shape[0]):
        h = (v + x_seq[t]).astype(np.float32)
        s = np.rint(np.clip(h, 0, D)).astype(np.float32)
        v = (h - s).astype(np.float32)
        out[t] = s.astype(np.uint8)
    return out, v


# ----------------------------------------------------------------------------- a9: DCNv3 core
def dcnv3_core(x, offset, mask, G, Cg, K=3, stride=1, pad=1, dil=1, offset_scale=1.0):
    """Direct pixel-coordinate restatement of `dcnv3_core_pytorch`
    (ops_dcnv3/functions/dcnv3_func.py:147-189; formula derived in SURVEY.md Appendix C.5).

    x [N,H,W,G*Cg], offset [N,Ho,Wo,G*K*K*2] (x then y per tap), mask [N,Ho,Wo,G*K*K] -> [N,Ho,Wo,G*Cg].
    Tap order k = i_w*K + j_h (kernel-w outer).  Bilinear with zero padding outside the *padded* input.
    Written as explicit gathers (not grid_sample) so that it is an independent check of the formula.
    """
    N, H, W, C = x.shape
    xp = F.pad(x, [0, 0, pad, pad, pad, pad])
    Hp, Wp = H + 2 * pad, W + 2 * pad
    Ho = (Hp - (dil * (K - 1) + 1)) // stride + 1
    Wo = (Wp - (dil * (K - 1) + 1)) // stride + 1
    P = K * K
    c0 = (dil * (K - 1)) // 2
    dev, dt = x.device, x.dtype
    ho = torch.arange(Ho, device=dev, dtype=dt).view(1, Ho, 1, 1, 1)
    wo = torch.arange(Wo, device=dev, dtype=dt).view(1, 1, Wo, 1, 1)
    kk = torch.arange(P, device=dev)
    iw = (kk // K).to(dt).view(1, 1, 1, 1, P)
    jh = (kk % K).to(dt).view(1, 1, 1, 1, P)
    off = offset.view(N, Ho, Wo, G, P, 2)
    px = wo * stride + c0 + (iw - (K - 1) // 2) * dil * offset_scale + off[..., 0] * offset_scale
    py = ho * stride + c0 + (jh - (K - 1) // 2) * dil * offset_scale + off[..., 1] * offset_scale
    x0 = torch.floor(px); y0 = torch.floor(py)
    lx = px - x0; ly = py - y0
    xg = xp.view(N, Hp * Wp, G, Cg)
    out = torch.zeros(N, Ho, Wo, G, Cg, device=dev, dtype=dt)
    m = mask.view(N, Ho, Wo, G, P)
    nidx = torch.arange(N, device=dev).view(N, 1, 1, 1, 1).expand(N, Ho, Wo, G, P)
    gidx = torch.arange(G, device=dev).view(1, 1, 1, G, 1).expand(N, Ho, Wo, G, P)
    for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
        yy = y0 + dy; xx = x0 + dx
        valid = (yy >= 0) & (yy <= Hp - 1) & (xx >= 0) & (xx <= Wp - 1)
        idx = (yy.clamp(0, Hp - 1) * Wp + xx.clamp(0, Wp - 1)).long()
        val = xg[nidx, idx, gidx]                                  # [N,Ho,Wo,G,P,Cg]
        out = out + (val * (wgt * valid.to(dt) * m).unsqueeze(-1)).sum(4)
    return out.reshape(N, Ho, Wo, G * Cg)


# ----------------------------------------------------------------------------- a11: sine positional encoding
def sine_pos_embed(B, H, W, num_feats, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """`SinePositionalEncoding.forward` with an all-False mask and normalize=True
    (mmdet/models/layers/positional_encoding.py:59-98) -> [B, 2*num_feats, H, W]."""
    y_embed = torch.arange(1, H + 1, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    x_embed = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------- the network
class OracleNet:
    """Functional forward of backbone (a3-a7) + MaskFormer head (a8-a12) over a flat params dict.

    `membranes` persists between `forward` calls like the reference's `MemoryModule` memories;
    `reset()` is `reset_net` (what `ResetModelHook` does before every iteration,
    mmseg/engine/hooks/resetmodel_hook.py:17-37).  `firing[name]` collects `mean(y * D)` per
    `Q_IFNode` of the last forward (the quantity `cal_firing_num.py:138-160` accumulates).
    """

    def __init__(self, params: Dict[str, torch.Tensor], cfg: ModelCfg, training: bool = True):
        self.p = params
        self.cfg = cfg
        self.training = training
        self.membranes: Dict[str, torch.Tensor] = {}
        self.firing: "OrderedDict[str, float]" = OrderedDict()
        self.keep_membrane_graph = False
        self.tap = None      # optional callable(name, tensor) for intermediate captures
        self.tap_in = None   # optional callable(name, h): every neuron's membrane before the threshold (the STE mask is 0 <= h <= D)
        self.stages = None   # optional dict: stage name -> (input, output) of every backbone stage / encoder layer
        self.force_diffs = {}   # per forced neuron: (elements whose own count differs, largest difference in levels, elements)
        self.force = None    # optional dict: neuron name -> spike COUNTS (uint8, this net's layout) to emit instead of the neuron's own:
        #                      re-seeds a teacher-forced comparison behind a neuron that legitimately flipped a level (tests only)

    def reset(self):
        self.membranes.clear()

    # ---- primitives
    def lif(self, name, x):
        if self.tap_in is not None:
            v = self.membranes.get(name)
            self.tap_in(name, (x if v is None else v + x).detach())
        y, v_new, s = lif_step(x, self.membranes.get(name), self.cfg.D)
        if self.force is not None and name in self.force:
            # emit the given counts; the straight-through gradient still flows through this neuron's own quantiser (same h)
            v_in = self.membranes.get(name)
            sf = self.force[name].to(x.dtype).reshape(x.shape)
            d = (sf - s.detach()).abs()
            self.force_diffs[name] = (int((d != 0).sum()), float(d.max()) if d.numel() else 0.0, d.numel())
            y = y + (sf / self.cfg.D - y).detach()
            v_new = v_new + (((x if v_in is None else v_in + x) - sf) - v_new).detach()
            s = sf
        self.membranes[name] = v_new if (self.keep_membrane_graph or not v_new.requires_grad) else v_new.detach()
        self.firing[name] = float(s.detach().mean())
        if self.tap is not None:
            self.tap(name, y)
        return y

    def conv2d(self, name, x, stride=1, padding=0, groups=1):
        return F.conv2d(x, self.p[name + ".weight"], self.p.get(name + ".bias"), stride, padding, 1, groups)

    def conv1d(self, name, x):
        return F.conv1d(x, self.p[name + ".weight"], self.p.get(name + ".bias"))

    def bn(self, name, x):
        p = self.p
        return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                            p[name + ".bias"], self.training, 0.1, 1e-5)

    # ---- backbone blocks (mmseg/models/backbones/sdtv2.py)
    def _down(self, n, x, k, s, pad, first):            # MS_DownSampling :386-421
        T, B = x.shape[:2]
        if not first:
            x = self.lif(n + ".encode_spike", x)
        x = self.bn(n + ".encode_bn", self.conv2d(n + ".encode_conv", x.flatten(0, 1), s, pad))
        return x.reshape(T, B, *x.shape[1:])

    def _sepconv(self, n, x):                            # SepConv :135-180
        T, B, C, H, W = x.shape
        x = self.lif(n + ".spike1", x)
        x = self.bn(n + ".bn1", self.conv2d(n + ".pwconv1", x.flatten(0, 1))).reshape(T, B, -1, H, W)
        x = self.lif(n + ".spike2", x)
        x = self.conv2d(n + ".dwconv", x.flatten(0, 1), 1, 3, x.shape[2])
        return self.bn(n + ".bn2", self.conv2d(n + ".pwconv2", x)).reshape(T, B, C, H, W)

    def _convblock(self, n, x):                          # MS_ConvBlock :183-219
        T, B, C, H, W = x.shape
        x = self._sepconv(n + ".Conv", x) + x
        feat = x
        x = self.lif(n + ".spike1", x)
        x = self.bn(n + ".bn1", self.conv2d(n + ".conv1", x.flatten(0, 1), 1, 1)).reshape(T, B, 4 * C, H, W)
        x = self.lif(n + ".spike2", x)
        x = self.bn(n + ".bn2", self.conv2d(n + ".conv2", x.flatten(0, 1), 1, 1)).reshape(T, B, C, H, W)
        return feat + x

    def _repconv_bn(self, n, x):                         # Sequential(RepConv, BN) :112-132, 48-89, 280-296
        p = self.p
        x = self.conv2d(n + ".0.body.0", x)
        b = n + ".0.body.1.bn"
        x = self.bn(b, x)
        # pad with the BN-of-zero value computed from the (just updated) *running* statistics, detached
        padv = (p[b + ".bias"].detach() - p[b + ".running_mean"] * p[b + ".weight"].detach()
                / torch.sqrt(p[b + ".running_var"] + 1e-5)).view(1, -1, 1, 1)
        N, C, H, W = x.shape
        xp = padv.expand(N, C, H + 2, W + 2).clone()
        xp[:, :, 1:-1, 1:-1] = x
        x = self.conv2d(n + ".0.body.2.0", xp, 1, 0, C)
        x = self.bn(n + ".0.body.2.2", self.conv2d(n + ".0.body.2.1", x))
        return self.bn(n + ".1", x)

    def _attn(self, n, x):                               # MS_Attention_RepConv_qkv_id :258-344
        T, B, C, H, W = x.shape
        N, nh = H * W, self.cfg.num_heads
        d = C // nh
        x = self.lif(n + ".head_spike", x)
        xf = x.flatten(0, 1)
        q = self._repconv_bn(n + ".q_conv", xf).reshape(T, B, C, H, W)
        k = self._repconv_bn(n + ".k_conv", xf).reshape(T, B, C, H, W)
        v = self._repconv_bn(n + ".v_conv", xf).reshape(T, B, C, H, W)

        def heads(name, t):
            t = self.lif(name, t).flatten(3)
            return t.transpose(-1, -2).reshape(T, B, N, nh, d).permute(0, 1, 3, 2, 4)

        q, k, v = heads(n + ".q_spike", q), heads(n + ".k_spike", k), heads(n + ".v_spike", v)
        kv = k.transpose(-2, -1) @ v
        o = (q @ kv) * (d ** -0.5)
        o = o.transpose(3, 4).reshape(T, B, C, N)
        o = self.lif(n + ".attn_spike", o)
        o = o.reshape(T, B, C, H, W).flatten(0, 1)
        return self._repconv_bn(n + ".proj_conv", o).reshape(T, B, C, H, W)

    def _mlp(self, n, x):                                # MS_MLP :222-255
        T, B, C, H, W = x.shape
        x = self.lif(n + ".fc1_spike", x.flatten(3))
        x = self.bn(n + ".fc1_bn", self.conv1d(n + ".fc1_conv", x.flatten(0, 1))).reshape(T, B, -1, H * W)
        x = self.lif(n + ".fc2_spike", x)
        return self.bn(n + ".fc2_bn", self.conv1d(n + ".fc2_conv", x.flatten(0, 1))).reshape(T, B, C, H, W)

    def _block(self, n, x):                              # MS_Block :347-383
        x = x + self._attn(n + ".attn", x)
        return x + self._mlp(n + ".mlp", x)

    def _stage(self, name, fn, x, *a):
        y = fn(name, x, *a)
        if self.stages is not None:
            self.stages[name] = (x.detach(), y.detach())
        return y

    def run_backbone_stage(self, name, x):
        """one stage of `backbone()` on its own (name as in `stages`): the teacher-forced re-runs of the full-size tests"""
        short = name[len("backbone."):]
        if short.startswith("downsample"):
            k, s, pad, first = {"downsample1_1": (7, 2, 3, True), "downsample4": (3, 1, 1, False)}.get(short, (3, 2, 1, False))
            return self._down(name, x, k, s, pad, first)
        return (self._convblock if short.startswith("ConvBlock") else self._block)(name, x)

    def backbone(self, img):                             # Spiking_vit_MetaFormer.forward_features :614-651
        b = "backbone."
        S = self._stage
        x = img.unsqueeze(0).repeat(self.cfg.T, 1, 1, 1, 1)
        x = S(b + "downsample1_1", self._down, x, 7, 2, 3, True)
        x = S(b + "ConvBlock1_1.0", self._convblock, x); x1 = x
        x = S(b + "downsample1_2", self._down, x, 3, 2, 1, False)
        x = S(b + "ConvBlock1_2.0", self._convblock, x); x2 = x
        x = S(b + "downsample2", self._down, x, 3, 2, 1, False)
        x = S(b + "ConvBlock2_1.0", self._convblock, x)
        x = S(b + "ConvBlock2_2.0", self._convblock, x); x3 = x
        x = S(b + "downsample3", self._down, x, 3, 2, 1, False)
        for i in range(6):
            x = S(b + f"block3.{i}", self._block, x)
        x = S(b + "downsample4", self._down, x, 3, 1, 1, False)
        for i in range(2):
            x = S(b + f"block4.{i}", self._block, x)
        return [x1, x2, x3, x]

    # ---- head blocks
    def _sepconv_spike(self, n, x, k):                   # SepConv_Spike, mmcv_spike/SNN_core.py:11-63 (NHWC in/out)
        T, B, H, W, C = x.shape
        x = x.permute(0, 1, 4, 2, 3)
        x = self.lif(n + ".spike1", x)
        x = self.bn(n + ".pwconv1.1", self.conv2d(n + ".pwconv1.0", x.flatten(0, 1))).reshape(T, B, 2 * C, H, W)
        x = self.lif(n + ".spike2", x)
        x = self.bn(n + ".dwconv.1", self.conv2d(n + ".dwconv.0", x.flatten(0, 1), 1, (k - 1) // 2, 2 * C))
        x = self.lif(n + ".spike3", x.reshape(T, B, 2 * C, H, W))
        x = self.bn(n + ".pwconv2.1", self.conv2d(n + ".pwconv2.0", x.flatten(0, 1))).reshape(T, B, C, H, W)
        return x.permute(0, 1, 3, 4, 2)

    def _dcn(self, n, inp):                              # DCNv3_pytorch.forward, ops_dcnv3/modules/dcnv3.py:198-233
        cfg = self.cfg
        T, N, H, W, C = inp.shape
        G = cfg.group
        x = self._sepconv_spike(n + ".input_proj", inp, cfg.dw_kernel_size)
        x1 = self.lif(n + ".dw_spike", inp.permute(0, 1, 4, 2, 3))
        x1 = self.bn(n + ".dw_conv.1", self.conv2d(n + ".dw_conv.0", x1.flatten(0, 1), 1,
                                                   (cfg.dw_kernel_size - 1) // 2, C)).reshape(T, N, C, H, W)
        x1 = self.lif(n + ".offset_spike", x1)
        # NOTE (bug-compatible): [T*N, C', H, W] is *reinterpreted* as [T,N,H,W,C'] (dcnv3.py:213-214)
        offset = self.bn(n + ".offset.1", self.conv2d(n + ".offset.0", x1.flatten(0, 1))).reshape(T, N, H, W, -1)
        mask = self.bn(n + ".mask.1", self.conv2d(n + ".mask.0", x1.flatten(0, 1))).reshape(T, N, H, W, -1)
        mask = self.lif(n + ".mask_spike", mask)
        y = dcnv3_core(x.flatten(0, 1), offset.flatten(0, 1), mask.flatten(0, 1), G, C // G)
        return self._sepconv_spike(n + ".output_proj", y.reshape(T, N, H, W, C), cfg.dw_kernel_size)

    def _pd_mlp(self, n, x):                             # MS_MLP, mmcv_spike/transformer.py:787-831
        T, B, H, W, C = x.shape
        x = x.permute(0, 1, 4, 2, 3).flatten(3)
        x = self.lif(n + ".fc1_spike", x)
        x = self.bn(n + ".fc1_bn", self.conv1d(n + ".fc1_conv", x.flatten(0, 1))).reshape(T, B, -1, H * W)
        x = self.lif(n + ".fc2_spike", x)
        # NOTE (bug-compatible): [T*B, C, N] reinterpreted as [T,B,H,W,C] (:829)
        return self.bn(n + ".fc2_bn", self.conv1d(n + ".fc2_conv", x.flatten(0, 1))).reshape(T, B, H, W, C)

    def _enc_layer(self, n, q):                          # DCNDetrTransformerEncoderLayer, detr_layers.py:263-339
        p = self.p
        q = q + p[n + ".gamma1"] * self._sepconv_spike(n + ".Conv", q, 3)
        q = q + p[n + ".gamma2"] * self._dcn(n + ".dcn", q)
        return q + p[n + ".gamma3"] * self._pd_mlp(n + ".ffn", q)

    def pixel_decoder(self, feats):                      # DCNTransformerEncoderPixelDecoder.forward, pixel_decoder.py:417-472
        n = "decode_head.pixel_decoder"
        Fc = self.cfg.feat_channels
        x4 = feats[-1]
        t, bs, c, h, w = x4.shape
        y = self.lif(n + ".last_feat_conv_spike", x4)
        y = self.bn(n + ".encoder_in_proj.1", self.conv2d(n + ".encoder_in_proj.0", y.flatten(0, 1)))
        q = y.reshape(t, bs, Fc, h, w).permute(0, 1, 3, 4, 2)
        for i in range(self.cfg.pd_layers):
            q = self._stage(n + f".encoder.layers.{i}", self._enc_layer, q)
        memory = q.permute(0, 1, 4, 2, 3).contiguous()
        memory = self.lif(n + ".encoder_out_proj_spike", memory)
        y = self.bn(n + ".encoder_out_proj.1", self.conv2d(n + ".encoder_out_proj.0", memory.flatten(0, 1)))
        y = y.reshape(t, bs, Fc, h, w)
        out = [y]
        for i in (2, 1, 0):
            x = self.lif(n + f".lateral_convs_spike.{i}", feats[i])
            cur = self.bn(n + f".lateral_convs.{i}.1", self.conv2d(n + f".lateral_convs.{i}.0", x.flatten(0, 1)))
            y = cur + F.interpolate(y.flatten(0, 1), size=cur.shape[-2:], mode="bilinear", align_corners=False)
            _, C, H, W = y.shape
            y = self.lif(n + f".output_convs_spike.{i}", y.reshape(t, bs, C, H, W))
            y = self.bn(n + f".output_convs.{i}.1", self.conv2d(n + f".output_convs.{i}.0", y.flatten(0, 1), 1, 1, C))
            y = y.reshape(t, bs, C, H, W)
            out.append(y)
        y = self.lif(n + ".mask_feature_spike", y)
        mf = self.conv2d(n + ".mask_feature", y.flatten(0, 1))
        return mf.reshape(t, bs, *mf.shape[1:]), memory, out[:3]

    def _dec_attn(self, n, query, key, value, attn_mask=None):   # (Cross)MultiHeadAttentionBlock, mmcv_spike/transformer.py:196-361
        t, b, nq, dim = query.shape
        nk = key.shape[2]
        nh = self.cfg.num_heads
        d = dim // nh

        def proj(name, x, L):
            x = self.lif(n + f".{name}_conv_spike", x).permute(0, 1, 3, 2)
            x = self.bn(n + f".{name}_conv.1", self.conv1d(n + f".{name}_conv.0", x.flatten(0, 1)))
            x = self.lif(n + f".{name}_spike", x.permute(0, 2, 1).reshape(t, b, L, dim))
            return x.reshape(t, b, L, nh, d).permute(0, 1, 3, 2, 4)

        q, k, v = proj("q", query, nq), proj("k", key, nk), proj("v", value, nk)
        scores = (q @ k.transpose(3, 4)) / (dim ** 0.5)   # embed_dim**0.5, not head-dim; no softmax (:263, :346)
        if attn_mask is not None:
            # :266-269, :349-352 -- reshaped with querys.shape[0] = t where the comment says bs, then broadcast against
            # [t, b, heads, nq, nk]: runs only for t == b (or b == 1), applying mask[b, h] to every time step
            scores = scores.masked_fill(attn_mask.reshape(q.shape[0], nh, nq, nk), 0)
        o = scores @ v
        o = o.permute(0, 1, 3, 2, 4).reshape(t, b, nq, dim)
        o = self.lif(n + ".attn_spike", o).permute(0, 1, 3, 2)
        o = self.bn(n + ".out_conv.1", self.conv1d(n + ".out_conv.0", o.flatten(0, 1)))
        return o.permute(0, 2, 1).reshape(t, b, nq, dim)

    def _dec_ffn(self, n, x):                            # MSDA_FFN, mmcv_spike/transformer.py:776-784
        t, bs, N, C = x.shape
        Fh = self.cfg.dec_ffn
        a = self.lif(n + ".fc1_spike", x).reshape(t, bs, C, N)        # reinterpretation, not a transpose (:777)
        a = self.bn(n + ".bn1", self.conv1d(n + ".fc1", a.flatten(0, 1))).reshape(t, bs, Fh, N)
        a = self.lif(n + ".fc2_spike", a)
        return self.bn(n + ".bn2", self.conv1d(n + ".fc2", a.flatten(0, 1))).reshape(t, bs, N, C)   # (:781)

    def _dec_layer(self, n, query, key, query_pos, key_pos, self_attn_mask=None, cross_attn_mask=None):
        # DetrTransformerDecoderLayer.forward, detr_layers.py:491-559
        ca = self._dec_attn(n + ".cross_attn.attn", query + query_pos, key + key_pos, key, cross_attn_mask)
        query = query + ca
        sa = self._dec_attn(n + ".self_attn.attn", query + query_pos, query + query_pos, query, self_attn_mask)
        query = query + sa
        return query + self._dec_ffn(n + ".ffn", query)

    def head(self, feats):                               # mmdet MaskFormerHead.forward, dense_heads/maskformer_head.py:498-586
        cfg, p = self.cfg, self.p
        h = "decode_head."
        mask_features, memory, msm = self.pixel_decoder(feats)
        t, bs = memory.shape[:2]
        query_feat = p[h + "query_feat.weight"].unsqueeze(0).repeat(t, bs, 1, 1)
        query_embed = p[h + "query_embed.weight"].unsqueeze(0).repeat(bs, 1, 1)
        dec_in, dec_pos = [], []
        for i in range(3):
            x = msm[i].flatten(3).permute(0, 1, 3, 2) + p[h + "level_embed.weight"][i].view(1, 1, -1)
            pe = sine_pos_embed(bs, msm[i].shape[-2], msm[i].shape[-1], cfg.num_feats).to(x.dtype)
            dec_in.append(x)
            dec_pos.append(pe.flatten(2).permute(0, 2, 1))
        outs = [query_feat]
        for i in range(cfg.dec_layers):
            lv = i % 3
            query_feat = self._dec_layer(h + f"transformer_decoder.layers.{i}", query_feat, dec_in[lv],
                                         query_embed, dec_pos[lv])
            outs.append(query_feat)
        return self._sdme(torch.stack(outs), mask_features)

    def _sdme(self, O, mask_features):                   # SDME block + mask contraction, dense_heads/maskformer_head.py:568-586
        p = self.p
        h = "decode_head."
        ln, t, bs, nq, C = O.shape
        Z = torch.sigmoid(O)
        A = 4 * self.lif(h + "decoder_out_spike", Z)
        cls = F.linear(A, p[h + "cls_embed.weight"], p[h + "cls_embed.bias"]).mean(1)
        m = F.linear(A, p[h + "mask_embed.fc1.weight"])
        m = self.lif(h + "mask_embed.spike1", m) * 4
        m = F.linear(m, p[h + "mask_embed.fc2.weight"])
        m = self.lif(h + "mask_embed.spike2", m) * 4
        m = F.linear(m, p[h + "mask_embed.fc_out.weight"], p[h + "mask_embed.fc_out.bias"])
        sc = (4 * self.lif(h + "shortcut_conv_spike", Z)).reshape(ln * t * bs, nq, C)
        sc = self.bn(h + "shortcut_conv.1", self.conv1d(h + "shortcut_conv.0", sc)).view(ln, t, bs, nq, C)
        E = m + p[h + "w"] * sc
        E = 4 * self.lif(h + "mask_embed_spike", E)
        masks = torch.einsum("ltbqc,tbchw->ltbqhw", E, mask_features).mean(1)
        return cls, masks

    def forward(self, img):
        self.firing = OrderedDict()
        return self.head(self.backbone(img))

    def predict(self, img):                              # mmseg MaskFormerHead.predict, decode_heads/maskformer_head.py:138-180
        cls, masks = self.forward(img)
        mp = F.interpolate(masks[-1], size=img.shape[-2:], mode="bilinear", align_corners=False)
        return torch.einsum("bqc,bqhw->bchw", F.softmax(cls[-1], dim=-1)[..., :-1], mp.sigmoid())


def headline_loss(cls, masks):
    """The benchmark's scalar loss: keeps the step on-device and data-independent."""
    return cls.float().mean() + masks.float().mean()


def make_params(cfg: ModelCfg, requires_grad=True) -> "OrderedDict[str, torch.Tensor]":
    st = seeded_state(param_shapes(cfg))
    if requires_grad:
        for k, v in st.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
    return st


def synthetic_image(cfg: ModelCfg, seed=0):
    return torch.randn(cfg.B, 3, cfg.H, cfg.W, generator=torch.Generator().manual_seed(seed))
