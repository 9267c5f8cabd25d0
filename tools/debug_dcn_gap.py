"""Bisect the C1_64 gradient gap: the DCN module of encoder layer 0 / 1 on a seeded input against the live oracle, per parameter;
`core=oracle` swaps the HIP sampling core for the oracle's torch restatement (on the GPU) inside this build's module."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f  # noqa: E402
from oracle import s2f_oracle as so  # noqa: E402
from spike2former_amd import head_layers, ops  # noqa: E402

cfg = so.CONFIGS["C1_64"]
st0 = so.make_params(cfg, requires_grad=False)
model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
model.load_state_dict(st0, strict=True)
model.cuda().train()
T, B, Fc = cfg.T, cfg.B, cfg.feat_channels


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def run(layer, H, W, seed, core):
    name = f"decode_head.pixel_decoder.encoder.layers.{layer}.dcn"
    mod = model.decode_head.pixel_decoder.encoder.layers[layer].dcn
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, H, W, Fc, generator=g) * 2
    gy = torch.randn(T, B, H, W, Fc, generator=g)
    model.load_state_dict(st0, strict=True)
    for p in mod.parameters():
        p.grad = None
    s2f.reset_net(model)
    real = ops.dcnv3_core
    if core == "oracle":
        def fake(xx, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, Cg, osc):
            return so.dcnv3_core(xx, off, msk, G, Cg, kh, sh, ph, dh, osc)
        head_layers.ops.dcnv3_core = fake
    try:
        xg = x.cuda().requires_grad_(True)
        y = mod(xg)
        y.backward(gy.cuda())
    finally:
        head_layers.ops.dcnv3_core = real
    st = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k) if k.startswith(name) else v) for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    xo = x.clone().requires_grad_(True)
    yo = net._dcn(name, xo)
    yo.backward(gy)
    print(f"layer {layer} map {H}x{W} core={core}: y {rel(y.detach().cpu(), yo.detach()):.1e}  gx {rel(xg.grad.cpu(), xo.grad):.1e}")
    for n, p in mod.named_parameters():
        r = st[name + "." + n].grad
        if r is None or p.grad is None or r.abs().max() < 1e-12:
            continue
        e = rel(p.grad.cpu(), r)
        if e > 1e-4:
            print(f"     {e:9.2e}  {n}")


for core in ("hip", "oracle"):
    for layer in (0, 1):
        for (H, W) in ((4, 4), (6, 5), (8, 8)):
            run(layer, H, W, 11 + layer, core)
