"""Pixel-decoder encoder layer i fed the oracle's input: per-neuron flip counts vs the oracle's taps (where does a difference start?)."""
import os, sys, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from oracle import s2f_oracle as so
cfg = dataclasses.replace(so.CONFIGS["C2"], B=1)
st0 = so.make_params(cfg, requires_grad=False)
model = s2f.MODELS.build(s2f.model_cfg("C2")); model.load_state_dict(st0); model.cuda().train()
st = {k: v.clone() for k, v in st0.items()}
net = so.OracleNet(st, cfg, True); net.stages = {}
taps = {}
net.tap = lambda n, y: taps.__setitem__(n, y)
img = so.synthetic_image(cfg, seed=7)
with torch.no_grad():
    feats = net.backbone(img); net.pixel_decoder(feats)
def rel(a, b): return ((a - b).norm() / b.norm()).item()
pd = model.decode_head.pixel_decoder
for i in range(cfg.pd_layers):
    name = f"decode_head.pixel_decoder.encoder.layers.{i}"
    x, y = net.stages[name]
    mod = pd.encoder.layers[i]
    model.load_state_dict(st0); s2f.reset_net(model)
    sp = {}
    hooks = [m.register_forward_hook(lambda mm, inp, o, n=n: sp.__setitem__(n, o.detach().cpu())) for n, m in mod.named_modules() if isinstance(m, s2f.Q_IFNode)]
    with torch.no_grad(): out = mod(x.cuda())
    for h in hooks: h.remove()
    print(f"layer {i}: relL2 {rel(out.cpu(), y):.2e}")
    for n, v in sp.items():
        r = taps.get(name + "." + n)
        if r is None: continue
        v = v.reshape(-1); r = r.reshape(-1)
        if v.numel() != r.numel(): print("   ", n, "shape mismatch"); continue
        # NCHW (ours) vs NHWC (oracle) orders differ for some neurons: compare sorted-invariant count and direct
        d = (v - r).abs()
        print(f"    {n:28s} differing elements {int((d > 0).sum()):8d} of {v.numel()}  max |d| {d.max().item():.3f}  sum(v)-sum(r) {float(v.double().sum() - r.double().sum()):+.3f}")
