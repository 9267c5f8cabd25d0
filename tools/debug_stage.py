import os, sys, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from oracle import s2f_oracle as so
cfg = dataclasses.replace(so.CONFIGS["C2"], B=1)
st0 = so.make_params(cfg, requires_grad=False)
model = s2f.MODELS.build(s2f.model_cfg("C2")); model.load_state_dict(st0); model.cuda().train()
st = {k: v.clone() for k, v in st0.items()}
net = so.OracleNet(st, cfg, True); net.stages = {}
taps = {}
net.tap = lambda n, y: taps.__setitem__(n, y)
img = so.synthetic_image(cfg, seed=7)
with torch.no_grad(): net.backbone(img)
def rel(a, b): return ((a-b).norm()/b.norm()).item()
for name in ("backbone.ConvBlock2_1.0", "backbone.ConvBlock2_2.0", "backbone.block3.4", "backbone.block4.0"):
    x, y = net.stages[name]
    mod = model.backbone
    for part in name[9:].split("."): mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
    for gemm in (True, False):
        ops.SPIKE_GEMM_ENABLED = gemm
        model.load_state_dict(st0); s2f.reset_net(model)
        sp = {}
        hooks = [m.register_forward_hook(lambda mm, i, o, n=n: sp.__setitem__(n, o.detach().cpu())) for n, m in mod.named_modules() if isinstance(m, s2f.Q_IFNode)]
        with torch.no_grad(): out = mod(x.cuda())
        for h in hooks: h.remove()
        line = f"{name} spike_gemm={gemm} relL2={rel(out.cpu(), y):.2e}"
        for n, v in sp.items():
            r = taps[name + "." + n]
            d = (v.reshape(r.shape) - r).abs()
            line += f" | {n}: flips {(d>0).float().mean().item():.1e}"
        print(line)
