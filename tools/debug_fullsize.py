import os, sys, dataclasses, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from oracle import s2f_oracle as so
if len(sys.argv) > 1 and sys.argv[1] == "nogemm": ops.SPIKE_GEMM_ENABLED = False
cfg = dataclasses.replace(so.CONFIGS["C2"], B=1)
st = so.make_params(cfg, requires_grad=False)
model = s2f.MODELS.build(s2f.model_cfg("C2")); model.load_state_dict(st, strict=True); model.cuda().train()
img = so.synthetic_image(cfg, seed=7)
net = so.OracleNet(st, cfg, training=True)
with torch.no_grad(): ocls, omasks = net.forward(img)
model.load_state_dict(st, strict=True)
s2f.reset_net(model)
with torch.no_grad(), s2f.FiringRecorder(model) as rec:
    cls, masks = model(img.cuda()); rec.collect()
t = rec.result()["t0"]
names = list(t)
d = np.array([abs(t[k]-net.firing[k]) for k in names])
for i,k in enumerate(names):
    if d[i] > 2e-4 or i % 25 == 0: print(f"{i:3d} {d[i]:.2e} {t[k]:.4f} {k}")
print("max", d.max(), "cls rel", ((cls.cpu()-ocls).abs().max()/ocls.abs().max()).item(), "mask rel", ((masks.cpu()-omasks).abs().max()/omasks.abs().max()).item())
