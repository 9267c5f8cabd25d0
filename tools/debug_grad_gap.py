"""Where does this build's gradient leave the oracle's?  C1_64 train step on the fixture image: per-parameter gap in the generator's
metric max|d| / (max|g| + 5e-3 gradient scale), in module order (the backward runs from the bottom of the list to the top).
    python tools/debug_grad_gap.py [switch=value ...]      e.g.  SPIKES_BF16=0"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f  # noqa: E402
from oracle import s2f_oracle as so  # noqa: E402
from spike2former_amd import ops  # noqa: E402

for a in sys.argv[1:]:
    k, v = a.split("=")
    setattr(ops, k, type(getattr(ops, k))(int(v)))
cfg = so.CONFIGS["C1_64"]
st = so.make_params(cfg)
model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
model.load_state_dict({k: v.detach() for k, v in st.items()}, strict=True)
model.cuda().train()
img = so.synthetic_image(cfg)
s2f.reset_net(model)
cls, masks = model(img.cuda())
s2f.headline_loss(cls, masks).backward()
net = so.OracleNet(st, cfg, training=True)
ocls, omasks = net.forward(img)
so.headline_loss(ocls, omasks).backward()
print("out rel", ((cls.detach().cpu() - ocls).abs().max() / ocls.abs().max()).item(), ((masks.detach().cpu() - omasks).abs().max() / omasks.abs().max()).item())
gscale = max(v.grad.abs().max().item() for v in st.values() if v.grad is not None)
print("gradient scale", gscale)
for k, p in model.named_parameters():
    ref = st[k].grad
    if ref is None:
        continue
    mine = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
    d = (mine - ref).abs().max().item()
    gap = d / (ref.abs().max().item() + 5e-3 * gscale)
    flag = " <<<" if gap > 1e-4 else ""
    print(f"{gap:9.2e} {d / (ref.abs().max().item() + 1e-30):9.2e} {ref.abs().max().item():10.3e}  {k}{flag}")
