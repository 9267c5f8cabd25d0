"""Debug: fused key/value neurons and branch streams vs the plain path on the tiny model (max differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spike2former_amd as s2f
from spike2former_amd import ops, maskformer_head as mh
from oracle import s2f_oracle as so

cfg = so.CONFIGS["C1_64"]
model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
model.load_state_dict(so.make_params(cfg, requires_grad=False), strict=True)
model = model.cuda().train()
s2f.set_keep_membrane(model, False)
img = so.synthetic_image(cfg, seed=5).cuda()
state = {k: v.clone() for k, v in model.state_dict().items()}


LONG = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(fused, side, long=False):
    mh.FUSED_KV_NEURONS = fused
    ops.BRANCH_STREAMS = side
    ops.LONG_STREAMS = LONG if long else None
    model.load_state_dict(state)
    s2f.reset_net(model); model.zero_grad(set_to_none=True)
    cls, masks = model(img)
    s2f.headline_loss(cls, masks).backward()
    torch.cuda.synchronize()
    return cls.detach().clone(), masks.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


def cmp(a, b, tag):
    gscale = max(v.abs().max().item() for v in b[2].values())
    worst = max(((a[2][k] - b[2][k]).abs().max().item() / (b[2][k].abs().max().item() + 1e-3 * gscale), k) for k in a[2])
    print(f"{tag}: cls {(a[0]-b[0]).abs().max().item():.3e} masks {(a[1]-b[1]).abs().max().item():.3e} worst grad rel {worst[0]:.3e} at {worst[1]}")


side = [torch.cuda.Stream(), torch.cuda.Stream()]
r0 = run(False, None); r1 = run(False, None); cmp(r1, r0, "plain vs plain")
r2 = run(True, None); cmp(r2, r0, "fused kv vs plain")
r3 = run(False, side); cmp(r3, r0, "branch streams vs plain")
r4 = run(False, side); cmp(r4, r3, "branch streams vs branch streams")
r5 = run(True, side); cmp(r5, r0, "both vs plain")
r6 = run(True, side, True); cmp(r6, r0, "all streams vs plain")
r7 = run(False, None, True); cmp(r7, r0, "long streams, unfused kv vs plain")
r8 = run(True, side, True); cmp(r8, r6, "all streams vs all streams")
