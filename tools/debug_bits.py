"""Debug: bench-path forward vs hooked (module-call) forward vs oracle on the oracle's C2 stage inputs."""
import sys, dataclasses, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import spike2former_amd as s2f
from oracle import s2f_oracle as so
import test_gpu_full_size as T
cfg = dataclasses.replace(so.CONFIGS["C2"], B=1)
st0 = so.make_params(cfg, requires_grad=False)
model = s2f.MODELS.build(s2f.model_cfg("C2")); model.load_state_dict(st0, strict=True); model = model.cuda().train()
img = so.synthetic_image(cfg, seed=7)
st = {k: v.clone() for k, v in st0.items()}
net = so.OracleNet(st, cfg, training=True)
net.stages = {}
with torch.no_grad():
    net.backbone(img)
for k in range(6):
    name = f"backbone.block3.{k}"
    x, y = net.stages[name]
    mod = model.backbone.block3[k]
    outs = []
    for hooked in (False, True, False):
        model.load_state_dict(st0, strict=True)
        s2f.reset_net(model)
        hooks = [m.register_forward_hook(lambda *a: None) for n, m in mod.named_modules() if isinstance(m, s2f.Q_IFNode)] if hooked else []
        with torch.no_grad():
            outs.append(mod(x.cuda()).cpu())
        for h in hooks: h.remove()
    print(name, "bench-oracle %.2e hooked-oracle %.2e bench-hooked %.2e bench-bench %.2e" % (
        T.rel_l2(outs[0], y), T.rel_l2(outs[1], y), T.rel_l2(outs[0], outs[1]), T.rel_l2(outs[0], outs[2])))
