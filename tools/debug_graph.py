import os, sys, dataclasses, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd.graph import GraphedStep
from spike2former_amd.init_utils import seeded_init
for name, B in (("C1_64", 2), ("C2", 1)):
    w = s2f.WORKLOADS[name]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg(name))).cuda().train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    s2f.set_keep_membrane(model, False)
    img = torch.randn(B, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()
    def eager():
        model.load_state_dict(sd); s2f.reset_net(model)
        with torch.no_grad():
            return float(s2f.headline_loss(*model(img)))
    def eager_grad():
        model.load_state_dict(sd); s2f.reset_net(model); model.zero_grad(set_to_none=True)
        l = s2f.headline_loss(*model(img)); l.backward(); return float(l)
    print(name, "eager nograd", eager(), eager(), "eager grad", eager_grad(), eager_grad())
    gs = GraphedStep(model, s2f.headline_loss, img, warmup=1)
    outs = []
    for _ in range(3):
        model.load_state_dict(sd); outs.append(float(gs()))
    print(name, "graph", outs, "eager after", eager_grad())
