import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f
from spike2former_amd import ops
from spike2former_amd.init_utils import seeded_init
from spike2former_amd.neuron import reset_net
NMOD = int(os.environ.get("NMOD", "3"))
HOLD = os.environ.get("HOLD", "1") == "1"
w = s2f.WORKLOADS["C1_64"]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
s2f.set_keep_membrane(model, False)
bb = model.backbone
img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()
chain = [bb.downsample1_1, *bb.ConvBlock1_1, bb.downsample1_2, *bb.ConvBlock1_2, bb.downsample2, *bb.ConvBlock2_1, *bb.ConvBlock2_2,
         bb.downsample3, *bb.block3, bb.downsample4, *bb.block4][:NMOD]
params = [p for m in chain for p in m.parameters()]
held = []
orig_init = ops.Spikes.__init__
def spy_init(self, data, tok=None):
    orig_init(self, data, tok)
    if HOLD and torch.cuda.is_current_stream_capturing() and not any(d is data for d in held):
        held.append(data)
ops.Spikes.__init__ = spy_init
def step():
    reset_net(model)
    for p in params: p.grad = None
    x = img.unsqueeze(0).repeat(bb.T, 1, 1, 1, 1)
    for i, m in enumerate(chain):
        x = m(x, next_lif=chain[i + 1].first_lif if i + 1 < len(chain) else None)
    loss = (x * x).mean()
    loss.backward()
    return loss.detach()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
ops.resplit_all(img.device)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    sl = step()
torch.cuda.synchronize()
static = [p.grad for p in params]
g.replay(); torch.cuda.synchronize()
g1 = [None if t is None else t.clone() for t in static]; h1 = [t.clone() for t in held]
for p in params:
    junk = 0.05 * p.abs().mean() * torch.randn(p.shape, device=p.device)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
garb = [i for i, (a, b) in enumerate(zip(static, g1)) if a is not None and (a - b).abs().max().item() > 1e3 * max(b.abs().max().item(), 1e-3)]
print("NMOD", len(chain), "HOLD", HOLD, "held spike tensors", len(held), "garbage grads:", len(garb), garb[:8])
for i, (a, b) in enumerate(zip(held, h1)):
    if not torch.equal(a, b):
        d = (a.float() - b.float()).flatten()
        idx = d.nonzero().flatten()
        print("  spike tensor", i, tuple(a.shape), a.dtype, "differs in", idx.numel(), "elements; first", idx[:8].tolist(), "last", idx[-3:].tolist(),
              "values now", a.flatten()[idx[:4]].tolist())
