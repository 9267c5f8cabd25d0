"""Worker of tests/test_gpu_dist.py::test_rccl_communicator_and_hipgraph_capture_coexist: the N > 1 bench path in ONE process --
with S2F_FORCE_DIST=1 under a live RCCL communicator (backend nccl, world 1), without it under no process group at all.
Writes {loss of the second replay, flat gradient buffer after reduce()} to argv[1]."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f                                               # noqa: E402
from spike2former_amd import ops                                             # noqa: E402
from spike2former_amd.dist import FlatGradAllReduce, broadcast_params, init_process_group      # noqa: E402
from spike2former_amd.graph import GraphedStep                              # noqa: E402
from spike2former_amd.init_utils import seeded_init                         # noqa: E402

rank, world, local = init_process_group()
backend = dist.get_backend() if dist.is_initialized() else "none"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = s2f.WORKLOADS["C1_64"]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).to(dev).train()
broadcast_params(model)
s2f.set_keep_membrane(model, False)
red = FlatGradAllReduce(model.parameters(), world)
red.install_sinks()
img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).to(dev)
state = {k: v.clone() for k, v in model.state_dict().items()}


def eager():
    s2f.reset_net(model)
    red.zero()
    cls, masks = model(img)
    s2f.headline_loss(cls, masks).backward()
    ops.wgrad_join()
    red.gather()


eager()
red.compact()
model.load_state_dict(state)
step = GraphedStep(model, s2f.headline_loss, img, grad_buffer=red, warmup=2)
model.load_state_dict(state)
step()
red.reduce()
red.wait()
model.load_state_dict(state)
loss = step()
red.reduce()                                   # the averaging all-reduce over one rank: RCCL's ncclAvg on the flat buffer
red.wait()
torch.cuda.synchronize()
torch.save({"loss": loss.detach().cpu().clone(), "flat": red.flat.detach().cpu().clone()}, sys.argv[1])
print(f"backend {backend} world {world} loss {float(loss):.9e}", flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
