"""Worker of tests/test_gpu_dist.py: one rank of a world-size-2 job whose ranks SHARE GPU 0 (gloo carries the collectives).
Runs the N > 1 bench path on the real model: broadcast_params, FlatGradAllReduce with gradient sinks + deferred grouped weight
gradients, GraphedStep capture with a process group alive, replay, reduce(); checks the reduced buffer against the mean of the
two shards' plain-autograd gradients.  Prints `RANK r OK` or raises."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f                                               # noqa: E402
from spike2former_amd import ops                                             # noqa: E402
from spike2former_amd.dist import FlatGradAllReduce, broadcast_params, init_process_group, shard_batch      # noqa: E402
from spike2former_amd.graph import GraphedStep                              # noqa: E402
from spike2former_amd.init_utils import seeded_init                         # noqa: E402

rank, world, local = init_process_group()
assert world == 2 and dist.get_backend() == "gloo"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = s2f.WORKLOADS["C1_64"]
model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).to(dev).train()
with torch.no_grad():                                        # rank 1 starts from different weights: the broadcast must fix it
    if rank == 1:
        for p in model.parameters():
            p.mul_(1.25)
broadcast_params(model)
chk = torch.cat([p.detach().flatten() for p in model.parameters()] + [b.detach().flatten().float() for b in model.buffers()])
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1]), "parameters differ after broadcast_params"
s2f.set_keep_membrane(model, False)

G = 2 * w["B"]                                               # global batch, identical on every rank; each takes its shard
imgs = torch.randn(G, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).to(dev)
start, per = shard_batch(G, rank, world)

# reference: plain autograd gradients of BOTH shards on this rank (no sinks, no graph), averaged
state = {k: v.clone() for k, v in model.state_dict().items()}


def plain_grads(x):
    model.load_state_dict(state)                             # the same running statistics before every step
    for p in model.parameters():
        p.grad = None
    s2f.reset_net(model)
    cls, masks = model(x)
    s2f.headline_loss(cls, masks).backward()
    return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}


ga, gb = plain_grads(imgs[:per]), plain_grads(imgs[per:])
want = {n: (ga[n] + gb[n]) / 2 for n in ga}

# the bench path
model.load_state_dict(state)
red = FlatGradAllReduce(model.parameters(), world)
red.install_sinks()
mine = imgs[start:start + per].clone()


def eager():
    s2f.reset_net(model)
    red.zero()
    cls, masks = model(mine)
    s2f.headline_loss(cls, masks).backward()
    ops.wgrad_join()
    red.gather()


eager()
red.compact()
model.load_state_dict(state)
step = GraphedStep(model, s2f.headline_loss, mine, grad_buffer=red, warmup=2)
model.load_state_dict(state)
step()
red.reduce()
red.wait()
torch.cuda.synchronize()
name_of = {id(p): n for n, p in model.named_parameters()}
gscale = max(v.abs().max().item() for v in want.values())
worst = 0.0
for p, v in zip(red.params, red.views):
    ref = want[name_of[id(p)]]
    worst = max(worst, (v - ref).abs().max().item() / (ref.abs().max().item() + 1e-3 * gscale))
assert worst <= 2e-3, f"reduced gradients differ from the mean of the shards' gradients: {worst}"
assert all(len(j) == 0 for j in list(ops._DW_PENDING.values()))
# the real training step under the live process group: graph.GraphedHungarianStep (the assignment's num_masks is averaged over
# the ranks between its two graphs, maskformer_head.py:459) against the eager mode="loss" step on this rank's shard
from spike2former_amd.graph import GraphedHungarianStep                     # noqa: E402
del step
gen = torch.Generator().manual_seed(17)
seg = torch.empty(G, 1, w["H"], w["W"], dtype=torch.int64)
for i in range(G):                                            # rank-dependent number of classes: the averaged num_masks matters
    classes = torch.randperm(w["K"], generator=gen)[:3 + 2 * (i // per)]
    for j, (y0, x0) in enumerate((y, x) for y in range(0, w["H"], w["H"] // 4) for x in range(0, w["W"], w["W"] // 4)):
        seg[i, 0, y0:y0 + w["H"] // 4, x0:x0 + w["W"] // 4] = classes[j % len(classes)]
seg = seg.to(dev)[start:start + per]
model.load_state_dict(state)
s2f.reset_net(model)
red.zero()
losses = model(mine, [seg[i] for i in range(per)], mode="loss")
sum(losses.values()).backward()
ops.wgrad_join()
red.gather()
want_loss, want_flat = {k: float(v) for k, v in losses.items()}, red.flat.clone()
del losses
for p in model.parameters():
    p.grad = None
import gc
gc.collect()
model.load_state_dict(state)
hstep = GraphedHungarianStep(model, mine, seg, red, warmup=1)
model.load_state_dict(state)
got = hstep()
torch.cuda.synchronize()
assert all(abs(float(v) - want_loss[k]) <= 1e-6 * max(abs(want_loss[k]), 1e-3) for k, v in got.items()), "Hungarian graph step: losses"
assert (red.flat - want_flat).abs().max().item() <= 1e-4 * want_flat.abs().max().item(), "Hungarian graph step: gradients"
# the parameter update behind the all-reduce: train.FlatAdamW on the averaged flat buffer against
# clip_grad_norm_ + torch.optim.AdamW on the same gradients; both ranks must end with identical parameters
from spike2former_amd.train import FlatAdamW                                 # noqa: E402
del hstep
model.load_state_dict(state)
CUSTOM = dict(custom_keys={"backbone": dict(lr_mult=0.1, decay_mult=1.0), "query_embed": dict(lr_mult=1.0, decay_mult=0.0),
                           "query_feat": dict(lr_mult=1.0, decay_mult=0.0), "level_embed": dict(lr_mult=1.0, decay_mult=0.0)})
opt = FlatAdamW(model, red, lr=0.001, weight_decay=0.005, paramwise_cfg=CUSTOM, clip_grad=dict(max_norm=0.01, norm_type=2))
ref = [p.detach().clone().requires_grad_(True) for p in red.params]
ref_opt = torch.optim.AdamW([{"params": [r], "lr": g["lr"], "weight_decay": g["weight_decay"]} for r, g in zip(ref, opt.param_groups)],
                            lr=0.001, betas=(0.9, 0.999), weight_decay=0.005)
for it in range(2):
    eager()
    red.reduce()
    red.wait()
    for r, v in zip(ref, red.views):
        r.grad = v.clone()
    torch.nn.utils.clip_grad_norm_(ref, 0.01, 2)
    ref_opt.step()
    opt.step()
    torch.cuda.synchronize()
    for p, r in zip(red.params, ref):
        assert (p - r).abs().max().item() <= 1e-6 * max(r.abs().max().item(), 1e-3), f"FlatAdamW vs torch.optim.AdamW, iteration {it}"
chk = torch.cat([p.detach().flatten() for p in red.params])
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1]), "parameters differ between the ranks after two iterations"
print(f"RANK {rank} OK worst {worst:.2e} ranks_seen {dist.get_world_size()} hungarian_graph_step OK flat_adamw OK", flush=True)
dist.barrier()
dist.destroy_process_group()
