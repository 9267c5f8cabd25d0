"""SURVEY section 8 row f4, "merge RepConv into one 3x3": an eval-mode RepConv + BatchNorm + neuron of block3 (C = 256, T B = 8 maps
of 32 x 32) as the three-launch chain of the product path against ONE fused implicit-3x3 launch on the merged kernel
(reparam.merge_repconv), both replayed from a hipGraph of 20 repetitions.  Prints us per RepConv and the agreement of the spikes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spike2former_amd as s2f  # noqa: E402
from spike2former_amd import ops, reparam  # noqa: E402
from spike2former_amd.backbone_sdtv2 import RepConv  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda")
for C, M, HW in ((256, 256, 32), (256, 768, 32), (256, 256, 64)):
    rep = RepConv(C, M).to(dev).eval()
    outer = torch.nn.BatchNorm2d(M).to(dev).eval()
    lif = s2f.Q_IFNode().to(dev)
    s2f.set_keep_membrane(lif, False)
    with torch.no_grad():
        for bn in (rep.body[1].bn, rep.body[2][2], outer):
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
    x = ops.Spikes((torch.randint(0, 9, (8, C, HW, HW), device=dev).float() / 8).to(torch.bfloat16))
    k, b = reparam.merge_repconv(rep, outer)
    one, zero = torch.ones(M, device=dev), torch.zeros(M, device=dev)
    var = torch.full((M,), 1.0 - 1e-5, device=dev)          # mean 0, var + eps = 1: the epilogue's BatchNorm is the identity + bias

    def chain():
        lif.reset()
        return rep(x, outer_bn=outer, lif=lif)[1]

    def merged():
        return ops.conv3x3_bn_lif_eval(x, k, zero, var, one, b, 1e-5, lif=True, want_pre=False)[1]

    with torch.no_grad():
        ya, yb = chain().data.float(), merged().data.float()
        res = {}
        for name, fn in (("chain (3 launches)", chain), ("merged 3x3 (1 launch)", merged)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / 200
    d = (ya - yb).abs()
    print(f"RepConv {C}->{M} on [8, {C}, {HW}, {HW}]: " + ", ".join(f"{n} {u:.1f} us" for n, u in res.items())
          + f"; spikes differing {float((d > 0).float().mean()):.1e} (max {float(d.max()) * 8:.0f} level)")
