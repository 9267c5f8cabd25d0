"""The DCNv3 sampling core alone, forward + backward, against the oracle's torch restatement on the same tensors (GPU vs CPU), over
many seeds at tiny maps: which gradient leaves, by how much, and in which (n, group) slices."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import s2f_oracle as so  # noqa: E402
from spike2former_amd import ops  # noqa: E402


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


worst = {}
for H, W, G, Cg in ((4, 4, 8, 8), (6, 5, 8, 8), (8, 8, 8, 8), (4, 4, 4, 16)):
    for seed in range(40):
        g = torch.Generator().manual_seed(seed)
        N, C, K = 2, G * Cg, 9
        x = torch.randn(N, H, W, C, generator=g) * (10.0 ** torch.randint(-2, 3, (1,), generator=g).item())
        off = torch.randn(N, H, W, G * K * 2, generator=g) * 1.5
        msk = (torch.randint(0, 9, (N, H, W, G * K), generator=g).float() / 8) * (torch.rand(N, H, W, G * K, generator=g) > 0.3)
        gy = torch.randn(N, H, W, C, generator=g) * (10.0 ** torch.randint(-3, 3, (1,), generator=g).item())
        # a heavy tail on the gradient: a few entries 1e4 x larger (what a spiking network's backward looks like)
        gy = gy * (1 + 1e4 * (torch.rand(gy.shape, generator=g) > 0.999))
        a = [t.clone().requires_grad_(True) for t in (x, off, msk)]
        yo = so.dcnv3_core(a[0], a[1], a[2], G, Cg, 3, 1, 1, 1, 1.0)
        yo.backward(gy)
        b = [t.clone().cuda().requires_grad_(True) for t in (x, off, msk)]
        y = ops.dcnv3_core(b[0], b[1], b[2], 3, 3, 1, 1, 1, 1, 1, 1, G, Cg, 1.0)
        y.backward(gy.cuda())
        r = (rel(y.detach().cpu(), yo.detach()), rel(b[0].grad.cpu(), a[0].grad), rel(b[1].grad.cpu(), a[1].grad), rel(b[2].grad.cpu(), a[2].grad))
        key = (H, W, G, Cg)
        if key not in worst or r[1] > worst[key][1][1]:
            worst[key] = (seed, r)
        if r[1] > 1e-3:
            d = (b[0].grad.cpu() - a[0].grad).abs().view(N, H * W, G, Cg).amax((1, 3)) / a[0].grad.abs().max()
            print(f"{key} seed {seed}: y {r[0]:.1e} gin {r[1]:.1e} goff {r[2]:.1e} gmask {r[3]:.1e}; per (n, group) gin error:", [f"{v:.0e}" for v in d.flatten().tolist()])
for k, v in worst.items():
    print(k, "worst seed", v[0], "y / gin / goff / gmask", " ".join(f"{t:.1e}" for t in v[1]))
