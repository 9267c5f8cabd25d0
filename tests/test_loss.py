"""Hungarian-matched MaskFormer loss against vectors produced by the reference's own loss code
(oracle/gen_golden_loss.py -> tests/golden/loss_f1.npz): values 1e-5 relative, gradients 1e-6 absolute (fp32 round-off of
differently ordered sums).  Host logic + torch glue: runs on CPU; the GPU variant checks the same vectors on cuda tensors."""
import numpy as np
import pytest
import torch

from spike2former_amd.loss import MaskFormerLoss, seg_to_instances


def _run(g, name, device):
    K = int(g[f"{name}_K"])
    cls = torch.from_numpy(g[f"{name}_cls"]).to(device).requires_grad_(True)
    masks = torch.from_numpy(g[f"{name}_masks"]).to(device).requires_grad_(True)
    seg = torch.from_numpy(g[f"{name}_seg"]).to(device)
    gts = [seg_to_instances(seg[i]) for i in range(seg.shape[0])]
    assert torch.cat([l for l, _ in gts]).cpu().tolist() == g[f"{name}_labels"].tolist()
    out = MaskFormerLoss(K, cls.shape[2]).loss_by_feat(cls, masks, gts)
    sum(out.values()).backward()
    assert list(out.keys()) == g[f"{name}_keys"].tolist()
    got = np.array([float(v) for v in out.values()])
    assert np.allclose(got, g[f"{name}_losses"], rtol=1e-5, atol=1e-7), (got, g[f"{name}_losses"])
    assert np.allclose(cls.grad.cpu().numpy(), g[f"{name}_gcls"], atol=1e-6)
    assert np.allclose(masks.grad.cpu().numpy(), g[f"{name}_gmasks"], atol=1e-6)


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_loss_vs_reference_vectors_cpu(golden, name):
    _run(golden("loss_f1.npz"), name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_loss_vs_reference_vectors_gpu(golden, name):
    _run(golden("loss_f1.npz"), name, "cuda")


def test_matching_is_the_optimal_assignment():
    """The assignment minimises the summed cost: brute force over all injections on a tiny case."""
    import itertools
    g = torch.Generator().manual_seed(3)
    K, Q, L = 3, 4, 2
    crit = MaskFormerLoss(K, Q)
    cls, masks = torch.randn(L, 1, Q, K + 1, generator=g), torch.randn(L, 1, Q, 6, 6, generator=g)
    seg = torch.randint(0, K, (1, 12, 12), generator=g)
    labels, gm = seg_to_instances(seg)
    small = torch.nn.functional.interpolate(gm.unsqueeze(1).float(), (6, 6), mode="nearest").squeeze(1)
    cost = crit.match_costs(cls[:, 0], masks[:, 0], labels, small)
    (pq, pg), = crit.assign(cls, masks, [(labels, gm)])
    n = labels.numel()
    for l in range(L):
        best = min(sum(float(cost[l, q, j]) for j, q in enumerate(perm)) for perm in itertools.permutations(range(Q), n))
        mine = sum(float(cost[l, q, j]) for q, j in zip(pq[l], pg[l]))
        assert abs(mine - best) < 1e-5 and sorted(pg[l].tolist()) == list(range(n)) and list(pq[l]) == sorted(pq[l])


def test_seg_to_instances_drops_ignored_and_handles_empty():
    seg = torch.tensor([[[0, 0, 255], [3, 255, 3]]])
    labels, masks = seg_to_instances(seg)
    assert labels.tolist() == [0, 3] and masks.shape == (2, 2, 3) and masks[1].tolist() == [[False, False, False], [True, False, True]]
    labels, masks = seg_to_instances(torch.full((1, 4, 4), 255))
    assert labels.numel() == 0 and masks.shape == (0, 4, 4)


def test_unsupported_variants_raise():
    with pytest.raises(NotImplementedError):
        MaskFormerLoss(3, 4, loss_cls=dict(use_sigmoid=True))
    with pytest.raises(NotImplementedError):
        MaskFormerLoss(3, 4, train_cfg=dict(assigner=dict(match_costs=[dict(type="mmdet.IoUCost")])))
