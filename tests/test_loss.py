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


def test_label_maps_outside_the_byte_range_raise():
    """seg_as_u8 must not wrap an out-of-range label into a valid class id (the reference fails loudly in cross-entropy)."""
    crit = MaskFormerLoss(3, 4)
    ok = torch.tensor([[[0, 2, 255], [1, 255, 2]]])
    assert crit.seg_as_u8(ok).tolist() == [[[0, 2, 255], [1, 255, 2]]]
    assert crit.seg_as_u8(torch.tensor([[[0, -100, 2]]]), ignore_index=-100).tolist() == [[[0, 255, 2]]]
    for bad in (torch.tensor([[[0, 256, 1]]]), torch.tensor([[[0, -1, 1]]]), torch.tensor([[[0, 511, 1]]])):
        with pytest.raises(ValueError, match="outside 0..255"):
            crit.seg_as_u8(bad)


# ------------------------------------------------------------------------------------------------ semantic-map path (GPU)
def _run_semantic(g, name):
    K = int(g[f"{name}_K"])
    cls = torch.from_numpy(g[f"{name}_cls"]).cuda().requires_grad_(True)
    masks = torch.from_numpy(g[f"{name}_masks"]).cuda().requires_grad_(True)
    seg = torch.from_numpy(g[f"{name}_seg"]).cuda()
    crit = MaskFormerLoss(K, cls.shape[2])
    assert crit.semantic_ok(masks, seg)
    out = crit.loss_semantic(cls, masks, seg)
    sum(out.values()).backward()
    assert list(out.keys()) == g[f"{name}_keys"].tolist()
    got = np.array([float(v) for v in out.values()])
    assert np.allclose(got, g[f"{name}_losses"], rtol=1e-5, atol=1e-7), (got, g[f"{name}_losses"])
    assert np.allclose(cls.grad.cpu().numpy(), g[f"{name}_gcls"], atol=1e-6)
    assert np.allclose(masks.grad.cpu().numpy(), g[f"{name}_gmasks"], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_semantic_path_vs_reference_vectors(golden, name):
    """loss_semantic (segmented-sum costs, label-map targets, static shapes) against the same vectors of the reference's loss."""
    _run_semantic(golden("loss_f1.npz"), name)


def _regions(B, H, W, K, n, seed, ignore=True):
    g = torch.Generator().manual_seed(seed)
    seg = torch.empty(B, H, W, dtype=torch.int64)
    for b in range(B):
        classes = torch.randperm(K, generator=g)[:n]
        for i, (y0, x0) in enumerate((y, x) for y in range(0, H, H // 4) for x in range(0, W, W // 4)):
            seg[b, y0:y0 + H // 4, x0:x0 + W // 4] = classes[i % n]
        if ignore:
            seg[b, :3, 5:40] = 255
    return seg


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["regions", "noise"])
def test_cost_bins_are_the_three_products(kind):
    """ops.mask_cost_bins + costs_all_classes == match_costs (the reference's three [L*Q, hw] x [hw, n] products) on the columns of
    the classes present, to fp32 round-off; bit-identical from run to run (fixed-point accumulation); same assignment."""
    from spike2former_amd.loss import seg_to_instances as s2i
    L, B, Q, K, h, w = 3, 2, 20, 150, 32, 48
    g = torch.Generator().manual_seed(5)
    cls = torch.randn(L, B, Q, K + 1, generator=g).cuda()
    masks = (torch.randn(L, B, Q, h, w, generator=g) * 3).cuda()
    seg = (_regions(B, 2 * h, 2 * w, K, 7, 11) if kind == "regions" else torch.randint(0, K, (B, 2 * h, 2 * w), generator=g)).cuda()
    crit = MaskFormerLoss(K, Q)
    seg_u8 = crit.seg_as_u8(seg)
    cost, count = crit.costs_all_classes(cls, masks, seg_u8)
    cost2, _ = crit.costs_all_classes(cls, masks, seg_u8)
    assert torch.equal(cost, cost2)
    gts = [s2i(seg[b]) for b in range(B)]
    for b, (labels, gm) in enumerate(gts):
        assert torch.nonzero(count[b, :K]).flatten().tolist() == labels.tolist()
        small = torch.nn.functional.interpolate(gm.unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1)
        want = crit.match_costs(cls[:, b].double(), masks[:, b].double(), labels, small.double())
        got = cost[:, b][:, :, labels]
        assert (got.double() - want).abs().max().item() <= 2e-5 * max(want.abs().max().item(), 1.0)
    tgt, rows, avg = crit.match_tables(cost.cpu().numpy(), count.cpu().numpy())
    ref = crit.assign(cls, masks, gts)
    for b in range(B):
        labels = gts[b][0].cpu().numpy()
        small = torch.nn.functional.interpolate(gts[b][1].unsqueeze(1).float(), (h, w), mode="nearest").squeeze(1)
        c64 = crit.match_costs(cls[:, b].double(), masks[:, b].double(), gts[b][0], small.double()).cpu().numpy()
        for l in range(L):
            pq, pg = ref[b][0][l], ref[b][1][l]
            assert np.nonzero(rows[b].reshape(L, Q)[l] >= 0)[0].tolist() == pq.tolist()
            if kind == "regions":
                assert tgt[l, b, pq].tolist() == labels[pg].tolist()
            else:
                # per-pixel noise: 150 near-identical targets, the optimum is degenerate to fp32 round-off -- both assignments
                # must be optimal for the fp64 costs, and one-to-one
                mine = np.searchsorted(labels, tgt[l, b, pq])
                assert len(set(mine.tolist())) == len(mine)
                assert abs(c64[l, pq, mine].sum() - c64[l, pq, pg].sum()) <= 1e-5 * abs(c64[l, pq, pg].sum())
    assert avg.tolist() == [float(sum(max(len(ref[b][0][l]), 1) for b in range(B))) for l in range(L)]


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(32, 48), (20, 70), (8, 6)])
def test_label_map_mask_loss_is_the_gathered_mask_loss(h, w):
    """ops.mask_loss_seg (targets = label map == class, gradient straight to the low-resolution logits through an LDS tile) against
    ops.mask_loss_sums on gathered binary masks + s2f_upsample2x_bwd: sums 1e-5 relative, gradients 1e-5 of their maximum;
    unmatched rows give zeros."""
    from spike2former_amd import ops
    B, R, K = 2, 9, 12
    g = torch.Generator().manual_seed(h * 100 + w)
    pred = (torch.randn(B, R, h, w, generator=g) * 3).cuda().requires_grad_(True)
    seg = _regions(B, 2 * h, 2 * w, K, 5, 3).cuda()
    crit = MaskFormerLoss(K, R)
    seg_u8 = crit.seg_as_u8(seg)
    row_class = torch.randint(-1, K, (B, R), generator=g).to(torch.int32)
    row_class[0, 0], row_class[1, 3] = -1, int(seg[1, -1, -1])
    rc = row_class.cuda()
    wts = torch.randn(B * R, 4, generator=g).cuda()
    sums = ops.mask_loss_seg(pred, seg_u8, rc.reshape(-1), 0.25, 2.0)
    (sums * wts).sum().backward()
    # the gathered form
    valid = (row_class.reshape(-1) >= 0).nonzero().flatten()
    tgt = torch.stack([(seg[int(i) // R] == int(row_class.reshape(-1)[i])) for i in valid]).to(torch.uint8)
    p2 = pred.detach().reshape(B * R, h, w)[valid.cuda()].clone().requires_grad_(True)
    want = ops.mask_loss_sums(p2, tgt, torch.arange(len(valid), device="cuda"), 0.25, 2.0)
    (want * wts[valid.cuda()]).sum().backward()
    got = sums[valid.cuda()]
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    inval = (row_class.reshape(-1) < 0).nonzero().flatten().cuda()
    assert sums[inval].abs().max().item() == 0.0
    gp = pred.grad.reshape(B * R, h, w)
    assert gp[inval].abs().max().item() == 0.0
    assert (gp[valid.cuda()] - p2.grad).abs().max().item() <= 1e-5 * p2.grad.abs().max().item()


@pytest.mark.gpu
def test_label_map_mask_loss_is_bit_repeatable():
    """Round 5: the chunk partials of a row are stored and added in chunk order (csrc/maskloss.hip) -- the four sums of every row
    are identical from launch to launch, at the C2 size (h = w = 256: 32 chunks per row) where rounds 2-4 used fp32 atomics."""
    from spike2former_amd import ops
    B, R, K, h, w = 2, 40, 150, 256, 256
    g = torch.Generator().manual_seed(3)
    pred = (torch.randn(B, R, h, w, generator=g) * 3).cuda()
    seg = _regions(B, 2 * h, 2 * w, K, 10, 4).cuda()
    seg_u8 = MaskFormerLoss(K, R).seg_as_u8(seg)
    rc = torch.randint(-1, K, (B * R,), generator=g).to(torch.int32).cuda()
    first = ops.mask_loss_seg(pred, seg_u8, rc, 0.25, 2.0).clone()
    for _ in range(5):
        assert torch.equal(ops.mask_loss_seg(pred, seg_u8, rc, 0.25, 2.0), first)
