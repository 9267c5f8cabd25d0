"""E-SpikeFormer backbone on the GPU against vectors produced by the reference's own sdtv3.py
(oracle/gen_golden_sdtv3.py -> tests/golden/sdtv3_tiny.npz).  Tolerances as for the SDT-v2 path (tests/test_gpu_model.py): the exact
per-neuron census decides -- a flip-free step is held to 1e-5 (feature maps) / 1e-4 (gradients); 2e-2 / 5e-2 only explain a flip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(img_size_h=64, img_size_w=64, patch_size=16, in_channels=3, num_classes=20, embed_dim=[16, 32, 64, 72], num_heads=8,
          mlp_ratios=4, qkv_bias=False, depths=8, sr_ratios=1, T=2, decode_mode="QTrick")


def test_stateless_four_level_neuron():
    import spike2former_amd as s2f
    n = s2f.Multispike_norm().cuda()
    x = torch.linspace(-1, 6, 4001, device="cuda", requires_grad=True)
    y = n(x)
    assert torch.equal(y, torch.round(torch.clamp(x.detach(), 0, 4)) / 4)
    y.sum().backward()
    assert torch.equal(x.grad, ((x.detach() >= 0) & (x.detach() <= 4)).float() / 4)
    assert isinstance(n.v, float) and torch.equal(n(x.detach()), y.detach())          # no state between calls


def test_sdtv3_backbone_vs_reference_vectors(golden):
    import spike2former_amd as s2f
    from spike2former_amd.init_utils import seeded_init
    g = golden("sdtv3_tiny.npz")
    model = seeded_init(s2f.MODELS.build(dict(type="Spiking_vit_MetaFormerv2", **KW))).cuda().train()
    # the census rule of tests/test_gpu_model.py: exact per-neuron {sum of spike counts, non-zero counts, elements with 0 <= x <= 4}
    # against the reference's own Multispike_norm modules (written by the generator's hooks); no neuron differs => feature maps to
    # 1e-5, gradients to 1e-4 of the gradient scale.  The loose bounds only explain a step in which a neuron flipped.
    census = {}

    def grab(mod, inp, out, n):
        if n not in census:
            u = inp[0].detach()
            census[n] = (int((out.detach() * mod.D).round().sum().item()), int((out.detach() != 0).sum().item()),
                         int(((u >= 0) & (u <= mod.D)).sum().item()))
    hooks = [m.register_forward_hook(lambda m_, i_, o_, n=n: grab(m_, i_, o_, n)) for n, m in model.named_modules()
             if isinstance(m, s2f.Multispike_norm)]
    outs = model(torch.from_numpy(g["img"]).cuda())
    for h in hooks:
        h.remove()
    sum((o * o).mean() for o in outs).backward()            # a plain mean of BatchNorm outputs has no gradient
    want = {str(n): tuple(int(v) for v in c) for n, c in zip(g["census_names"], g["census"])}
    assert set(census) == set(want), set(census) ^ set(want)
    flipped = sorted(n for n in want if census[n] != want[n])
    tol_out, tol_grad = (2e-2, 5e-2) if flipped else (1e-5, 1e-4)
    assert [tuple(o.shape) for o in outs] == [tuple(g[f"x{i + 1}"].shape) for i in range(4)]
    for i, o in enumerate(outs):
        ref = torch.from_numpy(g[f"x{i + 1}"])
        assert (o.detach().cpu() - ref).abs().max().item() <= tol_out * ref.abs().max().item(), (i, flipped)
        assert all(torch.equal(o[0], o[t]) for t in range(1, o.shape[0]))              # reset step: T identical slices
    params = dict(model.named_parameters())
    keys = [k[6:] for k in g.files if k.startswith("grad__")]
    gscale = float(g["grad_absmax"].max())
    for k in keys:
        ref = torch.from_numpy(g["grad__" + k])
        err = (params[k].grad.cpu() - ref).abs().max().item()
        assert err <= tol_grad * (ref.abs().max().item() + 5e-3 * gscale), (k, err, ref.abs().max().item(), flipped)
    rm = dict(model.named_buffers())["block3.0.attn.q_conv.1.running_mean"].cpu()
    assert torch.allclose(rm, torch.from_numpy(g["running_mean__block3.0.attn.q_conv.1"]), rtol=1e-3, atol=1e-5)


def test_sdtv3_attention_is_the_reference_product_order():
    """(q k^T) v * 2 scale with 4x wider value heads == four d-wide q (k^T v) problems, bit for bit on spike operands."""
    from spike2former_amd import ops
    g = torch.Generator().manual_seed(1)
    TB, h, d, N, r = 3, 8, 9, 36, 4
    C = h * d
    q, k = (torch.randint(0, 5, (TB, C, N), generator=g).float() / 4 for _ in range(2))
    v = torch.randint(0, 5, (TB, r * C, N), generator=g).float() / 4
    qh = q.view(TB, h, d, N).transpose(2, 3)
    kh = k.view(TB, h, d, N).transpose(2, 3)
    vh = v.view(TB, h, r * d, N).transpose(2, 3)
    want = ((qh @ kh.transpose(-2, -1)) @ vh * (d ** -0.5 * 2)).transpose(2, 3).reshape(TB, r * C, N)
    vj = v.cuda().view(TB, h, r, d, N).permute(2, 0, 1, 3, 4).contiguous()
    o = torch.stack([ops.sdsa(q.cuda(), k.cuda(), vj[j].reshape(TB, C, N), h, d ** -0.5 * 2) for j in range(r)], 0)
    o = o.view(r, TB, h, d, N).permute(1, 2, 0, 3, 4).reshape(TB, r * C, N)
    assert torch.allclose(o.cpu(), want, rtol=1e-6, atol=1e-6)


def test_c5_tiny_step_with_the_panoptic_shaped_head():
    """BASELINE configs[4] in its shrunken-width form: E-SpikeFormer backbone + MaskFormer head with things + stuff classes on a
    non-square input; one fwd+bwd step with the Hungarian-matched loss -- shapes, finite loss dictionary, gradients reach the
    backbone's first convolution."""
    import spike2former_amd as s2f
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C5_tiny"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C5_tiny"))).cuda().train()
    img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(2)).cuda()
    seg = torch.randint(0, w["K"], (w["B"], 1, w["H"], w["W"]), generator=torch.Generator().manual_seed(3)).cuda()
    s2f.reset_net(model)
    cls, masks = model(img)
    assert cls.shape == (w["dec"][0] + 1, w["B"], w["Q"], w["K"] + 1) and masks.shape[-2:] == (w["H"] // 2, w["W"] // 2)
    s2f.reset_net(model); model.zero_grad(set_to_none=True)
    losses = model(img, [seg[i] for i in range(w["B"])], mode="loss")
    total = sum(losses.values())
    assert torch.isfinite(total)
    total.backward()
    g = model.backbone.downsample1_1.encode_conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0


def test_c5_size_attention_and_unaligned_batchnorm_at_their_own_size():
    """BASELINE configs[4] at its own size (800x1344 -> 50x84 = 4 200 tokens at stride 16, D = 4 neurons, 256 channels):
    (1) the attention core of MS_Attention_linear -- the reference's O(N^2) product order (q k^T) v with 4x wider value heads
    (sdtv3.py:295-307) evaluated in fp64 per (tb, head) -- against the four d-wide q (k^T v) problems of ops.sdsa: spike operands
    make every sum exact in fp32 (asserted: below 2^24 ulps), so the two orders give the SAME bits at N = 4 200;
    (2) the row-walking BatchNorm + neuron kernels in their ALIGNED = false form (L = 4 200: a 256-element tile straddles two
    channel rows) against ATen's batch_norm + the oracle's neuron: pre-activations to 1e-5, spikes equal except one-level flips in
    <= 1e-4 of the elements, input / affine gradients to 1e-4 of their scale."""
    from oracle import s2f_oracle as so
    from spike2former_amd import ops
    g = torch.Generator().manual_seed(2)
    TB, h, d, r, N = 2, 8, 32, 4, 4200
    C = h * d
    q, k = (torch.randint(0, 5, (TB, C, N), generator=g).float() / 4 for _ in range(2))
    v = torch.randint(0, 5, (TB, r * C, N), generator=g).float() / 4
    scale = d ** -0.5 * 2
    vj = v.cuda().view(TB, h, r, d, N).permute(2, 0, 1, 3, 4).contiguous()
    got = torch.stack([ops.sdsa(q.cuda(), k.cuda(), vj[j].reshape(TB, C, N), h, scale) for j in range(r)], 0)
    got = got.view(r, TB, h, d, N).permute(1, 2, 0, 3, 4).reshape(TB, r * C, N)
    qh = q.cuda().double().view(TB, h, d, N).transpose(2, 3)
    kh = k.cuda().double().view(TB, h, d, N).transpose(2, 3)
    vh = v.cuda().double().view(TB, h, r * d, N).transpose(2, 3)
    want = torch.empty(TB, h, r * d, N, dtype=torch.float64, device="cuda")
    for tb in range(TB):
        for hh in range(h):
            want[tb, hh] = ((qh[tb, hh] @ kh[tb, hh].T) @ vh[tb, hh]).T          # the reference's order, exact in fp64
    assert float(want.abs().max()) * 16 < 2 ** 24                                   # integer multiples of 1/16 below 2^24: exact in fp32
    # the kernel multiplies the exact fp32 sum by the fp32 scale, rounding once
    assert torch.equal(got, (want.float() * torch.tensor(scale, dtype=torch.float32, device="cuda")).reshape(TB, r * C, N))
    # (2) unaligned row-walking BatchNorm + D = 4 neuron, forward and backward
    Nn, Cc, L = 4, 64, 4200
    z = (torch.randn(Nn, Cc, L, generator=g) * 1.5 + 0.7)
    res = torch.randn(Nn, Cc, L, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.2
    gyo, guo = torch.randn(Nn, Cc, L, generator=g), torch.randn(Nn, Cc, L, generator=g)
    zc, rc, gc, bc = (t.clone().requires_grad_(True) for t in (z, res, gamma, beta))
    uo = torch.nn.functional.batch_norm(zc, None, None, gc, bc, True, 0.1, 1e-5) + rc
    yo, _, _ = so.lif_step(uo, None, 4)
    (yo * gyo).sum().backward(retain_graph=True); (uo * guo).sum().backward()
    zg, rg, gg, bg = (t.clone().cuda().requires_grad_(True) for t in (z, res, gamma, beta))
    u, y, _ = ops.bn_act(zg, None, gg, bg, None, None, None, True, 0.1, 1e-5, residual=rg, lif=True, want_pre=True, D=4)
    ((y.float() * gyo.cuda()).sum() + (u * guo.cuda()).sum()).backward()
    assert (u.detach().cpu() - uo.detach()).abs().max().item() <= 1e-5 * uo.detach().abs().max().item()
    dlev = ((y.float().detach().cpu() - yo.detach()) * 4).round()
    assert dlev.abs().max().item() <= 1 and (dlev != 0).float().mean().item() <= 1e-4
    for mine, ref in ((zg.grad, zc.grad), (rg.grad, rc.grad), (gg.grad, gc.grad), (bg.grad, bc.grad)):
        off = ((mine.cpu() - ref).abs() > 1e-4 * ref.abs().max()).float().mean().item()
        assert off <= 1e-3, off          # a one-level / mask-bit difference moves single elements of gz, nothing systematic
