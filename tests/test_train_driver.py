"""SURVEY section 8 row f2 -- the thin training driver: SegDataPreProcessor against vectors produced by the reference's own
file (oracle/gen_golden_f2.py -> tests/golden/preproc_f2.npz, bit-exact), and the restated mmengine pieces (parse_losses,
custom_keys parameter groups, LinearLR -> PolyLR, clip_grad + AdamW) against their defining formulas."""
import math

import numpy as np
import pytest
import torch

import spike2former_amd as s2f
from spike2former_amd.data_preprocessor import SegDataPreProcessor, SegDataSample
from spike2former_amd.train import LinearThenPoly, OptimWrapper, param_groups, parse_losses

CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], bgr_to_rgb=True, pad_val=0, seg_pad_val=255)


@pytest.mark.parametrize("name,kw,training", [("train", dict(size=(32, 32)), True),
                                               ("test", dict(size=(32, 32), test_cfg=dict(size_divisor=16)), False),
                                               ("plain", dict(size=(32, 32)), False)])
def test_data_preprocessor_vs_reference_vectors(golden, name, kw, training):
    g = golden("preproc_f2.npz")
    imgs = [torch.from_numpy(g[f"{name}_img{i}"]) for i in range(2)]
    ds = [SegDataSample(torch.from_numpy(g[f"{name}_seg{i}"]).clone()) for i in range(2)]
    out = SegDataPreProcessor(**CFG, **kw)(dict(inputs=imgs, data_samples=ds), training)
    assert np.array_equal(out["inputs"].numpy(), g[f"{name}_inputs"])
    for i, d in enumerate(out["data_samples"]):
        assert np.array_equal(d.gt_sem_seg.data.numpy(), g[f"{name}_outseg{i}"])
        for k in ("img_shape", "pad_shape", "padding_size", "img_padding_size"):
            key = f"{name}_meta{i}_{k}"
            if key in g.files:
                assert tuple(int(v) for v in d.metainfo[k]) == tuple(g[key].tolist()), (k, d.metainfo)


def test_parse_losses_sums_the_loss_keys():
    losses = {"loss_cls": torch.tensor([1.0, 3.0]), "d0.loss_mask": [torch.tensor(2.0), torch.tensor([4.0, 6.0])],
              "acc_seg": torch.tensor(0.5)}
    loss, log = parse_losses(losses)
    assert float(loss) == 2.0 + (2.0 + 5.0) and list(log) == ["loss", "loss_cls", "d0.loss_mask", "acc_seg"]
    with pytest.raises(TypeError):
        parse_losses({"loss": 1.0})


def test_param_groups_follow_the_config_custom_keys():
    model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
    custom = {"backbone": dict(lr_mult=0.1, decay_mult=1.0), "query_embed": dict(lr_mult=1.0, decay_mult=0.0),
              "query_feat": dict(lr_mult=1.0, decay_mult=0.0), "level_embed": dict(lr_mult=1.0, decay_mult=0.0),
              "backbone.block4": dict(lr_mult=0.5, decay_mult=0.0)}
    g = {x["name"]: (x["lr"], x["weight_decay"]) for x in param_groups(model, 1e-3, 5e-3, dict(custom_keys=custom))}
    assert len(g) == sum(1 for _ in model.parameters())
    assert g["backbone.downsample1_1.encode_conv.weight"] == (1e-4, 5e-3)
    assert g["backbone.block4.0.mlp.fc1_conv.weight"] == (5e-4, 0.0)          # the longer key wins
    assert g["decode_head.query_embed.weight"] == (1e-3, 0.0) and g["decode_head.level_embed.weight"] == (1e-3, 0.0)
    assert g["decode_head.pixel_decoder.mask_feature.weight"] == (1e-3, 5e-3)


def test_linear_then_poly_schedule():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-3}, {"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 1e-4}])
    s = LinearThenPoly(opt, warmup=1500, total=160000, start_factor=1e-6, eta_min=0.0, power=1.0)
    lrs = {}
    for t in range(0, 1502):
        lrs[t] = [g["lr"] for g in opt.param_groups]
        s.step()
    assert math.isclose(lrs[0][0], 1e-9, rel_tol=1e-9) and math.isclose(lrs[0][1], 1e-10, rel_tol=1e-9)
    assert all(lrs[t + 1][0] > lrs[t][0] for t in range(0, 1499)) and math.isclose(lrs[1499][0], 1e-3, rel_tol=1e-9)
    assert math.isclose(lrs[1500][0], 1e-3, rel_tol=1e-12) and lrs[1501][0] < 1e-3
    assert math.isclose(s.factor(1500 + (160000 - 1500) // 2), 0.5, rel_tol=1e-4) and s.factor(160000) == 0.0


def test_optim_wrapper_is_clip_then_adamw():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ref = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ref.load_state_dict(net.state_dict())
    x = torch.randn(5, 4)
    ow = OptimWrapper(net, optimizer=dict(type="AdamW", lr=1e-2, betas=(0.9, 0.999), weight_decay=0.1),
                      clip_grad=dict(max_norm=0.01, norm_type=2), paramwise_cfg=dict(custom_keys={"0.": dict(lr_mult=0.1, decay_mult=0.0)}))
    norm = ow.update_params(net(x).square().sum())
    ref(x).square().sum().backward()
    total = torch.sqrt(sum(p.grad.square().sum() for p in ref.parameters()))
    assert math.isclose(float(norm), float(total), rel_tol=1e-6)
    for name, p in ref.named_parameters():
        lr, wd = (1e-3, 0.0) if name.startswith("0.") else (1e-2, 0.1)
        gclip = p.grad * (0.01 / (total + 1e-6))
        m, v = 0.1 * gclip, 0.001 * gclip.square()
        want = p.data * (1 - lr * wd) - lr * (m / 0.1) / ((v / 0.001).sqrt() + 1e-8)
        assert torch.allclose(dict(net.named_parameters())[name].data, want, rtol=1e-5, atol=1e-8), name
    assert all(p.grad is None for p in net.parameters())


@pytest.mark.gpu
def test_train_steps_on_the_tiny_model():
    """Three iterations of the whole driver on the GPU: preprocessing of ragged uint8 images, reset, Hungarian-matched loss,
    backward, clip to 0.01, AdamW with the config's multipliers, schedule -- finite losses, clipped norm reported, every
    trainable parameter that receives a gradient moves, membranes are reset at the start of each iteration."""
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    cfg = s2f.model_cfg("C1_64")
    cfg["data_preprocessor"] = dict(type="SegDataPreProcessor", size=(w["H"], w["W"]), **CFG)
    model = seeded_init(s2f.MODELS.build(cfg)).cuda().train()
    custom = {"backbone": dict(lr_mult=0.1, decay_mult=1.0), "query_embed": dict(lr_mult=1.0, decay_mult=0.0)}
    ow = OptimWrapper(model, optimizer=dict(type="AdamW", lr=1e-3, betas=(0.9, 0.999), weight_decay=5e-3),
                      clip_grad=dict(max_norm=0.01, norm_type=2), paramwise_cfg=dict(custom_keys=custom))
    sched = LinearThenPoly(ow.optimizer, warmup=2, total=10, start_factor=0.1)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    g = torch.Generator().manual_seed(3)
    logs = []
    for it in range(3):
        imgs = [torch.randint(0, 256, (3, w["H"] - 4 * i, w["W"] - 8 * i), generator=g).to(torch.uint8) for i in range(2)]
        ds = [SegDataSample(torch.randint(0, w["K"], (1, *im.shape[-2:]), generator=g)) for im in imgs]
        logs.append(s2f.train_step(model, dict(inputs=imgs, data_samples=ds), ow, sched))
    L = w["dec"][0] + 1
    for log in logs:
        assert len([k for k in log if k.endswith("loss_cls")]) == L and all(math.isfinite(v) for v in log.values())
        assert log["grad_norm"] > 0 and math.isclose(log["loss"], sum(v for k, v in log.items() if "loss_" in k), rel_tol=1e-5)
    moved = [k for k, p in model.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert len(moved) > 0.9 * len(before)
    neurons = [m for m in model.modules() if isinstance(m, s2f.Q_IFNode)]
    assert any(torch.is_tensor(m.v) for m in neurons)          # the iteration left membranes behind ...
    s2f.reset_net(model)
    assert all(isinstance(m.v, float) and m.v == 0.0 for m in neurons)      # ... which the next one starts by clearing
