"""GPU, BASELINE.json full sizes (C2: ADE20K-150, 512x512, T=4, full widths), against the oracle run on the host cores.

Why stage-wise ("teacher-forced") and not end-to-end at this size: the reference algorithm itself is chaotic in its
decoder at C2 with random weights -- train-mode BatchNorm over only 100 distinct query rows, no softmax, 6 x 3 residual
sub-layers.  Measured with the oracle alone (CPU, fp32): adding 1e-6 * N(0,1) noise to the input image changes the
decoder neurons' firing rates by up to 1.8e-2 and the final logits by O(1) relative; backbone / pixel-decoder rates move
by up to 4e-3 (the attention neurons, whose input sums over all 1 024 tokens).  So a round-off-level difference (GPU GEMM/BN vs ATen-CPU) cannot be bounded at the logits; it is bounded
where the map is well conditioned: every stage is fed the ORACLE's inputs and compared at its own outputs, and the firing
table of the free-running model is compared with a tolerance per region (1e-2 before the decoder, 5e-2 inside it: ~2.5x
the oracle's own sensitivity).  The oracle is pinned bit-exactly to the reference at the tiny config (tests/golden)."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# Every GPU test runs under ops.STRICT (tests/conftest.py `_fallback_census`): no shape of any configuration -- BASELINE's C2-C5 or the
# shrunken C1_64 of the graph-mechanics tests -- may leave the package's kernels for a vendor library or an ATen convolution.


@pytest.fixture(scope="module")
def c2():
    import spike2former_amd as s2f
    from oracle import s2f_oracle as so
    cfg = dataclasses.replace(so.CONFIGS["C2"], B=1)
    st0 = so.make_params(cfg, requires_grad=False)
    model = s2f.MODELS.build(s2f.model_cfg("C2"))
    model.load_state_dict(st0, strict=True)
    model = model.cuda().train()
    img = so.synthetic_image(cfg, seed=7)
    # oracle: one train-mode forward, keeping the stage boundaries (its BN running stats are updated in a private copy)
    st = {k: v.clone() for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    net.stages = {}
    pd_taps = {}                                          # spike maps of the pixel decoder's encoder-layer neurons (uint8 counts)
    bb_taps = {}                                          # ... and of every backbone neuron (242 M elements as uint8)
    with torch.no_grad():
        net.tap = lambda name, y: bb_taps.__setitem__(name, (y * 8).round().to(torch.uint8))
        feats = net.backbone(img)
        fire_bb = dict(net.firing)
        net.tap = lambda name, y: pd_taps.__setitem__(name, (y * 8).round().to(torch.uint8)) if ".encoder.layers." in name else None
        mf, memory, msm = net.pixel_decoder(feats)
        net.tap = None
    ref = dict(feats=feats, mask_features=mf, memory=memory, msm=msm, firing=dict(net.firing), fire_bb=fire_bb,
               stages=net.stages, pd_taps=pd_taps, bb_taps=bb_taps, net=net)
    return s2f, so, cfg, st0, model, img, ref


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


def frac_off(a, b, tol):
    """fraction of elements whose difference exceeds tol * max|b|"""
    return ((a - b).abs() > tol * b.abs().max()).float().mean().item()


class NeuronWalk:
    """Records the spike maps of every Q_IFNode under `root` in EXECUTION order (nn.Module forward hooks: fused neurons report
    through them as well) and compares them with the oracle's taps where the comparison is well conditioned: every neuron
    up to and including the first one that differs at all must differ in <= `frac` of its elements, each by exactly ONE level
    (an input within fp32 round-off of k + 0.5 rounded to the other side).  Downstream of a flip the maps legitimately diverge."""

    def __init__(self, s2f, root, prefix):
        self.order, self.mine, self.prefix = [], {}, prefix
        self.hooks = [m.register_forward_hook(lambda mod, inp, o, n=n: self._grab(n, o))
                      for n, m in root.named_modules() if isinstance(m, s2f.Q_IFNode)]

    def _grab(self, n, o):
        if n not in self.mine:                            # a neuron called twice keeps its first map
            self.order.append(n)
            self.mine[n] = (o.detach() * 8).round().to(torch.uint8).cpu()

    def close(self):
        for h in self.hooks:
            h.remove()

    def counts_for(self, full_name, like):
        """this build's spike counts of neuron `full_name` (uint8) in the oracle's layout `like` -- what OracleNet.force takes"""
        n = full_name[len(self.prefix) + 1:] if full_name != self.prefix else ""
        return self.mine[n].reshape(like.shape)

    def check(self, taps, frac=1e-4, channel_major=False, forced=()):
        """-> (neurons compared, name of the first differing neuron or None); `forced`: full names of neurons whose oracle output was
        replaced by this build's map (OracleNet.force) -- equal by construction, the walk continues behind them"""
        checked = 0
        for n in self.order:
            full = f"{self.prefix}.{n}" if n else self.prefix
            if full in forced:
                continue
            r = taps.get(full)
            if r is None or r.numel() != self.mine[n].numel():
                continue
            m = self.mine[n]
            if channel_major and tuple(m.shape[-2:]) != tuple(r.shape[-2:]) and tuple(m.shape[-2:]) == tuple(r.shape[-2:])[::-1]:
                m = m.reshape(-1, *m.shape[-2:]).transpose(1, 2)          # product [T*B, C, L] vs oracle [T, B, L, C]
            d = m.reshape(-1).to(torch.int16) - r.reshape(-1).to(torch.int16)
            checked += 1
            if (d != 0).any():
                assert d.abs().max().item() == 1 and (d != 0).float().mean().item() <= frac, (self.prefix, n, (d != 0).float().mean().item())
                return checked, full
        return checked, None


@pytest.mark.timeout(900)
def test_c2_backbone_stages_teacher_forced(c2):
    """Every backbone stage at 512x512, T=4 (5 down-samplings, 4 MS_ConvBlocks, 6 + 2 spike-driven attention blocks), each
    fed the ORACLE's input for that stage.  Most stages agree to ~2e-7 relative L2 (no spike flips at all); where a few of
    the stage's inner neurons flip by one level (measured: 3e-5 of the elements, inputs within 1e-6 of k + 0.5), each flip
    perturbs ~4 600 downstream pre-activations (3x3 conv, 512 channels) by ~w/8 and so flips ~18 further neurons: the
    stage output then differs by 1e-2..4e-2 in relative L2.  Bound: 6e-2, and at least half of the stages below 1e-5.
    Where the comparison is well conditioned it is exact: the stage's neurons are walked in execution order (NeuronWalk) --
    identical spike maps up to the first neuron that differs, which differs in <= 1e-4 of its elements by exactly one level;
    a stage without any flip must reproduce the oracle's output to 1e-5."""
    s2f, so, cfg, st0, model, img, ref = c2
    model.load_state_dict(st0, strict=True)
    bb = model.backbone
    worst, walked, reseeded = {}, {}, {}
    for name, (x, y) in ref["stages"].items():
        if not name.startswith("backbone."):
            continue
        mod = bb
        for part in name[len("backbone."):].split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        s2f.reset_net(model)
        walk = NeuronWalk(s2f, mod, name)
        with torch.no_grad():
            out = mod(x.cuda())
        walk.close()
        worst[name] = rel_l2(out.cpu(), y)
        # the stage's neurons in execution order: identical to the oracle's up to the first one-level flip
        checked, first = walk.check(ref["bb_taps"])
        walked[name] = (checked, first)
        assert first is not None or checked >= (0 if name.endswith("downsample1_1") else 1 if "downsample" in name else 3), (name, walk.order)
        if first is None:                                 # no neuron flipped anywhere in the stage: round-off only
            assert worst[name] <= 1e-5, (name, worst[name])
            continue
        # RE-SEED behind every flipped neuron (round 5): the oracle re-runs THIS stage emitting this build's spike map for the neuron
        # that flipped (OracleNet.force), so the neurons downstream see identical inputs again and are walked exactly as well; repeated
        # until no unforced neuron differs.  Every neuron of the stage is then either bit-identical to the oracle's or one of the few
        # that flipped <= 1e-4 of their elements by one level, and the stage output must agree to round-off.
        net, forced = ref["net"], {}
        for _ in range(12):
            forced[first] = walk.counts_for(first, ref["bb_taps"][first])
            for k, v in st0.items():                      # the stage's BatchNorm statistics as before its first oracle run
                if k.startswith(name + ".") and ("running_" in k or "num_batches" in k):
                    net.p[k].copy_(v)
            taps = {}
            net.reset()
            net.force, net.tap = forced, lambda nm, yy: taps.__setitem__(nm, (yy * 8).round().to(torch.uint8))
            with torch.no_grad():
                y2 = net.run_backbone_stage(name, x)
            net.force = net.tap = None
            _, first = walk.check(taps, forced=forced)
            if first is None:
                break
        assert first is None, (name, sorted(forced))
        reseeded[name] = (len(forced), rel_l2(out.cpu(), y2))
        assert reseeded[name][1] <= 1e-5, (name, reseeded[name])
    assert len(worst) == 17
    assert max(worst.values()) <= 6e-2, worst
    assert sorted(worst.values())[len(worst) // 2] <= 1e-5, worst
    print("re-seeded stages (flipped neurons, rel L2 after re-seeding):", reseeded)


@pytest.mark.timeout(900)
def test_c2_backbone_free_running_firing(c2):
    """Free-running backbone: all 76 neuron firing rates within 1e-2 of the oracle's (its own sensitivity to 1e-6 input
    noise is 4e-3 here)."""
    s2f, so, cfg, st0, model, img, ref = c2
    model.load_state_dict(st0, strict=True)
    s2f.reset_net(model)
    with torch.no_grad(), s2f.FiringRecorder(model.backbone) as rec:
        model.backbone(img.cuda())
        rec.collect()
    table = rec.result()["t0"]
    assert len(table) == 76
    for k, v in table.items():
        assert abs(v - ref["fire_bb"]["backbone." + k]) <= 1e-2, k


@pytest.mark.timeout(900)
def test_c2_pixel_decoder_teacher_forced(c2):
    """The 6 DCN encoder layers (SepConv_Spike + DCNv3 + MS_MLP, 32x32x256, G=32) each fed the oracle's input.

    A layer whose neurons all see inputs away from a rounding boundary reproduces the oracle to ~2e-7 relative L2 (5 of 6
    layers, every run so far).  When one neuron input sits within fp32 round-off of k + 0.5, the two implementations may
    round it to different levels; measured: 16 of 1 048 576 DCN-output spikes flipped by one level turn into 2.8 % different
    FFN input spikes and a layer output 1.2e-1 apart in relative L2 -- and WHICH borderline element flips changes with any
    reordering of an fp32 sum (e.g. the split-K of the forward GEMM).  So the layer is checked where the comparison is well
    conditioned: walking its neurons in execution order, every neuron up to and including the first one that differs at all
    must differ in <= 1e-4 of its elements, each by exactly ONE level; the layer output is bounded by 2.5e-1, and at least 4
    of the 6 layers must be exact to 1e-5.
    The whole pixel decoder fed the oracle's backbone features: firing rates within 1e-2."""
    s2f, so, cfg, st0, model, img, ref = c2
    model.load_state_dict(st0, strict=True)
    pd = model.decode_head.pixel_decoder
    worst = {}
    for i in range(cfg.pd_layers):
        name = f"decode_head.pixel_decoder.encoder.layers.{i}"
        x, y = ref["stages"][name]
        layer = pd.encoder.layers[i]
        s2f.reset_net(model)
        order, mine = [], {}
        def grab(mod, inp, o, n):
            order.append(n)
            mine[n] = (o.detach() * 8).round().to(torch.uint8).cpu()      # returns None: the output is left alone
        hooks = [m.register_forward_hook(lambda mod, inp, o, n=n: grab(mod, inp, o, n))
                 for n, m in layer.named_modules() if isinstance(m, s2f.Q_IFNode)]
        with torch.no_grad():
            out = layer(x.cuda())
        for h in hooks:
            h.remove()
        worst[i] = rel_l2(out.cpu(), y)
        checked = 0
        for n in order:                                   # execution order
            r = ref["pd_taps"].get(f"{name}.{n}")
            if r is None or r.numel() != mine[n].numel():
                continue
            d = mine[n].reshape(-1).to(torch.int16) - r.reshape(-1).to(torch.int16)
            checked += 1
            if (d != 0).any():
                assert d.abs().max().item() == 1 and (d != 0).float().mean().item() <= 1e-4, (i, n)
                break                                     # downstream of a flip the maps legitimately diverge
        assert checked >= 3, (i, order)
    assert max(worst.values()) <= 2.5e-1, worst
    assert sorted(worst.values())[1] <= 1e-5, worst
    model.load_state_dict(st0, strict=True)
    s2f.reset_net(model)
    with torch.no_grad(), s2f.FiringRecorder(pd) as rec:
        mf, memory, msm = pd([f.cuda() for f in ref["feats"]], None)
        rec.collect()
    assert mf.shape == ref["mask_features"].shape and memory.shape == ref["memory"].shape
    for k, v in rec.result()["t0"].items():
        assert abs(v - ref["firing"]["decode_head.pixel_decoder." + k]) <= 1e-2, k


@pytest.mark.timeout(900)
def test_c2_decoder_layers_teacher_forced(c2):
    """Each of the 6 decoder layers (cross-attention over 1 024 / 4 096 / 16 384 keys, self-attention, FFN) fed the oracle's
    query and memory: output within 2e-2 of its max for >= 99 % of the elements (a flipped spike of the 100-row query moves a
    whole BatchNorm'd column, hence the looser bound than in the backbone) -- and exact where that is well defined: the
    layer's neurons walked in execution order agree with the oracle's spike maps up to the first one-level flip; a layer
    without any flip reproduces the oracle's output to 1e-5."""
    s2f, so, cfg, st0, model, img, ref = c2
    model.load_state_dict(st0, strict=True)
    hd = model.decode_head
    st = {k: v.clone() for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    p = st
    h = "decode_head."
    t, bs = cfg.T, cfg.B
    query = p[h + "query_feat.weight"].unsqueeze(0).repeat(t, bs, 1, 1)
    qpos = p[h + "query_embed.weight"].unsqueeze(0).repeat(bs, 1, 1)
    for i in range(cfg.dec_layers):
        lv = i % 3
        m = ref["msm"][lv]
        key = m.flatten(3).permute(0, 1, 3, 2) + p[h + "level_embed.weight"][lv].view(1, 1, -1)
        kpos = so.sine_pos_embed(bs, m.shape[-2], m.shape[-1], cfg.num_feats).flatten(2).permute(0, 2, 1)
        lname = h + f"transformer_decoder.layers.{i}"
        taps = {}
        with torch.no_grad():
            net.reset()
            net.tap = lambda name, y: taps.__setitem__(name, (y * 8).round().to(torch.uint8))
            out_ref = net._dec_layer(lname, query, key, qpos, kpos)
            net.tap = None
            s2f.reset_net(model)
            walk = NeuronWalk(s2f, hd.transformer_decoder.layers[i], lname)
            out = hd.transformer_decoder.layers[i](query=query.cuda(), key=key.cuda(), value=key.cuda(),
                                                   query_pos=qpos.cuda(), key_pos=kpos.cuda())
            walk.close()
        assert frac_off(out.cpu(), out_ref, 2e-2) <= 1e-2 and rel_l2(out.cpu(), out_ref) <= 2e-2, i
        # the layer's 15 neurons in execution order (CA: q/k/v in, q/k/v out, attn; SA: the same; FFN: 2): identical to the
        # oracle's up to the first one-level flip (<= 1e-3 of a map here: a BatchNorm over only 100 distinct query rows
        # turns one borderline element into a shifted column)
        checked, first = walk.check(taps, frac=1e-3, channel_major=True)
        assert first is not None or checked >= 10, (i, checked, walk.order)
        if first is None:
            assert rel_l2(out.cpu(), out_ref) <= 1e-5, (i, rel_l2(out.cpu(), out_ref))
        query = out_ref                                   # teacher forcing: the next layer starts from the oracle's output


@pytest.mark.timeout(900)
def test_c2_free_running_firing_table(c2):
    """cal_firing_num's table for the free-running model at full size: 270 rows in the reference's order; rates within 1e-2
    before the decoder and within 5e-2 inside it (the oracle's own sensitivity to 1e-6 input noise: 4e-3 / 1.8e-2)."""
    s2f, so, cfg, st0, model, img, ref = c2
    st = {k: v.clone() for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    with torch.no_grad():
        net.head(ref["feats"])
    model.load_state_dict(st0, strict=True)
    s2f.reset_net(model)
    with torch.no_grad(), s2f.FiringRecorder(model) as rec:
        cls, masks = model(img.cuda())
        rec.collect()
    table = rec.result()["t0"]
    want = dict(ref["fire_bb"]); want.update(net.firing)
    assert len(table) == 270 and set(table) == set(want)
    assert cls.shape == (7, 1, 100, 151) and masks.shape == (7, 1, 100, 256, 256)
    for k, v in table.items():
        tol = 5e-2 if ("transformer_decoder" in k or k.startswith("decode_head.decoder_out") or "mask_embed" in k
                       or "shortcut" in k) else 1e-2
        assert abs(v - want[k]) <= tol, (k, v, want[k])


def test_c2_step_properties(c2):
    """Size-independent properties of the full-size fwd+bwd step: keep_membrane on/off identical logits, every activation
    handed to the bf16 spike GEMM sits on the spike grid, finite non-zero gradients, hipGraph replay runs the same step."""
    s2f, so, cfg, st0, model, img, ref = c2
    from spike2former_amd import ops
    from spike2former_amd.graph import GraphedStep
    x = img.cuda()

    def step(keep):
        model.load_state_dict(st0, strict=True)
        s2f.set_keep_membrane(model, keep)
        s2f.reset_net(model)
        model.zero_grad(set_to_none=True)
        cls, masks = model(x)
        s2f.headline_loss(cls, masks).backward()
        g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
        return cls.detach().clone(), masks.detach().clone(), g

    ops.SPIKE_GEMM_CHECK = True
    try:
        a = step(True)
    finally:
        ops.SPIKE_GEMM_CHECK = False
    b = step(False)
    # keep_membrane on / off run the SAME arithmetic (the membrane is only written, never read, after a reset).  The one
    # non-deterministic ingredient of a forward pass is the order of the fp64 atomics that sum the BatchNorm statistics of
    # the large maps (bn_stats_kernel): a last-bit difference in a mean can flip a borderline neuron, and the chaotic
    # decoder (module docstring) then amplifies it.  So: exact equality is asserted where no atomic reduction precedes --
    # test_keep_membrane_is_output_identical_up_to_the_first_atomic_reduction below -- and here only that both runs are
    # finite and of the same scale.
    assert torch.isfinite(a[0]).all() and torch.isfinite(b[0]).all() and torch.isfinite(a[1]).all() and torch.isfinite(b[1]).all()
    assert 0.5 <= a[1].abs().mean().item() / b[1].abs().mean().item() <= 2.0
    assert torch.isfinite(a[2]).all() and a[2].abs().max().item() > 0
    s2f.set_keep_membrane(model, False)
    model.load_state_dict(st0, strict=True)
    gs = GraphedStep(model, s2f.headline_loss, x, warmup=1)
    model.load_state_dict(st0, strict=True)
    loss_g = float(gs())
    assert np.isfinite(loss_g)
    s2f.set_keep_membrane(model, True)


def test_keep_membrane_is_output_identical_up_to_the_first_atomic_reduction(c2):
    """`Q_IFNode.keep_membrane = False` (bench.py: the membrane write is skipped because a reset precedes every step) must not
    change a single output bit.  At full size two runs of the SAME configuration already differ in the last bit of a
    BatchNorm mean (fp64 atomics over the workgroups of bn_stats_kernel, large maps only), so the identity is asserted (a)
    where no atomic reduction is involved: every backbone stage on the 32x32 maps (downsample4, the six block3 and two block4
    attention blocks) -- their BatchNorms run in the single-pass kernels (one workgroup per channel, fixed summation order),
    their GEMMs and the fused attention kernel reduce through LDS in a fixed order -- keep on vs off from identical inputs
    (the oracle's): bitwise equal outputs."""
    s2f, so, cfg, st0, model, img, ref = c2
    model.load_state_dict(st0, strict=True)
    bb = model.backbone
    stages = {n: xy for n, xy in ref["stages"].items() if n.startswith("backbone.block") or n.startswith("backbone.downsample4")}
    assert len(stages) >= 9
    for name, (x, _) in stages.items():
        mod = bb
        for part in name[len("backbone."):].split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        outs = []
        for keep in (True, False):
            model.load_state_dict(st0, strict=True)
            s2f.set_keep_membrane(model, keep)
            s2f.reset_net(model)
            with torch.no_grad():
                outs.append(mod(x.cuda()).clone())
        assert torch.equal(outs[0], outs[1]), name
    s2f.set_keep_membrane(model, True)


@pytest.mark.timeout(900)
def test_c3_size_decoder_cross_attention_vs_oracle():
    """BASELINE configs[2] (Cityscapes 1024x512): the decoder's cross-attention at ITS key counts -- 2 048 / 8 192 / 32 768
    keys (levels 32x64, 64x128, 128x256) against 100 queries, 256 channels, 8 heads, T = 4, B = 1 -- one decoder layer per
    level, teacher-forced against the oracle with name-seeded weights: neurons walked in execution order (identical up to
    the first one-level flip), output within 2e-2."""
    import spike2former_amd as s2f
    from oracle import s2f_oracle as so
    cfg = dataclasses.replace(so.CONFIGS["C3"], B=1)
    st0 = so.make_params(cfg, requires_grad=False)
    model = s2f.MODELS.build(s2f.model_cfg("C3"))
    model.load_state_dict(st0, strict=True)
    model = model.cuda().train()
    hd = model.decode_head
    net = so.OracleNet({k: v.clone() for k, v in st0.items()}, cfg, training=True)
    h = "decode_head."
    t, bs, C = cfg.T, 1, cfg.feat_channels
    g = torch.Generator().manual_seed(31)
    query = st0[h + "query_feat.weight"].unsqueeze(0).repeat(t, bs, 1, 1)
    qpos = st0[h + "query_embed.weight"].unsqueeze(0).repeat(bs, 1, 1)
    for i, (hh, ww) in enumerate([(32, 64), (64, 128), (128, 256)]):
        mem = torch.randn(1, bs, hh * ww, C, generator=g).repeat(t, 1, 1, 1) * 1.5          # T identical slices, as a reset step has
        key = mem + st0[h + "level_embed.weight"][i].view(1, 1, -1)
        kpos = so.sine_pos_embed(bs, hh, ww, cfg.num_feats).flatten(2).permute(0, 2, 1)
        lname = h + f"transformer_decoder.layers.{i}"
        taps = {}
        with torch.no_grad():
            net.reset()
            net.tap = lambda name, y: taps.__setitem__(name, (y * 8).round().to(torch.uint8))
            out_ref = net._dec_layer(lname, query, key, qpos, kpos)
            net.tap = None
            s2f.reset_net(model)
            walk = NeuronWalk(s2f, hd.transformer_decoder.layers[i], lname)
            out = hd.transformer_decoder.layers[i](query=query.cuda(), key=key.cuda(), value=key.cuda(),
                                                   query_pos=qpos.cuda(), key_pos=kpos.cuda())
            walk.close()
        checked, first = walk.check(taps, frac=1e-3, channel_major=True)
        assert first is not None or checked >= 10, (i, checked)
        assert frac_off(out.cpu(), out_ref, 2e-2) <= 1e-2 and rel_l2(out.cpu(), out_ref) <= 2e-2, (i, hh * ww)
        if first is None:
            assert rel_l2(out.cpu(), out_ref) <= 1e-5
    del model
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_c5_full_size_step_properties():
    """BASELINE configs[4] at FULL size (COCO-panoptic-shaped 800x1344, T = 4, E-SpikeFormer backbone + 133-class head, per-GPU
    batch 1) through size-independent properties: output shapes, every activation handed to the bf16 spike GEMM on the spike
    grid (D = 4 backbone, D = 8 head), finite non-zero gradients, the T-fold structure of a reset step, backbone firing rate
    inside (0, 1)."""
    import spike2former_amd as s2f
    from spike2former_amd import ops
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C5"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C5"))).cuda().train()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(9)).cuda()
    seen = {}
    hk = model.backbone.register_forward_hook(lambda m, i, o: seen.__setitem__("x4", o[-1].detach()))
    s2f.reset_net(model); model.zero_grad(set_to_none=True)
    ops.SPIKE_GEMM_CHECK = True
    try:
        cls, masks = model(img)
    finally:
        ops.SPIKE_GEMM_CHECK = False
        hk.remove()
    assert cls.shape == (7, w["B"], w["Q"], w["K"] + 1) and masks.shape == (7, w["B"], w["Q"], w["H"] // 2, w["W"] // 2)
    s2f.headline_loss(cls, masks).backward()
    flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    assert torch.isfinite(cls).all() and torch.isfinite(masks).all() and torch.isfinite(flat).all() and flat.abs().max() > 0
    x4 = seen["x4"]
    assert x4.shape[0] == w["T"] and all(torch.equal(x4[0], x4[t]) for t in range(1, w["T"]))
    y, _ = ops.lif(x4, None, D=4, keep_v=False)
    c = y * 4
    assert torch.equal(c, torch.round(c)) and 0 <= float(c.min()) and float(c.max()) <= 4 and 0 < float(y.mean()) < 1
    del model, flat
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload", ["C3", "C4"])
def test_other_baseline_configs_step_properties(workload):
    """BASELINE.json configs[2] (Cityscapes 1024x512, per-GPU shard of 2) and configs[3] (T = 8) at full size, through
    properties that need no oracle run: output shapes, every activation handed to the bf16 spike GEMM on the spike grid,
    finite non-zero gradients for the same parameter set as at C2, the T-fold structure of the workload (a reset step repeats the input T
    times, so every time slice of a feature map is identical) and a firing rate inside (0, 1) on the last backbone tap."""
    import spike2former_amd as s2f
    from spike2former_amd import ops
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS[workload]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg(workload))).cuda().train()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(w["B"], 3, w["H"], w["W"], generator=torch.Generator().manual_seed(9)).cuda()
    seen = {}
    h = model.backbone.register_forward_hook(lambda m, i, o: seen.__setitem__("x4", o[-1].detach()))
    s2f.reset_net(model); model.zero_grad(set_to_none=True)
    ops.SPIKE_GEMM_CHECK = True
    try:
        cls, masks = model(img)
    finally:
        ops.SPIKE_GEMM_CHECK = False
        h.remove()
    assert cls.shape == (7, w["B"], w["Q"], w["K"] + 1) and masks.shape == (7, w["B"], w["Q"], w["H"] // 2, w["W"] // 2)
    s2f.headline_loss(cls, masks).backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch.cat([g.flatten() for g in grads])
    assert torch.isfinite(cls).all() and torch.isfinite(masks).all() and torch.isfinite(flat).all() and flat.abs().max() > 0
    x4 = seen["x4"]                                          # last backbone tap [T, B, C, H/16, W/16]
    assert x4.shape == (w["T"], w["B"], w["embed_dim"][3], w["H"] // 16, w["W"] // 16)
    assert all(torch.equal(x4[0], x4[t]) for t in range(1, w["T"]))
    y, _ = ops.lif(x4, None, keep_v=False)
    c = y * 8
    assert torch.equal(c, torch.round(c)) and 0 <= float(c.min()) and float(c.max()) <= 8 and 0 < float(y.mean()) < 1
    del model, grads, flat
    torch.cuda.empty_cache()


def test_tiny_graph_replay_is_bitwise_the_eager_step():
    """On the well-conditioned tiny config the hipGraph replay must reproduce the eager step bit for bit."""
    import spike2former_amd as s2f
    from spike2former_amd.graph import GraphedStep
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()

    def eager():
        model.load_state_dict(sd); s2f.reset_net(model); model.zero_grad(set_to_none=True)
        loss = s2f.headline_loss(*model(img)); loss.backward()
        return float(loss.detach()), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    l0, g0 = eager()
    gs = GraphedStep(model, s2f.headline_loss, img, warmup=1)
    model.load_state_dict(sd)
    l1 = float(gs())
    g1 = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    assert l0 == l1
    assert (g0 - g1).abs().max().item() <= 1e-4 * g0.abs().max().item()       # LDS/atomic accumulation order in dW


def test_graph_replay_follows_weight_updates():
    """A captured step multiplies by the LIVE weights: the bf16 hi / mid / lo terms of every weight are re-split inside the
    graph (ops.resplit_all, one launch), so a replay after an optimiser-style in-place update equals the eager step on the
    updated weights -- not the step of capture time (the stale-split hazard of a cached weight split)."""
    import spike2former_amd as s2f
    from spike2former_amd import ops
    from spike2former_amd.graph import GraphedStep
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}

    def eager():
        s2f.reset_net(model); model.zero_grad(set_to_none=True)
        loss = s2f.headline_loss(*model(img)); loss.backward()
        return float(loss.detach()), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    gs = GraphedStep(model, s2f.headline_loss, img, warmup=1)
    assert ops.resplit_all(img.device) > 50                  # every spike-GEMM weight of the model is a registered job
    model.load_state_dict(sd0)                               # capture / warm-up steps moved the BatchNorm running statistics
    l_before = float(gs())
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():                                    # an "optimiser step": every parameter changed in place
        for p in model.parameters():
            p.add_(0.05 * p.abs().mean() * torch.randn(p.shape, generator=g).to(p.device))
    sd1 = {k: v.clone() for k, v in model.state_dict().items()}
    l_replay = float(gs())
    g_replay = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    model.load_state_dict(sd1)
    l_eager, g_eager = eager()
    assert l_replay != l_before
    assert l_replay == l_eager
    assert (g_replay - g_eager).abs().max().item() <= 1e-4 * g_eager.abs().max().item()
    # replay -> optimiser step -> EAGER forward -> replay: the eager step sees stale versions of every cached split / pack and
    # re-converts.  It must convert INTO the buffers the graph has baked in (ops._cache_buffer) and may not touch the job tables
    # the graph recorded: a fresh allocation would leave the replay writing bf16 terms into freed, possibly reused memory.
    ptrs = {k: v[1].data_ptr() for k, v in ops._SPLIT_CACHE.items()}
    tables = (ops._SPLIT_TABLE["jobs"], ops._SPLIT_TABLE["pack_jobs"])
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.97)
    sd2 = {k: v.clone() for k, v in model.state_dict().items()}
    l_e2, g_e2 = eager()                                      # eager step on the new weights (re-converts every weight)
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]      # churn the allocator: freed blocks would be reused here
    del junk
    assert all(ops._SPLIT_CACHE[k][1].data_ptr() == ptr for k, ptr in ptrs.items() if k in ops._SPLIT_CACHE)
    assert ops._SPLIT_TABLE["jobs"] is tables[0] and ops._SPLIT_TABLE["pack_jobs"] is tables[1]
    model.load_state_dict(sd2)
    l_r2 = float(gs())
    g_r2 = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    assert l_r2 == l_e2 and (g_r2 - g_e2).abs().max().item() <= 1e-4 * g_e2.abs().max().item()


def test_overlap_step_is_the_single_graph_step():
    """graph.GraphedOverlapStep (forward graph | backward graph, all-reduce of step k under the forward of step k + 1; here
    without a process group, so the collectives are no-ops) leaves in the flat gradient buffer what graph.GraphedStep leaves,
    step after step: same loss bit for bit, gradients to the round-off of the split-K atomics."""
    import spike2former_amd as s2f
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.graph import GraphedOverlapStep, GraphedStep
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()
    red = FlatGradAllReduce(model.parameters(), 1)
    red.install_sinks()
    try:
        single = GraphedStep(model, s2f.headline_loss, img, grad_buffer=red, warmup=1)
        model.load_state_dict(sd)
        l0 = float(single()); g0 = red.flat.clone()
        over = GraphedOverlapStep(model, s2f.headline_loss, img, red, warmup=1, buckets=2)
        for _ in range(2):
            model.load_state_dict(sd)
            l1 = float(over()); over.finish()
            torch.cuda.synchronize()
            assert l1 == l0
            assert (red.flat - g0).abs().max().item() <= 1e-4 * g0.abs().max().item()
    finally:
        red.close()


def test_tiny_split_graph_step_with_the_hungarian_loss_is_the_eager_step():
    """graph.GraphedSplitStep (forward graph | eager Hungarian-matched loss | backward graph) against the eager
    `mode="loss"` step on the tiny config: same loss values bit for bit, same gradients (flat buffer) to the run-to-run
    noise of the split-K atomics."""
    import spike2former_amd as s2f
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.graph import GraphedSplitStep
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5)).cuda()
    seg = _region_maps(2, w["H"], w["W"], w["K"], 6, 6).cuda()
    gts = [s2f.seg_to_instances(seg[i]) for i in range(2)]
    red = FlatGradAllReduce(model.parameters(), 1)

    model.load_state_dict(sd); s2f.reset_net(model); red.zero()
    losses = model(img, [seg[i] for i in range(2)], mode="loss")     # the semantic-map path (loss.MaskFormerLoss.loss_semantic)
    sum(losses.values()).backward()
    red.gather()
    want_loss = {k: float(v) for k, v in losses.items()}
    want = red.flat.clone()
    del losses                                            # no autograd graph of the eager step may outlive this point
    for p in model.parameters():
        p.grad = None
    import gc
    gc.collect()

    model.load_state_dict(sd)
    step = GraphedSplitStep(model, img, red, warmup=1)
    for _ in range(2):                                    # replayed twice: the second replay must not depend on the first
        model.load_state_dict(sd)
        leaves = step.forward()
        got = model.decode_head.loss_by_feat(leaves[0], leaves[1], gts)
        sum(got.values()).backward()
        step.backward(leaves)
        torch.cuda.synchronize()
        # eager step: label-map kernels; here: the generic instance-mask path -- the same numbers to fp32 round-off
        assert list(got.keys()) == list(want_loss.keys())
        assert all(abs(float(v) - want_loss[k]) <= 1e-5 * max(abs(want_loss[k]), 1e-3) for k, v in got.items())
        scale = want.abs().max().item()
        assert (red.flat - want).abs().max().item() <= 1e-4 * scale


def _region_maps(B, H, W, K, n, seed):
    """[B, 1, H, W] semantic maps of n classes each, as a 4 x 4 grid of rectangles (+ a strip of the ignored label)."""
    g = torch.Generator().manual_seed(seed)
    seg = torch.empty(B, 1, H, W, dtype=torch.int64)
    for b in range(B):
        classes = torch.randperm(K, generator=g)[:n]
        for i, (y0, x0) in enumerate((y, x) for y in range(0, H, H // 4) for x in range(0, W, W // 4)):
            seg[b, 0, y0:y0 + H // 4, x0:x0 + W // 4] = classes[i % n]
        seg[b, 0, :2, 3:17] = 255
    return seg


def test_hungarian_graph_step_is_the_eager_loss_step():
    """graph.GraphedHungarianStep (forward + costs graph | host assignment | losses + backward graph) against the eager
    `mode="loss"` step: same loss dictionary (to the round-off of the float atomics in the mask-loss sums), same gradients in the
    flat buffer; a replay on NEW inputs (image and semantic map) follows them."""
    import spike2former_amd as s2f
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.graph import GraphedHungarianStep
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    s2f.set_keep_membrane(model, False)
    imgs = [torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(2)]
    segs = [_region_maps(2, w["H"], w["W"], w["K"], 5 + i, 6 + i).cuda() for i in range(2)]
    red = FlatGradAllReduce(model.parameters(), 1)
    want = []
    for img, seg in zip(imgs, segs):
        model.load_state_dict(sd); s2f.reset_net(model); red.zero()
        losses = model(img, [seg[i] for i in range(2)], mode="loss")
        sum(losses.values()).backward()
        s2f.ops.wgrad_join()
        red.gather()
        want.append(({k: float(v) for k, v in losses.items()}, red.flat.clone()))
        del losses
    for p in model.parameters():
        p.grad = None
    import gc
    gc.collect()
    model.load_state_dict(sd)
    step = GraphedHungarianStep(model, imgs[0], segs[0], red, warmup=1)
    for i in (0, 1, 0):
        model.load_state_dict(sd)
        got = step(imgs[i], segs[i])
        torch.cuda.synchronize()
        wl, wg = want[i]
        assert list(got.keys()) == list(wl.keys())
        assert all(abs(float(v) - wl[k]) <= 1e-6 * max(abs(wl[k]), 1e-3) for k, v in got.items()), i
        assert (red.flat - wg).abs().max().item() <= 1e-4 * wg.abs().max().item(), i


_BB_STAGES = [("downsample1_1", "_down", (7, 2, 3, True)), ("ConvBlock1_1.0", "_convblock", ()), ("downsample1_2", "_down", (3, 2, 1, False)),
              ("ConvBlock1_2.0", "_convblock", ()), ("downsample2", "_down", (3, 2, 1, False)), ("ConvBlock2_1.0", "_convblock", ()),
              ("ConvBlock2_2.0", "_convblock", ()), ("downsample3", "_down", (3, 2, 1, False))] + \
             [(f"block3.{i}", "_block", ()) for i in range(6)] + [("downsample4", "_down", (3, 1, 1, False))] + \
             [(f"block4.{i}", "_block", ()) for i in range(2)]


class SteBits:
    """What explains a gradient gap between two correct fp32 implementations of a stage: the neurons whose SPIKE differs by a level
    (an input within round-off of k + 0.5) and the neurons whose straight-through MASK bit 1[0 <= h <= D] differs although the
    spike does not (h within round-off of 0 or D).  The oracle's side is collected while it runs (OracleNet.tap / tap_in); this
    build's side by a second, hooked forward of the stage (nn.Module forward hooks see (input, fp32 spikes) of every Q_IFNode,
    fused ones included; that module-call path agrees with the bench path to fp32 round-off, not bit for bit, so it can meet a
    borderline element the bench path does not and vice versa -- the caller combines it with the bench path's own output
    comparison).  Neurons are matched by name and walked in execution order up to and including the first one whose spikes
    differ; a map whose two layouts differ (token- vs channel-major) is transposed when that makes the shapes agree, otherwise it
    counts as not comparable."""

    def __init__(self, net, D=8, state=None):
        self.D, self.masks, self.spikes, self.state = D, {}, {}, state
        net.tap = lambda n, y: self.spikes.__setitem__(n, (y.detach() * D).round().to(torch.uint8))
        net.tap_in = lambda n, h: self.masks.__setitem__(n, (h >= 0) & (h <= D))
        self.net = net

    def detach(self):
        self.net.tap = self.net.tap_in = None

    def count(self, s2f, model, mod, prefix, run):
        """run(): one forward of `mod` on this build -> (neurons compared, not comparable, elements, differing spikes, differing mask
        bits where the spike agrees)"""
        mine = {}

        def grab(m, inp, out, n):
            if n not in mine:
                u = inp[0].detach()
                mine[n] = (((u >= 0) & (u <= self.D)).cpu(), (out.detach() * self.D).round().to(torch.uint8).cpu())
        hooks = [m.register_forward_hook(lambda m_, i_, o_, n=n: grab(m_, i_, o_, n)) for n, m in mod.named_modules()
                 if isinstance(m, s2f.Q_IFNode)]
        if self.state is not None:
            # the first forward updated the BatchNorm running statistics, which BNAndPadLayer's border value reads even in training
            model.load_state_dict(self.state, strict=True)
        s2f.reset_net(model)
        with torch.no_grad():
            run()
        for hk in hooks:
            hk.remove()
        compared = skipped = elems = dspk = dmask = 0
        for n, (mk, spk) in mine.items():          # execution order (dict order = first call)
            key = f"{prefix}.{n}" if n else prefix
            if key not in self.spikes or self.spikes[key].numel() != spk.numel():
                skipped += 1
                continue
            rs, rm = self.spikes[key], self.masks[key]
            if tuple(spk.shape[-2:]) != tuple(rs.shape[-2:]) and tuple(spk.shape[-2:]) == tuple(rs.shape[-2:])[::-1]:
                spk, mk = spk.reshape(-1, *spk.shape[-2:]).transpose(1, 2), mk.reshape(-1, *mk.shape[-2:]).transpose(1, 2)
            ds = spk.reshape(-1) != rs.reshape(-1)
            if ds.float().mean().item() > 1e-2:          # different layouts of the same map (or a stage downstream of a flip)
                skipped += 1
                continue
            compared += 1
            elems += spk.numel()
            dspk += int(ds.sum())
            dmask += int(((mk.reshape(-1) != rm.reshape(-1)) & ~ds).sum())
            if ds.any():          # downstream of a spike flip the maps legitimately diverge: the walk ends at the first one
                break
        return compared, skipped, elems, dspk, dmask


def _stage_backward(s2f, so, cfg, st0, model, mod, name, fn, args, x, seed):
    """One stage, forward AND backward, on both sides from the oracle's input x and a seeded output gradient:
    -> (flipped?, gx gap, worst parameter-gradient gap, its name, SteBits.count), gaps relative to the gradient scale."""
    # oracle (CPU autograd on a private copy of the parameters of this stage)
    pref = name + "."
    st = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k.startswith(pref)) if k.startswith(pref) else v)
          for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    bits = SteBits(net, state=st0)
    xo = x.clone().requires_grad_(True)
    yo = getattr(net, fn)(name, xo, *args)
    bits.detach()
    gy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed)) / yo.numel() ** 0.5
    yo.backward(gy)
    # this build: the bench path (bf16 spike maps, fused kernels, no hooks)
    model.load_state_dict(st0, strict=True)
    for p in mod.parameters():
        p.grad = None
    s2f.reset_net(model)
    xg = x.cuda().requires_grad_(True)
    out = mod(xg)
    out.backward(gy.cuda())
    s2f.ops.wgrad_join()
    flipped = rel_l2(out.detach().cpu(), yo.detach()) > 1e-5
    gx_gap = (rel_l2(xg.grad.cpu(), xo.grad), frac_off(xg.grad.cpu(), xo.grad, 1e-4))
    grads = {n: p.grad for n, p in mod.named_parameters() if p.grad is not None}
    ref = {n: st[pref + n].grad for n in grads if st[pref + n].grad is not None}
    gscale = max(v.abs().max().item() for v in ref.values())
    gaps = {n: (grads[n].cpu() - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-3 * gscale) for n in ref}
    assert len(ref) >= 2, (name, list(grads))
    xc = x.cuda()
    counted = bits.count(s2f, model, mod, name, lambda: mod(xc))
    return flipped, gx_gap, max(gaps.values()), max(gaps, key=gaps.get), counted


@pytest.mark.timeout(1800)
def test_c2_stage_gradients_teacher_forced(c2):
    """Full-size BACKWARD parity: every C2 backbone stage (512x512, T = 4) and every pixel-decoder encoder layer is fed the
    oracle's stage input and a seeded output gradient; the input gradient and every parameter gradient of the stage are
    compared with the oracle's autograd.  These are the kernel variants only the full size uses: row-walking BatchNorm backward
    on 256x256 maps, the grouped weight gradients, the implicit 3x3 input gradients, the 6-pass input-gradient GEMM on the
    packed weights, the DCN backward at 32x32xG32.  The straight-through mask 1[0 <= h <= D] has its own round-off boundaries
    (h within fp32 round-off of 0 or D: the spike is the same, the mask bit is not); ONE such bit at an inner neuron changes the
    input gradient over that neuron's whole receptive field (3x3 x 7x7 x all input channels ~ 14 000 elements in a ConvBlock;
    measured on ConvBlock1_1: 23 000 of 67 M elements off, relative L2 6e-4; in an attention block it reaches every token of
    the head).  Measured (one run, 23 stages): the five down-samplings, block3.0, block3.4 and four of the six pixel-decoder
    layers meet no such boundary and agree to 2e-7 .. 1.4e-6 in relative L2 of the input gradient and <= 5e-4 of the gradient
    scale in every parameter gradient -- the kernels themselves are exact to round-off; the stages that do meet one range up to
    3.9e-2 / 1.6e-1 (block4.0, which also has a forward spike flip).
    Round 4: the explanation is CHECKED, not assumed.  Every neuron of the stage is compared with the oracle's in its spike and in
    its mask bit (SteBits: a second, hooked forward, walked in execution order up to the first spike flip).  Asserted: (1) a stage
    whose output agrees with the oracle's and in which no spike and no mask bit differs agrees to round-off -- 1e-5 (input gradient,
    relative L2) and 1e-3 (parameter gradients); (2) the differing bits are few: <= 1e-4 of the neuron elements walked (measured:
    0, 4 or 8 bits -- the T = 4 replicas of one or two elements -- of 10^7..10^8); (3) a stage that does meet such a bit stays within
    5e-2 / 2e-1; at least 8 stages exact."""
    s2f, so, cfg, st0, model, img, ref = c2
    bb, pd = model.backbone, model.decode_head.pixel_decoder
    rows = []
    for short, fn, args in _BB_STAGES:
        name = "backbone." + short
        mod = bb
        for part in short.split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        rows.append((name,) + _stage_backward(s2f, so, cfg, st0, model, mod, name, fn, args, ref["stages"][name][0], 100 + len(rows)))
    for i in range(cfg.pd_layers):
        name = f"decode_head.pixel_decoder.encoder.layers.{i}"
        rows.append((name,) + _stage_backward(s2f, so, cfg, st0, model, pd.encoder.layers[i], name, "_enc_layer", (),
                                              ref["stages"][name][0], 200 + i))
    model.load_state_dict(st0, strict=True)
    print("stage-gradient gaps (stage, output differs, gx rel-L2, gx fraction off, worst parameter gap, (neurons compared, not "
          "comparable, elements, differing spikes, differing mask bits)):",
          [(r[0].split(".", 1)[1], r[1], f"{r[2][0]:.1e}", f"{r[2][1]:.1e}", f"{r[3]:.1e}", r[5]) for r in rows])
    exact = 0
    for name, flipped, (gx_l2, gx_off), p_gap, worst_p, (compared, skipped, elems, dspk, dmask) in rows:
        assert gx_l2 <= 5e-2 and p_gap <= 2e-1, (name, flipped, gx_l2, gx_off, p_gap, worst_p)
        assert compared >= (0 if name.endswith("downsample1_1") else 1), (name, compared, skipped)
        assert dspk + dmask <= max(1e-4 * elems, 4), (name, elems, dspk, dmask)
        if not flipped and skipped == 0 and dspk == 0 and dmask == 0:          # nothing that could explain a gap: there must be none
            assert gx_l2 <= 1e-5 and p_gap <= 1e-3, (name, gx_l2, p_gap, worst_p)
        exact += (gx_l2 <= 1e-5 and p_gap <= 1e-3)
    assert len(rows) == 17 + cfg.pd_layers and exact >= 8, [(r[0], r[2][0], r[3]) for r in rows]


def _grad_gaps(grads, ref):
    """-> (worst relative gap of a parameter gradient, its name); gaps relative to max|ref| + 1e-3 of the overall gradient scale"""
    gscale = max(v.abs().max().item() for v in ref.values())
    gaps = {n: (grads[n].cpu() - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-3 * gscale) for n in ref}
    worst = max(gaps, key=gaps.get)
    return gaps[worst], worst


@pytest.mark.timeout(1800)
def test_c2_decoder_layer_gradients_teacher_forced(c2):
    """Full-size BACKWARD parity of the six transformer-decoder layers on the bench path (channel-major query stream, key / value
    neurons fused with the level / position adds, s2f_sdsa_bwd_bf16 with the straight-through mask in its loaders, the generic
    BatchNorm backward on 100-token rows, the decoder's grouped weight gradients): each layer is fed the oracle's query, the
    oracle's memory level (1 024 / 4 096 / 16 384 keys) and a seeded output gradient; the gradients of the query, of the memory
    map (key + value paths summed), of the level embedding and of every parameter of the layer are compared with the oracle's
    autograd (detr_layers.py:491-559, mmcv_spike/transformer.py:196-361, 776-784).  A BatchNorm over 100 distinct query rows
    turns one borderline spike into a shifted column, and one straight-through mask bit at a key neuron changes the memory gradient
    of that token; so the layer's 15 neurons are compared with the oracle's in spike AND mask bit (SteBits): a layer in which none
    differs must agree to round-off, the others are bounded loosely, and the differing bits must be few."""
    s2f, so, cfg, st0, model, img, ref = c2
    hd = model.decode_head
    h = "decode_head."
    t, bs = cfg.T, cfg.B
    query = st0[h + "query_feat.weight"].unsqueeze(0).repeat(t, bs, 1, 1)
    qpos = st0[h + "query_embed.weight"].unsqueeze(0).repeat(bs, 1, 1)
    rows = []
    s2f.set_keep_membrane(model, False)                # as bench.py: a reset precedes every step (the fused key / value neurons need it)
    for i in range(cfg.dec_layers):
        lv = i % 3
        lname = h + f"transformer_decoder.layers.{i}"
        pref = lname + "."
        lev = h + "level_embed.weight"
        st = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k and (k.startswith(pref) or k == lev))
                  if (k.startswith(pref) or k == lev) else v) for k, v in st0.items()}
        net = so.OracleNet(st, cfg, training=True)
        bits = SteBits(net, state=st0)
        qo = query.clone().requires_grad_(True)
        mo = ref["msm"][lv].clone().requires_grad_(True)
        key = mo.flatten(3).permute(0, 1, 3, 2) + st[lev][lv].view(1, 1, -1)
        kpos = so.sine_pos_embed(bs, mo.shape[-2], mo.shape[-1], cfg.num_feats).flatten(2).permute(0, 2, 1)
        yo = net._dec_layer(lname, qo, key, qpos, kpos)
        bits.detach()
        gy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(300 + i)) / yo.numel() ** 0.5
        yo.backward(gy)
        # this build, as the head drives a layer (MaskFormerHead.decoder_inputs / run_decoder)
        model.load_state_dict(st0, strict=True)
        layer = hd.transformer_decoder.layers[i]
        for p in list(layer.parameters()) + [hd.level_embed.weight]:
            p.grad = None
        s2f.reset_net(model)
        msm = [m.cuda() for m in ref["msm"]]
        msm[lv].requires_grad_(True)
        qg = query.cuda().requires_grad_(True)
        def run(q):
            dec_in, dec_key, kv = hd.decoder_inputs(msm, bs)
            return kv[lv], layer.forward_stream(q.transpose(2, 3).contiguous(), qpos.cuda().transpose(1, 2).contiguous(), key=dec_key[lv],
                                                value=dec_in[lv], kv_spikes=kv[lv], kv_projected=None, last=True)[0]
        fused_kv, out = run(qg)
        assert fused_kv is not None                    # the fused key / value neurons (s2f_sum2_lif_fwd / _bwd)
        out.backward(gy.cuda())
        s2f.ops.wgrad_join()
        flipped = rel_l2(out.detach().cpu(), yo.detach()) > 1e-5
        grads = {pref + n: p.grad for n, p in layer.named_parameters() if p.grad is not None}
        grads[lev] = hd.level_embed.weight.grad
        refg = {n: st[n].grad for n in grads if st[n].grad is not None}
        assert len(refg) >= 20, (i, len(refg), len(grads))
        p_gap, worst = _grad_gaps(grads, refg)
        counted = bits.count(s2f, model, layer, lname, lambda: run(qg.detach()))
        rows.append((i, flipped, rel_l2(qg.grad.cpu(), qo.grad), rel_l2(msm[lv].grad.cpu(), mo.grad), p_gap, worst, counted))
        query = yo.detach()                            # teacher forcing: the next layer starts from the oracle's output
    s2f.set_keep_membrane(model, True)
    model.load_state_dict(st0, strict=True)
    print("decoder-layer gradient gaps (layer, output differs, gq, gmemory, worst parameter, (neurons compared, not comparable, "
          "elements, differing spikes, differing mask bits)):",
          [(r[0], r[1], f"{r[2]:.1e}", f"{r[3]:.1e}", f"{r[4]:.1e}", r[5].split("layers.")[-1], r[6]) for r in rows])
    for i, flipped, gq, gm, p_gap, worst, (compared, skipped, elems, dspk, dmask) in rows:
        assert gq <= 5e-2 and gm <= 5e-2 and p_gap <= 2e-1, (i, flipped, gq, gm, p_gap, worst)
        assert compared >= 1 and dspk + dmask <= max(1e-4 * elems, 4), (i, compared, skipped, elems, dspk, dmask)
        if not flipped and skipped == 0 and dspk == 0 and dmask == 0:          # nothing that could explain a gap: there must be none
            assert gq <= 1e-4 and gm <= 1e-4 and p_gap <= 2e-3, (i, gq, gm, p_gap, worst)


@pytest.mark.timeout(1800)
def test_c2_sdme_and_folded_mask_contraction_gradients(c2):
    """Full-size BACKWARD parity of the head's tail (dense_heads/maskformer_head.py:568-586): sigmoid -> neurons -> cls_embed,
    the mask-embedding MLP, the query-mixing shortcut Conv1d + BatchNorm1d(100), mask_embed_spike, and the mask contraction with the
    pixel decoder's mask_feature 1x1 convolution FOLDED into it (ops.mask_einsum_folded: the T-mean inside the contraction, dE on
    the spike operand, dS = W^T G).  Inputs: the oracle's seven decoder states and a seeded spike map in place of
    mask_feature_spike's output (both sides apply the same neuron to the same counts); outputs weighted by seeded gradients.
    Compared: the gradients of the decoder states, of the spike map's input, and of every parameter of the tail including the
    folded convolution's weight and bias."""
    s2f, so, cfg, st0, model, img, ref = c2
    hd = model.decode_head
    h = "decode_head."
    t, bs = cfg.T, cfg.B
    names = [k for k in st0 if k.startswith(h) and any(k.startswith(h + n) for n in
             ("cls_embed.", "mask_embed.", "shortcut_conv.", "w", "pixel_decoder.mask_feature.")) and "running" not in k
             and "num_batches" not in k and not k.startswith(h + "pixel_decoder.mask_feature_")]
    st = {k: (v.clone().requires_grad_(k in names) if k in names else v.clone()) for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    # the seven decoder states: the oracle's own forward through the six layers from the oracle's memory levels
    query = st0[h + "query_feat.weight"].unsqueeze(0).repeat(t, bs, 1, 1)
    qpos = st0[h + "query_embed.weight"].unsqueeze(0).repeat(bs, 1, 1)
    outs = [query]
    with torch.no_grad():
        for i in range(cfg.dec_layers):
            m = ref["msm"][i % 3]
            key = m.flatten(3).permute(0, 1, 3, 2) + st0[h + "level_embed.weight"][i % 3].view(1, 1, -1)
            kpos = so.sine_pos_embed(bs, m.shape[-2], m.shape[-1], cfg.num_feats).flatten(2).permute(0, 2, 1)
            query = net._dec_layer(h + f"transformer_decoder.layers.{i}", query, key, qpos, kpos)
            outs.append(query)
    O = torch.stack(outs)
    C, Hm, Wm = cfg.feat_channels, cfg.H // 2, cfg.W // 2
    g = torch.Generator().manual_seed(77)
    counts = (torch.randn(t, bs, C, Hm, Wm, generator=g) * 1.5 + 0.5).round().clamp_(0, 8)        # the neuron's input: its own counts
    Oo, xo = O.clone().requires_grad_(True), counts.clone().requires_grad_(True)
    net.reset()
    so_spk = net.lif(h + "pixel_decoder.mask_feature_spike", xo)
    mf = net.conv2d(h + "pixel_decoder.mask_feature", so_spk.flatten(0, 1))
    cls_o, masks_o = net._sdme(Oo, mf.reshape(t, bs, *mf.shape[1:]))
    g_cls = torch.randn(cls_o.shape, generator=g) / cls_o.numel() ** 0.5
    g_masks = torch.randn(masks_o.shape, generator=g) / masks_o.numel() ** 0.5
    ((cls_o * g_cls).sum() + (masks_o * g_masks).sum()).backward()
    # this build
    model.load_state_dict(st0, strict=True)
    for p in hd.parameters():
        p.grad = None
    s2f.set_keep_membrane(model, False)
    s2f.reset_net(model)
    Og, xg = O.cuda().requires_grad_(True), counts.cuda().requires_grad_(True)
    spk = hd.pixel_decoder.mask_feature_spike.fire(xg)
    assert isinstance(spk, s2f.ops.Spikes) and spk.tok is not None          # the bf16 pair the folded contraction takes
    cls, masks = hd.sdme(Og, spk)
    ((cls * g_cls.cuda()).sum() + (masks * g_masks.cuda()).sum()).backward()
    s2f.ops.wgrad_join()
    flipped = rel_l2(masks.detach().cpu(), masks_o.detach()) > 1e-5 or rel_l2(cls.detach().cpu(), cls_o.detach()) > 1e-5
    params = dict(model.named_parameters())
    grads = {n: params[n].grad for n in names if params[n].grad is not None}
    refg = {n: st[n].grad for n in grads if st[n].grad is not None}
    assert set(refg) == set(names), sorted(set(names) - set(refg))
    p_gap, worst = _grad_gaps(grads, refg)
    gO, gx = rel_l2(Og.grad.cpu(), Oo.grad), rel_l2(xg.grad.cpu(), xo.grad)
    print("head-tail gradient gaps (flipped, g_states, g_spike_input, worst parameter):", flipped, f"{gO:.1e}", f"{gx:.1e}", f"{p_gap:.1e}", worst)
    assert rel_l2(masks.detach().cpu(), masks_o.detach()) <= 2e-2 and rel_l2(cls.detach().cpu(), cls_o.detach()) <= 2e-2
    assert gO <= 5e-2 and gx <= 5e-2 and p_gap <= 2e-1, (flipped, gO, gx, p_gap, worst)
    if not flipped:
        assert gO <= 1e-4 and gx <= 1e-4 and p_gap <= 2e-3, (gO, gx, p_gap, worst)
    s2f.set_keep_membrane(model, True)
    model.load_state_dict(st0, strict=True)


@pytest.mark.timeout(900)
def test_c2_batchnorm_statistics_come_from_the_gemm_epilogues(c2):
    """Round 4: in a C2 training step every BatchNorm whose input is produced by one of the packed-weight GEMM / implicit 3x3 kernels
    takes its statistics from that kernel's epilogue partials (ops.BN_PARTIALS; s2f_bn_partials_finalize) -- the statistics pass
    s2f_bn_stats is left only for the depthwise-convolution outputs of the large FPN levels -- and the step's outputs agree with the
    statistics-pass form to the fp64 summation order (spikes may flip at borderline elements: compared per stage elsewhere; here the
    first backbone stage, which has no neuron, must agree to 1e-6)."""
    s2f, so, cfg, st0, model, img, ref = c2
    ops = s2f.ops
    assert ops.BN_PARTIALS
    model.load_state_dict(st0, strict=True)
    s2f.set_keep_membrane(model, False)
    x = img.cuda()
    outs = []
    for on in (True, False):
        ops.BN_PARTIALS = on
        try:
            model.load_state_dict(st0, strict=True)
            s2f.reset_net(model)
            before = list(ops.BN_PARTIALS_USED)
            with torch.no_grad():
                cls, masks = model(x)
                x1 = model.backbone.downsample1_1(x.unsqueeze(0).repeat(cfg.T, 1, 1, 1, 1))
            outs.append((cls, masks, x1, [a - b for a, b in zip(ops.BN_PARTIALS_USED, before)]))
        finally:
            ops.BN_PARTIALS = True
    s2f.set_keep_membrane(model, True)
    model.load_state_dict(st0, strict=True)
    (c1, m1, d1, used_on), (c0, m0, d0, used_off) = outs
    print("BatchNorm launches fed by partials / statistics passes:", used_on, "with the switch off:", used_off)
    # (the 32x32-stage maps and the decoder's 100-token maps compute their statistics inside their single-pass kernels either way)
    # (and, with sixteen-wavefront single-pass workgroups, every map of up to 16 384 elements per channel: the SDME block's too)
    assert used_off[0] == 0 and used_off[1] >= 15, used_off
    assert used_on[0] >= 15 and used_on[1] <= 8 and used_on[0] + used_on[1] == used_off[1], (used_on, used_off)
    assert rel_l2(d1.cpu(), d0.cpu()) <= 1e-6
    assert torch.isfinite(m1).all() and m1.shape == m0.shape and c1.shape == c0.shape


# ------------------------------------------------------------------------------------------------ other configs, one stage each vs the oracle
def _stage_forced(s2f, so, cfg, st0, model, mod, name, fn, x, seed, conditioning=False):
    """One stage forward + backward with the comparison RE-SEEDED at every neuron: this build runs first (forward hooks record the
    spike counts and the straight-through mask of every Q_IFNode of the run that is compared), then the oracle runs the stage
    emitting exactly these counts (OracleNet.force: the gradient still flows through its own quantiser) -- every neuron of the oracle
    then sees the inputs this build's neuron saw, and `force_diffs` says in how many elements ITS OWN rounding differs (the
    borderline flips).  -> dict(out, gx, p_gap, worst_p, elems, dspk, max_level, dmask, neurons)"""
    pref = name + "."
    model.load_state_dict(st0, strict=True)
    for p in mod.parameters():
        p.grad = None
    s2f.reset_net(model)
    mine, order = {}, []

    def grab(m, inp, out, n):
        if n not in mine:
            order.append(n)
            u = inp[0].detach()
            mine[n] = ((out.detach() * m.D).round().to(torch.uint8).cpu(), ((u >= 0) & (u <= m.D)).cpu())
    hooks = [m.register_forward_hook(lambda m_, i_, o_, n=n: grab(m_, i_, o_, n)) for n, m in mod.named_modules() if isinstance(m, s2f.Q_IFNode)]
    xg = x.cuda().requires_grad_(True)
    out = mod(xg)
    for h in hooks:
        h.remove()
    gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed)) / out.numel() ** 0.5
    out.backward(gy.cuda())
    s2f.ops.wgrad_join()
    # the oracle, forced
    st = {k: (v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k.startswith(pref)) if k.startswith(pref) else v)
          for k, v in st0.items()}
    net = so.OracleNet(st, cfg, training=True)
    net.force = {pref + n if n else name: v[0] for n, v in mine.items()}
    masks = {}
    net.tap_in = lambda n, h: masks.__setitem__(n, (h >= 0) & (h <= cfg.D))
    xo = x.clone().requires_grad_(True)
    yo = getattr(net, fn)(name, xo)
    net.tap_in = None
    yo.backward(gy)
    assert set(net.force_diffs) == set(net.force), set(net.force) ^ set(net.force_diffs)          # every neuron was reached and forced
    dspk = sum(v[0] for v in net.force_diffs.values())
    elems = sum(v[2] for v in net.force_diffs.values())
    max_level = max(v[1] for v in net.force_diffs.values())
    dmask = sum(int((masks[pref + n if n else name].reshape(-1) != v[1].reshape(-1)).sum()) for n, v in mine.items())
    grads = {n: p.grad for n, p in mod.named_parameters() if p.grad is not None}
    ref = {n: st[pref + n].grad for n in grads if st[pref + n].grad is not None}
    p_gap, worst_p = _grad_gaps(grads, ref)
    row = dict(out=rel_l2(out.detach().cpu(), yo.detach()), gx=rel_l2(xg.grad.cpu(), xo.grad), p_gap=p_gap, worst_p=worst_p, elems=elems,
               dspk=dspk, max_level=max_level, dmask=dmask, neurons=len(mine))
    if conditioning:
        # How well conditioned is this stage's gradient in fp32?  The oracle again in fp64 (same forced spikes): `ref64` is what the
        # fp32 ORACLE misses the fp64 result by -- the distance between two correct fp32 implementations -- and `vs64` what this build
        # misses it by.  (Measured at C2, B = 2: the DCN encoder layer's input gradient carries 6e-4 of fp32 noise in the oracle itself
        # and 0.9 % on dcn.input_proj's parameters -- the sampling core's offset gradient is a difference of neighbouring values.)
        st64 = {k: (v.clone().double().requires_grad_(v.is_floating_point() and "running" not in k and k.startswith(pref))
                    if (k.startswith(pref) and v.is_floating_point()) else v) for k, v in st0.items()}
        net64 = so.OracleNet(st64, cfg, training=True)
        net64.force = net.force
        x64 = x.double().requires_grad_(True)
        getattr(net64, fn)(name, x64).backward(gy.double())
        ref64 = {n: st64[pref + n].grad for n in grads if st64[pref + n].grad is not None}
        row.update(gx_vs64=rel_l2(xg.grad.cpu().double(), x64.grad), gx_ref64=rel_l2(xo.grad.double(), x64.grad),
                   p_vs64=_grad_gaps({n: g_.double() for n, g_ in grads.items()}, ref64)[0],
                   p_ref64=_grad_gaps({n: ref[n].double() for n in ref64 if n in ref}, ref64)[0])
    return row


def _other_config_stage_rows(workload, B, H16, W16, stages, seed, conditioning=()):
    """Builds the modules of another BASELINE config at FULL width and runs `stages` = [(name, oracle fn, input shape)] forward +
    backward on both sides from a seeded stage input (`_stage_forced`) -- no full oracle forward is needed."""
    import spike2former_amd as s2f
    from oracle import s2f_oracle as so
    base = so.CONFIGS["C3" if workload == "C3" else "C2"]
    cfg = dataclasses.replace(base, B=B, H=16 * H16, W=16 * W16)
    st0 = so.make_params(cfg, requires_grad=False)
    model = s2f.MODELS.build(s2f.model_cfg("C3" if workload == "C3" else "C2"))
    model.load_state_dict(st0, strict=True)
    model = model.cuda().train()
    rows = {}
    g = torch.Generator().manual_seed(seed)
    for name, fn, shape in stages:
        mod = model
        for part in name.split("."):
            mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
        rows[name] = _stage_forced(s2f, so, cfg, st0, model, mod, name, fn, torch.randn(*shape, generator=g), seed + len(rows),
                                   conditioning=name in conditioning)
    return rows


def _assert_stage_rows(rows, min_neurons):
    print("re-seeded stage gaps:", {k: {a: (f"{b:.1e}" if isinstance(b, float) else b) for a, b in v.items()} for k, v in rows.items()})
    for name, r in rows.items():
        assert r["neurons"] >= min_neurons, (name, r)
        # the borderline flips are few and by one level; the output agrees to round-off whatever flipped (the oracle was re-seeded)
        assert r["dspk"] <= max(1e-4 * r["elems"], 4) and r["max_level"] <= 1.0 and r["dmask"] <= max(1e-4 * r["elems"], 4), (name, r)
        assert r["out"] <= 1e-5, (name, r)
        if r["dmask"] == 0 and "gx_ref64" in r:
            # self-calibrated: this build may miss the fp64 gradient by at most 10x what the fp32 oracle misses it by
            assert r["gx_vs64"] <= max(1e-5, 10 * r["gx_ref64"]) and r["p_vs64"] <= max(1e-3, 10 * r["p_ref64"]), (name, r)
        elif r["dmask"] == 0:        # no straight-through mask bit differs either: the gradients agree to round-off
            assert r["gx"] <= 1e-5 and r["p_gap"] <= 1e-3, (name, r)
        else:
            assert r["gx"] <= 5e-2 and r["p_gap"] <= 2e-1, (name, r)


@pytest.mark.timeout(1800)
def test_c3_stage_forward_backward_vs_oracle():
    """C3 (Cityscapes 1024 x 512, T = 4, 2 per GPU) at FULL size, one stage of each kind that only this shape reaches (round-4 kernels
    that had kernel-level tests only): `block3.0` on the 64 x 32 map -- T B = 8 maps of 2 048 tokens = 16 384 elements per channel,
    the 64-tile single-pass BatchNorm on sixteen wavefronts forward and backward, the attention core over 2 048 tokens -- and a
    pixel-decoder encoder layer on the same map (SepConv_Spike + DCNv3 at 64 x 32 x G32: the BANDED LDS backward, two bands per
    (image, group) + MS_MLP with the 2 048-wide FFN).  Forward + backward against the oracle's autograd with the comparison re-seeded
    at every neuron (`_stage_forced`): outputs to 1e-5, gradients to round-off unless a straight-through mask bit differs."""
    rows = _other_config_stage_rows("C3", 2, 32, 64, [
        ("backbone.block3.0", "_block", (4, 2, 256, 32, 64)),
        ("decode_head.pixel_decoder.encoder.layers.0", "_enc_layer", (4, 2, 32, 64, 256))], 300)
    _assert_stage_rows(rows, 7)


@pytest.mark.timeout(1800)
def test_c5_size_dcn_layer_forward_backward_vs_oracle():
    """The pixel-decoder encoder layer at C5's map (800 x 1344 / 16 = 50 x 84 = 4 200 positions, T = 4, 1 per GPU): widths that are
    not a power of two, the banded DCNv3 backward with three bands, row-walking BatchNorm on rows of 4 200 elements.  (The head is
    the same class for every config; C5's backbone stages have their own fixture, tests/test_gpu_sdtv3.py.)"""
    rows = _other_config_stage_rows("C5", 1, 50, 84, [
        ("decode_head.pixel_decoder.encoder.layers.0", "_enc_layer", (4, 1, 50, 84, 256))], 400)
    _assert_stage_rows(rows, 8)


@pytest.mark.timeout(1800)
def test_c2_stages_at_the_bench_batch_forward_backward_vs_oracle():
    """C2 at the BENCH's own per-GPU batch (B = 2; the other C2 parity tests of this file run the oracle at B = 1).  B changes the
    per-channel row length that selects the BatchNorm kernel form -- at the 32 x 32 stage T B H W = 8 192 elements per channel, exactly
    the single-pass limit (B = 1: 4 096) -- and the GEMM column count (8 x 1 024).  One stage of each kind that lives on that map,
    forward + backward against the oracle's autograd with the comparison re-seeded at every neuron (`_stage_forced`):
    `downsample4` (3 x 3 stride 1, 256 -> 360: implicit 3 x 3 + BatchNorm over 8 192-element rows), `block3.2` and `block4.0`
    (RepConv q | k | v chains, attention over 1 024 tokens, the MLP; 256 and 360 channels), and a pixel-decoder encoder layer
    (SepConv_Spike + DCNv3 + MS_MLP on [4, 2, 32, 32, 256]).  Reference: configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:23-93."""
    rows = _other_config_stage_rows("C2", 2, 32, 32, [
        ("backbone.downsample4", "run_backbone_stage", (4, 2, 256, 32, 32)),
        ("backbone.block3.2", "_block", (4, 2, 256, 32, 32)),
        ("backbone.block4.0", "_block", (4, 2, 360, 32, 32)),
        ("decode_head.pixel_decoder.encoder.layers.0", "_enc_layer", (4, 2, 32, 32, 256))], 600,
        conditioning=("decode_head.pixel_decoder.encoder.layers.0",))
    _assert_stage_rows(rows, 1)
    assert rows["backbone.block3.2"]["neurons"] >= 7 and rows["decode_head.pixel_decoder.encoder.layers.0"]["neurons"] >= 8
