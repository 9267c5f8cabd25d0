"""Round 6: the kernels and host paths added this round, on the GPU, against the oracle / the reference's vectors / fp64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_leaky_node_kernels_vs_reference_vectors(golden):
    """neuron.LIFNode on csrc/lif.hip `lif_leaky_*` against the reference's own LIFNode (tests/golden/lif_leaky_kat.npz): five
    stateful calls, BPTT through the membrane chain -- spikes, final membrane and both gradients BIT-EXACT."""
    import spike2former_amd as s2f
    g = golden("lif_leaky_kat.npz")
    for tag in "abcd":
        for state in ("reset", "v0"):
            k = f"{tag}_{state}"
            tau, di = float(g[f"{k}_cfg"][0]), bool(g[f"{k}_cfg"][1])
            xs = T(g[f"{k}_x"]).cuda().requires_grad_(True)
            n = s2f.LIFNode(tau=tau, decay_input=di, surrogate_function=s2f.Quant())
            v0 = None
            if state == "v0":
                v0 = T(g[f"{k}_v0"]).cuda().requires_grad_(True)
                n.v = v0
            ys = torch.stack([n(xs[t]) for t in range(xs.shape[0])])
            ((ys * T(g[f"{k}_wy"]).cuda()).sum() + (n.v * T(g[f"{k}_wv"]).cuda()).sum()).backward()
            assert torch.equal(ys.detach().cpu(), T(g[f"{k}_y"])), k
            assert torch.equal(n.v.detach().cpu(), T(g[f"{k}_vT"])), k
            assert torch.equal(xs.grad.cpu(), T(g[f"{k}_gx"])), k
            if v0 is not None:
                assert torch.equal(v0.grad.cpu(), T(g[f"{k}_gv0"])), k


@pytest.mark.parametrize("n", [0, 1, 3, 255, 257, 4099, 1 << 18])
def test_leaky_node_kernels_vs_c_oracle_ragged(n):
    """ragged sizes, both charge forms, D = 4 and 8, a stateful second call: bit-exact against oracle/lif_ref.c"""
    from oracle import lif_ref
    from spike2former_amd import ops
    rng = np.random.default_rng(n + 5)
    for di in (True, False):
        for D, tau in ((8, 2.0), (4, 1.7)):
            x = (rng.standard_normal((2, n)) * 3 + 1).astype(np.float32)
            gy = rng.standard_normal((2, n)).astype(np.float32)
            gv = rng.standard_normal(n).astype(np.float32)
            wy, vT, _, inr = lif_ref.leaky_seq_fwd(x, None, D=D, tau=tau, decay_input=di)
            wgx, _ = lif_ref.leaky_seq_bwd(gy, inr, gv, D=D, tau=tau, decay_input=di)
            xs = T(x).cuda().requires_grad_(True)
            y0, v = ops.lif_leaky(xs[0], None, D, 1.0, tau, di)
            y1, v = ops.lif_leaky(xs[1], v, D, 1.0, tau, di)
            if n:
                ((y0 * T(gy[0]).cuda()).sum() + (y1 * T(gy[1]).cuda()).sum() + (v * T(gv).cuda()).sum()).backward()
                assert torch.equal(xs.grad.cpu(), T(wgx)), (n, di, D)
            assert torch.equal(torch.stack([y0, y1]).detach().cpu(), T(wy)) and torch.equal(v.detach().cpu(), T(vT)), (n, di, D)


@pytest.mark.parametrize("B,M,N,K", [(1, 1, 1, 1), (3, 5, 7, 9), (2, 64, 64, 16), (2, 65, 130, 33), (8, 72, 10, 72), (4, 21, 16, 64),
                                     (1, 300, 100, 257)])
def test_bmm_small_vs_fp64(B, M, N, K):
    """csrc/bmm.hip: any shape, strided / transposed / broadcast operands read in place, the batch-reduced (weight-gradient) form;
    fp32 multiply-adds in ascending k: within 2e-6 of sum |a||b| of the fp64 product, and bit-repeatable."""
    from spike2former_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + M)
    a = torch.randn(B, M, K, generator=g).cuda()
    b = torch.randn(B, K, N, generator=g).cuda()
    want = torch.bmm(a.double().cpu(), b.double().cpu())
    bound = 2e-6 * torch.bmm(a.abs().double().cpu(), b.abs().double().cpu()) + 1e-30
    got = ops.bmm_small(a, b)
    assert ((got.double().cpu() - want).abs() <= bound).all()
    assert torch.equal(got, ops.bmm_small(a, b))
    # transposed views and a broadcast (stride 0) operand
    at = a.transpose(1, 2).contiguous().transpose(1, 2)
    assert torch.equal(ops.bmm_small(at, b), got)
    w = a[0]
    gotw = ops.bmm_small(w.unsqueeze(0).expand(B, -1, -1), b)
    wantw = torch.matmul(w.double().cpu(), b.double().cpu())
    assert ((gotw.double().cpu() - wantw).abs() <= 2e-6 * torch.matmul(w.abs().double().cpu(), b.abs().double().cpu()) + 1e-30).all()
    # batch-reduced
    red = ops.bmm_small(a, b, reduce_batch=True)
    assert ((red.double().cpu() - want.sum(0)).abs() <= bound.sum(0)).all()


def test_no_vendor_gemm_on_the_plumbing_configuration():
    """One train step of C1_64 (10-query rows, 4 x 4 .. 32 x 32 maps: every shape the matrix-core kernels refuse) under ops.STRICT:
    no op may leave the package's kernels, and rocBLAS / hipBLASLt are never entered (torch.bmm / matmul / einsum are patched to
    raise for the duration)."""
    import spike2former_amd as s2f
    from oracle import s2f_oracle as so
    from spike2former_amd import ops
    cfg = so.CONFIGS["C1_64"]
    model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
    model.load_state_dict(so.make_params(cfg, requires_grad=False), strict=True)
    model.cuda().train()
    names = ("bmm", "matmul", "einsum", "mm", "baddbmm", "addmm")
    saved = {k: getattr(torch, k) for k in names}
    lin = torch.nn.functional.linear

    def refuse(*a, **k):
        raise AssertionError("a vendor GEMM entry point was called on the product path")
    before, strict = dict(ops.FALLBACKS), ops.STRICT
    try:
        for k in names:
            setattr(torch, k, refuse)
        torch.nn.functional.linear = refuse
        ops.STRICT = True
        s2f.reset_net(model)
        cls, masks = model(so.synthetic_image(cfg).cuda())
        s2f.headline_loss(cls, masks).backward()
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            setattr(torch, k, v)
        torch.nn.functional.linear = lin
        ops.STRICT = strict
    assert dict(ops.FALLBACKS) == before
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    assert len(grads) > 500 and all(torch.isfinite(g).all() for g in grads)


def test_eager_forward_after_graphed_training_reads_the_updated_weights():
    """ADVICE r5 (medium).  A replayed hipGraph that holds the AdamW update re-splits the weights at its START and updates them at
    its END: after a replay the version-keyed conversion caches (bf16 terms / packs, composed eval-mode BatchNorm affines) hold the
    weights from BEFORE the last update under unchanged version counters.  GraphedStep bumps the versions after every replay
    (FlatAdamW.mark_updated), so an eager forward -- validation, predict() -- re-converts: it must equal a FRESH model that loads
    the trained state_dict, in train mode (batch statistics) and in eval mode (the fused eval path with its cached affines)."""
    import spike2former_amd as s2f
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.graph import GraphedStep
    from spike2former_amd.init_utils import seeded_init
    from spike2former_amd.train import FlatAdamW
    w = s2f.WORKLOADS["C1_64"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1_64"))).cuda().train()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(7)).cuda()
    with torch.no_grad():                                    # an eager eval forward BEFORE training fills every version-keyed cache
        model.eval(); s2f.reset_net(model); model(img); model.train()
    red = FlatGradAllReduce(model.parameters(), 1)
    red.install_sinks()
    s2f.reset_net(model); red.zero()
    s2f.headline_loss(*model(img)).backward()
    s2f.ops.wgrad_join(); red.gather(); red.compact()
    # a large learning rate: one stale step must be visible far above round-off
    opt = FlatAdamW(model, red, lr=0.05, weight_decay=0.005, clip_grad=dict(max_norm=1e9))
    gs = GraphedStep(model, s2f.headline_loss, img, grad_buffer=red, warmup=1, optimizer=opt)
    for _ in range(3):
        gs()
    torch.cuda.synchronize()
    fresh = s2f.MODELS.build(s2f.model_cfg("C1_64")).cuda()
    fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()}, strict=True)
    s2f.set_keep_membrane(fresh, False)
    for mode in ("eval", "train"):
        getattr(model, mode)(); getattr(fresh, mode)()
        with torch.no_grad():
            s2f.reset_net(model); got = model(img)
            s2f.reset_net(fresh); want = fresh(img)
        for a, b in zip(got, want):
            assert torch.equal(a, b), (mode, (a - b).abs().max().item())
    red.close()


@pytest.mark.parametrize("N,C,L", [(8, 32, 1024),          # bn_fused_fwd_kernel (8 192 elements per channel: the C2 bench shape)
                                   (8, 32, 100),           # bn_small_fwd_kernel (one wavefront per channel)
                                   (4, 32, 4200),          # bn_mid_fwd_kernel (rows that are no whole tiles)
                                   (2, 16, 16)])           # the 4 x 4 maps of the plumbing configuration
def test_single_pass_batchnorm_statistics_survive_a_large_mean(N, C, L):
    """Channels whose variance is far below their squared mean (mean 50, std 1e-2 .. 1): the statistics as shifted sums (round 6,
    S2F_BN_PIVOT) against an fp64 two-pass BatchNorm.  E[x^2] - E[x]^2 on fp32 squares carries ~1e-7 mean^2 = 2.5e-4 of absolute
    variance error here -- larger than the variance of the quiet channels: their rstd, and with it every input gradient of the
    channel, came out wrong by tens of per cent while the forward (x - mean is tiny) looked fine.  One channel is exactly constant."""
    from spike2former_amd import ops
    g = torch.Generator().manual_seed(L)
    std = torch.logspace(-2, 0, C).view(1, C, 1)
    z = 50.0 + std * torch.randn(N, C, L, generator=g)
    z[:, 0] = 3.25                                                     # a constant channel: var must be 0 exactly
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gu = torch.randn(N, C, L, generator=g)
    zd = z.double().requires_grad_(True)
    mean = zd.mean((0, 2), keepdim=True)
    var = ((zd - mean) ** 2).mean((0, 2), keepdim=True)
    u64 = (zd - mean) / torch.sqrt(var + 1e-5) * gamma.double().view(1, C, 1) + beta.double().view(1, C, 1)
    (u64 * gu.double()).sum().backward()
    zc = z.cuda().requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    u = ops.bn_act(zc, None, gamma.cuda(), beta.cuda(), rm, rv, nbt, True, 0.1, 1e-5, lif=False, want_pre=True)[0]
    (u * gu.cuda()).sum().backward()
    # x_hat = (z - mean) * rstd is O(1): an rstd off by 1 % moves u by 1e-2.  fp32 round-off of z - mean at |z| = 50 is 4e-6 absolute,
    # i.e. up to 4e-6 / 1e-2 = 4e-4 of a quiet channel's x_hat -- the bound below leaves room for exactly that
    assert (u.detach().cpu().double() - u64.detach()).abs().max().item() <= 2e-3
    ref = zd.grad
    err = (zc.grad.cpu().double() - ref).abs().amax((0, 2)) / ref.abs().amax((0, 2)).clamp_min(1e-30)
    assert err[1:].max().item() <= 5e-3, err
    assert abs(rv[0].item() - 0.9) <= 1e-7                              # running_var of the constant channel: 0.9 * 1 + 0.1 * 0


@pytest.mark.parametrize("T,B,heads,d,Nq,Nk", [(2, 2, 8, 8, 8, 8), (4, 2, 8, 32, 100, 1000), (1, 3, 2, 45, 7, 130), (3, 1, 4, 64, 65, 64)])
def test_masked_attention_core_vs_fp64(T, B, heads, d, Nq, Nk):
    """ops.sdsa_masked (csrc/sdsa_masked.hip): out = (scale q k^T).masked_fill(mask, 0) v, mask [B, heads, Nq, Nk] shared over the time
    steps, on spike operands -- every product and partial sum is exact in fp32, so the result must equal the fp64 evaluation of the
    reference's expression (mmcv_spike/transformer.py:262-272) BIT FOR BIT, forward and all three gradients (with an integer-valued
    output gradient); ragged token counts, head dimensions up to 64, t != b."""
    from spike2former_amd import ops
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    C = heads * d
    q = torch.randint(0, 9, (T * B, C, Nq), generator=g).float() / 8
    k = torch.randint(0, 9, (T * B, C, Nk), generator=g).float() / 8
    v = torch.randint(0, 9, (T * B, C, Nk), generator=g).float() / 8
    mask = torch.rand(B * heads, Nq, Nk, generator=g) < 0.4
    go = torch.randint(-4, 5, (T * B, C, Nq), generator=g).float()
    scale = 1.0 / 16
    qd, kd, vd = (t.double().view(T, B, heads, d, -1).transpose(3, 4).requires_grad_(True) for t in (q, k, v))          # [t, b, h, n, d]
    scores = (qd @ kd.transpose(3, 4)) * scale
    scores = scores.masked_fill(mask.view(1, B, heads, Nq, Nk), 0)
    o64 = (scores @ vd).transpose(3, 4).reshape(T * B, C, Nq)
    o64.backward(go.double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    o = ops.sdsa_masked(qc, kc, vc, mask.cuda(), heads, scale, B)
    o.backward(go.cuda())
    assert torch.equal(o.detach().cpu().double(), o64.detach())
    for mine, ref in ((qc, qd), (kc, kd), (vc, vd)):
        want = ref.grad.transpose(3, 4).reshape(mine.shape)
        assert torch.equal(mine.grad.cpu().double(), want)
    # no mask bit set: the unmasked core (q (k^T v) on the matrix cores / VALU kernels of sdsa.hip)
    o0 = ops.sdsa_masked(qc.detach(), kc.detach(), vc.detach(), torch.zeros_like(mask).cuda(), heads, scale, B)
    assert torch.equal(o0, ops.sdsa(qc.detach(), kc.detach(), vc.detach(), heads, scale))


def test_glue_mode_routes_the_residual_aten_calls_of_a_step():
    """ops.GlueMode on a C1_64 train step: the aten calls that the module code and autograd's engine still make (gradient accumulation
    where two consumers meet, alpha * spikes, sigmoid, stack / cat, copies, fills, small sums) run on csrc/glue.hip.  Element-wise
    routes are the same IEEE operations: mask logits bit-identical; the class scores pass one routed mean over T (fixed-order sum): 1e-6;
    gradients 1e-5.  Under STRICT_GLUE nothing that touches a CUDA tensor may reach ATen."""
    import spike2former_amd as s2f
    from oracle import s2f_oracle as so
    from spike2former_amd import ops
    cfg = so.CONFIGS["C1_64"]
    model = s2f.MODELS.build(s2f.model_cfg("C1_64"))
    model.load_state_dict(so.make_params(cfg, requires_grad=False), strict=True)
    model.cuda().train()
    s2f.set_keep_membrane(model, False)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    img = so.synthetic_image(cfg, seed=9).cuda()
    runs = []
    strict = ops.STRICT_GLUE
    try:
        for on in (False, True):
            model.load_state_dict(state)
            s2f.reset_net(model); model.zero_grad(set_to_none=True)
            ops.glue.reset_counts()
            ops.STRICT_GLUE = on
            with ops.glue_mode(force=on):
                cls, masks = model(img)
                s2f.headline_loss(cls, masks).backward()
            torch.cuda.synchronize()
            runs.append((cls.detach().clone(), masks.detach().clone(),
                         {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        ops.STRICT_GLUE = strict
    assert sum(ops.glue.ROUTED.values()) > 100 and not ops.glue.UNROUTED, (dict(ops.glue.ROUTED), dict(ops.glue.UNROUTED))
    (c0, m0, g0), (c1, m1, g1) = runs
    assert torch.equal(m0, m1)
    assert (c0 - c1).abs().max().item() <= 1e-6 * c0.abs().max().item()
    assert g0.keys() == g1.keys()
    gscale = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        # the generators' metric (oracle/gen_golden.py): gradients that are mathematically zero (a bias in front of a BatchNorm) are pure
        # rounding noise of size 1e-8 x the largest gradient, and the split-contraction atomics reorder it from run to run
        assert (g0[k] - g1[k]).abs().max().item() <= 1e-5 * (g0[k].abs().max().item() + 5e-3 * gscale), k


def test_fanout_ports_sum_the_same_gradients_as_the_autograd_engine():
    """cfg.FANOUT_PORTS: the residual branch of `x + f(Q_IFNode(x))` and the second reader of a spike map send their gradients to
    extra outputs of the neuron's autograd node, and its backward kernel sums them -- (g1 + g2) / D and STE(.) + skip are the IEEE
    operations the engine's add launches performed, so one C2 training step gives the same outputs and gradients with the ports off
    and on: equal up to what two runs of the SAME setting differ by (the split-contraction weight gradients add partial tiles with
    fp32 atomics: ~5e-6 relative) -- the unit tests of the ports assert bit-identity on the kernels themselves."""
    import spike2former_amd as s2f
    from spike2former_amd import ops
    from spike2former_amd.init_utils import seeded_init
    w = s2f.WORKLOADS["C2"]
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C2"))).cuda().train()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(1, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(11)).cuda()
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(on):
        was = ops.FANOUT_PORTS
        ops.FANOUT_PORTS = on
        try:
            # (the running statistics are an INPUT of a training forward: BNAndPadLayer pads with BN(0) of the updated running
            # statistics, sdtv2.py:48-83 -- every run starts from the same buffers)
            model.load_state_dict(state)
            s2f.reset_net(model); model.zero_grad(set_to_none=True)
            cls, masks = model(img)
            s2f.headline_loss(cls, masks).backward()
            ops.wgrad_join()
            torch.cuda.synchronize()
        finally:
            ops.FANOUT_PORTS = was
        return cls.detach().clone(), masks.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    c0, m0, g0 = run(False)
    c0b, m0b, g0b = run(False)
    c1, m1, g1 = run(True)
    for a, b, r in ((c0, c1, c0b), (m0, m1, m0b)):          # (the forward does not depend on the setting at all)
        noise = (a - r).abs().max().item()
        assert (a - b).abs().max().item() <= 4 * noise + 1e-6 * a.abs().max().item(), ((a - b).abs().max().item(), noise)
    assert g0.keys() == g1.keys() and len(g0) > 500
    worst = 0.0
    for k in g0:
        noise = (g0[k] - g0b[k]).abs().max().item()
        d = (g0[k] - g1[k]).abs().max().item()
        scale = g0[k].abs().max().item() + 1e-30
        assert d <= 4 * noise + 1e-4 * scale, (k, d, noise, scale)          # (a dropped addend would be O(1); atomics' order is ~5e-6)
        worst = max(worst, d / scale)
    print(f"ports on vs off: worst relative gradient difference {worst:.2e}; forward off vs off again "
          f"{(c0 - c0b).abs().max().item():.2e} / {(m0 - m0b).abs().max().item():.2e}, off vs on {(c0 - c1).abs().max().item():.2e} / "
          f"{(m0 - m1).abs().max().item():.2e}")


@pytest.mark.parametrize("N,C,L", [(8, 16, 1024), (2, 32, 65536), (8, 3, 16800), (8, 8, 100), (2, 7, 1052)])
def test_batchnorm_neuron_second_reader_port_is_the_engine_sum(N, C, L):
    """Two readers of the spike map of a fused BatchNorm + neuron: with cfg.FANOUT_PORTS the second reader's gradient reaches the
    backward kernels on the spare handle and is summed where g_y is read (single-pass and row-walking kernels; the other forms add on
    the host) -- bit-identical to the autograd engine's own add, for z, gamma, beta and the residual."""
    from spike2former_amd import ops
    out = []
    for on in (False, True):
        was = ops.FANOUT_PORTS
        ops.FANOUT_PORTS = on
        try:
            g = torch.Generator().manual_seed(N + C + L)
            z = (torch.randn(N, C, L, generator=g) * 2 + 0.5).cuda().requires_grad_(True)
            gamma = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
            beta = torch.randn(C, generator=g).cuda().requires_grad_(True)
            r = torch.randn(N, C, L, generator=g).cuda().requires_grad_(True)
            rm, rv, nbt = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros((), dtype=torch.int64, device="cuda")
            u, y, _ = ops.bn_act(z, None, gamma, beta, rm, rv, nbt, True, 0.1, 1e-5, residual=r, lif=True, want_pre=True)
            w0, w1, w2 = (torch.randn(u.shape, generator=g).cuda() for _ in range(3))
            y2 = y.second()
            assert (y2.tok is not y.tok) == (on and y.tok2 is not None)
            ((u * w0).sum() + (y.float() * w1).sum() + (y2.float() * w2).sum()).backward()
            out.append((z.grad.clone(), gamma.grad.clone(), beta.grad.clone(), r.grad.clone()))
        finally:
            ops.FANOUT_PORTS = was
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [1024, 4100, 1 << 20])
def test_neuron_ports_are_the_engine_sums(n):
    """ops.lif / ops.sum2_lif with a second reader and a residual branch on the input: ports on == ports off, bit for bit."""
    from spike2former_amd import ops
    out = []
    for on in (False, True):
        was = ops.FANOUT_PORTS
        ops.FANOUT_PORTS = on
        try:
            g = torch.Generator().manual_seed(n)
            x = (torch.randn(n, generator=g) * 3 + 1).cuda().requires_grad_(True)
            w1, w2, w3 = (torch.randn(n, generator=g).cuda() for _ in range(3))
            y, _, xs = ops.lif(x, None, 8, 1.0, False, None, spikes=True, skip=True)
            ((y.float() * w1).sum() + (y.second().float() * w2).sum() + (xs * w3).sum()).backward()
            res = [x.grad.clone()]
            if n % 4 == 0:
                B, C, L = 2, 4, n // 32
                x3 = (torch.randn(4 * B, C, L, generator=g) * 3 + 1).cuda().requires_grad_(True)
                pos = torch.randn(B, C, L, generator=g).cuda().requires_grad_(True)
                e = torch.zeros(C).cuda()
                ws = [torch.randn(4 * B, C, L, generator=g).cuda() for _ in range(5)]
                yk, yv, xs3 = ops.sum2_lif(x3, e, pos, B, 8, 1.0, skip=True)
                ((yk.float() * ws[0]).sum() + (yk.second().float() * ws[1]).sum() + (yv.float() * ws[2]).sum()
                 + (yv.second().float() * ws[3]).sum() + (xs3 * ws[4]).sum()).backward()
                res += [x3.grad.clone(), pos.grad.clone()]
            out.append(res)
        finally:
            ops.FANOUT_PORTS = was
    assert len(out[0]) == len(out[1])
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,M,C,H,W", [(2, 128, 64, 8, 64), (1, 256, 32, 20, 168), (3, 130, 64, 4, 56), (2, 32, 128, 16, 16), (2, 64, 32, 8, 84)])
def test_conv3x3_weight_gradient_in_the_weights_layout(B, M, C, H, W):
    """Both implicit 3x3 weight-gradient kernels with the weight-layout option (dW [M, C, 3, 3], added into its destination): the same
    value as the tap-major product permuted -- same products, same sums per element up to the order of the partial tiles' atomics --
    against fp64 conv2d; accumulates onto what the destination holds."""
    import ctypes
    import torch.nn.functional as F
    from spike2former_amd._lib import check, lib
    from spike2former_amd.ops.core import _stream
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + C)
    x = (torch.randint(0, 9, (B, C, H, W), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
    gy = torch.randn(B, M, H, W, device="cuda", generator=g)
    w = torch.zeros(M, C, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(gy.double())
    wabs = torch.zeros(M, C, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wabs, None, 1, 1).backward(gy.abs().double())
    scale = wabs.grad.clamp_min(1e-30)
    base = torch.randn(M, C, 3, 3, device="cuda", generator=g)
    out = base.clone()
    check(lib.s2f_spike_conv3x3_dw_bf16(gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, C, H, W, 3, _stream()), "conv3x3_dw_bf16")
    assert (((out - base).double() - w.grad).abs() / scale).max().item() <= 4e-6
    tap = torch.empty(M, 3, 3, C, device="cuda")
    check(lib.s2f_spike_conv3x3_dw_bf16(gy.data_ptr(), x.data_ptr(), tap.data_ptr(), B, M, C, H, W, 0, _stream()), "conv3x3_dw_bf16")
    assert ((tap.permute(0, 3, 1, 2).double() - w.grad).abs() / scale).max().item() <= 2e-6
    if lib.s2f_spike_conv3x3_dw_pipe_ok(B, M, C, H, W):
        xs = torch.empty(x.numel() + 16, dtype=x.dtype, device=x.device)
        check(lib.s2f_shift1_bf16(x.data_ptr(), xs.data_ptr(), x.numel(), _stream()), "shift1")
        out = base.clone()
        arr = (ctypes.c_int64 * 9)(gy.data_ptr(), x.data_ptr(), xs.data_ptr(), out.data_ptr(), B, M, C, H, W)
        check(lib.s2f_spike_conv3x3_dw_pipe(arr, 1, 2, 0, _stream()), "conv3x3_dw_pipe")
        assert (((out - base).double() - w.grad).abs() / scale).max().item() <= 4e-6


def test_transpose_pass_through_and_fan_out_are_the_engine_sums():
    """ops.transpose_last2(skip=True) and ops.fan_out: the second reader's gradient summed inside the transposition's adjoint, the
    gradients of n readers summed by one launch in the engine's accumulation order -- ports on == ports off, bit for bit."""
    from spike2former_amd import ops
    out = []
    for on in (False, True):
        was = ops.FANOUT_PORTS
        ops.FANOUT_PORTS = on
        try:
            g = torch.Generator().manual_seed(3)
            x = torch.randn(8, 100, 256, generator=g).cuda().requires_grad_(True)
            w1, w2 = torch.randn(8, 256, 100, generator=g).cuda(), torch.randn(8, 100, 256, generator=g).cuda()
            y, xs = ops.transpose_last2(x, skip=True)
            ((y * w1).sum() + (xs * w2).sum()).backward()
            p = torch.randn(2, 256, 100, generator=g).cuda().requires_grad_(True)
            ws = [torch.randn(2, 256, 100, generator=g).cuda() for _ in range(12)]
            fans = ops.fan_out(p, 12)
            assert len(fans) == 12 and all(torch.equal(f, p) for f in fans)
            loss = 0
            for f, w in zip(fans, ws):          # readers in forward order: the engine accumulates their gradients last reader first
                loss = loss + (f * w).sum()
            loss.backward()
            out.append((x.grad.clone(), p.grad.clone()))
        finally:
            ops.FANOUT_PORTS = was
    for a, b in zip(*out):
        assert torch.equal(a, b)
