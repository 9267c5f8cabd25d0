"""GPU parity tests of the round-5 kernels, through the C ABI: the pipelined weight gradient (csrc/dwp.hip) against fp64 and
against the round-2 kernel it replaces; the fused clip + AdamW update (csrc/optim.hip) against torch.optim.AdamW behind
torch.nn.utils.clip_grad_norm_ (what the reference's OptimWrapper runs,
configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:137-155)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _operands(B, M, K, L, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    gy = torch.randn(B, M, L, device="cuda", generator=g) * torch.rand(B, M, 1, device="cuda", generator=g) * 1e-3
    x = (torch.randint(0, 9, (B, K, L), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
    return gy, x


# ragged M / K (tile 128 x 256), contraction lengths from one step to many batches, both schedules; L % 32 != 0 (the decoder's
# 100-token maps, C5's 50 x 84 maps, one and a bit steps) on the two-halves schedule
@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("B,M,K,L", [(8, 256, 256, 1024), (2, 700, 256, 4096), (3, 130, 300, 96), (1, 64, 32, 32), (5, 360, 1440, 64),
                                     (2, 1024, 256, 1024), (1, 5, 7, 2048), (8, 256, 256, 100), (4, 360, 256, 4200), (3, 130, 300, 36),
                                     (8, 2048, 256, 100), (1, 64, 520, 1052)])
def test_pipelined_weight_gradient_matches_fp64(cfg, B, M, K, L):
    from spike2former_amd._lib import check, lib
    assert lib.s2f_spike_gemm_dw_pipe_ok(B, M, K, L) == 1
    if L % 32 and cfg == 1:
        gy, x = _operands(B, M, K, L, 3)
        out = torch.zeros(M, K, device="cuda")
        assert lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, K, L, 0, 1, 0, _stream()) == -1
        return
    gy, x = _operands(B, M, K, L, 3)
    want = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    scale = torch.einsum("bml,bkl->mk", gy.abs().double(), x.double()).clamp_min(1e-30)
    for wgs in (0, 7, 1000):                     # default (one workgroup per CU), few long pieces, many short ones
        out = torch.full((M, K), float("nan"), device="cuda")
        check(lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, K, L, 0, cfg, wgs, _stream()), "dw_pipe")
        torch.cuda.synchronize()
        # fp32-equivalent: three exact bf16 terms of dY (24 bits), exact spikes, fp32 accumulation
        assert ((out.double() - want).abs() / scale).max().item() <= 2e-6
    # accumulate != 0 adds into the destination
    base = torch.randn(M, K, device="cuda")
    acc = base.clone()
    check(lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), acc.data_ptr(), B, M, K, L, 1, cfg, 0, _stream()), "dw_pipe")
    torch.cuda.synchronize()
    assert torch.allclose(acc.double(), base.double() + want, rtol=0, atol=2e-6 * scale.max().item() + 1e-6 * base.abs().max().item())


def test_pipelined_weight_gradient_refuses_what_it_cannot_take():
    from spike2former_amd._lib import lib
    assert lib.s2f_spike_gemm_dw_pipe_ok(8, 256, 256, 1050) == 0           # L % 4 != 0 (C5's 25 x 42 maps): rows are not float4-aligned
    assert lib.s2f_spike_gemm_dw_pipe_ok(8, 256, 256, 28) == 0             # less than one step
    assert lib.s2f_spike_gemm_dw_pipe_ok(1, 1 << 16, 64, 1 << 15) == 0     # M L >= 2^30
    assert lib.s2f_spike_gemm_dw_pipe_ok(64, 1 << 10, 64, (1 << 14) + 4) == 0     # ragged: batch M L >= 2^30
    gy, x = _operands(1, 32, 32, 38, 0)
    out = torch.zeros(32, 32, device="cuda")
    assert lib.s2f_spike_gemm_dw_pipe(gy.data_ptr(), x.data_ptr(), out.data_ptr(), 1, 32, 32, 38, 0, 0, 0, _stream()) == -1
    assert b"L % 4" in lib.s2f_last_error()


def test_pipelined_grouped_launch_is_the_single_launches():
    from spike2former_amd._lib import check, lib
    spec = [(8, 256, 256, 1024), (8, 1024, 256, 1024), (2, 700, 256, 4096), (3, 130, 300, 96), (8, 288, 256, 1024), (1, 40, 520, 64),
            (8, 256, 2048, 100), (4, 256, 360, 4200)]          # (the last two: L % 32 != 0 -- the whole launch takes the ragged form)
    keep, flat, want = [], [], []
    for i, (B, M, K, L) in enumerate(spec):
        gy, x = _operands(B, M, K, L, 10 + i)
        pre = torch.randn(M, K, device="cuda")
        out = pre.clone()
        keep.append((gy, x, out))
        flat += [gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, K, L]
        want.append((pre.double() + torch.einsum("bml,bkl->mk", gy.double(), x.double()),
                     torch.einsum("bml,bkl->mk", gy.abs().double(), x.double()).max().item() + pre.abs().max().item()))
    arr = (ctypes.c_int64 * len(flat))(*flat)
    check(lib.s2f_spike_gemm_dw_pipe_grouped(arr, len(spec), 0, 0, _stream()), "dw_pipe_grouped")
    torch.cuda.synchronize()
    for (gy, x, out), (w, sc) in zip(keep, want):
        assert (out.double() - w).abs().max().item() <= 2e-6 * sc


def test_deferred_weight_gradients_take_the_pipelined_kernel(monkeypatch):
    """ops.wgrad_flush routes the jobs the pipelined kernel takes to ONE grouped launch and leaves the others (L < 32) where they
    were; the sums in the sinks are those of the round-2 grouped kernel to fp32 round-off."""
    from spike2former_amd import ops
    from spike2former_amd.ops import core
    res = {}
    for pipe in (True, False):
        monkeypatch.setattr(ops.cfg, "DW_PIPE", pipe)
        sinks = []
        for i, (B, M, K, L) in enumerate([(8, 256, 256, 1024), (8, 256, 2048, 100), (8, 512, 256, 1024), (2, 64, 96, 4096), (4, 128, 128, 28)]):
            gy, x = _operands(B, M, K, L, 20 + i)
            sink = torch.zeros(M, K, device="cuda")
            core._defer_dw(gy, x, sink, B, M, K, L)
            sinks.append(sink)
        core.wgrad_flush()
        torch.cuda.synchronize()
        res[pipe] = sinks
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


# ------------------------------------------------------------------------------------------------ clip + AdamW
class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.backbone = torch.nn.ModuleDict({"a": torch.nn.Linear(37, 19), "b": torch.nn.Conv2d(5, 7, 3)})
        self.head = torch.nn.Linear(130, 66, bias=False)
        self.query_embed = torch.nn.Embedding(10, 6)
        self.big = torch.nn.Parameter(torch.randn(3, 4099, generator=g))        # > one 4 096-element chunk, ragged
        self.scalar = torch.nn.Parameter(torch.randn(1, generator=g))


CUSTOM = {"custom_keys": {"backbone": dict(lr_mult=0.1, decay_mult=1.0), "query_embed": dict(lr_mult=1.0, decay_mult=0.0)}}


@pytest.mark.parametrize("max_norm", [0.01, 1e9, None])
def test_flat_adamw_is_clip_grad_norm_plus_torch_adamw(max_norm):
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.train import FlatAdamW, LinearThenPoly, OptimWrapper
    torch.manual_seed(1)
    ours, ref = _Toy().cuda(), _Toy().cuda()
    ref.load_state_dict(ours.state_dict())
    clip = dict(max_norm=max_norm, norm_type=2) if max_norm else None
    red = FlatGradAllReduce(ours.parameters(), 1)
    opt = FlatAdamW(ours, red, lr=0.001, betas=(0.9, 0.999), weight_decay=0.005, paramwise_cfg=CUSTOM, clip_grad=clip)
    sched = LinearThenPoly(opt, warmup=2, total=10, start_factor=0.1)
    wrap = OptimWrapper(ref, dict(type="AdamW", lr=0.001, betas=(0.9, 0.999), weight_decay=0.005), clip_grad=clip, paramwise_cfg=CUSTOM)
    rsched = LinearThenPoly(wrap.optimizer, warmup=2, total=10, start_factor=0.1)
    g = torch.Generator(device="cuda").manual_seed(5)
    for it in range(4):
        grads = [torch.randn(p.shape, device="cuda", generator=g) * (10.0 ** (it - 2)) for p in ours.parameters()]
        for p, q, gr in zip(ours.parameters(), ref.parameters(), grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        red.gather()
        norm = opt.step()
        sched.step()
        want = torch.nn.utils.clip_grad_norm_(list(ref.parameters()), max_norm, 2) if max_norm else None
        wrap.optimizer.step()
        rsched.step()
        if want is not None:
            assert abs(norm.item() - want.item()) <= 1e-6 * want.item()
        for (n, p), q in zip(ours.named_parameters(), ref.parameters()):
            assert (p - q).abs().max().item() <= 1e-6 * max(q.abs().max().item(), 1e-3), (it, n)
    # versions were bumped (eager use): caches keyed on a weight's version re-convert
    assert all(p._version > 0 for p in ours.parameters())
    sd = opt.state_dict()
    assert sd["step"] == 4 and set(sd["state"]) == {n for n, _ in ours.named_parameters()}
    ref_state = wrap.optimizer.state_dict()["state"]
    for i, (n, _) in enumerate(ref.named_parameters()):
        assert (sd["state"][n]["exp_avg_sq"] - ref_state[i]["exp_avg_sq"]).abs().max().item() <= 1e-6 * ref_state[i]["exp_avg_sq"].abs().max().item() + 1e-30


def test_flat_adamw_follows_compact_and_is_capturable():
    from spike2former_amd.dist import FlatGradAllReduce
    from spike2former_amd.train import FlatAdamW
    m = _Toy().cuda()
    red = FlatGradAllReduce(m.parameters(), 1)
    opt = FlatAdamW(m, red, clip_grad=dict(max_norm=0.01))
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    m.head.weight.grad = None                    # as if its gradient had arrived through a sink
    red.gather()
    red.compact()
    with pytest.raises(RuntimeError, match="rebuild"):
        opt.step()
    opt.rebuild()
    before = [p.detach().clone() for p in m.parameters()]
    red.gather()
    opt.sync_hyper()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step(sync_hyper=False)               # warm-up (step 1)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        opt.step(sync_hyper=False)               # captured, not executed
    graph.replay()                               # step 2
    graph.replay()                               # step 3
    torch.cuda.synchronize()
    assert int(opt.state[4].item()) == 3
    moved = {n: (p.detach() - b).abs().max().item() for (n, p), b in zip(m.named_parameters(), before)}
    # head.weight received no gradient and has no sink: torch.optim.AdamW skips a parameter whose .grad is None entirely (no decay, no
    # moment update), and so does the kernel (round 6: lr < 0 in its table row); everything else moved
    assert moved["head.weight"] == 0.0 and all(d > 0 for n, d in moved.items() if n != "head.weight"), moved


# ------------------------------------------------------------------------------------------------ eval fusion gates
def test_eval_fusion_is_not_taken_when_a_parameter_wants_a_gradient():
    """Frozen input, trainable convolution and BatchNorm affine under bn.eval() (norm_eval fine-tuning): the fused eval kernels
    build no autograd graph, so they must step aside -- the weight and the affine still receive gradients."""
    import spike2former_amd as s2f
    from spike2former_amd import fused, ops
    from spike2former_amd.conv import Conv2d
    torch.manual_seed(0)
    conv = Conv2d(64, 32, 1, bias=False).cuda()
    bn = torch.nn.BatchNorm2d(32).cuda().eval()
    lif = s2f.Q_IFNode().cuda()
    x = ops.Spikes((torch.randint(0, 9, (4, 64, 16, 16), device="cuda").float() / 8).to(torch.bfloat16))
    s2f.reset_net(lif)
    u, y = fused.conv_bn_act(conv, x, bn, lif=lif, want_pre=True)
    u.square().mean().backward()
    assert conv.weight.grad is not None and bn.weight.grad is not None and bn.bias.grad is not None
    assert conv.weight.grad.abs().max().item() > 0
    # with everything frozen the fused kernel IS taken, and gives the same pre-activation
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad_(False)
    s2f.reset_net(lif)
    u2, y2 = fused.conv_bn_act(conv, x, bn, lif=lif, want_pre=True)
    assert not u2.requires_grad
    assert torch.allclose(u2, u.detach(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ implicit 3x3 weight gradient, pipelined
@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("B,M,C,H,W", [(2, 128, 32, 64, 64), (1, 40, 64, 5, 32), (2, 300, 96, 9, 64), (3, 64, 256, 4, 32), (1, 512, 128, 64, 64),
                                       (2, 128, 64, 12, 40), (1, 256, 32, 20, 168), (3, 130, 64, 4, 56), (1, 128, 64, 50, 336)])
def test_pipelined_conv3x3_weight_gradient_matches_conv2d(cfg, B, M, C, H, W):
    """s2f_spike_conv3x3_dw_pipe (tap-major dW [M, 3, 3, C]) against the weight gradient of F.conv2d in fp64: every border (first /
    last row, first / last column), tiles that span several taps (C < 256), ragged M, batch boundaries inside a workgroup's piece,
    widths that are no multiple of the 32-pixel step (a step then spans two image rows: C5's 336- and 168-wide maps)."""
    import ctypes
    import torch.nn.functional as F
    from spike2former_amd._lib import check, lib
    assert lib.s2f_spike_conv3x3_dw_pipe_ok(B, M, C, H, W) == 1
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + C)
    x = (torch.randint(0, 9, (B, C, H, W), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
    gy = torch.randn(B, M, H, W, device="cuda", generator=g)
    w = torch.zeros(M, C, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(gy.double())
    want = w.grad.permute(0, 2, 3, 1).contiguous()                       # [M, ky, kx, C]
    wabs = torch.zeros(M, C, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wabs, None, 1, 1).backward(gy.abs().double())
    scale = wabs.grad.permute(0, 2, 3, 1).clamp_min(1e-30)
    xs = torch.empty(x.numel() + 16, dtype=x.dtype, device=x.device)
    check(lib.s2f_shift1_bf16(x.data_ptr(), xs.data_ptr(), x.numel(), _stream()), "shift1")
    assert torch.equal(xs[7:7 + x.numel()], x.view(-1)) and float(xs[:7].abs().sum()) == 0.0 and float(xs[7 + x.numel():].abs().sum()) == 0.0
    for wgs in (0, 5, 700):
        out = torch.zeros(M, 3, 3, C, device="cuda")
        arr = (ctypes.c_int64 * 9)(gy.data_ptr(), x.data_ptr(), xs.data_ptr(), out.data_ptr(), B, M, C, H, W)
        check(lib.s2f_spike_conv3x3_dw_pipe(arr, 1, cfg, wgs, _stream()), "conv3x3_dw_pipe")
        torch.cuda.synchronize()
        assert ((out.double() - want).abs() / scale).max().item() <= 2e-6, wgs
    assert lib.s2f_spike_conv3x3_dw_pipe_ok(2, 64, 32, 8, 84) == 0          # W % 8 != 0 (C5's 84-wide maps): stays on the round-2 kernel
    assert lib.s2f_spike_conv3x3_dw_pipe_ok(2, 64, 32, 5, 40) == 0          # H W % 32 != 0: a step would span two images


# ------------------------------------------------------------------------------------------------ inference post-processing
@pytest.mark.parametrize("shape", [(2, 5, 16, 32), (1, 3, 7, 4), (1, 100, 64, 64)])
def test_upsample2x_sigmoid_is_interpolate_then_sigmoid(shape):
    """ops.upsample_bilinear(.., sigmoid=True) (s2f_upsample2x_sigmoid_fwd) against F.interpolate(..).sigmoid(): the same taps as the
    plain kernel (bit-identical up-sampled logits), expf's last bits on top."""
    import torch.nn.functional as F
    from spike2former_amd import ops
    x = torch.randn(*shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 6
    size = (2 * shape[2], 2 * shape[3])
    with torch.no_grad():
        got = ops.upsample_bilinear(x, size, sigmoid=True)
        plain = ops.upsample_bilinear(x, size)
    want = F.interpolate(x.double(), size=size, mode="bilinear", align_corners=False).sigmoid()
    assert (got.double() - want).abs().max().item() <= 3e-7
    assert (got - plain.sigmoid()).abs().max().item() <= 2e-7
    # with a gradient wanted the two-kernel form (autograd) is taken
    xg = x.clone().requires_grad_(True)
    y = ops.upsample_bilinear(xg, size, sigmoid=True)
    assert y.requires_grad and torch.allclose(y, got, atol=2e-7, rtol=0)


def test_class_mask_product_writes_every_image_into_its_slice():
    from spike2former_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    cls = torch.rand(3, 100, 19, device="cuda", generator=g)
    mp = torch.rand(3, 100, 24, 40, device="cuda", generator=g)
    with torch.no_grad():
        got = ops.class_mask_product(cls, mp)
    want = torch.einsum("bqc,bqhw->bchw", cls.double(), mp.double())
    assert got.shape == (3, 19, 24, 40) and (got.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("Nq,Nk", [(100, 100), (100, 1024), (100, 16384), (36, 4096)])
def test_inference_attention_and_its_neuron_as_one_launch_pair(Nq, Nk):
    """Under no_grad ops.sdsa(.., lif=) on the decoder's 100-query maps takes s2f_sdsa_lif_fwd_bf16_nomask (attention core + neuron,
    o never written): the same spikes, bit for bit, as the core followed by the neuron's own kernel."""
    from spike2former_amd import ops
    from spike2former_amd.neuron import Q_IFNode, Quant
    from spike2former_amd.ops.core import _new_tok
    TB, C, heads = 8, 256, 8
    g = torch.Generator(device="cuda").manual_seed(Nq + Nk)

    def spikes(n):
        d = (torch.randint(0, 9, (TB, C, n), device="cuda", generator=g).float() / 8).to(torch.bfloat16)
        return ops.Spikes(d, _new_tok(d))
    q, k, v = spikes(Nq), spikes(Nk), spikes(Nk)
    scale = 1.0 / 16
    a, b = Q_IFNode(surrogate_function=Quant()).cuda(), Q_IFNode(surrogate_function=Quant()).cuda()
    with torch.no_grad():
        fused = ops.sdsa(q, k, v, heads, scale, lif=a)
        o = ops.sdsa(q, k, v, heads, scale)
        two = b.fire(o)
    assert isinstance(fused, ops.Spikes) and fused.data.dtype == torch.bfloat16
    assert torch.equal(fused.data, two.data if isinstance(two, ops.Spikes) else two.to(torch.bfloat16))
    assert fused.data.float().abs().sum().item() > 0


@pytest.mark.parametrize("G,B,K,M,N", [(1, 8, 512, 256, 1024), (3, 8, 256, 256, 1024), (1, 2, 147, 32, 4096), (2, 3, 96, 72, 260)])
def test_dense_gemm_bn_lif_eval_is_gemm_then_batchnorm_then_neuron(G, B, K, M, N):
    """s2f_dense_gemm_bn_lif_fwd (dense fp32 input, G weights on channel groups, eval BatchNorm + residual + neuron in the epilogue)
    against the fp64 composition; the pre-activation to fp32 round-off, the spikes equal wherever the membrane is not within round-off
    of a rounding boundary."""
    from spike2former_amd import ops
    g = torch.Generator(device="cuda").manual_seed(G * 100 + K)
    x = torch.randn(B, G * K, N, device="cuda", generator=g)
    ws = [torch.randn(M, K, device="cuda", generator=g) / K ** 0.5 for _ in range(G)]
    C = G * M
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    mean, var = torch.randn(C, device="cuda", generator=g) * 0.2, torch.rand(C, device="cuda", generator=g) + 0.5
    gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.3
    res = torch.randn(B, C, N, device="cuda", generator=g)
    D, eps = 8, 1e-5
    with torch.no_grad():
        u, y = ops.dense_gemm_bn_lif_eval(x, ws, bias, mean, var, gamma, beta, eps, residual=res, want_pre=True, lif=True, D=D, vth=1.0)
        u_only, none = ops.dense_gemm_bn_lif_eval(x, ws, None, mean, var, gamma, beta, eps, want_pre=True, lif=False)
    assert none is None
    z = torch.cat([torch.einsum("mk,bkn->bmn", w.double(), x[:, i * K:(i + 1) * K].double()) for i, w in enumerate(ws)], 1)
    aff = lambda t: (t - mean.double().view(1, -1, 1)) / torch.sqrt(var.double().view(1, -1, 1) + eps) * gamma.double().view(1, -1, 1) \
        + beta.double().view(1, -1, 1)          # noqa: E731
    want_u = aff(z + bias.double().view(1, -1, 1)) + res.double()
    assert (u.double() - want_u).abs().max().item() <= 2e-5 * want_u.abs().max().item()
    assert (u_only.double() - aff(z)).abs().max().item() <= 2e-5 * aff(z).abs().max().item()
    h = want_u.clamp(0, D)
    want_y = torch.round(h) / D
    near = ((h - torch.floor(h)) - 0.5).abs() < 1e-3                      # within round-off of a rounding boundary: either neighbour
    bad = (y.data.double() != want_y) & ~near
    assert int(bad.sum()) == 0 and y.data.dtype == torch.bfloat16


def test_predict_on_the_last_layer_alone_gives_the_logits_of_the_full_head():
    """maskformer_head.PREDICT_LAST_ONLY: the SDME block and the mask contraction for the last decoder layer only -- the same
    segmentation logits as with all L + 1 layers evaluated (eval mode: every op of the block is per sample)."""
    import spike2former_amd as s2f
    from spike2former_amd import maskformer_head
    from spike2former_amd.init_utils import seeded_init
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1"))).cuda().eval()
    s2f.set_keep_membrane(model, False)
    w = s2f.WORKLOADS["C1"]
    img = torch.randn(2, 3, w["H"], w["W"], generator=torch.Generator().manual_seed(3)).cuda()
    outs = {}
    was = maskformer_head.PREDICT_LAST_ONLY
    try:
        for flag in (True, False):
            maskformer_head.PREDICT_LAST_ONLY = flag
            s2f.reset_net(model)
            with torch.no_grad():
                outs[flag] = model(img, mode="logits").clone()
    finally:
        maskformer_head.PREDICT_LAST_ONLY = was
    assert outs[True].shape == outs[False].shape
    assert (outs[True] - outs[False]).abs().max().item() <= 1e-6 * outs[False].abs().max().item()


def test_stateless_inference_fusions_run_and_match_the_unfused_path():
    """Row f4, the fusions of the second half of round 5 at model level (plumbing widths, 256x256 images, reset neurons): the dense
    1x1 + BatchNorm(-pair) + neuron launches (s2f_dense_gemm_bn_lif_fwd) and the sigmoid inside the final up-sampling are really
    taken, and the segmentation logits agree with the two-kernel path (fused.EVAL_FUSION =
    False) to fp32 round-off where no spike sits on a rounding boundary: 10^-4 of the logits' range, same arg-max on 99.9 % of the pixels."""
    import spike2former_amd as s2f
    from spike2former_amd import fused
    from spike2former_amd._lib import lib
    from spike2former_amd.init_utils import seeded_init
    model = seeded_init(s2f.MODELS.build(s2f.model_cfg("C1"))).cuda().eval()
    s2f.set_keep_membrane(model, False)
    img = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(21)).cuda()
    counts = {"s2f_dense_gemm_bn_lif_fwd": 0, "s2f_sdsa_lif_fwd_bf16_nomask": 0, "s2f_upsample2x_sigmoid_fwd": 0}
    origs = {n: getattr(lib, n) for n in counts}

    def wrap(n):
        def f(*a):
            counts[n] += 1
            return origs[n](*a)
        return f
    for n in counts:
        setattr(lib, n, wrap(n))
    outs = {}
    try:
        for on in (True, False):
            fused.EVAL_FUSION = on
            s2f.reset_net(model)
            with torch.no_grad():
                outs[on] = model(img, mode="logits").clone()
            if on:
                taken = dict(counts)
    finally:
        fused.EVAL_FUSION = True
        for n in counts:
            setattr(lib, n, origs[n])
    # (the plumbing config has 10 queries: the attention + neuron pair needs Nq % 4 == 0 and is covered by its own test above)
    assert taken["s2f_dense_gemm_bn_lif_fwd"] >= 8 and taken["s2f_upsample2x_sigmoid_fwd"] == 1, taken
    assert counts["s2f_dense_gemm_bn_lif_fwd"] == taken["s2f_dense_gemm_bn_lif_fwd"]          # ... and not with the switch off
    a, b = outs[True], outs[False]
    rng = (b.max() - b.min()).item()
    assert (a - b).abs().max().item() <= 1e-4 * rng, ((a - b).abs().max().item(), rng)
    assert (a.argmax(1) == b.argmax(1)).float().mean().item() >= 0.999


def test_decoder_layer_with_fused_query_neurons_is_the_add_and_neuron_form():
    """head_layers.FUSED_QUERY_NEURONS: `query + query_pos` and the query (cross-attention) resp. query / key / value (self-attention)
    neurons as one ops.sum2_lif launch.  One decoder layer on the channel-major query stream, 100 queries: outputs bit-identical to the
    add + neuron form, gradients (query, the learnable position term, a projection weight) to fp32 round-off of their accumulation."""
    import spike2former_amd as s2f
    from spike2former_amd import head_layers, ops
    from spike2former_amd.init_utils import seeded_init
    T, B, C, Nq, L = 4, 2, 256, 100, 64
    layer = seeded_init(head_layers.DetrTransformerDecoderLayer(
        self_attn_cfg=dict(embed_dims=C, num_heads=8, batch_first=True), cross_attn_cfg=dict(embed_dims=C, num_heads=8, batch_first=True),
        ffn_cfg=dict(embed_dims=C, feedforward_channels=512, num_fcs=2))).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(4)
    q0 = torch.randn(T, B, C, Nq, device="cuda", generator=g) * 2
    pos0 = torch.randn(B, C, Nq, device="cuda", generator=g)
    mem = torch.randn(T * B, C, L, device="cuda", generator=g) * 2
    kpos = torch.randn(B, C, L, device="cuda", generator=g)
    lvl = torch.randn(C, device="cuda", generator=g) * 0.1
    wq = torch.randn(T, B, Nq, C, device="cuda", generator=g)
    res = {}
    was = head_layers.FUSED_QUERY_NEURONS
    try:
        for flag in (True, False):
            head_layers.FUSED_QUERY_NEURONS = flag
            s2f.reset_net(layer)                      # (also forgets the neurons' shared-launch memo of the previous pass)
            for p in layer.parameters():
                p.grad = None
            q = q0.clone().requires_grad_(True)
            pos = pos0.clone().requires_grad_(True)
            yk, yv = ops.sum2_lif(mem, lvl, kpos, B, 8, 1.0)
            out, _ = layer.forward_stream(q, pos, kv_spikes=(yk.view(T, B, C, L), yv.view(T, B, C, L)), last=True)
            (out * wq).sum().backward()
            res[flag] = (out.detach().clone(), q.grad.clone(), pos.grad.clone(), layer.self_attn.attn.q_conv[0].weight.grad.clone(),
                         layer.cross_attn.attn.q_conv[0].weight.grad.clone())
    finally:
        head_layers.FUSED_QUERY_NEURONS = was
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert (x - y).abs().max().item() <= 2e-6 * max(y.abs().max().item(), 1e-6), (x - y).abs().max().item()
    assert a[2].abs().max().item() > 0
