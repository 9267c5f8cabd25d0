"""GPU parity tests proper: the HIP kernels, called through the C ABI (spike2former_amd.ops -> libs2f_hip.so), against
the oracle and the committed golden vectors.  Bit-exact for the neuron and the spike attention core; stated fp
tolerances for DCNv3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(True) if grad else t


@pytest.fixture(scope="module")
def ops():
    from spike2former_amd import ops
    return ops


@pytest.fixture(scope="module")
def so():
    from oracle import s2f_oracle
    return s2f_oracle


def unpack_mask(words, n):
    """Inverse of the wave-ballot layout documented in include/s2f.h."""
    w = words.cpu().numpy().view(np.uint64).reshape(-1, 4)
    tiles = w.shape[0]
    out = np.zeros(tiles * 256, dtype=bool)
    lanes = np.arange(64, dtype=np.uint64)
    for j in range(4):
        bits = (w[:, j][:, None] >> lanes[None, :]) & np.uint64(1)
        out.reshape(tiles, 64, 4)[:, :, j] = bits.astype(bool)
    return out[:n]


# ----------------------------------------------------------------------------------------------- neuron
def test_lif_known_answers(ops, golden):
    g = golden("lif_kat.npz")
    x = T(g["kat_x"])
    v = None
    for t in range(4):
        y, v = ops.lif(x, v)
        assert np.array_equal((y * 8).cpu().numpy(), g["kat_counts"][t])
    assert np.array_equal(v.cpu().numpy(), g["kat_v_final"])
    y, _ = ops.lif(T(g["half_x"]))
    assert np.array_equal((y * 8).cpu().numpy(), g["half_counts"])       # round-half-to-even, clamp edges
    x = T(g["grad2_x"], grad=True)
    y1, v = ops.lif(x)
    y2, v = ops.lif(x, v)
    (y1.sum() + y2.sum()).backward()
    assert np.array_equal(x.grad.cpu().numpy(), g["grad2_gx"])


@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 257, 1000, 4099, 1 << 20])
def test_lif_vs_c_oracle_ragged_sizes(ops, n):
    from oracle import lif_ref
    from spike2former_amd._lib import lib
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 4 + 2).astype(np.float32)
    x[: n // 7] = np.round(x[: n // 7] * 2) / 2          # plenty of exact .5 ties
    v0 = rng.standard_normal(n).astype(np.float32)
    if n == 0:
        y, v = ops.lif(torch.zeros(0, device="cuda"))
        assert y.numel() == 0
        return
    ry, rv, rc, rin = lif_ref.seq_fwd(x[None], v0)
    xt, vt = T(x, grad=True), T(v0, grad=True)
    stats = ops.new_stats("cuda")
    y, v = ops.lif(xt, vt, stats=stats)
    assert np.array_equal(y.detach().cpu().numpy(), ry[0]) and np.array_equal(v.detach().cpu().numpy(), rv)
    assert ops.read_stats(stats).tolist() == [int(rc.sum()), int((rc != 0).sum())]
    gy, gv = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    (y * T(gy)).sum().backward(retain_graph=True)
    rgx, _ = lif_ref.seq_bwd(gy[None], rin)
    assert np.array_equal(xt.grad.cpu().numpy(), rgx[0]) and np.array_equal(vt.grad.cpu().numpy(), rgx[0])
    xt.grad = None; vt.grad = None
    ((y * T(gy)).sum() + (v * T(gv)).sum()).backward()
    rgx, _ = lif_ref.seq_bwd(gy[None], rin, gv)
    assert np.array_equal(xt.grad.cpu().numpy(), rgx[0])
    # the packed in-range mask and the optional u8 counts, straight through the C ABI
    yy = torch.empty(n, device="cuda"); cnt = torch.empty(n, dtype=torch.uint8, device="cuda")
    mask = torch.zeros(int(lib.s2f_lif_mask_words(n)), dtype=torch.int64, device="cuda")
    assert lib.s2f_lif_fwd(xt.data_ptr(), vt.data_ptr(), yy.data_ptr(), None, mask.data_ptr(), cnt.data_ptr(), None, n,
                           1.0, 8, 0, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(unpack_mask(mask, n), rin[0].astype(bool)) and np.array_equal(cnt.cpu().numpy(), rc[0])
    # the same call writing its spikes as bf16 (y_bf16 = 1): exactly the fp32 values
    if n % 4 == 0 and n > 0:
        yb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        assert lib.s2f_lif_fwd(xt.data_ptr(), vt.data_ptr(), yb.data_ptr(), None, None, None, None, n, 1.0, 8, 1, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(yb.float(), yy)


def test_lif_four_levels_stateless(ops):
    """The D=4 stateless variant of the E-SpikeFormer backbone (mmseg/models/utils/Qtrick.py:4-38)."""
    x = torch.linspace(-1, 6, 2001, device="cuda")
    y, _ = ops.lif(x, None, D=4, keep_v=False)
    assert torch.equal(y, torch.round(torch.clamp(x, 0, 4)) / 4)


def test_lif_seq_matches_chained_calls_and_golden(ops, golden):
    g = golden("lif_kat.npz")
    xs, v0 = T(g["seq_x"], grad=True), T(g["seq_v0"], grad=True)
    stats = ops.new_stats("cuda", T=xs.shape[0])
    y, vT = ops.lif_seq(xs, v0, stats=stats)
    assert np.array_equal(y.detach().cpu().numpy(), g["seq_y"]) and np.array_equal(vT.detach().cpu().numpy(), g["seq_vT"])
    ((y * T(g["seq_wy"])).sum() + (vT * T(g["seq_wv"])).sum()).backward()
    assert np.array_equal(xs.grad.cpu().numpy(), g["seq_gx"]) and np.array_equal(v0.grad.cpu().numpy(), g["seq_gv0"])
    assert ops.read_stats(stats)[:, 0].tolist() == (g["seq_y"] * 8).sum(1).astype(np.int64).tolist()
    # fused-over-T == T single-step launches
    v, ys = T(g["seq_v0"]), []
    for t in range(xs.shape[0]):
        yt, v = ops.lif(xs.detach()[t], v)
        ys.append(yt)
    assert torch.equal(torch.stack(ys), y.detach()) and torch.equal(v, vT.detach())
    y2, _ = ops.lif_seq(T(np.stack([g["kat_x"]] * 4)))                       # no initial membrane
    assert np.array_equal((y2 * 8).cpu().numpy(), g["kat_counts"])


def test_lif_seq_deep_temporal_loop_T8(ops):
    """BASELINE configs[3] (T = 8, membrane carried across the whole loop in registers): the fused-over-T kernel against eight
    chained single-step launches on a 16.8 M-element map (block3's largest neuron call at C4), with a non-zero initial membrane,
    forward and backward, bit for bit; firing counters per time step."""
    T_, n = 8, 2 * 256 * 32 * 32 * 8
    g = torch.Generator().manual_seed(8)
    xs = (torch.randn(T_, n, generator=g) * 1.5 + 0.4).cuda().requires_grad_(True)
    v0 = torch.rand(n, generator=g).cuda().requires_grad_(True)
    wy, wv = torch.randn(T_, n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    stats = ops.new_stats("cuda", T=T_)
    y, vT = ops.lif_seq(xs, v0, stats=stats)
    ((y * wy).sum() + (vT * wv).sum()).backward()
    gx, gv = xs.grad.clone(), v0.grad.clone()
    xs.grad = v0.grad = None
    v, ys = v0, []
    for t in range(T_):
        yt, v = ops.lif(xs[t], v)
        ys.append(yt)
    yc = torch.stack(ys)
    ((yc * wy).sum() + (v * wv).sum()).backward()
    assert torch.equal(y, yc) and torch.equal(vT, v)
    assert torch.equal(gx, xs.grad) and torch.equal(gv, v0.grad)
    counts = (y.detach() * 8).round().long()
    assert ops.read_stats(stats)[:, 0].tolist() == counts.sum(1).tolist()
    assert ops.read_stats(stats)[:, 1].tolist() == (counts != 0).sum(1).tolist()


def test_lif_full_size_properties(ops):
    """BASELINE full size (the largest single neuron call at C2, [4,2,256,256,256] = 134 M elements): properties that
    do not need the oracle -- output on the 9-point grid, y + v' == x exactly, idempotence of the stateless map."""
    n = 4 * 2 * 256 * 256 * 256
    x = torch.randn(n, device="cuda") * 3 + 1
    stats = ops.new_stats("cuda")
    y, v = ops.lif(x, None, stats=stats)
    c = y * 8
    assert torch.equal(c, torch.round(c)) and float(c.min()) == 0 and float(c.max()) == 8
    assert torch.equal(x - c, v)
    st = ops.read_stats(stats)
    assert int(st[0]) == int(c.double().sum()) and int(st[1]) == int((c != 0).sum())
    y2, _ = ops.lif(c, None, keep_v=False)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("bf16", [True, False])
@pytest.mark.parametrize("T_,B,C,L", [(4, 2, 256, 1024), (1, 1, 5, 4), (2, 3, 7, 36), (3, 1, 16, 260)])
def test_sum2_lif_is_the_two_neurons_on_the_two_sums(ops, spike_mode, bf16, T_, B, C, L):
    """Decoder key / value neurons from memory + level_embed (+ pos) in one launch (maskformer_head.py:535-540,
    transformer.py:626-629): bit-identical, forward and backward, to the stand-alone neuron on the materialised sums."""
    g = torch.Generator().manual_seed(T_ * 100 + L)
    x = (torch.randn(T_ * B, C, L, generator=g) * 2 + 1).cuda().requires_grad_(True)
    e = torch.randn(C, generator=g).cuda().requires_grad_(True)
    pos = torch.randn(B, C, L, generator=g).cuda()
    wk, wv = torch.randn(T_ * B, C, L, generator=g).cuda(), torch.randn(T_ * B, C, L, generator=g).cuda()
    spike_mode(bf16)
    yk, yv = ops.sum2_lif(x, e, pos, B)
    assert yk.data.dtype == (torch.bfloat16 if bf16 else torch.float32)
    yk, yv = yk.float(), yv.float()          # Spikes -> the reference's fp32 tensors (gradients flow through the handle)
    ((yk * wk).sum() + (yv * wv).sum()).backward()
    gx, ge = x.grad.clone(), e.grad.clone()
    x.grad = e.grad = None
    a = x.view(T_, B, C, L) + e.view(1, 1, C, 1)
    rv, _ = ops.lif(a, None, keep_v=False)
    rk, _ = ops.lif(a + pos, None, keep_v=False)
    ((rk.view_as(wk) * wk).sum() + (rv.view_as(wv) * wv).sum()).backward()
    assert torch.equal(yk, rk.view_as(yk)) and torch.equal(yv, rv.view_as(yv))
    assert torch.equal(gx, x.grad)
    assert torch.allclose(ge, e.grad, rtol=1e-5, atol=1e-5)        # a [C] reduction: summation order differs
    with pytest.raises(RuntimeError):
        ops.sum2_lif(x[:, :, :L - 1].contiguous(), e, pos[:, :, :L - 1].contiguous(), B)     # L % 4 != 0


# ----------------------------------------------------------------------------------------------- attention core
@pytest.mark.parametrize("TB,heads,d,Nq,Nk", [(4, 8, 32, 1024, 1024), (2, 8, 45, 256, 256), (3, 8, 8, 10, 30),
                                               (2, 8, 32, 100, 4096), (1, 1, 64, 7, 513)])
def test_sdsa_exact_for_spikes(ops, TB, heads, d, Nq, Nk):
    """With spike operands all partial sums are exact in fp32 (< 2^24 ulp), so the kernel must equal the reference's
    (q k^T) v association bit for bit -- the bound is asserted, not assumed."""
    g = torch.Generator().manual_seed(TB * 1000 + d)
    C = heads * d

    def spikes(n):
        return torch.clamp(torch.round(torch.randn(TB, C, n, generator=g) + 0.6), 0, 8) / 8
    q, k, v = spikes(Nq), spikes(Nk), spikes(Nk)
    scale = d ** -0.5

    def heads_view(t):
        return t.view(TB, heads, d, -1).transpose(2, 3)                        # [TB, h, N, d]
    kv = heads_view(k).transpose(-2, -1).double() @ heads_view(v).double()
    o64 = heads_view(q).double() @ kv
    assert float(kv.abs().max()) * 64 < 2 ** 24 and float(o64.abs().max()) * 512 < 2 ** 24
    ref = ((heads_view(q) @ (heads_view(k).transpose(-2, -1) @ heads_view(v))) * scale)   # sdtv2.py:335-336, fp32
    ref_dec = (heads_view(q) @ heads_view(k).transpose(-2, -1)) / 16.0 @ heads_view(v)    # transformer.py:262-272
    ref = ref.transpose(2, 3).reshape(TB, C, Nq)
    ref_dec = ref_dec.transpose(2, 3).reshape(TB, C, Nq)
    out = ops.sdsa(q.cuda(), k.cuda(), v.cuda(), heads, scale)
    assert torch.equal(out.cpu(), ref)
    out = ops.sdsa(q.cuda(), k.cuda(), v.cuda(), heads, 1.0 / 16.0)
    assert torch.equal(out.cpu(), ref_dec)


def test_sdsa_backward(ops):
    TB, heads, d, Nq, Nk = 2, 8, 32, 100, 300
    g = torch.Generator().manual_seed(1)
    C = heads * d
    q, k, v = (torch.randn(TB, C, n, generator=g).double().requires_grad_(True) for n in (Nq, Nk, Nk))
    go = torch.randn(TB, C, Nq, generator=g).double()

    def hv(t):
        return t.view(TB, heads, d, -1).transpose(2, 3)
    o = (hv(q) @ (hv(k).transpose(-2, -1) @ hv(v)) * 0.37).transpose(2, 3).reshape(TB, C, Nq)
    o.backward(go)
    qc, kc, vc = (t.detach().float().cuda().requires_grad_(True) for t in (q, k, v))
    oc = ops.sdsa(qc, kc, vc, heads, 0.37)
    oc.backward(go.float().cuda())
    for a, b, name in ((oc, o, "o"), (qc.grad, q.grad, "gq"), (kc.grad, k.grad, "gk"), (vc.grad, v.grad, "gv")):
        err = (a.detach().cpu().double() - b.detach()).abs().max().item()
        assert err <= 2e-5 * b.abs().max().item(), name          # fp32 accumulation over <= 300 terms


# ----------------------------------------------------------------------------------------------- DCNv3
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_dcnv3_vs_reference_vectors(ops, golden, tag):
    """Tolerances of test_oracle_golden.py::test_dcnv3_core_vs_reference (the reference's grid arithmetic is not
    bit-reproducible by a pixel-coordinate kernel)."""
    g = golden("dcnv3_core.npz")
    N, H, W, G, Cg, K, s, p, d = (int(v) for v in g[f"{tag}_geom"])
    x, off, m = (T(g[f"{tag}_{k}"], grad=True) for k in ("x", "offset", "mask"))
    y = ops.dcnv3_core(x, off, m, K, K, s, s, p, p, d, d, G, Cg, float(g[f"{tag}_offset_scale"]))
    y.backward(T(g[f"{tag}_gy"]))

    def close(a, key, rel):
        b = g[f"{tag}_{key}"]
        return np.abs(a.detach().cpu().numpy() - b).max() <= rel * np.abs(b).max()
    assert close(y, "y", 2e-4) and close(x.grad, "gx", 2e-4) and close(m.grad, "gmask", 2e-4)
    assert close(off.grad, "goffset", 2e-3)


def test_dcnv3_vs_oracle_hot_path_geometry(ops, so):
    """C2 geometry per (t,b): 32x32, G=32, Cg=8, K=3, spikes as mask -- against the oracle's gather restatement."""
    g = torch.Generator().manual_seed(2)
    N, H, W, G, Cg = 2, 32, 32, 32, 8
    x = torch.randn(N, H, W, G * Cg, generator=g)
    off = torch.randn(N, H, W, G * 18, generator=g) * 2
    m = torch.clamp(torch.round(torch.randn(N, H, W, G * 9, generator=g) + 1), 0, 8) / 8
    gy = torch.randn(N, H, W, G * Cg, generator=g)
    xo, oo, mo = (t.clone().requires_grad_(True) for t in (x, off, m))
    so.dcnv3_core(xo, oo, mo, G, Cg).backward(gy)
    xc, oc, mc = (t.cuda().requires_grad_(True) for t in (x, off, m))
    y = ops.dcnv3_core(xc, oc, mc, 3, 3, 1, 1, 1, 1, 1, 1, G, Cg, 1.0)
    y.backward(gy.cuda())
    yo = so.dcnv3_core(x, off, m, G, Cg)
    for a, b, tol, name in ((y, yo, 1e-5, "y"), (xc.grad, xo.grad, 1e-5, "gx"), (mc.grad, mo.grad, 1e-5, "gm"),
                            (oc.grad, oo.grad, 1e-4, "goff")):
        assert (a.detach().cpu() - b).abs().max().item() <= tol * b.abs().max().item(), name


def test_dcnv3_backward_fixed_point_accumulator(ops):
    """grad_input is accumulated in LDS in 64-bit fixed point scaled per (n, group) slice: the result must be (1) exactly
    homogeneous under power-of-two scalings of grad_output over the whole fp32 range that gradients live in, (2) bit-wise
    reproducible (order-independent, unlike an fp32 atomic), (3) independent of what grad_input's buffer held before, and
    (4) equal to the large-map global-atomic path on a map that does not fit in LDS."""
    from spike2former_amd._lib import lib
    g = torch.Generator().manual_seed(5)
    N, H, W, G, Cg = 2, 32, 32, 32, 8
    x = torch.randn(N, H, W, G * Cg, generator=g).cuda()
    off = (torch.randn(N, H, W, G * 18, generator=g) * 2).cuda()
    m = torch.rand(N, H, W, G * 9, generator=g).cuda()
    gy = torch.randn(N, H, W, G * Cg, generator=g).cuda()

    def bwd(go, fill):
        gx = torch.full_like(x, fill)
        goff, gm = torch.empty_like(off), torch.empty_like(m)
        rc = lib.s2f_dcnv3_bwd(x.data_ptr(), off.data_ptr(), m.data_ptr(), go.data_ptr(), gx.data_ptr(), goff.data_ptr(),
                               gm.data_ptr(), N, H, W, G, Cg, 3, 3, 1, 1, 1, 1, 1, 1, 1.0, None)
        assert rc == 0
        torch.cuda.synchronize()
        return gx
    base = bwd(gy, 0.0)
    assert torch.equal(base, bwd(gy, 123.0))
    for e in (-60, -30, 30, 60):
        assert torch.equal(bwd(gy * 2.0 ** e, float("nan")) * 2.0 ** -e, base), e
    assert torch.equal(bwd(torch.zeros_like(gy), 7.0), torch.zeros_like(x))
    # one slice with a huge outlier: the other elements of that slice keep >= 2e-6 relative accuracy
    go2 = gy.clone(); go2[0, 0, 0, 0] = 1.0e4
    ref = bwd(gy, 0.0); out = bwd(go2, 0.0)
    far = torch.ones_like(x, dtype=torch.bool); far[0, :4, :4, :Cg] = False
    assert (out[far] - ref[far]).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # large map (64x64x32 channels per group = 1.5 MiB per slice): global-atomic path, same numbers as autograd of forward
    N2, H2, G2, Cg2 = 1, 64, 2, 32
    x2 = torch.randn(N2, H2, H2, G2 * Cg2, device="cuda", requires_grad=True)
    o2 = (torch.randn(N2, H2, H2, G2 * 18, device="cuda")).requires_grad_(True)
    m2 = torch.rand(N2, H2, H2, G2 * 9, device="cuda", requires_grad=True)
    y2 = ops.dcnv3_core(x2, o2, m2, 3, 3, 1, 1, 1, 1, 1, 1, G2, Cg2, 1.0)
    g2 = torch.randn_like(y2)
    y2.backward(g2)
    # linearity in x: <dcn(x'), g2> = <x', gx> for any x'
    xp = torch.randn_like(x2)
    lhs = (ops.dcnv3_core(xp, o2.detach(), m2.detach(), 3, 3, 1, 1, 1, 1, 1, 1, G2, Cg2, 1.0) * g2).double().sum()
    rhs = (xp * x2.grad).double().sum()
    assert abs(lhs.item() - rhs.item()) <= 1e-4 * (xp.abs() * x2.grad.abs()).double().sum().item()


def test_dcnv3_zero_offset_is_a_modulated_3x3_average(ops):
    """Size-independent property: zero offsets + constant mask 1 turn DCNv3 into a 3x3 box filter with zero padding."""
    N, H, W, G, Cg = 8, 32, 32, 32, 8
    x = torch.randn(N, H, W, G * Cg, device="cuda")
    y = ops.dcnv3_core(x, torch.zeros(N, H, W, G * 18, device="cuda"), torch.ones(N, H, W, G * 9, device="cuda"),
                       3, 3, 1, 1, 1, 1, 1, 1, G, Cg, 1.0)
    ref = torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=True) * 9
    assert torch.allclose(y, ref.permute(0, 2, 3, 1), atol=1e-5, rtol=1e-5)


def test_errors_are_raised_not_swallowed(ops):
    from spike2former_amd._lib import S2FError
    with pytest.raises(S2FError, match="head dim|d <="):
        ops.sdsa(torch.zeros(1, 130, 4, device="cuda"), torch.zeros(1, 130, 4, device="cuda"),
                 torch.zeros(1, 130, 4, device="cuda"), 2, 1.0)
    with pytest.raises(RuntimeError, match="fp32"):
        ops.lif(torch.zeros(8, device="cuda", dtype=torch.float16))


# ----------------------------------------------------------------------------------------------- fused BN (+bias, +residual, +LIF)
@pytest.mark.parametrize("N,C,L,training,res,lif", [(8, 32, 1024, True, False, True), (4, 20, 64, True, True, True),
                                                     (2, 7, 36, True, True, False), (3, 16, 256, False, True, True),
                                                     (8, 256, 4096, True, False, True),
                                                     # single-pass kernels (s2f_bn_single_pass): full / ragged wave counts
                                                     (8, 256, 1024, True, False, True), (8, 64, 1024, True, True, True),
                                                     (2, 128, 512, True, True, False), (3, 64, 768, True, False, True),
                                                     # row-walking kernels with tiles that straddle two channel rows / a ragged
                                                     # last tile (L % 256 != 0; C5's 1050- and 4200-pixel maps), train and eval
                                                     (3, 5, 260, True, True, True), (2, 7, 1052, True, False, True),
                                                     (1, 3, 300, True, True, False), (2, 6, 4200, False, True, True),
                                                     (8, 3, 16800, True, False, True),
                                                     # rows of L % 4 != 0 elements (element-wise ANYL kernels): the tiny
                                                     # configuration's 10-query decoder rows; odd totals; eval mode
                                                     (8, 64, 10, True, False, True), (8, 256, 10, True, True, True),
                                                     (3, 5, 7, True, True, True), (2, 3, 5, True, False, False),
                                                     (1, 9, 333, True, True, True), (4, 6, 10, False, True, True),
                                                     (2, 4, 1027, True, False, True),
                                                     # short rows, one wavefront per channel (s2f_bn_mask_words): the decoder's
                                                     # 100-token maps; one / several / all eight rounds of 64 groups
                                                     (8, 256, 100, True, False, True), (8, 64, 100, True, True, True),
                                                     (8, 2048, 100, True, False, False), (2, 32, 1000, True, True, True),
                                                     (3, 40, 36, True, True, True), (2, 33, 4, True, False, True),
                                                     # 33 .. 64 tiles per channel: the sixteen-wavefront single-pass form (C3's 64 x 32
                                                     # stage, C4's T = 8 batch)
                                                     (8, 64, 2048, True, True, True), (16, 96, 1024, True, False, True),
                                                     (5, 64, 3072, True, True, False),
                                                     # rows that are not whole tiles, 2 049 .. 20 480 elements per channel: the per-channel
                                                     # single-pass form on 2 .. 16 wavefronts (C5's 50 x 84 maps: [4, C, 4 200])
                                                     (4, 64, 4200, True, True, True), (4, 32, 4200, True, False, True),
                                                     (2, 40, 1052, True, True, True), (5, 33, 4092, True, True, False),
                                                     (3, 64, 1000, True, False, True)])
@pytest.mark.parametrize("bf16", [True, False])
def test_bn_act_vs_oracle(so, spike_mode, bf16, N, C, L, training, res, lif):
    """The fused kernels against the oracle's chain  BN(z + b) [+ r] -> Q_IFNode  on CPU (F.batch_norm + lif_step).
    Pre-activation: rtol 2e-5 (different but equally valid fp32 evaluation orders); spikes: at most 1e-4 of the
    elements may differ, each by exactly one level; gradients: 1e-4 of their max."""
    import torch.nn.functional as F

    from spike2former_amd import ops
    g = torch.Generator().manual_seed(N * 131 + C)
    z = torch.randn(N, C, L, generator=g) * 2 + 0.5
    b = torch.randn(C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3 + 0.5
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    r = torch.randn(N, C, L, generator=g) if res else None
    gu_w, gy_w = torch.randn(N, C, L, generator=g), torch.randn(N, C, L, generator=g)
    # oracle
    zo, bo, go, beo = (t.clone().requires_grad_(True) for t in (z, b, gamma, beta))
    ro = r.clone().requires_grad_(True) if res else None
    rmo, rvo = rm.clone(), rv.clone()
    u = F.batch_norm(zo + bo.view(1, -1, 1), rmo, rvo, go, beo, training, 0.1, 1e-5)
    if res:
        u = u + ro
    loss = (u * gu_w).sum()
    if lif:
        y, _, _ = so.lif_step(u, None)
        loss = loss + (y * gy_w).sum()
    loss.backward()
    # product
    zc, bc, gc, bec = (t.clone().cuda().requires_grad_(True) for t in (z, b, gamma, beta))
    rc = r.clone().cuda().requires_grad_(True) if res else None
    rmc, rvc = rm.clone().cuda(), rv.clone().cuda()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    fstats = ops.new_stats("cuda") if lif else None
    spike_mode(bf16)
    uu, yy, _, border = ops.bn_act(zc, bc, gc, bec, rmc, rvc, nbt if training else None, training, 0.1, 1e-5, residual=rc,
                                   lif=lif, want_pre=True, stats=fstats, want_border=True)
    if lif:
        assert yy.data.dtype == (torch.bfloat16 if (bf16 and (N * C * L) % 4 == 0) else torch.float32)
        yy = yy.float()
    # BNAndPadLayer's padding value BN(0), from the running statistics AFTER this call's update (sdtv2.py:68-78)
    bref = beta - rmo * gamma / torch.sqrt(rvo + 1e-5)
    assert (border.cpu() - bref).abs().max().item() <= 1e-5 * max(bref.abs().max().item(), 1.0)
    if lif:
        cnt = torch.round(yy.detach() * 8).long()
        assert ops.read_stats(fstats).tolist() == [int(cnt.sum()), int((cnt != 0).sum())]
    lossc = (uu * gu_w.cuda()).sum()
    if lif:
        lossc = lossc + (yy * gy_w.cuda()).sum()
    lossc.backward()

    def close(a, bb, tol):
        return (a.detach().cpu() - bb.detach()).abs().max().item() <= tol * max(bb.abs().max().item(), 1e-6)
    assert close(uu, u, 2e-5)
    if lif:
        d = (yy.detach().cpu() - y.detach()) * 8
        assert d.abs().max().item() <= 1 and (d != 0).float().mean().item() <= 1e-4
    if training:
        assert close(rmc, rmo, 1e-5) and close(rvc, rvo, 1e-5) and int(nbt) == 1
        assert bc.grad is None or bc.grad.abs().max().item() == 0.0   # train-mode BN removes the conv bias gradient exactly
    else:
        assert close(bc.grad, bo.grad, 1e-4)
    flips = lif and (d != 0).any().item()
    tol = 5e-3 if flips else 1e-4
    assert close(zc.grad, zo.grad, tol) and close(gc.grad, go.grad, tol) and close(bec.grad, beo.grad, tol)
    if res:
        assert close(rc.grad, ro.grad, tol)


def test_bn_act_stateful_lif_matches_unfused(ops):
    """Fused BN+neuron with a carried membrane == BN-only kernel followed by the stand-alone neuron kernel, bit for bit."""
    g = torch.Generator().manual_seed(9)
    N, C, L = 4, 24, 512
    z = (torch.randn(N, C, L, generator=g) * 2).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    v = torch.randn(N, C, L, generator=g).cuda()
    stats_a = ops.new_stats("cuda"); stats_b = ops.new_stats("cuda")
    u, y, v_out = ops.bn_act(z, None, gamma, beta, rm, rv, None, False, 0.1, 1e-5, lif=True, want_pre=True, v_in=v,
                             keep_v=True, stats=stats_a)
    u2, _, _ = ops.bn_act(z, None, gamma, beta, rm, rv, None, False, 0.1, 1e-5, lif=False, want_pre=True)
    y2, v2 = ops.lif(u2, v, stats=stats_b)
    y = y.float()
    assert torch.equal(u, u2) and torch.equal(y, y2) and torch.equal(v_out, v2) and torch.equal(ops.read_stats(stats_a), ops.read_stats(stats_b))


def test_mask_einsum_matrix_core_path_vs_fp64(ops):
    """out[b] = scale * sum_t E[t,b] @ MF[t,b] (maskformer_head.py:582-583 with the t-mean folded in) and its gradients.
    With E exact in bf16 (alpha * spikes) the forward and d(MF) run as split GEMMs on the matrix cores: exact products,
    fp32 accumulation -> same error class as an fp32 GEMM (<= 2e-6 * sum|e||mf|); ragged Q / HW exercise the padding."""
    g = torch.Generator().manual_seed(11)
    for (T, B, Q, C, HW) in ((4, 2, 100, 64, 1024), (3, 1, 37, 40, 260)):
        e = (torch.randint(0, 9, (T, B, Q, C), generator=g).float() / 2).cuda().requires_grad_(True)     # multiples of 1/2
        mf = torch.randn(T, B, C, HW, generator=g).cuda().requires_grad_(True)
        go = torch.randn(B, Q, HW, generator=g).cuda()
        out = ops.mask_einsum(e, mf, 1.0 / T, e_exact=True)
        out.backward(go)
        e64, mf64 = e.detach().double(), mf.detach().double()
        ref = torch.einsum("tbqc,tbcn->bqn", e64, mf64) / T
        bound = torch.einsum("tbqc,tbcn->bqn", e64.abs(), mf64.abs()) / T
        assert ((out.detach().double() - ref).abs() <= 2e-6 * bound + 1e-12).all()
        gmf_ref = torch.einsum("tbqc,bqn->tbcn", e64, go.double()) / T
        gmf_bound = torch.einsum("tbqc,bqn->tbcn", e64.abs(), go.double().abs()) / T
        assert ((mf.grad.double() - gmf_ref).abs() <= 2e-6 * gmf_bound + 1e-12).all()
        ge_ref = torch.einsum("bqn,tbcn->tbqc", go.double(), mf64) / T
        ge_bound = torch.einsum("bqn,tbcn->tbqc", go.double().abs(), mf64.abs()) / T
        assert ((e.grad.double() - ge_ref).abs() <= 4e-6 * ge_bound + 1e-12).all()      # 6-pass split GEMM: fp32 class
        # library path (E not flagged exact) gives the same numbers to fp32 round-off
        e2, mf2 = e.detach().clone().requires_grad_(True), mf.detach().clone().requires_grad_(True)
        out2 = ops.mask_einsum(e2, mf2, 1.0 / T)
        out2.backward(go)
        assert (out2 - out).abs().max().item() <= 1e-4 * ref.abs().max().item()
        assert (mf2.grad - mf.grad).abs().max().item() <= 1e-4 * gmf_ref.abs().max().item()


def test_scale_affine_matches_the_vector_ops(ops):
    """Layer scale folded into a BatchNorm affine pair (detr_layers.py:331-337): (gamma*s, beta*s) and all three gradients."""
    g = torch.Generator().manual_seed(4)
    ga, be, sc = (torch.randn(300, generator=g).cuda().requires_grad_(True) for _ in range(3))
    w, b = ops.scale_affine(ga, be, sc)
    cw, cb = torch.randn(300, generator=g).cuda(), torch.randn(300, generator=g).cuda()
    ((w * cw).sum() + (b * cb).sum()).backward()
    got = [t.grad.clone() for t in (ga, be, sc)]
    for t in (ga, be, sc):
        t.grad = None
    ((ga * sc * cw).sum() + (be * sc * cb).sum()).backward()
    assert torch.equal(w, ga * sc) and torch.equal(b, be * sc)
    assert torch.equal(got[0], ga.grad) and torch.equal(got[1], be.grad) and torch.allclose(got[2], sc.grad, rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------------------------------------- depthwise stencils
@pytest.mark.parametrize("N,C,H,W,K,pad,border", [(2, 8, 32, 32, 3, 1, True), (2, 6, 37, 45, 7, 3, False),
                                                   (1, 5, 4, 4, 5, 2, False), (3, 16, 64, 64, 5, 2, False),
                                                   (2, 4, 9, 7, 3, 1, True), (1, 3, 70, 33, 7, 3, False),
                                                   (2, 4, 10, 10, 3, 0, False),
                                                   # wide-map form of the 3x3 (W % 4 == 0, W >= 128, H >= 32, pad 1): full and
                                                   # partial 64 x 32 tiles, with / without border; other sizes on wide maps
                                                   (2, 3, 96, 128, 3, 1, True), (1, 2, 33, 132, 3, 1, False),
                                                   (1, 2, 40, 256, 3, 1, True), (1, 4, 64, 128, 7, 3, False),
                                                   (1, 2, 40, 136, 5, 2, False)])
def test_dwconv_vs_aten_cpu(ops, N, C, H, W, K, pad, border):
    """fp32, K*K <= 49 terms per output: rtol 2e-6 forward / input gradient; the weight gradient sums N*H*W terms in a
    different order than ATen: 2e-5.  `border` reproduces conv(pad-with-constant(x), padding=0) of BNAndPadLayer."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(H * 7 + K)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, K, K, generator=g)
    b = torch.randn(C, generator=g) if border else None
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if border:
        full = b.view(1, C, 1, 1).expand(N, C, H + 2 * pad, W + 2 * pad).clone()
        full[:, :, pad:-pad, pad:-pad] = xo
        yo = F.conv2d(full, wo, None, 1, 0, 1, C)
    else:
        yo = F.conv2d(xo, wo, None, 1, pad, 1, C)
    gy = torch.randn(yo.shape, generator=g)
    yo.backward(gy)
    xc, wc = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    yc = ops.dwconv(xc, wc, pad, None if b is None else b.cuda())
    yc.backward(gy.cuda())

    def close(a, bb, tol):
        return (a.detach().cpu() - bb.detach()).abs().max().item() <= tol * bb.abs().max().item()
    assert yc.shape == yo.shape and close(yc, yo, 2e-6) and close(xc.grad, xo.grad, 2e-6) and close(wc.grad, wo.grad, 2e-5)


# ----------------------------------------------------------------------------------------------- spike GEMM (bf16 MFMA)
def test_mfma_fragment_layout_identity_check(ops):
    """A = I with an ASYMMETRIC B catches row/column swaps in the MFMA operand / accumulator mapping."""
    M = K = 64
    N = 128
    w = torch.eye(M, K, device="cuda")
    x = (torch.arange(K * N, device="cuda").reshape(1, K, N) % 97).float() / 8          # spikes-like grid, asymmetric
    y = ops.spike_gemm(x, w, None)
    assert torch.equal(y, x)


@pytest.mark.parametrize("B,M,K,N", [(2, 256, 256, 1024), (1, 32, 288, 4096), (3, 360, 360, 1024), (2, 151, 100, 100),
                                     (1, 1024, 256, 256), (2, 64, 1152, 512), (1, 576, 256, 36)])
def test_spike_gemm_matches_fp64(ops, B, M, K, N):
    """Spike activations x three-term bf16 weights, fp32 accumulation: error vs an fp64 reference must be that of an fp32
    GEMM (<= 2e-6 of sum|w||x|; measured fp32 rocBLAS on the same data is reported for comparison), ragged M/K/N included."""
    g = torch.Generator().manual_seed(M + K)
    w = torch.randn(M, K, generator=g) * K ** -0.5
    x = torch.clamp(torch.round(torch.randn(B, K, N, generator=g) + 0.7), 0, 8) / 8
    bias = torch.randn(M, generator=g)
    ref = torch.matmul(w.double(), x.double()) + bias.double().view(1, -1, 1)
    scale = torch.matmul(w.abs().double(), x.abs().double()).max().item()
    wc, xc = w.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
    y = ops.spike_gemm(xc, wc, bias.cuda())
    err = (y.detach().cpu().double() - ref).abs().max().item()
    err32 = (torch.matmul(w.cuda(), x.cuda()).cpu().double() + bias.double().view(1, -1, 1) - ref).abs().max().item()
    assert err <= 2e-6 * scale, (err, err32, scale)
    # backward (rocBLAS fp32) is the plain GEMM adjoint
    gy = torch.randn(B, M, N, generator=g)
    y.backward(gy.cuda())
    gx_ref = torch.matmul(w.t().double(), gy.double())
    gw_ref = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    assert (xc.grad.cpu().double() - gx_ref).abs().max().item() <= 1e-5 * gx_ref.abs().max().item()
    assert (wc.grad.cpu().double() - gw_ref).abs().max().item() <= 1e-5 * gw_ref.abs().max().item()


def test_spike_gemm_terms_and_resplit(ops):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(128, 256, generator=g).cuda()
    x = (torch.randint(0, 9, (1, 256, 512), generator=g).float() / 8).cuda()
    ref = torch.matmul(w.double(), x.double())
    errs = []
    for t in (1, 2, 3):
        ops.SPIKE_GEMM_TERMS = t
        errs.append((ops.spike_gemm(x, w).double() - ref).abs().max().item())
    ops.SPIKE_GEMM_TERMS = 3
    assert errs[0] > 100 * errs[1] and errs[1] > 3 * errs[2] and errs[2] < 3e-5     # 8 / 16 / 24 mantissa bits (x3 sits on the fp32 accumulation floor)
    w.mul_(2.0)                                                                           # in-place update -> re-split
    assert (ops.spike_gemm(x, w).double() - 2 * ref).abs().max().item() < 6e-5
    # a freed weight's address is handed to the next allocation of the same size: the cached split must not outlive its owner
    addr = w.data_ptr()
    del w
    keep = []
    for _ in range(16):                                   # the caching allocator recycles the block sooner or later
        w2 = torch.randn(128, 256, generator=g).cuda()
        if w2.data_ptr() == addr:
            break
        keep.append(w2)
    if w2.data_ptr() != addr:
        pytest.skip("allocator did not recycle the block (premise of the address-reuse check)")
    w2.mul_(1.0)                                          # same version counter as the freed weight had
    assert w2._version == 1
    assert (ops.spike_gemm(x, w2).double() - torch.matmul(w2.double(), x.double())).abs().max().item() < 6e-5


# ----------------------------------------------------------------------------------------------- 2x bilinear up-sampling
@pytest.mark.parametrize("N,C,h,w", [(2, 3, 4, 4), (1, 5, 7, 6), (2, 16, 32, 32), (1, 2, 1, 2), (1, 2, 2, 4), (2, 3, 6, 8),
                                     (1, 4, 64, 64), (1, 2, 5, 8), (1, 2, 6, 10)])
def test_upsample2x_matches_interpolate(ops, N, C, h, w):
    """Against F.interpolate on CPU (forward and adjoint through autograd): 2e-6 of the max -- ATen's CPU kernel
    associates the four weighted taps differently, so the last bit may differ."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(h * 10 + w)
    x = torch.randn(N, C, h, w, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = F.interpolate(xo, size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
    gy = torch.randn(yo.shape, generator=g)
    yo.backward(gy)
    xc = x.cuda().requires_grad_(True)
    yc = ops.upsample_bilinear(xc, (2 * h, 2 * w))
    yc.backward(gy.cuda())
    assert (yc.detach().cpu() - yo.detach()).abs().max().item() <= 2e-6 * yo.abs().max().item()
    assert (xc.grad.cpu() - xo.grad).abs().max().item() <= 2e-6 * xo.grad.abs().max().item()


@pytest.mark.parametrize("B,M,K,L", [(2, 256, 256, 1024), (1, 32, 288, 4096), (3, 360, 360, 100), (2, 151, 100, 36),
                                     (8, 128, 64, 4096), (1, 576, 256, 1024),
                                     # narrow output tiles: 32 rows (M <= 32), 64 rows (M <= 64), ragged M / K / L
                                     (2, 64, 96, 512), (2, 20, 64, 256), (3, 33, 130, 100), (2, 32, 1152, 4096)])
def test_spike_gemm_weight_gradient_matches_fp64(ops, B, M, K, L):
    """dW = sum_b dY[b] X[b]^T on the bf16 matrix cores (dY split hi+mid+lo, X spikes): error vs fp64 <= 3e-6 of
    sum|dY||X| -- fp32-GEMM class (split-K partials are combined with fp32 atomics)."""
    from spike2former_amd._lib import lib
    g = torch.Generator().manual_seed(M * 3 + L)
    gy = torch.randn(B, M, L, generator=g) * torch.rand(B, M, 1, generator=g) * 10
    x = torch.clamp(torch.round(torch.randn(B, K, L, generator=g) + 0.7), 0, 8) / 8
    ref = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    scale = torch.einsum("bml,bkl->mk", gy.abs().double(), x.abs().double()).max().item()
    gyc, xc = gy.cuda(), x.cuda()
    out = torch.full((M, K), float("nan"), device="cuda")
    assert lib.s2f_spike_gemm_dw(gyc.data_ptr(), xc.data_ptr(), out.data_ptr(), B, M, K, L, 0, 1, None) == 0
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 3e-6 * scale, (err, scale)


@pytest.mark.parametrize("N,C,M,H,W,k,s,p,spike", [(2, 16, 64, 12, 16, 3, 1, 1, True), (2, 64, 16, 12, 16, 3, 1, 1, True),
                                                     (1, 8, 24, 9, 8, 3, 2, 1, True), (2, 3, 16, 16, 16, 7, 2, 3, False),
                                                     (1, 128, 32, 8, 8, 3, 1, 1, True),
                                                     # implicit-GEMM 3x3 (W a power of two): ragged M / C / H, several tiles
                                                     (2, 32, 72, 19, 32, 3, 1, 1, True), (3, 160, 40, 5, 64, 3, 1, 1, True),
                                                     (1, 64, 200, 33, 4, 3, 1, 1, True),
                                                     # widths that are not a power of two (C5's maps are 672 / 336 / 168 / 84 wide): the
                                                     # implicit weight-gradient kernels divide instead of shifting (round 5)
                                                     (2, 32, 72, 11, 84, 3, 1, 1, True), (1, 64, 40, 7, 168, 3, 1, 1, True),
                                                     (2, 96, 32, 9, 12, 3, 1, 1, True)])
def test_conv_dense_vs_aten_cpu(ops, N, C, M, H, W, k, s, p, spike):
    """The GEMM lowerings of the dense convolution (incl. the transposed-convolution form of dX used when M < C) against
    F.conv2d on CPU: 2e-5 of the max for outputs and all three gradients."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C * 7 + M)
    x = (torch.randint(0, 9, (N, C, H, W), generator=g).float() / 8) if spike else torch.randn(N, C, H, W, generator=g)
    w = torch.randn(M, C, k, k, generator=g) * (C * k * k) ** -0.5
    b = torch.randn(M, generator=g)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = F.conv2d(xo, wo, bo, s, p)
    gy = torch.randn(yo.shape, generator=g)
    yo.backward(gy)
    for implicit in (False, True):            # im2col + GEMM lowering, and the implicit-GEMM kernels where they apply
        ops.CONV3X3_IMPLICIT, min_pixels, ops.CONV3X3_IMPLICIT_MIN_PIXELS = implicit, ops.CONV3X3_IMPLICIT_MIN_PIXELS, 0
        ops.CONV3X3_DX_IMPLICIT = implicit
        try:
            xc, wc, bc = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
            yc = ops.conv_dense(xc, wc, bc, s, p, spike)
            yc.backward(gy.cuda())
        finally:
            ops.CONV3X3_IMPLICIT, ops.CONV3X3_IMPLICIT_MIN_PIXELS, ops.CONV3X3_DX_IMPLICIT = True, min_pixels, True
        for a, r in ((yc, yo), (xc.grad, xo.grad), (wc.grad, wo.grad), (bc.grad, bo.grad)):
            assert (a.detach().cpu() - r.detach()).abs().max().item() <= 2e-5 * r.abs().max().item(), implicit


def test_resplit_all_redoes_every_cached_split(ops):
    """ops.resplit_all (s2f_split_bf16x3_multi: all weights in one launch) writes, for each of the three source layouts, exactly
    the bf16 terms the per-weight split wrote -- checked after changing the weights THROUGH `.data` (no version bump), i.e.
    only the multi-launch can have produced the new terms."""
    g = torch.Generator().manual_seed(3)
    w2d = torch.randn(100, 72, generator=g).cuda()
    wc = torch.randn(40, 32, 3, 3, generator=g).cuda()
    a, b, c = ops.split_weight(w2d), ops.split_weight_conv3(wc), ops.split_weight_tconv3(wc)
    assert ops.resplit_all(w2d.device) >= 3
    torch.cuda.synchronize()
    a0, b0, c0 = a.clone(), b.clone(), c.clone()
    assert torch.equal(a0, a) and torch.equal(b0, b) and torch.equal(c0, c)          # same weights -> same terms
    w2d.data.mul_(1.7); wc.data.mul_(-0.3)
    assert ops.resplit_all(w2d.device) >= 3
    ops._SPLIT_CACHE.clear()
    a1, b1, c1 = ops.split_weight(w2d), ops.split_weight_conv3(wc), ops.split_weight_tconv3(wc)      # fresh per-weight splits
    assert not torch.equal(a0, a) and torch.equal(a1, a) and torch.equal(b1, b) and torch.equal(c1, c)
    # hi + mid + lo reproduces the fp32 weight to 2^-24
    terms = (a.view(torch.bfloat16).float().sum(0))[:100, :72]
    assert (terms - w2d).abs().max().item() <= 2.0 ** -22 * w2d.abs().max().item()


@pytest.mark.parametrize("TB,heads,d,N,packed", [(2, 8, 32, 1024, True), (1, 8, 45, 256, False), (3, 4, 9, 512, True),
                                                 (2, 2, 64, 256, False)])
def test_fused_attention_neuron_kernel_is_core_plus_neuron(ops, TB, heads, d, N, packed):
    """s2f_sdsa_lif_fwd_bf16 (kv on the matrix cores, q kv and the neuron in one kernel, bf16 spikes out) and its backward
    (straight-through estimator inside the loaders) against the separate fp32 kernels: attention core -> Q_IFNode.
    Spike operands make every forward sum exact: spikes and firing counters must be IDENTICAL; input gradients to round-off."""
    from spike2former_amd.neuron import Q_IFNode
    g = torch.Generator().manual_seed(TB * 1000 + d)
    C = heads * d
    scale = 0.125
    ycat = (torch.randint(0, 9, (TB, 3 * C, N), generator=g).float() / 8).cuda()
    ycat = ycat * (torch.rand(TB, 3 * C, 1, generator=g).cuda() < 0.6)          # sparse like real spike maps
    wgt = torch.randn(TB, C, N, generator=g).cuda()
    # reference: fp32 tensors through the unfused ops
    ref_in = ycat.clone().requires_grad_(True)
    q, k, v = ref_in[:, :C], ref_in[:, C:2 * C], ref_in[:, 2 * C:]
    lif_a = Q_IFNode(); lif_a.keep_membrane = False; lif_a.stats = ops.new_stats("cuda")
    o = ops.sdsa(q.contiguous(), k.contiguous(), v.contiguous(), heads, scale)
    ya = lif_a(o)
    (ya * wgt).sum().backward()
    # product path: bf16 Spikes with autograd handles
    src = ycat.clone().requires_grad_(True)
    yb16, _ = ops.lif(src * 8.0, None, keep_v=False, spikes=True)          # Q_IFNode(8 * (k / 8)) = k / 8: reproduces the map
    assert yb16.data.dtype == torch.bfloat16 and torch.equal(yb16.float().detach(), ycat)
    lif_b = Q_IFNode(); lif_b.keep_membrane = False; lif_b.stats = ops.new_stats("cuda")
    if packed:
        yb = ops.sdsa_packed(yb16, heads, scale, lif=lif_b)
    else:
        # three separate spike maps, each with its own handle
        hs = [ops.lif((ycat[:, i * C:(i + 1) * C].clone().requires_grad_(True)) * 8.0, None, keep_v=False, spikes=True)[0] for i in range(3)]
        yb = ops.sdsa(hs[0], hs[1], hs[2], heads, scale, lif=lif_b)
    assert isinstance(yb, ops.Spikes) and yb.data.dtype == torch.bfloat16
    assert torch.equal(yb.data.float(), ya.detach())
    assert torch.equal(ops.read_stats(lif_a.stats), ops.read_stats(lif_b.stats))
    if packed:
        (yb.float() * wgt).sum().backward()
        # d Q_IFNode(8 src) / d src = 8 * (in-range / 8) = 1 on [0, 1]; the gradient itself is a general fp32 tensor whose
        # k^T v-sized partial sums are combined with atomics: equal to fp32 round-off, not bit for bit
        err = (src.grad - ref_in.grad).abs().max().item()
        assert err <= 1e-5 * ref_in.grad.abs().max().item(), err


@pytest.mark.parametrize("N,C,H,W,K", [(2, 8, 16, 16, 7), (1, 4, 128, 128, 3), (2, 3, 12, 20, 5)])
def test_dwconv_on_bf16_spikes_is_the_fp32_stencil(ops, N, C, H, W, K):
    """The depthwise kernels reading a bf16 spike map (SepConv.dwconv, output_convs, DCNv3.dw_conv follow a neuron) give
    bit for bit the result of the fp32 input: forward, input gradient (through the autograd handle), weight gradient."""
    g = torch.Generator().manual_seed(N * 7 + K)
    src = torch.randint(0, 9, (N, C, H, W), generator=g).float().cuda()
    w = torch.randn(C, 1, K, K, generator=g).cuda()
    gy = torch.randn(N, C, H, W, generator=g).cuda()
    xa = (src / 8).requires_grad_(True); wa = w.clone().requires_grad_(True)
    ya = ops.dwconv(xa, wa, K // 2)
    ya.backward(gy)
    xs = src.clone().requires_grad_(True); wb = w.clone().requires_grad_(True)
    spk, _ = ops.lif(xs, None, keep_v=False, spikes=True)                 # Q_IFNode(k) = k / 8 as a bf16 Spikes pair
    assert spk.data.dtype == torch.bfloat16
    yb = ops.dwconv(spk, wb, K // 2)
    yb.backward(gy)
    assert torch.equal(ya, yb)
    # the weight gradient is summed over workgroups with atomics: equal to round-off
    assert (wa.grad - wb.grad).abs().max().item() <= 1e-5 * wa.grad.abs().max().item()
    assert torch.equal(xa.grad / 8, xs.grad)                              # d(k/8)/dk = 1/8 inside [0, 8]


def test_grouped_weight_gradients_are_the_single_launches(ops):
    """s2f_spike_gemm_dw_grouped (many short-contraction weight gradients in one launch, job table in the kernel arguments)
    accumulates into each destination exactly what one s2f_spike_gemm_dw_bf16 launch per job does (to the fp32 round-off of
    the split-K atomics), ragged shapes and both contraction-step classes included."""
    import ctypes
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(21)
    s = torch.cuda.current_stream().cuda_stream
    for bkv, shapes in ((64, [(8, 256, 256, 1024), (8, 768, 256, 1024), (2, 360, 360, 256), (8, 72, 40, 64), (1, 1024, 256, 4096)]),
                        (32, [(8, 2048, 256, 100), (8, 256, 2048, 100), (3, 151, 100, 36)])):
        jobs, keep = [], []
        for B, M, K, L in shapes:
            gy = torch.randn(B, M, L, generator=g).cuda()
            x = (torch.randint(0, 9, (B, K, L), generator=g).float() / 8).cuda().to(torch.bfloat16)
            base = torch.randn(M, K, generator=g).cuda()                     # accumulate semantics: dW += ...
            got, want = base.clone(), base.clone()
            check(lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), want.data_ptr(), B, M, K, L, 1, s), "single")
            jobs += [gy.data_ptr(), x.data_ptr(), got.data_ptr(), B, M, K, L]
            keep.append((gy, x, got, want, base))
        arr = (ctypes.c_int64 * len(jobs))(*jobs)
        check(lib.s2f_spike_gemm_dw_grouped(arr, len(shapes), bkv, s), "grouped")
        torch.cuda.synchronize()
        for gy, x, got, want, base in keep:
            ref = base.double() + torch.einsum("bml,bkl->mk", gy.double(), x.double())
            scale = ref.abs().max().item()
            assert (got.double() - ref).abs().max().item() <= 2e-6 * scale * 4
            assert (got - want).abs().max().item() <= 1e-5 * scale


@pytest.mark.gpu
def test_grouped_general_weight_gradients_are_the_single_launches(ops):
    """s2f_gemm_dw_general_grouped: several dW += sum_b dY[b] X[b]^T (both operands general fp32, strided groups) in one launch
    against the single launches and fp64."""
    import ctypes

    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(21)
    st = torch.cuda.current_stream().cuda_stream
    jobs, keep, want = [], [], []
    for (B, G, M, K, L) in [(8, 3, 256, 256, 1024), (2, 1, 100, 72, 260), (8, 1, 360, 360, 1024), (4, 2, 64, 130, 4096)]:
        gy, x = torch.randn(B, G * M, L, generator=g).cuda(), torch.randn(B, G * K, L, generator=g).cuda()
        for gi in range(G):
            sink = torch.randn(M, K, generator=g).cuda()
            ref = sink.double() + torch.einsum("bml,bkl->mk", gy[:, gi * M:(gi + 1) * M].double(), x[:, gi * K:(gi + 1) * K].double())
            single = sink.clone()
            check(lib.s2f_gemm_dw_general(gy.data_ptr() + 4 * gi * M * L, G * M * L, x.data_ptr() + 4 * gi * K * L, G * K * L,
                                          single.data_ptr(), B, M, K, L, 1, st), "single")
            jobs += [gy.data_ptr() + 4 * gi * M * L, G * M * L, x.data_ptr() + 4 * gi * K * L, G * K * L, sink.data_ptr(), B, M, K, L]
            keep.append((gy, x, sink, single))
            want.append(ref)
    arr = (ctypes.c_int64 * len(jobs))(*jobs)
    check(lib.s2f_gemm_dw_general_grouped(arr, len(want), st), "grouped")
    for (gy, x, sink, single), ref in zip(keep, want):
        scale = ref.abs().max().item()
        assert (sink.double() - ref).abs().max().item() <= 2e-6 * scale * 8
        assert (sink - single).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("shape", [(8, 256, 1024), (3, 100, 256), (2, 70, 36), (1, 5, 3), (2, 64, 130)])
def test_transpose_last2_is_the_permuted_copy(shape):
    """s2f_transpose_last2 == x.transpose(-1, -2).contiguous(), forward and adjoint, incl. ragged / unaligned tiles."""
    from spike2former_amd import ops
    torch.manual_seed(0)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    y = ops.transpose_last2(x)
    assert y.is_contiguous() and torch.equal(y, x.detach().transpose(-1, -2).contiguous())
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad, g.transpose(-1, -2).contiguous())
    x4 = torch.randn(2, 3, *shape[1:], device="cuda")
    assert torch.equal(ops.transpose_last2(x4), x4.transpose(-1, -2).contiguous())


@pytest.mark.gpu
def test_bf16_spike_storage_needs_a_power_of_two_D():
    """k / D is exact in bf16 only for a power-of-two D: other D keep fp32 spikes, and the C entry point refuses."""
    from spike2former_amd import ops
    from spike2former_amd._lib import S2FError, lib
    x = torch.randn(4, 64, 256, device="cuda") * 3
    y6, _ = ops.lif(x, D=6, keep_v=False, spikes=True)
    assert y6.tok is None and y6.data.dtype == torch.float32
    # the CPU quotient: ATen's CUDA `tensor / scalar` multiplies by the rounded reciprocal, the reference's CPU path divides
    assert torch.equal(y6.data.cpu(), torch.round(torch.clamp(x.cpu(), 0, 6)) / 6)
    y8, _ = ops.lif(x, D=8, keep_v=False, spikes=True)
    assert y8.data.dtype == torch.bfloat16 and torch.equal(y8.data.float(), torch.round(torch.clamp(x, 0, 8)) / 8)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    rc = lib.s2f_lif_fwd(x.data_ptr(), None, out.data_ptr(), None, None, None, None, x.numel(), 1.0, 6, 1,
                         torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    with pytest.raises(S2FError):
        ops.check(rc, "s2f_lif_fwd")


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,K,L", [(2, 130, 300, 260), (3, 256, 256, 1024), (1, 97, 1152, 520), (8, 128, 4608, 64), (5, 512, 512, 36)])
def test_wide_weight_gradients_vs_fp64(B, M, K, L):
    """Weight gradients with wide outputs (K >= 256) against the fp64 contraction: ragged M / K / L, odd and even step counts
    per split, fresh and accumulating destinations; 3e-6 of the scale."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 1000 + M + K + L)
    gy = (torch.randn(B, M, L, generator=g) * torch.rand(B, M, 1, generator=g) * 10).cuda()
    x = (torch.randint(0, 9, (B, K, L), generator=g).float() / 8).cuda().to(torch.bfloat16)
    ref = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    scale = torch.einsum("bml,bkl->mk", gy.abs().double(), x.abs().double()).max().item()
    s = torch.cuda.current_stream().cuda_stream
    out = torch.full((M, K), float("nan"), device="cuda")
    check(lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), out.data_ptr(), B, M, K, L, 0, s), "dw")
    assert (out.double() - ref).abs().max().item() <= 3e-6 * scale
    base = torch.randn(M, K, generator=g).cuda()
    acc = base.clone()
    check(lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), acc.data_ptr(), B, M, K, L, 1, s), "dw accumulate")
    assert (acc.double() - (base.double() + ref)).abs().max().item() <= 3e-6 * scale + 1e-6 * base.abs().max().item()


# ----------------------------------------------------------------------------------------------- LDS-DMA pipelined GEMMs (round 3)
def _spikes_bf16(shape, g):
    return (torch.randint(0, 9, shape, generator=g).float() / 8).cuda().bfloat16()


@pytest.mark.parametrize("B,M,K,N", [(2, 64, 32, 128), (3, 100, 72, 136), (2, 256, 256, 1024), (1, 360, 360, 1024), (2, 700, 96, 384),
                                     (8, 1024, 256, 1024), (1, 33, 1152, 256)])
def test_pgemm_forward_is_the_round2_spike_gemm(ops, B, M, K, N):
    """s2f_pgemm_nn_bf16 (packed weight, LDS-DMA, 2-3 stages) on ragged M / K and partial N tiles: against fp64 to the spike-GEMM
    bound, and -- same products, same accumulation order over (k slice, term) -- BIT-IDENTICAL to s2f_spike_gemm_fwd_bf16 without
    its in-workgroup K split; every tile configuration gives the same bits."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 1000 + M + K + N)
    w = (torch.randn(M, K, generator=g) * K ** -0.5).cuda()
    bias = torch.randn(M, generator=g).cuda()
    x = _spikes_bf16((B, K, N), g)
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.matmul(w.double(), x.double()) + bias.double().view(1, -1, 1)
    bound = 2e-6 * (torch.matmul(w.abs().double(), x.double()) + bias.abs().double().view(1, -1, 1)).max().item()
    outs = []
    for cfg in (1, 2, 3, 4) + ((5,) if M >= 256 else ()):
        y = torch.full((B, M, N), float("nan"), device="cuda")
        check(lib.s2f_pgemm_nn_bf16(ops.pack_weight(w).data_ptr(), x.data_ptr(), bias.data_ptr(), y.data_ptr(), B, M, N, K, 3, cfg,
                                    st), "pgemm")
        assert (y.double() - ref).abs().max().item() <= bound, cfg
        outs.append(y)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    if B * ((N + 127) // 128) * ((M + 63) // 64) >= 512 or K < 128:          # the round-2 kernel does not split K here
        ws = ops.split_weight(w)
        y0 = torch.empty(B, M, N, device="cuda")
        check(lib.s2f_spike_gemm_fwd_bf16(ws.data_ptr(), x.data_ptr(), bias.data_ptr(), y0.data_ptr(), B, M, N, K, ws.shape[1],
                                          ws.shape[2], 3, st), "old")
        assert torch.equal(y0, outs[0])


@pytest.mark.parametrize("B,Mo,Ki,N", [(2, 64, 32, 128), (3, 100, 72, 132), (2, 256, 256, 1024), (1, 360, 360, 1024),
                                       (2, 1440, 360, 256), (8, 256, 1024, 1024), (2, 40, 600, 100), (1, 2048, 256, 100)])
def test_pgemm_input_gradient_matches_fp64(ops, B, Mo, Ki, N):
    """s2f_pgemm_dx_f32: dX = W^T dY from the FORWARD pack of W (transpose reads of both operands, dY split hi + mid + lo in the
    kernel, 6 passes) on every kind of raggedness; fp32-GEMM accuracy, and beta = 1 accumulates."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 77 + Mo + Ki + N)
    w = (torch.randn(Mo, Ki, generator=g) * Mo ** -0.5).cuda()
    gy = torch.randn(B, Mo, N, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.matmul(w.t().double(), gy.double())
    bound = 2e-6 * torch.matmul(w.t().abs().double(), gy.abs().double()).max().item()
    for cfg in (3, 4, 5, 7, 8, 9):          # eight-wavefront tiles, 32-row tiles, 32-row steps: the plain-store forms
        if Mo >= 512 and N < 128:
            continue                         # (the launcher splits such contractions over gridDim.z with its own tile)
        dx = torch.full((B, Ki, N), float("nan"), device="cuda")
        check(lib.s2f_pgemm_dx_f32(ops.pack_weight(w).data_ptr(), gy.data_ptr(), 0, dx.data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg, st), "dx")
        assert (dx.double() - ref).abs().max().item() <= bound, cfg
    for cfg in (1, 2):
        dx = torch.full((B, Ki, N), float("nan"), device="cuda")
        check(lib.s2f_pgemm_dx_f32(ops.pack_weight(w).data_ptr(), gy.data_ptr(), 0, dx.data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg, st), "dx")
        assert (dx.double() - ref).abs().max().item() <= bound, cfg
        check(lib.s2f_pgemm_dx_f32(ops.pack_weight(w).data_ptr(), gy.data_ptr(), 0, dx.data_ptr(), 0, B, Mo, Ki, N, 1.0, cfg, st), "dx")
        assert (dx.double() - 2 * ref).abs().max().item() <= 2 * bound, cfg


@pytest.mark.parametrize("B,Mo,Ki,N", [(2, 64, 32, 128), (3, 100, 72, 136), (2, 256, 256, 1024), (1, 360, 360, 1024),
                                       (2, 1440, 360, 256), (8, 256, 1024, 1024), (1, 2048, 256, 104)])
def test_pgemm_input_gradient_from_presplit_planes(ops, B, Mo, Ki, N):
    """s2f_pgemm_dx_split: the same product from dY as three bf16 planes hi | mid | lo (what s2f_bn_act_bwd_split writes): the
    terms are the ones the fp32 kernel forms in registers, so the two agree to the reordering of 6 x K products (fp32-GEMM
    accuracy against fp64), every configuration."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 79 + Mo + Ki + N)
    w = (torch.randn(Mo, Ki, generator=g) * Mo ** -0.5).cuda()
    gy = torch.randn(B, Mo, N, generator=g).cuda()
    hi = gy.bfloat16(); r1 = gy - hi.float(); mid = r1.bfloat16(); lo = (r1 - mid.float()).bfloat16()
    planes = torch.stack([hi, mid, lo]).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.matmul(w.t().double(), gy.double())
    bound = 2e-6 * torch.matmul(w.t().abs().double(), gy.abs().double()).max().item()
    for cfg in (1, 2, 3, 4):
        dx = torch.full((B, Ki, N), float("nan"), device="cuda")
        check(lib.s2f_pgemm_dx_split(ops.pack_weight(w).data_ptr(), planes.data_ptr(), gy.numel(), dx.data_ptr(), B, Mo, Ki, N, cfg,
                                     st), "dxs")
        assert (dx.double() - ref).abs().max().item() <= bound, cfg
    with pytest.raises(RuntimeError):
        check(lib.s2f_pgemm_dx_split(ops.pack_weight(w).data_ptr(), planes.data_ptr(), gy.numel(), dx.data_ptr(), B, Mo, Ki, N + 4, 0,
                                     st), "dxs")


@pytest.mark.parametrize("N,C,L,training,res,lif", [(8, 16, 1024, True, True, True), (2, 7, 1052, True, False, True),
                                                    (8, 3, 16800, True, False, True), (2, 6, 4200, False, True, True),
                                                    (8, 8, 100, True, False, False), (2, 32, 65536, True, True, True)])
def test_bn_backward_presplit_planes_are_the_fp32_gradient(ops, monkeypatch, N, C, L, training, res, lif):
    """s2f_bn_act_bwd_split writes gz as hi | mid | lo bf16 planes: their sum IS the fp32 gz of s2f_bn_act_bwd (a 24-bit value
    splits exactly into three 8-bit terms), bit for bit, in every form of the backward (single pass, row-walking, generic) --
    and dgamma / dbeta / the residual gradient are untouched."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(N + C + L)
    z = (torch.randn(N, C, L, generator=g) * 2 + 0.5).cuda().requires_grad_(True)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True), torch.randn(C, generator=g).cuda().requires_grad_(True)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    r = torch.randn(N, C, L, generator=g).cuda().requires_grad_(True) if res else None
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    seen = {}
    orig = lib.s2f_bn_act_bwd_ports          # (what ops.bn_act's backward calls: s2f_bn_act_bwd + the second spike-gradient port)

    def both(*a):
        planes = torch.empty(3, N, C, L, dtype=torch.bfloat16, device="cuda")
        assert not a[6]                        # one reader here: no second gradient
        a2 = list(a[:6] + a[7:])               # the argument list of s2f_bn_act_bwd / _split
        tmp = [torch.empty(C, device="cuda") for _ in range(2)]
        gres2 = torch.empty(N, C, L, device="cuda") if a2[10] else None
        ws = a2[8]
        ws2 = torch.zeros(2 * C, dtype=torch.float64, device="cuda") if ws else None
        a2[8], a2[9], a2[10], a2[11], a2[12] = (ws2.data_ptr() if ws else None, planes.data_ptr(),
                                                gres2.data_ptr() if gres2 is not None else None, tmp[0].data_ptr(), tmp[1].data_ptr())
        check(lib.s2f_bn_act_bwd_split(*a2), "split")
        seen["planes"], seen["dg"], seen["db"], seen["gres"] = planes, tmp[0], tmp[1], gres2
        return orig(*a)
    monkeypatch.setattr(lib, "s2f_bn_act_bwd_ports", both)
    u, y, _ = ops.bn_act(z, None, gamma, beta, rm, rv, nbt if training else None, training, 0.1, 1e-5, residual=r, lif=lif,
                         want_pre=True)
    loss = (u * torch.randn(u.shape, generator=g).cuda()).sum()
    if lif:
        loss = loss + (y.float() * torch.randn(u.shape, generator=g).cuda()).sum()
    loss.backward()
    p = seen["planes"].float()
    assert torch.equal((p[0] + p[1]) + p[2], z.grad)
    assert torch.equal(seen["dg"], gamma.grad) and torch.equal(seen["db"], beta.grad)
    if res:
        assert torch.equal(seen["gres"], r.grad)


@pytest.mark.parametrize("B,G,M,K,L", [(3, 3, 72, 40, 256), (8, 3, 256, 256, 1024), (8, 3, 360, 360, 1024), (2, 2, 32, 96, 512),
                                       (1, 4, 64, 24, 132)])
def test_dense_gemm_groups_forward_backward(ops, B, G, M, K, L):
    """ops.dense_gemm (general fp32 input, G weights on consecutive channel groups -- the second 1x1 of the batched q / k / v
    chain): forward on the pack of W^T, input gradient on the pack of W, weight gradient on the 6-pass kernel; against fp64.  The
    G products run as ONE grouped launch each way (s2f_pgemm_dx_f32_grouped): bit-identical to the G separate launches, and its
    BatchNorm partials are the tile sums of every group's rows."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, G * K, L, generator=g).cuda().requires_grad_(True)
    ws = [(torch.randn(M, K, generator=g) * K ** -0.5).cuda().requires_grad_(True) for _ in range(G)]
    gy = torch.randn(B, G * M, L, generator=g).cuda()
    assert ops.DENSE_GROUPED
    ops.DENSE_GROUPED = False
    try:
        y0 = ops.dense_gemm(x, ws)
        y0.backward(gy)
        gx0 = x.grad.clone()
        x.grad = None
        for w in ws:
            w.grad = None
    finally:
        ops.DENSE_GROUPED = True
    single = ops.lib.s2f_bn_single_pass(B, G * M, L)
    ops.BN_PARTIALS_SINGLE = True
    try:
        y = ops.dense_gemm(x, ws, stats=True)
    finally:
        ops.BN_PARTIALS_SINGLE = False
    assert torch.equal(y, y0)
    part = ops.stats_of(y)
    assert part is not None and tuple(part.shape) == (G * M, B * ((L + 127) // 128), 2), single
    assert ((part.double() - _tile_sums(y.detach())).abs() <= 1e-5 * _tile_sums(y.detach().abs()) + 1e-30).all()
    y.backward(gy)
    assert torch.equal(x.grad, gx0)
    xd, gyd = x.detach().double().view(B, G, K, L), gy.double().view(B, G, M, L)
    for i, w in enumerate(ws):
        wd = w.detach().double()
        assert (y.detach().double().view(B, G, M, L)[:, i] - torch.matmul(wd, xd[:, i])).abs().max().item() <= 1e-5
        assert (x.grad.double().view(B, G, K, L)[:, i] - torch.matmul(wd.t(), gyd[:, i])).abs().max().item() <= 1e-5
        gw = torch.einsum("bml,bkl->mk", gyd[:, i], xd[:, i])
        assert (w.grad.double() - gw).abs().max().item() <= 2e-5 * gw.abs().max().item()


def test_pack_is_refreshed_in_place(ops):
    """A cached pack follows its weight: an in-place update bumps the version and the next use converts INTO THE SAME buffer (a
    captured hipGraph holds its address); resplit_all redoes packs and splits alike from the live weights."""
    g = torch.Generator().manual_seed(9)
    w = torch.randn(100, 72, generator=g).cuda()
    p0 = ops.pack_weight(w)
    ptr, snap = p0.data_ptr(), p0.clone()
    w.mul_(1.5)
    p1 = ops.pack_weight(w)
    assert p1.data_ptr() == ptr and not torch.equal(p1, snap)
    w.data.mul_(2.0)                                        # no version bump: only resplit_all can refresh it
    snap = p1.clone()
    assert ops.resplit_all(w.device) >= 1
    torch.cuda.synchronize()
    assert ops.pack_weight(w).data_ptr() == ptr and not torch.equal(p1, snap)
    pt = ops.pack_weight(w, transposed=True)
    assert pt.numel() == ((72 + 63) // 64) * ((100 + 31) // 32) * 6144


# ----------------------------------------------------------------------------------------------- eval-mode fusion (row f4)
@pytest.mark.parametrize("B,M,K,N,res,pre,stateful", [(2, 64, 32, 128, False, False, False), (3, 100, 72, 136, True, True, False),
                                                      (2, 256, 256, 1024, True, True, True), (1, 360, 360, 256, False, False, True)])
def test_gemm_bn_lif_eval_is_the_two_kernel_path(ops, B, M, K, N, res, pre, stateful):
    """s2f_gemm_bn_lif_fwd (conv1x1 -> BatchNorm(running statistics) [+ residual] -> Q_IFNode in the GEMM epilogue) against the
    unfused eval path  s2f_pgemm_nn_bf16 -> s2f_bn_act_fwd: same accumulator, same per-element expressions -> the spikes, the
    pre-activation, the carried membrane and the firing counters are IDENTICAL."""
    g = torch.Generator().manual_seed(B + M + K + N)
    w = (torch.randn(M, K, generator=g) * K ** -0.5).cuda()
    cb = torch.randn(M, generator=g).cuda()
    rm, rv = torch.randn(M, generator=g).cuda() * 0.1, (torch.rand(M, generator=g) + 0.5).cuda()
    ga, be = (torch.rand(M, generator=g) + 0.5).cuda(), torch.randn(M, generator=g).cuda() * 0.1
    x = ops.Spikes(_spikes_bf16((B, K, N), g), None)
    r = torch.randn(B, M, N, generator=g).cuda() if res else None
    v = torch.rand(B, M, N, generator=g).cuda() if stateful else None
    st_a, st_b = ops.new_stats("cuda"), ops.new_stats("cuda")
    with torch.no_grad():
        z = ops.spike_gemm(x, w)
        u0, y0, v0 = ops.bn_act(z, cb, ga, be, rm, rv, None, False, 0.1, 1e-5, residual=r, lif=True, want_pre=pre, v_in=v,
                                keep_v=stateful, stats=st_a)
        u1, y1, v1 = ops.gemm_bn_lif_eval(x, w, cb, rm, rv, ga, be, 1e-5, residual=r, want_pre=pre, lif=True, v_in=v,
                                          keep_v=stateful, stats=st_b)
    assert torch.equal(y0.data, y1.data)
    assert (u0 is None and u1 is None) or torch.equal(u0, u1)
    assert (v0 is None and v1 is None) or torch.equal(v0, v1)
    assert torch.equal(ops.read_stats(st_a), ops.read_stats(st_b)) and int(ops.read_stats(st_b)[1]) > 0


def test_reparam_helpers_fold_like_the_reference():
    """reparam.fused_conv2d_{weight,bias}_of_convbn2d / fuse_convbn2d (functional.py:574-692): conv(x; w', b') == BN_eval(conv(x; w))."""
    import torch.nn as nn
    from spike2former_amd import reparam
    g = torch.Generator().manual_seed(1)
    conv = nn.Conv2d(6, 10, 3, padding=1, bias=False)
    bn = nn.BatchNorm2d(10).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
        bn.weight.copy_(torch.rand(10, generator=g) + 0.5); bn.bias.copy_(torch.randn(10, generator=g))
        bn.running_mean.copy_(torch.randn(10, generator=g)); bn.running_var.copy_(torch.rand(10, generator=g) + 0.5)
        x = torch.randn(2, 6, 8, 8, generator=g)
        want = bn(conv(x))
        w2 = (conv.weight.transpose(0, 3) * bn.weight / (bn.running_var + bn.eps).sqrt()).transpose(0, 3)      # the reference's line
        assert torch.allclose(reparam.fused_conv2d_weight_of_convbn2d(conv, bn), w2, atol=0, rtol=1e-6)
        got = reparam.fuse_convbn2d(conv, bn)(x)
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,K,L", [(8, 256, 256, 1024), (2, 100, 72, 260), (8, 32, 64, 4096), (4, 360, 360, 1024)])
def test_weight_gradient_from_presplit_planes(ops, B, M, K, L):
    """s2f_spike_gemm_dw_bf16_split / s2f_spike_gemm_dw_grouped_split: dW from dY as bf16 planes == the fp32-dY kernels (the terms
    they form in registers are the planes), up to the order of the split-K atomics; fp64 as the yardstick."""
    import ctypes

    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B + M + K + L)
    gy = torch.randn(B, M, L, generator=g).cuda()
    x = (torch.randint(0, 9, (B, K, L), generator=g).float() / 8).cuda().bfloat16()
    hi = gy.bfloat16(); r1 = gy - hi.float(); mid = r1.bfloat16(); lo = (r1 - mid.float()).bfloat16()
    planes = torch.stack([hi, mid, lo]).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.einsum("bml,bkl->mk", gy.double(), x.double())
    scale = ref.abs().max().item()
    a, b_, c = (torch.zeros(M, K, device="cuda") for _ in range(3))
    check(lib.s2f_spike_gemm_dw_bf16(gy.data_ptr(), x.data_ptr(), a.data_ptr(), B, M, K, L, 0, st), "fp32")
    check(lib.s2f_spike_gemm_dw_bf16_split(planes.data_ptr(), gy.numel(), x.data_ptr(), b_.data_ptr(), B, M, K, L, 0, st), "split")
    arr = (ctypes.c_int64 * 7)(planes.data_ptr(), x.data_ptr(), c.data_ptr(), B, M, K, L)
    check(lib.s2f_spike_gemm_dw_grouped_split(arr, 1, 64 if (L % 64 == 0 or L >= 512) else 32, st), "grouped")
    for got in (a, b_, c):
        assert (got.double() - ref).abs().max().item() <= 4e-6 * scale
    assert (a - b_).abs().max().item() <= 2e-6 * scale and (a - c).abs().max().item() <= 2e-6 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,C,H,W", [(2, 64, 32, 32, 32), (1, 100, 64, 16, 64), (2, 32, 128, 64, 64), (1, 360, 256, 32, 32),
                                       (2, 128, 32, 20, 36), (1, 40, 96, 8, 8)])
def test_pgemm_conv3x3_is_the_round2_implicit_convolution(ops, N, M, C, H, W):
    """s2f_pgemm_conv3x3_bf16 / _f32 (LDS-DMA weight panels, register-staged shifted activation rows, transpose reads) against
    the round-2 implicit kernels -- same products in the same order: bit-identical -- and against F.conv2d in fp64, every tile
    configuration; the input-gradient form convolves dY with the flipped transposed weight (pack mode 2)."""
    import torch.nn.functional as F

    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(N * 7 + M + C + H + W)
    w = (torch.randn(M, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
    xs = (torch.randint(0, 9, (N, C, H, W), generator=g).float() / 8).cuda()
    xb = xs.bfloat16()
    bias = torch.randn(M, generator=g).cuda()
    st = torch.cuda.current_stream().cuda_stream
    # forward
    ref = F.conv2d(xs.double(), w.double(), bias.double(), padding=1)
    ws = ops.split_weight_conv3(w)
    y0 = torch.empty(N, M, H, W, device="cuda")
    check(lib.s2f_spike_conv3x3_fwd_bf16(ws.data_ptr(), xb.data_ptr(), bias.data_ptr(), y0.data_ptr(), N, M, C, H, W, ws.shape[1],
                                         ws.shape[2], 3, st), "old")
    for cfg in (1, 2, 3, 4):
        y1 = torch.full((N, M, H, W), float("nan"), device="cuda")
        check(lib.s2f_pgemm_conv3x3_bf16(ops.pack_weight_conv3(w).data_ptr(), xb.data_ptr(), bias.data_ptr(), y1.data_ptr(), N, M, C, H,
                                         W, cfg, st), "new")
        # same products; the round-2 kernel splits the contraction over wavefront groups when the grid is small, otherwise the
        # sums are ordered alike and the results bit-identical (tools/probe_pgemm.py conv prints '=')
        assert (y1 - y0).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1.0), cfg
        assert (y1.double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), 1.0)
    # input gradient of the convolution [M <- C]: dX[C] = conv(dY[M], flip(W)^T); needs M % 32 == 0 as the contraction channels
    if M % 32 == 0:
        gy = torch.randn(N, M, H, W, generator=g).cuda()
        refx = F.conv_transpose2d(gy.double(), w.double(), padding=1)
        wt = ops.split_weight_tconv3(w)
        g0 = torch.empty(N, C, H, W, device="cuda")
        check(lib.s2f_conv3x3_general(wt.data_ptr(), gy.data_ptr(), g0.data_ptr(), N, C, M, H, W, wt.shape[1], wt.shape[2], st), "oldx")
        for cfg in (1, 2, 3, 4):
            g1 = torch.full((N, C, H, W), float("nan"), device="cuda")
            check(lib.s2f_pgemm_conv3x3_f32(ops.pack_weight_conv3(w, transposed=True).data_ptr(), gy.data_ptr(), g1.data_ptr(), N, C, M,
                                            H, W, cfg, st), "newx")
            assert (g1 - g0).abs().max().item() <= 1e-6 * max(refx.abs().max().item(), 1.0), cfg
            assert (g1.double() - refx).abs().max().item() <= 4e-6 * max(refx.abs().max().item(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,R,C", [(8, 1024, 256), (2, 64, 128), (3, 192, 64)])
def test_transpose_scale_add_is_transpose_plus_addcmul(ops, B, R, C):
    """ops.transpose_scale_add (q + gamma * x^T in one pass, pixel-decoder FFN residual) against transpose + addcmul under autograd:
    forward bit-identical (same two roundings per element), gradients to fp32 round-off of the per-channel reduction."""
    g = torch.Generator().manual_seed(B + R + C)
    x = torch.randn(B, R, C, generator=g).cuda().requires_grad_(True)
    q = torch.randn(B, C, R, generator=g).cuda().requires_grad_(True)
    gam = (torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True)
    w = torch.randn(B, C, R, generator=g).cuda()
    y = ops.transpose_scale_add(x, q, gam)
    (y * w).sum().backward()
    got = (y.detach().clone(), x.grad.clone(), q.grad.clone(), gam.grad.clone())
    for t in (x, q, gam):
        t.grad = None
    ref = torch.addcmul(q, x.transpose(1, 2), gam.view(1, C, 1))
    (ref * w).sum().backward()
    assert torch.equal(got[0], ref.detach())
    assert torch.equal(got[1], x.grad) and torch.equal(got[2], q.grad)
    assert (got[3] - gam.grad).abs().max().item() <= 1e-5 * gam.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,o,bias", [(5600, 256, 256, False), (5600, 256, 151, True), (37, 64, 10, True), (700, 128, 96, False)])
def test_token_major_linear_on_own_kernels(ops, n, c, o, bias):
    """ops.linear_tm (nn.Linear on a token-major activation through s2f_gemm_dw_general / s2f_pgemm_dx_f32, no vendor GEMM) against
    F.linear in fp64: output and all three gradients to fp32-GEMM accuracy."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n + c + o)
    x = torch.randn(2, n // 2 if n % 2 == 0 else n, c, generator=g)[:, :n // 2 if n % 2 == 0 else n]
    x = (x.reshape(-1, c)[:n].reshape(1, n, c)).cuda().requires_grad_(True)
    w = (torch.randn(o, c, generator=g) * c ** -0.5).cuda().requires_grad_(True)
    b = torch.randn(o, generator=g).cuda().requires_grad_(True) if bias else None
    gy = torch.randn(1, n, o, generator=g).cuda()
    y = ops.linear_tm(x, w, b)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.double())

    def close(a, r, tol=4e-6):
        return (a.double() - r).abs().max().item() <= tol * max(r.abs().max().item(), 1e-6) * (c if r is yd else 1) ** 0.0 + 1e-12
    scale = lambda r: r.abs().max().item()
    assert (y.double() - yd).abs().max().item() <= 4e-6 * scale(yd)
    assert (x.grad.double() - xd.grad).abs().max().item() <= 4e-6 * scale(xd.grad)
    assert (w.grad.double() - wd.grad).abs().max().item() <= 4e-6 * scale(wd.grad)
    if bias:
        assert (b.grad.double() - bd.grad).abs().max().item() <= 1e-5 * scale(bd.grad)


# ----------------------------------------------------------------------------------------------- BatchNorm statistics from the GEMM epilogue
def _tile_sums(y):
    """fp64 reference of the partials: per (batch, 128-column tile, row) the sum and the sum of squares of y [B, M, N]"""
    B, M, N = y.shape
    nt = (N + 127) // 128
    yp = torch.nn.functional.pad(y.double(), (0, nt * 128 - N)).view(B, M, nt, 128)
    return torch.stack([yp.sum(-1), (yp * yp).sum(-1)], -1).permute(1, 0, 2, 3).reshape(M, B * nt, 2)          # [row, (b, tile), 2]


@pytest.mark.parametrize("B,M,K,N", [(2, 64, 32, 128), (3, 100, 72, 136), (8, 256, 256, 100), (2, 256, 256, 1024), (1, 360, 1440, 1024),
                                     (2, 40, 96, 4096), (8, 2048, 256, 100), (1, 33, 1152, 260)])
def test_gemm_epilogue_partials_are_the_tile_sums(ops, B, M, K, N):
    """s2f_pgemm_nn_bf16_stats / s2f_pgemm_dx_f32_stats: the product is bit-identical to the plain entry point, and the partials
    bn_partials[row, p] are the sum / sum of squares of the 128-column tile p of that row (fp32 sums of <= 128 values against
    fp64: 1e-5 of the tile's sum of |.|), ragged M / N included -- columns past N and rows past M contribute nothing."""
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(B * 1000 + M + K + N)
    w = (torch.randn(M, K, generator=g) * K ** -0.5).cuda()
    x = _spikes_bf16((B, K, N), g)
    st = torch.cuda.current_stream().cuda_stream
    P = lib.s2f_bn_partials_count(B, N)
    assert P == B * ((N + 127) // 128)
    y0 = torch.empty(B, M, N, device="cuda")
    check(lib.s2f_pgemm_nn_bf16(ops.pack_weight(w).data_ptr(), x.data_ptr(), None, y0.data_ptr(), B, M, N, K, 3, 0, st), "plain")
    y1 = torch.full((B, M, N), float("nan"), device="cuda")
    part = torch.full((M, P, 2), float("nan"), device="cuda")
    check(lib.s2f_pgemm_nn_bf16_stats(ops.pack_weight(w).data_ptr(), x.data_ptr(), y1.data_ptr(), part.data_ptr(), B, M, N, K, st), "stats")
    assert torch.equal(y0, y1)
    ref = _tile_sums(y1)
    scale = _tile_sums(y1.abs())
    assert ((part.double() - ref).abs() <= 1e-5 * scale + 1e-30).all()
    # the dense-input forward product (the transposed kernel on the pack of W^T), also as a group of a wider channel table
    xf = torch.randn(B, K, N, generator=g).cuda()
    z0 = torch.empty(B, M, N, device="cuda")
    pk = ops.pack_weight(w, transposed=True)          # the pack of w^T [K, M]: its transposed product IS  w @ x  (ops._DenseGemm)
    check(lib.s2f_pgemm_dx_f32(pk.data_ptr(), xf.data_ptr(), 0, z0.data_ptr(), 0, B, K, M, N, 0.0, 0, st), "dense plain")
    z1 = torch.full((B, M, N), float("nan"), device="cuda")
    wide = torch.full((M + 24, P, 2), float("nan"), device="cuda")
    check(lib.s2f_pgemm_dx_f32_stats(pk.data_ptr(), xf.data_ptr(), 0, z1.data_ptr(), 0, wide.data_ptr() + 8 * 16 * P, B, K, M, N, st),
          "dense stats")
    # (the plain entry point may split a long contraction over gridDim.z with atomics; the statistics form never does)
    assert torch.equal(z0, z1) or (z0 - z1).abs().max().item() <= 2e-6 * torch.matmul(w.abs().double(), xf.abs().double()).max().item()
    assert ((wide[16:16 + M].double() - _tile_sums(z1)).abs() <= 1e-5 * _tile_sums(z1.abs()) + 1e-30).all()
    assert torch.isnan(wide[:16]).all() and torch.isnan(wide[16 + M:]).all()          # only this group's rows are written
    assert (z1.double() - torch.matmul(w.double(), xf.double())).abs().max().item() <= 2e-6 * torch.matmul(w.abs().double(), xf.abs().double()).max().item()


@pytest.mark.parametrize("N,M,C,H,W", [(2, 64, 32, 16, 16), (1, 128, 64, 32, 32), (2, 32, 128, 64, 64), (1, 100, 32, 8, 32)])
def test_conv3x3_epilogue_partials_are_the_tile_sums(ops, N, M, C, H, W):
    from spike2former_amd._lib import check, lib
    g = torch.Generator().manual_seed(N + M + C + H)
    w = (torch.randn(M, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
    x = _spikes_bf16((N, C, H, W), g)
    st = torch.cuda.current_stream().cuda_stream
    P = lib.s2f_bn_partials_count(N, H * W)
    y0 = torch.empty(N, M, H * W, device="cuda")
    check(lib.s2f_pgemm_conv3x3_bf16(ops.pack_weight_conv3(w).data_ptr(), x.data_ptr(), None, y0.data_ptr(), N, M, C, H, W, 0, st), "plain")
    y1 = torch.full((N, M, H * W), float("nan"), device="cuda")
    part = torch.full((M, P, 2), float("nan"), device="cuda")
    check(lib.s2f_pgemm_conv3x3_bf16_stats(ops.pack_weight_conv3(w).data_ptr(), x.data_ptr(), y1.data_ptr(), part.data_ptr(), N, M, C, H, W,
                                           st), "stats")
    assert torch.equal(y0, y1)
    assert ((part.double() - _tile_sums(y1)).abs() <= 1e-5 * _tile_sums(y1.abs()) + 1e-30).all()


@pytest.mark.parametrize("N,C,L,res,lif", [(8, 256, 100, False, True), (8, 64, 4096, True, True), (2, 7, 1052, False, True),
                                           (8, 32, 16384, False, True), (3, 20, 640, True, False), (8, 256, 1024, False, True),
                                           (1, 8, 65536, False, True), (4, 6, 12, True, True)])
def test_bn_from_partials_is_bn_from_the_statistics_pass(ops, spike_mode, N, C, L, res, lif):
    """s2f_bn_partials_finalize + s2f_bn_act_fwd == s2f_bn_stats + s2f_bn_act_fwd: the partials of a GEMM epilogue (here: exact fp64 tile sums rounded
    to fp32, the kernel's storage format) give the same mean / rstd to fp32 round-off (1e-6), the same pre-activation to 1e-5,
    spikes that differ in <= 1e-4 of the elements by one level, the same running statistics; every apply kernel: generic
    (L = 100), row-walking aligned / straddling, few (<= 32: one thread per channel) and many (wave per channel) partials, the
    conv bias shifting the mean."""
    spike_mode(True)
    g = torch.Generator().manual_seed(N * 31 + C + L)
    z = (torch.randn(N, C, L, generator=g) * 2 + 0.7).cuda()
    b = torch.randn(C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3 + 0.5).cuda()
    r = torch.randn(N, C, L, generator=g).cuda() if res else None
    part = _tile_sums(z).float().contiguous()
    outs = []
    for p in (None, part):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda")
        before = list(ops.BN_PARTIALS_USED)
        u, y, _, border = ops.bn_act(z, b, gamma, beta, rm, rv, nbt, True, 0.1, 1e-5, residual=r, lif=lif, want_pre=True,
                                     want_border=True, partials=p)
        used = [a - c for a, c in zip(ops.BN_PARTIALS_USED, before)]
        outs.append((u, y.float() if lif else None, rm, rv, border, int(nbt), used))
    (u0, y0, rm0, rv0, b0, n0, used0), (u1, y1, rm1, rv1, b1, n1, used1) = outs
    single = bool(__import__("spike2former_amd")._lib.lib.s2f_bn_single_pass(N, C, L))
    assert used1 == ([1, 0] if (not single or ops.BN_PARTIALS_SINGLE) else [0, 0]) and used0[0] == 0
    close = lambda a, c, tol: (a - c).abs().max().item() <= tol * max(c.abs().max().item(), 1e-6)
    assert close(u1, u0, 1e-5) and close(rm1, rm0, 1e-6) and close(rv1, rv0, 1e-5) and close(b1, b0, 1e-5) and n0 == n1 == 1
    if lif:
        d = (y1 - y0) * 8
        assert d.abs().max().item() <= 1 and (d != 0).float().mean().item() <= 1e-4


@pytest.mark.parametrize("N,C,L,res,lif", [(8, 256, 1024, False, True), (8, 768, 1024, False, True), (8, 360, 1024, True, False),
                                           (4, 64, 512, True, True), (8, 256, 1024, True, True), (8, 128, 2048, True, True),
                                           (16, 64, 1024, False, True)])
def test_batchnorm_pair_as_one_kernel_is_two_batchnorms(ops, spike_mode, N, C, L, res, lif):
    """s2f_bn2_act_fwd / _bwd (train-mode BN2(BN1(z)) [+ residual] [-> neuron] in one single-pass kernel each way: the pair closing
    every RepConv chain, sdtv2.py:280-296) against two s2f_bn_act calls: pre-activation 1e-5, spikes <= 1e-4 of the elements by one
    level, both BatchNorms' running statistics 1e-5, gz / g_residual / dgamma2 / dbeta2 1e-4 of their scale, and the first
    BatchNorm's parameter gradients -- dbeta1 = 0 exactly (round-off noise in the two-kernel form) and dgamma1 = O(eps) -- to 1e-4
    of the scale of dgamma2."""
    import types
    spike_mode(True)
    g = torch.Generator().manual_seed(N + C + L)
    z = (torch.randn(N, C, L, generator=g) * 1.7 + 0.3).cuda()
    r = torch.randn(N, C, L, generator=g).cuda() if res else None
    gu_w, gy_w = torch.randn(N, C, L, generator=g).cuda(), torch.randn(N, C, L, generator=g).cuda()

    def bns():
        out = []
        for k in range(2):
            gg = torch.Generator().manual_seed(100 + k)
            out.append(types.SimpleNamespace(weight=(torch.rand(C, generator=gg) + 0.5).cuda().requires_grad_(True),
                                             bias=(torch.randn(C, generator=gg) * 0.3 + 0.4).cuda().requires_grad_(True),
                                             running_mean=torch.zeros(C, device="cuda"), running_var=torch.ones(C, device="cuda"),
                                             num_batches_tracked=torch.zeros((), dtype=torch.int64, device="cuda"), momentum=0.1,
                                             eps=1e-5, training=True))
        return out
    outs = []
    for fusedv in (False, True):
        b1, b2 = bns()
        zc = z.clone().requires_grad_(True)
        rc = r.clone().requires_grad_(True) if res else None
        if fusedv:
            assert ops.bn2_act_ok(zc)
            u, y, _ = ops.bn2_act(zc, b1, b2, residual=rc, lif=lif, want_pre=True)
        else:
            x, _, _ = ops.bn_act(zc, None, b1.weight, b1.bias, b1.running_mean, b1.running_var, b1.num_batches_tracked, True, 0.1, 1e-5)
            u, y, _ = ops.bn_act(x, None, b2.weight, b2.bias, b2.running_mean, b2.running_var, b2.num_batches_tracked, True, 0.1, 1e-5,
                                 residual=rc, lif=lif, want_pre=True)
        loss = (u * gu_w).sum()
        if lif:
            loss = loss + (y.float() * gy_w).sum()
        loss.backward()
        outs.append((u.detach(), y.float().detach() if lif else None, zc.grad, rc.grad if res else None, b1, b2))
    (u0, y0, gz0, gr0, a1, a2), (u1, y1, gz1, gr1, c1, c2) = outs
    close = lambda a, b, tol, scale=None: (a - b).abs().max().item() <= tol * max((b.abs().max().item() if scale is None else scale), 1e-6)
    assert close(u1, u0, 1e-5)
    flips = False
    if lif:
        d = (y1 - y0) * 8
        assert d.abs().max().item() <= 1 and (d != 0).float().mean().item() <= 1e-4
        flips = bool((d != 0).any())
    for x, y_ in ((c1, a1), (c2, a2)):
        assert close(x.running_mean, y_.running_mean, 1e-5, 1.0) and close(x.running_var, y_.running_var, 1e-5) and int(x.num_batches_tracked) == 1
    tol = 5e-3 if flips else 1e-4
    assert close(gz1, gz0, tol) and close(c2.weight.grad, a2.weight.grad, tol) and close(c2.bias.grad, a2.bias.grad, tol)
    if res:
        assert close(gr1, gr0, tol)
    s2 = a2.weight.grad.abs().max().item()
    assert close(c1.weight.grad, a1.weight.grad, tol, s2) and close(c1.bias.grad, a1.bias.grad, tol, s2)
    assert float(c1.bias.grad.abs().max()) == 0.0


@pytest.mark.parametrize("N,M,C,H,W,res,lif", [(2, 64, 32, 16, 16, False, True), (1, 128, 64, 32, 32, True, True), (2, 32, 128, 64, 64, True, False),
                                               (8, 512, 128, 64, 64, False, True), (1, 100, 32, 8, 32, True, True)])
def test_eval_conv3x3_bn_lif_fusion_is_the_unfused_path(ops, spike_mode, N, M, C, H, W, res, lif):
    """s2f_conv3x3_bn_lif_fwd (eval-mode conv3x3 -> BatchNorm -> [+ residual] -> neuron in the implicit convolution's epilogue) against
    s2f_pgemm_conv3x3_bf16 + s2f_bn_act_fwd in eval mode: same accumulator, same per-element expressions -> identical spikes,
    pre-activation and firing counters."""
    spike_mode(True)
    g = torch.Generator().manual_seed(N + M + C + H)
    w = (torch.randn(M, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
    x = ops.Spikes(_spikes_bf16((N, C, H, W), g), None)
    x.tok = ops._new_tok(x.data)
    gamma, beta = (torch.rand(M, generator=g) + 0.5).cuda(), (torch.randn(M, generator=g) * 0.3 + 0.5).cuda()
    rm, rv = (torch.randn(M, generator=g) * 0.1).cuda(), (torch.rand(M, generator=g) + 0.5).cuda()
    r = torch.randn(N, M, H, W, generator=g).cuda() if res else None
    st1, st0 = ops.new_stats("cuda"), ops.new_stats("cuda")
    with torch.no_grad():
        u1, y1, _ = ops.conv3x3_bn_lif_eval(x, w, rm, rv, gamma, beta, 1e-5, residual=r, want_pre=True, lif=lif, stats=st1 if lif else None)
        from spike2former_amd._lib import check, lib
        z = torch.empty(N, M, H, W, device="cuda")          # the implicit kernel itself (ops.conv_dense takes im2col below 32 x 32)
        check(lib.s2f_pgemm_conv3x3_bf16(ops.pack_weight_conv3(w).data_ptr(), x.data.data_ptr(), None, z.data_ptr(), N, M, C, H, W, 0,
                                         torch.cuda.current_stream().cuda_stream), "conv")
        u0, y0, _ = ops.bn_act(z, None, gamma, beta, rm, rv, None, False, 0.1, 1e-5, residual=r, lif=lif, want_pre=True,
                               stats=st0 if lif else None)
    assert torch.equal(u1, u0)
    if lif:
        assert torch.equal(y1.data, y0.data) and ops.read_stats(st1).tolist() == ops.read_stats(st0).tolist()


@pytest.mark.parametrize("N,C,H,W,K,lif,xbf", [(2, 64, 32, 32, 3, True, True), (1, 32, 64, 64, 5, True, True), (2, 16, 32, 32, 7, True, False),
                                               (1, 8, 128, 128, 3, True, True), (2, 24, 20, 36, 5, False, True), (1, 8, 256, 256, 3, True, False)])
def test_eval_dwconv_bn_lif_fusion_is_the_unfused_path(ops, spike_mode, N, C, H, W, K, lif, xbf):
    """s2f_dwconv_bn_lif_fwd (eval mode: depthwise stencil with the BatchNorm and the neuron in its store, both stencil forms) against
    s2f_dwconv_fwd + s2f_bn_act_fwd in eval mode: identical pre-activation, spikes and firing counters."""
    spike_mode(True)
    g = torch.Generator().manual_seed(N + C + H + K)
    w = (torch.randn(C, 1, K, K, generator=g) * K ** -1.0).cuda()
    xs = _spikes_bf16((N, C, H, W), g)
    x = ops.Spikes(xs, ops._new_tok(xs)) if xbf else xs.float()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3 + 0.5).cuda()
    rm, rv = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    st1, st0 = ops.new_stats("cuda"), ops.new_stats("cuda")
    with torch.no_grad():
        u1, y1 = ops.dwconv_bn_lif_eval(x, w, K // 2, rm, rv, gamma, beta, 1e-5, want_pre=True, lif=lif, stats=st1 if lif else None)
        z = ops.dwconv(x, w, K // 2)
        u0, y0, _ = ops.bn_act(z, None, gamma, beta, rm, rv, None, False, 0.1, 1e-5, lif=lif, want_pre=True, stats=st0 if lif else None)
    assert torch.equal(u1, u0)
    if lif:
        assert torch.equal(y1.data, y0.data) and ops.read_stats(st1).tolist() == ops.read_stats(st0).tolist()


@pytest.mark.parametrize("N,C,H,W,k,s,p,bf", [(2, 3, 64, 64, 7, 2, 3, False), (2, 32, 32, 32, 3, 2, 1, True), (1, 64, 16, 16, 3, 1, 1, True),
                                              (2, 5, 13, 22, 3, 2, 1, False), (1, 8, 9, 7, 3, 1, 1, True), (1, 4, 31, 30, 5, 3, 2, False),
                                              (2, 16, 128, 128, 3, 2, 1, True)])
def test_im2col_and_col2im_are_unfold_and_fold(ops, N, C, H, W, k, s, p, bf):
    """s2f_im2col is torch.nn.functional.unfold bit for bit (a gather); s2f_col2im is fold up to the order of its <= k*k additions per
    pixel (fold scatters with atomics), and exactly the adjoint of im2col."""
    g = torch.Generator().manual_seed(N * 7 + C + H + k)
    x = (_spikes_bf16((N, C, H, W), g) if bf else torch.randn(N, C, H, W, generator=g).cuda())
    cols = ops.im2col(x, k, k, s, p)
    want = torch.nn.functional.unfold(x.float(), (k, k), 1, p, s)
    assert cols.dtype == x.dtype and torch.equal(cols.float(), want)
    d = torch.randn(want.shape, generator=g).cuda()
    gx = ops.col2im(d, C, H, W, k, k, s, p)
    ref = torch.nn.functional.fold(d.double(), (H, W), (k, k), 1, p, s)
    assert (gx.double() - ref).abs().max().item() <= 1e-6 * k * k
    # adjoint: <im2col(x), d> == <x, col2im(d)>
    a = (want.double() * d.double()).sum().item()
    b = (x.double() * gx.double()).sum().item()
    assert abs(a - b) <= 1e-9 * max(1.0, abs(a)) + 1e-6 * want.abs().sum().item() * 1e-3


@pytest.mark.parametrize("B,M,K,N", [(2, 64, 32, 4), (3, 40, 72, 4), (1, 256, 256, 4)])
def test_spike_gemm_on_a_2x2_map(ops, spike_mode, B, M, K, N):
    """N = 4 (a 2 x 2 map: a 32 x 32 crop at stride 16) is below the packed-weight kernels' N >= 8: ops.spike_gemm takes the round-2
    kernel there (forward, weight gradient) and the input gradient whatever path it picks -- all against fp64."""
    spike_mode(True)
    g = torch.Generator().manual_seed(B + M + K)
    xs = _spikes_bf16((B, K, N), g)
    w = (torch.randn(M, K, generator=g) * K ** -0.5).cuda().requires_grad_()
    tok = ops._new_tok(xs).clone().requires_grad_()
    y = ops.spike_gemm(ops.Spikes(xs, tok), w)
    gy = torch.randn(B, M, N, generator=g).cuda()
    y.backward(gy)
    xd, wd = xs.double(), w.detach().double()
    assert (y.double() - wd @ xd).abs().max().item() <= 2e-6 * (wd.abs() @ xd).max().item()
    gx = torch.einsum("mk,bmn->bkn", wd, gy.double())
    gw = torch.einsum("bmn,bkn->mk", gy.double(), xd)
    assert (tok.grad.double() - gx).abs().max().item() <= 4e-6 * torch.einsum("mk,bmn->bkn", wd.abs(), gy.double().abs()).max().item()
    assert (w.grad.double() - gw).abs().max().item() <= 4e-6 * torch.einsum("bmn,bkn->mk", gy.double().abs(), xd).max().item()


@pytest.mark.parametrize("bands", [2, 3, 7])
def test_dcnv3_backward_in_bands_is_the_one_workgroup_backward(ops, bands):
    """Maps whose (n, group) slice does not fit in one CU's LDS (C3: 64 x 32, C5: 50 x 84) run the DCNv3 backward in bands of input
    rows (dcn_bwd_lds_kernel<true>): forced on the C2 geometry, all three gradients are bit-identical to the one-workgroup kernel
    (same fixed-point scale, integer accumulation; the offset / mask gradients are formed by exactly one band each)."""
    import os
    from spike2former_amd._lib import lib
    g = torch.Generator().manual_seed(40 + bands)
    N, H, W, G, Cg = 2, 32, 32, 32, 8
    x = torch.randn(N, H, W, G * Cg, generator=g).cuda()
    off = (torch.randn(N, H, W, G * 18, generator=g) * 3).cuda()
    m = torch.rand(N, H, W, G * 9, generator=g).cuda()
    gy = torch.randn(N, H, W, G * Cg, generator=g).cuda()

    def bwd():
        gx, goff, gm = torch.full_like(x, float("nan")), torch.full_like(off, float("nan")), torch.full_like(m, float("nan"))
        rc = lib.s2f_dcnv3_bwd(x.data_ptr(), off.data_ptr(), m.data_ptr(), gy.data_ptr(), gx.data_ptr(), goff.data_ptr(), gm.data_ptr(),
                               N, H, W, G, Cg, 3, 3, 1, 1, 1, 1, 1, 1, 1.0, None)
        assert rc == 0
        torch.cuda.synchronize()
        return gx, goff, gm
    want = bwd()
    os.environ["S2F_DCN_FORCE_BANDS"] = str(bands)
    try:
        got = bwd()
    finally:
        del os.environ["S2F_DCN_FORCE_BANDS"]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,W", [(1, 50, 84), (2, 64, 32), (1, 37, 20)])
def test_dcnv3_on_maps_larger_than_one_cu_vs_oracle(ops, so, N, H, W):
    """C5's 50 x 84 and C3's 64 x 32 pixel-decoder maps (G = 32, Cg = 8): forward and banded backward against the oracle's gather
    restatement, tolerances of the C2-geometry test."""
    g = torch.Generator().manual_seed(H * W)
    G, Cg = 32, 8
    x = torch.randn(N, H, W, G * Cg, generator=g)
    off = torch.randn(N, H, W, G * 18, generator=g) * 2
    m = torch.clamp(torch.round(torch.randn(N, H, W, G * 9, generator=g) + 1), 0, 8) / 8
    gy = torch.randn(N, H, W, G * Cg, generator=g)
    xo, oo, mo = (t.clone().requires_grad_(True) for t in (x, off, m))
    yo = so.dcnv3_core(xo, oo, mo, G, Cg)
    yo.backward(gy)
    xc, oc, mc = (t.cuda().requires_grad_(True) for t in (x, off, m))
    y = ops.dcnv3_core(xc, oc, mc, 3, 3, 1, 1, 1, 1, 1, 1, G, Cg, 1.0)
    y.backward(gy.cuda())
    for a, b, tol, name in ((y, yo.detach(), 1e-5, "y"), (xc.grad, xo.grad, 1e-5, "gx"), (mc.grad, mo.grad, 1e-5, "gm"),
                            (oc.grad, oo.grad, 1e-4, "goff")):
        assert (a.detach().cpu() - b).abs().max().item() <= tol * b.abs().max().item(), name
