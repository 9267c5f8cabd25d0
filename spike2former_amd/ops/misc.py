"""Tiled transposes, exact-2x bilinear up-sampling, the mask-loss kernels of the Hungarian-matched loss."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)


# ------------------------------------------------------------------------------------------------ transposition
import os as _os
_SMALL_PORTS = _os.environ.get("S2F_FANOUT_SMALL", "1") != "0"          # A/B switch of the two ports on small decoder tensors


class _TransposeLast2(torch.autograd.Function):
    """x [B, R, C] -> [B, C, R], contiguous (the adjoint is the same kernel the other way round).  -> (x^T, pass-through of x or an
    empty stand-in): the pass-through serves a second reader of x, whose gradient the adjoint sums (s2f_transpose_last2_add)."""

    @staticmethod
    def forward(ctx, x, skip):
        _need_cuda(x)
        x_in = x
        x = x.contiguous()
        B, R, C = x.shape
        y = torch.empty(B, C, R, dtype=torch.float32, device=x.device)
        check(lib.s2f_transpose_last2(_ptr(x), _ptr(y), B, R, C, _stream()), "s2f_transpose_last2")
        ctx.set_materialize_grads(False)
        if skip:
            return y, x_in
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        return y, aux

    @staticmethod
    def backward(ctx, gy, gskip):
        if gy is None:
            return gskip, None
        gy = gy.contiguous()
        B, C, R = gy.shape
        gx = torch.empty(B, R, C, dtype=torch.float32, device=gy.device)
        if gskip is not None:
            gskip = gskip.contiguous()
        check(lib.s2f_transpose_last2_add(_ptr(gy), _ptr(gskip), _ptr(gx), B, C, R, _stream()), "s2f_transpose_last2_add")
        return gx, None


class _FanOut(torch.autograd.Function):
    """x -> n aliases of x, one per reader; backward: the readers' gradients summed by ONE launch (s2f_sum_n) in the order the autograd
    engine would have accumulated them (last reader first) -- instead of n - 1 add launches (cfg.FANOUT_PORTS; the decoder's query
    position embedding has twelve readers per step)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g.contiguous() for g in reversed(gs) if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        import ctypes
        out = torch.empty_like(live[0])
        while len(live) > 1:          # (16 addends per launch)
            part = live[:16]
            arr = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
            check(lib.s2f_sum_n(arr, len(part), _ptr(out), out.numel(), _stream()), "s2f_sum_n")
            live = [out] + live[16:]
            if len(live) > 1:
                out = torch.empty_like(out)
        return live[0], None


def fan_out(x, n):
    """-> n tensors that are x, for n readers whose gradients one launch sums (float32 CUDA tensors with cfg.FANOUT_PORTS; otherwise
    x itself n times: the autograd engine adds)"""
    if n > 1 and cfg.FANOUT_PORTS and _SMALL_PORTS and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled():
        return list(_FanOut.apply(x, n))
    return [x] * n


class _TransposeScaleAdd(torch.autograd.Function):
    """q + g[c] * x^T: x [B, R, C] token-major, q [B, C, R] channel-major, g [C] (s2f.h s2f_transpose_scale_add_fwd/bwd)."""

    @staticmethod
    def forward(ctx, x, q, g):
        _need_cuda(x, q, g)
        x, q, g = x.contiguous(), q.contiguous(), g.contiguous()
        B, R, C = x.shape
        y = torch.empty_like(q)
        check(lib.s2f_transpose_scale_add_fwd(_ptr(x), _ptr(q), _ptr(g), _ptr(y), B, R, C, _stream()), "s2f_transpose_scale_add_fwd")
        ctx.save_for_backward(x, g)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g = ctx.saved_tensors
        gy = gy.contiguous()
        B, R, C = x.shape
        gx = torch.empty_like(x)
        gg = torch.zeros_like(g)
        check(lib.s2f_transpose_scale_add_bwd(_ptr(gy), _ptr(x), _ptr(g), _ptr(gx), _ptr(gg), B, R, C, _stream()),
              "s2f_transpose_scale_add_bwd")
        return gx, gy, gg


def transpose_scale_add(x, q, g):
    """q [..., C, R] + g[c] * x[..., R, C]^T in one pass (the layer-scaled FFN residual of the pixel decoder); falls back to
    transpose + addcmul for shapes the kernel does not take."""
    R, C = x.shape[-2:]
    if x.dtype == torch.float32 and x.is_cuda and R % 64 == 0 and C % 64 == 0 and g.data_ptr() % 16 == 0:
        return _TransposeScaleAdd.apply(x.reshape(-1, R, C), q.reshape(-1, C, R), g).view(q.shape)
    return torch.addcmul(q, transpose_last2(x).view(q.shape), g.view(*([1] * (q.dim() - 2)), C, 1))


def transpose_last2(x, skip=False):
    """x [..., R, C] (fp32, CUDA) -> contiguous [..., C, R]: the `.permute(...).contiguous()` copies around the DCNv3 sampling
    core as one tiled kernel (s2f_transpose_last2).  `skip`: -> (x^T, x') with x' = x for a second reader of x, whose gradient the
    adjoint kernel sums (cfg.FANOUT_PORTS; x' is x itself where that does not apply)."""
    lead = x.shape[:-2]
    R, C = x.shape[-2:]
    if x.dtype != torch.float32 or x.numel() == 0:
        y = x.transpose(-1, -2).contiguous()
        return (y, x) if skip else y
    if skip and cfg.FANOUT_PORTS and _SMALL_PORTS and x.is_cuda:
        y, through = _TransposeLast2.apply(x.reshape(-1, R, C), True)
        return y.view(*lead, C, R), through.view(x.shape)
    y = _TransposeLast2.apply(x.reshape(-1, R, C), False)[0].view(*lead, C, R)
    return (y, x) if skip else y


# ------------------------------------------------------------------------------------------------ reductions / fills (csrc/glue.hip)
def channel_sum(x):
    """x [N, C, L] fp32 -> [C] = x.sum((0, 2)) on s2f_channel_sum (partials stored, added in order: bit-repeatable); ATen for rows
    that are no whole 16-byte groups."""
    N, C, L = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and L % 4 == 0 and x.numel() > 0 and C < 65536):
        return x.sum((0, 2))
    x = x.contiguous()
    ws = torch.empty(C * int(lib.s2f_channel_sum_slices(N, C, L)), dtype=torch.float32, device=x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    check(lib.s2f_channel_sum(_ptr(x), N, C, L, _ptr(ws), _ptr(out), 0, _stream()), "s2f_channel_sum")
    return out


def sum_lead(x):
    """x [T, ...] fp32 -> x.sum(0) on s2f_sum_lead"""
    T = x.shape[0]
    M = x.numel() // max(T, 1)
    if not (x.is_cuda and x.dtype == torch.float32 and M % 4 == 0 and x.numel() > 0):
        return x.sum(0)
    x = x.contiguous()
    out = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
    check(lib.s2f_sum_lead(_ptr(x), T, M, _ptr(out), _stream()), "s2f_sum_lead")
    return out


def _dense_flat(x):
    """x as a flat view of the memory it covers when its elements are a permutation of one dense block (any permuted view of a
    contiguous tensor), else None"""
    if x.numel() == 0:
        return None
    dims = sorted((st, sz) for st, sz in zip(x.stride(), x.shape) if sz > 1)
    run = 1
    for st, sz in dims:
        if st != run:
            return None
        run *= sz
    return x.as_strided((x.numel(),), (1,))


class _MeanAll(torch.autograd.Function):
    """x.mean() over a whole tensor (the benchmark's headline loss terms).  Forward: s2f_sum_all over the memory x covers (a permuted
    view sums to the same value).  Backward: the constant g / n written ONCE, by s2f_fill, with x's own strides -- the gradient of
    the permuted logits view is then a plain tensor in the contraction's layout and the consumer's .contiguous() is a no-op.  (torch's
    formula materialises expand(g) / n in the view's layout and the consumer copies it into its own: two passes over 367 MB at C2.)"""

    @staticmethod
    def forward(ctx, x):
        flat = _dense_flat(x) if (x.is_cuda and x.dtype == torch.float32) else None
        ctx.meta = (x.shape, x.stride(), flat is not None)
        if flat is None or flat.data_ptr() % 16 != 0:
            ctx.meta = (x.shape, x.stride(), False)
            return x.mean()
        n = flat.numel()
        part = torch.empty(int(lib.s2f_sum_all_parts(n)), dtype=torch.float32, device=x.device)
        out = torch.empty((), dtype=torch.float32, device=x.device)
        check(lib.s2f_sum_all(_ptr(flat), n, 1.0 / n, _ptr(part), _ptr(out), _stream()), "s2f_sum_all")
        return out

    @staticmethod
    def backward(ctx, g):
        shape, stride, ours = ctx.meta
        n = 1
        for d in shape:
            n *= d
        if not ours:
            return (g / n).expand(shape)
        gx = torch.empty_strided(shape, stride, dtype=torch.float32, device=g.device)
        check(lib.s2f_fill(_ptr(gx), n, _ptr(g.contiguous()), 1.0 / n, _stream()), "s2f_fill")
        return gx


def mean_all(x):
    return _MeanAll.apply(x)


# ------------------------------------------------------------------------------------------------ 2x bilinear up-sampling
class _Up2x(torch.autograd.Function):
    """-> (up-sampled map, pass-through of x or an empty stand-in): the pass-through serves a second reader of x; the gradient it
    sends back is summed inside the adjoint kernel (s2f_upsample2x_bwd_add) instead of by an add of the autograd engine."""

    @staticmethod
    def forward(ctx, x, skip):
        _need_cuda(x)
        x_in = x
        x = x.contiguous()
        N, C, h, w = x.shape
        y = torch.empty(N, C, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        check(lib.s2f_upsample2x_fwd(_ptr(x), _ptr(y), N * C, h, w, _stream()), "s2f_upsample2x_fwd")
        ctx.shape = (N, C, h, w)
        ctx.set_materialize_grads(False)
        if skip:
            return y, x_in
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        return y, aux

    @staticmethod
    def backward(ctx, gy, gskip):
        N, C, h, w = ctx.shape
        if gy is None:
            return gskip, None
        gy = gy.contiguous()
        if gskip is not None:
            gskip = gskip.contiguous()
        gx = torch.empty(N, C, h, w, dtype=torch.float32, device=gy.device)
        check(lib.s2f_upsample2x_bwd_add(_ptr(gy), _ptr(gskip), _ptr(gx), N * C, h, w, _stream()), "s2f_upsample2x_bwd_add")
        return gx, None


def upsample_bilinear(x, size, sigmoid=False, skip=False):
    """F.interpolate(x, size, mode='bilinear', align_corners=False); the exact-2x case runs the HIP kernel.  `sigmoid`: followed by
    .sigmoid() -- inside the same pass where no gradient is wanted (the inference post-processing).
    `skip`: -> (y, x') with x' = x for a second reader of x, whose gradient the adjoint kernel then sums (cfg.FANOUT_PORTS; x' is x
    itself where that does not apply)."""
    if skip:
        h, w = x.shape[-2:]
        if cfg.FANOUT_PORTS and tuple(size) == (2 * h, 2 * w) and w % 2 == 0 and x.dim() == 4 and x.is_cuda and not sigmoid:
            return _Up2x.apply(x, True)
        return upsample_bilinear(x, size, sigmoid), x
    h, w = x.shape[-2:]
    if tuple(size) == (2 * h, 2 * w) and w % 2 == 0:
        if sigmoid and w % 4 == 0 and x.dim() == 4 and not (torch.is_grad_enabled() and x.requires_grad):
            _need_cuda(x)
            x = x.contiguous()
            N, C = x.shape[:2]
            y = torch.empty(N, C, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
            check(lib.s2f_upsample2x_sigmoid_fwd(_ptr(x), _ptr(y), N * C, h, w, _stream()), "s2f_upsample2x_sigmoid_fwd")
            return y
        y = _Up2x.apply(x, False)[0]
        return y.sigmoid() if sigmoid else y
    fallback("upsample_bilinear", f"{(h, w)} -> {tuple(size)}")
    y = torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)
    return y.sigmoid() if sigmoid else y


# ------------------------------------------------------------------------------------------------ mask losses (row f1)
class _MaskLossSums(torch.autograd.Function):
    """sums[p] = {sum s t, sum s, sum t, sum focal} over the 2x up-sampled logits of matched prediction p against its binary
    target (s2f.h s2f_mask_loss_fwd/bwd); nothing of size [P, 2h, 2w] exists forward, one such buffer backward."""

    @staticmethod
    def forward(ctx, pred, tgt, gt_index, alpha, gamma):
        _need_cuda(pred)
        pred = pred.contiguous()
        tgt = tgt.contiguous()
        P, h, w = pred.shape
        assert tgt.dtype == torch.uint8 and tgt.shape[1:] == (2 * h, 2 * w) and gt_index.dtype == torch.int64
        sums = torch.empty(P, 4, dtype=torch.float32, device=pred.device)
        check(lib.s2f_mask_loss_fwd(_ptr(pred), _ptr(tgt), _ptr(gt_index), _ptr(sums), P, h, w, alpha, gamma, _stream()),
              "s2f_mask_loss_fwd")
        ctx.save_for_backward(pred, tgt, gt_index)
        ctx.cfg = (alpha, gamma)
        return sums

    @staticmethod
    def backward(ctx, g):
        pred, tgt, gt_index = ctx.saved_tensors
        P, h, w = pred.shape
        g = g.contiguous()
        gup = torch.empty(P, 2 * h, 2 * w, dtype=torch.float32, device=pred.device)
        check(lib.s2f_mask_loss_bwd(_ptr(pred), _ptr(tgt), _ptr(gt_index), _ptr(g), _ptr(gup), P, h, w, *ctx.cfg, _stream()),
              "s2f_mask_loss_bwd")
        gp = torch.empty_like(pred)
        check(lib.s2f_upsample2x_bwd(_ptr(gup), _ptr(gp), P, h, w, _stream()), "s2f_upsample2x_bwd")
        return gp, None, None, None, None


def mask_loss_sums(pred, tgt_u8, gt_index, alpha, gamma):
    """pred [P, h, w] fp32 logits, tgt_u8 [G, 2h, 2w] uint8 0/1, gt_index [P] int64 -> [P, 4]"""
    return _MaskLossSums.apply(pred, tgt_u8, gt_index, float(alpha), float(gamma))


def mask_cost_bins(pred, seg_small, K, alpha, gamma, eps):
    """pred [B, R, hw] fp32 logits, seg_small [B, hw] uint8 label map -> [B, R, 2K + 2]: per class id the segmented sums of
    (pos - neg) and of s, then sum neg and sum s over all pixels (s2f.h s2f_mask_cost_bins; match_cost.py:289-297, :361-371)."""
    _need_cuda(pred)
    pred, seg_small = pred.contiguous(), seg_small.contiguous()
    B, R, hw = pred.shape
    assert seg_small.dtype == torch.uint8 and seg_small.shape == (B, hw) and pred.dtype == torch.float32
    out = torch.empty(B, R, 2 * K + 2, dtype=torch.float32, device=pred.device)
    check(lib.s2f_mask_cost_bins(_ptr(pred), _ptr(seg_small), _ptr(out), B, R, hw, K, alpha, gamma, eps, _stream()),
          "s2f_mask_cost_bins")
    return out


class _MaskLossSeg(torch.autograd.Function):
    """sums[(b, r)] = {sum s t, sum s, sum t, sum focal} of the 2x up-sampled logits pred[b, r] against  seg[b] == row_class[b, r]
    (rows with row_class < 0: zeros, zero gradient); s2f.h s2f_mask_loss_seg_fwd/bwd."""

    @staticmethod
    def forward(ctx, pred, seg, row_class, alpha, gamma):
        _need_cuda(pred)
        pred = pred.contiguous()
        B, R, h, w = pred.shape
        assert seg.dtype == torch.uint8 and seg.shape == (B, 2 * h, 2 * w) and seg.is_contiguous()
        assert row_class.dtype == torch.int32 and row_class.numel() == B * R and row_class.is_contiguous()
        sums = torch.empty(B * R, 4, dtype=torch.float32, device=pred.device)
        part = torch.empty(int(lib.s2f_mask_loss_seg_partials(B, R, h, w)), dtype=torch.float32, device=pred.device)
        check(lib.s2f_mask_loss_seg_fwd(_ptr(pred), _ptr(seg), _ptr(row_class), _ptr(sums), _ptr(part), B, R, h, w, alpha, gamma,
                                        _stream()), "s2f_mask_loss_seg_fwd")
        ctx.save_for_backward(pred, seg, row_class)
        ctx.cfg = (alpha, gamma)
        return sums

    @staticmethod
    def backward(ctx, g):
        pred, seg, row_class = ctx.saved_tensors
        B, R, h, w = pred.shape
        gp = torch.empty_like(pred)
        check(lib.s2f_mask_loss_seg_bwd(_ptr(pred), _ptr(seg), _ptr(row_class), _ptr(g.contiguous()), _ptr(gp), B, R, h, w, *ctx.cfg,
                                        _stream()), "s2f_mask_loss_seg_bwd")
        return gp, None, None, None, None


def mask_loss_seg(pred, seg_u8, row_class, alpha, gamma):
    """pred [B, R, h, w] fp32 logits, seg_u8 [B, 2h, 2w] uint8 label map, row_class [B * R] int32 -> sums [B * R, 4]"""
    return _MaskLossSeg.apply(pred, seg_u8, row_class, float(alpha), float(gamma))



__all__ = [n for n in dir() if not n.startswith('__')]
