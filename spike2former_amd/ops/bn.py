"""BatchNorm fused with the conv bias, the residual add and the following Q_IFNode (csrc/bn_lif.hip)."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)


# ------------------------------------------------------------------------------------------------ BN (+bias, +residual, +LIF)
class _BNAct(torch.autograd.Function):
    """t = z + conv_bias ; u = BatchNorm(t) [+ residual] ; y = Q_IFNode(u)   in two streaming kernels forward
    (statistics, apply) and two backward (reduce, apply).  Returns (u, y, v_out); unrequested ones are empty."""

    @staticmethod
    def forward(ctx, z, conv_bias, gamma, beta, residual, v_in, running_mean, running_var, nbt, training, momentum,
                eps, lif_on, want_pre, keep_v, D, vth, stats, bf16, partials=None):
        _need_cuda(z, conv_bias, gamma, beta, residual, v_in)
        z = z.contiguous()
        N, C = z.shape[0], z.shape[1]
        L = z.numel() // (N * C)
        dev = z.device
        stat = torch.empty(3 * C, dtype=torch.float32, device=dev)      # mean, rstd, BN(0) border (s2f.h)
        s = _stream()
        ws = None
        single = bool(training) and bool(lib.s2f_bn_single_pass(N, C, L))    # small map: statistics inside s2f_bn_act_fwd
        if not (training and partials is not None and partials.dim() == 3 and partials.shape[0] == C and partials.shape[2] == 2
                and partials.shape[1] == lib.s2f_bn_partials_count(N, L) and (cfg.BN_PARTIALS_SINGLE or not single)):
            partials = None          # the statistics the producing GEMM stored with z (BN_PARTIALS), when they describe this view of it
        if partials is not None:
            # the producer's per-tile partials -> the sums of the statistics pass: one small launch over P * C * 8 bytes
            BN_PARTIALS_USED[0] += 1
            ws = _take_zeroed(2 * C, dev)          # (plain stores: the workspace need not be zero)
            _time_next("bn_stats", 8 * partials.shape[1] * C)
            check(lib.s2f_bn_partials_finalize(_ptr(partials), partials.shape[1], _ptr(conv_bias), _ptr(ws), N, C, L, s),
                  "s2f_bn_partials_finalize")
        elif training and not single:
            BN_PARTIALS_USED[1] += 1
            ws = _take_zeroed(2 * C, dev)
            _time_next("bn_stats", 4 * z.numel())
            check(lib.s2f_bn_stats(_ptr(z), _ptr(conv_bias), _ptr(ws), N, C, L, s), "s2f_bn_stats")
        if residual is not None:
            residual = residual.contiguous()
        if v_in is not None:
            v_in = v_in.contiguous()
        bf16 = bool(bf16) and lif_on
        u = torch.empty_like(z) if want_pre else None
        y = torch.empty(z.shape, dtype=torch.bfloat16 if bf16 else torch.float32, device=dev) if lif_on else None
        v_out = torch.empty_like(z) if (lif_on and keep_v) else None
        need_grad = any(ctx.needs_input_grad[:5])
        # (training-mode short rows keep a per-channel mask layout of their own: s2f_bn_mask_words)
        nmask = int(lib.s2f_bn_mask_words(N, C, L)) if training else mask_words(z.numel())
        mask = torch.empty(nmask, dtype=torch.int64, device=dev) if (lif_on and need_grad) else None
        n = z.numel()
        # algorithmic bytes: read z, [read residual], [write u], [write y]  (SURVEY 8d per-element figures)
        alg = 4 * n * (1 + (residual is not None) + bool(want_pre) + bool(lif_on))
        _time_next("bn_lif_fwd" if lif_on else "bn_fwd", alg, moved=alg - (2 * n if bf16 else 0))
        check(lib.s2f_bn_act_fwd(_ptr(z), _ptr(conv_bias), _ptr(ws), _ptr(stat), _ptr(running_mean), _ptr(running_var),
                                 _ptr(nbt), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(u), _ptr(v_in), _ptr(y),
                                 _ptr(v_out), _ptr(mask), _ptr(stats), N, C, L, momentum, eps, int(training), vth, D,
                                 int(bf16), s), "s2f_bn_act_fwd")
        buf = stat
        stat, border = buf[:2 * C], buf[2 * C:]
        ctx.save_for_backward(z, conv_bias, gamma, stat, mask)
        ctx.cfg = (N, C, L, bool(training), D, vth, residual is not None, conv_bias is not None)
        ctx.set_materialize_grads(False)
        # bf16 spikes: slot 1 carries the autograd handle, slot 4 the (non-differentiable) bf16 tensor
        ydata = z.new_empty(0)
        tok2 = z.new_empty(0)          # bf16 spikes: a second handle for a second reader of the spike map (ops.Spikes.second)
        if bf16:
            ydata, y = y, _new_tok(z)
            if cfg.FANOUT_PORTS:
                tok2 = _new_tok(z)
        outs = [t if t is not None else z.new_empty(0) for t in (u, y, v_out)]
        ctx.mark_non_differentiable(border, ydata, *[o for o, t in zip(outs, (u, y, v_out)) if t is None])
        if tok2.numel() == 0:
            ctx.mark_non_differentiable(tok2)
        return tuple(outs) + (border, ydata, tok2)

    @staticmethod
    def backward(ctx, g_u, g_y, g_v, g_border, _g_ydata, g_y2):
        z, conv_bias, gamma, stat, mask = ctx.saved_tensors
        N, C, L, training, D, vth, has_res, has_bias = ctx.cfg

        def prep(g):
            return None if (g is None or g.numel() == 0) else g.contiguous()
        g_u, g_y, g_v, g_y2 = prep(g_u), prep(g_y), prep(g_v), prep(g_y2)
        if g_y is None:
            g_y, g_y2 = g_y2, None
        if g_y2 is not None and not lib.s2f_bn_bwd_ports_ok(N, C, L, int(training), D):
            g_y, g_y2 = g_y + g_y2, None          # (a kernel form without the second port: the add the autograd engine would have made)
        if g_u is None and g_y is None and g_v is None:
            return (None,) * 20
        dev = z.device
        gz = torch.empty_like(z)
        g_res = torch.empty_like(z) if (has_res and ctx.needs_input_grad[4]) else None
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = None if (training and lib.s2f_bn_single_pass(N, C, L)) else _take_zeroed(2 * C, dev)
        # read z + incoming grads, write gz [, g_residual]
        alg = 4 * z.numel() * (2 + (g_u is not None) + (g_y is not None) + (g_res is not None))
        _time_next("bn_lif_bwd" if g_y is not None else "bn_bwd", alg, moved=alg + 4 * z.numel() * (g_y2 is not None))
        check(lib.s2f_bn_act_bwd_ports(_ptr(z), _ptr(conv_bias), _ptr(stat), _ptr(gamma), _ptr(g_u), _ptr(g_y), _ptr(g_y2), _ptr(g_v),
                                       _ptr(mask), _ptr(ws), _ptr(gz), _ptr(g_res), _ptr(dgamma), _ptr(dbeta), N, C, L,
                                       int(training), vth, D, _stream()), "s2f_bn_act_bwd_ports")
        g_bias = None
        if has_bias:
            # train-mode BN removes any per-channel constant: d/d(bias) == 0 exactly -> no gradient tensor at all (None, a
            # zero-fill launch per BatchNorm otherwise); eval mode: sum(gz) = gamma * rstd * dbeta
            g_bias = None if training else gamma * stat[C:2 * C] * dbeta
        if ctx.needs_input_grad[5]:
            raise RuntimeError("gradient w.r.t. the incoming membrane is not supported by the fused BN+LIF op")
        return (gz, g_bias, dgamma, dbeta, g_res) + (None,) * 15


def bn_act(z, conv_bias, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, residual=None,
           lif=False, want_pre=True, v_in=None, keep_v=False, D=8, vth=1.0, stats=None, want_border=False, partials=None):
    """-> (u or None, y or None, v_out or None [, border]); y is a Spikes pair (bf16 when cfg.SPIKES_BF16); border [C] = BN(0)
    from the updated running statistics (BNAndPadLayer's padding value), produced by the same kernel.
    partials: the [C, P, 2] per-tile sums the producing GEMM stored for z (cfg.BN_PARTIALS; default: z's `_s2f_part` attribute)."""
    if partials is None:
        from .gemm import stats_of
        partials = stats_of(z)
    bf16 = bool(lif) and spikes_bf16_ok(D) and z.numel() % 4 == 0          # as ops.lif: consumers read bf16 spikes in 8-byte groups
    u, y, v, border, ydata, tok2 = _BNAct.apply(z, conv_bias, gamma, beta, residual, v_in, running_mean, running_var, nbt, training,
                                                momentum, eps, lif, want_pre, keep_v, D, vth, stats, bf16, partials)
    if lif:
        y = Spikes(ydata, y, tok2 if tok2.numel() else None) if bf16 else Spikes(y, None)
    out = (u if want_pre else None), (y if lif else None), (v if (lif and keep_v) else None)
    return out + (border,) if want_border else out




class _BN2Act(torch.autograd.Function):
    """u = BN2(BN1(z)) [+ residual] ; y = Q_IFNode(u)  in ONE single-pass kernel forward and one backward (s2f.h "BatchNorm o
    BatchNorm": the pair that closes every RepConv chain, sdtv2.py:280-296, 304-306).  Training mode, single-pass shapes."""

    @staticmethod
    def forward(ctx, z, g1, b1, g2, b2, residual, v_in, rm1, rv1, nbt1, mom1, eps1, rm2, rv2, nbt2, mom2, eps2, lif_on, want_pre,
                keep_v, D, vth, stats, bf16):
        _need_cuda(z, g1, b1, g2, b2, residual, v_in)
        z = z.contiguous()
        N, C = z.shape[0], z.shape[1]
        L = z.numel() // (N * C)
        dev = z.device
        stat = torch.empty(4 * C, dtype=torch.float32, device=dev)      # mean, r1, BN1(0) border, r2
        if residual is not None:
            residual = residual.contiguous()
        if v_in is not None:
            v_in = v_in.contiguous()
        bf16 = bool(bf16) and lif_on
        u = torch.empty_like(z) if want_pre else None
        y = torch.empty(z.shape, dtype=torch.bfloat16 if bf16 else torch.float32, device=dev) if lif_on else None
        v_out = torch.empty_like(z) if (lif_on and keep_v) else None
        need_grad = any(ctx.needs_input_grad[:6])
        mask = torch.empty(mask_words(z.numel()), dtype=torch.int64, device=dev) if (lif_on and need_grad) else None
        n = z.numel()
        alg = 4 * n * (1 + (residual is not None) + bool(want_pre) + bool(lif_on))
        _time_next("bn_lif_fwd" if lif_on else "bn_fwd", alg, moved=alg - (2 * n if bf16 else 0))
        check(lib.s2f_bn2_act_fwd(_ptr(z), 0, _ptr(stat), _ptr(rm1), _ptr(rv1), _ptr(nbt1), _ptr(g1), _ptr(b1), mom1, eps1, _ptr(g2),
                                  _ptr(b2), _ptr(rm2), _ptr(rv2), _ptr(nbt2), mom2, eps2, _ptr(residual), _ptr(u), _ptr(v_in), _ptr(y),
                                  _ptr(v_out), _ptr(mask), _ptr(stats), N, C, L, vth, D, int(bf16), _stream()), "s2f_bn2_act_fwd")
        ctx.save_for_backward(z, g1, g2, stat, mask)
        ctx.cfg = (N, C, L, D, vth, residual is not None, eps2)
        ctx.set_materialize_grads(False)
        ydata = z.new_empty(0)
        if bf16:
            ydata, y = y, _new_tok(z)
        outs = [t if t is not None else z.new_empty(0) for t in (u, y, v_out)]
        ctx.mark_non_differentiable(ydata, *[o for o, t in zip(outs, (u, y, v_out)) if t is None])
        return tuple(outs) + (ydata,)

    @staticmethod
    def backward(ctx, g_u, g_y, g_v, _g_ydata):
        z, g1, g2, stat, mask = ctx.saved_tensors
        N, C, L, D, vth, has_res, eps2 = ctx.cfg

        def prep(g):
            return None if (g is None or g.numel() == 0) else g.contiguous()
        g_u, g_y, g_v = prep(g_u), prep(g_y), prep(g_v)
        if g_u is None and g_y is None and g_v is None:
            return (None,) * 24
        dev = z.device
        gz = torch.empty_like(z)
        g_res = torch.empty_like(z) if (has_res and ctx.needs_input_grad[5]) else None
        d = torch.empty(4, C, dtype=torch.float32, device=dev)          # dgamma1, dbeta1, dgamma2, dbeta2
        alg = 4 * z.numel() * (2 + (g_u is not None) + (g_y is not None) + (g_res is not None))
        _time_next("bn_lif_bwd" if g_y is not None else "bn_bwd", alg)
        check(lib.s2f_bn2_act_bwd(_ptr(z), 0, _ptr(stat), _ptr(g1), _ptr(g2), eps2, _ptr(g_u), _ptr(g_y), _ptr(g_v), _ptr(mask), _ptr(gz),
                                  _ptr(g_res), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(d[3]), N, C, L, vth, D, _stream()),
              "s2f_bn2_act_bwd")
        if ctx.needs_input_grad[6]:
            raise RuntimeError("gradient w.r.t. the incoming membrane is not supported by the fused BN+LIF op")
        return (gz, d[0], d[1], d[2], d[3], g_res) + (None,) * 18


def bn2_act_ok(z):
    N, C = z.shape[0], z.shape[1]
    return bool(cfg.BN2_FUSED and z.is_cuda and z.numel() and lib.s2f_bn2_fused_ok(N, C, z.numel() // (N * C)))


def bn2_act(z, bn1, bn2, residual=None, lif=False, want_pre=True, v_in=None, keep_v=False, D=8, vth=1.0, stats=None):
    """bn1 / bn2: objects with weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps (nn.BatchNorm or the
    concatenated twins of the batched q / k / v chain) -> (u or None, y (Spikes) or None, v_out or None)"""
    bf16 = bool(lif) and spikes_bf16_ok(D) and z.numel() % 4 == 0
    u, y, v, ydata = _BN2Act.apply(z, bn1.weight, bn1.bias, bn2.weight, bn2.bias, residual, v_in, bn1.running_mean, bn1.running_var,
                                   bn1.num_batches_tracked, bn1.momentum, bn1.eps, bn2.running_mean, bn2.running_var,
                                   bn2.num_batches_tracked, bn2.momentum, bn2.eps, lif, want_pre, keep_v, D, vth, stats, bf16)
    if lif:
        y = Spikes(ydata, y) if bf16 else Spikes(y, None)
    return (u if want_pre else None), (y if lif else None), (v if (lif and keep_v) else None)


class _ScaleAffine(torch.autograd.Function):
    """(gamma * s, beta * s) in one launch, gradients of all three in one launch (s2f.h s2f_scale_affine_*)."""

    @staticmethod
    def forward(ctx, gamma, beta, s):
        _need_cuda(gamma, beta, s)
        gamma, beta, s = gamma.contiguous(), beta.contiguous(), s.contiguous()
        w, b = torch.empty_like(gamma), torch.empty_like(beta)
        check(lib.s2f_scale_affine_fwd(_ptr(gamma), _ptr(beta), _ptr(s), _ptr(w), _ptr(b), gamma.numel(), _stream()),
              "s2f_scale_affine_fwd")
        ctx.save_for_backward(gamma, beta, s)
        ctx.set_materialize_grads(False)
        return w, b

    @staticmethod
    def backward(ctx, gw, gb):
        gamma, beta, s = ctx.saved_tensors
        if gw is None and gb is None:
            return None, None, None
        dg, db, ds = torch.empty_like(gamma), torch.empty_like(beta), torch.empty_like(s)
        check(lib.s2f_scale_affine_bwd(_ptr(None if gw is None else gw.contiguous()), _ptr(None if gb is None else gb.contiguous()),
                                       _ptr(gamma), _ptr(beta), _ptr(s), _ptr(dg), _ptr(db), _ptr(ds), gamma.numel(), _stream()),
              "s2f_scale_affine_bwd")
        return dg, db, ds


def scale_affine(gamma, beta, s):
    return _ScaleAffine.apply(gamma, beta, s)



__all__ = [n for n in dir() if not n.startswith('__')]
