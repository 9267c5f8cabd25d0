"""Q_IFNode as autograd ops (csrc/lif.hip): one call, the fused two-neuron form of the decoder's keys / values, T chained calls."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)
from .misc import channel_sum, sum_lead


# ------------------------------------------------------------------------------------------------ LIF
class _LIF(torch.autograd.Function):
    """One Q_IFNode call (neuron.py:166-197 + surrogate.py:522-538).  Saves 1 bit/element for backward.
    Outputs (y or its handle, membrane, bf16 spikes, second handle, pass-through of x): the second handle serves a second consumer of
    the spike map, the pass-through a residual branch that reads the neuron's input beside it (`x + f(Q_IFNode(x))`); the backward
    kernel sums the gradients arriving on them (s2f_lif_bwd_ports) -- the sums autograd's engine would launch an add for."""

    @staticmethod
    def forward(ctx, x, v_in, D, vth, keep_v, stats, bf16, skip):
        _need_cuda(x, v_in)
        x_in = x
        x = x.contiguous()
        if v_in is not None:
            v_in = v_in.contiguous()
        n = x.numel()
        y = torch.empty(x.shape, dtype=torch.bfloat16 if bf16 else torch.float32, device=x.device)
        v_out = torch.empty_like(x) if keep_v else None
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        mask = torch.empty(mask_words(n), dtype=torch.int64, device=x.device) if need_grad else None
        _time_next("lif_fwd", 8 * n, moved=(6 if bf16 else 8) * n)
        check(lib.s2f_lif_fwd(_ptr(x), _ptr(v_in), _ptr(y), _ptr(v_out), _ptr(mask), 0, _ptr(stats), n, vth, D, int(bf16),
                              _stream()), "s2f_lif_fwd")
        ctx.save_for_backward(mask)
        ctx.D, ctx.vth, ctx.has_v = D, vth, v_in is not None
        ctx.set_materialize_grads(False)          # no zero-filled stand-ins for the gradients of unused / bf16 outputs
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        if v_out is None:
            v_out = x.new_empty(0)
            ctx.mark_non_differentiable(v_out)
        through = x_in if skip else aux          # (an input returned as it is: autograd hands out a view of it)
        if bf16:                       # (autograd handle, membrane, bf16 spikes, second handle, pass-through)
            ctx.mark_non_differentiable(y)
            return _new_tok(x), v_out, y, _new_tok(x), through
        return y, v_out, aux, aux, through

    @staticmethod
    def backward(ctx, gy, gv, _g2, gy2, gskip):
        (mask,) = ctx.saved_tensors
        if gy is None and gy2 is not None:
            gy, gy2 = gy2, None
        if gy is None and gv is None:
            return (gskip,) + (None,) * 7
        if gy is None:                              # only the membrane carries a gradient
            gy = torch.zeros_like(gv)
        gy = gy.contiguous()
        if gv is not None and gv.numel() != gy.numel():
            gv = None
        if gv is not None:
            gv = gv.contiguous()
        if gy2 is not None:
            gy2 = gy2.contiguous()
        fold = gskip is not None and gv is None and not ctx.has_v          # (a membrane gradient shares gx: keep the pass-through apart)
        if gskip is not None:
            gskip = gskip.contiguous()
        gx = torch.empty_like(gy)
        _time_next("lif_bwd", 12 * gy.numel(), moved=(12 + 4 * (gy2 is not None) + 4 * bool(fold)) * gy.numel())
        check(lib.s2f_lif_bwd_ports(_ptr(gy), _ptr(gy2), _ptr(gv), _ptr(mask), _ptr(gskip) if fold else 0, _ptr(gx), gy.numel(), ctx.vth,
                                    ctx.D, _stream()), "s2f_lif_bwd_ports")
        gx_in = gx if (gskip is None or fold) else gx + gskip.view(gx.shape)
        return gx_in, (gx if ctx.has_v else None), None, None, None, None, None, None


class _LIFLeaky(torch.autograd.Function):
    """One LIFNode call: leaky charge (neuron.py:803-814) in front of the fork's quantised firing rule (s2f.h s2f_lif_leaky_fwd)."""

    @staticmethod
    def forward(ctx, x, v_in, D, vth, tau, decay_input, keep_v, stats):
        _need_cuda(x, v_in)
        x = _aligned16(x)
        if v_in is not None:
            v_in = _aligned16(v_in)
        n = x.numel()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        v_out = torch.empty(x.shape, dtype=torch.float32, device=x.device) if keep_v else None
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        mask = torch.empty(mask_words(n), dtype=torch.int64, device=x.device) if need_grad else None
        check(lib.s2f_lif_leaky_fwd(_ptr(x), _ptr(v_in), _ptr(y), _ptr(v_out), _ptr(mask), _ptr(stats), n, vth, D, tau,
                                    int(bool(decay_input)), 0, _stream()), "s2f_lif_leaky_fwd")
        ctx.save_for_backward(mask)
        ctx.cfg = (D, vth, tau, int(bool(decay_input)), v_in is not None)
        ctx.set_materialize_grads(False)
        if v_out is None:
            v_out = x.new_empty(0)
            ctx.mark_non_differentiable(v_out)
        return y, v_out

    @staticmethod
    def backward(ctx, gy, gv):
        (mask,) = ctx.saved_tensors
        D, vth, tau, di, has_v = ctx.cfg
        if gy is None and gv is None:
            return (None,) * 8
        if gy is None:
            gy = torch.zeros_like(gv)
        gy = _aligned16(gy)
        if gv is not None and gv.numel() != gy.numel():
            gv = None
        if gv is not None:
            gv = _aligned16(gv)
        gx = torch.empty(gy.shape, dtype=torch.float32, device=gy.device)
        gvi = torch.empty(gy.shape, dtype=torch.float32, device=gy.device) if has_v else None
        check(lib.s2f_lif_leaky_bwd(_ptr(gy), _ptr(gv), _ptr(mask), _ptr(gx), _ptr(gvi), gy.numel(), vth, D, tau, di, _stream()),
              "s2f_lif_leaky_bwd")
        return gx, gvi, None, None, None, None, None, None


def lif_leaky(x, v_in=None, D=8, vth=1.0, tau=2.0, decay_input=True, keep_v=True, stats=None):
    """-> (y fp32, v_out or None): LIFNode (neuron.py:694-814) under the fork's BaseNode.forward"""
    y, v = _LIFLeaky.apply(x, v_in, int(D), float(vth), float(tau), bool(decay_input), bool(keep_v), stats)
    return y, (v if keep_v else None)


def lif(x, v_in=None, D=8, vth=1.0, keep_v=True, stats=None, spikes=False, skip=False):
    """-> (y, v_out or None); `spikes`: y as a Spikes pair (bf16 when cfg.SPIKES_BF16 and the size allows 8-byte stores) that carries a
    spare handle for a second consumer; `skip`: -> (y, v_out, x'), x' = x for a residual branch -- its gradient is summed inside the
    neuron's backward kernel (cfg.FANOUT_PORTS)"""
    bf16 = bool(spikes) and spikes_bf16_ok(D) and x.numel() % 4 == 0 and x.numel() > 0
    ports = cfg.FANOUT_PORTS
    fold = bool(skip) and ports and v_in is None and x.numel() > 0
    y, v, data, tok2, through = _LIF.apply(x, v_in, D, vth, keep_v, stats, bf16, fold)
    if spikes:
        y = Spikes(data, y, tok2 if ports else None) if bf16 else Spikes(y, None)
    if skip:
        return y, (v if keep_v else None), (through if fold else x)
    return y, (v if keep_v else None)


class _Sum2LIF(torch.autograd.Function):
    """The decoder's value / key neurons on  a = x + e[c]  and  a + pos[b]  in one pass, neither sum materialised
    (maskformer_head.py:535-540 + transformer.py:626-629; s2f.h s2f_sum2_lif_fwd).  Reset, stateless neurons only.
    Outputs (key handle, value handle, bf16 key spikes, bf16 value spikes, second key handle, second value handle, pass-through of
    x): see _LIF -- a memory level's spikes are read by two decoder layers, the decoder's `query + attention(query)` reads the query
    beside its neurons; the gradients arriving on the extra outputs are summed in the backward kernel (s2f_sum2_lif_bwd_ports)."""

    @staticmethod
    def forward(ctx, x, e, pos, B, D, vth, bf16, skip):
        _need_cuda(x, e, pos)
        x_in = x
        x, e, pos = x.contiguous(), e.contiguous(), pos.contiguous()
        TB, C, L = x.shape
        n = x.numel()
        dt = torch.bfloat16 if bf16 else torch.float32
        yk, yv = torch.empty(x.shape, dtype=dt, device=x.device), torch.empty(x.shape, dtype=dt, device=x.device)
        mk = torch.empty(mask_words(n), dtype=torch.int64, device=x.device)
        mv = torch.empty_like(mk)
        _time_next("lif_fwd", 12 * n, moved=(8 if bf16 else 12) * n)          # read x, write two spike maps (pos is 1/T of a map)
        check(lib.s2f_sum2_lif_fwd(_ptr(x), _ptr(e), _ptr(pos), _ptr(yk), _ptr(yv), _ptr(mk), _ptr(mv), TB, B, C, L, vth, D,
                                   int(bf16), _stream()), "s2f_sum2_lif_fwd")
        ctx.save_for_backward(mk, mv)
        ctx.D = D
        ctx.tb = (TB // B, B)
        ctx.set_materialize_grads(False)
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        through = x_in if skip else aux
        if bf16:
            ctx.mark_non_differentiable(yk, yv)
            return _new_tok(x), _new_tok(x), yk, yv, _new_tok(x), _new_tok(x), through
        return yk, yv, aux, aux, aux, aux, through

    @staticmethod
    def backward(ctx, gk, gv, _a, _b, gk2, gv2, gskip):
        mk, mv = ctx.saved_tensors
        if gk is None and gk2 is not None:
            gk, gk2 = gk2, None
        if gv is None and gv2 is not None:
            gv, gv2 = gv2, None
        if gk is None and gv is None:
            return (gskip,) + (None,) * 7
        # (either gradient may be missing -- a neuron whose spikes nobody differentiated: the kernel takes NULL for zero)
        gk, gv, gk2, gv2 = (None if g is None else g.contiguous() for g in (gk, gv, gk2, gv2))
        like = gk if gk is not None else gv
        gx = torch.empty_like(like)
        want_pos = ctx.needs_input_grad[2] and gk is not None
        gxk = torch.empty_like(like) if want_pos else None          # STE_k(g_k) alone: its sum over the time steps is d/d(pos)
        # the pass-through's gradient belongs to x alone: with a level embedding to differentiate (ge = channel sums of the NEURONS'
        # gradient) it stays a separate add
        fold = gskip is not None and not ctx.needs_input_grad[1]
        if gskip is not None:
            gskip = gskip.contiguous()
        _time_next("lif_bwd", 12 * like.numel(), moved=(12 + 4 * (gk2 is not None) + 4 * (gv2 is not None) + 4 * bool(fold)) * like.numel())
        check(lib.s2f_sum2_lif_bwd_ports(_ptr(gk), _ptr(gk2), _ptr(gv), _ptr(gv2), _ptr(mk), _ptr(mv), _ptr(gskip) if fold else 0,
                                         _ptr(gx), _ptr(gxk), like.numel(), ctx.D, _stream()), "s2f_sum2_lif_bwd_ports")
        ge = channel_sum(gx) if ctx.needs_input_grad[1] else None
        gpos = None
        if ctx.needs_input_grad[2]:
            T, B = ctx.tb
            gpos = sum_lead(gxk.view(T, B, *like.shape[1:])) if want_pos else torch.zeros(B, *like.shape[1:], dtype=like.dtype, device=like.device)
        gx_in = gx if (gskip is None or fold) else gx + gskip.view(gx.shape)
        return gx_in, ge, gpos, None, None, None, None, None


def sum2_lif(x, e, pos, B, D=8, vth=1.0, skip=False):
    """x [T*B, C, L], e [C], pos [B, C, L] -> (Q_IFNode(x + e + pos), Q_IFNode(x + e))  (key spikes, value spikes), as Spikes with a
    spare handle each; `skip`: -> (key, value, x') with x' = x for a residual branch (see ops.lif)."""
    bf16 = spikes_bf16_ok(D)
    ports = cfg.FANOUT_PORTS
    fold = bool(skip) and ports
    hk, hv, dk, dv, hk2, hv2, through = _Sum2LIF.apply(x, e, pos, B, D, vth, bf16, fold)
    out = (Spikes(dk, hk, hk2 if ports else None), Spikes(dv, hv, hv2 if ports else None)) if bf16 else (Spikes(hk), Spikes(hv))
    return out + ((through if fold else x),) if skip else out


class _LIFSeq(torch.autograd.Function):
    """T chained stateful calls on one neuron, membrane in registers (cal_firing_num.py:203-225 across images)."""

    @staticmethod
    def forward(ctx, x_seq, v0, D, vth, stats):
        _need_cuda(x_seq, v0)
        x_seq = x_seq.contiguous()
        T = x_seq.shape[0]
        n = x_seq[0].numel()
        y = torch.empty_like(x_seq)
        vT = torch.empty_like(x_seq[0])
        mask = torch.empty(T * mask_words(n), dtype=torch.int64, device=x_seq.device)
        check(lib.s2f_lif_seq_fwd(_ptr(x_seq), _ptr(None if v0 is None else v0.contiguous()), _ptr(y), _ptr(vT),
                                  _ptr(mask), _ptr(stats), T, n, vth, D, _stream()), "s2f_lif_seq_fwd")
        ctx.save_for_backward(mask)
        ctx.D, ctx.vth, ctx.has_v0, ctx.T, ctx.n = D, vth, v0 is not None, T, n
        return y, vT

    @staticmethod
    def backward(ctx, gy, gvT):
        (mask,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        gv0 = torch.empty_like(gy[0]) if ctx.has_v0 else None
        check(lib.s2f_lif_seq_bwd(_ptr(gy), _ptr(None if gvT is None else gvT.contiguous()), _ptr(mask), _ptr(gx),
                                  _ptr(gv0), ctx.T, ctx.n, ctx.vth, ctx.D, _stream()), "s2f_lif_seq_bwd")
        return gx, gv0, None, None, None


def lif_seq(x_seq, v0=None, D=8, vth=1.0, stats=None):
    return _LIFSeq.apply(x_seq, v0, D, vth, stats)



__all__ = [n for n in dir() if not n.startswith('__')]
