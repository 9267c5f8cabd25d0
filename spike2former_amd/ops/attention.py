"""The spike-driven attention core (csrc/sdsa.hip) and the DCNv3 sampling core (csrc/dcnv3.hip)."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)


# ------------------------------------------------------------------------------------------------ attention core
class _SDSA(torch.autograd.Function):
    """o = scale * q (k^T v) on channel-major spikes [TB, C, N] (sdtv2.py:335-339; transformer.py:253-274)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        _need_cuda(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        TB, C, Nq = q.shape
        Nk = k.shape[2]
        d = C // heads
        o = torch.empty_like(q)
        kv = torch.empty(TB, heads, d, d, dtype=torch.float32, device=q.device)
        check(lib.s2f_sdsa_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(kv), TB, heads, d, Nq, Nk, scale, _stream()),
              "s2f_sdsa_fwd")
        ctx.save_for_backward(q, k, v, kv)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, kv = ctx.saved_tensors
        go = go.contiguous()
        TB, C, Nq = q.shape
        Nk = k.shape[2]
        d = C // ctx.heads
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty_like(kv)
        check(lib.s2f_sdsa_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(kv), _ptr(go), _ptr(gq), _ptr(gk), _ptr(gv), _ptr(ws),
                               TB, ctx.heads, d, Nq, Nk, ctx.scale, _stream()), "s2f_sdsa_bwd")
        return gq, gk, gv, None, None


class _SDSASpikes(torch.autograd.Function):
    """The attention core on bf16 spike operands (s2f.h s2f_sdsa_*_bf16), optionally fused with the neuron that follows it
    (s2f_sdsa_lif_fwd_bf16: kv on the matrix cores, o never written, spikes + mask + counters from the epilogue).
    packed: `qd` is the stacked q|k|v map [TB, 3C, N] of the batched projection chain (one autograd handle `qt`, one
    [TB, 3C, N] gradient -- no split / stack copies); otherwise three [TB, C, N*] tensors with a handle each."""

    @staticmethod
    def forward(ctx, qd, kd, vd, qt, kt, vt, heads, scale, packed, fuse, D, vth, stats):
        qd = qd.contiguous()
        TB = qd.shape[0]
        if packed:
            C, Nq = qd.shape[1] // 3, qd.shape[2]
            Nk = Nq
            qs = ks = vs = 3 * C * Nq
            qp = qd.data_ptr()
            kp, vp = qp + 2 * C * Nq, qp + 4 * C * Nq
        else:
            kd, vd = kd.contiguous(), vd.contiguous()
            C, Nq, Nk = qd.shape[1], qd.shape[2], kd.shape[2]
            qs, ks, vs = C * Nq, C * Nk, C * Nk
            qp, kp, vp = qd.data_ptr(), kd.data_ptr(), vd.data_ptr()
        d = C // heads
        dev = qd.device
        kv = torch.empty(TB, heads, d, d, dtype=torch.float32, device=dev)
        ctx.cfg = (TB, C, Nq, Nk, d, heads, scale, packed, fuse, D, (qs, ks, vs))
        ctx.ptrs_of = (kp - qp, vp - qp)
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad[3:6])
        if fuse:
            y = torch.empty(TB, C, Nq, dtype=torch.bfloat16, device=dev)
            mask = torch.empty(mask_words(y.numel()), dtype=torch.int64, device=dev) if need else None
            _time_next("sdsa_lif_fwd", 4 * TB * C * (2 * Nk + 2 * Nq), 4 * TB * heads * d * d * (Nk + Nq),
                       moved=2 * TB * C * (2 * Nk + 2 * Nq))
            check(lib.s2f_sdsa_lif_fwd_bf16(qp, kp, vp, qs, ks, vs, _ptr(y), _ptr(mask), _ptr(stats), _ptr(kv), TB, heads, d, Nq,
                                            scale, vth, D, _stream()), "s2f_sdsa_lif_fwd_bf16")
            ctx.save_for_backward(qd, kd, vd, kv, mask)
            ctx.mark_non_differentiable(y)
            return _new_tok(y), y
        o = torch.empty(TB, C, Nq, dtype=torch.float32, device=dev)
        check(lib.s2f_sdsa_fwd_bf16(qp, kp, vp, qs, ks, vs, _ptr(o), _ptr(kv), TB, heads, d, Nq, Nk, scale, _stream()),
              "s2f_sdsa_fwd_bf16")
        ctx.save_for_backward(qd, kd, vd, kv, None)
        aux = o.new_empty(0)
        ctx.mark_non_differentiable(aux)
        return o, aux

    @staticmethod
    def backward(ctx, g, _g1):
        qd, kd, vd, kv, mask = ctx.saved_tensors
        TB, C, Nq, Nk, d, heads, scale, packed, fuse, D, (qs, ks, vs) = ctx.cfg
        if g is None:
            return (None,) * 13
        g = g.contiguous()
        dev = g.device
        qp = qd.data_ptr()
        if packed:
            G = torch.empty(TB, 3 * C, Nq, dtype=torch.float32, device=dev)
            gq, gk, gv = G.data_ptr(), G.data_ptr() + 4 * C * Nq, G.data_ptr() + 8 * C * Nq
            gs = (3 * C * Nq,) * 3
            kp, vp = qp + ctx.ptrs_of[0], qp + ctx.ptrs_of[1]
            out = (G, None, None)
        else:
            Gq = torch.empty(TB, C, Nq, dtype=torch.float32, device=dev)
            Gk = torch.empty(TB, C, Nk, dtype=torch.float32, device=dev)
            Gv = torch.empty(TB, C, Nk, dtype=torch.float32, device=dev)
            gq, gk, gv = Gq.data_ptr(), Gk.data_ptr(), Gv.data_ptr()
            gs = (C * Nq, C * Nk, C * Nk)
            kp, vp = kd.data_ptr(), vd.data_ptr()
            out = (Gq, Gk, Gv)
        ws = torch.empty_like(kv)
        check(lib.s2f_sdsa_bwd_bf16(qp, kp, vp, qs, ks, vs, _ptr(kv), _ptr(g), _ptr(mask) if fuse else 0, D, gq, gk, gv, *gs,
                                    _ptr(ws), TB, heads, d, Nq, Nk, scale, _stream()), "s2f_sdsa_bwd_bf16")
        return (None, None, None) + out + (None,) * 7


class _SDSAMasked(torch.autograd.Function):
    """out = (scale q k^T).masked_fill(mask, 0) v on channel-major fp32 maps (transformer.py:259-272, 343-355; s2f.h
    s2f_sdsa_masked_fwd / _bwd).  mask: uint8 [B, heads, Nq, Nk], shared over the time steps."""

    @staticmethod
    def forward(ctx, q, k, v, mask, heads, scale):
        _need_cuda(q, k, v)
        q, k, v, mask = q.contiguous(), k.contiguous(), v.contiguous(), mask.contiguous()
        TB, C, Nq = q.shape
        Nk = k.shape[2]
        B = mask.shape[0]
        o = torch.empty_like(q)
        check(lib.s2f_sdsa_masked_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(o), TB, B, heads, C // heads, Nq, Nk, scale,
                                      _stream()), "s2f_sdsa_masked_fwd")
        ctx.save_for_backward(q, k, v, mask)
        ctx.cfg = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, mask = ctx.saved_tensors
        heads, scale = ctx.cfg
        go = go.contiguous()
        TB, C, Nq = q.shape
        Nk = k.shape[2]
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(lib.s2f_sdsa_masked_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(mask), _ptr(go), _ptr(gq), _ptr(gk), _ptr(gv), TB, mask.shape[0],
                                      heads, C // heads, Nq, Nk, scale, _stream()), "s2f_sdsa_masked_bwd")
        return gq, gk, gv, None, None, None


def sdsa_masked(q, k, v, attn_mask, heads, scale, B):
    """The attention core with an `attn_mask` (True / non-zero = the score is replaced by 0).  q [T*B, C, Nq], k, v [T*B, C, Nk]: fp32
    tensors or Spikes.  attn_mask: b * heads * Nq * Nk booleans in any shape -- the reference reshapes to (t, heads, nq, nk) and
    broadcasts against (t, b, heads, nq, nk), which only runs for t == b and then applies mask[b, h] to every time step; that is the
    semantics here, for any t.  -> fp32 [T*B, C, Nq]."""
    TB, C, Nq = q.shape
    Nk = k.shape[2]
    if C // heads > 64:
        raise NotImplementedError("sdsa_masked: head dimension > 64")
    if attn_mask.numel() != B * heads * Nq * Nk:
        raise RuntimeError(f"attn_mask has {attn_mask.numel()} elements; (batch {B}) x (heads {heads}) x (nq {Nq}) x (nk {Nk}) wanted")
    m = attn_mask.reshape(B, heads, Nq, Nk).to(device=q.device, dtype=torch.uint8)
    return _SDSAMasked.apply(spikes_float(q), spikes_float(k), spikes_float(v), m, heads, float(scale))


def sdsa(q, k, v, heads, scale, lif=None):
    """o = scale * q (k^T v) on channel-major spike maps.  q, k, v: fp32 tensors or Spikes.  `lif`: the Q_IFNode applied to o
    (the attention's attn_spike); given, the result is its spike map (Spikes) -- from ONE fused kernel when the neuron starts
    from a reset membrane and keeps none and the shapes allow it (backbone self-attention), else core + neuron."""
    bf = all(isinstance(t, Spikes) and t.tok is not None for t in (q, k, v))
    bf = bf and q.shape[2] % 4 == 0 and k.shape[2] % 4 == 0 and q.shape[1] // heads <= 64
    if not bf:
        o = _SDSA.apply(spikes_float(q), spikes_float(k), spikes_float(v), heads, scale)
        return o if lif is None else lif.fire(o)
    return _sdsa_spikes(q.data, k.data, v.data, q.tok, k.tok, v.tok, heads, scale, False, lif)


def sdsa_packed(y, heads, scale, lif=None):
    """The same with q | k | v = the three channel ranges of one spike map y [TB, 3C, N] (the batched projection chain)."""
    if not (isinstance(y, Spikes) and y.tok is not None and y.shape[2] % 4 == 0 and y.shape[1] // 3 // heads <= 64):
        q, k, v = split3(spikes_float(y))
        return sdsa(q, k, v, heads, scale, lif)
    return _sdsa_spikes(y.data, None, None, y.tok, None, None, heads, scale, True, lif)


def _sdsa_spikes(qd, kd, vd, qt, kt, vt, heads, scale, packed, lif):
    C = qd.shape[1] // 3 if packed else qd.shape[1]
    Nq = qd.shape[2]
    Nk = Nq if packed else kd.shape[2]
    pure = (lif is not None and isinstance(lif.v, float) and not lif.keep_membrane
            and not lif._forward_hooks and not lif._forward_pre_hooks)          # hooks want the module call
    fuse = pure and Nq == Nk and Nq % 256 == 0 and (C * Nq) % 8 == 0 and spikes_bf16_ok(lif.D)
    no_grad = not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (qt, kt, vt)))
    if (pure and no_grad and not packed and Nq % 4 == 0 and Nk % 4 == 0 and spikes_bf16_ok(lif.D) and C // heads <= 64
            and not (Nq == Nk and Nq % 256 == 0)):
        # inference on maps the masked kernel does not take (the decoder's 100 queries): attention core + neuron as one launch pair,
        # o never written (s2f_sdsa_lif_fwd_bf16_nomask)
        qd, kd, vd = qd.contiguous(), kd.contiguous(), vd.contiguous()
        TB = qd.shape[0]
        d = C // heads
        y = torch.empty(TB, C, Nq, dtype=torch.bfloat16, device=qd.device)
        kv = torch.empty(TB, heads, d, d, dtype=torch.float32, device=qd.device)
        if lif.stats is not None:
            lif.stats_elems += TB * C * Nq
        check(lib.s2f_sdsa_lif_fwd_bf16_nomask(_ptr(qd), _ptr(kd), _ptr(vd), C * Nq, C * Nk, C * Nk, _ptr(y), _ptr(lif.stats),
                                               _ptr(kv), TB, heads, d, Nq, Nk, scale, lif.v_threshold, lif.D, _stream()),
              "s2f_sdsa_lif_fwd_bf16_nomask")
        lif.v = 0.0
        return Spikes(y, _new_tok(y))
    if fuse and lif.stats is not None:
        lif.stats_elems += qd.shape[0] * C * Nq
    o, ydata = _SDSASpikes.apply(qd, kd, vd, qt, kt, vt, heads, scale, packed, fuse, lif.D if fuse else 8,
                                 lif.v_threshold if fuse else 1.0, lif.stats if fuse else None)
    if fuse:
        lif.v = 0.0
        return Spikes(ydata, o)
    return o if lif is None else lif.fire(o)


# ------------------------------------------------------------------------------------------------ DCNv3 core
class _DCNv3(torch.autograd.Function):
    """dcnv3_core_pytorch (ops_dcnv3/functions/dcnv3_func.py:147-189), NHWC in/out."""

    @staticmethod
    def forward(ctx, x, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, G, Cg, offset_scale):
        _need_cuda(x, offset, mask)
        x, offset, mask = x.contiguous(), offset.contiguous(), mask.contiguous()
        N, H, W, _ = x.shape
        Ho, Wo = offset.shape[1], offset.shape[2]
        out = torch.empty(N, Ho, Wo, G * Cg, dtype=torch.float32, device=x.device)
        geo = (N, H, W, G, Cg, kh, kw, sh, sw, ph, pw, dh, dw)
        check(lib.s2f_dcnv3_fwd(_ptr(x), _ptr(offset), _ptr(mask), _ptr(out), *geo, offset_scale, _stream()),
              "s2f_dcnv3_fwd")
        ctx.save_for_backward(x, offset, mask)
        ctx.geo, ctx.osc = geo, offset_scale
        return out

    @staticmethod
    def backward(ctx, go):
        x, offset, mask = ctx.saved_tensors
        go = go.contiguous()
        N, H, W, G, Cg = ctx.geo[:5]
        Ho, Wo = offset.shape[1], offset.shape[2]
        gx = torch.empty_like(x)                      # s2f_dcnv3_bwd overwrites all three gradients
        goff, gm = torch.empty_like(offset), torch.empty_like(mask)
        check(lib.s2f_dcnv3_bwd(_ptr(x), _ptr(offset), _ptr(mask), _ptr(go), _ptr(gx), _ptr(goff), _ptr(gm), *ctx.geo,
                                ctx.osc, _stream()), "s2f_dcnv3_bwd")
        return (gx, goff, gm) + (None,) * 11


def dcnv3_core(x, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, G, Cg, offset_scale):
    """`mask` may arrive as a Spikes pair (the module's mask neuron): the gather reads it as fp32."""
    return _DCNv3.apply(x, offset, spikes_float(mask), kh, kw, sh, sw, ph, pw, dh, dw, G, Cg, float(offset_scale))



__all__ = [n for n in dir() if not n.startswith('__')]
