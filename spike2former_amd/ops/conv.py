"""Depthwise stencils (csrc/dwconv.hip) and dense k x k convolutions lowered to the GEMM kernels."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)
from .gemm import *          # noqa: F401,F403


# ------------------------------------------------------------------------------------------------ depthwise conv
class _DWConv(torch.autograd.Function):
    """Depthwise KxK, stride 1 (nn.Conv2d(groups=C)); `border` = per-channel constant padding value (detached).
    x: fp32, or a bf16 spike map with its autograd handle `tok`."""

    @staticmethod
    def forward(ctx, x, tok, w, border, pad):
        _need_cuda(w, border, spikes=x)
        x, w = x.contiguous(), w.contiguous()
        N, C, H, W = x.shape
        K = w.shape[-1]
        Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
        y = torch.empty(N, C, Ho, Wo, dtype=torch.float32, device=x.device)
        if border is not None:
            border = border.contiguous()
        xb = int(x.dtype == torch.bfloat16)
        check(lib.s2f_dwconv_fwd(_ptr(x), _ptr(w), _ptr(border), _ptr(y), N, C, H, W, K, pad, xb, _stream()),
              "s2f_dwconv_fwd")
        ctx.save_for_backward(x, w, border)
        ctx.pad, ctx.has_tok = pad, tok is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, border = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, H, W = x.shape
        K = w.shape[-1]
        gx = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
            check(lib.s2f_dwconv_bwd_input(_ptr(gy), _ptr(w), _ptr(gx), N, C, H, W, K, ctx.pad, _stream()),
                  "s2f_dwconv_bwd_input")
        if ctx.needs_input_grad[2]:
            sink = _sink_for(w)
            gw = torch.empty_like(w) if sink is None else None
            side = _wgrad_stream(sink, gy, x, border)
            check(lib.s2f_dwconv_bwd_weight(_ptr(x), _ptr(border), _ptr(gy), _ptr(gw if sink is None else sink), N, C, H, W,
                                            K, ctx.pad, int(sink is not None), int(x.dtype == torch.bfloat16),
                                            side.cuda_stream if side is not None else _stream()), "s2f_dwconv_bwd_weight")
        return _grad_pair(ctx.has_tok, gx) + (gw, None, None)


def dwconv_bn_lif_eval(x, w, pad, running_mean, running_var, gamma, beta, eps, border=None, want_pre=False, lif=False, D=8, vth=1.0,
                       stats=None):
    """Eval-mode  depthwise conv -> BatchNorm(running statistics) [-> Q_IFNode (reset, no membrane kept)]  as ONE launch
    (s2f_dwconv_bn_lif_fwd; row f4).  x: fp32 tensor or Spikes [N, C, H, W]; w [C, 1, K, K] -> (u fp32 or None, Spikes or None)."""
    data, _ = _unpack(x)
    data = data.contiguous()
    N, C, H, W = data.shape
    K = w.shape[-1]
    Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
    dev = data.device
    with torch.no_grad():
        u = torch.empty(N, C, Ho, Wo, dtype=torch.float32, device=dev) if want_pre else None
        y = torch.empty(N, C, Ho, Wo, dtype=torch.bfloat16, device=dev) if lif else None
        check(lib.s2f_dwconv_bn_lif_fwd(_ptr(data), _ptr(w.detach().contiguous()), _ptr(None if border is None else border.contiguous()),
                                        _ptr(running_mean), _ptr(running_var), _ptr(gamma), _ptr(beta), float(eps), _ptr(u), _ptr(y),
                                        _ptr(stats), N, C, H, W, K, pad, int(data.dtype == torch.bfloat16), float(vth), int(D), _stream()),
              "s2f_dwconv_bn_lif_fwd")
    return u, (Spikes(y, _new_tok(y)) if lif else None)


def dwconv(x, w, pad, border=None):
    """x: fp32 tensor or Spikes"""
    data, tok = _unpack(x)
    return _DWConv.apply(data, tok, w, border, pad)




# ------------------------------------------------------------------------------------------------ dense k x k convolution
def im2col(x, kh, kw, stride, padding):
    """torch.nn.functional.unfold(x, (kh, kw), 1, padding, stride) for fp32 maps and bf16 spike maps, as one gather kernel
    (csrc/im2col.hip): [N, C, H, W] -> [N, C kh kw, Ho Wo], the column matrix of the reference's stride-2 / 7x7 nn.Conv2d
    (mmseg/models/backbones/sdtv2.py:386-421)."""
    N, C, H, W = x.shape
    Ho = (H + 2 * padding - kh) // stride + 1
    Wo = (W + 2 * padding - kw) // stride + 1
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("im2col: needs an fp32 or bf16 map in device memory (there is no host path)")
    x = x.contiguous()
    cols = torch.empty(N, C * kh * kw, Ho * Wo, dtype=x.dtype, device=x.device)
    check(lib.s2f_im2col(_ptr(x), _ptr(cols), N, C, H, W, kh, kw, stride, padding, int(x.dtype == torch.bfloat16), _stream()),
          "s2f_im2col")
    return cols


def col2im(dcols, C, H, W, kh, kw, stride, padding):
    """torch.nn.functional.fold(dcols, (H, W), (kh, kw), 1, padding, stride) as a gather: every input pixel sums the column entries
    that cover it, in a fixed order, with no zero fill (csrc/im2col.hip)."""
    N = dcols.shape[0]
    if not dcols.is_cuda or dcols.dtype != torch.float32:
        raise RuntimeError("col2im: needs fp32 columns in device memory (there is no host path)")
    dcols = dcols.contiguous()
    gx = torch.empty(N, C, H, W, dtype=torch.float32, device=dcols.device)
    check(lib.s2f_col2im(_ptr(dcols), _ptr(gx), N, C, H, W, kh, kw, stride, padding, _stream()), "s2f_col2im")
    return gx


class _ConvDense(torch.autograd.Function):
    """Dense k x k Conv2d lowered to GEMMs (MIOpen is not usable on this image, see conv.py).

    forward : cols = im2col(x) ; y = W2d @ cols                       (spike GEMM when x is a neuron output)
    dW      : dY @ cols^T                                              (bf16-MFMA batch-reduce kernel for spike inputs)
              3x3 / stride 1 / padding 1 on spikes: both as IMPLICIT GEMMs -- the kernels' loaders read x itself, `cols`
              (9x the activation: 2.4 GB for MS_ConvBlock1_1.conv2 at C2) is neither written, read nor saved
    dX      : the cheaper of two equivalent lowerings --
                M >= C : dcols = W2d^T @ dY ; dX = col2im(dcols)       (the adjoint of im2col; C*k*k rows)
                M <  C : dX = flip(W)^T (*) dY = W_t2d @ im2col(dY)     (transposed convolution; M*k*k rows)
              the second form moves k*k*M instead of k*k*C rows through HBM and needs no col2im scatter; it applies to
              stride 1 (MS_ConvBlock.conv2: 4C -> C, sdtv2.py:202-204)."""

    @staticmethod
    def forward(ctx, x, tok, weight, bias, stride, padding, spike_input, stats=False):
        _need_cuda(weight, bias, spikes=x)
        ctx.has_tok = tok is not None
        part = None
        xb = x.dtype == torch.bfloat16
        N, C, H, W = x.shape
        M, _, kh, kw = weight.shape
        Ho = (H + 2 * padding - kh) // stride + 1
        Wo = (W + 2 * padding - kw) // stride + 1
        w2d = weight.view(M, -1)
        # implicit GEMM: 3x3, stride 1, padding 1 on spikes -- the kernels' loaders read the activation itself
        implicit = (spike_input and cfg.SPIKE_GEMM_ENABLED and cfg.CONV3X3_IMPLICIT and kh == 3 and kw == 3 and stride == 1
                    and padding == 1 and C % 32 == 0 and W % 4 == 0
                    and H * W >= cfg.CONV3X3_IMPLICIT_MIN_PIXELS)
        if implicit:
            x = x.contiguous()
            if cfg.SPIKE_GEMM_CHECK:
                assert _is_spike_grid(x), "not a spike tensor"
            y = torch.empty(N, M, H * W, dtype=torch.float32, device=x.device)
            _time_next("spike_gemm_fwd", 4 * N * H * W * (C + M), 2 * N * M * H * W * C * 9,
                       moved=N * H * W * ((2 if xb else 4) * C + 4 * M))
            P = _want_partials(stats and cfg.PGEMM_CONV and xb and cfg.SPIKE_GEMM_TERMS == 3 and bias is None, N, M, H * W)
            if P:
                part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device)
                check(lib.s2f_pgemm_conv3x3_bf16_stats(_ptr(pack_weight_conv3(weight)), _ptr(x), _ptr(y), _ptr(part), N, M, C, H, W,
                                                       _stream()), "s2f_pgemm_conv3x3_bf16_stats")
            elif cfg.PGEMM_CONV and xb and cfg.SPIKE_GEMM_TERMS == 3:
                check(lib.s2f_pgemm_conv3x3_bf16(_ptr(pack_weight_conv3(weight)), _ptr(x), _ptr(bias), _ptr(y), N, M, C, H, W, 0,
                                                 _stream()), "s2f_pgemm_conv3x3_bf16")
            else:
                ws = split_weight_conv3(weight)
                fn = lib.s2f_spike_conv3x3_fwd_bf16 if xb else lib.s2f_spike_conv3x3_fwd
                check(fn(_ptr(ws), _ptr(x), _ptr(bias), _ptr(y), N, M, C, H, W, ws.shape[1], ws.shape[2], cfg.SPIKE_GEMM_TERMS,
                         _stream()), "s2f_spike_conv3x3_fwd")
            ctx.save_for_backward(x, weight)
            ctx.geo = (N, C, H, W, M, kh, kw, Ho, Wo, stride, padding, bias is not None, True)
            ctx.implicit = True
            part = y.new_empty(0) if part is None else part
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y.view(N, M, Ho, Wo), part
        ctx.implicit = False
        cols = im2col(x, kh, kw, stride, padding)                                       # [N, C*kh*kw, Ho*Wo]
        use_mfma = spike_input and cfg.SPIKE_GEMM_ENABLED and cols.shape[2] % 4 == 0
        L = Ho * Wo
        if use_mfma:
            y = torch.empty(N, M, L, dtype=torch.float32, device=x.device)
            _time_next("spike_gemm_fwd", 4 * N * L * (cols.shape[1] + M), 2 * N * M * L * cols.shape[1],
                       moved=N * L * ((2 if xb else 4) * cols.shape[1] + 4 * M))
            pg = cfg.PGEMM and xb and L % 4 == 0 and L >= 8 and cfg.SPIKE_GEMM_TERMS == 3
            P = _want_partials(stats and pg and bias is None, N, M, L)
            if P:
                part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device)
                check(lib.s2f_pgemm_nn_bf16_stats(_ptr(pack_weight(w2d)), _ptr(cols), _ptr(y), _ptr(part), N, M, L, cols.shape[1],
                                                  _stream()), "s2f_pgemm_nn_bf16_stats")
            elif pg:
                check(lib.s2f_pgemm_nn_bf16(_ptr(pack_weight(w2d)), _ptr(cols), _ptr(bias), _ptr(y), N, M, L, cols.shape[1],
                                            cfg.SPIKE_GEMM_TERMS, 0, _stream()), "s2f_pgemm_nn_bf16")
            else:
                ws = split_weight(w2d)
                fn = lib.s2f_spike_gemm_fwd_bf16 if xb else lib.s2f_spike_gemm_fwd
                check(fn(_ptr(ws), _ptr(cols), _ptr(bias), _ptr(y), N, M, L, cols.shape[1], ws.shape[1], ws.shape[2],
                         cfg.SPIKE_GEMM_TERMS, _stream()), "s2f_spike_gemm_fwd")
        else:
            if xb:
                cols = cols.float()
            if cfg.PGEMM_DX and L % 4 == 0 and L >= cfg.PGEMM_MIN_N:
                # general fp32 input (the stem reads the image): the transposed product on the pack of W^T, 6 passes
                y = torch.empty(N, M, L, dtype=torch.float32, device=x.device)
                _time_next("dx_gemm", 4 * N * L * (cols.shape[1] + M), 2 * N * M * L * cols.shape[1])
                P = _want_partials(stats and bias is None, N, M, L)
                if P:
                    part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device)
                    check(lib.s2f_pgemm_dx_f32_stats(_ptr(pack_weight(w2d, transposed=True)), _ptr(cols), 0, _ptr(y), 0, _ptr(part), N,
                                                     cols.shape[1], M, L, _stream()), "s2f_pgemm_dx_f32_stats")
                else:
                    check(lib.s2f_pgemm_dx_f32(_ptr(pack_weight(w2d, transposed=True)), _ptr(cols), 0, _ptr(y), 0, N, cols.shape[1], M,
                                               L, 0.0, 0, _stream()), "s2f_pgemm_dx_f32")
            else:
                y = bmm_small(w2d.unsqueeze(0).expand(N, -1, -1), cols)
            if bias is not None:
                y = y + bias.view(1, -1, 1)
        ctx.save_for_backward(cols, weight)
        ctx.geo = (N, C, H, W, M, kh, kw, Ho, Wo, stride, padding, bias is not None, use_mfma)
        part = y.new_empty(0) if part is None else part
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)
        return y.view(N, M, Ho, Wo), part

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        cols, weight = ctx.saved_tensors
        if gy is None:
            return (None,) * 8
        N, C, H, W, M, kh, kw, Ho, Wo, stride, padding, has_bias, use_mfma = ctx.geo
        gy = gy.contiguous().view(N, M, Ho * Wo)
        w2d = weight.view(M, -1)
        gx = gw = gb = None
        xb = cols.dtype == torch.bfloat16
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if (cfg.CONV3X3_DX_IMPLICIT and kh == 3 and kw == 3 and stride == 1 and padding == 1 and M % 32 == 0 and W % 4 == 0
                    and gy.is_cuda and H * W >= cfg.CONV3X3_DX_MIN_PIXELS):
                # transposed convolution dX = flip(W)^T (*) dY as an implicit 6-pass split GEMM: no unfold(dY), no col2im
                gx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
                _time_next("dx_gemm", 4 * N * H * W * (C + M), 2 * N * M * H * W * C * 9)
                # measured (tools/probe_pgemm.py conv): the pipelined kernel wins for <= 64 output rows (narrow tiles: 442 vs 729 us
                # on [32 <- 128] at 256 x 256) and for long contractions (>= 256 channels); the round-2 kernel keeps a 5-10 % edge
                # on wide outputs over short contractions
                if cfg.PGEMM_CONV and (cfg.CONV3X3_DX_PIPE == 2 or (cfg.CONV3X3_DX_PIPE == 1 and (C <= 64 or M >= 256))):
                    check(lib.s2f_pgemm_conv3x3_f32(_ptr(pack_weight_conv3(weight, transposed=True)), _ptr(gy), _ptr(gx), N, C, M, H,
                                                    W, 0, _stream()), "s2f_pgemm_conv3x3_f32")
                else:
                    wt = split_weight_tconv3(weight)
                    check(lib.s2f_conv3x3_general(_ptr(wt), _ptr(gy), _ptr(gx), N, C, M, H, W, wt.shape[1], wt.shape[2], _stream()),
                          "s2f_conv3x3_general")
            elif M < C and stride == 1 and kh == kw and Ho == H and Wo == W:
                wt = weight.flip(2, 3).permute(1, 0, 2, 3).reshape(C, M * kh * kw)        # [C, M*k*k], tiny
                gcols = im2col(gy.view(N, M, Ho, Wo), kh, kw, 1, kh - 1 - padding)
                gx = bmm_small(wt.unsqueeze(0).expand(N, -1, -1), gcols).view(N, C, H, W)
            else:
                dcols = dx_gemm(w2d, gy)
                gx = col2im(dcols, C, H, W, kh, kw, stride, padding)
        if ctx.needs_input_grad[2]:
            K = w2d.shape[1]
            if ctx.implicit:
                x = cols                                                  # the saved tensor is the activation itself
                # bf16 spikes: the kernels store the gradient in the weight's own layout [M, C, 3, 3] (they contract tap-major) and add
                # into the parameter's slot of the flat gradient buffer when there is one -- no zeroed staging tensor, no permuted add
                sink = _sink_for(weight)
                direct = xb and cfg.CONV_DW_DIRECT
                if direct:
                    gt = sink if sink is not None else torch.zeros(M, C, 3, 3, dtype=torch.float32, device=gy.device)
                else:
                    gt = torch.empty(M, 3, 3, C, dtype=torch.float32, device=gy.device)    # tap-major, as the kernel contracts
                _time_next("spike_gemm_dw", 4 * N * H * W * (C + M), 2 * N * M * H * W * K,
                           moved=N * H * W * ((2 if xb else 4) * C + 4 * M))
                if xb and cfg.DW_PIPE_CONV and M >= 128 and C >= 64 and lib.s2f_spike_conv3x3_dw_pipe_ok(N, M, C, H, W):
                    # the LDS-DMA pipelined kernel (csrc/dwp.hip): the horizontal taps come from a copy shifted by one element.
                    # Its tile is 128 output channels x 256 (tap, input channel) rows: narrower layers stay on the round-2 kernel
                    # (measured, tools/probe_dwp_conv.py: M = 32 / 64 lose 10-70 %, C = 32 ties)
                    import ctypes
                    xs = torch.empty(x.numel() + 16, dtype=x.dtype, device=x.device)
                    check(lib.s2f_shift1_bf16(_ptr(x), _ptr(xs), x.numel(), _stream()), "s2f_shift1_bf16")
                    arr = (ctypes.c_int64 * 9)(_ptr(gy), _ptr(x), _ptr(xs), _ptr(gt), N, M, C, H, W)
                    if not direct:
                        gt.zero_()
                    check(lib.s2f_spike_conv3x3_dw_pipe(arr, 1, (cfg.DWP_SCHEDULE & 1) | (2 if direct else 0), cfg.DWP_WGS, _stream()),
                          "s2f_spike_conv3x3_dw_pipe")
                elif xb:
                    check(lib.s2f_spike_conv3x3_dw_bf16(_ptr(gy), _ptr(x), _ptr(gt), N, M, C, H, W, 3 if direct else 0, _stream()),
                          "s2f_spike_conv3x3_dw_bf16")
                else:
                    check(lib.s2f_spike_conv3x3_dw(_ptr(gy), _ptr(x), _ptr(gt), N, M, C, H, W, 0, _stream()), "s2f_spike_conv3x3_dw")
                if direct:
                    gw = None if sink is not None else gt
                elif sink is not None:
                    sink.view(M, C, 3, 3).add_(gt.permute(0, 3, 1, 2))
                    gw = None
                else:
                    gw = gt.permute(0, 3, 1, 2).contiguous()
            elif use_mfma and cfg.SPIKE_GEMM_DW and M >= 16:
                sink = _sink_for(weight)
                gw = torch.empty(M, K, dtype=torch.float32, device=gy.device) if sink is None else None
                if (cfg.DEFER_DW and sink is not None and xb and N * Ho * Wo <= cfg.DEFER_DW_MAX_CONTRACTION and cfg.WGRAD_STREAM is None):
                    _defer_dw(gy, cols, sink, N, M, K, Ho * Wo)
                    return _grad_pair(ctx.has_tok, gx) + (None, gy.sum((0, 2)) if (has_bias and ctx.needs_input_grad[3]) else None,
                                                          None, None, None, None)
                _time_next("spike_gemm_dw", 4 * N * Ho * Wo * (K + M), 2 * N * M * Ho * Wo * K,
                           moved=N * Ho * Wo * ((2 if xb else 4) * K + 4 * M))
                side = _wgrad_stream(sink, gy, cols)
                st = side.cuda_stream if side is not None else _stream()
                if xb:
                    check(lib.s2f_spike_gemm_dw_bf16(_ptr(gy), _ptr(cols), _ptr(gw if sink is None else sink), N, M, K, Ho * Wo,
                                                     int(sink is not None), st), "s2f_spike_gemm_dw_bf16")
                else:
                    check(lib.s2f_spike_gemm_dw(_ptr(gy), _ptr(cols), _ptr(gw if sink is None else sink), N, M, K, Ho * Wo,
                                                int(sink is not None), 1, st), "s2f_spike_gemm_dw")
            elif cfg.PGEMM_DX and (Ho * Wo) % 4 == 0 and cols.dtype == torch.float32:
                # both operands general fp32 (the stem): 6-pass weight-gradient kernel, straight into the sink when there is one
                sink = _sink_for(weight)
                gw = torch.empty(M, K, dtype=torch.float32, device=gy.device) if sink is None else None
                check(lib.s2f_gemm_dw_general(_ptr(gy), 0, _ptr(cols), 0, _ptr(gw if sink is None else sink), N, M, K, Ho * Wo,
                                              int(sink is not None), _stream()), "s2f_gemm_dw_general")
            else:
                gw = bmm_small(gy, cols.float().transpose(1, 2), reduce_batch=True)          # ragged / tiny maps (csrc/bmm.hip)
            gw = gw.view_as(weight) if gw is not None else None
        if has_bias and ctx.needs_input_grad[3]:
            gb = gy.sum((0, 2))
        return _grad_pair(ctx.has_tok, gx) + (gw, gb, None, None, None, None)


def conv_dense(x, weight, bias, stride, padding, spike_input, stats=False):
    """x: fp32 tensor, or Spikes (then `spike_input` is implied).  stats: as spike_gemm"""
    data, tok = _unpack(x)
    return _with_part(*_ConvDense.apply(data, tok, weight, bias, stride, padding, spike_input or isinstance(x, Spikes), bool(stats)))


__all__ = [n for n in dir() if not n.startswith('__')]
