"""GEMM-shaped ops: the packed-weight spike GEMMs and their gradients (csrc/pgemm.hip, gemm.hip, gemm_bf16.hip), the weight
conversion caches, dense-input products, token-major nn.Linear, the mask contraction."""
import torch

from .config import cfg
from .core import *          # noqa: F401,F403  (the shared plumbing: _ptr, _stream, check, lib, Spikes, ...)
from .misc import channel_sum, transpose_last2


# ------------------------------------------------------------------------------------------------ small / ragged products
# The matrix-core GEMM families want rows of whole 16-byte groups (L % 4 == 0) and at least one 128-column tile.  Everything else --
# the plumbing configuration's 4 x 4 .. 16 x 16 maps and 10-query rows, odd test shapes -- runs on
# s2f_bmm_f32 (csrc/bmm.hip): a strided fp32 product on the vector ALUs, ascending-k fp32 multiply-adds, bit-repeatable.  Rounds 1-5
# sent these to rocBLAS / hipBLASLt through torch.bmm (`bmm_tuned`, timed per shape): the last vendor GEMMs on the path, and exactly
# the shapes of the fixtures the reference itself produced -- a parity test could not tell this build's arithmetic from the library's.
# No vendor GEMM is called from this package any more.


def bmm_small(a, b, reduce_batch=False):
    """a [B, M, K] @ b [B, K, N] -> [B, M, N] (or, `reduce_batch`, their sum over B -> [M, N]); fp32, any strides (expanded /
    transposed views are read in place)."""
    _need_cuda(a, b)
    B, M, K = a.shape
    N = b.shape[2]
    if b.shape[0] != B or b.shape[1] != K:
        raise RuntimeError(f"bmm_small: {tuple(a.shape)} @ {tuple(b.shape)}")
    c = torch.empty((M, N) if reduce_batch else (B, M, N), dtype=torch.float32, device=a.device)
    if c.numel() == 0:
        return c
    if K == 0 or B == 0:
        return c.zero_()
    check(lib.s2f_bmm_f32(_ptr(a), a.stride(0), a.stride(1), a.stride(2), _ptr(b), b.stride(0), b.stride(1), b.stride(2), _ptr(c),
                          0 if reduce_batch else M * N, N, 1, B, M, N, K, int(bool(reduce_batch)), _stream()), "s2f_bmm_f32")
    return c




class _DenseGemm(torch.autograd.Function):
    """Y[b] = W @ X[b] for a general fp32 X (the 1x1 convolutions that do not read spikes: SepConv.pwconv2, RepConv's second
    1x1, sdtv2.py:124-125, 164) with G independent weights applied to consecutive channel groups: ws = G matrices [M, K],
    X [B, G*K, L] -> [B, G*M, L].  Forward and input gradient on s2f_pgemm_dx_f32 (the transposed product on the pack of W^T
    resp. W, X / dY split hi + mid + lo in the kernel: 6 passes), weight gradient on s2f_gemm_dw_general; a group is a call
    with offset pointers and the batch strides of the full tensors (no copies).  Shapes the kernels do not take (L % 4 != 0,
    L < 128) fall back to the library GEMM."""

    @staticmethod
    def forward(ctx, stats, x, *ws):
        _need_cuda(x, *ws)
        G = len(ws)
        M, K = ws[0].shape
        B, _, L = x.shape
        x = x.contiguous()
        ctx.save_for_backward(x, *ws)
        ctx.fast = cfg.PGEMM_DX and L % 4 == 0 and L >= cfg.PGEMM_MIN_N
        if ctx.fast:
            y = torch.empty(B, G * M, L, dtype=torch.float32, device=x.device)
            P = _want_partials(stats, B, G * M, L)
            part = torch.empty(G * M, P, 2, dtype=torch.float32, device=x.device) if P else x.new_empty(0)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            if 1 < G <= 4 and cfg.DENSE_GROUPED:
                # the G products as ONE launch (blockIdx.z = group): three workgroups per CU instead of one on the 32x32-stage maps
                import ctypes
                packs = (ctypes.c_void_p * G)(*[pack_weight(w, transposed=True).data_ptr() for w in ws])
                _time_next("dx_gemm", 4 * B * L * G * (K + M), 2 * B * G * M * L * K)
                check(lib.s2f_pgemm_dx_f32_grouped(packs, G, _ptr(x), G * K * L, K * L, _ptr(y), G * M * L, M * L, _ptr(part) if P else 0,
                                                   2 * M * P, B, K, M, L, _stream()), "s2f_pgemm_dx_f32_grouped")
                return y, part
            for g, w in enumerate(ws):
                _time_next("dx_gemm", 4 * B * L * (K + M), 2 * B * M * L * K)
                if P:
                    check(lib.s2f_pgemm_dx_f32_stats(_ptr(pack_weight(w, transposed=True)), x.data_ptr() + 4 * g * K * L, G * K * L,
                                                     y.data_ptr() + 4 * g * M * L, G * M * L, part.data_ptr() + 8 * g * M * P,
                                                     B, K, M, L, _stream()), "s2f_pgemm_dx_f32_stats")
                else:
                    check(lib.s2f_pgemm_dx_f32(_ptr(pack_weight(w, transposed=True)), x.data_ptr() + 4 * g * K * L, G * K * L,
                                               y.data_ptr() + 4 * g * M * L, G * M * L, B, K, M, L, 0.0, 0, _stream()),
                          "s2f_pgemm_dx_f32")
            return y, part
        part = x.new_empty(0)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)
        wb = torch.stack(ws, 0).unsqueeze(0).expand(B, G, M, K).reshape(B * G, M, K) if G > 1 else ws[0].expand(B, M, K)
        return bmm_small(wb, x.view(B * G, K, L)).view(B, G * M, L), part

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, *ws = ctx.saved_tensors
        G = len(ws)
        if gy is None:
            return (None,) * (2 + G)
        M, K = ws[0].shape
        B, _, L = x.shape
        gy = gy.contiguous()
        gx, gws = None, [None] * G
        if ctx.fast:
            if ctx.needs_input_grad[1] and 1 < G <= 4 and cfg.DENSE_GROUPED:
                import ctypes
                gx = torch.empty(B, G * K, L, dtype=torch.float32, device=gy.device)
                packs = (ctypes.c_void_p * G)(*[pack_weight(w).data_ptr() for w in ws])
                _time_next("dx_gemm", 4 * B * L * G * (K + M), 2 * B * G * M * L * K)
                check(lib.s2f_pgemm_dx_f32_grouped(packs, G, _ptr(gy), G * M * L, M * L, _ptr(gx), G * K * L, K * L, 0, 0, B, M, K, L,
                                                   _stream()), "s2f_pgemm_dx_f32_grouped")
            elif ctx.needs_input_grad[1]:
                gx = torch.empty(B, G * K, L, dtype=torch.float32, device=gy.device)
                for g, w in enumerate(ws):
                    _time_next("dx_gemm", 4 * B * L * (K + M), 2 * B * M * L * K)
                    check(lib.s2f_pgemm_dx_f32(_ptr(pack_weight(w)), gy.data_ptr() + 4 * g * M * L, G * M * L,
                                               gx.data_ptr() + 4 * g * K * L, G * K * L, B, M, K, L, 0.0, 0, _stream()),
                          "s2f_pgemm_dx_f32")
            for g, w in enumerate(ws):
                if not ctx.needs_input_grad[2 + g]:
                    continue
                sink = _sink_for(w)
                if _defer_dw_general(gy, g * M * L, G * M * L, x, g * K * L, G * K * L, sink, B, M, K, L):
                    continue
                if sink is None:
                    gws[g] = torch.empty(M, K, dtype=torch.float32, device=gy.device)
                check(lib.s2f_gemm_dw_general(gy.data_ptr() + 4 * g * M * L, G * M * L, x.data_ptr() + 4 * g * K * L, G * K * L,
                                              _ptr(gws[g] if sink is None else sink), B, M, K, L, int(sink is not None),
                                              _stream()), "s2f_gemm_dw_general")
            return (None, gx, *gws)
        gyv = gy.view(B * G, M, L)
        if ctx.needs_input_grad[1]:
            wt = torch.stack(ws, 0).transpose(1, 2)
            wb = wt.unsqueeze(0).expand(B, G, K, M).reshape(B * G, K, M) if G > 1 else wt[0].expand(B, K, M)
            gx = bmm_small(wb, gyv).view(B, G * K, L)
        if any(ctx.needs_input_grad[2:]):
            gw = bmm_small(gyv, x.view(B * G, K, L).transpose(1, 2)).view(B, G, M, K).sum(0)
            gws = [gw[g] for g in range(G)]
        return (None, gx, *gws)


def dense_gemm(x, w, stats=False):
    """x [B, G*K, L]; w: one matrix [M, K], a stack [G, M, K], or a list of G matrices (e.g. views of G parameters: each then
    keeps its own cached pack and gradient sink) -> [B, G*M, L].  stats: as spike_gemm"""
    if torch.is_tensor(w):
        w = [w] if w.dim() == 2 else list(w.unbind(0))
    y, part = _DenseGemm.apply(bool(stats), x, *w)
    return _with_part(y, part)




# ------------------------------------------------------------------------------------------------ spike GEMM (bf16 MFMA)
_SPLIT_CACHE = {}
# 3x3 / stride 1 / pad 1 spike convolutions as implicit GEMMs (no im2col matrix; s2f_spike_conv3x3_fwd / _dw).  Round 1 measured
# the pair (forward + weight gradient) as a win on the >= 128x128 maps only (tools/probe_conv3.py: 64x64 maps 485 vs 420 and
# 795 vs 660 us against the saved column matrix).  With the loaders' prefetches freed of their predicates (round 2, conv3_fix)
# the implicit form wins from 32x32 up -- same-box A/B of the step: threshold 128x128 43.92, 64x64 43.39, 32x32 43.44 ms -- and
# the bf16 column matrices of the 64x64 / 32x32 stages (ATen im2col) are gone.
# input gradient of the 3x3 convolutions as an implicit transposed convolution on the 6-pass split GEMM (no unfold / col2im)
# Round 3: the LDS-DMA pipelined kernels (csrc/pgemm.hip).  PGEMM: forward spike GEMMs on s2f_pgemm_nn_bf16 (packed weight, bf16
# spikes, any N % 4 == 0 since the register-staged form); PGEMM_DX: every fp32 x fp32 product that ran on the library in rounds 1-2 -- the input
# gradients of the 1x1 convolutions and the forward products of the convolutions whose input is not a spike map -- on
# s2f_pgemm_dx_f32 (6 bf16 passes = fp32 accuracy), their weight gradients on s2f_gemm_dw_general.
# Round 4: BatchNorm statistics from the producing GEMM's epilogue, without atomics (s2f.h "BatchNorm statistics from the producing
# GEMM's epilogue").  A forward convolution of a module in training mode stores per-(tile, row) partial sums next to its output
# and hands them over as the attribute `_s2f_part` of the tensor it returns (`carry_stats` moves it across a view); fused.bn_act
# passes them to the BatchNorm apply kernel instead of launching s2f_bn_stats.  A tensor that lost the attribute on the way simply
# takes the statistics pass.  BN_PARTIALS_SINGLE: also for the small maps whose BatchNorm computes its statistics itself in one
# pass (then the row-walking apply kernel runs instead of the single-pass kernel).


def _want_partials(stats, B, M, L):
    """-> number of partials per channel (> 0) if the product [B, M, L] should store BatchNorm partials, else 0"""
    if not (stats and cfg.BN_PARTIALS and L % 4 == 0):
        return 0
    if not cfg.BN_PARTIALS_SINGLE and lib.s2f_bn_single_pass(B, M, L):
        return 0
    return int(lib.s2f_bn_partials_count(B, L))


def carry_stats(src, dst):
    """dst is a view / reshape of the GEMM output src: the BatchNorm partials stored with src describe dst as well"""
    part = stats_of(src)
    if part is not None:
        dst._s2f_part = (part, dst._version, dst.data_ptr())
    return dst


def stats_of(z):
    """The BatchNorm partials the producing GEMM stored with z, IF z still holds what the GEMM wrote: the attribute carries the
    tensor's version counter and address at hand-over; an in-place change since (add_, mul_, a hook) or a carry onto another
    tensor invalidates them and the BatchNorm takes its statistics pass instead."""
    hit = getattr(z, "_s2f_part", None)
    if hit is None:
        return None
    part, version, ptr = hit
    return part if (z._version == version and z.data_ptr() == ptr) else None


def _owner(t):
    """The long-lived tensor object a cached split belongs to: the parameter a view was taken from (or the first twin of a
    zero-copy concatenation).  The cache keeps a weak reference to it -- an address is not an identity: once a model is
    freed, another model's weight of the same shape lands on the same address with the same version counter."""
    o = getattr(t, "_s2f_owner", None)
    if o is not None:
        return o
    return t._base if t._base is not None else t


# A cache entry: (version, out, shape, weakref(owner), job) with job = (src address, mode, C, M, K) -- what
# s2f_split_bf16x3_multi needs to redo this split from the live weight (resplit_all).  Only the address is kept (a tensor
# would keep a freed model's weights allocated); it is used only while the owner is alive and its storage still covers it.
_TRUST_ALL = [False]          # set by resplit_all() inside a capture: every registered split was just redone from the live weights


def _cache_get(key, version, shape, owner):
    hit = _SPLIT_CACHE.get(key)
    if hit is not None and hit[2] == shape and hit[3]() is owner:
        if hit[0] == version:
            return hit[1]
        if _TRUST_ALL[0] and hit[4] is not None and torch.cuda.is_current_stream_capturing():
            _SPLIT_CACHE[key] = (version,) + hit[1:]
            return hit[1]
    return None


def _cache_buffer(key, shape, owner, out_shape, device):
    """The destination of a (re-)conversion: the buffer of a stale entry of the same weight is converted INTO again -- a
    captured hipGraph (and the job tables of resplit_all) hold its address, a fresh allocation would leave them writing into
    freed memory -- otherwise a new one."""
    hit = _SPLIT_CACHE.get(key)
    if (hit is not None and hit[2] == shape and hit[3]() is owner and hit[1].device == device
            and tuple(hit[1].shape) == tuple(out_shape)):
        return hit[1]
    return torch.empty(out_shape, dtype=torch.int16, device=device)


def _cache_put(key, version, out, shape, owner, job=None, kind="split"):
    import weakref
    if len(_SPLIT_CACHE) > 4096:                       # dead entries of freed models
        for k in [k for k, v in _SPLIT_CACHE.items() if v[3]() is None]:
            del _SPLIT_CACHE[k]
    old = _SPLIT_CACHE.get(key)
    _SPLIT_CACHE[key] = (version, out, shape, weakref.ref(owner), job, kind)
    if old is None or old[1] is not out or old[4] != job:
        _SPLIT_TABLE["keys"] = None                    # a new destination: the job tables must be rebuilt (never mutated)


# Job tables of resplit_all: one per conversion kernel.  A table tensor is REPLACED, never written again, once built: a
# captured graph keeps reading the tensor it recorded (graph.py holds references to the tables and buffers of its capture).
_SPLIT_TABLE = {"keys": None, "jobs": None, "blocks": 0, "njobs": 0, "pack_jobs": None, "pack_blocks": 0, "pack_njobs": 0}


def conversion_state():
    """What a captured step must keep alive: the job tables resplit_all launched with and every cached conversion buffer."""
    return (_SPLIT_TABLE["jobs"], _SPLIT_TABLE["pack_jobs"], [v[1] for v in _SPLIT_CACHE.values()])


def resplit_all(device, build=True):
    """Redo EVERY cached weight conversion (bf16 hi/mid/lo splits and packs) from the live fp32 weights: one launch per
    conversion kernel (s2f_split_bf16x3_multi, s2f_pack_bf16x3_multi).  A training step owes this after each optimiser update;
    a captured step (graph.GraphedStep) records it, so every replay multiplies by the current weights -- without it the graph
    would replay the bf16 terms of capture time while its backward reads the live fp32 weights.  -> number of weights
    converted; -1 when the job tables would have to be (re)built and `build` is False (they are uploaded from the host, which
    a stream capture does not allow: GraphedStep calls this once before capturing)."""
    def covered(v):
        o = v[3]()
        if o is None or v[4] is None or v[1].device != device:
            return False
        st = o.untyped_storage()
        return st.data_ptr() <= v[4][0] and v[4][0] + 4 * v[4][3] * v[4][4] <= st.data_ptr() + st.nbytes()
    live = [(k, v) for k, v in _SPLIT_CACHE.items() if covered(v)]
    if not live:
        return 0
    keys = tuple(k for k, _ in live)
    tab = _SPLIT_TABLE
    if tab["keys"] != keys or (tab["jobs"] is None and tab["pack_jobs"] is None) or \
            (tab["jobs"] if tab["jobs"] is not None else tab["pack_jobs"]).device != device:
        if not build:
            return -1
        rows, first, prows, pfirst = [], 0, [], 0
        for _, (_ver, out, _shape, _own, (src, mode, cdim, M, K), kind) in live:
            if kind == "pack":
                prows.append([src, out.data_ptr(), M, K, mode | (cdim << 8), pfirst, 0, 0])
                pfirst += ((M + 63) // 64) * ((K + 31) // 32) * 2
            else:
                Mpad, Kpad = out.shape[1], out.shape[2]
                rows.append([src, out.data_ptr(), M, K, Mpad, Kpad, mode | (cdim << 8), first])
                first += (Mpad * Kpad + 1023) // 1024
        tab.update(keys=keys, blocks=first, njobs=len(rows), pack_blocks=pfirst, pack_njobs=len(prows),
                   jobs=torch.tensor(rows, dtype=torch.int64).to(device) if rows else None,
                   pack_jobs=torch.tensor(prows, dtype=torch.int64).to(device) if prows else None)
    if tab["njobs"]:
        check(lib.s2f_split_bf16x3_multi(_ptr(tab["jobs"]), tab["njobs"], tab["blocks"], _stream()), "s2f_split_bf16x3_multi")
    if tab["pack_njobs"]:
        check(lib.s2f_pack_bf16x3_multi(_ptr(tab["pack_jobs"]), tab["pack_njobs"], tab["pack_blocks"], _stream()),
              "s2f_pack_bf16x3_multi")
    return len(live)


def split_weight(w2d):
    """fp32 [M, K] -> cached bf16 [3, Mpad, Kpad] (hi, mid, lo).  Re-split when the parameter is modified in place
    (optimiser step, load_state_dict) -- tracked through the tensor version counter; weights must not be mutated through
    `.data` (its own version counter).  Inside a captured step the splits are redone by resplit_all()."""
    key = (w2d.data_ptr(), w2d.numel())
    M, K = w2d.shape
    # a zero-copy concatenation of sibling parameters (cat_params) is a fresh tensor every call: it carries the sum of the
    # parameters' version counters instead of its own
    version = getattr(w2d, "_s2f_version", w2d._version)
    owner = _owner(w2d)
    hit = _cache_get(key, version, (M, K), owner)
    if hit is not None:
        return hit
    Mpad, Kpad = (M + 63) // 64 * 64, (K + 31) // 32 * 32
    out = _cache_buffer(key, (M, K), owner, (3, Mpad, Kpad), w2d.device)
    src = w2d.detach()
    job = (src.data_ptr(), 0, 0, M, K) if src.is_contiguous() else None
    check(lib.s2f_split_bf16x3(_ptr(src.contiguous()), _ptr(out), M, K, Mpad, Kpad, _stream()), "s2f_split_bf16x3")
    _cache_put(key, version, out, (M, K), owner, job)
    return out


def pack_weight(w2d, transposed=False):
    """fp32 [M, K] -> the cached bf16 PACK of it (s2f.h "pipelined GEMMs": blocks of [3 terms][64 rows][32 k], the LDS image of
    the LDS-DMA kernels), or of its transpose (`transposed`: the pack of w2d^T, the A operand of the forward product of a
    convolution whose input is a general fp32 tensor).  The pack of W serves its forward product (s2f_pgemm_nn_bf16) AND the
    input gradient W^T dY (s2f_pgemm_dx_f32).  Versioning and in-graph refresh as split_weight."""
    key = ("pack", bool(transposed), w2d.data_ptr(), w2d.numel())
    R, Cc = w2d.shape
    M, K = (Cc, R) if transposed else (R, Cc)
    version = getattr(w2d, "_s2f_version", w2d._version)
    owner = _owner(w2d)
    hit = _cache_get(key, version, (M, K), owner)
    if hit is not None:
        return hit
    out = _cache_buffer(key, (M, K), owner, (int(lib.s2f_pack_elems(M, K)),), w2d.device)
    src = w2d.detach()
    mode = 3 if transposed else 0
    job = (src.data_ptr(), mode, 0, M, K) if src.is_contiguous() else None
    check(lib.s2f_pack_bf16x3(_ptr(src.contiguous()), _ptr(out), M, K, mode, 0, _stream()), "s2f_pack_bf16x3")
    _cache_put(key, version, out, (M, K), owner, job, kind="pack")
    return out


def pack_weight_conv3(weight, transposed=False):
    """[M, C, 3, 3] -> the cached PACK (see pack_weight) of the TAP-MAJOR matrix [M, (ky, kx, c)] the implicit 3x3 kernels contract
    over (s2f_pgemm_conv3x3_bf16), or -- `transposed` -- of the transposed-convolution matrix [C, (ky, kx, m)] with flipped taps,
    Wt[c][(ky, kx), m] = weight[m][c][2 - ky][2 - kx]: the A operand of the input gradient (s2f_pgemm_conv3x3_f32)."""
    key = ("pack3", bool(transposed), weight.data_ptr())
    Mw, C = weight.shape[:2]
    M, K, mode, cdim = (C, 9 * Mw, 2, Mw) if transposed else (Mw, 9 * C, 1, C)
    owner = _owner(weight)
    hit = _cache_get(key, weight._version, (Mw, C), owner)
    if hit is not None:
        return hit
    out = _cache_buffer(key, (Mw, C), owner, (int(lib.s2f_pack_elems(M, K)),), weight.device)
    src = weight.detach()
    job = (src.data_ptr(), mode, cdim, M, K) if src.is_contiguous() else None
    check(lib.s2f_pack_bf16x3(_ptr(src.contiguous()), _ptr(out), M, K, mode, cdim, _stream()), "s2f_pack_bf16x3")
    _cache_put(key, weight._version, out, (Mw, C), owner, job, kind="pack")
    return out


def split_weight_conv3(weight):
    """[M, C, 3, 3] -> cached bf16 split of the TAP-MAJOR matrix [M, (ky, kx, c)] that the implicit 3x3 kernels contract over."""
    key = ("tap", weight.data_ptr())
    M, C = weight.shape[:2]
    hit = _cache_get(key, weight._version, (M, C), _owner(weight))
    if hit is not None:
        return hit
    w2d = weight.detach().permute(0, 2, 3, 1).reshape(M, 9 * C)
    Mpad, Kpad = (M + 63) // 64 * 64, (9 * C + 31) // 32 * 32
    out = _cache_buffer(key, (M, C), _owner(weight), (3, Mpad, Kpad), weight.device)
    check(lib.s2f_split_bf16x3(_ptr(w2d), _ptr(out), M, 9 * C, Mpad, Kpad, _stream()), "s2f_split_bf16x3")
    src = weight.detach()
    _cache_put(key, weight._version, out, (M, C), _owner(weight), (src.data_ptr(), 1, C, M, 9 * C) if src.is_contiguous() else None)
    return out


def split_weight_tconv3(weight):
    """[M, C, 3, 3] -> cached bf16 split of the transposed-convolution matrix [C, (ky, kx, m)] with flipped taps
    (Wt[c][(ky, kx), m] = weight[m][c][2 - ky][2 - kx]), rows padded to a multiple of 128 for s2f_conv3x3_general."""
    key = ("tconv", weight.data_ptr())
    M, C = weight.shape[:2]
    hit = _cache_get(key, weight._version, (M, C), _owner(weight))
    if hit is not None:
        return hit
    w2d = weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * M)
    Mpad, Kpad = (C + 127) // 128 * 128, (9 * M + 31) // 32 * 32
    out = _cache_buffer(key, (M, C), _owner(weight), (3, Mpad, Kpad), weight.device)
    check(lib.s2f_split_bf16x3(_ptr(w2d), _ptr(out), C, 9 * M, Mpad, Kpad, _stream()), "s2f_split_bf16x3")
    src = weight.detach()
    _cache_put(key, weight._version, out, (M, C), _owner(weight), (src.data_ptr(), 2, M, C, 9 * M) if src.is_contiguous() else None)
    return out


def _is_spike_grid(x):
    xf = x.float()
    return torch.equal(xf * 8, torch.round(xf * 8)) and float(xf.abs().max()) <= 16


class _SpikeGemm(torch.autograd.Function):
    """Y[b] = W @ X[b] (+ bias) with X spikes (bf16 pair or fp32): forward and weight gradient on the bf16 matrix cores (W /
    dY split hi+mid+lo), input gradient on the transposed packed-weight kernel (dx_gemm: s2f_pgemm_dx_f32, 6 passes)."""

    @staticmethod
    def forward(ctx, x, tok, w2d, bias, stats=False):
        _need_cuda(w2d, bias, spikes=x)
        x = x.contiguous()
        B, K, N = x.shape
        M = w2d.shape[0]
        if cfg.SPIKE_GEMM_CHECK:
            assert _is_spike_grid(x), "not a spike tensor"
        y = torch.empty(B, M, N, dtype=torch.float32, device=x.device)
        xb = x.dtype == torch.bfloat16
        _time_next("spike_gemm_fwd", 4 * B * N * (K + M), 2 * B * M * N * K, moved=B * N * ((2 if xb else 4) * K + 4 * M))
        pg = cfg.PGEMM and xb and N % 4 == 0 and N >= 8 and cfg.SPIKE_GEMM_TERMS == 3
        P = _want_partials(stats and pg and bias is None, B, M, N)
        part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device) if P else None
        if P:
            check(lib.s2f_pgemm_nn_bf16_stats(_ptr(pack_weight(w2d)), _ptr(x), _ptr(y), _ptr(part), B, M, N, K, _stream()),
                  "s2f_pgemm_nn_bf16_stats")
        elif pg:
            check(lib.s2f_pgemm_nn_bf16(_ptr(pack_weight(w2d)), _ptr(x), _ptr(bias), _ptr(y), B, M, N, K, cfg.SPIKE_GEMM_TERMS, 0,
                                        _stream()), "s2f_pgemm_nn_bf16")
        else:
            ws = split_weight(w2d)
            fn = lib.s2f_spike_gemm_fwd_bf16 if xb else lib.s2f_spike_gemm_fwd
            check(fn(_ptr(ws), _ptr(x), _ptr(bias), _ptr(y), B, M, N, K, ws.shape[1], ws.shape[2], cfg.SPIKE_GEMM_TERMS, _stream()),
                  "s2f_spike_gemm_fwd")
        ctx.save_for_backward(x, w2d)
        ctx.has_bias, ctx.has_tok = bias is not None, tok is not None
        if part is None:
            part = y.new_empty(0)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)          # (autograd would zero-fill a [P, M, 2] "gradient" of the partials per launch)
        return y, part

    @staticmethod
    def backward(ctx, gy, _gpart=None):
        x, w2d = ctx.saved_tensors
        if gy is None:
            return (None,) * 5
        gy = gy.contiguous()
        B = x.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx = dx_gemm(w2d, gy)
        if ctx.needs_input_grad[2]:
            M, K = w2d.shape
            L = x.shape[2]
            if cfg.SPIKE_GEMM_DW and L % 4 == 0 and M >= 16:     # 32- / 64- / 128-row tiles by M
                sink = _sink_for(w2d)
                gw = torch.empty(M, K, dtype=torch.float32, device=x.device) if sink is None else None
                xb = x.dtype == torch.bfloat16
                if (cfg.DEFER_DW and sink is not None and xb and B * L <= cfg.DEFER_DW_MAX_CONTRACTION and cfg.WGRAD_STREAM is None
                        and x.data_ptr() % 8 == 0):
                    _defer_dw(gy, x, sink, B, M, K, L)
                    return _grad_pair(ctx.has_tok, gx) + (None, gy.sum((0, 2)) if (ctx.has_bias and ctx.needs_input_grad[3]) else None,
                                                          None)
                _time_next("spike_gemm_dw", 4 * B * L * (K + M), 2 * B * M * L * K, moved=B * L * ((2 if xb else 4) * K + 4 * M))
                side = _wgrad_stream(sink, gy, x)
                st = side.cuda_stream if side is not None else _stream()
                if (xb and cfg.DW_PIPE and cfg.DW_PIPE_SINGLE and M >= 128 and K >= 128 and lib.s2f_spike_gemm_dw_pipe_ok(B, M, K, L)
                        and gy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
                    # long contractions launch on their own: the LDS-DMA pipeline where its 128 x 256 tile is filled
                    check(lib.s2f_spike_gemm_dw_pipe(_ptr(gy), _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                     int(sink is not None), 0, 0, st), "s2f_spike_gemm_dw_pipe")
                elif xb:
                    check(lib.s2f_spike_gemm_dw_bf16(_ptr(gy), _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                     int(sink is not None), st), "s2f_spike_gemm_dw_bf16")
                else:
                    check(lib.s2f_spike_gemm_dw(_ptr(gy), _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                int(sink is not None), 1, st), "s2f_spike_gemm_dw")
            else:
                gw = bmm_small(gy, x.float().transpose(1, 2), reduce_batch=True)          # shapes the matrix-core kernels do not take
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = gy.sum((0, 2))
        return _grad_pair(ctx.has_tok, gx) + (gw, gb, None)


def dx_gemm(w2d, gy):
    """Input gradient of a 1x1 convolution: gx[b] = W^T @ gy[b]  (two general fp32 operands): the transposed product on the
    forward pack of W with gy split hi + mid + lo in the kernel (s2f_pgemm_dx_f32, 6 passes)."""
    B, M, N = gy.shape
    if cfg.PGEMM_DX and N % 4 == 0 and gy.is_cuda:
        K = w2d.shape[1]
        gx = torch.empty(B, K, N, dtype=torch.float32, device=gy.device)
        _time_next("dx_gemm", 4 * B * N * (K + M), 2 * B * M * N * K)
        check(lib.s2f_pgemm_dx_f32(_ptr(pack_weight(w2d)), _ptr(gy), 0, _ptr(gx), 0, B, M, K, N, 0.0, 0, _stream()),
              "s2f_pgemm_dx_f32")
        return gx
    return bmm_small(w2d.t().unsqueeze(0).expand(B, -1, -1), gy)          # N % 4 != 0 (10-query rows of the plumbing configuration)


def gemm_bn_lif_eval_ok(x, N):
    """The eval-mode fusion takes bf16 spikes with N % 4 == 0, N >= 8 (the decoder's 100-token maps too), and builds no autograd
    graph."""
    return (cfg.PGEMM and isinstance(x, Spikes) and x.data.dtype == torch.bfloat16 and x.data.is_cuda and N % 4 == 0
            and N >= 8 and x.data.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and x.requires_grad))


def gemm_bn_lif_eval(x, w2d, conv_bias, running_mean, running_var, gamma, beta, eps, residual=None, want_pre=False, lif=False,
                     v_in=None, keep_v=False, D=8, vth=1.0, stats=None):
    """Eval-mode  conv1x1 -> BatchNorm(running statistics) [+ residual] [-> Q_IFNode]  as ONE launch (s2f_gemm_bn_lif_fwd: the
    packed-weight GEMM with the BatchNorm and neuron arithmetic in its epilogue).  x: bf16 Spikes
    [B, K, N].  -> (u fp32 or None, spikes as Spikes or None, v_out or None).  No backward: inference only."""
    data = x.data.contiguous()
    B, K, N = data.shape
    M = w2d.shape[0]
    dev = data.device
    with torch.no_grad():
        u = torch.empty(B, M, N, dtype=torch.float32, device=dev) if want_pre else None
        y = torch.empty(B, M, N, dtype=torch.bfloat16, device=dev) if lif else None
        v_out = torch.empty(B, M, N, dtype=torch.float32, device=dev) if (lif and keep_v) else None
        if residual is not None:
            residual = residual.contiguous()
        if v_in is not None:
            v_in = v_in.contiguous()
        _time_next("gemm_bn_lif", 4 * B * N * (K + M), 2 * B * M * N * K, moved=B * N * (2 * K + (4 if want_pre else 0) + (2 if lif else 0)))
        check(lib.s2f_gemm_bn_lif_fwd(_ptr(pack_weight(w2d)), _ptr(data), _ptr(conv_bias), _ptr(running_mean), _ptr(running_var),
                                      _ptr(gamma), _ptr(beta), float(eps), _ptr(residual), _ptr(u), _ptr(v_in), _ptr(y), _ptr(v_out),
                                      _ptr(stats), B, M, N, K, float(vth), int(D), _stream()), "s2f_gemm_bn_lif_fwd")
    return u, (Spikes(y, _new_tok(y)) if lif else None), v_out


def dense_gemm_bn_lif_eval_ok(x, N):
    """The eval-mode fusion of a 1x1 convolution whose input is a general fp32 map (not spikes): N % 4 == 0, at least one 128-column
    tile, no autograd graph."""
    return (cfg.PGEMM_DX and torch.is_tensor(x) and x.dtype == torch.float32 and x.is_cuda and N % 4 == 0 and N >= cfg.PGEMM_MIN_N
            and x.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and x.requires_grad))


def dense_gemm_bn_lif_eval(x, ws, conv_bias, running_mean, running_var, gamma, beta, eps, residual=None, want_pre=False, lif=False,
                           D=8, vth=1.0, stats=None):
    """Eval-mode  conv1x1 (dense fp32 input; G = len(ws) weights [M, K] on consecutive channel groups) -> BatchNorm(running
    statistics) [+ residual] [-> Q_IFNode, reset]  as ONE launch (s2f_dense_gemm_bn_lif_fwd; row f4).  x [B, G K, N] fp32; the
    per-channel vectors [G M].  -> (u fp32 [B, G M, N] or None, Spikes or None).  No backward: inference only."""
    import ctypes
    if torch.is_tensor(ws):
        ws = [ws] if ws.dim() == 2 else list(ws.unbind(0))
    G = len(ws)
    M, K = ws[0].shape
    x = x.contiguous()
    B, GK, N = x.shape
    assert GK == G * K and 1 <= G <= 4
    dev = x.device
    with torch.no_grad():
        u = torch.empty(B, G * M, N, dtype=torch.float32, device=dev) if want_pre else None
        y = torch.empty(B, G * M, N, dtype=torch.bfloat16, device=dev) if lif else None
        if residual is not None:
            residual = residual.contiguous()
        packs = (ctypes.c_void_p * G)(*[pack_weight(w, transposed=True).data_ptr() for w in ws])
        _time_next("dx_gemm", 4 * B * N * G * (K + M), 2 * B * G * M * N * K)
        check(lib.s2f_dense_gemm_bn_lif_fwd(packs, G, _ptr(x), G * K * N, K * N, _ptr(conv_bias), _ptr(running_mean), _ptr(running_var),
                                            _ptr(gamma), _ptr(beta), float(eps), _ptr(residual), _ptr(u), _ptr(y), _ptr(stats),
                                            B, K, M, N, float(vth), int(D), _stream()), "s2f_dense_gemm_bn_lif_fwd")
    return u, (Spikes(y, _new_tok(y)) if lif else None)


def conv3x3_bn_lif_eval(x, weight, running_mean, running_var, gamma, beta, eps, residual=None, want_pre=False, lif=False, v_in=None,
                        keep_v=False, D=8, vth=1.0, stats=None):
    """Eval-mode  conv3x3 (stride 1, padding 1, no bias) -> BatchNorm(running statistics) [+ residual] [-> Q_IFNode]  as ONE launch
    (s2f_conv3x3_bn_lif_fwd; row f4).  x: bf16 Spikes [B, C, H, W]; weight [M, C, 3, 3].  -> (u, spikes, v_out), [B, M, H, W]."""
    data = x.data.contiguous()
    B, C, H, W = data.shape
    M = weight.shape[0]
    dev = data.device
    with torch.no_grad():
        u = torch.empty(B, M, H, W, dtype=torch.float32, device=dev) if want_pre else None
        y = torch.empty(B, M, H, W, dtype=torch.bfloat16, device=dev) if lif else None
        v_out = torch.empty(B, M, H, W, dtype=torch.float32, device=dev) if (lif and keep_v) else None
        if residual is not None:
            residual = residual.contiguous()
        if v_in is not None:
            v_in = v_in.contiguous()
        n = B * H * W
        _time_next("gemm_bn_lif", 4 * n * (C + M), 2 * n * M * C * 9, moved=n * (2 * C + (4 if want_pre else 0) + (2 if lif else 0)))
        check(lib.s2f_conv3x3_bn_lif_fwd(_ptr(pack_weight_conv3(weight)), _ptr(data), 0, _ptr(running_mean), _ptr(running_var),
                                         _ptr(gamma), _ptr(beta), float(eps), _ptr(residual), _ptr(u), _ptr(v_in), _ptr(y), _ptr(v_out),
                                         _ptr(stats), B, M, C, H, W, float(vth), int(D), _stream()), "s2f_conv3x3_bn_lif_fwd")
    return u, (Spikes(y, _new_tok(y)) if lif else None), v_out


def _with_part(y, part):
    if part.numel():
        y._s2f_part = (part, y._version, y.data_ptr())
    return y


def spike_gemm(x, w2d, bias=None, stats=False):
    """x: Spikes or an fp32 spike tensor [B, K, N].  stats: a train-mode BatchNorm follows -- store its partial statistics with
    the output (attribute `_s2f_part`, see cfg.BN_PARTIALS) when the kernel path can"""
    data, tok = _unpack(x)
    return _with_part(*_SpikeGemm.apply(data, tok, w2d, bias, bool(stats)))


# ------------------------------------------------------------------------------------------------ token-major linear layers
class _LinearTM(torch.autograd.Function):
    """nn.Linear on a TOKEN-major activation, y[n, o] = sum_c x[n, c] W[o, c] + b[o]  (the SDME block's cls_embed and mask-embedding
    MLP, mmdet dense_heads/maskformer_head.py:568-582, SNN_core.py:95-123; ~5 600 tokens of 256 channels), on this package's kernels
    instead of rocBLAS: both operands of y are contraction-contiguous -- the layout of the general weight-gradient kernel
    (s2f_gemm_dw_general, 6 bf16 passes = fp32 accuracy), which also serves dX = dY W (on W^T); dW = dY^T X contracts over the
    tokens: the transposed packed-weight kernel with dY packed on the fly (s2f_pgemm_dx_f32, contraction split over gridDim.z)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x, w, b)
        x = x.contiguous()
        n, c = x.shape
        o = w.shape[0]
        y = b.detach().expand(n, o).contiguous() if b is not None else torch.zeros(n, o, dtype=torch.float32, device=x.device)
        # accumulate = 3: add into y (initialised with the bias) without a contraction split -- a forward product must repeat bit for bit
        check(lib.s2f_gemm_dw_general(_ptr(x), 0, _ptr(w.detach().contiguous()), 0, _ptr(y), 1, n, o, c, 3, _stream()), "s2f_gemm_dw_general")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        n, c = x.shape
        o = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = transpose_last2(w.detach().unsqueeze(0))[0]                                # [c, o]
            g2, op = gy, o
            if o % 4 != 0:                                                                  # the contraction runs in groups of 4
                op = (o + 3) // 4 * 4
                g2 = torch.nn.functional.pad(gy, (0, op - o))
                wt = torch.nn.functional.pad(wt, (0, op - o))
            gx = torch.zeros(n, c, dtype=torch.float32, device=gy.device)
            check(lib.s2f_gemm_dw_general(_ptr(g2), 0, _ptr(wt), 0, _ptr(gx), 1, n, c, op, 1, _stream()), "s2f_gemm_dw_general")
        if ctx.needs_input_grad[1]:
            sink = _sink_for(w)
            if sink is not None and sink.data_ptr() % 16 != 0:
                # the kernel adds 16-byte aligned rows.  dist.FlatGradAllReduce pads every slot to 16 bytes, so whether this weight
                # goes through its sink never depends on where compact() placed it; a foreign sink table must do the same
                raise RuntimeError("linear_tm: the gradient sink of this weight is not 16-byte aligned")
            gp = torch.empty(int(lib.s2f_pack_elems(n, o)), dtype=torch.int16, device=gy.device)
            check(lib.s2f_pack_bf16x3(_ptr(gy), _ptr(gp), n, o, 0, 0, _stream()), "s2f_pack_bf16x3")
            if sink is None:
                gw = torch.empty(o, c, dtype=torch.float32, device=gy.device)
            check(lib.s2f_pgemm_dx_f32(_ptr(gp), _ptr(x), 0, _ptr(gw if sink is None else sink), 0, 1, n, o, c,
                                       0.0 if sink is None else 1.0, 0, _stream()), "s2f_pgemm_dx_f32")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


def _mm_tm(x2d, w_oc):
    """x2d [n, c] @ w_oc[o, c]^T -> [n, o] on s2f_gemm_dw_general (both operands contraction-contiguous; no contraction split:
    repeats bit for bit); ops.bmm_small for c % 4 != 0."""
    n, c = x2d.shape
    if not (cfg.LINEAR_TM and c % 4 == 0 and x2d.is_cuda and n > 0):
        return bmm_small(x2d.unsqueeze(0), w_oc.t().unsqueeze(0))[0]
    y = torch.zeros(n, w_oc.shape[0], dtype=torch.float32, device=x2d.device)
    check(lib.s2f_gemm_dw_general(_ptr(x2d.contiguous()), 0, _ptr(w_oc.contiguous()), 0, _ptr(y), 1, n, w_oc.shape[0], c, 3, _stream()),
          "s2f_gemm_dw_general")
    return y


def _mtm_tm(a2d, b2d, out=None):
    """a2d [n, o]^T @ b2d [n, c] -> [o, c]: the contraction runs over the rows of both -- the transposed packed-operand kernel with a2d
    packed on the fly (s2f_pgemm_dx_f32, contraction split over gridDim.z); ops.bmm_small for c % 4 != 0.  `out`: a contiguous
    [o, c] fp32 destination (16-byte aligned)."""
    n, o = a2d.shape
    c = b2d.shape[1]
    if not (cfg.LINEAR_TM and c % 4 == 0 and a2d.is_cuda and n > 0):
        r = bmm_small(a2d.t().unsqueeze(0), b2d.unsqueeze(0))[0]
        return r if out is None else out.copy_(r)
    ap = torch.empty(int(lib.s2f_pack_elems(n, o)), dtype=torch.int16, device=a2d.device)
    check(lib.s2f_pack_bf16x3(_ptr(a2d.contiguous()), _ptr(ap), n, o, 0, 0, _stream()), "s2f_pack_bf16x3")
    if out is None:
        out = torch.empty(o, c, dtype=torch.float32, device=a2d.device)
    assert out.is_contiguous() and out.shape == (o, c) and out.dtype == torch.float32
    check(lib.s2f_pgemm_dx_f32(_ptr(ap), _ptr(b2d.contiguous()), 0, _ptr(out), 0, 1, n, o, c, 0.0, 0, _stream()), "s2f_pgemm_dx_f32")
    return out




class _LinearSmall(torch.autograd.Function):
    """nn.Linear with a contraction length that is no multiple of 4 (none on the shipped configurations): the three products on
    ops.bmm_small."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x, w, b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        y = bmm_small(x.unsqueeze(0), w.detach().t().unsqueeze(0))[0]
        return y + b.detach() if b is not None else y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = bmm_small(gy.unsqueeze(0), w.detach().unsqueeze(0))[0] if ctx.needs_input_grad[0] else None
        gw = bmm_small(gy.t().unsqueeze(0), x.unsqueeze(0))[0] if ctx.needs_input_grad[1] else None
        gb = gy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def linear_tm(x, weight, bias=None):
    """torch.nn.functional.linear(x, weight, bias) for x [..., c] fp32 on the GPU: the token-major matrix-core products for
    c % 4 == 0, ops.bmm_small otherwise."""
    c = x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.numel() > 0):
        fallback("linear_tm", f"c={c} dtype={x.dtype} cuda={x.is_cuda}")
        return torch.nn.functional.linear(x, weight, bias)
    fn = _LinearTM if (cfg.LINEAR_TM and c % 4 == 0) else _LinearSmall
    return fn.apply(x.reshape(-1, c), weight, bias).view(*x.shape[:-1], weight.shape[0])


# ------------------------------------------------------------------------------------------------ mask einsum (SDME)
def _split_rows(mat, slack_rows):
    """fp32 [R, K] -> bf16 terms [3, Rpad, Kpad] (s2f_split_bf16x3), Rpad >= R + slack_rows (zero rows: a row-block view of
    the matrix may over-read that many rows past its end)."""
    R, K = mat.shape
    Rpad, Kpad = (R + slack_rows + 63) // 64 * 64, (K + 31) // 32 * 32
    out = torch.empty(3, Rpad, Kpad, dtype=torch.int16, device=mat.device)
    check(lib.s2f_split_bf16x3(_ptr(mat), _ptr(out), R, K, Rpad, Kpad, _stream()), "s2f_split_bf16x3")
    return out, Rpad, Kpad


class _MaskEinsum(torch.autograd.Function):
    """out[b] = scale * sum_t E[t, b] @ MF[t, b]   (E [T,B,Q,C], MF [T,B,C,HW] -> [B,Q,HW]).

    = einsum('tbqc,tbchw->tbqhw').mean(t) of maskformer_head.py:582-583 with the mean folded into the contraction.
    `e_exact`: E is exactly representable in bf16 (the head passes alpha * spikes = multiples of 1/2): forward and
    d(mask_features) then run on the bf16 matrix cores with MF / the incoming gradient split hi+mid+lo in the kernel
    (s2f_split_gemm, 3 passes, exact products, fp32 accumulation) -- the forward as ONE GEMM of K = T*C, no partial-sum
    traffic.  dE (two general fp32 operands, K = HW) and the non-exact case stay on rocBLAS fp32.  The backward writes each
    dMF[t] / dE[t] slice straight into its final buffer: autograd's select_backward would zero-fill and add T full-size
    [T,B,C,HW] tensors (4 x 537 MB at C2)."""

    @staticmethod
    def forward(ctx, e, mf, scale, e_exact):
        _need_cuda(e, mf)
        T, B, Q, C = e.shape
        HW = mf.shape[-1]
        e = e.contiguous()
        mf = mf.contiguous()
        mfma = bool(e_exact) and HW % 4 == 0 and cfg.SPIKE_GEMM_ENABLED
        if cfg.SPIKE_GEMM_CHECK and mfma:
            assert torch.equal(e, e.bfloat16().float()), "mask_einsum: E is not exact in bf16"
        if mfma:
            acat = e.permute(1, 2, 0, 3).reshape(B * Q, T * C)                  # row (b, q), column (t, c)
            a_split, Rpad, Kpad = _split_rows(acat, 128)
            out = torch.empty(B, Q, HW, dtype=torch.float32, device=e.device)
            check(lib.s2f_split_gemm(_ptr(a_split), Q * Kpad, Rpad * Kpad, 1, _ptr(mf), C * HW, C, B * C * HW, 3, _ptr(out),
                                     Q * HW, scale, B, Q, HW, T * C, (Q + 127) // 128 * 128, Kpad, _stream()),
                  "s2f_split_gemm")
        else:
            # E not exact in bf16, or ragged rows: one strided product with the T slabs folded into the contraction
            # (row (b, q) x column (t, c) against row (t, c) x column hw), ascending-k fp32 multiply-adds
            out = bmm_small((e * scale).permute(1, 2, 0, 3).reshape(B, Q, T * C), mf.permute(1, 0, 2, 3).reshape(B, T * C, HW))
        ctx.save_for_backward(e, mf)
        ctx.scale, ctx.mfma = scale, mfma
        return out

    @staticmethod
    def backward(ctx, g):
        e, mf = ctx.saved_tensors
        g = g.contiguous()
        T, B, Q, C = e.shape
        HW = mf.shape[-1]
        ge = gmf = None
        if ctx.needs_input_grad[0]:
            if ctx.mfma and cfg.MASK_EINSUM_DE_MFMA:
                # dE[t, b] = g[b] (Q x HW) @ MF[t, b]^T: both operands contraction-contiguous fp32 -> the weight-gradient
                # kernel with both sides split hi+mid+lo (6 passes), split-K over HW with fp32 atomics
                ge = torch.zeros_like(e)
                for t in range(T):
                    for b in range(B):
                        check(lib.s2f_spike_gemm_dw(_ptr(g[b]), _ptr(mf[t, b]), _ptr(ge[t, b]), 1, Q, C, HW, 1, 3, _stream()),
                              "s2f_spike_gemm_dw")
            else:
                ge = torch.stack([bmm_small(g, mf[t].transpose(1, 2)) for t in range(T)], 0)
            ge.mul_(ctx.scale)
        if ctx.needs_input_grad[1]:
            gmf = torch.empty_like(mf)
            if ctx.mfma:
                et = e.permute(0, 1, 3, 2).reshape(T * B * C, Q)               # row (t, b, c), column q
                a_split, Rpad, Kpad = _split_rows(et, 128)
                for t in range(T):
                    check(lib.s2f_split_gemm(_ptr(a_split) + 2 * t * B * C * Kpad, C * Kpad, Rpad * Kpad, 1, _ptr(g), Q * HW, Q,
                                             0, 3, _ptr(gmf[t]), C * HW, ctx.scale, B, C, HW, Q, (C + 127) // 128 * 128, Kpad,
                                             _stream()), "s2f_split_gemm")
            else:
                es = e * ctx.scale
                for t in range(T):
                    gmf[t].copy_(bmm_small(es[t].transpose(1, 2), g))
        return ge, gmf, None, None


def mask_einsum(e, mf, scale, e_exact=False):
    return _MaskEinsum.apply(e, mf, float(scale), bool(e_exact))


class _MaskEinsumFolded(torch.autograd.Function):
    """The mask contraction of the head with the pixel decoder's mask_feature 1x1 convolution FOLDED into it:

        out[b] = scale * sum_t E[t, b] @ (W S[t, b] + bias)          (maskformer_head.py:582-583 on pixel_decoder.py:467-470)
               = scale * ( sum_t (E[t, b] W) @ S[t, b]  +  (sum_t E[t, b] bias) 1^T )

    S = mask_feature_spike's output, a bf16 spike map [T*B, C, HW]; W [Co, C], bias [Co] = the mask_feature convolution;
    E [T, B, Q, Co].  (E W) is a [Q, C] product per (t, b) -- 0.1 GFLOP -- after which the contraction runs over the SPIKES:
    the convolution's own forward (69 GFLOP at C2), its weight gradient and the 537 MB fp32 mask_features tensor it wrote for
    the einsum to read back never exist, and dE needs 3 MFMA passes (spike operand) instead of 6.  Same value as the
    reference's two steps up to the association of fp32 sums (nothing thresholds this output).
    Backward: G[t,b] = scale E[t,b]^T g[b] (3 passes, E exact in bf16) -> dS = W^T G (the convolution's input gradient);
    H[t,b] = g[b] S[t,b]^T (weight-gradient kernel on a spike operand) -> dE = scale (H W^T + rowsum(g) bias^T),
    dW = scale sum E^T H, dbias = scale sum E^T rowsum(g)."""

    @staticmethod
    def forward(ctx, e, sdata, stok, W, bias, scale, T, B, e_exact):
        _need_cuda(e, W, bias, spikes=sdata)
        Q, Co = e.shape[2], e.shape[3]
        C, HW = sdata.shape[1], sdata.shape[2]
        e = e.contiguous()
        sdata = sdata.contiguous()
        dev = e.device
        ew = _mm_tm(e.reshape(-1, Co), transpose_last2(W.detach().unsqueeze(0))[0]).view(T, B, Q, C)      # e @ W: [T, B, Q, C]
        acat = ew.permute(1, 2, 0, 3).reshape(B, Q, T * C).contiguous()       # row (b, q), column (t, c)
        rowb = None
        if bias is not None:
            rowb = (e.sum(0) * bias.view(1, 1, -1)).sum(-1).contiguous()       # [B, Q]: sum_t E[t, b] bias (a reduction, no GEMV)
        out = torch.empty(B, Q, HW, dtype=torch.float32, device=dev)
        if cfg.PGEMM and cfg.MASK_FWD_PGEMM and HW % 8 == 0 and C % 32 == 0:
            # the pipelined NN kernel (LDS-DMA fed, csrc/pgemm.hip): one pack of (E W)[b] per batch element
            pe = int(lib.s2f_pack_elems(Q, T * C))
            a_pack = torch.empty(B, pe, dtype=torch.int16, device=dev)
            for b in range(B):
                check(lib.s2f_pack_bf16x3(_ptr(acat[b]), _ptr(a_pack[b]), Q, T * C, 0, 0, _stream()), "s2f_pack_bf16x3")
            _time_next("spike_gemm_fwd", 4 * B * HW * (T * C + Q), 2 * B * Q * HW * T * C, moved=B * HW * (2 * T * C + 4 * Q))
            check(lib.s2f_pgemm_nn_bf16_ex(_ptr(a_pack), pe, _ptr(sdata), C * HW, C, B * C * HW, _ptr(rowb), Q if rowb is not None else 0,
                                           scale, _ptr(out), B, Q, HW, T * C, _stream()), "s2f_pgemm_nn_bf16_ex")
        else:
            Mpad = (Q + 255) // 256 * 256 if Q > 256 else (Q + 63) // 64 * 64
            Kpad = T * C
            a_split = torch.empty(B, 3, Mpad, Kpad, dtype=torch.int16, device=dev)
            for b in range(B):
                check(lib.s2f_split_bf16x3(_ptr(acat[b]), _ptr(a_split[b]), Q, T * C, Mpad, Kpad, _stream()), "s2f_split_bf16x3")
            _time_next("spike_gemm_fwd", 4 * B * HW * (T * C + Q), 2 * B * Q * HW * T * C, moved=B * HW * (2 * T * C + 4 * Q))
            check(lib.s2f_spike_gemm_fwd_bf16_ex(_ptr(a_split), 3 * Mpad * Kpad, _ptr(sdata), C * HW, C, B * C * HW, _ptr(rowb),
                                                 Q if rowb is not None else 0, scale, _ptr(out), B, Q, HW, T * C, Mpad, Kpad, _stream()),
                  "s2f_spike_gemm_fwd_bf16_ex")
        ctx.save_for_backward(e, sdata, W, bias, ew)
        ctx.cfg = (scale, T, B, bool(e_exact))
        return out

    @staticmethod
    def backward(ctx, g):
        e, sdata, W, bias, ew = ctx.saved_tensors
        scale, T, B, e_exact = ctx.cfg
        Q, Co = e.shape[2], e.shape[3]
        C, HW = sdata.shape[1], sdata.shape[2]
        g = g.contiguous()
        dev = g.device
        gs = ge = gW = gb = None
        S = sdata.view(T, B, C, HW)
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and cfg.MASK_BWD_FOLDED and cfg.LINEAR_TM and HW % 4 == 0:
            # dS[t, b] = W^T (scale E[t, b]^T g[b]) = (scale E[t, b] W)^T g[b]: the fold of the forward, read backwards -- ONE product
            # per (t, b) over the Q queries with the [Q, C] matrix the forward already formed, instead of the [Co x Q] product into a
            # [T, B, Co, HW] intermediate (537 MB at C2) followed by the convolution's input-gradient GEMM over it.  Both operands
            # general: 6 passes of 2 C Q HW against 3 of 2 Co Q HW + 6 of 2 C Co HW -- 46 % fewer matrix-core FLOPs at C2.
            ews = ew * scale
            gs = torch.empty(T * B, C, HW, dtype=torch.float32, device=dev)
            _time_next("dx_gemm", 4 * T * B * HW * (C + Q), 2 * T * B * Q * HW * C)
            for t in range(T):
                for b in range(B):
                    _mtm_tm(ews[t, b], g[b], out=gs[t * B + b])
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # dS[t, b] = W^T (scale E[t, b]^T g[b]): the first product as in _MaskEinsum (E exact in bf16: 3 passes), the
            # second is the mask_feature convolution's input gradient
            G = torch.empty(T, B, Co, HW, dtype=torch.float32, device=dev)
            if e_exact and HW % 4 == 0:
                et = e.permute(0, 1, 3, 2).reshape(T * B * Co, Q)
                a_split, Rpad, Kp = _split_rows(et, 128)
                for t in range(T):
                    check(lib.s2f_split_gemm(_ptr(a_split) + 2 * t * B * Co * Kp, Co * Kp, Rpad * Kp, 1, _ptr(g), Q * HW, Q, 0, 3,
                                             _ptr(G[t]), Co * HW, scale, B, Co, HW, Q, (Co + 127) // 128 * 128, Kp, _stream()),
                          "s2f_split_gemm")
            else:
                es = e * scale
                for t in range(T):
                    G[t].copy_(bmm_small(es[t].transpose(1, 2), g))
            gs = dx_gemm(W, G.view(T * B, Co, HW))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            H = torch.empty(T, B, Q, C, dtype=torch.float32, device=dev)
            xb = sdata.dtype == torch.bfloat16
            _time_next("spike_gemm_dw", 4 * T * B * HW * (C + Q), 2 * T * B * Q * HW * C, moved=T * B * HW * ((2 if xb else 4) * C + 4 * Q))
            if xb and cfg.MASK_EINSUM_DW_GROUPED and T * B <= 56 and HW % 4 == 0:
                # the T * B products H[t, b] = g[b] S[t, b]^T as ONE grouped launch (12 output tiles each over a 65 536-long
                # contraction: one by one they run at 183 TF/s) into the zeroed H
                import ctypes
                H.zero_()
                flat = []
                for t in range(T):
                    for b in range(B):
                        flat += [g[b].data_ptr(), S[t, b].data_ptr(), H[t, b].data_ptr(), 1, Q, C, HW]
                arr = (ctypes.c_int64 * len(flat))(*flat)
                if (cfg.DW_PIPE and lib.s2f_spike_gemm_dw_pipe_ok(1, Q, C, HW) and g.data_ptr() % 16 == 0 and S.data_ptr() % 16 == 0
                        and (Q * HW) % 4 == 0 and (C * HW) % 8 == 0 and (HW % 32 == 0 or cfg.DWP_SCHEDULE == 0)):
                    check(lib.s2f_spike_gemm_dw_pipe_grouped(arr, T * B, cfg.DWP_SCHEDULE, cfg.DWP_WGS, _stream()), "s2f_spike_gemm_dw_pipe_grouped")
                else:
                    check(lib.s2f_spike_gemm_dw_grouped(arr, T * B, 64, _stream()), "s2f_spike_gemm_dw_grouped")
            else:
                for t in range(T):
                    for b in range(B):
                        if xb:
                            check(lib.s2f_spike_gemm_dw_bf16(_ptr(g[b]), _ptr(S[t, b]), _ptr(H[t, b]), 1, Q, C, HW, 0, _stream()),
                                  "s2f_spike_gemm_dw_bf16")
                        else:
                            check(lib.s2f_spike_gemm_dw(_ptr(g[b]), _ptr(S[t, b]), _ptr(H[t, b]), 1, Q, C, HW, 0, 1, _stream()),
                                  "s2f_spike_gemm_dw")
            rs = channel_sum(g.view(1, B * Q, HW)).view(B, Q) if bias is not None else None                      # [B, Q]
            if ctx.needs_input_grad[0]:
                ge = _mm_tm(H.view(-1, C), W).view(T, B, Q, Co)                                   # H @ W^T
                if bias is not None:
                    ge = ge + rs.unsqueeze(0).unsqueeze(-1) * bias.view(1, 1, 1, -1)
                ge = ge * scale
            if ctx.needs_input_grad[3]:
                gW = _mtm_tm(e.reshape(-1, Co), H.view(-1, C)) * scale                             # sum_{t,b,q} e^T H
            if bias is not None and ctx.needs_input_grad[4]:
                gb = (e * rs.view(1, B, Q, 1)).sum((0, 1, 2)) * scale
        return (ge,) + _grad_pair(True, gs) + (gW, gb, None, None, None, None)


def class_mask_product(cls_score, mask_probs):
    """einsum('bqc,bqhw->bchw') of the inference post-processing (mmseg decode_heads/maskformer_head.py:176-178): per image the
    [K x Q] x [Q x HW] product of two general fp32 operands on the transposed packed-operand kernel (6 bf16 passes = fp32 accuracy;
    the class scores packed on the fly) instead of the vendor GEMM the einsum lowers to.  No autograd (inference glue)."""
    B, Q, K = cls_score.shape
    h, w = mask_probs.shape[-2:]
    if not cls_score.is_cuda:
        # host tensors: the a13 callers are Python glue in the reference too and are pinned on the CPU bit for bit
        # (tests/test_a13_callers.py); nothing of the hot path runs there
        return torch.einsum("bqc,bqhw->bchw", cls_score, mask_probs)
    if torch.is_grad_enabled() and (cls_score.requires_grad or mask_probs.requires_grad):
        fallback("class_mask_product", "autograd wanted through the inference post-processing")
        return torch.einsum("bqc,bqhw->bchw", cls_score, mask_probs)
    if not (cfg.LINEAR_TM and cls_score.is_cuda and (h * w) % 4 == 0 and cls_score.dtype == torch.float32):
        return bmm_small(cls_score.transpose(1, 2), mask_probs.reshape(B, Q, h * w)).view(B, K, h, w)
    out = torch.empty(B, K, h, w, dtype=torch.float32, device=cls_score.device)          # every image's product lands in its slice
    for b in range(B):
        _mtm_tm(cls_score[b], mask_probs[b].reshape(Q, h * w), out=out[b].view(K, h * w))
    return out


def mask_einsum_folded(e, spikes, W, bias, scale, T, B, e_exact=False):
    """e [T, B, Q, Co], spikes: bf16 Spikes [T*B, C, HW] (mask_feature_spike's output), W [Co, C], bias [Co] or None -> [B, Q, HW]"""
    assert isinstance(spikes, Spikes) and spikes.tok is not None and spikes.data.dtype == torch.bfloat16
    return _MaskEinsumFolded.apply(e, spikes.data, spikes.tok, W, bias, float(scale), int(T), int(B), bool(e_exact))



__all__ = [n for n in dir() if not n.startswith('__')]
