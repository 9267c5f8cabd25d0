"""The residual ATen calls of a step on this package's kernels.

What the module code and autograd still ask ATen for inside a step is glue: gradient accumulation where two consumers of a tensor meet
(`at::add` issued by the engine itself), scalar multiples (`alpha * spikes`), `sigmoid`, layout copies and dtype casts, `torch.stack` /
`cat`, zero fills, small sums -- ~190 launches per C2 step that `S2F_STRICT` never saw because they are no GEMMs or convolutions.
`GlueMode` is a `TorchDispatchMode` (it travels with autograd's thread-local state, so it also sees the engine's own calls in the
backward pass) that

  * routes those calls to the generic strided kernels of csrc/glue.hip (`s2f_ew`, `s2f_reduce_sum`, `s2f_fill`): the same IEEE operations
    as ATen's element-wise kernels (bit-identical results for add / mul / div / copy / fill), sums in a fixed order;
  * lets view / metadata ops through (they launch nothing);
  * counts everything else that touches a CUDA tensor in `UNROUTED` -- and raises under `cfg.STRICT_GLUE`.

It costs host time per call (Python dispatch), which a captured hipGraph does not replay: `graph.GraphedStep` and friends enter it
for their warm-up and capture when `cfg.GLUE_MODE` is on, so the REPLAYED step consists of this package's kernels only
(`tools/rocpd_categories.py` counts `at::native` launches in the trace; `S2F_FORBID_ATEN=1` makes any a failure)."""
import collections
import ctypes

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .._lib import check, lib
from .config import cfg

aten = torch.ops.aten

# ops that launch nothing (views, metadata, allocation)
VIEW_OPS = {"view", "_unsafe_view", "reshape", "_reshape_alias", "expand", "permute", "transpose", "t", "select", "slice", "unbind", "detach",
            "alias", "as_strided", "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "unsqueeze", "squeeze", "split",
            "split_with_sizes", "unflatten", "flatten", "_local_scalar_dense", "is_same_size", "set_", "lift_fresh", "lift", "view_as",
            "unsafe_split", "chunk", "narrow", "movedim", "result_type", "size", "stride", "is_contiguous", "numel", "storage_offset",
            "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "dim", "is_pinned", "record_stream", "resize_", "unfold",
            "diagonal", "real", "is_non_overlapping_and_dense", "is_strides_like_format", "_has_compatible_shallow_copy_type",
            "item", "prim_layout"}

# aten calls that stay on ATen by decision (counted in ALLOWED, never an error): the gradient packing's batched copy of ~650 pieces into
# the flat buffer (dist.FlatGradAllReduce.gather: ATen's CatArrayBatchedCopy does it in ~6 launches; a launch per piece would cost more),
# and the BatchNorms' num_batches_tracked counters (one _foreach_add_ over int64 scalars)
ALLOW = {"aten.cat.out", "aten._foreach_add_.Scalar"}
ALLOWED = collections.Counter()
ROUTED = collections.Counter()
UNROUTED = collections.Counter()
_I64x6 = ctypes.c_int64 * 6


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ok(t):
    return torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0 and t.dim() <= 6 and not t.is_conj()


def _arr(v):
    v = list(v) + [1] * (6 - len(v))
    return _I64x6(*[int(x) for x in v])


def _coalesce(shape, strides):
    """merge adjacent dimensions that every operand walks contiguously (stride[i] == stride[i + 1] * size[i + 1] for all of them) and
    drop dimensions of extent 1: fewer index divisions per element in the strided kernel"""
    dims = [(n, tuple(st[i] for st in strides)) for i, n in enumerate(shape) if n != 1]
    if not dims:
        return [1], [[0] for _ in strides]
    out = [dims[0]]
    for n, st in dims[1:]:
        pn, pst = out[-1]
        if all(ps == s_ * n for ps, s_ in zip(pst, st)):
            out[-1] = (pn * n, st)
        else:
            out.append((n, st))
    return [n for n, _ in out], [[st[k] for _, st in out] for k in range(len(strides))]


def _ew(op, a, b, out, alpha=1.0, beta=0.0, a_bf16=False):
    """out[...] = f(a, b) over out's shape; a, b already expanded to it (stride 0 = broadcast)"""
    shape = tuple(out.shape)
    strides = [list(a.stride()), list(b.stride()) if b is not None else [0] * len(shape), list(out.stride())]
    size, (sa, sb, so) = _coalesce(shape, strides)
    nd = len(size)
    flat = (nd == 1 and not a_bf16 and sa[0] == 1 and so[0] == 1 and (b is None or sb[0] == 1) and a.data_ptr() % 16 == 0
            and out.data_ptr() % 16 == 0 and (b is None or b.data_ptr() % 16 == 0))
    check(lib.s2f_ew(op, a.data_ptr(), 0 if b is None else b.data_ptr(), out.data_ptr(), nd, _arr(size), _arr(sa), _arr(sb), _arr(so),
                     float(alpha), float(beta), int(a_bf16), int(flat), _stream()), "s2f_ew")
    return out


def _binary(op, a, b, alpha=1.0, out=None):
    """a, b fp32 CUDA tensors (broadcastable) -> op(a, b); `out`: in-place destination (= a)"""
    if not (_ok(a) and _ok(b)) or a.device != b.device:
        return NotImplemented
    shape = torch.broadcast_shapes(a.shape, b.shape)
    if len(shape) > 6:
        return NotImplemented
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=a.device)
    elif tuple(out.shape) != tuple(shape):
        return NotImplemented
    return _ew(op, a.expand(shape), b.expand(shape), out, alpha)


def _scalar(x):
    return isinstance(x, (int, float, bool)) or (torch.is_tensor(x) and x.dim() == 0 and not x.is_cuda)


def h_add(a, b, *, alpha=1):
    if _scalar(b) and _ok(a):
        return _ew(6, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device), 1.0, float(b) * float(alpha))
    if _scalar(a) and _ok(b):
        return _ew(6, b, None, torch.empty(b.shape, dtype=torch.float32, device=b.device), float(alpha), float(a))
    return _binary(1, a, b, alpha)


def h_add_(a, b, *, alpha=1):
    if _scalar(b) and _ok(a):
        return _ew(6, a, None, a, 1.0, float(b) * float(alpha))
    return _binary(1, a, b, alpha, out=a)


def h_sub(a, b, *, alpha=1):
    if _scalar(b) and _ok(a):
        return _ew(6, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device), 1.0, -float(b) * float(alpha))
    return _binary(8, a, b, alpha)


def h_mul(a, b):
    if _scalar(b) and _ok(a):
        return _ew(6, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device), float(b), 0.0)
    if _scalar(a) and _ok(b):
        return _ew(6, b, None, torch.empty(b.shape, dtype=torch.float32, device=b.device), float(a), 0.0)
    return _binary(2, a, b)


def h_mul_(a, b):
    if _scalar(b) and _ok(a):
        return _ew(6, a, None, a, float(b), 0.0)
    return _binary(2, a, b, out=a)


def h_div(a, b):
    if _scalar(b) and _ok(a):
        return _ew(7, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device), float(b), 0.0)
    return _binary(3, a, b)


def h_div_(a, b):
    if _scalar(b) and _ok(a):
        return _ew(7, a, None, a, float(b), 0.0)
    return _binary(3, a, b, out=a)


def h_neg(a):
    if not _ok(a):
        return NotImplemented
    return _ew(6, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device), -1.0, 0.0)


def h_addcmul(a, t1, t2, *, value=1):
    prod = _binary(2, t1, t2)
    if prod is NotImplemented:
        return NotImplemented
    return _binary(1, a, prod, value)


def h_sigmoid(a):
    if not _ok(a):
        return NotImplemented
    return _ew(4, a, None, torch.empty(a.shape, dtype=torch.float32, device=a.device))


def h_sigmoid_backward(g, y):
    return _binary(5, g, y)


def _like(x, memory_format=None):
    if memory_format in (None, torch.preserve_format):
        return torch.empty_like(x, dtype=torch.float32)
    return torch.empty(x.shape, dtype=torch.float32, device=x.device)


def h_clone(x, *, memory_format=None):
    if not _ok(x):
        return NotImplemented
    return _ew(0, x, None, _like(x, memory_format))


def h_copy_(dst, src, non_blocking=False):
    if not (_ok(dst) and torch.is_tensor(src) and src.is_cuda and src.device == dst.device and src.numel() > 0
            and src.dtype in (torch.float32, torch.bfloat16) and src.dim() <= 6):
        return NotImplemented
    try:
        s = src.expand(dst.shape)
    except RuntimeError:
        return NotImplemented
    _ew(0, s, None, dst, a_bf16=src.dtype == torch.bfloat16)
    return dst


def h_to_copy(x, *, dtype=None, layout=None, device=None, pin_memory=None, non_blocking=False, memory_format=None):
    if not (torch.is_tensor(x) and x.is_cuda and x.numel() > 0 and x.dim() <= 6 and dtype == torch.float32 and x.dtype in (torch.bfloat16, torch.float32)
            and (device is None or torch.device(device) == x.device) and layout in (None, torch.strided)):
        return NotImplemented
    return _ew(0, x, None, _like(x, memory_format), a_bf16=x.dtype == torch.bfloat16)


def _fill(t, value):
    if not (_ok(t)):
        return NotImplemented
    if t.is_contiguous():
        check(lib.s2f_fill(t.data_ptr(), t.numel(), 0, float(value), _stream()), "s2f_fill")
        return t
    return _ew(6, t, None, t, 0.0, float(value))


def h_zero_(t):
    if (torch.is_tensor(t) and t.is_cuda and t.numel() > 0 and t.is_contiguous() and t.dtype != torch.float32
            and (t.numel() * t.element_size()) % 4 == 0 and t.data_ptr() % 4 == 0):
        # any dtype whose zero is all-zero bytes (fp64 reduction arena, int64 counters, bf16 maps): cleared as 4-byte words
        check(lib.s2f_fill(t.data_ptr(), t.numel() * t.element_size() // 4, 0, 0.0, _stream()), "s2f_fill")
        return t
    return _fill(t, 0.0)


def h_fill_(t, value):
    if not _scalar(value):
        return NotImplemented
    return _fill(t, float(value))


def _new(size, dtype, device, value):
    if dtype not in (None, torch.float32) or device is None or torch.device(device).type != "cuda":
        return NotImplemented
    t = torch.empty(tuple(size), dtype=torch.float32, device=device)
    if t.numel() == 0:
        return t
    return _fill(t, value)


def h_zeros(size, *, dtype=None, layout=None, device=None, pin_memory=None):
    return _new(size, dtype, device, 0.0)


def h_ones(size, *, dtype=None, layout=None, device=None, pin_memory=None):
    return _new(size, dtype, device, 1.0)


def h_full(size, fill_value, *, dtype=None, layout=None, device=None, pin_memory=None):
    if not _scalar(fill_value):
        return NotImplemented
    return _new(size, dtype, device, float(fill_value))


def h_zeros_like(x, *, dtype=None, layout=None, device=None, pin_memory=None, memory_format=None):
    if not (_ok(x) and dtype in (None, torch.float32) and device is None):
        return NotImplemented
    return _fill(_like(x, memory_format), 0.0)


def h_ones_like(x, *, dtype=None, layout=None, device=None, pin_memory=None, memory_format=None):
    if not (_ok(x) and dtype in (None, torch.float32) and device is None):
        return NotImplemented
    return _fill(_like(x, memory_format), 1.0)


def _reduce(x, dims, keepdim, scale):
    if not _ok(x):
        return NotImplemented
    nd = x.dim()
    dims = sorted({d % nd for d in dims}) if nd else []
    if nd == 0 or not dims:
        return NotImplemented
    keep = [d for d in range(nd) if d not in dims]
    if len(keep) > 6 or len(dims) > 6:
        return NotImplemented
    out_shape = [x.shape[d] for d in keep]
    out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
    n_red = 1
    for d in dims:
        n_red *= x.shape[d]
    ws = torch.empty(int(lib.s2f_reduce_sum_workspace(max(out.numel(), 1), n_red)), dtype=torch.float32, device=x.device)
    check(lib.s2f_reduce_sum(x.data_ptr(), out.data_ptr(), ws.data_ptr(), len(keep), _arr(out_shape), _arr([x.stride(d) for d in keep]),
                             _arr(list(out.stride())), len(dims), _arr([x.shape[d] for d in dims]), _arr([x.stride(d) for d in dims]),
                             (1.0 / n_red) if scale == "mean" else 1.0, _stream()), "s2f_reduce_sum")
    if keepdim:
        shape = [1 if d in dims else x.shape[d] for d in range(nd)]
        out = out.view(shape)
    return out


def h_sum_dim(x, dim, keepdim=False, *, dtype=None):
    if dtype not in (None, torch.float32):
        return NotImplemented
    if dim is None or len(dim) == 0:
        dim = list(range(x.dim()))
    return _reduce(x, dim, keepdim, "sum")


def h_sum(x, *, dtype=None):
    if dtype not in (None, torch.float32) or not _ok(x) or x.dim() == 0:
        return NotImplemented
    return _reduce(x, list(range(x.dim())), False, "sum")


def h_mean_dim(x, dim, keepdim=False, *, dtype=None):
    if dtype not in (None, torch.float32):
        return NotImplemented
    if dim is None or len(dim) == 0:
        dim = list(range(x.dim()))
    return _reduce(x, dim, keepdim, "mean")


def h_mean(x, *, dtype=None):
    if dtype not in (None, torch.float32) or not _ok(x) or x.dim() == 0:
        return NotImplemented
    return _reduce(x, list(range(x.dim())), False, "mean")


def _segments(out, pieces):
    """contiguous pieces copied back to back into the contiguous `out`: one launch per eight pieces (s2f_copy_segments)"""
    at = 0
    for i in range(0, len(pieces), 8):
        grp = pieces[i:i + 8]
        srcs = (ctypes.c_void_p * 8)(*([t.data_ptr() for t in grp] + [0] * (8 - len(grp))))
        ns = (ctypes.c_int64 * 8)(*([t.numel() for t in grp] + [0] * (8 - len(grp))))
        check(lib.s2f_copy_segments(out.data_ptr() + 4 * at, srcs, ns, len(grp), _stream()), "s2f_copy_segments")
        at += sum(t.numel() for t in grp)
    return out


def _seg_ok(ts):
    return all(t.is_contiguous() and t.data_ptr() % 16 == 0 and t.numel() % 4 == 0 for t in ts)


def h_cat(tensors, dim=0):
    ts = [t for t in tensors if not (t.dim() == 1 and t.numel() == 0)]
    if not ts or not all(_ok(t) for t in ts) or len({t.dim() for t in ts}) != 1:
        return NotImplemented
    nd = ts[0].dim()
    dim = dim % nd
    shape = list(ts[0].shape)
    shape[dim] = sum(t.shape[dim] for t in ts)
    out = torch.empty(shape, dtype=torch.float32, device=ts[0].device)
    if dim == 0 and _seg_ok(ts):
        return _segments(out, ts)
    at = 0
    for t in ts:
        _ew(0, t, None, out.narrow(dim, at, t.shape[dim]))
        at += t.shape[dim]
    return out


def h_cat_out(tensors, dim=0, *, out):
    ts = [t for t in tensors if not (t.dim() == 1 and t.numel() == 0)]
    if not ts or not all(_ok(t) for t in ts) or not _ok(out) or len({t.dim() for t in ts}) != 1:
        return NotImplemented
    dim = dim % ts[0].dim()
    if out.dim() != ts[0].dim() or out.shape[dim] != sum(t.shape[dim] for t in ts):
        return NotImplemented
    if len(ts) > 64:
        return NotImplemented          # (hundreds of pieces: one launch per piece would cost more than ATen's batched copy; see dist.gather)
    at = 0
    for t in ts:
        _ew(0, t, None, out.narrow(dim, at, t.shape[dim]))
        at += t.shape[dim]
    return out


def h_stack(tensors, dim=0):
    ts = list(tensors)
    if not ts or not all(_ok(t) for t in ts) or len({tuple(t.shape) for t in ts}) != 1 or ts[0].dim() >= 6:
        return NotImplemented
    dim = dim % (ts[0].dim() + 1)
    shape = list(ts[0].shape)
    shape.insert(dim, len(ts))
    out = torch.empty(shape, dtype=torch.float32, device=ts[0].device)
    if dim == 0 and _seg_ok(ts):
        return _segments(out, ts)
    for i, t in enumerate(ts):
        _ew(0, t, None, out.select(dim, i))
    return out


def h_constant_pad_nd(x, pad, value=0):
    if not (_ok(x) and _scalar(value)) or len(pad) % 2 or any(p < 0 for p in pad):
        return NotImplemented
    shape = list(x.shape)
    view_at = []
    for i in range(len(pad) // 2):
        d = x.dim() - 1 - i
        shape[d] += pad[2 * i] + pad[2 * i + 1]
        view_at.append((d, pad[2 * i]))
    out = _fill(torch.empty(shape, dtype=torch.float32, device=x.device), float(value))
    v = out
    for d, lo in view_at:
        v = v.narrow(d, lo, x.shape[d])
    _ew(0, x, None, v)
    return out


def h_repeat(x, repeats):
    if not _ok(x) or len(repeats) < x.dim():
        return NotImplemented
    lead = len(repeats) - x.dim()
    xs = x.reshape((1,) * lead + tuple(x.shape))
    out = torch.empty([r * s for r, s in zip(repeats, xs.shape)], dtype=torch.float32, device=x.device)
    # out viewed as [r0, s0, r1, s1, ...], x broadcast over the r dimensions
    inter = []
    for r, s in zip(repeats, xs.shape):
        inter += [r, s]
    ov = out.view(inter)
    xe = xs.reshape([1 if i % 2 == 0 else inter[i] for i in range(len(inter))]).expand(inter)
    if len(_coalesce(inter, [list(xe.stride()), [0] * len(inter), list(ov.stride())])[0]) > 6:
        return NotImplemented
    _ew(0, xe, None, ov)
    return out


def h_flip(x, dims):
    """copy through negated strides from the far corner of the flipped dimensions"""
    if not _ok(x) or x.dim() == 0:
        return NotImplemented
    out = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return out
    sa, off = list(x.stride()), 0
    for d in {d % x.dim() for d in dims}:
        off += (x.shape[d] - 1) * sa[d]
        sa[d] = -sa[d]
    size, (ca, cb, co) = _coalesce(tuple(x.shape), [sa, [0] * x.dim(), list(out.stride())])
    if len(size) > 6:
        return NotImplemented
    check(lib.s2f_ew(0, x.data_ptr() + 4 * off, 0, out.data_ptr(), len(size), _arr(size), _arr(ca), _arr(cb), _arr(co), 1.0, 0.0, 0, 0,
                     _stream()), "s2f_ew")
    return out


def h_select_backward(g, input_sizes, dim, index):
    if not _ok(g):
        return NotImplemented
    out = _fill(torch.empty(tuple(input_sizes), dtype=torch.float32, device=g.device), 0.0)
    _ew(0, g, None, out.select(dim, index))
    return out


HANDLERS = {
    aten.add.Tensor: h_add, aten.add.Scalar: h_add, aten.add_.Tensor: h_add_, aten.add_.Scalar: h_add_,
    aten.sub.Tensor: h_sub, aten.sub.Scalar: h_sub,
    aten.mul.Tensor: h_mul, aten.mul.Scalar: h_mul, aten.mul_.Tensor: h_mul_, aten.mul_.Scalar: h_mul_,
    aten.div.Tensor: h_div, aten.div.Scalar: h_div, aten.div_.Tensor: h_div_, aten.div_.Scalar: h_div_,
    aten.neg.default: h_neg, aten.addcmul.default: h_addcmul,
    aten.sigmoid.default: h_sigmoid, aten.sigmoid_backward.default: h_sigmoid_backward,
    aten.clone.default: h_clone, aten.copy_.default: h_copy_, aten._to_copy.default: h_to_copy,
    aten.zero_.default: h_zero_, aten.fill_.Scalar: h_fill_,
    aten.zeros.default: h_zeros, aten.ones.default: h_ones, aten.full.default: h_full,
    aten.zeros_like.default: h_zeros_like, aten.ones_like.default: h_ones_like,
    aten.sum.dim_IntList: h_sum_dim, aten.sum.default: h_sum, aten.mean.dim: h_mean_dim, aten.mean.default: h_mean,
    aten.cat.default: h_cat, aten.cat.out: h_cat_out, aten.stack.default: h_stack,
    aten.constant_pad_nd.default: h_constant_pad_nd, aten.repeat.default: h_repeat,
    aten.select_backward.default: h_select_backward, aten.flip.default: h_flip,
}


def _touches_cuda(args, out):
    stack = list(args) + [out]
    while stack:
        a = stack.pop()
        if torch.is_tensor(a):
            if a.is_cuda and a.numel() > 0:
                return True
        elif isinstance(a, (list, tuple)):
            stack.extend(a)
    return False


class GlueMode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = HANDLERS.get(func)
        if h is not None:
            r = h(*args, **kwargs)
            if r is not NotImplemented:
                ROUTED[str(func)] += 1
                return r
        out = func(*args, **kwargs)
        name = func.__name__.split(".")[0]
        if name not in VIEW_OPS and _touches_cuda(args, out):
            if str(func) in ALLOW:
                ALLOWED[str(func)] += 1
                return out
            UNROUTED[str(func)] += 1
            if cfg.STRICT_GLUE:
                raise RuntimeError(f"spike2former_amd: {func} reached ATen inside a glue_mode step; S2F_STRICT_GLUE forbids that")
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def glue_mode(force=None):
    """context manager: GlueMode when cfg.GLUE_MODE (or `force`) is on, otherwise nothing"""
    on = cfg.GLUE_MODE if force is None else force
    return GlueMode() if on else _Null()


def reset_counts():
    ROUTED.clear()
    UNROUTED.clear()
    ALLOWED.clear()
