"""Shared plumbing of the op modules: launch timing, the per-step reduction arena, gradient sinks and deferred weight gradients,
concurrency helpers, spike maps (ops.Spikes), parameter groups.  Torch is plumbing here: it owns device memory and the stream; all
arithmetic of the ops happens in libs2f_hip.so.  Every op raises on non-CUDA input -- there is no fallback."""
import torch

from .._lib import check, lib
from .config import cfg


# When set to a list, the launches of the neuron / BatchNorm / spike-GEMM kernels are timed with the dispatch packets' own
# begin / end timestamps (include/s2f.h "measurement": s2f_time_next_call) and (name, algorithmic_bytes, flops, start,
# stop) is appended -- bench.py's live roofline measurement.  Algorithmic bytes are SURVEY section 8d's per-element
# figures: LIF forward 8 B (read x, write y), backward 12 B (read gy, read x, write gx).  Eager launches only.


def _time_next(name, nbytes, flops=0, moved=None):
    """nbytes: ALGORITHMIC bytes (the fp32 figures of the reference's tensors); moved: the bytes this build's kernel
    actually reads + writes for them (bf16 spike maps), reported next to the algorithmic figure; flops: ALGORITHMIC flops
    (2 M N K per GEMM -- not the 3 / 6 bf16 passes issued for them)."""
    if cfg.KERNEL_EVENTS is not None:
        e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
        if not e0 or not e1:
            raise RuntimeError("s2f_event_create failed: " + lib.s2f_last_error().decode())
        lib.s2f_time_next_call(e0, e1)
        cfg.KERNEL_EVENTS.append((name, int(nbytes), int(flops), e0, e1, int(nbytes if moved is None else moved)))


def drain_kernel_events():
    """-> [(name, algorithmic_bytes, algorithmic_flops, microseconds, moved_bytes)] of the launches timed since cfg.KERNEL_EVENTS
    was set; synchronises, frees the events and switches the timing off."""
    import ctypes
    torch.cuda.synchronize()
    out, us = [], ctypes.c_double()
    for name, nbytes, flops, e0, e1, moved in (cfg.KERNEL_EVENTS or []):
        check(lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(us)), "s2f_event_elapsed_us")
        out.append((name, nbytes, flops, us.value, moved))
        lib.s2f_event_destroy(e0)
        lib.s2f_event_destroy(e1)
    cfg.KERNEL_EVENTS = None
    return out


# One zero-initialised fp64 arena serves every per-channel reduction workspace of a step (BatchNorm statistics, BatchNorm
# backward sums): `begin_step()` clears the used prefix with ONE memset and rewinds the cursor; each op takes a slice.
# Regions are transient (produced and consumed inside one op call), so reuse across steps only needs them zero again.
# Without `begin_step()` (stand-alone use of an op) every request falls back to a fresh torch.zeros.
_ARENA = {"buf": None, "pos": 0, "high": 0, "armed": False}
_ARENA_DOUBLES = 1 << 20


def begin_step(device=None):
    # a step that is being captured into a hipGraph re-splits every weight first (one launch, recorded in the graph): the
    # replays then read the live fp32 weights instead of the bf16 terms of capture time
    from . import gemm as _g          # (the weight-conversion caches live with the GEMM ops)
    _g._TRUST_ALL[0] = False
    if device is not None and torch.cuda.is_current_stream_capturing() and cfg.RESPLIT_IN_GRAPH:
        n = _g.resplit_all(device, build=False)
        if n > 0:
            _g._TRUST_ALL[0] = True
        elif n < 0:
            # no job table for the current set of weights: forget every cached version instead, so that each weight is
            # re-split by its own launch inside this capture (correct, ~180 launches more per replay)
            for k, v in list(_g._SPLIT_CACHE.items()):
                _g._SPLIT_CACHE[k] = (None,) + tuple(v[1:])
    a = _ARENA
    if a["buf"] is None:
        if device is None:
            return
        a["buf"] = torch.zeros(_ARENA_DOUBLES, dtype=torch.float64, device=device)
    elif a["high"] > 0:
        a["buf"][:a["high"]].zero_()
    a["pos"], a["armed"] = 0, True


def _take_zeroed(n, device):
    a = _ARENA
    if a["armed"] and a["buf"] is not None and a["buf"].device == device and a["pos"] + n <= _ARENA_DOUBLES:
        t = a["buf"][a["pos"]:a["pos"] + n]
        a["pos"] += n
        a["high"] = max(a["high"], a["pos"])
        return t
    return torch.zeros(n, dtype=torch.float64, device=device)


# A call that leaves this package's kernels for an ATen / library routine (a grouped convolution, a resize that is not the exact 2x
# bilinear, autograd through the inference post-processing, a non-fp32 nn.Linear input) goes through ONE door:
#   * counted per site in FALLBACKS;
#   * a RuntimeWarning -- tests/conftest.py turns warnings from this package into errors, so a parity test cannot silently
#     validate ATen instead of the HIP kernels -- unless the caller has declared the site inside `with ops.allowed_fallbacks(...)`;
#   * a RuntimeError under cfg.STRICT (S2F_STRICT=1: bench.py, smoke() and the tiny- and full-size parity tests), allowed or not.
# Since round 6 no GEMM-shaped product is behind this door: shapes the matrix-core kernels do not take run on ops.bmm_small
# (csrc/bmm.hip), not on a vendor library.
FALLBACKS = {}
_ALLOWED = []          # stack of sets of site names (allowed_fallbacks contexts)


class allowed_fallbacks:
    """with ops.allowed_fallbacks("upsample_bilinear"): ...   -- the named sites may leave the package's kernels without a warning
    inside the block (they are still counted in ops.FALLBACKS, and still an error under ops.STRICT)."""

    def __init__(self, *sites):
        self.sites = frozenset(sites)

    def __enter__(self):
        _ALLOWED.append(self.sites)
        return self

    def __exit__(self, *exc):
        _ALLOWED.pop()
        return False


def fallback(site, detail=""):
    FALLBACKS[site] = FALLBACKS.get(site, 0) + 1
    what = (f"spike2former_amd: {site} left the package's kernels for a library / ATen path"
            f"{' (' + detail + ')' if detail else ''}")
    if cfg.STRICT:
        raise RuntimeError(what + "; S2F_STRICT forbids that")
    if not any(site in a for a in _ALLOWED):
        import warnings
        warnings.warn(what, RuntimeWarning, stacklevel=2)          # attributed to the op (module spike2former_amd.ops.*: the filter of tests/conftest.py)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _aligned16(t):
    """contiguous and 16-byte aligned (a row slice of a tensor whose rows are no whole 16-byte groups is neither a copy nor
    aligned): what the 16-byte loads of the streaming kernels need"""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts, spikes=None):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("spike2former_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"spike2former_amd ops compute in fp32; got {t.dtype}")
    if spikes is not None:
        if not spikes.is_cuda:
            raise RuntimeError("spike2former_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
        if spikes.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"a spike operand is fp32 or bf16; got {spikes.dtype}")


# Gradient sinks: {parameter data_ptr: fp32 view of a pre-zeroed flat gradient buffer}.  When a weight has a sink, the
# split-K weight-gradient kernels add straight into it (no clearing memset per launch, no copy when the gradients are
# packed) and autograd gets None for that weight.  Installed by dist.FlatGradAllReduce.install_sinks(); None = off.


# Weight gradients that go to a sink are off the critical path of the backward pass (nothing downstream reads them until
# the gradients are packed): with a side stream set they are launched there, so the short weight-gradient GEMMs of the
# 32x32 stages overlap the data-gradient chain instead of queueing in it.  wgrad_join() must precede any read of the sinks.
# Measured at C2 inside the replayed hipGraph: 69.4 ms/step with the side stream vs 67.2 without -- OFF by default.
_WGRAD_KEEP = []          # inputs of in-flight side-stream launches (kept alive until the join)


def _wgrad_stream(sink, *inputs):
    """-> (stream handle to launch on, context manager) for a weight-gradient launch into `sink`."""
    if cfg.WGRAD_STREAM is None or sink is None:
        return None
    cfg.WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
    _WGRAD_KEEP.append(inputs)
    return cfg.WGRAD_STREAM


def wgrad_join():
    """Every weight gradient of the backward pass is in its sink after this: joins the side stream (cfg.WGRAD_STREAM) and
    launches the deferred ones (cfg.DEFER_DW)."""
    wgrad_flush()
    if cfg.WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(cfg.WGRAD_STREAM)
        _WGRAD_KEEP.clear()


# Deferred weight gradients.  Nothing downstream of a weight gradient runs until the gradients are packed, and the short-
# contraction layers (32x32 / 64x64 stages, the decoder's 100-token layers) each owe one of only 1-4 GFLOP: launched one by
# one (~180 per step) they cost 18-35 us apiece because each must split its B*L = 8 192 .. 32 768 contraction 32-64 ways to
# fill the chip.  With gradient sinks installed they are collected instead -- (dY, X, sink) kept alive -- and launched
# together by wgrad_flush() as ONE grouped kernel per contraction-step class (s2f_spike_gemm_dw_grouped).
_DW_PENDING = {64: [], 32: []}


def _defer_dw(gy, x, sink, B, M, K, L):
    _DW_PENDING[64 if (L % 64 == 0 or L >= 512) else 32].append((gy, x, sink, B, M, K, L))


_DWG_PENDING = []          # general (fp32 x fp32) weight gradients: (dY ptr, dy batch stride, X ptr, x batch stride, sink, B, M, K, L, keep)


def _defer_dw_general(gy, gy_off, dy_bs, x, x_off, x_bs, sink, B, M, K, L):
    """-> True if the weight gradient  sink += sum_b dY[b] X[b]^T  was queued for the grouped launch of wgrad_flush()."""
    if not (cfg.DEFER_DW and sink is not None and B * L <= cfg.DEFER_DW_MAX_CONTRACTION and cfg.WGRAD_STREAM is None and L % 4 == 0):
        return False
    _DWG_PENDING.append((gy.data_ptr() + 4 * gy_off, dy_bs, x.data_ptr() + 4 * x_off, x_bs, sink, B, M, K, L, (gy, x)))
    return True


def wgrad_drop():
    """Forget the deferred weight gradients that were never flushed (a backward that raised, an abandoned step): their sink
    views and activations must not be added into the NEXT step's freshly zeroed buffer."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return                                 # inside a capture the step's own flush follows; nothing stale can be pending
    for jobs in _DW_PENDING.values():
        jobs.clear()
    _DWG_PENDING.clear()


def wgrad_flush():
    import ctypes
    while _DWG_PENDING:
        chunk = _DWG_PENDING[:56]
        flat = []
        for dy, dy_bs, x, x_bs, sink, B, M, K, L, _keep in chunk:
            flat += [dy, dy_bs, x, x_bs, sink.data_ptr(), B, M, K, L]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        _time_next("spike_gemm_dw", sum(4 * j[5] * j[8] * (j[6] + j[7]) for j in chunk),
                   sum(2 * j[5] * j[6] * j[7] * j[8] for j in chunk))
        check(lib.s2f_gemm_dw_general_grouped(arr, len(chunk), _stream()), "s2f_gemm_dw_general_grouped")
        del _DWG_PENDING[:56]
    if cfg.DW_PIPE:
        # the jobs the pipelined kernel takes (L % 4 == 0, L >= 32) leave their step classes for ONE list: a grouped launch spreads equal
        # shares of all its jobs over the CUs, so the more it holds the better
        pipe = []
        for bkv, jobs in _DW_PENDING.items():
            rest = []
            for j in jobs:
                # the pipelined kernel copies 16-byte pieces of both operands by LDS-DMA (S2F_EALIGN otherwise) and its symmetric
                # schedule takes whole 32-element steps only: ONE job it rejects would fail the whole grouped launch in the middle of
                # a backward pass, so anything else stays on the round-2 grouped kernel, which takes 8-byte-aligned operands
                ok = (lib.s2f_spike_gemm_dw_pipe_ok(j[3], j[4], j[5], j[6]) and j[0].data_ptr() % 16 == 0 and j[1].data_ptr() % 16 == 0
                      and (j[6] % 32 == 0 or cfg.DWP_SCHEDULE == 0))
                (pipe if ok else rest).append(j)
            jobs[:] = rest
        while pipe:
            chunk, pipe = pipe[:56], pipe[56:]
            flat = []
            for gy, x, sink, B, M, K, L in chunk:
                flat += [gy.data_ptr(), x.data_ptr(), sink.data_ptr(), B, M, K, L]
            arr = (ctypes.c_int64 * len(flat))(*flat)
            _time_next("spike_gemm_dw", sum(4 * B * L * (K + M) for _, _, _, B, M, K, L in chunk),
                       sum(2 * B * M * L * K for _, _, _, B, M, K, L in chunk),
                       moved=sum(B * L * (2 * K + 4 * M) for _, _, _, B, M, K, L in chunk))
            check(lib.s2f_spike_gemm_dw_pipe_grouped(arr, len(chunk), cfg.DWP_SCHEDULE, cfg.DWP_WGS, _stream()), "s2f_spike_gemm_dw_pipe_grouped")
    for bkv, jobs in _DW_PENDING.items():
        while jobs:
            chunk, rest = jobs[:56], jobs[56:]
            flat = []
            for gy, x, sink, B, M, K, L in chunk:
                flat += [gy.data_ptr(), x.data_ptr(), sink.data_ptr(), B, M, K, L]
            arr = (ctypes.c_int64 * len(flat))(*flat)
            _time_next("spike_gemm_dw", sum(4 * B * L * (K + M) for _, _, _, B, M, K, L in chunk),
                       sum(2 * B * M * L * K for _, _, _, B, M, K, L in chunk),
                       moved=sum(B * L * (2 * K + 4 * M) for _, _, _, B, M, K, L in chunk))
            check(lib.s2f_spike_gemm_dw_grouped(arr, len(chunk), bkv, _stream()), "s2f_spike_gemm_dw_grouped")
            jobs[:] = rest


# ---- concurrency inside one step -------------------------------------------------------------------------------------
# Most launches of a step are kernels of 128-256 workgroups (the 32x32-stage blocks, the decoder's 100-query chain): one
# wave per SIMD, nothing to hide latency with, while other parts of the step (the 256x256-level FPN stage and mask_feature
# convolution, the decoder's key / value projections over up to 16 384 tokens) are independent of them.  With side streams
# set, independent work is launched concurrently; inside the captured hipGraph this becomes parallel branches.  Autograd
# replays each branch's backward on the stream its forward ran on and orders the hand-overs itself.
#   BRANCH_STREAMS : short fork / join around sibling chains (q / k / v projections; DCN input / offset / mask chains)
#   LONG_STREAMS   : [0] lateral convolutions + the 256x256 FPN level + mask_feature,  [1] decoder key / value projections
# A tensor allocated on one stream and consumed on another is registered with the allocator (`record_stream`), so its
# block is not handed out again while the consumer may still be reading it.  None = everything on the current stream.


def _tensors(obj):
    if isinstance(obj, Spikes):
        yield obj.data
    elif torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)


def _seen_by(obj, stream):
    for t in _tensors(obj):
        if t.is_cuda and t.numel():
            t.record_stream(stream)


def use_here(*ts):
    """The tensors are about to be read on the current stream although another one may have allocated them."""
    if cfg.BRANCH_STREAMS or cfg.LONG_STREAMS:
        _seen_by(ts, torch.cuda.current_stream())


_RR = [0]          # round-robin cursor over BRANCH_STREAMS: consecutive forks land on different side streams


def branches(fns, inputs=()):
    """Run the independent callables `fns` (first one on the current stream, the others on ops.BRANCH_STREAMS) -> results.
    `inputs`: tensors of the current stream that the branches read.  Forks only from the step's own stream: a fork out of
    a side stream (nested branches) makes hipStreamEndCapture crash on ROCm 7.2 -- such calls run inline; fns[0] runs on
    the current stream AFTER the side branches are launched, so it may itself fork again."""
    side = cfg.BRANCH_STREAMS
    if not side or len(fns) < 2 or cfg.KERNEL_EVENTS is not None:
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    if main in side or (cfg.LONG_STREAMS and main in cfg.LONG_STREAMS):
        return [f() for f in fns]
    where = [main]
    for _ in range(1, len(fns)):
        where.append(side[_RR[0] % len(side)])
        _RR[0] += 1
    for st in set(where[1:]):
        st.wait_stream(main)                   # fork: everything enqueued so far (the common input) precedes the branch
        _seen_by(inputs, st)
    out = [None] * len(fns)
    for i in range(1, len(fns)):
        with torch.cuda.stream(where[i]):
            out[i] = fns[i]()
    out[0] = fns[0]()
    for st in set(where[1:]):
        main.wait_stream(st)                   # join
    for i in range(1, len(fns)):
        _seen_by(out[i], main)
    return out




def fork(which, fn, inputs=(), what=None):
    """Launch fn() on cfg.LONG_STREAMS[which] behind everything enqueued so far -> (result, handle); `join(handle, result)`
    before the current stream reads the result.  Without side streams: runs inline, handle None."""
    if not cfg.LONG_STREAMS or cfg.KERNEL_EVENTS is not None or (what is not None and what not in cfg.LONG_WHAT):
        return fn(), None
    st, main = cfg.LONG_STREAMS[which % len(cfg.LONG_STREAMS)], torch.cuda.current_stream()
    if st == main:
        return fn(), None
    st.wait_stream(main)
    _seen_by(inputs, st)
    with torch.cuda.stream(st):
        out = fn()
        done = torch.cuda.Event()
        done.record(st)
    return out, (done, st)


def join(handle, result=()):
    if handle is not None:
        main = torch.cuda.current_stream()
        if main != handle[1]:
            main.wait_event(handle[0])
            _seen_by(result, main)


def _sink_for(w):
    """The flat-buffer slice that collects this weight's gradient, if its parameter registered one.  `w` is the parameter or
    a view of it that starts at its first element (weight.view(M, -1))."""
    if cfg.GRAD_SINKS is None:
        return None
    hit = cfg.GRAD_SINKS.get(w.data_ptr())
    if hit is None:
        return None
    ref, v = hit
    p = ref()
    if p is None or p.data_ptr() != w.data_ptr() or v.numel() != w.numel():
        return None          # the address belonged to a parameter that has since been freed or re-allocated
    return v


STAT_SLOTS = 256         # include/s2f.h S2F_STAT_SLOTS


def new_stats(device, T=None):
    """Zeroed firing counters for one neuron (s2f.h `stats`): int64 [STAT_SLOTS, 2] (or [T, STAT_SLOTS, 2] for lif_seq)."""
    shape = (STAT_SLOTS, 2) if T is None else (T, STAT_SLOTS, 2)
    return torch.zeros(shape, dtype=torch.int64, device=device)


def read_stats(stats):
    """-> int64 [..., 2] = {sum of spike counts, number of non-zero counts}, summed over the contention slots."""
    return stats.sum(-2)


def mask_words(n):
    return ((n + 255) >> 8) * 4


# ------------------------------------------------------------------------------------------------ spike maps in bf16
# A Q_IFNode output holds multiples of 1/D with at most 8 significant bits: bf16 represents it exactly.  With SPIKES_BF16 the
# neuron kernels write their spikes as bf16 (2 B / element instead of the reference's 4) and every consumer on the path -- the
# spike GEMMs, the implicit 3x3 convolutions, the attention core, the depthwise stencils -- reads that tensor directly: the
# GEMM loops lose their fp32 -> bf16 conversion and every pass over a spike map moves half the bytes.
# Autograd casts a gradient to the dtype of the tensor it belongs to, so a bf16 tensor cannot carry an fp32 gradient across
# an autograd edge.  A spike map therefore travels as a pair: `data` (bf16, not differentiable) and `tok`, its autograd
# handle -- an fp32 tensor of the same SHAPE whose strides are all zero (4 bytes of storage, never read).  The producer
# returns `tok` as a differentiable output; a consumer takes (data, tok) and hands the fp32 gradient of the spike map back
# as the gradient of `tok`.  View-type reshapes act on both halves (any view of an all-zero-stride tensor is legal and free).
# In the fp32 mode (SPIKES_BF16 = False, or spikes handed in by an outside caller) `tok` is None and `data` is an ordinary
# fp32 autograd tensor.


def spikes_bf16_ok(D):
    """bf16 storage is exact only for k / D with D a power of two (D <= 128: k <= D needs <= 8 significant bits)."""
    D = int(D)
    return cfg.SPIKES_BF16 and 0 < D <= 128 and (D & (D - 1)) == 0


class Spikes:
    """A spike map as the kernels pass it on: `data` (bf16 or fp32 values) + `tok`, the fp32 autograd handle of the producing
    neuron (None: data is its own handle).  `tok2`: a spare handle of the same neuron for a SECOND consumer (`second()`): the
    neuron's backward kernel sums the gradients arriving on both, instead of the autograd engine adding them (cfg.FANOUT_PORTS)."""
    __slots__ = ("data", "tok", "tok2")

    def __init__(self, data, tok=None, tok2=None):
        self.data, self.tok, self.tok2 = data, tok, tok2

    shape = property(lambda self: self.data.shape)
    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)
    requires_grad = property(lambda self: self.data.requires_grad if self.tok is None else self.tok.requires_grad)

    def numel(self):
        return self.data.numel()

    def dim(self):
        return self.data.dim()

    def second(self):
        """the same spikes for another consumer, on the spare handle when there is one"""
        return self if self.tok2 is None else Spikes(self.data, self.tok2, self.tok2)

    def _both(self, f):
        return Spikes(f(self.data), None if self.tok is None else f(self.tok), None if self.tok2 is None else f(self.tok2))

    def view(self, *shape):
        return self._both(lambda t: t.view(*shape))

    def reshape(self, *shape):
        return self._both(lambda t: t.reshape(*shape))

    def flatten(self, *a):
        return self._both(lambda t: t.flatten(*a))

    def permute(self, *dims):
        return self._both(lambda t: t.permute(*dims))

    def unflatten(self, dim, sizes):
        return self._both(lambda t: t.unflatten(dim, sizes))

    def contiguous(self):
        return Spikes(self.data.contiguous(), self.tok, self.tok2)

    def float(self):
        """The fp32 tensor the reference would hold at this point (one conversion pass; only for consumers off the path)."""
        if self.tok is None:
            return self.data
        return _SpikesToFloat.apply(self.data, self.tok)


def _new_tok(like):
    return torch.empty((), dtype=torch.float32, device=like.device).expand(like.shape)


def as_spikes(x):
    return x if isinstance(x, Spikes) else Spikes(x, None)


def spikes_float(x):
    return x.float() if isinstance(x, Spikes) else x


def _unpack(x):
    """-> (data, tok or None) of a spike operand handed over as Spikes or as a plain fp32 tensor"""
    return (x.data, x.tok) if isinstance(x, Spikes) else (x, None)


class _SpikesToFloat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, tok):
        return data.float()

    @staticmethod
    def backward(ctx, g):
        return None, g


def _grad_pair(ctx_has_tok, g):
    """gradient slots of a (data, tok) operand pair"""
    return (None, g) if ctx_has_tok else (g, None)




# ------------------------------------------------------------------------------------------------ parameter groups
def adjacent(ts):
    """True when the tensors lie back to back in ONE storage (same dtype / device, contiguous): their concatenation along
    dim 0 then already exists in memory."""
    t0 = ts[0]
    base = t0.untyped_storage().data_ptr()
    end = t0.data_ptr()
    for t in ts:
        if (t.dtype != t0.dtype or t.device != t0.device or not t.is_contiguous() or t.shape[1:] != t0.shape[1:]
                or t.untyped_storage().data_ptr() != base or t.data_ptr() != end):
            return False
        end += t.numel() * t.element_size()
    return True


def _alias(ts):
    t0 = ts[0]
    rows = sum(t.shape[0] for t in ts)
    return torch.empty(0, dtype=t0.dtype, device=t0.device).set_(t0.untyped_storage(), t0.storage_offset(),
                                                                 (rows,) + tuple(t0.shape[1:]))


class _AliasCat(torch.autograd.Function):
    """torch.cat(params, 0) without the copy, for parameters that are adjacent views of one flat storage (the q / k / v
    twins of an attention block, flattened by the module): forward hands out the enclosing view, backward hands each
    parameter its slice of the gradient (views, no copy)."""

    @staticmethod
    def forward(ctx, *ps):
        ctx.rows = [p.shape[0] for p in ps]
        return _alias([p.detach() for p in ps])

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


def cat_params(ps):
    """Concatenation of sibling parameters along dim 0: zero-copy when they are adjacent in memory, torch.cat otherwise."""
    ps = list(ps)
    if adjacent([p.detach() for p in ps]):
        return _AliasCat.apply(*ps) if any(p.requires_grad for p in ps) else _alias(ps)
    return torch.cat(ps, 0)


def cat_buffers(bs):
    """-> (tensor, write_back): sibling buffers as one tensor that a kernel may update in place; `write_back()` copies the
    result into the buffers when they were not adjacent (then the tensor is a temporary concatenation)."""
    bs = list(bs)
    if adjacent(bs):
        return _alias(bs), (lambda: None)
    cat = torch.cat(bs, 0)

    def write_back():
        for b, c in zip(bs, cat.split([b.shape[0] for b in bs], 0)):
            b.copy_(c)
    return cat, write_back


def flatten_together(ts):
    """Re-point the `.data` of sibling parameters / buffers at consecutive slices of one new flat tensor (values kept)."""
    ts = list(ts)
    flat = torch.cat([t.data.reshape(-1) for t in ts])
    off = 0
    for t in ts:
        t.data = flat[off:off + t.numel()].view(t.shape)
        off += t.numel()


class _Split3(torch.autograd.Function):
    """[N, 3C, L] -> three contiguous [N, C, L] tensors (one transposing copy); backward: one concatenating copy (autograd's
    own slicing would zero-fill and add three full-size tensors)."""

    @staticmethod
    def forward(ctx, y):
        N, C3, L = y.shape
        yp = y.view(N, 3, C3 // 3, L).permute(1, 0, 2, 3).contiguous()
        return yp[0], yp[1], yp[2]

    @staticmethod
    def backward(ctx, ga, gb, gc):
        return torch.stack([ga, gb, gc], 1).flatten(1, 2)


def split3(y):
    return _Split3.apply(y)




BN_PARTIALS_USED = [0, 0]          # [BatchNorm launches fed by partials, s2f_bn_stats launches] since the last reset (tests, census)


__all__ = [n for n in dir() if not n.startswith('__')]
