"""torch.autograd wrappers around the C ABI.  Torch is plumbing here: it owns device memory and the stream; all
arithmetic of these ops happens in libs2f_hip.so.  Every op raises on non-CUDA input -- there is no fallback."""
import torch

from ._lib import check, lib


# When set to a list, the launches of the neuron / BatchNorm / spike-GEMM kernels are timed with the dispatch packets' own
# begin / end timestamps (include/s2f.h "measurement": s2f_time_next_call) and (name, algorithmic_bytes, flops, start,
# stop) is appended -- bench.py's live roofline measurement.  Algorithmic bytes are SURVEY section 8d's per-element
# figures: LIF forward 8 B (read x, write y), backward 12 B (read gy, read x, write gx).  Eager launches only.
KERNEL_EVENTS = None


def _time_next(name, nbytes, flops=0, moved=None):
    """nbytes: ALGORITHMIC bytes (the fp32 figures of the reference's tensors); moved: the bytes this build's kernel
    actually reads + writes for them (bf16 spike maps), reported next to the algorithmic figure; flops: ALGORITHMIC flops
    (2 M N K per GEMM -- not the 3 / 6 bf16 passes issued for them)."""
    if KERNEL_EVENTS is not None:
        e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
        if not e0 or not e1:
            raise RuntimeError("s2f_event_create failed: " + lib.s2f_last_error().decode())
        lib.s2f_time_next_call(e0, e1)
        KERNEL_EVENTS.append((name, int(nbytes), int(flops), e0, e1, int(nbytes if moved is None else moved)))


def drain_kernel_events():
    """-> [(name, algorithmic_bytes, algorithmic_flops, microseconds, moved_bytes)] of the launches timed since KERNEL_EVENTS
    was set; synchronises, frees the events and switches the timing off."""
    global KERNEL_EVENTS
    import ctypes
    torch.cuda.synchronize()
    out, us = [], ctypes.c_double()
    for name, nbytes, flops, e0, e1, moved in (KERNEL_EVENTS or []):
        check(lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(us)), "s2f_event_elapsed_us")
        out.append((name, nbytes, flops, us.value, moved))
        lib.s2f_event_destroy(e0)
        lib.s2f_event_destroy(e1)
    KERNEL_EVENTS = None
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
    _TRUST_ALL[0] = False
    _GRAD_SPLITS.clear()
    if device is not None and torch.cuda.is_current_stream_capturing():
        n = resplit_all(device, build=False)
        if n > 0:
            _TRUST_ALL[0] = True
        elif n < 0:
            # no job table for the current set of weights: forget every cached version instead, so that each weight is
            # re-split by its own launch inside this capture (correct, ~180 launches more per replay)
            for k, v in list(_SPLIT_CACHE.items()):
                _SPLIT_CACHE[k] = (None,) + tuple(v[1:])
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


def _ptr(t):
    return 0 if t is None else t.data_ptr()


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
GRAD_SINKS = None


# Weight gradients that go to a sink are off the critical path of the backward pass (nothing downstream reads them until
# the gradients are packed): with a side stream set they are launched there, so the short weight-gradient GEMMs of the
# 32x32 stages overlap the data-gradient chain instead of queueing in it.  wgrad_join() must precede any read of the sinks.
# Measured at C2 inside the replayed hipGraph: 69.4 ms/step with the side stream vs 67.2 without -- OFF by default.
WGRAD_STREAM = None
_WGRAD_KEEP = []          # inputs of in-flight side-stream launches (kept alive until the join)


def _wgrad_stream(sink, *inputs):
    """-> (stream handle to launch on, context manager) for a weight-gradient launch into `sink`."""
    if WGRAD_STREAM is None or sink is None:
        return None
    WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
    _WGRAD_KEEP.append(inputs)
    return WGRAD_STREAM


def wgrad_join():
    """Every weight gradient of the backward pass is in its sink after this: joins the side stream (WGRAD_STREAM) and
    launches the deferred ones (DEFER_DW)."""
    wgrad_flush()
    if WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)
        _WGRAD_KEEP.clear()


# Deferred weight gradients.  Nothing downstream of a weight gradient runs until the gradients are packed, and the short-
# contraction layers (32x32 / 64x64 stages, the decoder's 100-token layers) each owe one of only 1-4 GFLOP: launched one by
# one (~180 per step) they cost 18-35 us apiece because each must split its B*L = 8 192 .. 32 768 contraction 32-64 ways to
# fill the chip.  With gradient sinks installed they are collected instead -- (dY, X, sink) kept alive -- and launched
# together by wgrad_flush() as ONE grouped kernel per contraction-step class (s2f_spike_gemm_dw_grouped).
DEFER_DW = True
import os as _os
DEFER_DW_MAX_CONTRACTION = int(_os.environ.get("S2F_DEFER_DW_MAX", "32768"))
_DW_PENDING = {64: [], 32: []}


def _defer_dw(gy, x, sink, B, M, K, L):
    _DW_PENDING[64 if (L % 64 == 0 or L >= 512) else 32].append((gy, x, sink, B, M, K, L))


_DWG_PENDING = []          # general (fp32 x fp32) weight gradients: (dY ptr, dy batch stride, X ptr, x batch stride, sink, B, M, K, L, keep)


def _defer_dw_general(gy, gy_off, dy_bs, x, x_off, x_bs, sink, B, M, K, L):
    """-> True if the weight gradient  sink += sum_b dY[b] X[b]^T  was queued for the grouped launch of wgrad_flush()."""
    if not (DEFER_DW and sink is not None and B * L <= DEFER_DW_MAX_CONTRACTION and WGRAD_STREAM is None and L % 4 == 0):
        return False
    _DWG_PENDING.append((gy.data_ptr() + 4 * gy_off, dy_bs, x.data_ptr() + 4 * x_off, x_bs, sink, B, M, K, L, (gy, x)))
    return True


# ---- pre-split gradients -----------------------------------------------------------------------------------------------
# The gradient of a 1x1 convolution's output comes from a BatchNorm backward kernel and is read by two GEMMs (input and weight
# gradient) that each split it into bf16 hi / mid / lo terms in their inner loops.  When fused.conv_bn_act pairs the two ops
# (`split_request` around the convolution, `split_grad=` on bn_act), the BatchNorm backward writes the three bf16 planes instead
# (s2f_bn_act_bwd_split: 6 instead of 4 bytes per element) and hands autograd a one-element stand-in expanded to z's shape;
# _SpikeGemm.backward finds the planes under the stand-in's address and runs the all-DMA input-gradient kernel
# (s2f_pgemm_dx_split) and the copy-only weight-gradient kernels (s2f_spike_gemm_dw_*_split).  Nothing else may read that
# gradient -- z has exactly one consumer, the BatchNorm.
# Measured at C2 on one box (bench.py, S2F_GRAD_SPLIT=1 vs 0): 42.95 vs 42.84 ms per step -- the input-gradient launches get 6 %
# shorter (27.1 vs 28.9 us), the weight-gradient launches 4 % LONGER (they were not bound by the conversion arithmetic, and now
# read 6 instead of 4 bytes per element) and so do the BatchNorm backward passes that write the planes: no net gain, so the
# protocol is OFF by default and kept as a measured alternative (tests/test_gpu_kernels.py covers its kernels either way).
GRAD_SPLIT = _os.environ.get("S2F_GRAD_SPLIT", "0") != "0"
_SPLIT_STATE = {"req": False, "granted": False}
_GRAD_SPLITS = {}
_DW_PENDING_SPLIT = {64: [], 32: []}


def split_request(on):
    """fused.conv_bn_act: ask the next spike GEMM whether its backward takes the gradient as bf16 planes; -> granted?"""
    if on:
        _SPLIT_STATE["req"], _SPLIT_STATE["granted"] = True, False
        return False
    g = _SPLIT_STATE["granted"]
    _SPLIT_STATE["req"] = _SPLIT_STATE["granted"] = False
    return g


def wgrad_drop():
    """Forget the deferred weight gradients that were never flushed (a backward that raised, an abandoned step): their sink
    views and activations must not be added into the NEXT step's freshly zeroed buffer."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return                                 # inside a capture the step's own flush follows; nothing stale can be pending
    for jobs in list(_DW_PENDING.values()) + list(_DW_PENDING_SPLIT.values()):
        jobs.clear()
    _DWG_PENDING.clear()
    _GRAD_SPLITS.clear()


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
    for split, pending in ((False, _DW_PENDING), (True, _DW_PENDING_SPLIT)):
        for bkv, jobs in pending.items():
            while jobs:
                chunk, rest = jobs[:56], jobs[56:]
                flat = []
                for gy, x, sink, B, M, K, L in chunk:
                    flat += [gy.data_ptr(), x.data_ptr(), sink.data_ptr(), B, M, K, L]
                arr = (ctypes.c_int64 * len(flat))(*flat)
                _time_next("spike_gemm_dw", sum(4 * B * L * (K + M) for _, _, _, B, M, K, L in chunk),
                           sum(2 * B * M * L * K for _, _, _, B, M, K, L in chunk),
                           moved=sum(B * L * (2 * K + (6 if split else 4) * M) for _, _, _, B, M, K, L in chunk))
                fn = lib.s2f_spike_gemm_dw_grouped_split if split else lib.s2f_spike_gemm_dw_grouped
                check(fn(arr, len(chunk), bkv, _stream()), "s2f_spike_gemm_dw_grouped")
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
BRANCH_STREAMS = None
LONG_STREAMS = None


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
    if BRANCH_STREAMS or LONG_STREAMS:
        _seen_by(ts, torch.cuda.current_stream())


_RR = [0]          # round-robin cursor over BRANCH_STREAMS: consecutive forks land on different side streams


def branches(fns, inputs=()):
    """Run the independent callables `fns` (first one on the current stream, the others on ops.BRANCH_STREAMS) -> results.
    `inputs`: tensors of the current stream that the branches read.  Forks only from the step's own stream: a fork out of
    a side stream (nested branches) makes hipStreamEndCapture crash on ROCm 7.2 -- such calls run inline; fns[0] runs on
    the current stream AFTER the side branches are launched, so it may itself fork again."""
    side = BRANCH_STREAMS
    if not side or len(fns) < 2 or KERNEL_EVENTS is not None:
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    if main in side or (LONG_STREAMS and main in LONG_STREAMS):
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


LONG_WHAT = ("lat", "mf", "kv")


def fork(which, fn, inputs=(), what=None):
    """Launch fn() on LONG_STREAMS[which] behind everything enqueued so far -> (result, handle); `join(handle, result)`
    before the current stream reads the result.  Without side streams: runs inline, handle None."""
    if not LONG_STREAMS or KERNEL_EVENTS is not None or (what is not None and what not in LONG_WHAT):
        return fn(), None
    st, main = LONG_STREAMS[which % len(LONG_STREAMS)], torch.cuda.current_stream()
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
    if GRAD_SINKS is None:
        return None
    hit = GRAD_SINKS.get(w.data_ptr())
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
SPIKES_BF16 = True


def spikes_bf16_ok(D):
    """bf16 storage is exact only for k / D with D a power of two (D <= 128: k <= D needs <= 8 significant bits)."""
    D = int(D)
    return SPIKES_BF16 and 0 < D <= 128 and (D & (D - 1)) == 0


class Spikes:
    __slots__ = ("data", "tok")

    def __init__(self, data, tok=None):
        self.data, self.tok = data, tok

    shape = property(lambda self: self.data.shape)
    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)
    requires_grad = property(lambda self: self.data.requires_grad if self.tok is None else self.tok.requires_grad)

    def numel(self):
        return self.data.numel()

    def dim(self):
        return self.data.dim()

    def _both(self, f):
        return Spikes(f(self.data), None if self.tok is None else f(self.tok))

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
        return Spikes(self.data.contiguous(), self.tok)

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


# ------------------------------------------------------------------------------------------------ library GEMMs
# The fp32 GEMMs of two general operands (dX of the 1x1 / k x k convolutions, non-spike 1x1 convolutions) are plain library
# calls.  PyTorch-ROCm can route torch.bmm through rocBLAS or hipBLASLt; neither wins everywhere on gfx950 (tools/probe_blas.py,
# us: [256x1024]@[8x1024x1024] 56 vs 41, [256x512]@[8x512x1024] 35 vs 23, [256x2048]@[8x2048x100] 32 vs 20, but
# [256x256]@[8x256x1024] 12.6 vs 19.8, [360x360]@[8x360x1024] 23 vs 30), so the first call of every distinct (shape, stride)
# times both and the faster one is used from then on.  Never tunes inside a graph capture (the warm-up steps have seen
# every shape by then).
BLAS_AUTOTUNE = True
_BLAS_CHOICE = {}


def _time_bmm(a, b, backend, reps=5):
    torch.backends.cuda.preferred_blas_library(backend)
    for _ in range(2):
        torch.bmm(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.bmm(a, b)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1)


def bmm_tuned(a, b):
    if not BLAS_AUTOTUNE or not a.is_cuda:
        return torch.bmm(a, b)
    key = (tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()))
    choice = _BLAS_CHOICE.get(key)
    prev = torch.backends.cuda.preferred_blas_library()
    if choice is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.bmm(a, b)
        try:
            with torch.no_grad():
                t = {lib_: _time_bmm(a.detach(), b.detach(), lib_) for lib_ in ("cublas", "cublaslt")}
            choice = min(t, key=t.get)
        except RuntimeError:                      # a backend that cannot run this problem: stay with the default
            choice = "default"
        finally:
            torch.backends.cuda.preferred_blas_library(prev)
        _BLAS_CHOICE[key] = choice
    if choice == "default":
        return torch.bmm(a, b)
    torch.backends.cuda.preferred_blas_library(choice)
    try:
        return torch.bmm(a, b)
    finally:
        torch.backends.cuda.preferred_blas_library(prev)


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
        ctx.fast = PGEMM_DX and L % 4 == 0 and L >= PGEMM_MIN_N
        if ctx.fast:
            y = torch.empty(B, G * M, L, dtype=torch.float32, device=x.device)
            P = _want_partials(stats, B, G * M, L)
            part = torch.empty(G * M, P, 2, dtype=torch.float32, device=x.device) if P else x.new_empty(0)
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
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
        return bmm_tuned(wb, x.view(B * G, K, L)).view(B, G * M, L), part

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
            if ctx.needs_input_grad[1]:
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
            gx = bmm_tuned(wb, gyv).view(B, G * K, L)
        if any(ctx.needs_input_grad[2:]):
            gw = bmm_tuned(gyv, x.view(B * G, K, L).transpose(1, 2)).view(B, G, M, K).sum(0)
            gws = [gw[g] for g in range(G)]
        return (None, gx, *gws)


def dense_gemm(x, w, stats=False):
    """x [B, G*K, L]; w: one matrix [M, K], a stack [G, M, K], or a list of G matrices (e.g. views of G parameters: each then
    keeps its own cached pack and gradient sink) -> [B, G*M, L].  stats: as spike_gemm"""
    if torch.is_tensor(w):
        w = [w] if w.dim() == 2 else list(w.unbind(0))
    y, part = _DenseGemm.apply(bool(stats), x, *w)
    return _with_part(y, part)


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


# ------------------------------------------------------------------------------------------------ LIF
class _LIF(torch.autograd.Function):
    """One Q_IFNode call (neuron.py:166-197 + surrogate.py:522-538).  Saves 1 bit/element for backward."""

    @staticmethod
    def forward(ctx, x, v_in, D, vth, keep_v, stats, bf16):
        _need_cuda(x, v_in)
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
        if v_out is None:
            v_out = x.new_empty(0)
            ctx.mark_non_differentiable(v_out)
        if bf16:                       # (autograd handle, membrane, bf16 spikes)
            ctx.mark_non_differentiable(y)
            return _new_tok(x), v_out, y
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        return y, v_out, aux

    @staticmethod
    def backward(ctx, gy, gv, _g2):
        (mask,) = ctx.saved_tensors
        if gy is None and gv is None:
            return (None,) * 7
        if gy is None:                              # only the membrane carries a gradient
            gy = torch.zeros_like(gv)
        gy = gy.contiguous()
        if gv is not None and gv.numel() != gy.numel():
            gv = None
        if gv is not None:
            gv = gv.contiguous()
        gx = torch.empty_like(gy)
        _time_next("lif_bwd", 12 * gy.numel())
        check(lib.s2f_lif_bwd(_ptr(gy), _ptr(gv), _ptr(mask), _ptr(gx), gy.numel(), ctx.vth, ctx.D, _stream()),
              "s2f_lif_bwd")
        return gx, (gx if ctx.has_v else None), None, None, None, None, None


def lif(x, v_in=None, D=8, vth=1.0, keep_v=True, stats=None, spikes=False):
    """-> (y, v_out or None); `spikes`: y as a Spikes pair (bf16 when SPIKES_BF16 and the size allows 8-byte stores)"""
    bf16 = bool(spikes) and spikes_bf16_ok(D) and x.numel() % 4 == 0 and x.numel() > 0
    y, v, data = _LIF.apply(x, v_in, D, vth, keep_v, stats, bf16)
    if spikes:
        y = Spikes(data, y) if bf16 else Spikes(y, None)
    return y, (v if keep_v else None)


class _Sum2LIF(torch.autograd.Function):
    """The decoder's value / key neurons on  a = x + e[c]  and  a + pos[b]  in one pass, neither sum materialised
    (maskformer_head.py:535-540 + transformer.py:626-629; s2f.h s2f_sum2_lif_fwd).  Reset, stateless neurons only."""

    @staticmethod
    def forward(ctx, x, e, pos, B, D, vth, bf16):
        _need_cuda(x, e, pos)
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
        ctx.set_materialize_grads(False)
        if bf16:
            ctx.mark_non_differentiable(yk, yv)
            return _new_tok(x), _new_tok(x), yk, yv
        aux = x.new_empty(0)
        ctx.mark_non_differentiable(aux)
        return yk, yv, aux, aux

    @staticmethod
    def backward(ctx, gk, gv, _a, _b):
        mk, mv = ctx.saved_tensors
        if gk is None and gv is None:
            return (None,) * 7
        gk = torch.zeros_like(gv) if gk is None else gk
        gv = torch.zeros_like(gk) if gv is None else gv
        gk, gv = gk.contiguous(), gv.contiguous()
        gx = torch.empty_like(gk)
        _time_next("lif_bwd", 12 * gk.numel())
        check(lib.s2f_sum2_lif_bwd(_ptr(gk), _ptr(gv), _ptr(mk), _ptr(mv), _ptr(gx), gk.numel(), ctx.D, _stream()),
              "s2f_sum2_lif_bwd")
        ge = gx.sum((0, 2)) if ctx.needs_input_grad[1] else None
        return gx, ge, None, None, None, None, None


def sum2_lif(x, e, pos, B, D=8, vth=1.0):
    """x [T*B, C, L], e [C], pos [B, C, L] -> (Q_IFNode(x + e + pos), Q_IFNode(x + e))  (key spikes, value spikes), as Spikes."""
    bf16 = spikes_bf16_ok(D)
    hk, hv, dk, dv = _Sum2LIF.apply(x, e, pos, B, D, vth, bf16)
    return (Spikes(dk, hk), Spikes(dv, hv)) if bf16 else (Spikes(hk), Spikes(hv))


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


# ------------------------------------------------------------------------------------------------ BN (+bias, +residual, +LIF)
class _BNAct(torch.autograd.Function):
    """t = z + conv_bias ; u = BatchNorm(t) [+ residual] ; y = Q_IFNode(u)   in two streaming kernels forward
    (statistics, apply) and two backward (reduce, apply).  Returns (u, y, v_out); unrequested ones are empty."""

    @staticmethod
    def forward(ctx, z, conv_bias, gamma, beta, residual, v_in, running_mean, running_var, nbt, training, momentum,
                eps, lif_on, want_pre, keep_v, D, vth, stats, bf16, split_grad=False, partials=None):
        _need_cuda(z, conv_bias, gamma, beta, residual, v_in)
        ctx.split_grad = bool(split_grad)
        z = z.contiguous()
        N, C = z.shape[0], z.shape[1]
        L = z.numel() // (N * C)
        dev = z.device
        stat = torch.empty(3 * C, dtype=torch.float32, device=dev)      # mean, rstd, BN(0) border (s2f.h)
        s = _stream()
        ws = None
        single = bool(training) and bool(lib.s2f_bn_single_pass(N, C, L))    # small map: statistics inside s2f_bn_act_fwd
        if not (training and partials is not None and partials.dim() == 3 and partials.shape[0] == C and partials.shape[2] == 2
                and partials.shape[1] == lib.s2f_bn_partials_count(N, L) and (BN_PARTIALS_SINGLE or not single)):
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
        if bf16:
            ydata, y = y, _new_tok(z)
        outs = [t if t is not None else z.new_empty(0) for t in (u, y, v_out)]
        ctx.mark_non_differentiable(border, ydata, *[o for o, t in zip(outs, (u, y, v_out)) if t is None])
        return tuple(outs) + (border, ydata)

    @staticmethod
    def backward(ctx, g_u, g_y, g_v, g_border, _g_ydata):
        z, conv_bias, gamma, stat, mask = ctx.saved_tensors
        N, C, L, training, D, vth, has_res, has_bias = ctx.cfg

        def prep(g):
            return None if (g is None or g.numel() == 0) else g.contiguous()
        g_u, g_y, g_v = prep(g_u), prep(g_y), prep(g_v)
        if g_u is None and g_y is None and g_v is None:
            return (None,) * 21
        dev = z.device
        split = ctx.split_grad and z.numel() % 4 == 0
        planes = torch.empty((3, N, C, L), dtype=torch.bfloat16, device=dev) if split else None
        gz = torch.empty(1, dtype=torch.float32, device=dev).expand(z.shape) if split else torch.empty_like(z)
        g_res = torch.empty_like(z) if (has_res and ctx.needs_input_grad[4]) else None
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = None if (training and lib.s2f_bn_single_pass(N, C, L)) else _take_zeroed(2 * C, dev)
        # read z + incoming grads, write gz [, g_residual]
        alg = 4 * z.numel() * (2 + (g_u is not None) + (g_y is not None) + (g_res is not None))
        _time_next("bn_lif_bwd" if g_y is not None else "bn_bwd", alg, moved=alg + (2 * z.numel() if split else 0))
        if split:
            check(lib.s2f_bn_act_bwd_split(_ptr(z), _ptr(conv_bias), _ptr(stat), _ptr(gamma), _ptr(g_u), _ptr(g_y), _ptr(g_v),
                                           _ptr(mask), _ptr(ws), _ptr(planes), _ptr(g_res), _ptr(dgamma), _ptr(dbeta), N, C, L,
                                           int(training), vth, D, _stream()), "s2f_bn_act_bwd_split")
            _GRAD_SPLITS[gz.data_ptr()] = planes
        else:
            check(lib.s2f_bn_act_bwd(_ptr(z), _ptr(conv_bias), _ptr(stat), _ptr(gamma), _ptr(g_u), _ptr(g_y), _ptr(g_v),
                                     _ptr(mask), _ptr(ws), _ptr(gz), _ptr(g_res), _ptr(dgamma), _ptr(dbeta), N, C, L,
                                     int(training), vth, D, _stream()), "s2f_bn_act_bwd")
        g_bias = None
        if has_bias:
            # train-mode BN removes any per-channel constant: d/d(bias) == 0 exactly -> no gradient tensor at all (None, a
            # zero-fill launch per BatchNorm otherwise); eval mode: sum(gz) = gamma * rstd * dbeta
            g_bias = None if training else gamma * stat[C:2 * C] * dbeta
        if ctx.needs_input_grad[5]:
            raise RuntimeError("gradient w.r.t. the incoming membrane is not supported by the fused BN+LIF op")
        return (gz, g_bias, dgamma, dbeta, g_res) + (None,) * 16


def bn_act(z, conv_bias, gamma, beta, running_mean, running_var, nbt, training, momentum, eps, residual=None,
           lif=False, want_pre=True, v_in=None, keep_v=False, D=8, vth=1.0, stats=None, want_border=False, split_grad=False,
           partials=None):
    """-> (u or None, y or None, v_out or None [, border]); y is a Spikes pair (bf16 when SPIKES_BF16); border [C] = BN(0)
    from the updated running statistics (BNAndPadLayer's padding value), produced by the same kernel.
    partials: the [C, P, 2] per-tile sums the producing GEMM stored for z (BN_PARTIALS; default: z's `_s2f_part` attribute)."""
    if partials is None:
        partials = getattr(z, "_s2f_part", None)
    bf16 = bool(lif) and spikes_bf16_ok(D) and z.numel() % 4 == 0          # as ops.lif: consumers read bf16 spikes in 8-byte groups
    u, y, v, border, ydata = _BNAct.apply(z, conv_bias, gamma, beta, residual, v_in, running_mean, running_var, nbt, training,
                                          momentum, eps, lif, want_pre, keep_v, D, vth, stats, bf16, split_grad, partials)
    if lif:
        y = Spikes(ydata, y) if bf16 else Spikes(y, None)
    out = (u if want_pre else None), (y if lif else None), (v if (lif and keep_v) else None)
    return out + (border,) if want_border else out


BN2_FUSED = _os.environ.get("S2F_BN2_FUSED", "1") != "0"          # train-mode BatchNorm o BatchNorm pairs as one kernel (s2f_bn2_act_fwd / _bwd)


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
    return bool(BN2_FUSED and z.is_cuda and z.numel() and lib.s2f_bn2_fused_ok(N, C, z.numel() // (N * C)))


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


def dwconv(x, w, pad, border=None):
    """x: fp32 tensor or Spikes"""
    data, tok = _unpack(x)
    return _DWConv.apply(data, tok, w, border, pad)


# ------------------------------------------------------------------------------------------------ spike GEMM (bf16 MFMA)
_SPLIT_CACHE = {}
SPIKE_GEMM_TERMS = 3          # number of bf16 weight terms (3 == fp32-equivalent)
SPIKE_GEMM_ENABLED = True
# 3x3 / stride 1 / pad 1 spike convolutions as implicit GEMMs (no im2col matrix; s2f_spike_conv3x3_fwd / _dw).  Round 1 measured
# the pair (forward + weight gradient) as a win on the >= 128x128 maps only (tools/probe_conv3.py: 64x64 maps 485 vs 420 and
# 795 vs 660 us against the saved column matrix).  With the loaders' prefetches freed of their predicates (round 2, conv3_fix)
# the implicit form wins from 32x32 up -- same-box A/B of the step: threshold 128x128 43.92, 64x64 43.39, 32x32 43.44 ms -- and
# the bf16 column matrices of the 64x64 / 32x32 stages (ATen im2col) are gone.
CONV3X3_IMPLICIT = True
CONV3X3_IMPLICIT_MIN_PIXELS = int(_os.environ.get("S2F_CONV3_MIN_PIXELS", 32 * 32))
# input gradient of the 3x3 convolutions as an implicit transposed convolution on the 6-pass split GEMM (no unfold / col2im)
CONV3X3_DX_IMPLICIT = True
CONV3X3_DX_MIN_PIXELS = 0          # measured at C2: a win on every map size (61.7 vs 62.3 ms/step)
MASK_EINSUM_DW_GROUPED = _os.environ.get("S2F_MASK_DW_GROUPED", "1") != "0"   # the mask contraction's T*B dE products as one grouped launch
MASK_EINSUM_DE_MFMA = True    # dE of the mask einsum on the matrix cores (6-pass split GEMM) instead of rocBLAS fp32
SPIKE_GEMM_DW = True          # weight gradient on the bf16 matrix cores as well (dY split hi+mid+lo in-kernel)
SPIKE_GEMM_CHECK = False      # debug: assert that the activation really is a spike tensor
# Round 3: the LDS-DMA pipelined kernels (csrc/pgemm.hip).  PGEMM: forward spike GEMMs on s2f_pgemm_nn_bf16 (packed weight, bf16
# spikes, any N % 4 == 0 since the register-staged form); PGEMM_DX: every fp32 x fp32 product that ran on the library in rounds 1-2 -- the input
# gradients of the 1x1 convolutions and the forward products of the convolutions whose input is not a spike map -- on
# s2f_pgemm_dx_f32 (6 bf16 passes = fp32 accuracy), their weight gradients on s2f_gemm_dw_general.
PGEMM = _os.environ.get("S2F_PGEMM", "1") != "0"
PGEMM_DX = _os.environ.get("S2F_PGEMM_DX", "1") != "0"
PGEMM_MIN_N = 128
PGEMM_CONV = _os.environ.get("S2F_PGEMM_CONV", "1") != "0"          # implicit 3x3 convolutions on the pipelined kernels (pgemm.hip)
# Round 4: BatchNorm statistics from the producing GEMM's epilogue, without atomics (s2f.h "BatchNorm statistics from the producing
# GEMM's epilogue").  A forward convolution of a module in training mode stores per-(tile, row) partial sums next to its output
# and hands them over as the attribute `_s2f_part` of the tensor it returns (`carry_stats` moves it across a view); fused.bn_act
# passes them to the BatchNorm apply kernel instead of launching s2f_bn_stats.  A tensor that lost the attribute on the way simply
# takes the statistics pass.  BN_PARTIALS_SINGLE: also for the small maps whose BatchNorm computes its statistics itself in one
# pass (then the row-walking apply kernel runs instead of the single-pass kernel).
BN_PARTIALS = _os.environ.get("S2F_BN_PARTIALS", "1") != "0"
BN_PARTIALS_SINGLE = _os.environ.get("S2F_BN_PARTIALS_SINGLE", "0") != "0"
BN_PARTIALS_USED = [0, 0]          # [BatchNorm launches fed by partials, s2f_bn_stats launches] since the last reset (tests, census)


def _want_partials(stats, B, M, L):
    """-> number of partials per channel (> 0) if the product [B, M, L] should store BatchNorm partials, else 0"""
    if not (stats and BN_PARTIALS and L % 4 == 0):
        return 0
    if not BN_PARTIALS_SINGLE and lib.s2f_bn_single_pass(B, M, L):
        return 0
    return int(lib.s2f_bn_partials_count(B, L))


def carry_stats(src, dst):
    """dst is a view / reshape of the GEMM output src: the BatchNorm partials stored with src describe dst as well"""
    part = getattr(src, "_s2f_part", None)
    if part is not None:
        dst._s2f_part = part
    return dst


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
        if SPIKE_GEMM_CHECK:
            assert _is_spike_grid(x), "not a spike tensor"
        y = torch.empty(B, M, N, dtype=torch.float32, device=x.device)
        xb = x.dtype == torch.bfloat16
        _time_next("spike_gemm_fwd", 4 * B * N * (K + M), 2 * B * M * N * K, moved=B * N * ((2 if xb else 4) * K + 4 * M))
        pg = PGEMM and xb and N % 4 == 0 and N >= 8 and SPIKE_GEMM_TERMS == 3
        P = _want_partials(stats and pg and bias is None, B, M, N)
        part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device) if P else None
        if P:
            check(lib.s2f_pgemm_nn_bf16_stats(_ptr(pack_weight(w2d)), _ptr(x), _ptr(y), _ptr(part), B, M, N, K, _stream()),
                  "s2f_pgemm_nn_bf16_stats")
        elif pg:
            check(lib.s2f_pgemm_nn_bf16(_ptr(pack_weight(w2d)), _ptr(x), _ptr(bias), _ptr(y), B, M, N, K, SPIKE_GEMM_TERMS, 0,
                                        _stream()), "s2f_pgemm_nn_bf16")
        else:
            ws = split_weight(w2d)
            fn = lib.s2f_spike_gemm_fwd_bf16 if xb else lib.s2f_spike_gemm_fwd
            check(fn(_ptr(ws), _ptr(x), _ptr(bias), _ptr(y), B, M, N, K, ws.shape[1], ws.shape[2], SPIKE_GEMM_TERMS, _stream()),
                  "s2f_spike_gemm_fwd")
        ctx.save_for_backward(x, w2d)
        ctx.has_bias, ctx.has_tok = bias is not None, tok is not None
        # pre-split gradient protocol (see GRAD_SPLIT): granted when both gradient GEMMs of this layer can read planes
        ctx.takes_split = bool(GRAD_SPLIT and _SPLIT_STATE["req"] and PGEMM and PGEMM_DX and xb and bias is None and N % 8 == 0
                               and N >= PGEMM_MIN_N and SPIKE_GEMM_DW and M >= 16 and x.data_ptr() % 8 == 0
                               and any(ctx.needs_input_grad[:3]))
        if ctx.takes_split:
            _SPLIT_STATE["granted"] = True
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
        planes = _GRAD_SPLITS.pop(gy.data_ptr(), None) if ctx.takes_split else None
        if planes is not None:
            return _SpikeGemm._backward_split(ctx, x, w2d, planes) + (None,)
        gy = gy.contiguous()
        B = x.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx = dx_gemm(w2d, gy)
        if ctx.needs_input_grad[2]:
            M, K = w2d.shape
            L = x.shape[2]
            if SPIKE_GEMM_DW and L % 4 == 0 and M >= 16:     # 32- / 64- / 128-row tiles by M
                sink = _sink_for(w2d)
                gw = torch.empty(M, K, dtype=torch.float32, device=x.device) if sink is None else None
                xb = x.dtype == torch.bfloat16
                if (DEFER_DW and sink is not None and xb and B * L <= DEFER_DW_MAX_CONTRACTION and WGRAD_STREAM is None
                        and x.data_ptr() % 8 == 0):
                    _defer_dw(gy, x, sink, B, M, K, L)
                    return _grad_pair(ctx.has_tok, gx) + (None, gy.sum((0, 2)) if (ctx.has_bias and ctx.needs_input_grad[3]) else None,
                                                          None)
                _time_next("spike_gemm_dw", 4 * B * L * (K + M), 2 * B * M * L * K, moved=B * L * ((2 if xb else 4) * K + 4 * M))
                side = _wgrad_stream(sink, gy, x)
                st = side.cuda_stream if side is not None else _stream()
                if xb:
                    check(lib.s2f_spike_gemm_dw_bf16(_ptr(gy), _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                     int(sink is not None), st), "s2f_spike_gemm_dw_bf16")
                else:
                    check(lib.s2f_spike_gemm_dw(_ptr(gy), _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                int(sink is not None), 1, st), "s2f_spike_gemm_dw")
            else:
                gw = torch.bmm(gy, x.float().transpose(1, 2)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            gb = gy.sum((0, 2))
        return _grad_pair(ctx.has_tok, gx) + (gw, gb, None)


def _spike_gemm_backward_split(ctx, x, w2d, planes):
    """_SpikeGemm.backward from the gradient as bf16 planes [3, B, M, L] (hi | mid | lo)."""
    _, B, M, L = planes.shape
    K = w2d.shape[1]
    gx = gw = None
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
        gx = torch.empty(B, K, L, dtype=torch.float32, device=planes.device)
        _time_next("dx_gemm", 4 * B * L * (K + M), 2 * B * M * L * K, moved=B * L * (4 * K + 6 * M))
        check(lib.s2f_pgemm_dx_split(_ptr(pack_weight(w2d)), _ptr(planes), B * M * L, _ptr(gx), B, M, K, L, 0, _stream()),
              "s2f_pgemm_dx_split")
    if ctx.needs_input_grad[2]:
        sink = _sink_for(w2d)
        if DEFER_DW and sink is not None and B * L <= DEFER_DW_MAX_CONTRACTION and WGRAD_STREAM is None:
            _DW_PENDING_SPLIT[64 if (L % 64 == 0 or L >= 512) else 32].append((planes, x, sink, B, M, K, L))
        else:
            gw = torch.empty(M, K, dtype=torch.float32, device=x.device) if sink is None else None
            _time_next("spike_gemm_dw", 4 * B * L * (K + M), 2 * B * M * L * K, moved=B * L * (2 * K + 6 * M))
            side = _wgrad_stream(sink, planes, x)
            st = side.cuda_stream if side is not None else _stream()
            check(lib.s2f_spike_gemm_dw_bf16_split(_ptr(planes), B * M * L, _ptr(x), _ptr(gw if sink is None else sink), B, M, K, L,
                                                   int(sink is not None), st), "s2f_spike_gemm_dw_bf16_split")
    return _grad_pair(ctx.has_tok, gx) + (gw, None)


_SpikeGemm._backward_split = staticmethod(_spike_gemm_backward_split)


def dx_gemm(w2d, gy):
    """Input gradient of a 1x1 convolution: gx[b] = W^T @ gy[b]  (two general fp32 operands): the transposed product on the
    forward pack of W with gy split hi + mid + lo in the kernel (s2f_pgemm_dx_f32, 6 passes)."""
    B, M, N = gy.shape
    if PGEMM_DX and N % 4 == 0 and gy.is_cuda:
        K = w2d.shape[1]
        gx = torch.empty(B, K, N, dtype=torch.float32, device=gy.device)
        _time_next("dx_gemm", 4 * B * N * (K + M), 2 * B * M * N * K)
        check(lib.s2f_pgemm_dx_f32(_ptr(pack_weight(w2d)), _ptr(gy), 0, _ptr(gx), 0, B, M, K, N, 0.0, 0, _stream()),
              "s2f_pgemm_dx_f32")
        return gx
    if gy.shape[2] <= 128 and w2d.shape[0] <= 512 and w2d.shape[1] <= 512:
        # rocBLAS picks a 40 us kernel for the batched [256x256]^T @ [256x100] of the decoder (tools/probe_small_dx.py);
        # the same product through einsum's folding takes 12 us
        return torch.einsum("mk,bml->bkl", w2d, gy)
    return bmm_tuned(w2d.t().unsqueeze(0).expand(B, -1, -1), gy)


def gemm_bn_lif_eval_ok(x, N):
    """The eval-mode fusion takes bf16 spikes with N % 8 == 0, N >= PGEMM_MIN_N, and builds no autograd graph."""
    return (PGEMM and isinstance(x, Spikes) and x.data.dtype == torch.bfloat16 and x.data.is_cuda and N % 8 == 0
            and N >= PGEMM_MIN_N and not (torch.is_grad_enabled() and x.requires_grad))


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


def _with_part(y, part):
    if part.numel():
        y._s2f_part = part
    return y


def spike_gemm(x, w2d, bias=None, stats=False):
    """x: Spikes or an fp32 spike tensor [B, K, N].  stats: a train-mode BatchNorm follows -- store its partial statistics with
    the output (attribute `_s2f_part`, see BN_PARTIALS) when the kernel path can"""
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
    repeats bit for bit); the library for c % 4 != 0."""
    n, c = x2d.shape
    if not (LINEAR_TM and c % 4 == 0 and x2d.is_cuda and n > 0):
        return torch.matmul(x2d, w_oc.t())
    y = torch.zeros(n, w_oc.shape[0], dtype=torch.float32, device=x2d.device)
    check(lib.s2f_gemm_dw_general(_ptr(x2d.contiguous()), 0, _ptr(w_oc.contiguous()), 0, _ptr(y), 1, n, w_oc.shape[0], c, 3, _stream()),
          "s2f_gemm_dw_general")
    return y


def _mtm_tm(a2d, b2d):
    """a2d [n, o]^T @ b2d [n, c] -> [o, c]: the contraction runs over the rows of both -- the transposed packed-operand kernel with a2d
    packed on the fly (s2f_pgemm_dx_f32, contraction split over gridDim.z); the library for c % 4 != 0."""
    n, o = a2d.shape
    c = b2d.shape[1]
    if not (LINEAR_TM and c % 4 == 0 and a2d.is_cuda and n > 0):
        return torch.matmul(a2d.t(), b2d)
    ap = torch.empty(int(lib.s2f_pack_elems(n, o)), dtype=torch.int16, device=a2d.device)
    check(lib.s2f_pack_bf16x3(_ptr(a2d.contiguous()), _ptr(ap), n, o, 0, 0, _stream()), "s2f_pack_bf16x3")
    out = torch.empty(o, c, dtype=torch.float32, device=a2d.device)
    check(lib.s2f_pgemm_dx_f32(_ptr(ap), _ptr(b2d.contiguous()), 0, _ptr(out), 0, 1, n, o, c, 0.0, 0, _stream()), "s2f_pgemm_dx_f32")
    return out


LINEAR_TM = _os.environ.get("S2F_LINEAR_TM", "1") != "0"


def linear_tm(x, weight, bias=None):
    """torch.nn.functional.linear(x, weight, bias) for x [..., c] fp32 on the GPU with c % 4 == 0 (else the library)."""
    c = x.shape[-1]
    if not (LINEAR_TM and x.is_cuda and x.dtype == torch.float32 and c % 4 == 0 and x.numel() > 0):
        return torch.nn.functional.linear(x, weight, bias)
    return _LinearTM.apply(x.reshape(-1, c), weight, bias).view(*x.shape[:-1], weight.shape[0])


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
        mfma = bool(e_exact) and HW % 4 == 0 and SPIKE_GEMM_ENABLED
        if SPIKE_GEMM_CHECK and mfma:
            assert torch.equal(e, e.bfloat16().float()), "mask_einsum: E is not exact in bf16"
        if mfma:
            acat = e.permute(1, 2, 0, 3).reshape(B * Q, T * C)                  # row (b, q), column (t, c)
            a_split, Rpad, Kpad = _split_rows(acat, 128)
            out = torch.empty(B, Q, HW, dtype=torch.float32, device=e.device)
            check(lib.s2f_split_gemm(_ptr(a_split), Q * Kpad, Rpad * Kpad, 1, _ptr(mf), C * HW, C, B * C * HW, 3, _ptr(out),
                                     Q * HW, scale, B, Q, HW, T * C, (Q + 127) // 128 * 128, Kpad, _stream()),
                  "s2f_split_gemm")
        else:
            es = e * scale
            out = torch.bmm(es[0], mf[0])
            for t in range(1, T):
                torch.baddbmm(out, es[t], mf[t], out=out)
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
            if ctx.mfma and MASK_EINSUM_DE_MFMA:
                # dE[t, b] = g[b] (Q x HW) @ MF[t, b]^T: both operands contraction-contiguous fp32 -> the weight-gradient
                # kernel with both sides split hi+mid+lo (6 passes), split-K over HW with fp32 atomics
                ge = torch.zeros_like(e)
                for t in range(T):
                    for b in range(B):
                        check(lib.s2f_spike_gemm_dw(_ptr(g[b]), _ptr(mf[t, b]), _ptr(ge[t, b]), 1, Q, C, HW, 1, 3, _stream()),
                              "s2f_spike_gemm_dw")
            else:
                ge = torch.empty_like(e)
                for t in range(T):
                    torch.bmm(g, mf[t].transpose(1, 2), out=ge[t])
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
                    torch.bmm(es[t].transpose(1, 2), g, out=gmf[t])
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
        Mpad = (Q + 255) // 256 * 256 if Q > 256 else (Q + 63) // 64 * 64
        Kpad = T * C
        a_split = torch.empty(B, 3, Mpad, Kpad, dtype=torch.int16, device=dev)
        for b in range(B):
            check(lib.s2f_split_bf16x3(_ptr(acat[b]), _ptr(a_split[b]), Q, T * C, Mpad, Kpad, _stream()), "s2f_split_bf16x3")
        rowb = None
        if bias is not None:
            rowb = (e.sum(0) * bias.view(1, 1, -1)).sum(-1).contiguous()       # [B, Q]: sum_t E[t, b] bias (a reduction, no GEMV)
        out = torch.empty(B, Q, HW, dtype=torch.float32, device=dev)
        _time_next("spike_gemm_fwd", 4 * B * HW * (T * C + Q), 2 * B * Q * HW * T * C, moved=B * HW * (2 * T * C + 4 * Q))
        check(lib.s2f_spike_gemm_fwd_bf16_ex(_ptr(a_split), 3 * Mpad * Kpad, _ptr(sdata), C * HW, C, B * C * HW, _ptr(rowb),
                                             Q if rowb is not None else 0, scale, _ptr(out), B, Q, HW, T * C, Mpad, Kpad, _stream()),
              "s2f_spike_gemm_fwd_bf16_ex")
        ctx.save_for_backward(e, sdata, W, bias)
        ctx.cfg = (scale, T, B, bool(e_exact))
        return out

    @staticmethod
    def backward(ctx, g):
        e, sdata, W, bias = ctx.saved_tensors
        scale, T, B, e_exact = ctx.cfg
        Q, Co = e.shape[2], e.shape[3]
        C, HW = sdata.shape[1], sdata.shape[2]
        g = g.contiguous()
        dev = g.device
        gs = ge = gW = gb = None
        S = sdata.view(T, B, C, HW)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
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
                    torch.bmm(es[t].transpose(1, 2), g, out=G[t])
            gs = dx_gemm(W, G.view(T * B, Co, HW))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            H = torch.empty(T, B, Q, C, dtype=torch.float32, device=dev)
            xb = sdata.dtype == torch.bfloat16
            _time_next("spike_gemm_dw", 4 * T * B * HW * (C + Q), 2 * T * B * Q * HW * C, moved=T * B * HW * ((2 if xb else 4) * C + 4 * Q))
            if xb and MASK_EINSUM_DW_GROUPED and T * B <= 56 and HW % 4 == 0:
                # the T * B products H[t, b] = g[b] S[t, b]^T as ONE grouped launch (12 output tiles each over a 65 536-long
                # contraction: one by one they run at 183 TF/s) into the zeroed H
                import ctypes
                H.zero_()
                flat = []
                for t in range(T):
                    for b in range(B):
                        flat += [g[b].data_ptr(), S[t, b].data_ptr(), H[t, b].data_ptr(), 1, Q, C, HW]
                arr = (ctypes.c_int64 * len(flat))(*flat)
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
            rs = g.sum(-1) if bias is not None else None                      # [B, Q]
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


def mask_einsum_folded(e, spikes, W, bias, scale, T, B, e_exact=False):
    """e [T, B, Q, Co], spikes: bf16 Spikes [T*B, C, HW] (mask_feature_spike's output), W [Co, C], bias [Co] or None -> [B, Q, HW]"""
    assert isinstance(spikes, Spikes) and spikes.tok is not None and spikes.data.dtype == torch.bfloat16
    return _MaskEinsumFolded.apply(e, spikes.data, spikes.tok, W, bias, float(scale), int(T), int(B), bool(e_exact))


# ------------------------------------------------------------------------------------------------ transposition
class _TransposeLast2(torch.autograd.Function):
    """x [B, R, C] -> [B, C, R], contiguous (the adjoint is the same kernel the other way round)."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        B, R, C = x.shape
        y = torch.empty(B, C, R, dtype=torch.float32, device=x.device)
        check(lib.s2f_transpose_last2(_ptr(x), _ptr(y), B, R, C, _stream()), "s2f_transpose_last2")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B, C, R = gy.shape
        gx = torch.empty(B, R, C, dtype=torch.float32, device=gy.device)
        check(lib.s2f_transpose_last2(_ptr(gy), _ptr(gx), B, C, R, _stream()), "s2f_transpose_last2")
        return gx


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


def transpose_last2(x):
    """x [..., R, C] (fp32, CUDA) -> contiguous [..., C, R]: the `.permute(...).contiguous()` copies around the DCNv3 sampling
    core as one tiled kernel (s2f_transpose_last2)."""
    lead = x.shape[:-2]
    R, C = x.shape[-2:]
    if x.dtype != torch.float32 or x.numel() == 0:
        return x.transpose(-1, -2).contiguous()
    return _TransposeLast2.apply(x.reshape(-1, R, C)).view(*lead, C, R)


# ------------------------------------------------------------------------------------------------ 2x bilinear up-sampling
class _Up2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        N, C, h, w = x.shape
        y = torch.empty(N, C, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        check(lib.s2f_upsample2x_fwd(_ptr(x), _ptr(y), N * C, h, w, _stream()), "s2f_upsample2x_fwd")
        ctx.shape = (N, C, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, h, w = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(N, C, h, w, dtype=torch.float32, device=gy.device)
        check(lib.s2f_upsample2x_bwd(_ptr(gy), _ptr(gx), N * C, h, w, _stream()), "s2f_upsample2x_bwd")
        return gx


def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False); the exact-2x case runs the HIP kernel."""
    h, w = x.shape[-2:]
    if tuple(size) == (2 * h, 2 * w) and w % 2 == 0:
        return _Up2x.apply(x)
    return torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


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
        check(lib.s2f_mask_loss_seg_fwd(_ptr(pred), _ptr(seg), _ptr(row_class), _ptr(sums), B, R, h, w, alpha, gamma, _stream()),
              "s2f_mask_loss_seg_fwd")
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


# ------------------------------------------------------------------------------------------------ dense k x k convolution
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
        implicit = (spike_input and SPIKE_GEMM_ENABLED and CONV3X3_IMPLICIT and kh == 3 and kw == 3 and stride == 1
                    and padding == 1 and C % 32 == 0 and W % 4 == 0 and (W & (W - 1)) == 0
                    and H * W >= CONV3X3_IMPLICIT_MIN_PIXELS)
        if implicit:
            x = x.contiguous()
            if SPIKE_GEMM_CHECK:
                assert _is_spike_grid(x), "not a spike tensor"
            y = torch.empty(N, M, H * W, dtype=torch.float32, device=x.device)
            _time_next("spike_gemm_fwd", 4 * N * H * W * (C + M), 2 * N * M * H * W * C * 9,
                       moved=N * H * W * ((2 if xb else 4) * C + 4 * M))
            P = _want_partials(stats and PGEMM_CONV and xb and SPIKE_GEMM_TERMS == 3 and bias is None, N, M, H * W)
            if P:
                part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device)
                check(lib.s2f_pgemm_conv3x3_bf16_stats(_ptr(pack_weight_conv3(weight)), _ptr(x), _ptr(y), _ptr(part), N, M, C, H, W,
                                                       _stream()), "s2f_pgemm_conv3x3_bf16_stats")
            elif PGEMM_CONV and xb and SPIKE_GEMM_TERMS == 3:
                check(lib.s2f_pgemm_conv3x3_bf16(_ptr(pack_weight_conv3(weight)), _ptr(x), _ptr(bias), _ptr(y), N, M, C, H, W, 0,
                                                 _stream()), "s2f_pgemm_conv3x3_bf16")
            else:
                ws = split_weight_conv3(weight)
                fn = lib.s2f_spike_conv3x3_fwd_bf16 if xb else lib.s2f_spike_conv3x3_fwd
                check(fn(_ptr(ws), _ptr(x), _ptr(bias), _ptr(y), N, M, C, H, W, ws.shape[1], ws.shape[2], SPIKE_GEMM_TERMS,
                         _stream()), "s2f_spike_conv3x3_fwd")
            ctx.save_for_backward(x, weight)
            ctx.geo = (N, C, H, W, M, kh, kw, Ho, Wo, stride, padding, bias is not None, True)
            ctx.implicit = True
            part = y.new_empty(0) if part is None else part
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y.view(N, M, Ho, Wo), part
        ctx.implicit = False
        cols = torch.nn.functional.unfold(x, (kh, kw), 1, padding, stride)              # [N, C*kh*kw, Ho*Wo]
        use_mfma = spike_input and SPIKE_GEMM_ENABLED and cols.shape[2] % 4 == 0
        L = Ho * Wo
        if use_mfma:
            y = torch.empty(N, M, L, dtype=torch.float32, device=x.device)
            _time_next("spike_gemm_fwd", 4 * N * L * (cols.shape[1] + M), 2 * N * M * L * cols.shape[1],
                       moved=N * L * ((2 if xb else 4) * cols.shape[1] + 4 * M))
            pg = PGEMM and xb and L % 4 == 0 and L >= 8 and SPIKE_GEMM_TERMS == 3
            P = _want_partials(stats and pg and bias is None, N, M, L)
            if P:
                part = torch.empty(M, P, 2, dtype=torch.float32, device=x.device)
                check(lib.s2f_pgemm_nn_bf16_stats(_ptr(pack_weight(w2d)), _ptr(cols), _ptr(y), _ptr(part), N, M, L, cols.shape[1],
                                                  _stream()), "s2f_pgemm_nn_bf16_stats")
            elif pg:
                check(lib.s2f_pgemm_nn_bf16(_ptr(pack_weight(w2d)), _ptr(cols), _ptr(bias), _ptr(y), N, M, L, cols.shape[1],
                                            SPIKE_GEMM_TERMS, 0, _stream()), "s2f_pgemm_nn_bf16")
            else:
                ws = split_weight(w2d)
                fn = lib.s2f_spike_gemm_fwd_bf16 if xb else lib.s2f_spike_gemm_fwd
                check(fn(_ptr(ws), _ptr(cols), _ptr(bias), _ptr(y), N, M, L, cols.shape[1], ws.shape[1], ws.shape[2],
                         SPIKE_GEMM_TERMS, _stream()), "s2f_spike_gemm_fwd")
        else:
            if xb:
                cols = cols.float()
            if PGEMM_DX and L % 4 == 0 and L >= PGEMM_MIN_N:
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
                y = bmm_tuned(w2d.unsqueeze(0).expand(N, -1, -1), cols)
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
            if (CONV3X3_DX_IMPLICIT and kh == 3 and kw == 3 and stride == 1 and padding == 1 and M % 32 == 0 and W % 4 == 0
                    and gy.is_cuda and H * W >= CONV3X3_DX_MIN_PIXELS):
                # transposed convolution dX = flip(W)^T (*) dY as an implicit 6-pass split GEMM: no unfold(dY), no col2im
                gx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
                _time_next("dx_gemm", 4 * N * H * W * (C + M), 2 * N * M * H * W * C * 9)
                # measured (tools/probe_pgemm.py conv): the pipelined kernel wins for <= 64 output rows (narrow tiles: 442 vs 729 us
                # on [32 <- 128] at 256 x 256) and for long contractions (>= 256 channels); the round-2 kernel keeps a 5-10 % edge
                # on wide outputs over short contractions
                if PGEMM_CONV and (C <= 64 or M >= 256):
                    check(lib.s2f_pgemm_conv3x3_f32(_ptr(pack_weight_conv3(weight, transposed=True)), _ptr(gy), _ptr(gx), N, C, M, H,
                                                    W, 0, _stream()), "s2f_pgemm_conv3x3_f32")
                else:
                    wt = split_weight_tconv3(weight)
                    check(lib.s2f_conv3x3_general(_ptr(wt), _ptr(gy), _ptr(gx), N, C, M, H, W, wt.shape[1], wt.shape[2], _stream()),
                          "s2f_conv3x3_general")
            elif M < C and stride == 1 and kh == kw and Ho == H and Wo == W:
                wt = weight.flip(2, 3).permute(1, 0, 2, 3).reshape(C, M * kh * kw)        # [C, M*k*k], tiny
                gcols = torch.nn.functional.unfold(gy.view(N, M, Ho, Wo), (kh, kw), 1, kh - 1 - padding, 1)
                gx = bmm_tuned(wt.unsqueeze(0).expand(N, -1, -1), gcols).view(N, C, H, W)
            else:
                dcols = dx_gemm(w2d, gy)
                gx = torch.nn.functional.fold(dcols, (H, W), (kh, kw), 1, padding, stride)
        if ctx.needs_input_grad[2]:
            K = w2d.shape[1]
            if ctx.implicit:
                x = cols                                                  # the saved tensor is the activation itself
                gt = torch.empty(M, 3, 3, C, dtype=torch.float32, device=gy.device)        # tap-major, as the kernel contracts
                _time_next("spike_gemm_dw", 4 * N * H * W * (C + M), 2 * N * M * H * W * K,
                           moved=N * H * W * ((2 if xb else 4) * C + 4 * M))
                fn = lib.s2f_spike_conv3x3_dw_bf16 if xb else lib.s2f_spike_conv3x3_dw
                check(fn(_ptr(gy), _ptr(x), _ptr(gt), N, M, C, H, W, 0, _stream()), "s2f_spike_conv3x3_dw")
                sink = _sink_for(weight)
                if sink is not None:
                    sink.view(M, C, 3, 3).add_(gt.permute(0, 3, 1, 2))
                    gw = None
                else:
                    gw = gt.permute(0, 3, 1, 2).contiguous()
            elif use_mfma and SPIKE_GEMM_DW and M >= 16:
                sink = _sink_for(weight)
                gw = torch.empty(M, K, dtype=torch.float32, device=gy.device) if sink is None else None
                if (DEFER_DW and sink is not None and xb and N * Ho * Wo <= DEFER_DW_MAX_CONTRACTION and WGRAD_STREAM is None):
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
            elif PGEMM_DX and (Ho * Wo) % 4 == 0 and cols.dtype == torch.float32:
                # both operands general fp32 (the stem): 6-pass weight-gradient kernel, straight into the sink when there is one
                sink = _sink_for(weight)
                gw = torch.empty(M, K, dtype=torch.float32, device=gy.device) if sink is None else None
                check(lib.s2f_gemm_dw_general(_ptr(gy), 0, _ptr(cols), 0, _ptr(gw if sink is None else sink), N, M, K, Ho * Wo,
                                              int(sink is not None), _stream()), "s2f_gemm_dw_general")
            else:
                gw = torch.bmm(gy, cols.float().transpose(1, 2)).sum(0)
            gw = gw.view_as(weight) if gw is not None else None
        if has_bias and ctx.needs_input_grad[3]:
            gb = gy.sum((0, 2))
        return _grad_pair(ctx.has_tok, gx) + (gw, gb, None, None, None, None)


def conv_dense(x, weight, bias, stride, padding, spike_input, stats=False):
    """x: fp32 tensor, or Spikes (then `spike_input` is implied).  stats: as spike_gemm"""
    data, tok = _unpack(x)
    return _with_part(*_ConvDense.apply(data, tok, weight, bias, stride, padding, spike_input or isinstance(x, Spikes), bool(stats)))
