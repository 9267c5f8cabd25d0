"""BatchNorm(+neuron) streaming kernels at the step's large shapes: MOVED GB/s per call (HIP events around 20 back-to-back
launches; moved = the bytes the kernels of the call really read + write, two-pass forms read z / the gradient twice)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
from spike2former_amd._lib import lib


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


P = lambda t: None if t is None else t.data_ptr()
print(f"{'N,C,L':>18} {'1pass':>5} | {'stats':>13} {'apply(bf16 y)':>15} {'bwd gy':>13} {'bwd gu':>13} {'bwd gu+gy+res':>15}   moved GB/s (us)")
shapes = [(8, 256, 65536), (8, 512, 16384), (8, 128, 16384), (8, 1024, 4096), (8, 256, 4096), (8, 1536, 1024), (8, 384, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for N, C, L in shapes:
    n = N * C * L
    z = torch.randn(N, C, L, device="cuda") * 2 + 1
    g1 = torch.randn_like(z); g2 = torch.randn_like(z)
    y = torch.empty(N, C, L, dtype=torch.bfloat16, device="cuda")
    mask = torch.empty(ops.mask_words(n), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda")
    ws = torch.zeros(2 * C, dtype=torch.float64, device="cuda"); stat = torch.empty(3 * C, device="cuda")
    single = bool(lib.s2f_bn_single_pass(N, C, L))

    def stats():
        ws.zero_()
        lib.s2f_bn_stats(P(z), None, P(ws), N, C, L, s)
    t_stats = None if single else timeit(stats)
    if not single:
        stats()

    def apply():
        lib.s2f_bn_act_fwd(P(z), None, P(ws), P(stat), None, None, None, P(gamma), P(beta), None, None, None, P(y), None, P(mask),
                           None, N, C, L, 0.1, 1e-5, 1, 1.0, 8, 1, s)
    t_apply = timeit(apply)
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda"); gz = torch.empty_like(z); gres = torch.empty_like(z)
    ws2 = torch.zeros(2 * C, dtype=torch.float64, device="cuda")

    def bwd(gu, gy, res):
        def f():
            ws2.zero_()
            lib.s2f_bn_act_bwd(P(z), None, P(stat), P(gamma), P(gu), P(gy), None, P(mask), P(ws2), P(gz), P(gres if res else None),
                               P(dg), P(db), N, C, L, 1, 1.0, 8, s)
        return f
    t_b1 = timeit(bwd(None, g1, False)); t_b2 = timeit(bwd(g1, None, False)); t_b3 = timeit(bwd(g1, g2, True))
    passes = 1 if single else 2
    g = lambda b, t: "-" if t is None else f"{b * n / t / 1e9:6.0f} ({t * 1e6:6.1f})"
    print(f"{str((N, C, L)):>18} {int(single):>5} | {g(4, t_stats):>13} {g(6, t_apply):>15} {g(4 * (passes * 2 + 1), t_b1):>13} "
          f"{g(4 * (passes * 2 + 1), t_b2):>13} {g(4 * (passes * 3 + 2), t_b3):>15}")
