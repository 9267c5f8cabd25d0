"""Fixed cost of the streaming kernels: kernel-timestamp duration vs size, with/without mask and firing counters."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd._lib import lib
from spike2former_amd import ops

def timed(fn, reps=20):
    e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
    us = ctypes.c_double(); best = []
    for _ in range(reps):
        lib.s2f_time_next_call(e0, e1); fn(); torch.cuda.synchronize()
        lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(us)); best.append(us.value)
    best.sort(); return best[len(best) // 2]

for n in (1 << 16, 204800, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
    x = torch.randn(n, device="cuda") * 4
    y = torch.empty_like(x); mask = torch.empty(ops.mask_words(n), dtype=torch.int64, device="cuda")
    stats = ops.new_stats("cuda")
    r = []
    for m, st in ((None, None), (mask, None), (None, stats), (mask, stats)):
        r.append(timed(lambda: lib.s2f_lif_fwd(x.data_ptr(), None, y.data_ptr(), None, m.data_ptr() if m is not None else None, None,
                                               st.data_ptr() if st is not None else None, n, 1.0, 8, None)))
    # BN apply (C=256 channels) with LIF
    C = 256; L = n // C // 8 if n >= C * 8 * 4 else 0
    t_bn = t_st = float("nan")
    if L >= 4:
        z = x[:8 * C * L].view(8, C, L); sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
        stat = torch.empty(3 * C, device="cuda"); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda"); nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
        def st_():
            sums.zero_(); 
            return lib.s2f_bn_stats(z.data_ptr(), None, sums.data_ptr(), 8, C, L, None)
        t_st = timed(st_)
        t_bn = timed(lambda: lib.s2f_bn_act_fwd(z.data_ptr(), None, sums.data_ptr(), stat.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
                                                 g.data_ptr(), b.data_ptr(), None, None, None, y.data_ptr(), None, mask.data_ptr(), stats.data_ptr(),
                                                 8, C, L, 0.1, 1e-5, 1, 1.0, 8, None))
    print(f"n={n:9d} ({8*n/1e6:7.2f} MB)  lif: plain {r[0]:6.2f}  +mask {r[1]:6.2f}  +stats {r[2]:6.2f}  +both {r[3]:6.2f} us | bn_stats {t_st:6.2f}  bn_lif {t_bn:6.2f} us")
