"""Input-gradient product s2f_pgemm_dx_f32: tile configurations on the shapes of one C2 step (us per launch, isolated)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spike2former_amd import ops
from spike2former_amd._lib import check, lib
S = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=9):
    fn(); torch.cuda.synchronize()
    us, out = [], ctypes.c_double()
    for _ in range(reps):
        e0, e1 = lib.s2f_event_create(), lib.s2f_event_create()
        lib.s2f_time_next_call(e0, e1)
        fn(); torch.cuda.synchronize()
        check(lib.s2f_event_elapsed_us(e0, e1, ctypes.byref(out)), "elapsed")
        us.append(out.value)
        lib.s2f_event_destroy(e0), lib.s2f_event_destroy(e1)
    us.sort()
    return us[len(us) // 2]


shapes = [(8, 256, 1024, 1024), (8, 256, 512, 1024), (8, 512, 256, 1024), (8, 1024, 256, 1024), (8, 256, 256, 1024), (8, 256, 256, 4096),
          (8, 256, 256, 16384), (8, 1440, 360, 1024), (8, 360, 360, 1024), (8, 768, 256, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, Mo, Ki, N) in shapes:
    w = torch.randn(Mo, Ki, device="cuda") * Mo ** -0.5
    g = torch.randn(B, Mo, N, device="cuda")
    dx = torch.empty(B, Ki, N, device="cuda")
    pk = ops.pack_weight(w)
    res = {}
    for cfg in (0, 3, 9, 4, 7, 2, 8):
        try:
            res[cfg] = round(timed(lambda: check(lib.s2f_pgemm_dx_f32(pk.data_ptr(), g.data_ptr(), 0, dx.data_ptr(), 0, B, Mo, Ki, N, 0.0, cfg, S), "dx")), 1)
        except Exception as e:
            res[cfg] = "-"
    print(f"dX [{Ki} <- {Mo}] x {B} x {N}:", res)
