"""fp32 batched GEMMs of the path's dX shapes: rocBLAS vs hipBLASLt behind torch.bmm.   python tools/probe_blas.py"""
import torch, time
B = 8
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(256, 256, 1024), (1024, 256, 1024), (256, 1024, 1024), (512, 256, 1024), (768, 256, 1024), (360, 360, 1024),
          (256, 256, 16384), (256, 256, 65536), (2048, 256, 100)]
for lib_ in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib_)
    except Exception as e:
        print(lib_, "not selectable:", e); continue
    print("==", lib_, torch.backends.cuda.preferred_blas_library())
    for (M, K, L) in shapes:
        w = torch.randn(M, K, device="cuda"); gy = torch.randn(B, M, L, device="cuda")
        wt = w.t()
        t = timeit(lambda: torch.bmm(wt.unsqueeze(0).expand(B, -1, -1), gy))
        print(f"  dX [{K}x{M}] @ [{B}x{M}x{L}]: {t:8.1f} us  {2.0*B*M*K*L/t/1e6:7.1f} TF/s")
