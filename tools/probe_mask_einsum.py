"""Mask-einsum backward dE[b] = g[b] [Q x HW] @ MF[b]^T [HW x C] (K = 65536): rocBLAS formulations."""
import time, torch
def graphed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): out = fn()
    gr.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (5 * reps) * 1e6
B, Q, C, HW = 2, 700, 256, 65536
g = torch.randn(B, Q, HW, device="cuda"); mf = torch.randn(B, C, HW, device="cuda"); e = torch.randn(B, Q, C, device="cuda")
out = torch.empty(B, Q, C, device="cuda")
f1 = lambda: torch.bmm(g, mf.transpose(1, 2), out=out)
f2 = lambda: torch.bmm(mf, g.transpose(1, 2)).transpose(1, 2)
def f3(S=8):
    gs = g.view(B, Q, S, HW // S).permute(0, 2, 1, 3).reshape(B * S, Q, HW // S)      # copy
    ms = mf.view(B, C, S, HW // S).permute(0, 2, 3, 1).reshape(B * S, HW // S, C)     # copy
    return torch.bmm(gs, ms).view(B, S, Q, C).sum(1)
def f4(S=16):
    # strided batch without copies: batch = (b, s) via as_strided views
    gs = g.as_strided((B, S, Q, HW // S), (Q * HW, HW // S, HW, 1)).reshape(B * S, Q, HW // S) if False else None
    return None
f5 = lambda: torch.einsum("bqn,bcn->bqc", g, mf)
ref = f1().clone()
for name, f in (("bmm(g, mf^T)", f1), ("(mf @ g^T)^T", f2), ("split-K 8 with copies", f3), ("einsum", f5)):
    t = graphed(f); err = float((f() - ref).abs().max() / ref.abs().max())
    print(f"{name:28s} {t:8.1f} us  {2*B*Q*C*HW/t/1e6:6.1f} TF/s  relerr {err:.1e}")
# forward: out = sum_t E_t @ MF_t  as one GEMM with K = T*C through a [B, T*C, HW] copy of MF vs 4 baddbmm
T = 4
E = torch.randn(T, B, Q, C, device="cuda"); MF = torch.randn(T, B, C, HW, device="cuda")
def fwd_chain():
    o = torch.bmm(E[0], MF[0])
    for t in range(1, T): torch.baddbmm(o, E[t], MF[t], out=o)
    return o
def fwd_cat():
    return torch.bmm(E.permute(1, 2, 0, 3).reshape(B, Q, T * C), MF.permute(1, 0, 2, 3).reshape(B, T * C, HW))
for name, f in (("fwd: bmm + 3 baddbmm", fwd_chain), ("fwd: K-concatenated (copies MF)", fwd_cat)):
    t = graphed(f, 3); print(f"{name:34s} {t:8.1f} us  {2*T*B*Q*C*HW/t/1e6:6.1f} TF/s")
# dMF[t] = E_t^T @ g
f6 = lambda: torch.bmm(E[0].transpose(1, 2), g)
t = graphed(f6); print(f"dMF bmm(E^T, g)                {t:8.1f} us  {2*B*Q*C*HW/t/1e6:6.1f} TF/s")
