import time, torch
dev = "cuda"
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n
B, M, K, N = 8, 256, 256, 16384
w = torch.randn(M, K, device=dev); x = (torch.randint(0, 9, (B, K, N), device=dev).float()/8)
we = w.unsqueeze(0).expand(B, M, K)
t = bench(lambda: torch.bmm(we, x)); print(f"fp32 bmm {t*1e6:.1f} us  {2*B*M*K*N/t/1e12:.1f} TF")
xb = x.bfloat16(); wb = w.bfloat16().unsqueeze(0).expand(B, M, K)
t = bench(lambda: torch.bmm(wb, xb)); print(f"bf16 bmm (bf16 out) {t*1e6:.1f} us  {2*B*M*K*N/t/1e12:.1f} TF")
try:
    o = torch.bmm(wb, xb, out_dtype=torch.float32)
    t = bench(lambda: torch.bmm(wb, xb, out_dtype=torch.float32)); print(f"bf16 bmm (fp32 out) {t*1e6:.1f} us  {2*B*M*K*N/t/1e12:.1f} TF", o.dtype)
    acc = torch.zeros(B, M, N, device=dev)
    try:
        o2 = torch.baddbmm(acc, wb, xb, out_dtype=torch.float32); print("baddbmm out_dtype ok", o2.dtype)
        t = bench(lambda: torch.baddbmm(acc, wb, xb, out_dtype=torch.float32)); print(f"bf16 baddbmm (fp32 out) {t*1e6:.1f} us")
    except Exception as e: print("baddbmm out_dtype fail", repr(e)[:200])
except Exception as e:
    print("out_dtype fail", repr(e)[:300])
# big single GEMM: mask einsum shape: [700,1024] x [1024,65536]
A = torch.randn(700, 1024, device=dev); Bm = torch.randn(1024, 65536, device=dev)
t = bench(lambda: A @ Bm, 5); print(f"fp32 mm 700x1024x65536 {t*1e3:.2f} ms {2*700*1024*65536/t/1e12:.1f} TF")
Ab, Bb = A.bfloat16(), Bm.bfloat16()
t = bench(lambda: Ab @ Bb, 5); print(f"bf16 mm {t*1e3:.2f} ms {2*700*1024*65536/t/1e12:.1f} TF")
try:
    t = bench(lambda: torch.mm(Ab, Bb, out_dtype=torch.float32), 5); print(f"bf16 mm fp32-out {t*1e3:.2f} ms {2*700*1024*65536/t/1e12:.1f} TF")
except Exception as e: print("mm out_dtype fail", repr(e)[:200])
# 3x3 conv as im2col GEMM: W[128,288] x cols[8,288,65536]
w2 = torch.randn(128, 288, device=dev).unsqueeze(0).expand(8,128,288); c2 = torch.randn(8, 288, 65536, device=dev)
t = bench(lambda: torch.bmm(w2, c2), 5); print(f"fp32 bmm 128x288x65536 x8 {t*1e3:.2f} ms {2*8*128*288*65536/t/1e12:.1f} TF")
w3 = torch.randn(32, 1152, device=dev).unsqueeze(0).expand(8,32,1152); c3 = torch.randn(8, 1152, 65536, device=dev)
t = bench(lambda: torch.bmm(w3, c3), 5); print(f"fp32 bmm 32x1152x65536 x8 {t*1e3:.2f} ms {2*8*32*1152*65536/t/1e12:.1f} TF")
