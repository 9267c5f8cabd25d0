// NOTE (round 3): the -DGP_* / -DNO_* knock-out switches these probes were built with lived inside the product kernels in
// round 2 (git revision cd1a9db); they were removed from spike2former_amd/csrc in round 3.  The measurements are kept in
// profiles/r02_probe_*_knockouts.txt; to repeat them, check out that revision.  Without the switches this file times the
// product kernel as it is.
// What bounds sgemm_bf16_kernel?  Phases knocked out by -DGP_NO_MFMA / -DGP_NO_LDSREAD / -DGP_NO_STORE / -DGP_NO_GLOBAL.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spike2former_amd/csrc -I include tools/micro/gemm_fwd_probe.hip -o /tmp/gp && /tmp/gp
#include "../../spike2former_amd/csrc/gemm_bf16.hip"
#include <cstdio>
#include <cstdlib>
__global__ void s2f_zero_kernel(float* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
void s2f_set_error(const char*, ...) {}
static S2fTiming g_tm = {nullptr, nullptr};
S2fTiming* s2f_timing_tls() { return &g_tm; }
int main() {
  struct Shape { int B, M, K, N; } shapes[] = {{8, 512, 512, 1024}, {8, 512, 1536, 1024}, {8, 1024, 256, 1024}, {8, 256, 256, 16384},
                                               {8, 512, 1152, 4096}, {8, 256, 1024, 1024}};
  for (auto sh : shapes) {
    const int Mpad = (sh.M + 255) / 256 * 256, Kpad = (sh.K + 31) / 32 * 32;
    uint16_t *w, *x; float* y;
    hipMalloc(&w, (size_t)3 * Mpad * Kpad * 2); hipMalloc(&x, (size_t)sh.B * sh.K * sh.N * 2); hipMalloc(&y, (size_t)sh.B * sh.M * sh.N * 4);
    hipMemset(w, 0x3c, (size_t)3 * Mpad * Kpad * 2); hipMemset(x, 0x3c, (size_t)sh.B * sh.K * sh.N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) s2f_spike_gemm_fwd_bf16(w, x, nullptr, y, sh.B, sh.M, sh.N, sh.K, Mpad, Kpad, 3, nullptr);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) s2f_spike_gemm_fwd_bf16(w, x, nullptr, y, sh.B, sh.M, sh.N, sh.K, Mpad, Kpad, 3, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / 20, fl = 2.0 * sh.B * sh.M * sh.K * sh.N;
    printf("[%4dx%4d]@[%dx%4dx%5d]: %7.1f us  %6.1f TF/s alg (%4.2f issued of peak)  %6.0f GB/s\n", sh.M, sh.K, sh.B, sh.K, sh.N, us,
           fl / us / 1e6, 3 * fl / us / 1e6 / 2500, ((double)sh.B * sh.N * (2.0 * sh.K + 4.0 * sh.M)) / us / 1e3);
    hipFree(w); hipFree(x); hipFree(y);
  }
  return 0;
}
