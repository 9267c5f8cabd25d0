// NOTE (round 3): the -DGP_* / -DNO_* knock-out switches these probes were built with lived inside the product kernels in
// round 2 (git revision cd1a9db); they were removed from spike2former_amd/csrc in round 3.  The measurements are kept in
// profiles/r02_probe_*_knockouts.txt; to repeat them, check out that revision.  Without the switches this file times the
// product kernel as it is.
// What bounds the weight-gradient kernel?  -DGP_NO_SPLIT / -DGP_NO_MFMA / -DGP_NO_LDSREAD / -DGP_NO_STORE / -DGP_NO_GLOBAL
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
  struct Shape { int B, M, K, L; } shapes[] = {{8, 256, 256, 16384}, {8, 512, 1152, 4096}, {8, 128, 4608, 4096}, {8, 512, 512, 1024},
                                               {8, 256, 64, 65536}, {8, 1024, 256, 1024}};
  for (auto sh : shapes) {
    float *dy, *dw; uint16_t* x;
    hipMalloc(&dy, (size_t)sh.B * sh.M * sh.L * 4); hipMalloc(&x, (size_t)sh.B * sh.K * sh.L * 2); hipMalloc(&dw, (size_t)sh.M * sh.K * 4);
    hipMemset(dy, 0, (size_t)sh.B * sh.M * sh.L * 4); hipMemset(x, 0x3c, (size_t)sh.B * sh.K * sh.L * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) s2f_spike_gemm_dw_bf16(dy, x, dw, sh.B, sh.M, sh.K, sh.L, 0, nullptr);
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) s2f_spike_gemm_dw_bf16(dy, x, dw, sh.B, sh.M, sh.K, sh.L, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / 20, fl = 2.0 * sh.B * sh.M * sh.K * sh.L;
    printf("dW[%4dx%4d] over [%dx%5d]: %7.1f us  %6.1f TF/s alg (%4.2f issued of peak)  %6.0f GB/s\n", sh.M, sh.K, sh.B, sh.L, us,
           fl / us / 1e6, 3 * fl / us / 1e6 / 2500, ((double)sh.B * sh.L * (2.0 * sh.K + 4.0 * sh.M)) / us / 1e3);
    hipFree(dy); hipFree(x); hipFree(dw);
  }
  return 0;
}
