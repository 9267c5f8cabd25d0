// NOTE (round 3): the -DGP_* / -DNO_* knock-out switches these probes were built with lived inside the product kernels in
// round 2 (git revision cd1a9db); they were removed from spike2former_amd/csrc in round 3.  The measurements are kept in
// profiles/r02_probe_*_knockouts.txt; to repeat them, check out that revision.  Without the switches this file times the
// product kernel as it is.
// Where does dw_wgrad_kernel<5> spend its 35 us on [8,512,32,32]?  Phases knocked out by -DNO_STAGE / -DNO_MAC / -DNO_REDUCE.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I spike2former_amd/csrc -I include tools/micro/dw_wgrad_probe.hip -o /tmp/dwp && /tmp/dwp
#include "../../spike2former_amd/csrc/dwconv.hip"
#include <cstdio>
#include <vector>
__global__ void s2f_zero_kernel(float* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
void s2f_set_error(const char*, ...) {}
int main() {
  const int N = 8, C = 512, H = 32, W = 32, K = 5;
  const size_t n = (size_t)N * C * H * W;
  unsigned short* x; float *gy, *gw;
  hipMalloc(&x, n * 2); hipMalloc(&gy, n * 4); hipMalloc(&gw, C * K * K * 4);
  hipMemset(x, 0x3e, n * 2); hipMemset(gy, 0, n * 4); hipMemset(gw, 0, C * K * K * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bx = 1; bx <= 1; ++bx) {
    for (int it = 0; it < 3; ++it)
      hipLaunchKernelGGL((dw_wgrad_kernel<5, unsigned short>), dim3(1, N * C), dim3(256), 0, 0, x, nullptr, gy, gw, C, H, W, H, W, 2, 1, 1);
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it)
      hipLaunchKernelGGL((dw_wgrad_kernel<5, unsigned short>), dim3(1, N * C), dim3(256), 0, 0, x, nullptr, gy, gw, C, H, W, H, W, 2, 1, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dw_wgrad<5> [8,512,32,32]: %.1f us per launch\n", ms * 1000 / 50);
  }
  return 0;
}
