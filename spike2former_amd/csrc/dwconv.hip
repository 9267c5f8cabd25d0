// Depthwise KxK convolution (stride 1, dilation 1) on channel-major [N, C, H, W] fp32, for gfx950.
//
// Reference call sites: SepConv.dwconv 7x7 (mmseg/models/backbones/sdtv2.py:156-163), RepConv's un-padded 3x3 applied
// to the BNAndPadLayer output (sdtv2.py:48-89, 123-127), SepConv_Spike.dwconv 3x3/5x5 (mmcv_spike/SNN_core.py:36-40),
// DCNv3_pytorch.dw_conv 5x5 (ops_dcnv3/modules/dcnv3.py:161-169), pixel-decoder output_convs 3x3
// (mmdet/models/layers/pixel_decoder.py:374-378).  ATen's generic depthwise kernels spent 20 ms per C2 step here
// (9.7 ms in the weight gradient alone) and MIOpen is unusable on this image (no gfx950 kernel database).
//
// HBM/LDS-bound stencils.  One workgroup owns a 32x32 output tile of one (n, c) plane staged in LDS with its halo; the
// K*K weights of the channel are wave-uniform scalar loads.  `border` (per-channel, nullable) is the value read outside
// the plane inside the padding ring -- BNAndPadLayer's "BN(0)" border -- so the padded tensor of the reference is never
// materialised.  The input gradient is the same stencil with the kernel flipped; the weight gradient assigns one
// (tap, pixel-slice) to each thread and finishes with K*K atomics per workgroup.
#include "s2f_common.h"

#pragma clang fp contract(fast)

namespace {

constexpr int TS = 32;   // tile side (outputs)

// FLIP = false: y[oy][ox] = sum_{i,j} w[i][j] * x[oy + i - pad][ox + j - pad]           (x: H x W, y: Ho x Wo)
// FLIP = true : y[oy][ox] = sum_{i,j} w[i][j] * x[oy - i + pad][ox - j + pad]           (input gradient: x = gy)
template <int K, bool FLIP>
__global__ __launch_bounds__(256) void dw_stencil_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ border, float* __restrict__ y, int C,
                                                         int H, int W, int Ho, int Wo, int pad, int tiles_x) {
  constexpr int HS = TS + K - 1;
  __shared__ float s[HS][HS + 1];
  const int plane = blockIdx.y;
  const int c = plane % C;
  const int ty = (blockIdx.x / tiles_x) * TS, tx = (blockIdx.x % tiles_x) * TS;
  const float* xp = x + (int64_t)plane * H * W;
  const float fillv = border ? border[c] : 0.f;
  // halo origin in input coordinates
  const int oy0 = FLIP ? ty - (K - 1) + pad : ty - pad;
  const int ox0 = FLIP ? tx - (K - 1) + pad : tx - pad;
  for (int e = threadIdx.x; e < HS * HS; e += 256) {
    const int r = e / HS, q = e % HS;
    const int iy = oy0 + r, ix = ox0 + q;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = xp[(int64_t)iy * W + ix];
    else if (!FLIP && iy >= -pad && iy < H + pad && ix >= -pad && ix < W + pad)
      v = fillv;
    s[r][q] = v;
  }
  float wk[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wk[i] = w[c * K * K + i];
  __syncthreads();
  const int r = threadIdx.x >> 3;            // 0..31
  const int q0 = (threadIdx.x & 7) * 4;      // 0,4,..,28
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K; ++i) {
    float row[K + 3];
#pragma unroll
    for (int j = 0; j < K + 3; ++j) row[j] = s[r + i][q0 + j];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float wv = FLIP ? wk[(K - 1 - i) * K + (K - 1 - j)] : wk[i * K + j];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += wv * row[j + o];
    }
  }
  const int oy = ty + r;
  if (oy < Ho) {
    float* yp = y + (int64_t)plane * Ho * Wo + (int64_t)oy * Wo + tx + q0;
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (tx + q0 + o < Wo) yp[o] = acc[o];
  }
}

// gw[c][i][j] += sum over the tile of gy[oy][ox] * x[oy + i - pad][ox + j - pad]
template <int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ border,
                                                       const float* __restrict__ gy, float* __restrict__ gw, int C, int H,
                                                       int W, int Ho, int Wo, int pad, int tiles_x) {
  constexpr int HS = TS + K - 1;
  constexpr int KK = K * K;
  constexpr int S = 256 / KK;            // pixel slices per tap
  __shared__ float s[HS][HS + 1];
  __shared__ float g[TS][TS + 1];
  __shared__ float red[KK][S + 1];
  const int plane = blockIdx.y;
  const int c = plane % C;
  const int ty = (blockIdx.x / tiles_x) * TS, tx = (blockIdx.x % tiles_x) * TS;
  const float* xp = x + (int64_t)plane * H * W;
  const float* gp = gy + (int64_t)plane * Ho * Wo;
  const float fillv = border ? border[c] : 0.f;
  for (int e = threadIdx.x; e < HS * HS; e += 256) {
    const int r = e / HS, q = e % HS;
    const int iy = ty - pad + r, ix = tx - pad + q;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = xp[(int64_t)iy * W + ix];
    else if (iy >= -pad && iy < H + pad && ix >= -pad && ix < W + pad)
      v = fillv;
    s[r][q] = v;
  }
  for (int e = threadIdx.x; e < TS * TS; e += 256) {
    const int r = e / TS, q = e % TS;
    const int oy = ty + r, ox = tx + q;
    g[r][q] = (oy < Ho && ox < Wo) ? gp[(int64_t)oy * Wo + ox] : 0.f;
  }
  __syncthreads();
  const int tap = threadIdx.x / S, sl = threadIdx.x % S;
  if (tap < KK) {
    const int i = tap / K, j = tap % K;
    float acc = 0.f;
    for (int p = sl; p < TS * TS; p += S) {
      const int r = p / TS, q = p % TS;
      acc += g[r][q] * s[r + i][q + j];
    }
    red[tap][sl] = acc;
  }
  __syncthreads();
  if (threadIdx.x < KK) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < S; ++k) t += red[threadIdx.x][k];
    atomicAdd(gw + c * KK + threadIdx.x, t);
  }
}

int check(const char* who, int N, int C, int H, int W, int K, int pad, int& Ho, int& Wo) {
  S2F_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, S2F_EINVAL, "%s: bad shape", who);
  S2F_REQUIRE(K == 3 || K == 5 || K == 7, S2F_EINVAL, "%s: kernel size %d not in {3,5,7}", who, K);
  S2F_REQUIRE(pad >= 0 && pad < K, S2F_EINVAL, "%s: bad padding %d", who, pad);
  Ho = H + 2 * pad - K + 1;
  Wo = W + 2 * pad - K + 1;
  S2F_REQUIRE(Ho > 0 && Wo > 0, S2F_EINVAL, "%s: empty output", who);
  S2F_REQUIRE((int64_t)N * C < 65536 * 16, S2F_EINVAL, "%s: too many planes", who);
  return S2F_OK;
}

template <bool FLIP>
void launch_stencil(int K, dim3 grid, hipStream_t s, const float* x, const float* w, const float* border, float* y, int C,
                    int H, int W, int Ho, int Wo, int pad, int tiles_x) {
  if (K == 3)
    hipLaunchKernelGGL((dw_stencil_kernel<3, FLIP>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x);
  else if (K == 5)
    hipLaunchKernelGGL((dw_stencil_kernel<5, FLIP>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x);
  else
    hipLaunchKernelGGL((dw_stencil_kernel<7, FLIP>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x);
}

}  // namespace

extern "C" int s2f_dwconv_fwd(const float* x, const float* w, const float* border, float* y, int N, int C, int H, int W,
                              int K, int pad, void* stream) {
  S2F_REQUIRE(x && w && y, S2F_EINVAL, "s2f_dwconv_fwd: null pointer");
  int Ho, Wo;
  int rc = check("s2f_dwconv_fwd", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  const int tiles_x = (Wo + TS - 1) / TS, tiles_y = (Ho + TS - 1) / TS;
  launch_stencil<false>(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, x, w, border, y, C, H, W, Ho, Wo, pad,
                        tiles_x);
  return s2f_check_launch("s2f_dwconv_fwd");
}

extern "C" int s2f_dwconv_bwd_input(const float* gy, const float* w, float* gx, int N, int C, int H, int W, int K, int pad,
                                    void* stream) {
  S2F_REQUIRE(gy && w && gx, S2F_EINVAL, "s2f_dwconv_bwd_input: null pointer");
  int Ho, Wo;
  int rc = check("s2f_dwconv_bwd_input", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  // the stencil's "input" is gy (Ho x Wo) and its output is gx (H x W)
  const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
  launch_stencil<true>(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, gy, w, nullptr, gx, C, Ho, Wo, H, W, pad,
                       tiles_x);
  return s2f_check_launch("s2f_dwconv_bwd_input");
}

extern "C" int s2f_dwconv_bwd_weight(const float* x, const float* border, const float* gy, float* gw, int N, int C, int H,
                                     int W, int K, int pad, int accumulate, void* stream) {
  S2F_REQUIRE(x && gy && gw, S2F_EINVAL, "s2f_dwconv_bwd_weight: null pointer");
  int Ho, Wo;
  int rc = check("s2f_dwconv_bwd_weight", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(gw, 0, sizeof(float) * (size_t)C * K * K, s) != hipSuccess)
    return s2f_check_launch("s2f_dwconv_bwd_weight memset");
  const int tiles_x = (Wo + TS - 1) / TS, tiles_y = (Ho + TS - 1) / TS;
  const dim3 grid(tiles_x * tiles_y, N * C);
  if (K == 3)
    hipLaunchKernelGGL(dw_wgrad_kernel<3>, grid, dim3(256), 0, s, x, border, gy, gw, C, H, W, Ho, Wo, pad, tiles_x);
  else if (K == 5)
    hipLaunchKernelGGL(dw_wgrad_kernel<5>, grid, dim3(256), 0, s, x, border, gy, gw, C, H, W, Ho, Wo, pad, tiles_x);
  else
    hipLaunchKernelGGL(dw_wgrad_kernel<7>, grid, dim3(256), 0, s, x, border, gy, gw, C, H, W, Ho, Wo, pad, tiles_x);
  return s2f_check_launch("s2f_dwconv_bwd_weight");
}
