// im2col / col2im for the dense k x k convolutions that do not take the implicit 3x3 kernels (gfx950): the stride-2 down-samplings
// and the 7x7 stem of MS_DownSampling (mmseg/models/backbones/sdtv2.py:386-421), and the 3x3 convolutions of maps the implicit
// loaders do not take.  Replaces ATen's im2col_kernel / col2im_batched_kernel (0.68 ms per C2 step, round 3): both are pure data
// movement; here every thread owns four consecutive output elements -- one 8- or 16-byte store per thread, reads coalesced along
// the row -- and col2im is a GATHER (each input pixel sums the <= ceil(k / stride)^2 column entries that cover it: no atomics, no
// zero fill, fixed summation order).
//   cols[n][(c kh + ky) kw + kx][oy Wo + ox] = x[n][c][oy s + ky - p][ox s + kx - p]   (0 outside the plane), dilation 1.
#include "s2f_common.h"

namespace {

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  typedef float4 type;
};
template <>
struct Vec4<unsigned short> {
  typedef uint2 type;
};

template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, int C, int H, int W, int kh,
                                                     int kw, int stride, int pad, int Ho, int Wo, int64_t groups_total) {
  const int L = Ho * Wo, L4 = (L + 3) >> 2, kk = kh * kw;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups_total; g += (int64_t)gridDim.x * 256) {
    const int l4 = (int)(g % L4);
    const int64_t rowi = g / L4;                       // (n, c, ky, kx)
    const int tap = (int)(rowi % kk);
    const int64_t nc = rowi / kk;
    const int ky = tap / kw, kx = tap - ky * kw;
    const T* xp = x + nc * (int64_t)H * W;
    T v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int l = l4 * 4 + o;
      const int oy = l / Wo, ox = l - oy * Wo;
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      v[o] = (l < L && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xp[(int64_t)iy * W + ix] : (T)0;
    }
    T* cp = cols + rowi * (int64_t)L + (int64_t)l4 * 4;
    if ((L & 3) == 0) {
      typename Vec4<T>::type pk;
      __builtin_memcpy(&pk, v, sizeof(pk));
      *reinterpret_cast<typename Vec4<T>::type*>(cp) = pk;
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (l4 * 4 + o < L) cp[o] = v[o];
    }
  }
}

// gx[n][c][y][x] = sum over (ky, kx) with (y + p - ky) % s == 0, (x + p - kx) % s == 0 and the quotient inside [0, Ho) x [0, Wo)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ cols, float* __restrict__ gx, int C, int H, int W,
                                                     int kh, int kw, int stride, int pad, int Ho, int Wo, int64_t groups_total) {
  const int W4 = (W + 3) >> 2, kk = kh * kw;
  const int64_t L = (int64_t)Ho * Wo;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups_total; g += (int64_t)gridDim.x * 256) {
    const int x4 = (int)(g % W4);
    const int64_t r = g / W4;
    const int y = (int)(r % H);
    const int64_t nc = r / H;
    const float* cp = cols + nc * kk * L;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < kh; ++ky) {
      const int ty = y + pad - ky;
      if (ty < 0 || ty % stride != 0) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const float* rowp = cp + (int64_t)(ky * kw + kx) * L + (int64_t)oy * Wo;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int tx = x4 * 4 + o + pad - kx;
          if (tx < 0 || tx % stride != 0) continue;
          const int ox = tx / stride;
          if (ox < Wo && x4 * 4 + o < W) acc[o] += rowp[ox];
        }
      }
    }
    float* gp = gx + (nc * H + y) * (int64_t)W + (int64_t)x4 * 4;
    if ((W & 3) == 0) {
      *reinterpret_cast<float4*>(gp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (x4 * 4 + o < W) gp[o] = acc[o];
    }
  }
}

int geometry(const char* who, int N, int C, int H, int W, int kh, int kw, int stride, int pad, int& Ho, int& Wo) {
  S2F_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, S2F_EINVAL, "%s: bad geometry", who);
  Ho = (H + 2 * pad - kh) / stride + 1;
  Wo = (W + 2 * pad - kw) / stride + 1;
  S2F_REQUIRE(Ho > 0 && Wo > 0 && (int64_t)N * C * kh * kw < ((int64_t)1 << 40), S2F_EINVAL, "%s: empty output", who);
  return S2F_OK;
}

unsigned grid_for(int64_t groups) {
  int64_t b = (groups + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > (1 << 20) ? (1 << 20) : b);
}

}  // namespace

extern "C" int s2f_im2col(const void* x, void* cols, int N, int C, int H, int W, int kh, int kw, int stride, int pad, int x_bf16,
                          void* stream) {
  S2F_REQUIRE(x && cols, S2F_EINVAL, "s2f_im2col: null pointer");
  int Ho, Wo;
  int rc = geometry("s2f_im2col", N, C, H, W, kh, kw, stride, pad, Ho, Wo);
  if (rc) return rc;
  const int64_t L4 = ((int64_t)Ho * Wo + 3) >> 2, groups = (int64_t)N * C * kh * kw * L4;
  S2F_REQUIRE((((int64_t)Ho * Wo) & 3) != 0 || (reinterpret_cast<uintptr_t>(cols) & (x_bf16 ? 7u : 15u)) == 0, S2F_EALIGN,
              "s2f_im2col: cols must be 8-byte (bf16) / 16-byte (fp32) aligned");
  if (x_bf16)
    hipLaunchKernelGGL(im2col_kernel<unsigned short>, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned short*>(x), reinterpret_cast<unsigned short*>(cols), C, H, W, kh, kw, stride, pad,
                       Ho, Wo, groups);
  else
    hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(x), reinterpret_cast<float*>(cols), C, H, W, kh, kw, stride, pad, Ho, Wo, groups);
  return s2f_check_launch("s2f_im2col");
}

extern "C" int s2f_col2im(const float* cols, float* gx, int N, int C, int H, int W, int kh, int kw, int stride, int pad,
                          void* stream) {
  S2F_REQUIRE(cols && gx, S2F_EINVAL, "s2f_col2im: null pointer");
  int Ho, Wo;
  int rc = geometry("s2f_col2im", N, C, H, W, kh, kw, stride, pad, Ho, Wo);
  if (rc) return rc;
  S2F_REQUIRE((W & 3) != 0 || s2f_aligned16(gx), S2F_EALIGN, "s2f_col2im: gx must be 16-byte aligned");
  const int64_t groups = (int64_t)N * C * H * ((W + 3) >> 2);
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream, cols, gx, C, H, W, kh, kw, stride, pad,
                     Ho, Wo, groups);
  return s2f_check_launch("s2f_col2im");
}
