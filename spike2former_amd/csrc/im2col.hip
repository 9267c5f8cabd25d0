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

// grid (ceil(Ho Wo / 4 / 256), C kh kw, N): the row of the column matrix comes from the block index, a thread's four outputs lie in
// one output row when Wo % 4 == 0 (one division per thread; the flat-index form spent ~8 integer divisions per 16 bytes stored
// and ran the stem's 308 MB matrix at 2.2 TB/s).  S = the stride as a compile-time constant (0: run-time).
template <typename T, int S>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, int C, int H, int W, int kh,
                                                     int kw, int stride_rt, int pad, int Ho, int Wo) {
  const int stride = S > 0 ? S : stride_rt;
  const int L = Ho * Wo, l0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
  if (l0 >= L) return;
  const int rowi = blockIdx.y, kk = kh * kw;                 // (c, ky, kx)
  const int c = rowi / kk, tap = rowi - c * kk;              // wave-uniform
  const int ky = tap / kw, kx = tap - ky * kw;
  const T* xp = x + ((int64_t)blockIdx.z * C + c) * (int64_t)H * W;
  T v[4];
  int oy = l0 / Wo, ox = l0 - oy * Wo;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
    v[o] = (l0 + o < L && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xp[(int64_t)iy * W + ix] : (T)0;
    if (++ox == Wo) ox = 0, ++oy;
  }
  T* cp = cols + ((int64_t)blockIdx.z * gridDim.y + rowi) * (int64_t)L + l0;
  if ((L & 3) == 0) {
    typename Vec4<T>::type pk;
    __builtin_memcpy(&pk, v, sizeof(pk));
    *reinterpret_cast<typename Vec4<T>::type*>(cp) = pk;
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (l0 + o < L) cp[o] = v[o];
  }
}

// gx[n][c][y][x] = sum over (ky, kx) with (y + p - ky) % s == 0, (x + p - kx) % s == 0 and the quotient inside [0, Ho) x [0, Wo).
// grid (ceil(H ceil(W / 4) / 256), planes in y (, z)): the plane (n, c) comes from the block index, one division per thread.
template <int S>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ cols, float* __restrict__ gx, int64_t planes, int H,
                                                     int W, int kh, int kw, int stride_rt, int pad, int Ho, int Wo) {
  const int stride = S > 0 ? S : stride_rt;
  const int kk = kh * kw, W4 = (W + 3) >> 2;
  const int64_t L = (int64_t)Ho * Wo;
  const int cell = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int64_t nc = (int64_t)blockIdx.z * gridDim.y + blockIdx.y;
  if (cell >= H * W4 || nc >= planes) return;
  const int y = cell / W4, x0 = (cell - y * W4) * 4;
  const float* cp = cols + nc * kk * L;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < kh; ++ky) {
    const int ty = y + pad - ky;
    if (ty < 0) break;                                       // larger ky only lower it further
    const int oy = ty / stride;
    if (oy * stride != ty || oy >= Ho) continue;
    for (int kx = 0; kx < kw; ++kx) {
      const float* rowp = cp + (int64_t)(ky * kw + kx) * L + (int64_t)oy * Wo;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int tx = x0 + o + pad - kx;
        const int ox = tx / stride;
        if (tx >= 0 && ox * stride == tx && ox < Wo && x0 + o < W) acc[o] += rowp[ox];
      }
    }
  }
  float* gp = gx + (nc * H + y) * (int64_t)W + x0;
  if ((W & 3) == 0) {
    *reinterpret_cast<float4*>(gp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (x0 + o < W) gp[o] = acc[o];
  }
}

int geometry(const char* who, int N, int C, int H, int W, int kh, int kw, int stride, int pad, int& Ho, int& Wo) {
  S2F_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, S2F_EINVAL, "%s: bad geometry", who);
  Ho = (H + 2 * pad - kh) / stride + 1;
  Wo = (W + 2 * pad - kw) / stride + 1;
  S2F_REQUIRE(Ho > 0 && Wo > 0, S2F_EINVAL, "%s: empty output", who);
  S2F_REQUIRE((int64_t)C * kh * kw < 65536 && N < 65536 && (int64_t)N * C < ((int64_t)1 << 30) && (int64_t)H * W < ((int64_t)1 << 31) &&
                  (int64_t)Ho * Wo < ((int64_t)1 << 31),
              S2F_EINVAL, "%s: shape too large for the launch grid", who);
  return S2F_OK;
}

}  // namespace

extern "C" int s2f_im2col(const void* x, void* cols, int N, int C, int H, int W, int kh, int kw, int stride, int pad, int x_bf16,
                          void* stream) {
  S2F_REQUIRE(x && cols, S2F_EINVAL, "s2f_im2col: null pointer");
  int Ho, Wo;
  int rc = geometry("s2f_im2col", N, C, H, W, kh, kw, stride, pad, Ho, Wo);
  if (rc) return rc;
  S2F_REQUIRE((((int64_t)Ho * Wo) & 3) != 0 || (reinterpret_cast<uintptr_t>(cols) & (x_bf16 ? 7u : 15u)) == 0, S2F_EALIGN,
              "s2f_im2col: cols must be 8-byte (bf16) / 16-byte (fp32) aligned");
  const dim3 grid((unsigned)((((int64_t)Ho * Wo + 3) / 4 + 255) / 256), (unsigned)(C * kh * kw), (unsigned)N);
  hipStream_t s = (hipStream_t)stream;
#define S2F_IM2COL(T, SV)                                                                                                        \
  hipLaunchKernelGGL((im2col_kernel<T, SV>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(x), reinterpret_cast<T*>(cols), C, H, \
                     W, kh, kw, stride, pad, Ho, Wo)
  if (x_bf16) {
    if (stride == 1) S2F_IM2COL(unsigned short, 1);
    else if (stride == 2) S2F_IM2COL(unsigned short, 2);
    else S2F_IM2COL(unsigned short, 0);
  } else {
    if (stride == 1) S2F_IM2COL(float, 1);
    else if (stride == 2) S2F_IM2COL(float, 2);
    else S2F_IM2COL(float, 0);
  }
#undef S2F_IM2COL
  return s2f_check_launch("s2f_im2col");
}

extern "C" int s2f_col2im(const float* cols, float* gx, int N, int C, int H, int W, int kh, int kw, int stride, int pad,
                          void* stream) {
  S2F_REQUIRE(cols && gx, S2F_EINVAL, "s2f_col2im: null pointer");
  int Ho, Wo;
  int rc = geometry("s2f_col2im", N, C, H, W, kh, kw, stride, pad, Ho, Wo);
  if (rc) return rc;
  S2F_REQUIRE((W & 3) != 0 || s2f_aligned16(gx), S2F_EALIGN, "s2f_col2im: gx must be 16-byte aligned");
  const int64_t planes = (int64_t)N * C;
  const unsigned gy = (unsigned)(planes < 32768 ? planes : 32768);
  const dim3 grid((unsigned)(((int64_t)H * ((W + 3) / 4) + 255) / 256), gy, (unsigned)((planes + gy - 1) / gy));
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1)
    hipLaunchKernelGGL(col2im_kernel<1>, grid, dim3(256), 0, s, cols, gx, planes, H, W, kh, kw, stride, pad, Ho, Wo);
  else if (stride == 2)
    hipLaunchKernelGGL(col2im_kernel<2>, grid, dim3(256), 0, s, cols, gx, planes, H, W, kh, kw, stride, pad, Ho, Wo);
  else
    hipLaunchKernelGGL(col2im_kernel<0>, grid, dim3(256), 0, s, cols, gx, planes, H, W, kh, kw, stride, pad, Ho, Wo);
  return s2f_check_launch("s2f_col2im");
}
