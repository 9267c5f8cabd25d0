// DCNv3 core (deformable sampling + modulation) for gfx950.
//
// Reference semantics: dcnv3_core_pytorch, mmdet/models/layers/transformer/ops_dcnv3/functions/dcnv3_func.py:147-189
// (grid_sample(bilinear, zeros, align_corners=False) on the zero-padded input); the dormant CUDA op of the same
// function is ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-275 (forward) / :278-839 (backward).  Pixel-coordinate
// form of the sampling position, in zero-PADDED input coordinates:
//     px = wo*sw + c0w + (i_w - (Kw-1)/2) * dw * s + off_x * s        c0w = (dw*(Kw-1))/2,  s = offset_scale
//     py = ho*sh + c0h + (j_h - (Kh-1)/2) * dh * s + off_y * s        tap k = i_w*Kh + j_h
//
// Gather-bound (HBM/L2).  One thread owns one (n, ho, wo, g): it reads its 2*K offsets and K mask values once and
// loops over the Cg channels of the group in 16-byte pieces (NHWC: a group's channels are contiguous), so the
// backward needs NO cross-thread reduction for grad_offset / grad_mask (the reference CUDA op spreads a group over
// Cg threads and reduces through shared memory, .cuh:907-1039).  Consecutive threads are consecutive groups of one
// pixel: a wavefront reads 64*Cg contiguous floats per corner when the offsets agree.
#include "s2f_common.h"

namespace {

struct Geom {
  int N, H, W, G, Cg, Kh, Kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
  float osc;
};

struct Tap {
  int x0, y0;          // top-left corner in UNPADDED input coordinates
  float lx, ly;        // fractional parts
  bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ Tap make_tap(const Geom& g, int ho, int wo, int iw, int jh, float offx, float offy) {
  // same operation order as the oracle (padded coordinates, left to right), then shift the integer corner
  const float c0w = (float)((g.dw * (g.Kw - 1)) / 2), c0h = (float)((g.dh * (g.Kh - 1)) / 2);
  const float px = ((float)(wo * g.sw) + c0w + (float)(iw - (g.Kw - 1) / 2) * (float)g.dw * g.osc) + offx * g.osc;
  const float py = ((float)(ho * g.sh) + c0h + (float)(jh - (g.Kh - 1) / 2) * (float)g.dh * g.osc) + offy * g.osc;
  const float fx = floorf(px), fy = floorf(py);
  Tap t;
  t.lx = px - fx;
  t.ly = py - fy;
  // clamp before the int conversion so that wild offsets cannot overflow; anything outside is invalid anyway
  const float cx = fminf(fmaxf(fx, -4.0f), (float)(g.W + 2 * g.pw + 4));
  const float cy = fminf(fmaxf(fy, -4.0f), (float)(g.H + 2 * g.ph + 4));
  t.x0 = (int)cx - g.pw;
  t.y0 = (int)cy - g.ph;
  t.vx0 = t.x0 >= 0 && t.x0 < g.W;
  t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < g.W;
  t.vy0 = t.y0 >= 0 && t.y0 < g.H;
  t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < g.H;
  return t;
}

template <int V>
struct Vec;
template <>
struct Vec<4> {
  float4 v;
  __device__ __forceinline__ static Vec ld(const float* p) { Vec r; r.v = *reinterpret_cast<const float4*>(p); return r; }
  __device__ __forceinline__ static Vec zero() { Vec r; r.v = make_float4(0, 0, 0, 0); return r; }
  __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
  __device__ __forceinline__ void set(int i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else if (i == 2) v.z = f; else v.w = f; }
  __device__ __forceinline__ void st(float* p) const { *reinterpret_cast<float4*>(p) = v; }
};
template <>
struct Vec<1> {
  float v;
  __device__ __forceinline__ static Vec ld(const float* p) { Vec r; r.v = *p; return r; }
  __device__ __forceinline__ static Vec zero() { Vec r; r.v = 0.f; return r; }
  __device__ __forceinline__ float get(int) const { return v; }
  __device__ __forceinline__ void set(int, float f) { v = f; }
  __device__ __forceinline__ void st(float* p) const { *p = v; }
};

template <int V>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(const float* __restrict__ in, const float* __restrict__ off,
                                                      const float* __restrict__ msk, float* __restrict__ out, Geom g) {
  const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
  const int P = g.Kh * g.Kw;
  const int C = g.G * g.Cg;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(idx % g.G);
    int64_t r = idx / g.G;
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);
    const int64_t pix = ((int64_t)n * g.Ho + ho) * g.Wo + wo;
    const float* offp = off + (pix * g.G + gi) * P * 2;
    const float* mp = msk + (pix * g.G + gi) * P;
    const float* inb = in + (int64_t)n * g.H * g.W * C + gi * g.Cg;
    float* outp = out + pix * C + gi * g.Cg;
    for (int c = 0; c < g.Cg; c += V) {
      Vec<V> acc = Vec<V>::zero();
      for (int k = 0; k < P; ++k) {
        const int iw = k / g.Kh, jh = k % g.Kh;
        const Tap t = make_tap(g, ho, wo, iw, jh, offp[2 * k], offp[2 * k + 1]);
        const float m = mp[k];
        const float w00 = (1.f - t.ly) * (1.f - t.lx), w01 = (1.f - t.ly) * t.lx, w10 = t.ly * (1.f - t.lx),
                    w11 = t.ly * t.lx;
        const float* p00 = inb + ((int64_t)t.y0 * g.W + t.x0) * C + c;
        Vec<V> v00 = (t.vy0 && t.vx0) ? Vec<V>::ld(p00) : Vec<V>::zero();
        Vec<V> v01 = (t.vy0 && t.vx1) ? Vec<V>::ld(p00 + C) : Vec<V>::zero();
        Vec<V> v10 = (t.vy1 && t.vx0) ? Vec<V>::ld(p00 + (int64_t)g.W * C) : Vec<V>::zero();
        Vec<V> v11 = (t.vy1 && t.vx1) ? Vec<V>::ld(p00 + (int64_t)g.W * C + C) : Vec<V>::zero();
#pragma unroll
        for (int i = 0; i < V; ++i) {
          float s = v00.get(i) * (w00 * m) + v01.get(i) * (w01 * m) + v10.get(i) * (w10 * m) + v11.get(i) * (w11 * m);
          acc.set(i, acc.get(i) + s);
        }
      }
      acc.st(outp + c);
    }
  }
}

// The 3x3 / Cg = 8 geometry of the pixel decoder, fully unrolled: the 27 offset / mask values of a (pixel, group) are loaded
// first, then the 4 x 2 corner vectors of three taps at a time -- 1 + 3 dependent round trips per thread instead of the
// 2 x 9 of the generic loop (whose trip count is a run-time value, so nothing is hoisted).
__global__ __launch_bounds__(256) void dcn_fwd_k3c8_kernel(const float* __restrict__ in, const float* __restrict__ off,
                                                           const float* __restrict__ msk, float* __restrict__ out, Geom g) {
  typedef __attribute__((ext_vector_type(4))) float v4;
  const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
  const int C = g.G * 8;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(idx % g.G);
    int64_t r = idx / g.G;
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);
    const int64_t pix = ((int64_t)n * g.Ho + ho) * g.Wo + wo;
    const float* offp = off + (pix * g.G + gi) * 18;
    const float* mp = msk + (pix * g.G + gi) * 9;
    const float* inb = in + (int64_t)n * g.H * g.W * C + gi * 8;
    float ov[18], mv[9];
#pragma unroll
    for (int k = 0; k < 18; ++k) ov[k] = offp[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) mv[k] = mp[k];
    v4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < 9; k0 += 3) {
      v4 c0[3][4], c1[3][4];
      float wgt[3][4];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int k = k0 + u;
        const Tap t = make_tap(g, ho, wo, k / 3, k % 3, ov[2 * k], ov[2 * k + 1]);
        const float m = mv[k];
        wgt[u][0] = (1.f - t.ly) * (1.f - t.lx) * m; wgt[u][1] = (1.f - t.ly) * t.lx * m;
        wgt[u][2] = t.ly * (1.f - t.lx) * m;         wgt[u][3] = t.ly * t.lx * m;
        const bool b[4] = {t.vy0 && t.vx0, t.vy0 && t.vx1, t.vy1 && t.vx0, t.vy1 && t.vx1};
        const int64_t o00 = ((int64_t)t.y0 * g.W + t.x0) * C;
        const int64_t o[4] = {o00, o00 + C, o00 + (int64_t)g.W * C, o00 + (int64_t)g.W * C + C};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* p = inb + (b[q] ? o[q] : 0);
          const v4 z = {0.f, 0.f, 0.f, 0.f};
          c0[u][q] = b[q] ? *reinterpret_cast<const v4*>(p) : z;
          c1[u][q] = b[q] ? *reinterpret_cast<const v4*>(p + 4) : z;
        }
      }
      // same operation order as the generic kernel: ((v00*w00 + v01*w01) + v10*w10) + v11*w11, tap after tap
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        acc0 += ((c0[u][0] * wgt[u][0] + c0[u][1] * wgt[u][1]) + c0[u][2] * wgt[u][2]) + c0[u][3] * wgt[u][3];
        acc1 += ((c1[u][0] * wgt[u][0] + c1[u][1] * wgt[u][1]) + c1[u][2] * wgt[u][2]) + c1[u][3] * wgt[u][3];
      }
    }
    float* outp = out + pix * C + gi * 8;
    *reinterpret_cast<v4*>(outp) = acc0;
    *reinterpret_cast<v4*>(outp + 4) = acc1;
  }
}

template <int V>
__global__ __launch_bounds__(256) void dcn_bwd_kernel(const float* __restrict__ in, const float* __restrict__ off,
                                                      const float* __restrict__ msk, const float* __restrict__ gout,
                                                      float* __restrict__ gin, float* __restrict__ goff,
                                                      float* __restrict__ gmsk, Geom g) {
  const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G;
  const int P = g.Kh * g.Kw;
  const int C = g.G * g.Cg;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(idx % g.G);
    int64_t r = idx / g.G;
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int n = (int)(r / g.Ho);
    const int64_t pix = ((int64_t)n * g.Ho + ho) * g.Wo + wo;
    const float* offp = off + (pix * g.G + gi) * P * 2;
    const float* mp = msk + (pix * g.G + gi) * P;
    const int64_t img = (int64_t)n * g.H * g.W * C + gi * g.Cg;
    const float* gop = gout + pix * C + gi * g.Cg;
    for (int k = 0; k < P; ++k) {
      const int iw = k / g.Kh, jh = k % g.Kh;
      const Tap t = make_tap(g, ho, wo, iw, jh, offp[2 * k], offp[2 * k + 1]);
      const float m = mp[k];
      const float w00 = (1.f - t.ly) * (1.f - t.lx), w01 = (1.f - t.ly) * t.lx, w10 = t.ly * (1.f - t.lx),
                  w11 = t.ly * t.lx;
      const int64_t o00 = img + ((int64_t)t.y0 * g.W + t.x0) * C;
      const bool b00 = t.vy0 && t.vx0, b01 = t.vy0 && t.vx1, b10 = t.vy1 && t.vx0, b11 = t.vy1 && t.vx1;
      float am = 0.f, ax = 0.f, ay = 0.f;
      for (int c = 0; c < g.Cg; c += V) {
        const Vec<V> go = Vec<V>::ld(gop + c);
        const Vec<V> v00 = b00 ? Vec<V>::ld(in + o00 + c) : Vec<V>::zero();
        const Vec<V> v01 = b01 ? Vec<V>::ld(in + o00 + C + c) : Vec<V>::zero();
        const Vec<V> v10 = b10 ? Vec<V>::ld(in + o00 + (int64_t)g.W * C + c) : Vec<V>::zero();
        const Vec<V> v11 = b11 ? Vec<V>::ld(in + o00 + (int64_t)g.W * C + C + c) : Vec<V>::zero();
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float gv = go.get(i);
          const float a = v00.get(i), b = v01.get(i), cc = v10.get(i), d = v11.get(i);
          am += gv * (a * w00 + b * w01 + cc * w10 + d * w11);
          ax += gv * ((1.f - t.ly) * (b - a) + t.ly * (d - cc));
          ay += gv * ((1.f - t.lx) * (cc - a) + t.lx * (d - b));
          const float gm = gv * m;
          if (b00) atomicAdd(gin + o00 + c + i, gm * w00);
          if (b01) atomicAdd(gin + o00 + C + c + i, gm * w01);
          if (b10) atomicAdd(gin + o00 + (int64_t)g.W * C + c + i, gm * w10);
          if (b11) atomicAdd(gin + o00 + (int64_t)g.W * C + C + c + i, gm * w11);
        }
      }
      gmsk[(pix * g.G + gi) * P + k] = am;
      goff[((pix * g.G + gi) * P + k) * 2] = ax * m * g.osc;
      goff[((pix * g.G + gi) * P + k) * 2 + 1] = ay * m * g.osc;
    }
  }
}


// Backward with the whole (n, group) slice resident in LDS: one workgroup owns grad_input[n, :, :, g*Cg:(g+1)*Cg]
// (H*W*Cg values), so the scatter-add runs in LDS and the slice is written back with plain stores -- no global atomics,
// no pre-zeroed grad_input.  The input slice is staged in LDS as well, so every bilinear corner is an LDS read.
//
// The scatter-add accumulates in 64-bit FIXED POINT with ds_add_u64.  Measured on MI355X (tools/probe_dcn.py, C2 geometry,
// 75 M lane-adds per call): ds_add_f32 retires ~0.5 lane-adds per cycle per CU (260 us of a 420 us call, whatever the
// address pattern), the integer LDS atomics run at full LDS rate (< 10 us).  Each workgroup scales its contributions by a
// power of two chosen from its own max|grad_output| * max|mask| (bilinear weights are <= 1) so that the largest possible
// sum of 4*K*Ho*Wo contributions still fits in 62 bits: every contribution is then rounded at <= 2^-(acc_bits) of that
// bound -- finer than one fp32 ulp of any partial sum an fp32 atomic would have formed -- and the result no longer depends
// on the order of the adds (the reference's CUDA op, dcnv3_im2col_cuda.cuh:278-839, uses fp32 atomicAdd in global memory).
// float -> round-to-nearest-even 64-bit integer for |v| < 2^51 in three instructions (v_cvt_f64_f32, v_add_f64, 64-bit
// subtract): adding 1.5 * 2^52 leaves the integer in the low mantissa bits.  The generic float -> int64 conversion is a
// ~15-instruction emulation, issued 32 times per work item here.
__device__ __forceinline__ unsigned long long fix64(float v) {
  const double magic = 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong((double)v + magic) - __double_as_longlong(magic));
}

constexpr int kChunk = 256;   // output pixels whose offsets / mask / grad_output are staged in LDS at a time

// for e in [tid, n) step blockDim: use(e, load(e)) -- with the loads of U iterations issued TOGETHER from always-valid
// (clamped) indices.  Written as a plain loop with a run-time trip count, every iteration is a dependent global round trip
// (load, wait, use): the staging loops of the backward kernel were ~40 such round trips per workgroup, with one workgroup
// per CU and nothing else to run meanwhile.
template <int U, typename Load, typename Use>
__device__ __forceinline__ void grouped_loads(int n, Load load, Use use) {
  if (n <= 0) return;
  for (int e0 = threadIdx.x; e0 < n; e0 += U * (int)blockDim.x) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load(min(e0 + u * (int)blockDim.x, n - 1));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      if (e < n) use(e, v[u]);
    }
  }
}

// BANDED (round 4): the (n, group) slice of a larger map -- 64 x 32 at C3, 50 x 84 at C5 -- does not fit: `nb` workgroups share it,
// each owning the accumulators of a BAND of input rows [y_lo, y_hi).  Every workgroup walks ALL output pixels of the slice and
// scatters the corners that land in its band; grad_offset / grad_mask of a pixel are formed by ONE of them (the band its output
// row falls in) from input values read from global memory, as the forward kernel does.  Same fixed-point scale in every band (each
// scans the whole slice for it), integer adds: the result is bit-identical to the one-workgroup form.  This replaces the
// global-atomic kernel on those maps: 6.9 ms -> ~1 ms per call at C5 (6 calls per step).
template <bool BANDED>
__global__ __launch_bounds__(1024) void dcn_bwd_lds_kernel(const float* __restrict__ in, const float* __restrict__ off,
                                                          const float* __restrict__ msk, const float* __restrict__ gout,
                                                          float* __restrict__ gin, float* __restrict__ goff,
                                                          float* __restrict__ gmsk, Geom g, int acc_bits, int nb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float s_red[2][16];
  const int band = BANDED ? (int)blockIdx.x % nb : 0, ng = BANDED ? (int)blockIdx.x / nb : (int)blockIdx.x;
  const int n = ng / g.G, gi = ng % g.G;
  const int Cg = g.Cg, C = g.G * g.Cg, P = g.Kh * g.Kw;
  const int npix_in = g.H * g.W, npix_out = g.Ho * g.Wo;
  const int rows_b = BANDED ? (g.H + nb - 1) / nb : g.H;                 // input rows per band
  const int y_lo = band * rows_b, y_hi = min(g.H, y_lo + rows_b);
  const int npix_band = max(0, y_hi - y_lo) * g.W;
  const int rows_ob = BANDED ? (g.Ho + nb - 1) / nb : g.Ho;               // output rows whose offset / mask gradients this band forms
  // rows padded to Cg + 1 accumulators: with Cg = 8 consecutive pixels would lie 64 bytes apart and the 64 lanes of one
  // ds_add_u64 (same channel, neighbouring pixels) would share 4 bank pairs
  const int CA = Cg + 1;
  unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(smem_raw);      // [band pixels][Cg + 1] fixed-point grad_input
  float* s_in = reinterpret_cast<float*>(s_acc + (size_t)(BANDED ? rows_b * g.W : npix_in) * CA);      // [H*W][Cg] (not BANDED)
  float* s_off = s_in + (BANDED ? 0 : npix_in * Cg);       // [kChunk][P*2]
  float* s_msk = s_off + kChunk * P * 2;    // [kChunk][P]
  float* s_go = s_msk + kChunk * P;         // [kChunk][Cg]
  const float* inb = in + (int64_t)n * npix_in * C + gi * Cg;
  const float* gob = gout + (int64_t)n * npix_out * C + gi * Cg;
  if (BANDED) {
    for (int e = threadIdx.x; e < npix_band * CA; e += blockDim.x) s_acc[e] = 0ull;
  } else {
    grouped_loads<8>(npix_in * Cg, [&](int e) { return inb[(int64_t)(e / Cg) * C + e % Cg]; },
                     [&](int e, float v) {
                       s_in[e] = v;
                       s_acc[(e / Cg) * CA + e % Cg] = 0ull;
                     });
  }
  // scale of this slice: max|grad_output| * max|mask| bounds every contribution
  float mg = 0.f, mm = 0.f;
  grouped_loads<8>(npix_out * Cg, [&](int e) { return gob[(int64_t)(e / Cg) * C + e % Cg]; },
                   [&](int, float v) { mg = fmaxf(mg, fabsf(v)); });
  grouped_loads<8>(npix_out * P, [&](int e) { return msk[(((int64_t)n * npix_out + e / P) * g.G + gi) * P + e % P]; },
                   [&](int, float v) { mm = fmaxf(mm, fabsf(v)); });
  for (int o = 32; o > 0; o >>= 1) {
    mg = fmaxf(mg, __shfl_xor(mg, o, 64));
    mm = fmaxf(mm, __shfl_xor(mm, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_red[0][threadIdx.x >> 6] = mg;
    s_red[1][threadIdx.x >> 6] = mm;
  }
  __syncthreads();
  mg = 0.f, mm = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
    mg = fmaxf(mg, s_red[0][w]);
    mm = fmaxf(mm, s_red[1][w]);
  }
  // NaN/Inf gradients: fmaxf drops NaN; an infinite bound gives a negative shift and saturating conversions (garbage in,
  // garbage out, but no undefined behaviour)
  int ex = 0;
  const float bound = mg * mm;
  if (bound > 0.f) frexpf(fminf(bound, 3.0e38f), &ex);          // bound < 2^ex
  const int shift = max(-126, min(126, acc_bits - ex));
  const float scale = ldexpf(1.f, shift);
  const double inv_scale = ldexp(1.0, -shift);

  const int slots = blockDim.x;
  for (int p0 = 0; p0 < npix_out; p0 += kChunk) {
    // The offsets / mask / grad_output of a pixel chunk are staged with one bulk load (every thread of the 16 waves
    // issues its loads at once) instead of a dependent global round trip per item.
    const int pc = min(kChunk, npix_out - p0);
    grouped_loads<5>(pc * P * 2, [&](int e) { return off[(((int64_t)n * npix_out + p0 + e / (P * 2)) * g.G + gi) * P * 2 + e % (P * 2)]; },
                     [&](int e, float v) { s_off[e] = v; });
    grouped_loads<3>(pc * P, [&](int e) { return msk[(((int64_t)n * npix_out + p0 + e / P) * g.G + gi) * P + e % P]; },
                     [&](int e, float v) { s_msk[e] = v; });
    grouped_loads<2>(pc * Cg, [&](int e) { return gob[(int64_t)(p0 + e / Cg) * C + e % Cg]; }, [&](int e, float v) { s_go[e] = v; });
    __syncthreads();
    // Work item = (tap, pixel of the chunk), one per thread, the Cg channels in an inner loop: the tap geometry is computed
    // once per item.  (Spreading the channels of an item over Cg adjacent lanes -- conflict-free LDS adds -- mattered with
    // the slow ds_add_f32; with integer LDS atomics it only replicates the tap arithmetic Cg times: 199 vs 178 us.)
    const int items = pc * P;
    for (int it0 = 0; it0 < items; it0 += slots) {
      const int it = it0 + threadIdx.x;
      const bool live = it < items;
      const int k = live ? it / pc : 0, pl = live ? it % pc : 0;
      const int pix = p0 + pl;
      const int ho = pix / g.Wo, wo = pix % g.Wo;
      const float offx = s_off[(pl * P + k) * 2], offy = s_off[(pl * P + k) * 2 + 1];
      const float m = s_msk[pl * P + k];
      const Tap t = make_tap(g, ho, wo, k / g.Kh, k % g.Kh, offx, offy);
      const float w00 = (1.f - t.ly) * (1.f - t.lx), w01 = (1.f - t.ly) * t.lx, w10 = t.ly * (1.f - t.lx),
                  w11 = t.ly * t.lx;
      const bool b00 = live && t.vy0 && t.vx0, b01 = live && t.vy0 && t.vx1, b10 = live && t.vy1 && t.vx0,
                 b11 = live && t.vy1 && t.vx1;
      const int o00 = (t.y0 * g.W + t.x0) * Cg;
      const int o01 = o00 + Cg, o10 = o00 + g.W * Cg, o11 = o10 + Cg;
      // BANDED: scatter only the corners whose row lies in this band; the offset / mask gradients of a pixel belong to one band
      const bool in0 = !BANDED || (t.y0 >= y_lo && t.y0 < y_hi), in1 = !BANDED || (t.y0 + 1 >= y_lo && t.y0 + 1 < y_hi);
      const bool s00 = b00 && in0, s01 = b01 && in0, s10 = b10 && in1, s11 = b11 && in1;
      const bool mine = !BANDED || min(nb - 1, ho / rows_ob) == band;
      // BANDED: an item whose corners lie in other bands and whose offset / mask gradients another band forms costs this workgroup
      // its tap arithmetic only (neighbouring pixels share their band: whole wavefronts skip the channel loop; without the test
      // every band repeated the channel loop of every item, nb times the work)
      if (BANDED && !(mine || s00 || s01 || s10 || s11)) continue;
      const int a00 = ((t.y0 - y_lo) * g.W + t.x0) * CA;
      float am = 0.f, ax = 0.f, ay = 0.f;
      // channels four at a time: 16-byte LDS reads of the grad_output row and the four corner rows (scalar reads put the
      // lanes of a wave Cg floats apart = a Cg-way bank conflict)
      auto body = [&](int c, float gv, float a, float b, float cc, float d) __attribute__((always_inline)) {
        am += gv * (a * w00 + b * w01 + cc * w10 + d * w11);
        ax += gv * ((1.f - t.ly) * (b - a) + t.ly * (d - cc));
        ay += gv * ((1.f - t.lx) * (cc - a) + t.lx * (d - b));
        const float gm = gv * m * scale;
        if (s00) atomicAdd(&s_acc[a00 + c], fix64(gm * w00));
        if (s01) atomicAdd(&s_acc[a00 + CA + c], fix64(gm * w01));
        if (s10) atomicAdd(&s_acc[a00 + g.W * CA + c], fix64(gm * w10));
        if (s11) atomicAdd(&s_acc[a00 + g.W * CA + CA + c], fix64(gm * w11));
      };
      if (BANDED) {
        // input corners from global memory (rows of Cg floats, C apart), only for the pixels whose offset / mask gradients are ours
        const int64_t pc0 = (int64_t)(t.y0 * g.W + t.x0) * C;
        if ((Cg & 3) == 0 && (reinterpret_cast<uintptr_t>(inb) & 15u) == 0 && ((rows_b * g.W * CA) & 1) == 0) {      // (staging 16-byte aligned)
          const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int c = 0; c < Cg; c += 4) {
            const float4 gv = live ? *reinterpret_cast<const float4*>(&s_go[pl * Cg + c]) : z4;
            const float4 a = (mine && b00) ? *reinterpret_cast<const float4*>(inb + pc0 + c) : z4;
            const float4 b = (mine && b01) ? *reinterpret_cast<const float4*>(inb + pc0 + C + c) : z4;
            const float4 cc = (mine && b10) ? *reinterpret_cast<const float4*>(inb + pc0 + (int64_t)g.W * C + c) : z4;
            const float4 d = (mine && b11) ? *reinterpret_cast<const float4*>(inb + pc0 + (int64_t)g.W * C + C + c) : z4;
            body(c, gv.x, a.x, b.x, cc.x, d.x);
            body(c + 1, gv.y, a.y, b.y, cc.y, d.y);
            body(c + 2, gv.z, a.z, b.z, cc.z, d.z);
            body(c + 3, gv.w, a.w, b.w, cc.w, d.w);
          }
        } else {
          for (int c = 0; c < Cg; ++c) {
            const float gv = live ? s_go[pl * Cg + c] : 0.f;
            const float a = (mine && b00) ? inb[pc0 + c] : 0.f, b = (mine && b01) ? inb[pc0 + C + c] : 0.f,
                        cc = (mine && b10) ? inb[pc0 + (int64_t)g.W * C + c] : 0.f,
                        d = (mine && b11) ? inb[pc0 + (int64_t)g.W * C + C + c] : 0.f;
            body(c, gv, a, b, cc, d);
          }
        }
      } else if ((Cg & 3) == 0 && ((npix_in * CA) & 1) == 0) {          // second test: s_in starts 16-byte aligned
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < Cg; c += 4) {
          const float4 gv = live ? *reinterpret_cast<const float4*>(&s_go[pl * Cg + c]) : z4;
          const float4 a = b00 ? *reinterpret_cast<const float4*>(&s_in[o00 + c]) : z4;
          const float4 b = b01 ? *reinterpret_cast<const float4*>(&s_in[o01 + c]) : z4;
          const float4 cc = b10 ? *reinterpret_cast<const float4*>(&s_in[o10 + c]) : z4;
          const float4 d = b11 ? *reinterpret_cast<const float4*>(&s_in[o11 + c]) : z4;
          body(c, gv.x, a.x, b.x, cc.x, d.x);
          body(c + 1, gv.y, a.y, b.y, cc.y, d.y);
          body(c + 2, gv.z, a.z, b.z, cc.z, d.z);
          body(c + 3, gv.w, a.w, b.w, cc.w, d.w);
        }
      } else {
        for (int c = 0; c < Cg; ++c)
          body(c, live ? s_go[pl * Cg + c] : 0.f, b00 ? s_in[o00 + c] : 0.f, b01 ? s_in[o01 + c] : 0.f,
               b10 ? s_in[o10 + c] : 0.f, b11 ? s_in[o11 + c] : 0.f);
      }
      if (live && mine) {
        const int64_t ob = (((int64_t)n * npix_out + pix) * g.G + gi) * P + k;
        gmsk[ob] = am;
        goff[ob * 2] = ax * m * g.osc;
        goff[ob * 2 + 1] = ay * m * g.osc;
      }
    }
    __syncthreads();
  }
  float* ginb = gin + ((int64_t)n * npix_in + (int64_t)y_lo * g.W) * C + gi * Cg;
  for (int e = threadIdx.x; e < npix_band * Cg; e += blockDim.x) {
    const int p = e / Cg, c = e % Cg;
    ginb[(int64_t)p * C + c] = (float)((double)(long long)s_acc[p * CA + c] * inv_scale);
  }
}

int make_geom(Geom& g, int N, int H, int W, int G, int Cg, int Kh, int Kw, int sh, int sw, int ph, int pw, int dh, int dw,
              float osc, const char* who) {
  S2F_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && Cg > 0 && Kh > 0 && Kw > 0 && sh > 0 && sw > 0 && ph >= 0 &&
                  pw >= 0 && dh > 0 && dw > 0,
              S2F_EINVAL, "%s: bad geometry", who);
  g = Geom{N, H, W, G, Cg, Kh, Kw, sh, sw, ph, pw, dh, dw, 0, 0, osc};
  g.Ho = (H + 2 * ph - (dh * (Kh - 1) + 1)) / sh + 1;
  g.Wo = (W + 2 * pw - (dw * (Kw - 1) + 1)) / sw + 1;
  S2F_REQUIRE(g.Ho > 0 && g.Wo > 0, S2F_EINVAL, "%s: empty output", who);
  return S2F_OK;
}

inline int grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int s2f_dcnv3_fwd(const float* input, const float* offset, const float* mask, float* output, int N, int H,
                             int W, int G, int Cg, int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w,
                             int dil_h, int dil_w, float offset_scale, void* stream) {
  S2F_REQUIRE(input && offset && mask && output, S2F_EINVAL, "s2f_dcnv3_fwd: null pointer");
  Geom g;
  int rc = make_geom(g, N, H, W, G, Cg, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, offset_scale,
                     "s2f_dcnv3_fwd");
  if (rc != S2F_OK) return rc;
  const int64_t total = (int64_t)N * g.Ho * g.Wo * G;
  const bool vec = (Cg % 4 == 0) && s2f_aligned16(input) && s2f_aligned16(output);
  if (vec && Cg == 8 && Kh == 3 && Kw == 3)
    hipLaunchKernelGGL(dcn_fwd_k3c8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                       output, g);
  else if (vec)
    hipLaunchKernelGGL(dcn_fwd_kernel<4>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                       output, g);
  else
    hipLaunchKernelGGL(dcn_fwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                       output, g);
  return s2f_check_launch("s2f_dcnv3_fwd");
}

extern "C" int s2f_dcnv3_bwd(const float* input, const float* offset, const float* mask, const float* grad_output,
                             float* grad_input, float* grad_offset, float* grad_mask, int N, int H, int W, int G, int Cg,
                             int Kh, int Kw, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                             float offset_scale, void* stream) {
  S2F_REQUIRE(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, S2F_EINVAL,
              "s2f_dcnv3_bwd: null pointer");
  Geom g;
  int rc = make_geom(g, N, H, W, G, Cg, Kh, Kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, offset_scale,
                     "s2f_dcnv3_bwd");
  if (rc != S2F_OK) return rc;
  const int64_t total = (int64_t)N * g.Ho * g.Wo * G;
  const size_t lds = (size_t)H * W * (8 * (Cg + 1) + 4 * Cg) + sizeof(float) * (size_t)kChunk * (Kh * Kw * 3 + Cg);
  constexpr size_t kMaxDynLds = 160 * 1024 - 256;          // the kernel also holds 128 B of static LDS
  // accumulator bits so that 4*K*Ho*Wo contributions of magnitude < 2^acc_bits cannot overflow 62 bits
  int count_bits = 0;
  while (((int64_t)1 << count_bits) < (int64_t)4 * Kh * Kw * g.Ho * g.Wo) ++count_bits;
  // ... and so that a single contribution stays inside fix64's exact range |v| < 2^51: at most 50 bits.  (Round 6: maps of <= 28
  // output pixels -- the 4 x 4 maps of the 64 x 64 plumbing configuration -- gave count_bits = 10, i.e. 52 accumulator bits; a
  // contribution above half the slice's bound then left fix64's range and came back as garbage: up to 26 % error in grad_input of
  // single (n, group) slices, data-dependent, forward exact.  tools/debug_dcn_core.py; hidden by the 5e-2 tolerances until the
  // tiny-config tests were held to the measured reference-vs-oracle gap.)
  if (count_bits < 12) count_bits = 12;
  const char* fb = getenv("S2F_DCN_FORCE_BANDS");                   // tests: band a map that would fit (read per call)
  const int force_bands = fb ? atoi(fb) : 0;
  if (lds <= kMaxDynLds && force_bands <= 1) {
    // (n, group) slice fits in the CU's 160 KiB LDS: no global atomics
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_bwd_lds_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
      S2F_REQUIRE(e == hipSuccess, S2F_ELAUNCH, "s2f_dcnv3_bwd: cannot raise the dynamic LDS limit: %s",
                  hipGetErrorString(e));
      raised = true;
    }
    // 1024 threads = 16 wavefronts on the one CU that holds the slice
    hipLaunchKernelGGL(dcn_bwd_lds_kernel<false>, dim3(N * G), dim3(1024), lds, (hipStream_t)stream, input, offset, mask,
                       grad_output, grad_input, grad_offset, grad_mask, g, 62 - count_bits, 1);
    return s2f_check_launch("s2f_dcnv3_bwd");
  }
  // larger maps: bands of input rows, one workgroup each (the accumulators of a band in LDS, the input read from global memory)
  {
    const size_t staging = sizeof(float) * (size_t)kChunk * (Kh * Kw * 3 + Cg);
    int nb = force_bands > 1 ? force_bands : 2;
    auto band_lds = [&](int b) { return (size_t)((H + b - 1) / b) * W * 8 * (Cg + 1) + staging; };
    while (force_bands <= 1 && nb < H && nb < 64 && band_lds(nb) > kMaxDynLds) ++nb;
    if (force_bands <= 1) {
      // A workgroup pays ~57 us per 256-pixel chunk whose rows are its own (LDS atomics) and ~7 us for every other chunk it walks
      // (tools/probe_dcn.py, fitted over 3 .. 10 bands at 50 x 84); one workgroup per CU.  Among the next few band counts take
      // the one with the least  (8 / nb + 1) x rounds:  4 bands instead of 3 at C5 (512 workgroups = two full rounds, 853 -> 729 us)
      auto cost = [&](int b) { return (8.0 / b + 1.0) * (double)(((int64_t)N * G * b + 255) / 256); };
      int best = nb;
      for (int b = nb + 1; b <= nb + 3 && b <= H && b < 64; ++b)
        if (cost(b) < cost(best) - 1e-9) best = b;
      nb = best;
    }
    if (nb <= H && band_lds(nb) <= kMaxDynLds && (int64_t)N * G * nb < ((int64_t)1 << 31)) {
      static bool raised_b = false;
      if (!raised_b) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_bwd_lds_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds);
        S2F_REQUIRE(e == hipSuccess, S2F_ELAUNCH, "s2f_dcnv3_bwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        raised_b = true;
      }
      hipLaunchKernelGGL(dcn_bwd_lds_kernel<true>, dim3(N * G * nb), dim3(1024), band_lds(nb), (hipStream_t)stream, input, offset,
                         mask, grad_output, grad_input, grad_offset, grad_mask, g, 62 - count_bits, nb);
      return s2f_check_launch("s2f_dcnv3_bwd");
    }
  }
  // global-atomic path: grad_input is accumulated into, zero it first (stream-ordered, capturable)
  if (s2f_zero_async(grad_input, sizeof(float) * (size_t)N * H * W * G * Cg, (hipStream_t)stream) != S2F_OK)
    return s2f_check_launch("s2f_dcnv3_bwd");
  const bool vec = (Cg % 4 == 0) && s2f_aligned16(input) && s2f_aligned16(grad_output);
  if (vec)
    hipLaunchKernelGGL(dcn_bwd_kernel<4>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                       grad_output, grad_input, grad_offset, grad_mask, g);
  else
    hipLaunchKernelGGL(dcn_bwd_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, input, offset, mask,
                       grad_output, grad_input, grad_offset, grad_mask, g);
  return s2f_check_launch("s2f_dcnv3_bwd");
}
