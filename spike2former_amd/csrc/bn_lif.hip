// BatchNorm (train or eval) fused with the conv bias, the residual add and the following Q_IFNode, for gfx950.
//
// Reference chain (e.g. MS_ConvBlock / SepConv / MS_MLP, mmseg/models/backbones/sdtv2.py:167-255; Sequential(conv, BN)
// + Q_IFNode everywhere in the head):   t = conv(x) + b ;  u = BN(t) [+ residual] ;  y = Q_IFNode(u)
// which the reference runs as ~12 elementwise ATen kernels forward and as many backward.  Here:
//   forward :  bn_stats  (1 read of z)  ->  bn_finalize (C threads: mean, rstd, running-stat update)
//              bn_apply  (1 read of z [+ residual], writes u and/or the spikes y + 1-bit in-range mask)
//   backward:  bn_bwd_reduce (per-channel sum(gu), sum(gu * xhat)) -> bn_bwd_apply (writes gz [, g_residual])
//              with  gu = g_u + LIF-STE(g_y)  formed on the fly from the bit mask.
// All kernels are HBM-bound streams over [N, C, L] (channel-major, L contiguous): lane l of a wave owns 4 consecutive
// elements (one 16-byte access; L % 4 == 0 keeps them inside one channel row), the in-range bits are wave ballots as in
// lif.hip.  Per-channel parameters are 2-3 floats per lane served from L1/L2.
#include "gemm_common.h"
#include <cstring>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / S2F_WAVE;

__device__ __forceinline__ double wave_sum_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (a, b) = this thread's sum and sum of squares of d = x - pivot over `count` elements of the workgroup in total (round 6: shifted
// sums, see S2F_BN_PIVOT below -- the fp32 squares are taken of the SHIFTED values, whose mean is of the order of their spread);
// un-shifted in fp64 -- sum x = sum d + n K, sum x^2 = sum d^2 + 2 K sum d + n K^2, where the cancellation of the consumer's
// E[x^2] - E[x]^2 costs 1e-16 mean^2 instead of the 1e-7 mean^2 of fp32 squares -- and added to dst[0], dst[1].
__device__ __forceinline__ void block_atomic_add2(double a, double b, double* dst, double pivot = 0.0, double count = 0.0) {
  __shared__ double red[2 * kWaves];
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[2 * w] = a;
    red[2 * w + 1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
    for (int i = 0; i < kWaves; ++i) {
      sa += red[2 * i];
      sb += red[2 * i + 1];
    }
    sb += 2.0 * pivot * sa + count * pivot * pivot;
    sa += count * pivot;
    atomicAdd(dst, sa);
    atomicAdd(dst + 1, sb);
  }
}

// grid (C, S): block (c, s) reduces rows n*C + c, columns [s*slice, (s+1)*slice).  The (row, 16-byte column) items of the
// block are flattened and walked 4 at a time with all loads issued before the first use (the per-row loop of a naive
// version serialises N dependent HBM round trips per thread).
constexpr int kUnroll = 4;

__global__ __launch_bounds__(kBlock) void bn_stats_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                          double* __restrict__ sums, int N, int C, int L, int slice) {
  const int c = blockIdx.x;
  const int l0 = blockIdx.y * slice, l1 = min(L, l0 + slice);
  const float b = bias ? bias[c] : 0.0f;
  const int per_row = (l1 - l0) >> 2;                    // float4 items per row (slice and L are multiples of 4)
  const int total = N * per_row;
  const float piv = z[(int64_t)c * L + l0] + b;          // pivot of this (channel, slice): its first element (shifted sums)
  float ps = 0.f, pq = 0.f;                              // fp32 partials over <= total/256 * 4 elements per thread
  for (int it = threadIdx.x; it < total; it += kBlock * kUnroll) {
    float4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int id = it + u * kBlock;
      if (id < total) {
        const int n = id / per_row, q = id - n * per_row;
        v[u] = *reinterpret_cast<const float4*>(z + ((int64_t)n * C + c) * L + l0 + q * 4);
      } else {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const bool ok = it + u * kBlock < total;            // items past the slice contribute exactly zero
      const float a0 = ok ? (v[u].x + b) - piv : 0.f, a1 = ok ? (v[u].y + b) - piv : 0.f, a2 = ok ? (v[u].z + b) - piv : 0.f,
                  a3 = ok ? (v[u].w + b) - piv : 0.f;
      ps += (a0 + a1) + (a2 + a3);
      pq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  block_atomic_add2((double)ps, (double)pq, sums + 2 * c, (double)piv, 4.0 * (double)total);
}

// Any row length (L % 4 != 0: rows are not 16-byte aligned): the same reduction with one element per thread and step.  Only
// odd shapes land here (e.g. a decoder with 10 queries in the shrunken test configuration); nothing of the path at its real sizes.
__global__ __launch_bounds__(kBlock) void bn_stats_any_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              double* __restrict__ sums, int N, int C, int L, int slice) {
  const int c = blockIdx.x;
  const int l0 = blockIdx.y * slice, l1 = min(L, l0 + slice);
  const float b = bias ? bias[c] : 0.0f;
  const int len = l1 - l0, total = N * len;
  const float piv = total > 0 ? z[(int64_t)c * L + l0] + b : 0.f;
  float ps = 0.f, pq = 0.f;
  for (int id = threadIdx.x; id < total; id += kBlock) {
    const int n = id / len, q = id - n * len;
    const float a = (z[((int64_t)n * C + c) * L + l0 + q] + b) - piv;
    ps += a;
    pq += a * a;
  }
  block_atomic_add2((double)ps, (double)pq, sums + 2 * c, (double)piv, (double)total);
}

// Per-channel statistics derived identically by every lane that needs them (deterministic: same inputs, same ops).
// training: mean / biased variance from the fp64 sums of bn_stats_kernel -- fp64 multiplies by the precomputed 1/count,
// no fp64 division (a wave-level fp64 divide per tile made the streaming kernel VALU-bound: 3.7 vs 5.3 TB/s);
// eval: the running statistics.
struct ChanStat {
  float mean, rstd, var;
};
// (s1, s2) = sum and sum of squares of the channel; `shift` is added to the mean (statistics taken of z by a GEMM epilogue, the
// BatchNorm input being z + conv_bias: the variance does not see the constant)
__device__ __forceinline__ ChanStat chan_stat_of(double s1, double s2, double shift, double inv_count, float eps) {
  ChanStat r;
  const double m = s1 * inv_count;
  double v = s2 * inv_count - m * m;
  if (v < 0) v = 0;
  r.mean = (float)(m + shift);
  r.var = (float)v;
  r.rstd = 1.0f / sqrtf(r.var + eps);
  return r;
}
__device__ __forceinline__ ChanStat chan_stat(const double* __restrict__ sums, const float* __restrict__ running_mean,
                                              const float* __restrict__ running_var, int c, double inv_count, float eps,
                                              int training) {
  if (training) return chan_stat_of(sums[2 * c], sums[2 * c + 1], 0.0, inv_count, eps);
  ChanStat r;
  r.mean = running_mean[c];
  r.var = running_var[c];
  r.rstd = 1.0f / sqrtf(r.var + eps);
  return r;
}
// Statistics handed over as per-tile PARTIALS of the producing GEMM's epilogue (pgemm.hip epi_row_partials): part[(c * P + p) * 2 +
// {0, 1}], p < P (channel-major).  bn_partials_finalize_kernel adds the P partials of every channel in fp64 in a fixed order and
// writes the sums s2f_bn_stats would have produced (of z + bias: the shift by the conv bias is applied to the sums) -- a launch of
// 2.5-4 us that reads P * C * 8 bytes in place of a pass over z (5-47 us at C2).  (Built and measured first, round 4: the apply
// kernels adding the partials of the channels they touch in their own prologue, no extra launch -- +9.7 us per launch of the
// row-walking kernel on the large maps (thousands of partials per channel behind two block-wide barriers, in front of every
// workgroup's first tile), +2.4 us on the 100-token decoder rows: 0.41 ms per step against 0.31 for these launches, and the fused
// BatchNorm + neuron kernel itself -- the kernel the roofline figure is quoted on -- got 10 % slower.)
// grid: one workgroup per channel (P > 256) or one wavefront per channel (four channels per workgroup).
__global__ __launch_bounds__(kBlock) void bn_partials_finalize_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                      double* __restrict__ sums, int P, int C, double count,
                                                                      int per_wave) {
  __shared__ double red[2 * kWaves];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = per_wave ? blockIdx.x * kWaves + w : blockIdx.x;
  const int first = per_wave ? lane : threadIdx.x, stride = per_wave ? 64 : kBlock;
  double a = 0, b = 0;
  if (c < C) {
    for (int p0 = first; p0 < P; p0 += 8 * stride) {          // eight loads in flight
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * stride;
        v[u] = p < P ? *reinterpret_cast<const float2*>(part + ((int64_t)c * P + p) * 2) : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (double)v[u].x;
        b += (double)v[u].y;
      }
    }
  }
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  if (!per_wave) {
    if (lane == 0) {
      red[2 * w] = a;
      red[2 * w + 1] = b;
    }
    __syncthreads();
    a = 0, b = 0;
    for (int i = 0; i < kWaves; ++i) {
      a += red[2 * i];
      b += red[2 * i + 1];
    }
  }
  if (c < C && (per_wave ? lane == 0 : threadIdx.x == 0)) {
    const double sh = bias ? (double)bias[c] : 0.0;          // sum(z + sh) = s1 + n sh ; sum((z + sh)^2) = s2 + 2 sh s1 + n sh^2
    sums[2 * c] = a + count * sh;
    sums[2 * c + 1] = b + 2.0 * sh * a + count * sh * sh;
  }
}

// channel of the element at flat index `base` in [N, C, L]; 32-bit division when the tensor has < 2^32 elements
__device__ __forceinline__ int channel_of(int64_t base, int C, int L, bool small) {
  if (small) return (int)(((uint32_t)base / (uint32_t)L) % (uint32_t)C);
  return (int)((base / L) % C);
}

struct Tile4 {
  float a[4];
};
__device__ __forceinline__ Tile4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  Tile4 t; t.a[0] = v.x; t.a[1] = v.y; t.a[2] = v.z; t.a[3] = v.w;
  return t;
}
__device__ __forceinline__ void st4(float* p, const Tile4& t) {
  *reinterpret_cast<float4*>(p) = make_float4(t.a[0], t.a[1], t.a[2], t.a[3]);
}

// The gradient gz of a GEMM-produced pre-activation, stored the way its consumers want it: fp32, or -- `gzs` != null -- as
// three bf16 planes hi | mid | lo (gz = hi + mid + lo to 2^-24, plane p at gzs + p * total): the input-gradient and
// weight-gradient GEMMs copy those straight into LDS (s2f_pgemm_dx_split / s2f_spike_gemm_dw_*_split), their K loops convert
// nothing.  6 instead of 4 bytes per element written here, read back as bf16 instead of fp32 + a split per consuming tile.
__device__ __forceinline__ void store_gz(float* gz, unsigned short* gzs, int64_t total, int64_t base, const Tile4& o) {
  if (gzs != nullptr) {
    unsigned int h0, m0, l0, h1, m1, l1;
    s2f_split3x2(o.a[0], o.a[1], h0, m0, l0);
    s2f_split3x2(o.a[2], o.a[3], h1, m1, l1);
    *reinterpret_cast<uint2*>(gzs + base) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(gzs + total + base) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(gzs + 2 * total + base) = make_uint2(l0, l1);
  } else {
    *reinterpret_cast<float4*>(gz + base) = make_float4(o.a[0], o.a[1], o.a[2], o.a[3]);
  }
}

// spikes as bf16 (exact: s2f_spikes_to_bf16x4); p addresses uint16 storage
__device__ __forceinline__ void st4_bf16(float* p, int64_t base, const Tile4& t) {
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + base) = s2f_spikes_to_bf16x4(t.a[0], t.a[1], t.a[2], t.a[3]);
}

// Non-temporal loads for operands at their LAST use: bit 0 = z in the row-walking forward, bit 1 = z and the incoming gradients in
// the row-walking backward apply (the reduce pass before it reads them normally: this pass re-reads them), bit 2 = non-temporal
// mask stores, bit 3 / 4 = the single-pass forward / backward, bit 5 = the residual operand (below).  Same-box A/B of the C2 step through S2F_LIB (round 4, two
// alternations): 0: 37.23 / 37.22 ms, HBM-resident forward fraction 0.697 / 0.681;  1: 37.17 / 37.25, 0.728 / 0.713;
// 3: 37.04 / 37.05, 0.717 / 0.719;  7: 37.05 / 37.11, 0.701 / 0.698 (the mask words are read back by the backward: keep them cached).
// On another box, against 3 (37.63 / 37.61): 3 + 8: 37.55 / 37.56;  3 + 16: 37.53 / 37.46;  3 + 8 + 16: 37.46 / 37.40.
// The kernels' own durations barely move -- what improves is everybody else: a streamed-through operand no longer evicts what the
// next kernel is about to read (the gradient just written for the input- and weight-gradient products).  The same hint on the other
// last-use streams of the step (incoming gradients of the stand-alone neuron backward, the column matrix in col2im, the output
// gradient in the depthwise weight gradient) measured nothing: 37.16 / 37.07 without, 37.03 / 37.12 with all three.
// bit 5 = the residual operand of the forward kernels (the previous block's stream: its last reader): 27 -> 59 on a third box
// 37.28 / 37.38 / 37.28 -> 37.28 / 37.27 / 37.20, HBM-resident fraction 0.712 / 0.707 / 0.712 -> 0.729 / 0.734 / 0.732.
#ifndef S2F_BN_NT
#define S2F_BN_NT 59
#endif
__device__ __forceinline__ Tile4 ld4_nt(const float* p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  Tile4 t;
  t.a[0] = v.x, t.a[1] = v.y, t.a[2] = v.z, t.a[3] = v.w;
  return t;
}
// streaming (non-temporal) forms of the two stores
__device__ __forceinline__ void st4_nt(float* p, const Tile4& t) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v = {t.a[0], t.a[1], t.a[2], t.a[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
}
__device__ __forceinline__ void st4_bf16_nt(float* p, int64_t base, const Tile4& t) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const uint2 w = s2f_spikes_to_bf16x4(t.a[0], t.a[1], t.a[2], t.a[3]);
  u2 v = {w.x, w.y};
  __builtin_nontemporal_store(v, reinterpret_cast<u2*>(reinterpret_cast<unsigned short*>(p) + base));
}

// ANYL forms of the tile accesses: element by element, each guarded by the tensor's end (rows of L % 4 != 0 elements: a
// lane's four elements may lie in two channel rows, and only the tensor's start is known to be aligned)
template <bool ANYL>
__device__ __forceinline__ Tile4 ld4g(const float* p, int64_t base, int64_t total) {
  if (!ANYL) return ld4(p + base);
  Tile4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) t.a[j] = base + j < total ? p[base + j] : 0.f;
  return t;
}
template <bool ANYL>
__device__ __forceinline__ void st4g(float* p, int64_t base, int64_t total, const Tile4& t) {
  if (!ANYL) {
    st4(p + base, t);
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < total) p[base + j] = t.a[j];
}
template <bool ANYL>
__device__ __forceinline__ void st4g_bf16(float* p, int64_t base, int64_t total, const Tile4& t) {
  if (!ANYL) {
    st4_bf16(p, base, t);
    return;
  }
  unsigned short* q = reinterpret_cast<unsigned short*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < total) q[base + j] = (unsigned short)(__float_as_uint(t.a[j]) >> 16);      // exact for spikes
}

// u = ((z + b) - mean) * rstd * gamma + beta [+ res] ; optional LIF on u.   Flat 256-element tiles, L % 4 == 0.
// YB: the spikes y are written as bf16 (2 bytes / element).
// ANYL: any row length -- the channel, the validity and the accesses are per element (see ld4g).
template <bool LIF, bool HAS_V, bool YB, bool ANYL = false>
__global__ __launch_bounds__(kBlock) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                          const double* __restrict__ sums, float* __restrict__ stat,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ res,
                                                          float* __restrict__ u_out, const float* __restrict__ v_in,
                                                          float* __restrict__ y, float* __restrict__ v_out,
                                                          uint64_t* __restrict__ mask,
                                                          unsigned long long* __restrict__ stats, int64_t total, int C,
                                                          int L, double inv_count, float unbias, float momentum,
                                                          float eps, int training, float vth, float Df) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWaves;
  const int64_t ntiles = (total + 255) >> 8;
  uint32_t csum = 0, cnz = 0;
  // Prologue: every workgroup derives (mean, rstd) of ALL channels once into LDS (C/256 fp64 evaluations per thread);
  // the streaming loop then reads two LDS floats per lane.  Doing the fp64 arithmetic per tile instead kept the kernel
  // VALU-bound at 3.7 TB/s.
  extern __shared__ __attribute__((aligned(16))) float sstat[];      // [3][C]: mean, rstd, var
  // The operands of a wave's next tile are requested one iteration ahead -- the first ones before the prologue -- so the
  // statistics prologue and every tile's arithmetic run under a load instead of in front of one.
  Tile4 zn, rn, vn;
  auto request = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t base = tile * 256 + lane * 4;
    if (tile < ntiles && base < total) {
      zn = ld4g<ANYL>(z, base, total);
      if (res) rn = ld4g<ANYL>(res, base, total);
      if (LIF && HAS_V) vn = ld4g<ANYL>(v_in, base, total);
    }
  };
  request(wave0);
  for (int c = threadIdx.x; c < C; c += kBlock) {
    const ChanStat cs = chan_stat(sums, running_mean, running_var, c, inv_count, eps, training);
    sstat[c] = cs.mean;
    sstat[C + c] = cs.rstd;
    sstat[2 * C + c] = cs.var;
  }
  __syncthreads();
  const bool small = (total >> 32) == 0;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    const bool ok = base < total;                       // total % 4 == 0 -> whole float4 valid or not
    bool inr[4] = {false, false, false, false};
    const Tile4 zv = zn, rv = rn, vv = vn;
    request(tile + nwaves);
    if (ok) {
      // the lane owning the first element of channel c (row n = 0) publishes the statistics for the backward pass
      // and performs the running-statistics update (torch.nn.BatchNorm: momentum, unbiased variance)
      auto publish = [&](int c, float mean, float rstd, float g, float be) __attribute__((always_inline)) {
        stat[c] = mean;
        stat[C + c] = rstd;
        float rm = running_mean ? running_mean[c] : 0.f, rvv = running_var ? running_var[c] : 1.f;
        if (training && running_mean != nullptr) {
          rm = (1.f - momentum) * rm + momentum * mean;
          rvv = (1.f - momentum) * rvv + momentum * (sstat[2 * C + c] * unbias);
          running_mean[c] = rm;
          running_var[c] = rvv;
        }
        stat[2 * C + c] = be - rm * g / sqrtf(rvv + eps);      // BN(0) from the (updated) running statistics
        if (training && c == 0 && num_batches != nullptr) *num_batches += 1;
      };
      int c = channel_of(base, C, L, small);
      float b = bias ? bias[c] : 0.f;
      float mean = sstat[c], rstd = sstat[C + c], g = gamma[c], be = beta[c];
      if (!ANYL && base == (int64_t)c * L) publish(c, mean, rstd, g, be);
      Tile4 uo, yo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ANYL) {
          if (base + j >= total) {
            uo.a[j] = yo.a[j] = vo.a[j] = 0.f;
            continue;
          }
          c = channel_of(base + j, C, L, small);
          b = bias ? bias[c] : 0.f;
          mean = sstat[c], rstd = sstat[C + c], g = gamma[c], be = beta[c];
          if (base + j == (int64_t)c * L) publish(c, mean, rstd, g, be);
        }
        float u = ((zv.a[j] + b) - mean) * rstd * g + be;
        if (res) u += rv.a[j];
        uo.a[j] = u;
        if (LIF) {
          const float h = HAS_V ? (vv.a[j] + u) : u;
          float s, yy;
          s2f_lif_update(h, Df, 1.0f, vth, s, yy, vo.a[j], inr[j]);
          yo.a[j] = s / Df;
          csum += (uint32_t)s;
          cnz += ((uint32_t)s != 0);
        }
      }
      if (u_out) st4g<ANYL>(u_out, base, total, uo);
      if (LIF) {
        if (YB)
          st4g_bf16<ANYL>(y, base, total, yo);
        else
          st4g<ANYL>(y, base, total, yo);
        if (v_out) st4g<ANYL>(v_out, base, total, vo);
      }
    }
    if (LIF) {
      const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
      if (mask != nullptr && lane < 4) mask[tile * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
    }
  }
  if (LIF && stats != nullptr) {
    for (int o = 32; o > 0; o >>= 1) {
      csum += __shfl_xor(csum, o, 64);
      cnz += __shfl_xor(cnz, o, 64);
    }
    // one pair of atomics per workgroup (the prologue's LDS is free again after the streaming loop)
    __syncthreads();
    uint32_t* red = reinterpret_cast<uint32_t*>(sstat);
    if (lane == 0) {
      red[(threadIdx.x >> 6) * 2] = csum;
      red[(threadIdx.x >> 6) * 2 + 1] = cnz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long a = 0, b = 0;
      for (int w = 0; w < kWaves; ++w) {
        a += red[2 * w];
        b += red[2 * w + 1];
      }
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (a) atomicAdd(&slot[0], a);
      if (b) atomicAdd(&slot[1], b);
    }
  }
}

// gu = g_u + LIF-STE(g_y, g_v)   (any of the three may be absent)
__device__ __forceinline__ float form_gu(bool has_gu, float gu, bool has_gy, float gy, bool has_gv, float gv, bool m,
                                         float vth, float Df) {
  float r = has_gu ? gu : 0.f;
  if (has_gy || has_gv) {
    const float gvv = has_gv ? gv : 0.f;
    const float through = has_gy ? gy / Df : 0.f;
    const float lif = has_gv ? (m ? (gvv + (through - gvv * vth)) : gvv) : (m ? through : 0.f);
    r = has_gu ? r + lif : lif;
  }
  return r;
}

// grid (C, S) like bn_stats: sums[2c] = sum(gu), sums[2c+1] = sum(gu * xhat),  xhat = ((z + b) - mean) * rstd
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                               const float* __restrict__ stat,
                                                               const float* __restrict__ g_u, const float* __restrict__ g_y,
                                                               const float* __restrict__ g_v,
                                                               const uint64_t* __restrict__ mask,
                                                               double* __restrict__ sums, int N, int C, int L, int slice,
                                                               float vth, float Df) {
  const int c = blockIdx.x;
  const int l0 = blockIdx.y * slice, l1 = min(L, l0 + slice);
  const float b = bias ? bias[c] : 0.0f, mean = stat[c], rstd = stat[C + c];
  const int per_row = (l1 - l0) >> 2;
  const int total = N * per_row;
  float ps = 0.f, pq = 0.f;
  for (int it = threadIdx.x; it < total; it += kBlock * kUnroll) {
    int64_t e[kUnroll];
    bool ok[kUnroll];
    Tile4 zv[kUnroll], a[kUnroll], bb[kUnroll], cc[kUnroll];
    uint64_t mw[kUnroll][4];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {                 // issue every load of the group first
      const int id = it + u * kBlock;
      ok[u] = id < total;
      const int n = ok[u] ? id / per_row : 0, q = ok[u] ? id - n * per_row : 0;
      e[u] = ((int64_t)n * C + c) * L + l0 + q * 4;
      zv[u] = ld4(z + e[u]);
      if (g_u) a[u] = ld4(g_u + e[u]);
      if (g_y) bb[u] = ld4(g_y + e[u]);
      if (g_v) cc[u] = ld4(g_v + e[u]);
      if (mask) {
        const int64_t tile = e[u] >> 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) mw[u][j] = mask[tile * 4 + j];
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (!ok[u]) continue;
      const int ln = (int)((e[u] & 255) >> 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool m = mask ? ((mw[u][j] >> ln) & 1ull) : false;
        const float gu = form_gu(g_u != nullptr, a[u].a[j], g_y != nullptr, bb[u].a[j], g_v != nullptr, cc[u].a[j], m, vth, Df);
        const float xhat = ((zv[u].a[j] + b) - mean) * rstd;
        ps += gu;
        pq += gu * xhat;
      }
    }
  }
  block_atomic_add2((double)ps, (double)pq, sums + 2 * c);
}

// any row length (see bn_stats_any_kernel): one element per thread and step; the in-range bit of flat element e is bit
// (e & 255) >> 2 of mask word (e >> 8) * 4 + (e & 3)
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_any_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                                   const float* __restrict__ stat, const float* __restrict__ g_u,
                                                                   const float* __restrict__ g_y, const float* __restrict__ g_v,
                                                                   const uint64_t* __restrict__ mask, double* __restrict__ sums,
                                                                   int N, int C, int L, int slice, float vth, float Df) {
  const int c = blockIdx.x;
  const int l0 = blockIdx.y * slice, l1 = min(L, l0 + slice);
  const float b = bias ? bias[c] : 0.0f, mean = stat[c], rstd = stat[C + c];
  const int len = l1 - l0, total = N * len;
  float ps = 0.f, pq = 0.f;
  for (int id = threadIdx.x; id < total; id += kBlock) {
    const int n = id / len, q = id - n * len;
    const int64_t e = ((int64_t)n * C + c) * L + l0 + q;
    const bool m = mask ? ((mask[(e >> 8) * 4 + (e & 3)] >> ((e & 255) >> 2)) & 1ull) : false;
    const float gu = form_gu(g_u != nullptr, g_u ? g_u[e] : 0.f, g_y != nullptr, g_y ? g_y[e] : 0.f, g_v != nullptr,
                             g_v ? g_v[e] : 0.f, m, vth, Df);
    const float xhat = ((z[e] + b) - mean) * rstd;
    ps += gu;
    pq += gu * xhat;
  }
  block_atomic_add2((double)ps, (double)pq, sums + 2 * c);
}

// train: gz = gamma * rstd * (gu - sum_gu/cnt - xhat * sum_gux/cnt) ;  eval: gz = gamma * rstd * gu ;  g_res = gu
template <bool ANYL>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              const float* __restrict__ stat,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ g_u, const float* __restrict__ g_y,
                                                              const float* __restrict__ g_v,
                                                              const uint64_t* __restrict__ mask,
                                                              const double* __restrict__ sums, float* __restrict__ gz,
                                                              float* __restrict__ g_res, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int64_t total, int C, int L,
                                                              double inv_count, int training, float vth, float Df,
                                                              unsigned short* __restrict__ gzs) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWaves;
  const int64_t ntiles = (total + 255) >> 8;
  extern __shared__ __attribute__((aligned(16))) float sstat[];      // [2][C]: mean(gu), mean(gu * xhat)
  Tile4 zn, an, bn, cn;                       // next tile's operands, requested one iteration ahead (see bn_apply_kernel)
  uint64_t mn[4] = {0, 0, 0, 0};
  auto request = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t base = tile * 256 + lane * 4;
    if (tile < ntiles && base < total) {
      zn = ld4g<ANYL>(z, base, total);
      if (g_u) an = ld4g<ANYL>(g_u, base, total);
      if (g_y) bn = ld4g<ANYL>(g_y, base, total);
      if (g_v) cn = ld4g<ANYL>(g_v, base, total);
      if (mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mn[j] = mask[tile * 4 + j];
      }
    }
  };
  request(wave0);
  for (int c = threadIdx.x; c < C; c += kBlock) {
    sstat[c] = training ? (float)(sums[2 * c] * inv_count) : 0.f;
    sstat[C + c] = training ? (float)(sums[2 * c + 1] * inv_count) : 0.f;
  }
  __syncthreads();
  const bool small = (total >> 32) == 0;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    const Tile4 zv = zn, a = an, bb = bn, cc = cn;
    const uint64_t mw[4] = {mn[0], mn[1], mn[2], mn[3]};
    request(tile + nwaves);
    if (base >= total) continue;
    int c = channel_of(base, C, L, small);
    float b = bias ? bias[c] : 0.f, mean = stat[c], rstd = stat[C + c], g = gamma[c];
    float m1 = sstat[c], m2 = sstat[C + c];
    if (!ANYL && base == (int64_t)c * L) {          // dbeta = sum(gu), dgamma = sum(gu * xhat)
      dbeta[c] = (float)sums[2 * c];
      dgamma[c] = (float)sums[2 * c + 1];
    }
    Tile4 o, r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ANYL) {
        if (base + j >= total) {
          o.a[j] = r.a[j] = 0.f;
          continue;
        }
        c = channel_of(base + j, C, L, small);
        b = bias ? bias[c] : 0.f, mean = stat[c], rstd = stat[C + c], g = gamma[c];
        m1 = sstat[c], m2 = sstat[C + c];
        if (base + j == (int64_t)c * L) {
          dbeta[c] = (float)sums[2 * c];
          dgamma[c] = (float)sums[2 * c + 1];
        }
      }
      const bool m = mask ? ((mw[j] >> lane) & 1ull) : false;
      const float gu = form_gu(g_u != nullptr, a.a[j], g_y != nullptr, bb.a[j], g_v != nullptr, cc.a[j], m, vth, Df);
      const float xhat = ((zv.a[j] + b) - mean) * rstd;
      o.a[j] = (g * rstd) * ((gu - m1) - xhat * m2);
      r.a[j] = gu;
    }
    if (ANYL) {
      st4g<true>(gz, base, total, o);          // no bf16-plane form for odd rows (the pre-split protocol needs numel % 4 == 0)
      if (g_res) st4g<true>(g_res, base, total, r);
    } else {
      store_gz(gz, gzs, total, base, o);
      if (g_res) st4(g_res + base, r);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-walking forms of the three streaming kernels, for L % 4 == 0, L >= 256 and a power-of-two D (the ALIGNED variants: L %
// 256 == 0, every 256-element tile lies in ONE channel row; the others let a tile end one row and start the next).  The
// generic kernels above spend ~300 VALU instructions per tile on per-lane index arithmetic (two integer divisions for the
// channel, a float division per spike, 64-bit address checks) and ran at 3.1-3.9 TB/s moved where a pure read reaches
// 5.8-6.5 TB/s (tools/probe_bn_stream.py): they were VALU-bound, not HBM-bound.  Here a wave owns a CONTIGUOUS run of tiles;
// tile index, row, position in the row and channel are wave-uniform (readfirstlane on the wave id) and live in SGPRs: the
// channel advances by a compare instead of a division, the per-channel parameters and the four in-range mask words arrive
// through scalar loads, and a mask word is applied as a lane mask (inverse ballot -> v_cndmask) instead of 64-bit shifts.
// Same per-element expressions as the generic kernels, so the results are bit-identical.
__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

constexpr int kAhead = 2;                        // tiles of a wave in flight besides the one being worked on

// The wave-uniform walk over flat 256-element tiles of [N, C, L]: `row` = n * C + c of the tile's first element, `off` its
// position inside that row.  ALIGNED (L % 256 == 0, whole tiles): a tile lies in one row.  Otherwise (L % 4 == 0, L >= 256)
// the first `bnd` < 256 elements of a tile may end a row and the rest start the next one: the per-channel parameters of
// both channels are loaded (scalar) and each lane selects by its position -- at most one row boundary per tile.
struct RowWalk {
  uint32_t row, off, c;
};
__device__ __forceinline__ RowWalk walk_begin(uint32_t tile, uint32_t L, uint32_t C) {
  const uint64_t base = (uint64_t)tile * 256u;
  RowWalk w;
  w.row = (uint32_t)(base / L);
  w.off = (uint32_t)(base - (uint64_t)w.row * L);
  w.c = w.row % C;
  return w;
}

template <bool LIF, bool HAS_V, bool YB, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void bn_apply_rows_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const double* __restrict__ sums, float* __restrict__ stat,
    float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ num_batches,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ u_out,
    const float* __restrict__ v_in, float* __restrict__ y, float* __restrict__ v_out, uint64_t* __restrict__ mask,
    unsigned long long* __restrict__ stats, int64_t total, uint32_t ntiles, int C, uint32_t L, uint32_t chunk,
    double inv_count, float unbias, float momentum, float eps, int training, float vth, float Df) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * kWaves + (uint32_t)wave_id_uniform();
  uint32_t t = wave * chunk;
  const uint32_t t_end = min(ntiles, t + chunk);
  const uint32_t tail = (uint32_t)(total - (int64_t)(ntiles - 1) * 256);      // valid elements of the last tile (1..256)
  extern __shared__ __attribute__((aligned(16))) float sstat[];      // [3][C]: mean, rstd, var
  Tile4 zn[kAhead], rn[kAhead], vn[kAhead];
  auto request = [&](int slot, uint32_t tile) __attribute__((always_inline)) {
    if (tile < t_end && (ALIGNED || tile + 1 < ntiles || (uint32_t)lane * 4 < tail)) {
      const int64_t base = (int64_t)tile * 256 + lane * 4;
      zn[slot] = (S2F_BN_NT & 1) ? ld4_nt(z + base) : ld4(z + base);
      if (res) rn[slot] = (S2F_BN_NT & 32) ? ld4_nt(res + base) : ld4(res + base);
      if (LIF && HAS_V) vn[slot] = ld4(v_in + base);
    }
  };
#pragma unroll
  for (int i = 0; i < kAhead; ++i) request(i, t + i);
  for (int c = threadIdx.x; c < C; c += kBlock) {
    const ChanStat cs = chan_stat(sums, running_mean, running_var, c, inv_count, eps, training);
    sstat[c] = cs.mean;
    sstat[C + c] = cs.rstd;
    sstat[2 * C + c] = cs.var;
  }
  __syncthreads();
  RowWalk w = walk_begin(t, L, (uint32_t)C);
  struct Par {
    float b, mean, rstd, g, be;
  } p0, p1;
  auto params = [&](uint32_t c) __attribute__((always_inline)) {
    Par p;
    p.b = bias ? bias[c] : 0.f;
    p.mean = sstat[c];
    p.rstd = sstat[C + c];
    p.g = gamma[c];
    p.be = beta[c];
    return p;
  };
  if (t < t_end) p0 = params(w.c);
  p1 = p0;
  const float inv_d = 1.0f / Df;                          // exact: D is a power of two on this path
  const bool count = LIF && stats != nullptr;
  uint32_t csum = 0, cnz = 0;
  // first element of channel c in batch row 0: publish the statistics for the backward pass and update the running
  // statistics (torch.nn.BatchNorm: momentum, unbiased variance); executed by the one lane that owns that element
  auto publish = [&](uint32_t c, const Par& p) __attribute__((always_inline)) {
    stat[c] = p.mean;
    stat[C + c] = p.rstd;
    float rm = running_mean ? running_mean[c] : 0.f, rvv = running_var ? running_var[c] : 1.f;
    if (training && running_mean != nullptr) {
      rm = (1.f - momentum) * rm + momentum * p.mean;
      rvv = (1.f - momentum) * rvv + momentum * (sstat[2 * C + c] * unbias);
      running_mean[c] = rm;
      running_var[c] = rvv;
    }
    stat[2 * C + c] = p.be - rm * p.g / sqrtf(rvv + eps);
    if (training && c == 0 && num_batches != nullptr) *num_batches += 1;
  };
  // one tile: BN (+ residual) (+ neuron), stores, mask words; then the (row, channel) walk
  auto work = [&](uint32_t tile, const Tile4& zv, const Tile4& rv, const Tile4& vv) __attribute__((always_inline)) {
    const int64_t base = (int64_t)tile * 256 + lane * 4;
    const uint32_t bnd = L - w.off;                       // elements of this tile that still belong to `row`
    const bool split = !ALIGNED && bnd < 256u;
    const uint32_t c2 = w.c + 1 == (uint32_t)C ? 0u : w.c + 1;
    if (split) p1 = params(c2);
    if (w.off == 0 && w.row < (uint32_t)C && lane == 0) publish(w.c, p0);
    if (split && w.row + 1 < (uint32_t)C && (uint32_t)lane == (bnd >> 2)) publish(c2, p1);
    const bool first = !split || (uint32_t)lane * 4 < bnd;
    const bool ok = ALIGNED || tile + 1 < ntiles || (uint32_t)lane * 4 < tail;
    const float b = first ? p0.b : p1.b, mean = first ? p0.mean : p1.mean, rstd = first ? p0.rstd : p1.rstd,
                g = first ? p0.g : p1.g, be = first ? p0.be : p1.be;
    bool inr[4] = {false, false, false, false};
    if (ok) {
      Tile4 uo, yo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = ((zv.a[j] + b) - mean) * rstd * g + be;
        if (res) u += rv.a[j];
        uo.a[j] = u;
        if (LIF) {
          const float h = HAS_V ? (vv.a[j] + u) : u;
          float sp;
          s2f_lif_update(h, Df, inv_d, vth, sp, yo.a[j], vo.a[j], inr[j]);
          if (count) {
            csum += (uint32_t)sp;
            cnz += ((uint32_t)sp != 0);
          }
        }
      }
      if (u_out) st4(u_out + base, uo);
      if (LIF) {
        if (YB)
          st4_bf16(y, base, yo);
        else
          st4(y + base, yo);
        if (v_out) st4(v_out + base, vo);
      }
    }
    if (LIF) {
      const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
      if (mask != nullptr && lane < 4) {
        const uint64_t word = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
        if (S2F_BN_NT & 4)
          __builtin_nontemporal_store(word, mask + (int64_t)tile * 4 + lane);
        else
          mask[(int64_t)tile * 4 + lane] = word;
      }
    }
    w.off += 256u;
    if (w.off >= L) {
      w.off -= L;
      ++w.row;
      w.c = c2;
      p0 = split ? p1 : params(c2);
    }
  };
  for (; t < t_end; t += kAhead) {
    Tile4 zv[kAhead], rv[kAhead], vv[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) {
      zv[i] = zn[i];
      rv[i] = rn[i];
      vv[i] = vn[i];
    }
#pragma unroll
    for (int i = 0; i < kAhead; ++i) request(i, t + kAhead + i);
#pragma unroll
    for (int i = 0; i < kAhead; ++i)
      if (t + i < t_end) work(t + i, zv[i], rv[i], vv[i]);
  }
  if (count) {
    for (int o = 32; o > 0; o >>= 1) {
      csum += __shfl_xor(csum, o, 64);
      cnz += __shfl_xor(cnz, o, 64);
    }
    __syncthreads();
    uint32_t* red = reinterpret_cast<uint32_t*>(sstat);
    if (lane == 0) {
      red[(threadIdx.x >> 6) * 2] = csum;
      red[(threadIdx.x >> 6) * 2 + 1] = cnz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long a = 0, bq = 0;
      for (int wv = 0; wv < kWaves; ++wv) {
        a += red[2 * wv];
        bq += red[2 * wv + 1];
      }
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (a) atomicAdd(&slot[0], a);
      if (bq) atomicAdd(&slot[1], bq);
    }
  }
}

// gu of one element with the in-range bit delivered as a lane predicate
template <bool GU, bool GY, bool GV>
__device__ __forceinline__ float form_gu_rows(float gu, float gy, float gv, bool m, float vth, float inv_d) {
  float r = GU ? gu : 0.f;
  if (GY || GV) {
    const float gvv = GV ? gv : 0.f;
    const float through = GY ? gy * inv_d : 0.f;
    const float lif = GV ? (m ? (gvv + (through - gvv * vth)) : gvv) : (m ? through : 0.f);
    r = GU ? r + lif : lif;
  }
  return r;
}

// grid (C, S): block (c, s) reduces columns [s*slice, (s+1)*slice) of the rows n*C + c.  The columns of a row are walked as
// the FLAT 256-element tiles that overlap them (the in-range masks are stored per flat tile: a uniform tile index keeps the
// four mask words scalar loads); lanes outside [row start + l0, row start + l1) contribute nothing.  Wave w takes tiles
// w, w+4, ... of a row, kRowUnroll of them with every load issued before the first use.
constexpr int kRowUnroll = 4;
template <bool GU, bool GY, bool GV>
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce_rows_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ stat,
    const float* __restrict__ g_u, const float* __restrict__ g_y, const float* __restrict__ g_v,
    const uint64_t* __restrict__ mask, double* __restrict__ sums, int N, int C, int L, int slice, float vth, float Df,
    const float* __restrict__ g_y2) {          // g_y2?: a second reader's gradient of the spike map, summed here (s2f_bn_act_bwd_ports)
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = wave_id_uniform();
  const int l0 = blockIdx.y * slice, l1 = min(L, l0 + slice);
  const float b = bias ? bias[c] : 0.0f, mean = stat[c], rstd = stat[C + c];
  const float inv_d = 1.0f / Df;
  float ps = 0.f, pq = 0.f;
  for (int n = 0; n < N; ++n) {
    const int64_t e0 = ((int64_t)n * C + c) * L + l0, e1 = e0 + (l1 - l0);
    const uint32_t t_first = (uint32_t)(e0 >> 8), t_last = (uint32_t)((e1 - 1) >> 8);      // < 2^31 tiles (rows_ok)
    const int lo_first = (int)(e0 - (int64_t)t_first * 256);          // first valid element of the first tile
    const int hi_last = (int)(e1 - (int64_t)t_last * 256);            // end of the valid elements of the last tile
    for (uint32_t t0 = t_first + (uint32_t)wv; t0 <= t_last; t0 += kWaves * kRowUnroll) {
      Tile4 zv[kRowUnroll], a[kRowUnroll], bb[kRowUnroll], b2[kRowUnroll], cc[kRowUnroll];
      uint64_t mw[kRowUnroll][4];
      bool act[kRowUnroll];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        const uint32_t tile = t0 + u * kWaves;
        const bool live = tile <= t_last;                        // wave-uniform
        const int lo = tile == t_first ? lo_first : 0, hi = tile == t_last ? hi_last : 256;
        act[u] = live && lane * 4 >= lo && lane * 4 < hi;
        if (act[u]) {
          const int64_t base = (int64_t)tile * 256 + lane * 4;
          zv[u] = ld4(z + base);
          if (GU) a[u] = ld4(g_u + base);
          if (GY) bb[u] = ld4(g_y + base);
          if (GY && g_y2) b2[u] = ld4(g_y2 + base);
          if (GV) cc[u] = ld4(g_v + base);
        }
        if ((GY || GV) && live) {
#pragma unroll
          for (int j = 0; j < 4; ++j) mw[u][j] = mask[(int64_t)tile * 4 + j];
        }
      }
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        if (t0 + u * kWaves > t_last) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool m = (GY || GV) ? __builtin_amdgcn_inverse_ballot_w64(mw[u][j]) : false;
          const float gyv = GY ? ((g_y2 && act[u]) ? bb[u].a[j] + b2[u].a[j] : bb[u].a[j]) : 0.f;
          const float gu = form_gu_rows<GU, GY, GV>(GU ? a[u].a[j] : 0.f, gyv, GV ? cc[u].a[j] : 0.f, m, vth, inv_d);
          const float xhat = ((zv[u].a[j] + b) - mean) * rstd;
          if (act[u]) {
            ps += gu;
            pq += gu * xhat;
          }
        }
      }
    }
  }
  block_atomic_add2((double)ps, (double)pq, sums + 2 * c);
}

template <bool GU, bool GY, bool GV, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_rows_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ stat,
    const float* __restrict__ gamma, const float* __restrict__ g_u, const float* __restrict__ g_y,
    const float* __restrict__ g_v, const uint64_t* __restrict__ mask, const double* __restrict__ sums,
    float* __restrict__ gz, float* __restrict__ g_res, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int64_t total, uint32_t ntiles, int C, uint32_t L, uint32_t chunk, double inv_count, int training, float vth, float Df,
    unsigned short* __restrict__ gzs, const float* __restrict__ g_y2) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * kWaves + (uint32_t)wave_id_uniform();
  uint32_t t = wave * chunk;
  const uint32_t t_end = min(ntiles, t + chunk);
  const uint32_t tail = (uint32_t)(total - (int64_t)(ntiles - 1) * 256);
  Tile4 zn, an, bn, b2n, cn;
  uint64_t mn[4] = {0, 0, 0, 0};
  auto request = [&](uint32_t tile) __attribute__((always_inline)) {
    if (tile < t_end) {
      if (ALIGNED || tile + 1 < ntiles || (uint32_t)lane * 4 < tail) {
        const int64_t base = (int64_t)tile * 256 + lane * 4;
        zn = (S2F_BN_NT & 2) ? ld4_nt(z + base) : ld4(z + base);
        if (GU) an = (S2F_BN_NT & 2) ? ld4_nt(g_u + base) : ld4(g_u + base);
        if (GY) bn = (S2F_BN_NT & 2) ? ld4_nt(g_y + base) : ld4(g_y + base);
        if (GY && g_y2) b2n = (S2F_BN_NT & 2) ? ld4_nt(g_y2 + base) : ld4(g_y2 + base);
        if (GV) cn = ld4(g_v + base);
      }
      if (GY || GV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mn[j] = mask[(int64_t)tile * 4 + j];
      }
    }
  };
  request(t);
  RowWalk w = walk_begin(t, L, (uint32_t)C);
  struct Par {
    float b, mean, rstd, g, m1, m2;
  } p0, p1;
  auto params = [&](uint32_t c) __attribute__((always_inline)) {
    Par p;
    p.b = bias ? bias[c] : 0.f;
    p.mean = stat[c];
    p.rstd = stat[C + c];
    p.g = gamma[c];
    p.m1 = training ? (float)(sums[2 * c] * inv_count) : 0.f;
    p.m2 = training ? (float)(sums[2 * c + 1] * inv_count) : 0.f;
    return p;
  };
  if (t < t_end) p0 = params(w.c);
  p1 = p0;
  const float inv_d = 1.0f / Df;
  for (; t < t_end; ++t) {
    const int64_t base = (int64_t)t * 256 + lane * 4;
    const Tile4 zv = zn, a = an, cc = cn;
    Tile4 bb = bn;
    if (GY && g_y2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bb.a[j] += b2n.a[j];
    }
    const uint64_t mw[4] = {mn[0], mn[1], mn[2], mn[3]};
    request(t + 1);
    const uint32_t bnd = L - w.off;
    const bool split = !ALIGNED && bnd < 256u;
    const uint32_t c2 = w.c + 1 == (uint32_t)C ? 0u : w.c + 1;
    if (split) p1 = params(c2);
    // dbeta = sum(gu), dgamma = sum(gu * xhat): written by the lane that owns the channel's first element
    if (w.off == 0 && w.row < (uint32_t)C && lane == 0) {
      dbeta[w.c] = (float)sums[2 * w.c];
      dgamma[w.c] = (float)sums[2 * w.c + 1];
    }
    if (split && w.row + 1 < (uint32_t)C && (uint32_t)lane == (bnd >> 2)) {
      dbeta[c2] = (float)sums[2 * c2];
      dgamma[c2] = (float)sums[2 * c2 + 1];
    }
    const bool first = !split || (uint32_t)lane * 4 < bnd;
    const bool ok = ALIGNED || t + 1 < ntiles || (uint32_t)lane * 4 < tail;
    const float b = first ? p0.b : p1.b, mean = first ? p0.mean : p1.mean, rstd = first ? p0.rstd : p1.rstd,
                g = first ? p0.g : p1.g, m1 = first ? p0.m1 : p1.m1, m2 = first ? p0.m2 : p1.m2;
    if (ok) {
      Tile4 o, r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool m = (GY || GV) ? __builtin_amdgcn_inverse_ballot_w64(mw[j]) : false;
        const float gu = form_gu_rows<GU, GY, GV>(GU ? a.a[j] : 0.f, GY ? bb.a[j] : 0.f, GV ? cc.a[j] : 0.f, m, vth, inv_d);
        const float xhat = ((zv.a[j] + b) - mean) * rstd;
        o.a[j] = (g * rstd) * ((gu - m1) - xhat * m2);
        r.a[j] = gu;
      }
      store_gz(gz, gzs, total, base, o);
      if (g_res) st4(g_res + base, r);
    }
    w.off += 256u;
    if (w.off >= L) {
      w.off -= L;
      ++w.row;
      w.c = c2;
      p0 = split ? p1 : params(c2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-pass forms for small maps.  When one channel's N*L elements fit in the registers of one workgroup (4 waves x
// 8 tiles x 256 elements = 8 192; L % 256 == 0 so that the 256-element mask tiles do not straddle channels) the workgroup of
// channel c loads its slice ONCE, reduces the statistics through LDS and applies them from registers: z is read once instead
// of twice and one launch replaces two.  At the 32x32 / 64x64 stages of the path the two-kernel form is launch-bound
// (measured floors: bn_stats 4.3 us + bn_apply 5.5 us for <= 16 MB), which is where ~2/3 of the BatchNorm launches live.
// (round 4, same-box A/B of the C2 step: 4 tiles per wavefront on eight wavefronts 38.53 / 38.43 ms against 38.73 / 38.63 with 8 on four --
// two wavefronts per SIMD instead of one hide the load round trip of these launch-bound kernels; 2 on sixteen loses again, 39.09 / 38.99)
#ifndef S2F_BN_TPW
#define S2F_BN_TPW 4
#endif
constexpr int kTpw = S2F_BN_TPW;              // tiles per wave held in registers
// wavefronts of a single-pass workgroup: up to 16 (a channel of at most 64 tiles = 16 384 elements: round 4 -- the 64 x 32 stage of C3
// and the T = 8 batch of C4 are 16 384 elements per channel and ran the two-pass kernels for every BatchNorm of their deep stages)
#ifndef S2F_BN_MAX_TILES
#define S2F_BN_MAX_TILES 64
#endif
constexpr int kFusedWaves = S2F_BN_MAX_TILES / kTpw;
constexpr int kFusedBlock = 64 * kFusedWaves;

__device__ __forceinline__ void block_sum2(double& a, double& b, double* red, int nwaves) {
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();                            // red may still be read from a previous use
  if (lane == 0) {
    red[2 * w] = a;
    red[2 * w + 1] = b;
  }
  __syncthreads();
  a = 0, b = 0;
  for (int i = 0; i < nwaves; ++i) {          // every thread forms the same sum in the same order
    a += red[2 * i];
    b += red[2 * i + 1];
  }
}

// The SECOND BatchNorm of a train-mode BatchNorm o BatchNorm pair (every RepConv chain of the attention blocks closes with two:
// Sequential(RepConv(.. BN_b), BN_c), sdtv2.py:280-296, 304-306).  With xhat = (z - mean) r_b the first gives a = gamma_b xhat +
// beta_b, whose batch mean is beta_b and whose (biased) batch variance is gamma_b^2 var r_b^2 -- no second pass is needed to know
// them -- so  BN_c(BN_b(z)) = gamma_b gamma_c r_c xhat + beta_c,  r_c = 1 / sqrt(gamma_b^2 var r_b^2 + eps_c):  ONE kernel with a
// different scale, which also updates BN_c's running statistics with (beta_b, gamma_b^2 var r_b^2 unbias).  (The reference forms
// mean(a) numerically: beta_b + gamma_b mean(xhat), mean(xhat) ~ 1e-8.)
struct Bn2 {
  const float* gamma;            // gamma_c [C]; null = plain BatchNorm
  const float* beta;             // beta_c
  float* running_mean;           // BN_c's running statistics (or null)
  float* running_var;
  long long* num_batches;
  float eps, momentum;
};

// SHIFTED statistics of the single-pass kernels (round 6).  The sums are taken of (z + b) - pivot, pivot = the channel's first element:
// var = E[d^2] - E[d]^2 then cancels against (mean - pivot)^2 -- of the order of the variance itself -- instead of against mean^2.  The
// plain form E[x^2] - E[x]^2 with fp32 squares carries an absolute error of ~1e-7 mean^2: for a channel whose variance is far below
// its squared mean (tiny maps, near-constant channels: the 4 x 4 maps of the plumbing configuration) that is a PER-CENT error in
// rstd -- invisible forward (x - mean is tiny there) but multiplied into every input gradient of the channel (found by holding the
// tiny-config gradients to 10x the measured reference-vs-oracle gap: 2-7 % on dcn.input_proj at C1_64).  ATen's CPU BatchNorm
// (the reference) computes the variance in two passes; a constant channel now gives var == 0 exactly here too.
#ifndef S2F_BN_SHIFTED
#define S2F_BN_SHIFTED 1          // 0: the plain E[x^2] - E[x]^2 form (A/B builds only)
#endif
#define S2F_BN_PIVOT const float piv = S2F_BN_SHIFTED ? z[(int64_t)c * L] + b : 0.f

template <bool LIF, bool HAS_V, bool YB, bool DOUBLE = false>
__global__ __launch_bounds__(kFusedBlock) void bn_fused_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ stat, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ u_out, const float* __restrict__ v_in,
    float* __restrict__ y, float* __restrict__ v_out, uint64_t* __restrict__ mask, unsigned long long* __restrict__ stats,
    int N, int C, int L, double inv_count, float unbias, float momentum, float eps, float vth, float Df, Bn2 bn2) {
  __shared__ double red[2 * kFusedWaves];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int tpr = L >> 8, tiles = N * tpr;                  // tiles per row, tiles of this channel
  const float b = bias ? bias[c] : 0.f;
  Tile4 zv[kTpw];
  int64_t base[kTpw];
  bool ok[kTpw];
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {                          // all loads in flight before the first use
    const int t = w + i * nwaves;
    ok[i] = t < tiles;
    const int n = ok[i] ? t / tpr : 0, q = ok[i] ? t - n * tpr : 0;
    base[i] = ((int64_t)n * C + c) * L + q * 256 + lane * 4;
    zv[i] = (S2F_BN_NT & 8) ? ld4_nt(z + base[i]) : ld4(z + base[i]);
  }
  // the residual is independent of the statistics: its loads ride with z's instead of forming a second dependent round trip
  // after the block reduction (~1.5 us of a 9 us kernel)
  Tile4 rvp[kTpw];
  if (res) {
#pragma unroll
    for (int i = 0; i < kTpw; ++i) rvp[i] = (S2F_BN_NT & 32) ? ld4_nt(res + base[i]) : ld4(res + base[i]);
  }
  S2F_BN_PIVOT;
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {
    if (!ok[i]) continue;
    const float a0 = (zv[i].a[0] + b) - piv, a1 = (zv[i].a[1] + b) - piv, a2 = (zv[i].a[2] + b) - piv, a3 = (zv[i].a[3] + b) - piv;
    ps += (a0 + a1) + (a2 + a3);
    pq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  double s1 = (double)ps, s2 = (double)pq;
  block_sum2(s1, s2, red, nwaves);
  const double dm = s1 * inv_count;                       // mean of (z + b) - pivot
  double vd = s2 * inv_count - dm * dm;
  if (vd < 0) vd = 0;
  const double md = (double)piv + dm;
  const float mean = (float)md, var = (float)vd;
  const float rstd = 1.0f / sqrtf(var + eps);
  float g = gamma[c], be = beta[c];
  if (threadIdx.x == 0) {
    stat[c] = mean;
    stat[C + c] = rstd;
    float rm = 0.f, rv = 1.f;
    if (running_mean != nullptr) {
      rm = (1.f - momentum) * running_mean[c] + momentum * mean;
      rv = (1.f - momentum) * running_var[c] + momentum * (var * unbias);
      running_mean[c] = rm;
      running_var[c] = rv;
    }
    stat[2 * C + c] = be - rm * g / sqrtf(rv + eps);          // BN(0) from the (updated) running statistics
    if (c == 0 && num_batches != nullptr) *num_batches += 1;
  }
  if constexpr (DOUBLE) {
    const float var_a = (g * g) * (var * (rstd * rstd));      // biased batch variance of a = gamma_b xhat + beta_b
    const float r_c = 1.0f / sqrtf(var_a + bn2.eps);
    if (threadIdx.x == 0) {
      stat[3 * C + c] = r_c;
      if (bn2.running_mean != nullptr) {
        bn2.running_mean[c] = (1.f - bn2.momentum) * bn2.running_mean[c] + bn2.momentum * be;
        bn2.running_var[c] = (1.f - bn2.momentum) * bn2.running_var[c] + bn2.momentum * (var_a * unbias);
      }
      if (c == 0 && bn2.num_batches != nullptr) *bn2.num_batches += 1;
    }
    g = (g * bn2.gamma[c]) * r_c;
    be = bn2.beta[c];
  }
  uint32_t csum = 0, cnz = 0;
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {
    bool inr[4] = {false, false, false, false};
    if (ok[i]) {                                            // wave-uniform
      Tile4 vv, uo, yo, vo;
      if (LIF && HAS_V) vv = ld4(v_in + base[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = ((zv[i].a[j] + b) - mean) * rstd * g + be;
        if (res) u += rvp[i].a[j];
        uo.a[j] = u;
        if (LIF) {
          const float h = HAS_V ? (vv.a[j] + u) : u;
          float sp, yy;
          s2f_lif_update(h, Df, 1.0f, vth, sp, yy, vo.a[j], inr[j]);
          yo.a[j] = sp / Df;
          csum += (uint32_t)sp;
          cnz += ((uint32_t)sp != 0);
        }
      }
      if (u_out) st4(u_out + base[i], uo);
      if (LIF) {
        if (YB)
          st4_bf16(y, base[i], yo);
        else
          st4(y + base[i], yo);
        if (v_out) st4(v_out + base[i], vo);
        const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
        const int64_t tile = base[i] >> 8;
        if (mask != nullptr && lane < 4) mask[tile * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
      }
    }
  }
  if (LIF && stats != nullptr) {
    double a = (double)csum, bq = (double)cnz;              // < 2^53: exact
    block_sum2(a, bq, red, nwaves);
    if (threadIdx.x == 0) {
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (a > 0) atomicAdd(&slot[0], (unsigned long long)a);
      if (bq > 0) atomicAdd(&slot[1], (unsigned long long)bq);
    }
  }
}

// DOUBLE (the backward of the pair above): with S1 = sum gu, S2 = sum gu xhat,
//   dbeta_c = S1 ; dgamma_c = gamma_b r_c S2 ; dbeta_b = 0 ; dgamma_b = gamma_c eps_c r_c^3 S2 ;
//   gz = gamma_b r_b gamma_c r_c (gu - S1/n - kappa xhat S2/n),  kappa = r_c^2 (gamma_b^2 + eps_c)
// (BN_c's backward into a, then BN_b's: sum xhat = 0, sum xhat^2 = n var r_b^2 = n (1 / r_c^2 - eps_c) / gamma_b^2).
struct Bn2Bwd {
  const float* gamma;            // gamma_c; null = plain
  float* dgamma;                 // d gamma_c, d beta_c
  float* dbeta;
  float eps;
};

template <bool GU, bool GY, bool GV, bool DOUBLE = false>
__global__ __launch_bounds__(kFusedBlock) void bn_fused_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ stat,
    const float* __restrict__ gamma, const float* __restrict__ g_u, const float* __restrict__ g_y,
    const float* __restrict__ g_v, const uint64_t* __restrict__ mask, float* __restrict__ gz, float* __restrict__ g_res,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int L, double inv_count, float vth, float Df,
    unsigned short* __restrict__ gzs, Bn2Bwd bn2, const float* __restrict__ g_y2) {
  __shared__ double red[2 * kFusedWaves];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int tpr = L >> 8, tiles = N * tpr;
  const float b = bias ? bias[c] : 0.f, mean = stat[c], rstd = stat[C + c], g = gamma[c];
  // per tile: xhat (from z) and gu, both kept in registers for the second phase; the incoming gradients and the mask word
  // of a tile are consumed as soon as they arrive (lane j < 4 loads word j, the others get it by a lane read)
  Tile4 xh[kTpw], gu[kTpw];
  int tix[kTpw];
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {
    const int t = w + i * nwaves;
    tix[i] = t < tiles ? t : -1;
    const int tt = t < tiles ? t : 0;
    const int n = tt / tpr, q = tt - n * tpr;
    const int64_t base = ((int64_t)n * C + c) * L + q * 256 + lane * 4;
    xh[i] = (S2F_BN_NT & 16) ? ld4_nt(z + base) : ld4(z + base);
    if (GU) gu[i] = (S2F_BN_NT & 16) ? ld4_nt(g_u + base) : ld4(g_u + base);
    Tile4 bb, cc;
    uint64_t word = 0;
    if (GY) bb = (S2F_BN_NT & 16) ? ld4_nt(g_y + base) : ld4(g_y + base);
    if (GY && g_y2) {
      const Tile4 t2 = (S2F_BN_NT & 16) ? ld4_nt(g_y2 + base) : ld4(g_y2 + base);
#pragma unroll
      for (int j = 0; j < 4; ++j) bb.a[j] += t2.a[j];
    }
    if (GV) cc = ld4(g_v + base);
    if (GY || GV) word = mask[(base >> 8) * 4 + (lane & 3)];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool m = false;
      if (GY || GV) {
        const uint32_t lo = __shfl((uint32_t)word, j, 64), hi = __shfl((uint32_t)(word >> 32), j, 64);
        m = ((lane < 32 ? lo >> lane : hi >> (lane - 32)) & 1u) != 0;
      }
      gu[i].a[j] = form_gu(GU, GU ? gu[i].a[j] : 0.f, GY, GY ? bb.a[j] : 0.f, GV, GV ? cc.a[j] : 0.f, m, vth, Df);
      xh[i].a[j] = ((xh[i].a[j] + b) - mean) * rstd;
    }
  }
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {
    if (tix[i] < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ps += gu[i].a[j];
      pq += gu[i].a[j] * xh[i].a[j];
    }
  }
  double s1 = (double)ps, s2 = (double)pq;
  block_sum2(s1, s2, red, nwaves);
  float scale = g * rstd;
  float m1 = (float)(s1 * inv_count), m2 = (float)(s2 * inv_count);
  if constexpr (DOUBLE) {
    const float r_c = stat[3 * C + c], g_c = bn2.gamma[c];
    if (threadIdx.x == 0) {
      bn2.dbeta[c] = (float)s1;
      bn2.dgamma[c] = (g * r_c) * (float)s2;
      dbeta[c] = 0.f;
      dgamma[c] = (g_c * bn2.eps) * (r_c * r_c * r_c) * (float)s2;
    }
    scale = (g * rstd) * (g_c * r_c);
    m2 *= (r_c * r_c) * (g * g + bn2.eps);
  } else if (threadIdx.x == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
#pragma unroll
  for (int i = 0; i < kTpw; ++i) {
    if (tix[i] < 0) continue;
    const int n = tix[i] / tpr, q = tix[i] - n * tpr;
    const int64_t base = ((int64_t)n * C + c) * L + q * 256 + lane * 4;
    Tile4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.a[j] = scale * ((gu[i].a[j] - m1) - xh[i].a[j] * m2);
    store_gz(gz, gzs, (int64_t)N * C * L, base, o);
    if (g_res) st4(g_res + base, gu[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-pass forms for SHORT rows (round 4): the decoder's 100-token maps [T*B, C, 100] (and any L % 4 == 0 map with N * L <= 2 048
// whose rows are not whole 256-element tiles).  The two-kernel forms cost four launches of 5-7 us per BatchNorm and step there
// (statistics or partials-finalize + apply forward, reduce + apply backward; ~50 such BatchNorms per C2 step); round 3's one-launch
// attempt was tied to the flat 256-element mask tiles -- 64 channels per workgroup, FOUR workgroups for 256 channels -- and lost.
// The in-range mask of these maps is written by this forward and read by this backward only, so it gets its own layout: ONE
// WAVEFRONT PER CHANNEL holds the channel's N * L / 4 four-element groups (group g = n * (L / 4) + q of row n at lane g & 63,
// round g >> 6, at most kSmallIters rounds), and mask word [(c * kSmallIters + round) * 4 + j] is the ballot of component j of
// that round.  Statistics by shuffles inside the wavefront (fp64), everything else as the single-pass kernels above.
constexpr int kSmallIters = 8;          // rounds of 64 four-element groups per channel: N * L <= 2 048

inline bool small_rows_ok(int64_t N, int64_t C, int64_t L) {
  return (L & 3) == 0 && (L & 255) != 0 && N * L <= 64 * 4 * kSmallIters && N * L >= 8 && C >= 32;
}

template <bool LIF, bool HAS_V, bool YB>
__global__ __launch_bounds__(64) void bn_small_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ stat, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ u_out, const float* __restrict__ v_in,
    float* __restrict__ y, float* __restrict__ v_out, uint64_t* __restrict__ mask, unsigned long long* __restrict__ stats,
    int N, int C, int L, double inv_count, float unbias, float momentum, float eps, float vth, float Df) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const int per_row = L >> 2, groups = N * per_row;
  const float b = bias ? bias[c] : 0.f;
  Tile4 zv[kSmallIters], rvp[kSmallIters];
  int64_t base[kSmallIters];
  bool ok[kSmallIters];
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {                   // all loads in flight before the first use
    const int g = i * 64 + lane;
    ok[i] = g < groups;
    const int n = ok[i] ? g / per_row : 0, q = ok[i] ? g - n * per_row : 0;
    base[i] = ((int64_t)n * C + c) * L + q * 4;
    if (i * 64 < groups) {                                  // wave-uniform: rounds past the channel's groups issue nothing
      zv[i] = ld4(z + base[i]);
      if (res) rvp[i] = ld4(res + base[i]);
    }
  }
  S2F_BN_PIVOT;
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {
    if (!ok[i]) continue;
    const float a0 = (zv[i].a[0] + b) - piv, a1 = (zv[i].a[1] + b) - piv, a2 = (zv[i].a[2] + b) - piv, a3 = (zv[i].a[3] + b) - piv;
    ps += (a0 + a1) + (a2 + a3);
    pq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const double s1 = wave_sum_f64((double)ps), s2 = wave_sum_f64((double)pq);
  const double dm = s1 * inv_count;                       // mean of (z + b) - pivot
  double vd = s2 * inv_count - dm * dm;
  if (vd < 0) vd = 0;
  const double md = (double)piv + dm;
  const float mean = (float)md, var = (float)vd;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g_ = gamma[c], be = beta[c];
  if (lane == 0) {
    stat[c] = mean;
    stat[C + c] = rstd;
    float rm = 0.f, rv = 1.f;
    if (running_mean != nullptr) {
      rm = (1.f - momentum) * running_mean[c] + momentum * mean;
      rv = (1.f - momentum) * running_var[c] + momentum * (var * unbias);
      running_mean[c] = rm;
      running_var[c] = rv;
    }
    stat[2 * C + c] = be - rm * g_ / sqrtf(rv + eps);         // BN(0) from the (updated) running statistics
    if (c == 0 && num_batches != nullptr) *num_batches += 1;
  }
  uint32_t csum = 0, cnz = 0;
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {
    if (i * 64 >= groups) break;                            // wave-uniform
    bool inr[4] = {false, false, false, false};
    if (ok[i]) {
      Tile4 vv, uo, yo, vo;
      if (LIF && HAS_V) vv = ld4(v_in + base[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = ((zv[i].a[j] + b) - mean) * rstd * g_ + be;
        if (res) u += rvp[i].a[j];
        uo.a[j] = u;
        if (LIF) {
          const float h = HAS_V ? (vv.a[j] + u) : u;
          float sp, yy;
          s2f_lif_update(h, Df, 1.0f, vth, sp, yy, vo.a[j], inr[j]);
          yo.a[j] = sp / Df;
          csum += (uint32_t)sp;
          cnz += ((uint32_t)sp != 0);
        }
      }
      if (u_out) st4(u_out + base[i], uo);
      if (LIF) {
        if (YB)
          st4_bf16(y, base[i], yo);
        else
          st4(y + base[i], yo);
        if (v_out) st4(v_out + base[i], vo);
      }
    }
    if (LIF && mask != nullptr) {
      const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
      if (lane < 4) mask[((int64_t)c * kSmallIters + i) * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
    }
  }
  if (LIF && stats != nullptr) {
    for (int o = 32; o > 0; o >>= 1) {
      csum += __shfl_xor(csum, o, 64);
      cnz += __shfl_xor(cnz, o, 64);
    }
    if (lane == 0) {
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
      if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
    }
  }
}

template <bool GU, bool GY, bool GV>
__global__ __launch_bounds__(64) void bn_small_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ stat,
    const float* __restrict__ gamma, const float* __restrict__ g_u, const float* __restrict__ g_y,
    const float* __restrict__ g_v, const uint64_t* __restrict__ mask, float* __restrict__ gz, float* __restrict__ g_res,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int L, double inv_count, float vth, float Df) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const int per_row = L >> 2, groups = N * per_row;
  const float b = bias ? bias[c] : 0.f, mean = stat[c], rstd = stat[C + c], g_ = gamma[c];
  Tile4 xh[kSmallIters], gu[kSmallIters];
  int64_t base[kSmallIters];
  bool ok[kSmallIters];
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {
    const int g = i * 64 + lane;
    ok[i] = g < groups;
    const int n = ok[i] ? g / per_row : 0, q = ok[i] ? g - n * per_row : 0;
    base[i] = ((int64_t)n * C + c) * L + q * 4;
    if (i * 64 >= groups) continue;                         // wave-uniform
    xh[i] = ld4(z + base[i]);
    if (GU) gu[i] = ld4(g_u + base[i]);
    Tile4 bb, cc;
    if (GY) bb = ld4(g_y + base[i]);
    if (GV) cc = ld4(g_v + base[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool m = false;
      if (GY || GV) m = (mask[((int64_t)c * kSmallIters + i) * 4 + j] >> lane) & 1ull;
      gu[i].a[j] = form_gu(GU, GU ? gu[i].a[j] : 0.f, GY, GY ? bb.a[j] : 0.f, GV, GV ? cc.a[j] : 0.f, m, vth, Df);
      xh[i].a[j] = ((xh[i].a[j] + b) - mean) * rstd;
    }
  }
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {
    if (!ok[i]) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ps += gu[i].a[j];
      pq += gu[i].a[j] * xh[i].a[j];
    }
  }
  const double s1 = wave_sum_f64((double)ps), s2 = wave_sum_f64((double)pq);
  if (lane == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
  const float m1 = (float)(s1 * inv_count), m2 = (float)(s2 * inv_count);
#pragma unroll
  for (int i = 0; i < kSmallIters; ++i) {
    if (!ok[i]) continue;
    Tile4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.a[j] = (g_ * rstd) * ((gu[i].a[j] - m1) - xh[i].a[j] * m2);
    st4(gz + base[i], o);
    if (g_res) st4(g_res + base[i], gu[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-pass forms for rows that are NOT whole 256-element tiles but too long for one wavefront (round 5): C5's 50 x 84 maps,
// [4, C, 4 200] -- 16 800 elements per channel, 2.5 % past the 64 whole tiles of the single-pass kernels above and not tile-aligned, so
// every BatchNorm of that stage ran statistics / finalize / apply forward and reduce + apply backward (20 + 21 us against 8 + 13 us at
// [4, 256, 4 096], tools/probe_bn_stream.py; ~300 launches per C5 step).  The short-row idea on up to sixteen wavefronts: the
// channel's N * L / 4 four-element groups are dealt out 64 per wavefront and round -- group g = (round * W + wave) * 64 + lane, at
// most kMidIters rounds -- and the in-range mask keeps a per-channel layout of its own (written by this forward, read by this
// backward only): word [((c * kMidIters + round) * kMidWaves + wave) * 4 + j] is the ballot of component j.
constexpr int kMidIters = 5;
constexpr int kMidWaves = 16;

inline bool mid_rows_ok(int64_t N, int64_t C, int64_t L) {
  static const char* off = getenv("S2F_BN_MID");          // A/B switch: "0" keeps these shapes on the row-walking kernels
  if (off && off[0] == '0') return false;
  return (L & 3) == 0 && (L & 255) != 0 && N * L > 64 * 4 * kSmallIters && N * L <= (int64_t)64 * 4 * kMidIters * kMidWaves && C >= 32;
}
inline int mid_rows_threads(int64_t N, int64_t L) {
  const int64_t groups = N * (L >> 2);
  return 64 * (int)((groups + 64 * kMidIters - 1) / (64 * kMidIters));
}

template <bool LIF, bool HAS_V, bool YB>
__global__ __launch_bounds__(64 * kMidWaves) void bn_mid_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ stat, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ u_out, const float* __restrict__ v_in,
    float* __restrict__ y, float* __restrict__ v_out, uint64_t* __restrict__ mask, unsigned long long* __restrict__ stats,
    int N, int C, int L, double inv_count, float unbias, float momentum, float eps, float vth, float Df) {
  __shared__ double red[2 * kMidWaves];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int per_row = L >> 2, groups = N * per_row;
  const float b = bias ? bias[c] : 0.f;
  Tile4 zv[kMidIters], rvp[kMidIters];
  int64_t base[kMidIters];
  bool ok[kMidIters];
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {                     // all loads in flight before the first use
    const int g0 = (i * W + w) * 64, g = g0 + lane;
    ok[i] = g < groups;
    const int n = ok[i] ? g / per_row : 0, q = ok[i] ? g - n * per_row : 0;
    base[i] = ((int64_t)n * C + c) * L + q * 4;
    if (g0 < groups) {                                      // wave-uniform
      zv[i] = ld4(z + base[i]);
      if (res) rvp[i] = ld4(res + base[i]);
    }
  }
  S2F_BN_PIVOT;
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {
    if (!ok[i]) continue;
    const float a0 = (zv[i].a[0] + b) - piv, a1 = (zv[i].a[1] + b) - piv, a2 = (zv[i].a[2] + b) - piv, a3 = (zv[i].a[3] + b) - piv;
    ps += (a0 + a1) + (a2 + a3);
    pq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  double s1 = (double)ps, s2 = (double)pq;
  block_sum2(s1, s2, red, W);
  const double dm = s1 * inv_count;                       // mean of (z + b) - pivot
  double vd = s2 * inv_count - dm * dm;
  if (vd < 0) vd = 0;
  const double md = (double)piv + dm;
  const float mean = (float)md, var = (float)vd;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float g_ = gamma[c], be = beta[c];
  if (threadIdx.x == 0) {
    stat[c] = mean;
    stat[C + c] = rstd;
    float rm = 0.f, rv = 1.f;
    if (running_mean != nullptr) {
      rm = (1.f - momentum) * running_mean[c] + momentum * mean;
      rv = (1.f - momentum) * running_var[c] + momentum * (var * unbias);
      running_mean[c] = rm;
      running_var[c] = rv;
    }
    stat[2 * C + c] = be - rm * g_ / sqrtf(rv + eps);         // BN(0) from the (updated) running statistics
    if (c == 0 && num_batches != nullptr) *num_batches += 1;
  }
  uint32_t csum = 0, cnz = 0;
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {
    if ((i * W + w) * 64 >= groups) break;                  // wave-uniform
    bool inr[4] = {false, false, false, false};
    if (ok[i]) {
      Tile4 vv, uo, yo, vo;
      if (LIF && HAS_V) vv = ld4(v_in + base[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = ((zv[i].a[j] + b) - mean) * rstd * g_ + be;
        if (res) u += rvp[i].a[j];
        uo.a[j] = u;
        if (LIF) {
          const float h = HAS_V ? (vv.a[j] + u) : u;
          float sp, yy;
          s2f_lif_update(h, Df, 1.0f, vth, sp, yy, vo.a[j], inr[j]);
          yo.a[j] = sp / Df;
          csum += (uint32_t)sp;
          cnz += ((uint32_t)sp != 0);
        }
      }
      if (u_out) st4(u_out + base[i], uo);
      if (LIF) {
        if (YB)
          st4_bf16(y, base[i], yo);
        else
          st4(y + base[i], yo);
        if (v_out) st4(v_out + base[i], vo);
      }
    }
    if (LIF && mask != nullptr) {
      const uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
      if (lane < 4)
        mask[(((int64_t)c * kMidIters + i) * kMidWaves + w) * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
    }
  }
  if (LIF && stats != nullptr) {
    for (int o = 32; o > 0; o >>= 1) {
      csum += __shfl_xor(csum, o, 64);
      cnz += __shfl_xor(cnz, o, 64);
    }
    if (lane == 0) {
      unsigned long long* slot = stats + 2 * ((blockIdx.x * kMidWaves + w) % S2F_STAT_SLOTS);
      if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
      if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
    }
  }
}

template <bool GU, bool GY, bool GV>
__global__ __launch_bounds__(64 * kMidWaves) void bn_mid_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ bias, const float* __restrict__ stat,
    const float* __restrict__ gamma, const float* __restrict__ g_u, const float* __restrict__ g_y,
    const float* __restrict__ g_v, const uint64_t* __restrict__ mask, float* __restrict__ gz, float* __restrict__ g_res,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int L, double inv_count, float vth, float Df) {
  __shared__ double red[2 * kMidWaves];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int per_row = L >> 2, groups = N * per_row;
  const float b = bias ? bias[c] : 0.f, mean = stat[c], rstd = stat[C + c], g_ = gamma[c];
  Tile4 xh[kMidIters], gu[kMidIters];
  int64_t base[kMidIters];
  bool ok[kMidIters];
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {
    const int g0 = (i * W + w) * 64, g = g0 + lane;
    ok[i] = g < groups;
    const int n = ok[i] ? g / per_row : 0, q = ok[i] ? g - n * per_row : 0;
    base[i] = ((int64_t)n * C + c) * L + q * 4;
    if (g0 >= groups) continue;                             // wave-uniform
    xh[i] = ld4(z + base[i]);
    if (GU) gu[i] = ld4(g_u + base[i]);
    Tile4 bb, cc;
    if (GY) bb = ld4(g_y + base[i]);
    if (GV) cc = ld4(g_v + base[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bool m = false;
      if (GY || GV) m = (mask[(((int64_t)c * kMidIters + i) * kMidWaves + w) * 4 + j] >> lane) & 1ull;
      gu[i].a[j] = form_gu(GU, GU ? gu[i].a[j] : 0.f, GY, GY ? bb.a[j] : 0.f, GV, GV ? cc.a[j] : 0.f, m, vth, Df);
      xh[i].a[j] = ((xh[i].a[j] + b) - mean) * rstd;
    }
  }
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {
    if (!ok[i]) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ps += gu[i].a[j];
      pq += gu[i].a[j] * xh[i].a[j];
    }
  }
  double s1 = (double)ps, s2 = (double)pq;
  block_sum2(s1, s2, red, W);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
  const float m1 = (float)(s1 * inv_count), m2 = (float)(s2 * inv_count);
#pragma unroll
  for (int i = 0; i < kMidIters; ++i) {
    if (!ok[i]) continue;
    Tile4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.a[j] = (g_ * rstd) * ((gu[i].a[j] - m1) - xh[i].a[j] * m2);
    st4(gz + base[i], o);
    if (g_res) st4(g_res + base[i], gu[i]);
  }
}

// single-pass eligibility: whole 256-element tiles per row, one channel's tiles fit kWaves x kTpw, and enough channels
// to occupy the chip with one workgroup per channel
inline bool single_pass_ok(int64_t N, int64_t C, int64_t L) {
  if ((L & 255) != 0) return false;
  const int64_t tiles = N * (L >> 8);
  return tiles >= 4 && tiles <= kFusedWaves * kTpw && C >= 64 && N * C * L < ((int64_t)1 << 31);
}
inline int single_pass_threads(int64_t N, int64_t L) {
  const int tiles = (int)(N * (L >> 8));
  return 64 * ((tiles + kTpw - 1) / kTpw);
}

inline int pick_slices(int C, int L, int& slice) {
  int S = 1;
  while ((int64_t)C * S < 1024 && L / (S * 2) >= 2048) S *= 2;
  slice = ((L + S - 1) / S + 3) / 4 * 4;
  return S;
}

// Row-walking forms: eligibility, slice picker (in tiles) and the grid (grid_rows below).
inline bool rows_ok(int64_t N, int64_t C, int64_t L, int D) {
  return (L & 3) == 0 && L >= 256 && N * C * L < ((int64_t)1 << 39) && N * C < ((int64_t)1 << 31) && D >= 1 && (D & (D - 1)) == 0;
}
inline bool rows_aligned(int64_t total, int64_t L) { return (L & 255) == 0 && (total & 255) == 0; }
inline int pick_slices_rows(int C, int L, int& slice) {
  int S = 1;
  while ((int64_t)C * S < 2048 && L / (S * 2) >= 2048) S *= 2;       // >= 8 tiles per (row, slice): two per wave
  slice = ((L + S - 1) / S + 255) / 256 * 256;                      // whole tiles when the rows are tile-aligned
  return (L + slice - 1) / slice;
}
// Workgroups of `Kern` that can be resident on the whole chip at once (the round-2 grid of the row kernels, kept as the `want = 0`
// form of grid_rows for the probe: 2 048 equal shares on 1 792 slots ran a second, 14 %-full round -- and, round 4, equal shares
// on exactly the resident slots still lose 13-28 % to the slowest CU, see grid_rows).
template <auto Kern>
int resident_blocks(size_t lds) {
  static int cus = 0, per_cu_small = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < 1)
      cus = 256;
  }
  int per_cu = lds <= 16384 ? per_cu_small : 0;
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kern, kBlock, lds <= 16384 ? 16384 : lds) != hipSuccess || per_cu < 1)
      per_cu = 4;
    if (lds <= 16384) per_cu_small = per_cu;
  }
  return cus * per_cu;
}
// -> grid; chunk = contiguous tiles per wave.  `want` tiles per wave in as many workgroups as that takes -- NOT one resident
// workgroup per equal share of the tensor: equal static shares finish with the slowest CU, short workgroups handed out by the
// dispatcher as others retire balance themselves (round 4, tools/probe_bn_stream.py, [8, 512, 16384] BatchNorm + neuron
// forward: 93 us on a resident grid, 73 us with 8 tiles per wave; [8, 256, 65536] 191 -> 164 us; the backward pair 525 -> 472 us
// with 4.  A resident grid that strides over the same 8-tile runs is as slow as the equal shares, 187 us; twice the resident
// grid is as fast as the full one: it is the balancing, not the address pattern).  S2F_BN_ROWS_CHUNK=<forward>,<backward> overrides
// `want` (0: the resident grid) for that probe.
template <auto Kern>
int grid_rows(uint32_t ntiles, size_t lds, uint32_t& chunk, int want) {
  static const char* env = getenv("S2F_BN_ROWS_CHUNK");              // "<forward>,<backward>"
  if (env != nullptr) {
    const int fwd_want = atoi(env), bwd_want = strchr(env, ',') ? atoi(strchr(env, ',') + 1) : fwd_want;
    want = want == 8 ? fwd_want : bwd_want;
  }
  int64_t blocks = ((int64_t)ntiles + 4 * kWaves - 1) / (4 * kWaves);      // >= 4 tiles per wave amortise the prologue
  const int64_t cap = want > 0 ? ((int64_t)ntiles + (int64_t)want * kWaves - 1) / ((int64_t)want * kWaves) : resident_blocks<Kern>(lds);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  chunk = (uint32_t)(((int64_t)ntiles + blocks * kWaves - 1) / (blocks * kWaves));
  return (int)(((int64_t)ntiles + (int64_t)chunk * kWaves - 1) / ((int64_t)chunk * kWaves));
}

inline int grid_flat(int64_t total) {
  int64_t tiles = (total + 255) >> 8;
  int64_t blocks = (tiles + 4 * kWaves - 1) / (4 * kWaves);      // >= 4 tiles per wave amortise the per-block prologue
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (int)(blocks < 1 ? 1 : blocks);
}

// Any L >= 1 is taken: rows of L % 4 != 0 elements run on the element-wise ANYL kernels (odd shapes only; the vector forms need
// a lane's four elements in one 16-byte-aligned group of one channel row).
int check_shape(const char* who, int64_t N, int64_t C, int64_t L) {
  S2F_REQUIRE(N > 0 && C > 0 && L > 0, S2F_EINVAL, "%s: bad shape N=%lld C=%lld L=%lld", who, (long long)N, (long long)C,
              (long long)L);
  S2F_REQUIRE(N * C * L < (1ll << 40) && L < (1ll << 31) && C <= 5120 && N < (1ll << 31), S2F_EINVAL,
              "%s: shape too large (C <= 5120: per-channel statistics are staged in 60 KiB of LDS)", who);
  return S2F_OK;
}

}  // namespace

extern "C" int s2f_bn_single_pass(int64_t N, int64_t C, int64_t L) {
  return (single_pass_ok(N, C, L) || small_rows_ok(N, C, L) || mid_rows_ok(N, C, L)) ? 1 : 0;
}

// uint64 words of the in-range mask s2f_bn_act_fwd writes / s2f_bn_act_bwd reads for this shape: the flat-tile layout of
// s2f_lif_mask_words, or the per-channel layout of the short-row single-pass kernels (private to this fwd / bwd pair)
extern "C" int64_t s2f_bn_mask_words(int64_t N, int64_t C, int64_t L) {
  if (small_rows_ok(N, C, L)) return C * kSmallIters * 4;
  if (mid_rows_ok(N, C, L)) return C * kMidIters * kMidWaves * 4;
  return ((N * C * L + 255) >> 8) * 4;
}

extern "C" int s2f_bn_stats(const float* z, const float* conv_bias, double* sums_zeroed, int64_t N, int64_t C, int64_t L,
                            void* stream) {
  S2F_REQUIRE(z && sums_zeroed, S2F_EINVAL, "s2f_bn_stats: null z/workspace");
  int rc = check_shape("s2f_bn_stats", N, C, L);
  if (rc) return rc;
  int slice;
  const int S = pick_slices((int)C, (int)L, slice);
  if (L & 3) {
    S2F_LAUNCH(true, true, bn_stats_any_kernel, dim3((unsigned)C, S), dim3(kBlock), 0, (hipStream_t)stream, z, conv_bias,
               sums_zeroed, (int)N, (int)C, (int)L, slice);
    return s2f_check_launch("s2f_bn_stats");
  }
  S2F_REQUIRE(s2f_aligned16(z), S2F_EALIGN, "s2f_bn_stats: z must be 16-byte aligned");
  S2F_LAUNCH(true, true, bn_stats_kernel, dim3((unsigned)C, S), dim3(kBlock), 0, (hipStream_t)stream, z, conv_bias,
                     sums_zeroed, (int)N, (int)C, (int)L, slice);
  return s2f_check_launch("s2f_bn_stats");
}

extern "C" int s2f_bn_partials_finalize(const float* partials, int64_t P, const float* conv_bias, double* sums_out, int64_t N,
                                        int64_t C, int64_t L, void* stream) {
  S2F_REQUIRE(partials && sums_out && P > 0 && P < (1 << 24) && (reinterpret_cast<uintptr_t>(partials) & 7u) == 0, S2F_EINVAL,
              "s2f_bn_partials_finalize: partials 8-byte aligned, 0 < P < 2^24, sums workspace given");
  int rc = check_shape("s2f_bn_partials_finalize", N, C, L);
  if (rc) return rc;
  const int per_wave = P <= 256;
  const unsigned fgrid = per_wave ? (unsigned)((C + kWaves - 1) / kWaves) : (unsigned)C;
  S2F_LAUNCH(true, true, bn_partials_finalize_kernel, dim3(fgrid), dim3(kBlock), 0, (hipStream_t)stream, partials, conv_bias,
             sums_out, (int)P, (int)C, (double)N * (double)L, per_wave);
  return s2f_check_launch("s2f_bn_partials_finalize");
}

static int bn_act_fwd_impl(const float* z, const float* conv_bias, const double* sums, float* stat_out, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, const float* gamma, const float* beta,
                           const float* residual, float* u_out, const float* v_in, void* y_out, float* v_out, uint64_t* mask,
                           uint64_t* stats, int64_t N, int64_t C, int64_t L, float momentum, float eps, int training, float vth,
                           int D, int y_bf16, void* stream, Bn2 bn2) {
  float* y = reinterpret_cast<float*>(y_out);
  S2F_REQUIRE(z && stat_out && gamma && beta, S2F_EINVAL, "s2f_bn_act_fwd: null z/stat/gamma/beta");
  S2F_REQUIRE(!bn2.gamma || (training && sums == nullptr && single_pass_ok(N, C, L) && bn2.beta), S2F_EINVAL,
              "s2f_bn2_act_fwd: the BatchNorm pair runs on the single-pass kernels only (training mode, s2f_bn2_fused_ok)");
  constexpr bool first_launch = true;
  // sums given for a shape that could go single-pass: the caller already has the statistics (a producer's epilogue, or a probe) --
  // take the apply path, which is not tied to one workgroup per channel
  const bool single = training && sums == nullptr && single_pass_ok(N, C, L);
  S2F_REQUIRE(training ? (single || sums != nullptr || small_rows_ok(N, C, L) || mid_rows_ok(N, C, L)) : (running_mean && running_var), S2F_EINVAL,
              "s2f_bn_act_fwd: training needs the sums of s2f_bn_stats / s2f_bn_partials_finalize, eval needs the running statistics");
  S2F_REQUIRE(u_out || y, S2F_EINVAL, "s2f_bn_act_fwd: neither u_out nor y requested");
  S2F_REQUIRE(!(y && y_bf16) || s2f_bf16_spikes_exact(D), S2F_EINVAL,
              "s2f_bn_act_fwd: bf16 spikes need D a power of two <= 128 (D=%d)", D);
  int rc = check_shape("s2f_bn_act_fwd", N, C, L);
  if (rc) return rc;
  const bool anyl = (L & 3) != 0;
  S2F_REQUIRE(anyl || (s2f_aligned16(z) && s2f_aligned16(residual) && s2f_aligned16(u_out) && s2f_aligned16(v_in) &&
                       s2f_aligned16(y) && s2f_aligned16(v_out)),
              S2F_EALIGN, "s2f_bn_act_fwd: tensors must be 16-byte aligned");
  const int64_t total = N * C * L;
  hipStream_t s = (hipStream_t)stream;
  auto* st = reinterpret_cast<unsigned long long*>(stats);
  auto* nbt = reinterpret_cast<long long*>(num_batches_tracked);
  const dim3 grid(grid_flat(total)), block(kBlock);
  const double count = (double)N * (double)L;
  const double inv_count = 1.0 / count;
  const float unbias = count > 1 ? (float)(count / (count - 1.0)) : 1.0f;
  if (training && mid_rows_ok(N, C, L)) {
    // (always, given statistics or not: the mask layout of this shape is the per-channel one, which the backward reads)
#define S2F_BN_MID(LIFV, HASV, YBV)                                                                                      \
  S2F_LAUNCH(true, true, (bn_mid_fwd_kernel<LIFV, HASV, YBV>), dim3((unsigned)C), dim3(mid_rows_threads(N, L)), 0, s, z, conv_bias, \
             stat_out, running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st, (int)N, (int)C,   \
             (int)L, inv_count, unbias, momentum, eps, vth, (float)D)
    if (y == nullptr)
      S2F_BN_MID(false, false, false);
    else if (v_in == nullptr) {
      if (y_bf16)
        S2F_BN_MID(true, false, true);
      else
        S2F_BN_MID(true, false, false);
    } else {
      if (y_bf16)
        S2F_BN_MID(true, true, true);
      else
        S2F_BN_MID(true, true, false);
    }
#undef S2F_BN_MID
    return s2f_check_launch("s2f_bn_act_fwd");
  }
  if (training && small_rows_ok(N, C, L)) {
    // (always, given statistics or not: the mask layout of this shape is the short-row one, which the backward reads)
#define S2F_BN_SMALL(LIFV, HASV, YBV)                                                                                    \
  S2F_LAUNCH(true, true, (bn_small_fwd_kernel<LIFV, HASV, YBV>), dim3((unsigned)C), dim3(64), 0, s, z, conv_bias, stat_out, \
             running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st, (int)N, (int)C,      \
             (int)L, inv_count, unbias, momentum, eps, vth, (float)D)
    if (y == nullptr)
      S2F_BN_SMALL(false, false, false);
    else if (v_in == nullptr) {
      if (y_bf16)
        S2F_BN_SMALL(true, false, true);
      else
        S2F_BN_SMALL(true, false, false);
    } else {
      if (y_bf16)
        S2F_BN_SMALL(true, true, true);
      else
        S2F_BN_SMALL(true, true, false);
    }
#undef S2F_BN_SMALL
    return s2f_check_launch("s2f_bn_act_fwd");
  }
  if (single) {
    const dim3 fgrid((unsigned)C), fblock(single_pass_threads(N, L));
#define S2F_BN_FUSED(LIFV, HASV, YBV)                                                                                    \
  do {                                                                                                                   \
    if (bn2.gamma)                                                                                                       \
      S2F_LAUNCH(true, true, (bn_fused_fwd_kernel<LIFV, HASV, YBV, true>), fgrid, fblock, 0, s, z, conv_bias, stat_out,    \
                 running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st, (int)N, (int)C,  \
                 (int)L, inv_count, unbias, momentum, eps, vth, (float)D, bn2);                                          \
    else                                                                                                                 \
      S2F_LAUNCH(true, true, (bn_fused_fwd_kernel<LIFV, HASV, YBV>), fgrid, fblock, 0, s, z, conv_bias, stat_out,          \
                 running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st, (int)N, (int)C,  \
                 (int)L, inv_count, unbias, momentum, eps, vth, (float)D, bn2);                                          \
  } while (0)
    if (y == nullptr)
      S2F_BN_FUSED(false, false, false);
    else if (v_in == nullptr) {
      if (y_bf16)
        S2F_BN_FUSED(true, false, true);
      else
        S2F_BN_FUSED(true, false, false);
    } else {
      if (y_bf16)
        S2F_BN_FUSED(true, true, true);
      else
        S2F_BN_FUSED(true, true, false);
    }
#undef S2F_BN_FUSED
    return s2f_check_launch("s2f_bn_act_fwd");
  }
  const bool rows = rows_ok(N, C, L, D), aligned = rows_aligned(total, L);
  const size_t lds = 3 * C * sizeof(float) + 64;
#define S2F_BN_ROWS_FWD(LIFV, HASV, YBV, AL)                                                                              \
  do {                                                                                                                  \
    uint32_t chunk;                                                                                                     \
    const uint32_t ntiles = (uint32_t)((total + 255) >> 8);                                                             \
    const int rgrid = grid_rows<bn_apply_rows_kernel<LIFV, HASV, YBV, AL>>(ntiles, lds, chunk, 8);                         \
    S2F_LAUNCH(first_launch, true, (bn_apply_rows_kernel<LIFV, HASV, YBV, AL>), dim3(rgrid), block, lds, s, z, conv_bias, \
               sums, stat_out, running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st,   \
               total, ntiles, (int)C, (uint32_t)L, chunk, inv_count, unbias, momentum, eps, training, vth, (float)D);    \
  } while (0)
#define S2F_BN_APPLY(LIFV, HASV, YBV)                                                                                   \
  do {                                                                                                                  \
    if (rows && aligned) {                                                                                              \
      S2F_BN_ROWS_FWD(LIFV, HASV, YBV, true);                                                                           \
    } else if (rows) {                                                                                                  \
      S2F_BN_ROWS_FWD(LIFV, HASV, YBV, false);                                                                          \
    } else if (anyl) {                                                                                                  \
      S2F_LAUNCH(first_launch, true, (bn_apply_kernel<LIFV, HASV, YBV, true>), grid, block, lds, s, z, conv_bias, sums, \
                 stat_out, running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st,      \
                 total, (int)C, (int)L, inv_count, unbias, momentum, eps, training, vth, (float)D);                     \
    } else {                                                                                                            \
      S2F_LAUNCH(first_launch, true, (bn_apply_kernel<LIFV, HASV, YBV>), grid, block, lds, s, z, conv_bias, sums,        \
                 stat_out, running_mean, running_var, nbt, gamma, beta, residual, u_out, v_in, y, v_out, mask, st,      \
                 total, (int)C, (int)L, inv_count, unbias, momentum, eps, training, vth, (float)D);                     \
    }                                                                                                                   \
  } while (0)
  if (y == nullptr)
    S2F_BN_APPLY(false, false, false);
  else if (v_in == nullptr) {
    if (y_bf16)
      S2F_BN_APPLY(true, false, true);
    else
      S2F_BN_APPLY(true, false, false);
  } else {
    if (y_bf16)
      S2F_BN_APPLY(true, true, true);
    else
      S2F_BN_APPLY(true, true, false);
  }
#undef S2F_BN_APPLY
#undef S2F_BN_ROWS_FWD
  return s2f_check_launch("s2f_bn_act_fwd");
}

extern "C" int s2f_bn_act_fwd(const float* z, const float* conv_bias, const double* sums, float* stat_out,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                              const float* beta, const float* residual, float* u_out, const float* v_in, void* y_out,
                              float* v_out, uint64_t* mask, uint64_t* stats, int64_t N, int64_t C, int64_t L,
                              float momentum, float eps, int training, float vth, int D, int y_bf16, void* stream) {
  return bn_act_fwd_impl(z, conv_bias, sums, stat_out, running_mean, running_var, num_batches_tracked, gamma, beta, residual, u_out,
                         v_in, y_out, v_out, mask, stats, N, C, L, momentum, eps, training, vth, D, y_bf16, stream, Bn2{});
}

extern "C" int s2f_bn2_fused_ok(int64_t N, int64_t C, int64_t L) { return single_pass_ok(N, C, L) ? 1 : 0; }

extern "C" int s2f_bn2_act_fwd(const float* z, const float* conv_bias, float* stat_out, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, const float* gamma, const float* beta, float momentum, float eps,
                               const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                               int64_t* num_batches_tracked2, float momentum2, float eps2, const float* residual, float* u_out,
                               const float* v_in, void* y_out, float* v_out, uint64_t* mask, uint64_t* stats, int64_t N, int64_t C,
                               int64_t L, float vth, int D, int y_bf16, void* stream) {
  S2F_REQUIRE(gamma2 && beta2, S2F_EINVAL, "s2f_bn2_act_fwd: null second BatchNorm");
  return bn_act_fwd_impl(z, conv_bias, nullptr, stat_out, running_mean, running_var, num_batches_tracked, gamma, beta, residual, u_out,
                         v_in, y_out, v_out, mask, stats, N, C, L, momentum, eps, 1, vth, D, y_bf16, stream,
                         Bn2{gamma2, beta2, running_mean2, running_var2, reinterpret_cast<long long*>(num_batches_tracked2), eps2,
                             momentum2});
}

static int bn_act_bwd_impl(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* g_u,
                           const float* g_y, const float* g_v, const uint64_t* mask, double* sums_zeroed, float* gz,
                           float* g_residual, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t L, int training,
                           float vth, int D, void* stream, unsigned short* gzs, Bn2Bwd bn2 = Bn2Bwd{}, const float* g_y2 = nullptr) {
  S2F_REQUIRE(!g_y2 || (g_y && s2f_aligned16(g_y2) && !(training && (mid_rows_ok(N, C, L) || small_rows_ok(N, C, L))) &&
                        ((training && single_pass_ok(N, C, L)) || rows_ok(N, C, L, D))),
              S2F_EINVAL, "s2f_bn_act_bwd_ports: a second spike gradient only on the single-pass and row-walking kernels (s2f_bn_bwd_ports_ok)");
  S2F_REQUIRE(!bn2.gamma || (training && single_pass_ok(N, C, L) && bn2.dgamma && bn2.dbeta && !gzs), S2F_EINVAL,
              "s2f_bn2_act_bwd: the BatchNorm pair runs on the single-pass kernels only");
  const bool single = training && (single_pass_ok(N, C, L) || small_rows_ok(N, C, L) || mid_rows_ok(N, C, L));
  S2F_REQUIRE(z && stat && gamma && (single || sums_zeroed) && (gz || gzs) && dgamma && dbeta, S2F_EINVAL,
              "s2f_bn_act_bwd: null pointer");
  S2F_REQUIRE(!gzs || (reinterpret_cast<uintptr_t>(gzs) & 7u) == 0, S2F_EALIGN, "s2f_bn_act_bwd_split: gz_split must be 8-byte aligned");
  S2F_REQUIRE(g_u || g_y || g_v, S2F_EINVAL, "s2f_bn_act_bwd: no incoming gradient");
  S2F_REQUIRE(!(g_y || g_v) || mask, S2F_EINVAL, "s2f_bn_act_bwd: spike gradients need the in-range mask");
  int rc = check_shape("s2f_bn_act_bwd", N, C, L);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = N * C * L;
  if (training && mid_rows_ok(N, C, L)) {
    S2F_REQUIRE(gzs == nullptr && gz, S2F_EINVAL, "s2f_bn_act_bwd_split: rows that are not whole tiles have no bf16-plane form");
#define S2F_BN_MB(A, B, Cc)                                                                                               \
  S2F_LAUNCH(true, true, (bn_mid_bwd_kernel<A, B, Cc>), dim3((unsigned)C), dim3(mid_rows_threads(N, L)), 0, s, z, conv_bias, stat, \
             gamma, g_u, g_y, g_v, mask, gz, g_residual, dgamma, dbeta, (int)N, (int)C, (int)L, 1.0 / ((double)N * (double)L), vth, \
             (float)D)
    switch ((g_u ? 4 : 0) | (g_y ? 2 : 0) | (g_v ? 1 : 0)) {
      case 1: S2F_BN_MB(false, false, true); break;
      case 2: S2F_BN_MB(false, true, false); break;
      case 3: S2F_BN_MB(false, true, true); break;
      case 4: S2F_BN_MB(true, false, false); break;
      case 5: S2F_BN_MB(true, false, true); break;
      case 6: S2F_BN_MB(true, true, false); break;
      default: S2F_BN_MB(true, true, true); break;
    }
#undef S2F_BN_MB
    return s2f_check_launch("s2f_bn_act_bwd");
  }
  if (training && small_rows_ok(N, C, L)) {
    S2F_REQUIRE(gzs == nullptr && gz, S2F_EINVAL, "s2f_bn_act_bwd_split: short rows (N * L <= 2048) have no bf16-plane form");
#define S2F_BN_SB(A, B, Cc)                                                                                               \
  S2F_LAUNCH(true, true, (bn_small_bwd_kernel<A, B, Cc>), dim3((unsigned)C), dim3(64), 0, s, z, conv_bias, stat, gamma, g_u, \
             g_y, g_v, mask, gz, g_residual, dgamma, dbeta, (int)N, (int)C, (int)L, 1.0 / ((double)N * (double)L), vth,    \
             (float)D)
    switch ((g_u ? 4 : 0) | (g_y ? 2 : 0) | (g_v ? 1 : 0)) {
      case 1: S2F_BN_SB(false, false, true); break;
      case 2: S2F_BN_SB(false, true, false); break;
      case 3: S2F_BN_SB(false, true, true); break;
      case 4: S2F_BN_SB(true, false, false); break;
      case 5: S2F_BN_SB(true, false, true); break;
      case 6: S2F_BN_SB(true, true, false); break;
      default: S2F_BN_SB(true, true, true); break;
    }
#undef S2F_BN_SB
    return s2f_check_launch("s2f_bn_act_bwd");
  }
  if (single) {
#define S2F_BN_FB(A, B, Cc)                                                                                               \
  do {                                                                                                                    \
    if (bn2.gamma)                                                                                                        \
      S2F_LAUNCH(true, true, (bn_fused_bwd_kernel<A, B, Cc, true>), dim3((unsigned)C), dim3(single_pass_threads(N, L)), 0,  \
                 s, z, conv_bias, stat, gamma, g_u, g_y, g_v, mask, gz, g_residual, dgamma, dbeta, (int)N, (int)C,        \
                 (int)L, 1.0 / ((double)N * (double)L), vth, (float)D, gzs, bn2, g_y2);                                   \
    else                                                                                                                  \
      S2F_LAUNCH(true, true, (bn_fused_bwd_kernel<A, B, Cc>), dim3((unsigned)C), dim3(single_pass_threads(N, L)), 0, s, z,  \
                 conv_bias, stat, gamma, g_u, g_y, g_v, mask, gz, g_residual, dgamma, dbeta, (int)N, (int)C, (int)L,      \
                 1.0 / ((double)N * (double)L), vth, (float)D, gzs, bn2, g_y2);                                           \
  } while (0)
    const int combo = (g_u ? 4 : 0) | (g_y ? 2 : 0) | (g_v ? 1 : 0);
    switch (combo) {
      case 1: S2F_BN_FB(false, false, true); break;
      case 2: S2F_BN_FB(false, true, false); break;
      case 3: S2F_BN_FB(false, true, true); break;
      case 4: S2F_BN_FB(true, false, false); break;
      case 5: S2F_BN_FB(true, false, true); break;
      case 6: S2F_BN_FB(true, true, false); break;
      default: S2F_BN_FB(true, true, true); break;
    }
#undef S2F_BN_FB
    return s2f_check_launch("s2f_bn_act_bwd");
  }
  if (rows_ok(N, C, L, D)) {
    int slice;
    const int S = pick_slices_rows((int)C, (int)L, slice);
    const uint32_t ntiles = (uint32_t)((total + 255) >> 8);
    const double inv_count = 1.0 / ((double)N * (double)L);
    const bool aligned = rows_aligned(total, L);
#define S2F_BN_ROWS_APPLY(A, B, Cc, AL)                                                                                  \
  do {                                                                                                                   \
    uint32_t chunk;                                                                                                      \
    const int rgrid = grid_rows<bn_bwd_apply_rows_kernel<A, B, Cc, AL>>(ntiles, 0, chunk, 4);                               \
    S2F_LAUNCH(false, true, (bn_bwd_apply_rows_kernel<A, B, Cc, AL>), dim3(rgrid), dim3(kBlock), 0, s, z, conv_bias, stat, \
               gamma, g_u, g_y, g_v, mask, sums_zeroed, gz, g_residual, dgamma, dbeta, total, ntiles, (int)C, (uint32_t)L, \
               chunk, inv_count, training, vth, (float)D, gzs, g_y2);                                                    \
  } while (0)
#define S2F_BN_ROWS_BWD(A, B, Cc)                                                                                        \
  do {                                                                                                                   \
    S2F_LAUNCH(true, false, (bn_bwd_reduce_rows_kernel<A, B, Cc>), dim3((unsigned)C, S), dim3(kBlock), 0, s, z, conv_bias, \
               stat, g_u, g_y, g_v, mask, sums_zeroed, (int)N, (int)C, (int)L, slice, vth, (float)D, g_y2);               \
    if (aligned)                                                                                                         \
      S2F_BN_ROWS_APPLY(A, B, Cc, true);                                                                                 \
    else                                                                                                                 \
      S2F_BN_ROWS_APPLY(A, B, Cc, false);                                                                                \
  } while (0)
    switch ((g_u ? 4 : 0) | (g_y ? 2 : 0) | (g_v ? 1 : 0)) {
      case 1: S2F_BN_ROWS_BWD(false, false, true); break;
      case 2: S2F_BN_ROWS_BWD(false, true, false); break;
      case 3: S2F_BN_ROWS_BWD(false, true, true); break;
      case 4: S2F_BN_ROWS_BWD(true, false, false); break;
      case 5: S2F_BN_ROWS_BWD(true, false, true); break;
      case 6: S2F_BN_ROWS_BWD(true, true, false); break;
      default: S2F_BN_ROWS_BWD(true, true, true); break;
    }
#undef S2F_BN_ROWS_BWD
#undef S2F_BN_ROWS_APPLY
    return s2f_check_launch("s2f_bn_act_bwd");
  }
  int slice;
  const int S = pick_slices((int)C, (int)L, slice);
  if (L & 3) {
    S2F_REQUIRE(gzs == nullptr, S2F_EINVAL, "s2f_bn_act_bwd_split: rows of L %% 4 != 0 elements have no bf16-plane form");
    S2F_LAUNCH(true, false, bn_bwd_reduce_any_kernel, dim3((unsigned)C, S), dim3(kBlock), 0, s, z, conv_bias, stat, g_u, g_y,
               g_v, mask, sums_zeroed, (int)N, (int)C, (int)L, slice, vth, (float)D);
    S2F_LAUNCH(false, true, bn_bwd_apply_kernel<true>, dim3(grid_flat(total)), dim3(kBlock), 2 * C * sizeof(float), s, z,
               conv_bias, stat, gamma, g_u, g_y, g_v, mask, sums_zeroed, gz, g_residual, dgamma, dbeta, total, (int)C, (int)L,
               1.0 / ((double)N * (double)L), training, vth, (float)D, gzs);
    return s2f_check_launch("s2f_bn_act_bwd");
  }
  S2F_LAUNCH(true, false, bn_bwd_reduce_kernel, dim3((unsigned)C, S), dim3(kBlock), 0, s, z, conv_bias, stat, g_u, g_y, g_v,
                     mask, sums_zeroed, (int)N, (int)C, (int)L, slice, vth, (float)D);
  S2F_LAUNCH(false, true, bn_bwd_apply_kernel<false>, dim3(grid_flat(total)), dim3(kBlock), 2 * C * sizeof(float), s, z, conv_bias, stat, gamma, g_u, g_y,
                     g_v, mask, sums_zeroed, gz, g_residual, dgamma, dbeta, total, (int)C, (int)L,
                     1.0 / ((double)N * (double)L), training, vth, (float)D, gzs);
  return s2f_check_launch("s2f_bn_act_bwd");
}

extern "C" int s2f_bn_act_bwd(const float* z, const float* conv_bias, const float* stat, const float* gamma,
                              const float* g_u, const float* g_y, const float* g_v, const uint64_t* mask,
                              double* sums_zeroed, float* gz, float* g_residual, float* dgamma, float* dbeta, int64_t N,
                              int64_t C, int64_t L, int training, float vth, int D, void* stream) {
  S2F_REQUIRE(gz, S2F_EINVAL, "s2f_bn_act_bwd: null gz");
  return bn_act_bwd_impl(z, conv_bias, stat, gamma, g_u, g_y, g_v, mask, sums_zeroed, gz, g_residual, dgamma, dbeta, N, C, L,
                         training, vth, D, stream, nullptr);
}

// 1: s2f_bn_act_bwd_ports takes a second spike gradient for this shape (the kernels that carry the large maps and the 32 x 32 stage)
extern "C" int s2f_bn_bwd_ports_ok(int64_t N, int64_t C, int64_t L, int training, int D) {
  if (N <= 0 || C <= 0 || L <= 0) return 0;
  if (training && (mid_rows_ok(N, C, L) || small_rows_ok(N, C, L))) return 0;
  if (training && single_pass_ok(N, C, L)) return 1;
  return rows_ok(N, C, L, D) ? 1 : 0;
}

extern "C" int s2f_bn_act_bwd_ports(const float* z, const float* conv_bias, const float* stat, const float* gamma,
                                    const float* g_u, const float* g_y, const float* g_y2, const float* g_v, const uint64_t* mask,
                                    double* sums_zeroed, float* gz, float* g_residual, float* dgamma, float* dbeta, int64_t N,
                                    int64_t C, int64_t L, int training, float vth, int D, void* stream) {
  S2F_REQUIRE(gz, S2F_EINVAL, "s2f_bn_act_bwd_ports: null gz");
  return bn_act_bwd_impl(z, conv_bias, stat, gamma, g_u, g_y, g_v, mask, sums_zeroed, gz, g_residual, dgamma, dbeta, N, C, L,
                         training, vth, D, stream, nullptr, Bn2Bwd{}, g_y2);
}

extern "C" int s2f_bn_act_bwd_split(const float* z, const float* conv_bias, const float* stat, const float* gamma,
                                    const float* g_u, const float* g_y, const float* g_v, const uint64_t* mask,
                                    double* sums_zeroed, uint16_t* gz_split, float* g_residual, float* dgamma, float* dbeta,
                                    int64_t N, int64_t C, int64_t L, int training, float vth, int D, void* stream) {
  S2F_REQUIRE(gz_split, S2F_EINVAL, "s2f_bn_act_bwd_split: null gz_split");
  return bn_act_bwd_impl(z, conv_bias, stat, gamma, g_u, g_y, g_v, mask, sums_zeroed, nullptr, g_residual, dgamma, dbeta, N, C,
                         L, training, vth, D, stream, gz_split);
}

extern "C" int s2f_bn2_act_bwd(const float* z, const float* conv_bias, const float* stat, const float* gamma, const float* gamma2,
                               float eps2, const float* g_u, const float* g_y, const float* g_v, const uint64_t* mask, float* gz,
                               float* g_residual, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, int64_t N, int64_t C,
                               int64_t L, float vth, int D, void* stream) {
  S2F_REQUIRE(gz && gamma2, S2F_EINVAL, "s2f_bn2_act_bwd: null gz / second BatchNorm");
  return bn_act_bwd_impl(z, conv_bias, stat, gamma, g_u, g_y, g_v, mask, nullptr, gz, g_residual, dgamma, dbeta, N, C, L, 1, vth, D,
                         stream, nullptr, Bn2Bwd{gamma2, dgamma2, dbeta2, eps2});
}
