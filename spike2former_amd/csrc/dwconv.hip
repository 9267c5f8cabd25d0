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
#include <cstdlib>

#pragma clang fp contract(fast)

namespace {

constexpr int TS = 32;   // tile side (outputs)

// The stencil input is fp32, or a bf16 spike map (uint16 storage, exact) as the neuron kernels write it.
__device__ __forceinline__ float ldx(const float* p) { return *p; }
__device__ __forceinline__ float ldx(const unsigned short* p) { return __uint_as_float(((unsigned int)*p) << 16); }
__device__ __forceinline__ float4 ldx4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldx4(const unsigned short* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

// Stage the (TS + K - 1)^2 halo tile whose origin is (oy0, ox0) in plane coordinates.  The aligned TS x TS block inside
// it (offset `po` in both directions: plane pixel (ty, tx), tx % 4 == 0) is loaded with ONE 16-byte load per thread; the
// K - 1 wide ring around it with scalar loads by the first threads.  `fill` = value read in the padding ring of width
// `fpad` around the plane (BNAndPadLayer's border), zero further out.
template <int K, typename TX>
__device__ __forceinline__ void stage_halo(float (&s)[TS + K - 1][TS + K], const TX* __restrict__ xp, int H, int W, int oy0,
                                           int ox0, int po, float fillv, int fpad, bool vec_ok) {
  constexpr int HS = TS + K - 1;
  auto at = [&](int iy, int ix) -> float {
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) return ldx(xp + (int64_t)iy * W + ix);
    if (iy >= -fpad && iy < H + fpad && ix >= -fpad && ix < W + fpad) return fillv;
    return 0.f;
  };
  {
    const int ri = threadIdx.x >> 3, qi = (threadIdx.x & 7) * 4;
    const int iy = oy0 + po + ri, ix = ox0 + po + qi;
    float4 v;
    if (vec_ok && iy >= 0 && iy < H && ix >= 0 && ix + 3 < W) {
      v = ldx4(xp + (int64_t)iy * W + ix);
    } else {
      v = make_float4(at(iy, ix), at(iy, ix + 1), at(iy, ix + 2), at(iy, ix + 3));
    }
    float* d = &s[po + ri][po + qi];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  // ring: `po` rows above, K-1-po rows below (full width HS), then po / K-1-po columns beside the TS interior rows
  constexpr int RING = HS * HS - TS * TS;
  const int top = po * HS, bot = (K - 1 - po) * HS;
  for (int e = threadIdx.x; e < RING; e += 256) {
    int r, q;
    if (e < top) {
      r = e / HS; q = e - r * HS;
    } else if (e < top + bot) {
      const int f = e - top;
      r = po + TS + f / HS; q = f % HS;
    } else {
      const int f = e - top - bot;                      // TS rows x (K-1) side columns
      r = po + f / (K - 1);
      const int j = f % (K - 1);
      q = j < po ? j : TS + j;
    }
    s[r][q] = at(oy0 + r, ox0 + q);
  }
}

// Eval-mode epilogue of s2f_dwconv_bn_lif_fwd (row f4): the BatchNorm that follows a depthwise convolution everywhere on the path
// (SepConv_Spike.dwconv, DCNv3.dw_conv, the pixel decoder's output convolutions: SNN_core.py:36-45, dcnv3.py:161-169,
// pixel_decoder.py:374-378) with its running statistics, and the Q_IFNode after it, applied to the stencil's four outputs before
// they are stored: u = (acc - mean) rstd gamma + beta ; spikes as bf16.  Per-element expressions of bn_lif.hip / lif.hip.
struct DwEpi {
  const float* mean;             // running_mean [C]; null = plain stencil
  const float* var;
  const float* gamma;
  const float* beta;
  float* u_out;                  // fp32 pre-activation [N][C][Ho][Wo] or null
  unsigned short* y;             // bf16 spikes or null (no neuron)
  unsigned long long* stats;     // firing counters or null
  float eps, vth, Df, inv_d;
};

// four consecutive outputs of channel c at flat offset `off` (16-byte aligned when `vec`), `nvalid` of them inside the row
__device__ __forceinline__ void dw_epi_store(const DwEpi& ep, int c, int64_t off, const float (&acc)[4], int nvalid, bool vec,
                                             uint32_t& csum, uint32_t& cnz) {
#pragma clang fp contract(off)
  const float mean = ep.mean[c], rstd = 1.0f / sqrtf(ep.var[c] + ep.eps), g = ep.gamma[c], be = ep.beta[c];
  float u[4], yy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    u[o] = (acc[o] - mean) * rstd * g + be;
    if (ep.y && o < nvalid) {
      float sp, vn;
      bool inr;
      s2f_lif_update(u[o], ep.Df, ep.inv_d, ep.vth, sp, yy[o], vn, inr);
      csum += (uint32_t)sp;
      cnz += ((uint32_t)sp != 0);
    }
  }
  if (vec && nvalid == 4) {
    if (ep.u_out) *reinterpret_cast<float4*>(ep.u_out + off) = make_float4(u[0], u[1], u[2], u[3]);
    if (ep.y) *reinterpret_cast<uint2*>(ep.y + off) = s2f_spikes_to_bf16x4(yy[0], yy[1], yy[2], yy[3]);
  } else {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (o >= nvalid) continue;
      if (ep.u_out) ep.u_out[off + o] = u[o];
      if (ep.y) ep.y[off + o] = (unsigned short)(__float_as_uint(yy[o]) >> 16);
    }
  }
}

__device__ __forceinline__ void dw_epi_counters(const DwEpi& ep, uint32_t csum, uint32_t cnz) {
  if (ep.y == nullptr || ep.stats == nullptr) return;
  for (int o = 32; o > 0; o >>= 1) {
    csum += __shfl_xor(csum, o, 64);
    cnz += __shfl_xor(cnz, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* slot = ep.stats + 2 * ((blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y * 7) % S2F_STAT_SLOTS);
    if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
    if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
  }
}

// FLIP = false: y[oy][ox] = sum_{i,j} w[i][j] * x[oy + i - pad][ox + j - pad]           (x: H x W, y: Ho x Wo)
// FLIP = true : y[oy][ox] = sum_{i,j} w[i][j] * x[oy - i + pad][ox - j + pad]           (input gradient: x = gy)
template <int K, bool FLIP, typename TX, bool EPI = false>
__global__ __launch_bounds__(256) void dw_stencil_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ border, float* __restrict__ y, int C,
                                                         int H, int W, int Ho, int Wo, int pad, int tiles_x, DwEpi ep = DwEpi{}) {
  constexpr int HS = TS + K - 1;
  __shared__ float s[HS][HS + 1];
  const int plane = blockIdx.y;
  const int c = plane % C;
  const int ty = (blockIdx.x / tiles_x) * TS, tx = (blockIdx.x % tiles_x) * TS;
  const TX* xp = x + (int64_t)plane * H * W;
  const float fillv = (!FLIP && border) ? border[c] : 0.f;
  // halo origin in input coordinates; the aligned block (ty, tx) sits `po` inside it
  const int po = FLIP ? K - 1 - pad : pad;
  const bool vec_ok = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & (4 * sizeof(TX) - 1)) == 0;
  stage_halo<K, TX>(s, xp, H, W, ty - po, tx - po, po, fillv, FLIP ? 0 : pad, vec_ok);
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
  if constexpr (EPI) {
    uint32_t csum = 0, cnz = 0;
    if (oy < Ho && tx + q0 < Wo) {
      const int64_t off = (int64_t)plane * Ho * Wo + (int64_t)oy * Wo + tx + q0;
      dw_epi_store(ep, c, off, acc, min(4, Wo - tx - q0), (Wo & 3) == 0 && (off & 3) == 0, csum, cnz);
    }
    dw_epi_counters(ep, csum, cnz);
    return;
  }
  if (oy < Ho) {
    float* yp = y + (int64_t)plane * Ho * Wo + (int64_t)oy * Wo + tx + q0;
    if ((Wo & 3) == 0 && tx + q0 + 3 < Wo && (reinterpret_cast<uintptr_t>(yp) & 15u) == 0) {
      *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (tx + q0 + o < Wo) yp[o] = acc[o];
    }
  }
}

// Wide-map form of the same stencil (W % 4 == 0, W >= 64): a 64 x 32 output tile per workgroup, every thread a 2 x 4 output
// block.  The halo is staged as whole 16-byte groups of an ALIGNED superset of its columns ([tx - 4, tx + 68): every group
// lies entirely inside or entirely outside the plane, so the loads are branch-free vector loads with a select -- the scalar
// ring loads of the 32 x 32 form each sat behind their own bounds branch), LDS rows are read back as ds_read_b128 (16
// consecutive lanes read 256 consecutive bytes: conflict-free) and each staged row feeds two output rows.  Per output:
// 1.5 LDS reads of 16 bytes instead of 17.5 of 4 bytes.  'Same' padding (pad == (K - 1) / 2) only.
constexpr int WT = 64, HT = 32, WS = WT + 8;           // tile and staged row length (floats)

template <int K, bool FLIP, typename TX, bool EPI = false>
__global__ __launch_bounds__(256) void dw_stencil_wide_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ border, float* __restrict__ y,
                                                              int C, int H, int W, int Ho, int Wo, int pad, int tiles_x,
                                                              DwEpi ep = DwEpi{}) {
  constexpr int HR = HT + K - 1;                       // staged rows
  __shared__ __attribute__((aligned(16))) float s[HR][WS];
  const int plane = blockIdx.y;
  const int c = plane % C;
  const int ty = (blockIdx.x / tiles_x) * HT, tx = (blockIdx.x % tiles_x) * WT;
  const TX* xp = x + (int64_t)plane * H * W;
  const float fillv = (!FLIP && border) ? border[c] : 0.f;    // every off-plane value a valid output reads is the border
  const int po = pad;                                  // == K - 1 - pad
  // stage rows ty - po .. ty - po + HR - 1, columns tx - 4 .. tx + 67 as 18 groups of 4
  for (int e = threadIdx.x; e < HR * (WS / 4); e += 256) {
    const int r = e / (WS / 4), g = e - r * (WS / 4);
    const int iy = ty - po + r, ix = tx - 4 + g * 4;
    const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float4 v = ldx4(xp + (in ? (int64_t)iy * W + ix : 0));
    *reinterpret_cast<float4*>(&s[r][g * 4]) = in ? v : make_float4(fillv, fillv, fillv, fillv);
  }
  float wk[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wk[i] = w[c * K * K + i];
  __syncthreads();
  const int r2 = (threadIdx.x >> 4) * 2;               // first of the thread's two output rows (0, 2, .., 30)
  const int q0 = (threadIdx.x & 15) * 4;               // first of its four output columns
  // 'same' padding only (pad == (K - 1) / 2, so po == pad in both directions): output column q0 + o reads the staged
  // columns q0 + o + j + SH, j = 0 .. K-1 -- inside the 12 floats from q0
  constexpr int SH = 4 - (K - 1) / 2;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < K + 1; ++i) {
    float row[12];
#pragma unroll
    for (int v4 = 0; v4 < 3; ++v4) {
      const float4 t = *reinterpret_cast<const float4*>(&s[r2 + i][q0 + v4 * 4]);
      row[v4 * 4] = t.x; row[v4 * 4 + 1] = t.y; row[v4 * 4 + 2] = t.z; row[v4 * 4 + 3] = t.w;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ki = i - a;                            // kernel row that staged row i is for output row r2 + a
      if (ki < 0 || ki >= K) continue;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float wv = FLIP ? wk[(K - 1 - ki) * K + (K - 1 - j)] : wk[ki * K + j];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[a][o] += wv * row[j + o + SH];
      }
    }
  }
  if constexpr (EPI) {
    uint32_t csum = 0, cnz = 0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oy = ty + r2 + a, ox = tx + q0;
      if (oy >= Ho || ox >= Wo) continue;
      const int64_t off = (int64_t)plane * Ho * Wo + (int64_t)oy * Wo + ox;
      dw_epi_store(ep, c, off, acc[a], min(4, Wo - ox), (Wo & 3) == 0 && (off & 3) == 0, csum, cnz);
    }
    dw_epi_counters(ep, csum, cnz);
    return;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oy = ty + r2 + a, ox = tx + q0;
    if (oy >= Ho) continue;
    float* yp = y + (int64_t)plane * Ho * Wo + (int64_t)oy * Wo + ox;
    if ((Wo & 3) == 0 && ox + 3 < Wo && (reinterpret_cast<uintptr_t>(yp) & 15u) == 0) {
      *reinterpret_cast<float4*>(yp) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (ox + o < Wo) yp[o] = acc[a][o];
    }
  }
}

// gw[c][i][j] += sum over the tile of gy[oy][ox] * x[oy + i - pad][ox + j - pad].  Each thread owns 4 consecutive output
// pixels of one tile row and forms its K*K partial sums in registers (K rows of K+3 staged inputs, as in the stencil); the
// block sum runs over wave shuffles and one LDS round.  (A (tap, pixel-slice) thread mapping spent ~10 instructions per
// multiply-add on index arithmetic: 1.9 TB/s on the 537 MB maps.)
template <int K, typename TX>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const TX* __restrict__ x, const float* __restrict__ border,
                                                       const float* __restrict__ gy, float* __restrict__ gw, int C, int H,
                                                       int W, int Ho, int Wo, int pad, int tiles_x, int ntiles) {
  constexpr int HS = TS + K - 1;
  constexpr int KK = K * K;
  __shared__ float s[HS][HS + 1];
  __shared__ float red[4][KK];
  const int plane = blockIdx.y;
  const int c = plane % C;
  const TX* xp = x + (int64_t)plane * H * W;
  const float* gp = gy + (int64_t)plane * Ho * Wo;
  const float fillv = border ? border[c] : 0.f;
  const bool vec_ok = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & (4 * sizeof(TX) - 1)) == 0;
  const int r = threadIdx.x >> 3, q0 = (threadIdx.x & 7) * 4;
  float acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = 0.f;
  // a workgroup walks several tiles of its plane with the K*K partial sums in registers: the K*K block reductions (6
  // shuffle steps each) and the atomics are paid once per workgroup, not once per tile
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty = (tile / tiles_x) * TS, tx = (tile % tiles_x) * TS;
    __syncthreads();                          // the previous tile's readers are done with s
    stage_halo<K, TX>(s, xp, H, W, ty - pad, tx - pad, pad, fillv, pad, vec_ok);
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    const int oy = ty + r, ox = tx + q0;
    const float* p = gp + (int64_t)oy * Wo + ox;
    if (oy < Ho) {
      if ((Wo & 3) == 0 && ox + 3 < Wo && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
      } else {
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (ox + o < Wo) g[o] = p[o];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) {
      float row[K + 3];
#pragma unroll
      for (int j = 0; j < K + 3; ++j) row[j] = s[r + i][q0 + j];
#pragma unroll
      for (int j = 0; j < K; ++j)
        acc[i * K + j] += (g[0] * row[j] + g[1] * row[j + 1]) + (g[2] * row[j + 2] + g[3] * row[j + 3]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    const float v = s2f_wave_sum_lane63(acc[t]);
    if (lane == 63) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < KK) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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

template <typename TX>
void launch_stencil_epi(int K, dim3 grid, hipStream_t s, const TX* x, const float* w, const float* border, int C, int H, int W, int Ho,
                        int Wo, int pad, int tiles_x, DwEpi ep) {
  if (K == 3 && (W & 3) == 0 && W >= 2 * WT && H >= HT && pad == 1 && (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(TX) - 1)) == 0) {
    const int wx = (Wo + WT - 1) / WT, wy = (Ho + HT - 1) / HT;
    hipLaunchKernelGGL((dw_stencil_wide_kernel<3, false, TX, true>), dim3(wx * wy, grid.y), dim3(256), 0, s, x, w, border,
                       (float*)nullptr, C, H, W, Ho, Wo, pad, wx, ep);
    return;
  }
  if (K == 3)
    hipLaunchKernelGGL((dw_stencil_kernel<3, false, TX, true>), grid, dim3(256), 0, s, x, w, border, (float*)nullptr, C, H, W, Ho, Wo,
                       pad, tiles_x, ep);
  else if (K == 5)
    hipLaunchKernelGGL((dw_stencil_kernel<5, false, TX, true>), grid, dim3(256), 0, s, x, w, border, (float*)nullptr, C, H, W, Ho, Wo,
                       pad, tiles_x, ep);
  else
    hipLaunchKernelGGL((dw_stencil_kernel<7, false, TX, true>), grid, dim3(256), 0, s, x, w, border, (float*)nullptr, C, H, W, Ho, Wo,
                       pad, tiles_x, ep);
}

template <bool FLIP, typename TX>
void launch_stencil(int K, dim3 grid, hipStream_t s, const TX* x, const float* w, const float* border, float* y, int C,
                    int H, int W, int Ho, int Wo, int pad, int tiles_x) {
  // 3x3 on wide maps with 'same' padding: 64 x 32 tiles, vector staging, 2 x 4 outputs per thread.  Measured
  // (tools/probe_dw.py, us, 32x32-tile form -> wide form): 3x3 [8,256,256,256] 267 -> 222, [8,256,128,128] 70 -> 53,
  // [8,256,64,64] 17 -> 18; 7x7 [8,64,256,256] 109 -> 139 (49 multiply-adds per output: that stencil is VALU-bound and the
  // 2 x 4 block only adds register pressure), so the wide form is used for K = 3, W >= 128 only.
  static const bool wide_on = !getenv("S2F_DW_NO_WIDE");          // A/B switch for tools/probe_dw.py
  if (wide_on && K == 3 && (W & 3) == 0 && W >= 2 * WT && H >= HT && pad == 1 && (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(TX) - 1)) == 0) {
    const int wx = (Wo + WT - 1) / WT, wy = (Ho + HT - 1) / HT;
    hipLaunchKernelGGL((dw_stencil_wide_kernel<3, FLIP, TX>), dim3(wx * wy, grid.y), dim3(256), 0, s, x, w, border, y, C, H, W, Ho,
                       Wo, pad, wx, DwEpi{});
    return;
  }
  if (K == 3)
    hipLaunchKernelGGL((dw_stencil_kernel<3, FLIP, TX>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x, DwEpi{});
  else if (K == 5)
    hipLaunchKernelGGL((dw_stencil_kernel<5, FLIP, TX>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x, DwEpi{});
  else
    hipLaunchKernelGGL((dw_stencil_kernel<7, FLIP, TX>), grid, dim3(256), 0, s, x, w, border, y, C, H, W, Ho, Wo, pad, tiles_x, DwEpi{});
}

}  // namespace

extern "C" int s2f_dwconv_fwd(const void* x, const float* w, const float* border, float* y, int N, int C, int H, int W,
                              int K, int pad, int x_bf16, void* stream) {
  S2F_REQUIRE(x && w && y, S2F_EINVAL, "s2f_dwconv_fwd: null pointer");
  int Ho, Wo;
  int rc = check("s2f_dwconv_fwd", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  const int tiles_x = (Wo + TS - 1) / TS, tiles_y = (Ho + TS - 1) / TS;
  if (x_bf16)
    launch_stencil<false>(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, reinterpret_cast<const unsigned short*>(x), w,
                          border, y, C, H, W, Ho, Wo, pad, tiles_x);
  else
    launch_stencil<false>(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, reinterpret_cast<const float*>(x), w, border,
                          y, C, H, W, Ho, Wo, pad, tiles_x);
  return s2f_check_launch("s2f_dwconv_fwd");
}

extern "C" int s2f_dwconv_bn_lif_fwd(const void* x, const float* w, const float* border, const float* running_mean,
                                     const float* running_var, const float* gamma, const float* beta, float eps, float* u_out,
                                     void* y_bf16, uint64_t* stats, int N, int C, int H, int W, int K, int pad, int x_bf16, float vth,
                                     int D, void* stream) {
  S2F_REQUIRE(x && w && running_mean && running_var && gamma && beta && (u_out || y_bf16), S2F_EINVAL,
              "s2f_dwconv_bn_lif_fwd: null pointer / neither output requested");
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_dwconv_bn_lif_fwd: bf16 spikes need D a power of two <= 128");
  int Ho, Wo;
  int rc = check("s2f_dwconv_bn_lif_fwd", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  const int tiles_x = (Wo + TS - 1) / TS, tiles_y = (Ho + TS - 1) / TS;
  DwEpi ep{running_mean, running_var, gamma, beta, u_out, reinterpret_cast<unsigned short*>(y_bf16),
           reinterpret_cast<unsigned long long*>(stats), eps, vth, (float)D, 1.0f / (float)D};
  if (x_bf16)
    launch_stencil_epi(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, reinterpret_cast<const unsigned short*>(x), w, border, C,
                       H, W, Ho, Wo, pad, tiles_x, ep);
  else
    launch_stencil_epi(K, dim3(tiles_x * tiles_y, N * C), (hipStream_t)stream, reinterpret_cast<const float*>(x), w, border, C, H, W,
                       Ho, Wo, pad, tiles_x, ep);
  return s2f_check_launch("s2f_dwconv_bn_lif_fwd");
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

extern "C" int s2f_dwconv_bwd_weight(const void* x_in, const float* border, const float* gy, float* gw, int N, int C, int H,
                                     int W, int K, int pad, int accumulate, int x_bf16, void* stream) {
  const float* x = reinterpret_cast<const float*>(x_in);
  const unsigned short* xh = reinterpret_cast<const unsigned short*>(x_in);
  S2F_REQUIRE(x && gy && gw, S2F_EINVAL, "s2f_dwconv_bwd_weight: null pointer");
  int Ho, Wo;
  int rc = check("s2f_dwconv_bwd_weight", N, C, H, W, K, pad, Ho, Wo);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && s2f_zero_async(gw, sizeof(float) * (size_t)C * K * K, s) != S2F_OK)
    return s2f_check_launch("s2f_dwconv_bwd_weight memset");
  const int tiles_x = (Wo + TS - 1) / TS, tiles_y = (Ho + TS - 1) / TS;
  const int ntiles = tiles_x * tiles_y;
  // workgroups per plane: enough for >= 2048 in flight, at most one per tile
  // enough workgroups for ~32 per CU: a workgroup that walks several tiles pays a global round trip per tile (the block
  // reduction at the end is cheap since the wave sums run on DPP)
  int per_plane = (8192 + N * C - 1) / (N * C);
  if (per_plane > ntiles) per_plane = ntiles;
  if (per_plane < 1) per_plane = 1;
  const dim3 grid(per_plane, N * C);
#define S2F_WG(KV)                                                                                                          \
  do {                                                                                                                      \
    if (x_bf16)                                                                                                             \
      hipLaunchKernelGGL((dw_wgrad_kernel<KV, unsigned short>), grid, dim3(256), 0, s, xh, border, gy, gw, C, H, W, Ho, Wo, \
                         pad, tiles_x, ntiles);                                                                             \
    else                                                                                                                    \
      hipLaunchKernelGGL((dw_wgrad_kernel<KV, float>), grid, dim3(256), 0, s, x, border, gy, gw, C, H, W, Ho, Wo, pad,      \
                         tiles_x, ntiles);                                                                                  \
  } while (0)
  if (K == 3)
    S2F_WG(3);
  else if (K == 5)
    S2F_WG(5);
  else
    S2F_WG(7);
#undef S2F_WG
  return s2f_check_launch("s2f_dwconv_bwd_weight");
}
