// Quantised integrate-and-fire neuron (Q_IFNode + quant STE) for gfx950.
//
// Reference semantics: Qtrick_architecture/clock_driven/neuron.py:166-197 (forward), :459-460 (charge), :133-153
// (soft reset); surrogate.py:522-538 (round(clamp(.,0,D)) forward, in-range straight-through backward).
//
// HBM-bound elementwise kernels.  One wavefront owns a 256-element tile (lane l holds elements 4l..4l+3 as one
// 16-byte access, 1 KiB per wave instruction); the in-range bit of component j is collected with a wave ballot, so
// the backward pass reads 1 bit/element instead of the fp32 membrane the reference saves.  Firing statistics
// (sum of counts, number of non-zero counts) are reduced per wave with popcount / DPP adds, one atomic pair per block.
#include <stdarg.h>

#include "s2f_common.h"

static thread_local char g_err[512] = "";
void s2f_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* s2f_last_error(void) { return g_err; }
extern "C" int s2f_version(void) { return S2F_ABI_VERSION; }

static thread_local S2fTiming g_timing = {nullptr, nullptr};
S2fTiming* s2f_timing_tls() { return &g_timing; }
extern "C" void* s2f_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    s2f_set_error("s2f_event_create: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return e;
}
extern "C" void s2f_event_destroy(void* event) {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
}
extern "C" int s2f_time_next_call(void* start_event, void* stop_event) {
  g_timing.start = (hipEvent_t)start_event;
  g_timing.stop = (hipEvent_t)stop_event;
  return S2F_OK;
}
extern "C" int s2f_event_elapsed_us(void* start_event, void* stop_event, double* microseconds) {
  S2F_REQUIRE(start_event && stop_event && microseconds, S2F_EINVAL, "s2f_event_elapsed_us: null argument");
  float ms = 0.f;
  hipError_t e = hipEventElapsedTime(&ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
  S2F_REQUIRE(e == hipSuccess, S2F_ELAUNCH, "s2f_event_elapsed_us: %s", hipGetErrorString(e));
  *microseconds = (double)ms * 1e3;
  return S2F_OK;
}
extern "C" int64_t s2f_lif_mask_words(int64_t n) { return ((n + 255) >> 8) * 4; }

__global__ void s2f_zero_kernel(float* __restrict__ p, int64_t n) {
  const bool vec = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / S2F_WAVE;

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct Tile4 {
  float a[4];
};

__device__ __forceinline__ Tile4 load4(const float* p, int64_t base, int64_t n, float fill) {
  Tile4 t;
  if (base + 3 < n) {
    float4 v = *reinterpret_cast<const float4*>(p + base);
    t.a[0] = v.x; t.a[1] = v.y; t.a[2] = v.z; t.a[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) t.a[j] = (base + j < n) ? p[base + j] : fill;
  }
  return t;
}

__device__ __forceinline__ void store4(float* p, int64_t base, int64_t n, const Tile4& t) {
  if (base + 3 < n) {
    *reinterpret_cast<float4*>(p + base) = make_float4(t.a[0], t.a[1], t.a[2], t.a[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (base + j < n) p[base + j] = t.a[j];
  }
}

__device__ __forceinline__ void store4_bf16(unsigned short* p, int64_t base, int64_t n, const Tile4& t) {
  if (base + 3 < n) {
    *reinterpret_cast<uint2*>(p + base) = s2f_spikes_to_bf16x4(t.a[0], t.a[1], t.a[2], t.a[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (base + j < n) p[base + j] = (unsigned short)(__float_as_uint(t.a[j]) >> 16);
  }
}

__device__ __forceinline__ void write_mask(uint64_t* mask, int64_t tile, int lane, const bool inr[4]) {
  // wave ballot of each component; word j bit l <-> element 256*tile + 4*l + j
  uint64_t b0 = __ballot(inr[0]), b1 = __ballot(inr[1]), b2 = __ballot(inr[2]), b3 = __ballot(inr[3]);
  if (mask != nullptr && lane < 4) {
    uint64_t w = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
    mask[tile * 4 + lane] = w;
  }
}

// ------------------------------------------------------------------ single step
// YB: the spikes are written as bf16 (exact, see s2f_spikes_to_bf16x4) -- y then points to uint16 storage
template <bool HAS_V, bool YB>
__global__ __launch_bounds__(kBlock) void lif_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v_in,
                                                         float* __restrict__ y, float* __restrict__ v_out,
                                                         uint64_t* __restrict__ mask, uint8_t* __restrict__ cnt,
                                                         unsigned long long* __restrict__ stats, int64_t n, float vth,
                                                         float Df) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  uint32_t csum = 0, cnz = 0;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 xv = load4(x, base, n, -1.0f);
    Tile4 vv;
    if (HAS_V) vv = load4(v_in, base, n, 0.0f);
    Tile4 yv, vo;
    bool inr[4];
    uint32_t c4 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float h = HAS_V ? (vv.a[j] + xv.a[j]) : xv.a[j];
      float s;
      s2f_lif_update(h, Df, 1.0f, vth, s, yv.a[j], vo.a[j], inr[j]);
      yv.a[j] = s / Df;
      inr[j] = inr[j] && (base + j < n);
      uint32_t si = (uint32_t)s;
      if (base + j < n) {
        csum += si;
        cnz += (si != 0);
      }
      c4 |= si << (8 * j);
    }
    if (YB)
      store4_bf16(reinterpret_cast<unsigned short*>(y), base, n, yv);
    else
      store4(y, base, n, yv);
    if (v_out != nullptr) store4(v_out, base, n, vo);
    if (cnt != nullptr) {
      if (base + 3 < n) {
        *reinterpret_cast<uint32_t*>(cnt + base) = c4;
      } else {
        for (int j = 0; j < 4; ++j)
          if (base + j < n) cnt[base + j] = (uint8_t)(c4 >> (8 * j));
      }
    }
    write_mask(mask, tile, lane, inr);
  }
  if (stats != nullptr) {
    csum = wave_sum_u32(csum);
    cnz = wave_sum_u32(cnz);
    __shared__ uint32_t red[2 * kWavesPerBlock];
    if (lane == 0) {
      red[(threadIdx.x >> 6) * 2] = csum;
      red[(threadIdx.x >> 6) * 2 + 1] = cnz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long a = 0, b = 0;
      for (int w = 0; w < kWavesPerBlock; ++w) {
        a += red[2 * w];
        b += red[2 * w + 1];
      }
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (a) atomicAdd(&slot[0], a);
      if (b) atomicAdd(&slot[1], b);
    }
  }
}

// gy2 (optional): the gradient of a SECOND consumer of the same spike map, skip (optional): the gradient of a residual branch that
// read the neuron's input itself -- both sums the autograd engine would otherwise form with an add launch of its own
// (ops/neuron.py: the neuron's second handle and its pass-through output); (gy + gy2) / D + skip are the same IEEE operations.
template <bool HAS_GV>
__global__ __launch_bounds__(kBlock) void lif_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ gv,
                                                         const uint64_t* __restrict__ mask, float* __restrict__ gx,
                                                         int64_t n, float vth, float Df, const float* __restrict__ gy2,
                                                         const float* __restrict__ skip) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 g = load4(gy, base, n, 0.0f);
    if (gy2) {
      const Tile4 g2 = load4(gy2, base, n, 0.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j) g.a[j] += g2.a[j];
    }
    Tile4 gvv;
    if (HAS_GV) gvv = load4(gv, base, n, 0.0f);
    Tile4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool m = (mask[tile * 4 + j] >> lane) & 1ull;
      // autograd of  y = s/D ; v' = h - s*vth  with ds/dh = m:  dL/dh = gv + (gy/D - gv*vth) * m
      const float through = g.a[j] / Df;
      if (HAS_GV)
        o.a[j] = m ? (gvv.a[j] + (through - gvv.a[j] * vth)) : gvv.a[j];
      else
        o.a[j] = m ? through : 0.0f;
    }
    if (skip) {
      const Tile4 sk = load4(skip, base, n, 0.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j) o.a[j] += sk.a[j];
    }
    store4(gx, base, n, o);
  }
}

// ------------------------------------------------------------------ two neurons on a biased tensor (decoder keys / values)
// a = x + e[c] ;  y_v = Q_IFNode(a) ;  y_k = Q_IFNode(a + pos[b, c, l])      x: [TB, C, L], tb = t*B + b, pos: [B, C, L]
// = the decoder's value / key neurons applied to  memory + level_embed  and  memory + level_embed + key_pos
// (mmdet/models/dense_heads/maskformer_head.py:535-540; mmcv_spike/transformer.py:626-629, 213-236) without materialising
// the two sums: they are read by nothing but these neurons.  Reset, stateless neurons only (no membrane in / out).
template <bool YB>
__global__ __launch_bounds__(kBlock) void sum2_lif_fwd_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                              const float* __restrict__ pos, float* __restrict__ yk,
                                                              float* __restrict__ yv, uint64_t* __restrict__ mk,
                                                              uint64_t* __restrict__ mv, int64_t n, int C, int L, int B,
                                                              float vth, float Df) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;                  // L % 4 == 0: the four elements share (tb, c)
    bool ik[4] = {false, false, false, false}, iv[4] = {false, false, false, false};
    if (base < n) {
      const int64_t row = base / L;
      const int l = (int)(base - row * L), c = (int)(row % C), b = (int)((row / C) % B);
      const Tile4 xv = load4(x, base, n, 0.f), pv = load4(pos, ((int64_t)b * C + c) * L + l, (int64_t)B * C * L, 0.f);
      const float ec = e[c];
      Tile4 ok, ov;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = xv.a[j] + ec;
        float sp, yy, vn;
        s2f_lif_update(a, Df, 1.0f, vth, sp, yy, vn, iv[j]);
        ov.a[j] = sp / Df;
        s2f_lif_update(a + pv.a[j], Df, 1.0f, vth, sp, yy, vn, ik[j]);
        ok.a[j] = sp / Df;
      }
      if (YB) {
        store4_bf16(reinterpret_cast<unsigned short*>(yv), base, n, ov);
        store4_bf16(reinterpret_cast<unsigned short*>(yk), base, n, ok);
      } else {
        store4(yv, base, n, ov);
        store4(yk, base, n, ok);
      }
    }
    write_mask(mk, tile, lane, ik);
    write_mask(mv, tile, lane, iv);
  }
}

// gx = STE_k(g_k) + STE_v(g_v)   (the gradient with respect to x; the one with respect to e is its per-channel sum)
// gk / gv may be null (a neuron whose output nobody differentiated); gxk? receives STE_k(g_k) alone -- the gradient with respect to
// `pos` is its sum over the T time steps (the decoder's self-attention, whose position term is the learnable query embedding)
// gk2 / gv2 / skip (optional): a second consumer's gradient of either spike map and the gradient of a residual branch on x, summed
// here instead of by the autograd engine (see lif_bwd_kernel)
__global__ __launch_bounds__(kBlock) void sum2_lif_bwd_kernel(const float* __restrict__ gk, const float* __restrict__ gv,
                                                              const uint64_t* __restrict__ mk,
                                                              const uint64_t* __restrict__ mv, float* __restrict__ gx,
                                                              float* __restrict__ gxk, int64_t n, float Df,
                                                              const float* __restrict__ gk2,
                                                              const float* __restrict__ gv2,
                                                              const float* __restrict__ skip) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 a, b, o, ok;
#pragma unroll
    for (int j = 0; j < 4; ++j) a.a[j] = b.a[j] = 0.f;
    if (gk) a = load4(gk, base, n, 0.f);
    if (gv) b = load4(gv, base, n, 0.f);
    if (gk2) {
      const Tile4 t = load4(gk2, base, n, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) a.a[j] = gk ? a.a[j] + t.a[j] : t.a[j];
    }
    if (gv2) {
      const Tile4 t = load4(gv2, base, n, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) b.a[j] = gv ? b.a[j] + t.a[j] : t.a[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool bk = (mk[tile * 4 + j] >> lane) & 1ull, bv = (mv[tile * 4 + j] >> lane) & 1ull;
      ok.a[j] = bk ? a.a[j] / Df : 0.f;
      o.a[j] = (bv ? b.a[j] / Df : 0.f) + ok.a[j];
    }
    if (skip) {
      const Tile4 t = load4(skip, base, n, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) o.a[j] += t.a[j];
    }
    store4(gx, base, n, o);
    if (gxk) store4(gxk, base, n, ok);
  }
}

// ------------------------------------------------------------------ layer scale folded into a BatchNorm affine pair
__global__ void scale_affine_fwd_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ s, float* __restrict__ w, float* __restrict__ b, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    w[c] = gamma[c] * s[c];
    b[c] = beta[c] * s[c];
  }
}

__global__ void scale_affine_bwd_kernel(const float* __restrict__ gw, const float* __restrict__ gb,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ s, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                        float* __restrict__ ds, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float a = gw ? gw[c] : 0.f, bb = gb ? gb[c] : 0.f;
    dgamma[c] = a * s[c];
    dbeta[c] = bb * s[c];
    ds[c] = a * gamma[c] + bb * beta[c];
  }
}

// ------------------------------------------------------------------ T chained steps, membrane in registers
template <bool HAS_V0>
__global__ __launch_bounds__(kBlock) void lif_seq_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v0,
                                                             float* __restrict__ y, float* __restrict__ vT,
                                                             uint64_t* __restrict__ mask,
                                                             unsigned long long* __restrict__ stats, int T, int64_t n,
                                                             float vth, float Df) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  const int64_t mwords = ntiles * 4;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 v;
    if (HAS_V0) {
      v = load4(v0, base, n, 0.0f);
    } else {
      v.a[0] = v.a[1] = v.a[2] = v.a[3] = 0.0f;
    }
    Tile4 xn = load4(x, base, n, -1.0f);
    for (int t = 0; t < T; ++t) {
      Tile4 xv = xn;
      if (t + 1 < T) xn = load4(x + (int64_t)(t + 1) * n, base, n, -1.0f);  // prefetch next step
      Tile4 yv;
      bool inr[4];
      uint32_t csum = 0, cnz = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // first step after a reset: the reference computes 0. + x == x; adding the 0.0f register is identical
        float h = v.a[j] + xv.a[j];
        float s, yy;
        s2f_lif_update(h, Df, 1.0f, vth, s, yy, v.a[j], inr[j]);
        yv.a[j] = s / Df;
        inr[j] = inr[j] && (base + j < n);
        if (base + j < n) {
          csum += (uint32_t)s;
          cnz += ((uint32_t)s != 0);
        }
      }
      store4(y + (int64_t)t * n, base, n, yv);
      write_mask(mask == nullptr ? nullptr : mask + (int64_t)t * mwords, tile, lane, inr);
      if (stats != nullptr) {
        csum = wave_sum_u32(csum);
        cnz = wave_sum_u32(cnz);
        if (lane == 0) {
          unsigned long long* slot = stats + 2 * ((int64_t)t * S2F_STAT_SLOTS + blockIdx.x % S2F_STAT_SLOTS);
          if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
          if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
        }
      }
    }
    if (vT != nullptr) store4(vT, base, n, v);
  }
}

__global__ __launch_bounds__(kBlock) void lif_seq_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ gvT,
                                                             const uint64_t* __restrict__ mask,
                                                             float* __restrict__ gx, float* __restrict__ gv0, int T,
                                                             int64_t n, float vth, float Df) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  const int64_t mwords = ntiles * 4;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 gh;
    if (gvT != nullptr) {
      gh = load4(gvT, base, n, 0.0f);
    } else {
      gh.a[0] = gh.a[1] = gh.a[2] = gh.a[3] = 0.0f;
    }
    for (int t = T - 1; t >= 0; --t) {
      Tile4 g = load4(gy + (int64_t)t * n, base, n, 0.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool m = (mask[(int64_t)t * mwords + tile * 4 + j] >> lane) & 1ull;
        gh.a[j] = m ? (gh.a[j] + (g.a[j] / Df - gh.a[j] * vth)) : gh.a[j];
      }
      store4(gx + (int64_t)t * n, base, n, gh);
    }
    if (gv0 != nullptr) store4(gv0, base, n, gh);
  }
}


// ------------------------------------------------------------------ leaky charge (LIFNode.neuronal_charge, neuron.py:803-814)
// No Spike2Former module instantiates LIFNode (Q_IFNode is a copy of IFNode), but the class is part of the
// neuron file the path imports, with the fork's forward (multi-level quantised firing, soft reset, y = s / D; neuron.py:153, 197):
//     decay_input:      h = v + (x - v) / tau          (the reference's expression order; true fp32 division)
//     not decay_input:  h = v * (1 - 1/tau) + x        (the factor formed in double by Python, rounded to fp32 by the scalar multiply)
// then s = rint(clamp(h, 0, D)), v' = h - s * vth, y = s / D as in lif_fwd_kernel.  Backward (autograd of the same expressions):
//     g_h = gv' + (gy / D - gv' * vth) * m ;   decay_input: gx = g_h / tau, gv = g_h - g_h / tau ;   else: gx = g_h, gv = g_h * c.
template <bool HAS_V, bool YB, bool DECAY_INPUT>
__global__ __launch_bounds__(kBlock) void lif_leaky_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v_in,
                                                               float* __restrict__ y, float* __restrict__ v_out,
                                                               uint64_t* __restrict__ mask, unsigned long long* __restrict__ stats,
                                                               int64_t n, float vth, float Df, float tau, float keep) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  uint32_t csum = 0, cnz = 0;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 xv = load4(x, base, n, -1.0f);
    Tile4 vv;
    if (HAS_V) vv = load4(v_in, base, n, 0.0f);
    Tile4 yv, vo;
    bool inr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float h;
      if (DECAY_INPUT)
        h = HAS_V ? (vv.a[j] + (xv.a[j] - vv.a[j]) / tau) : (xv.a[j] / tau);          // v == 0.: 0. + (x - 0.) / tau
      else
        h = HAS_V ? (vv.a[j] * keep + xv.a[j]) : xv.a[j];                             // v == 0.: 0. * c + x
      float s;
      s2f_lif_update(h, Df, 1.0f, vth, s, yv.a[j], vo.a[j], inr[j]);
      yv.a[j] = s / Df;
      inr[j] = inr[j] && (base + j < n);
      const uint32_t si = (uint32_t)s;
      if (base + j < n) {
        csum += si;
        cnz += (si != 0);
      }
    }
    if (YB)
      store4_bf16(reinterpret_cast<unsigned short*>(y), base, n, yv);
    else
      store4(y, base, n, yv);
    if (v_out != nullptr) store4(v_out, base, n, vo);
    write_mask(mask, tile, lane, inr);
  }
  if (stats != nullptr) {
    csum = wave_sum_u32(csum);
    cnz = wave_sum_u32(cnz);
    if (lane == 0) {
      unsigned long long* slot = stats + 2 * (blockIdx.x % S2F_STAT_SLOTS);
      if (csum) atomicAdd(&slot[0], (unsigned long long)csum);
      if (cnz) atomicAdd(&slot[1], (unsigned long long)cnz);
    }
  }
}

template <bool HAS_GV, bool DECAY_INPUT>
__global__ __launch_bounds__(kBlock) void lif_leaky_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ gv,
                                                               const uint64_t* __restrict__ mask, float* __restrict__ gx,
                                                               float* __restrict__ gv_in, int64_t n, float vth, float Df, float tau,
                                                               float keep) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const int64_t ntiles = (n + 255) >> 8;
  for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
    const int64_t base = tile * 256 + lane * 4;
    Tile4 g = load4(gy, base, n, 0.0f);
    Tile4 gvv;
    if (HAS_GV) gvv = load4(gv, base, n, 0.0f);
    Tile4 ox, ov;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool m = (mask[tile * 4 + j] >> lane) & 1ull;
      const float through = g.a[j] / Df;
      float gh;
      if (HAS_GV)
        gh = m ? (gvv.a[j] + (through - gvv.a[j] * vth)) : gvv.a[j];
      else
        gh = m ? through : 0.0f;
      if (DECAY_INPUT) {
        ox.a[j] = gh / tau;
        ov.a[j] = gh - ox.a[j];
      } else {
        ox.a[j] = gh;
        ov.a[j] = gh * keep;
      }
    }
    store4(gx, base, n, ox);
    if (gv_in != nullptr) store4(gv_in, base, n, ov);
  }
}

inline int grid_for(int64_t n) {
  int64_t tiles = (n + 255) >> 8;
  int64_t blocks = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = 256 * 8;  // 256 CUs x 8 resident 256-thread blocks
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int s2f_lif_fwd(const float* x, const float* v_in, void* y_out, float* v_out, uint64_t* mask, uint8_t* count_u8,
                           uint64_t* stats, int64_t n, float vth, int D, int y_bf16, void* stream) {
  float* y = reinterpret_cast<float*>(y_out);
  if (n == 0) return S2F_OK;  // empty tensors are legal (null pointers included)
  S2F_REQUIRE(x && y, S2F_EINVAL, "s2f_lif_fwd: null x/y");
  S2F_REQUIRE(n >= 0 && D >= 1 && D <= 255, S2F_EINVAL, "s2f_lif_fwd: bad n=%lld or D=%d", (long long)n, D);
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_lif_fwd: bf16 spikes need D a power of two <= 128 (D=%d)", D);
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(y) && s2f_aligned16(v_in) && s2f_aligned16(v_out), S2F_EALIGN,
              "s2f_lif_fwd: x/y/v must be 16-byte aligned");
  S2F_REQUIRE(count_u8 == nullptr || (reinterpret_cast<uintptr_t>(count_u8) & 3u) == 0, S2F_EALIGN,
              "s2f_lif_fwd: count_u8 must be 4-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  auto* st = reinterpret_cast<unsigned long long*>(stats);
#define S2F_LIF_GO(HV, YBV)                                                                                            \
  S2F_LAUNCH(true, true, (lif_fwd_kernel<HV, YBV>), dim3(grid_for(n)), dim3(kBlock), 0, s, x, v_in, y, v_out, mask,     \
             count_u8, st, n, vth, (float)D)
  if (v_in != nullptr) {
    if (y_bf16)
      S2F_LIF_GO(true, true);
    else
      S2F_LIF_GO(true, false);
  } else {
    if (y_bf16)
      S2F_LIF_GO(false, true);
    else
      S2F_LIF_GO(false, false);
  }
#undef S2F_LIF_GO
  return s2f_check_launch("s2f_lif_fwd");
}

extern "C" int s2f_lif_bwd_ports(const float* gy, const float* gy2, const float* gv_out, const uint64_t* mask, const float* skip,
                                 float* gx, int64_t n, float vth, int D, void* stream) {
  if (n == 0) return S2F_OK;  // empty tensors are legal (null pointers included)
  S2F_REQUIRE(gy && mask && gx, S2F_EINVAL, "s2f_lif_bwd: null gy/mask/gx");
  S2F_REQUIRE(n >= 0 && D >= 1 && D <= 255, S2F_EINVAL, "s2f_lif_bwd: bad n or D");
  S2F_REQUIRE(!(gv_out && skip), S2F_EINVAL, "s2f_lif_bwd_ports: a pass-through gradient only for a neuron without a membrane gradient");
  S2F_REQUIRE(s2f_aligned16(gy) && s2f_aligned16(gx) && s2f_aligned16(gv_out) && s2f_aligned16(gy2) && s2f_aligned16(skip), S2F_EALIGN,
              "s2f_lif_bwd: gy/gx/gv must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (gv_out != nullptr)
    S2F_LAUNCH(true, true, lif_bwd_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, s, gy, gv_out, mask, gx, n, vth,
                       (float)D, gy2, skip);
  else
    S2F_LAUNCH(true, true, lif_bwd_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, s, gy, gv_out, mask, gx, n, vth,
                       (float)D, gy2, skip);
  return s2f_check_launch("s2f_lif_bwd");
}

extern "C" int s2f_lif_bwd(const float* gy, const float* gv_out, const uint64_t* mask, float* gx, int64_t n, float vth,
                           int D, void* stream) {
  return s2f_lif_bwd_ports(gy, nullptr, gv_out, mask, nullptr, gx, n, vth, D, stream);
}

extern "C" int s2f_lif_leaky_fwd(const float* x, const float* v_in, void* y_out, float* v_out, uint64_t* mask, uint64_t* stats,
                                 int64_t n, float vth, int D, float tau, int decay_input, int y_bf16, void* stream) {
  float* y = reinterpret_cast<float*>(y_out);
  if (n == 0) return S2F_OK;
  S2F_REQUIRE(x && y, S2F_EINVAL, "s2f_lif_leaky_fwd: null x/y");
  S2F_REQUIRE(n >= 0 && D >= 1 && D <= 255, S2F_EINVAL, "s2f_lif_leaky_fwd: bad n=%lld or D=%d", (long long)n, D);
  S2F_REQUIRE(tau > 1.0f, S2F_EINVAL, "s2f_lif_leaky_fwd: tau must exceed 1 (neuron.py:792), got %g", (double)tau);
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_lif_leaky_fwd: bf16 spikes need D a power of two <= 128 (D=%d)", D);
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(y) && s2f_aligned16(v_in) && s2f_aligned16(v_out), S2F_EALIGN,
              "s2f_lif_leaky_fwd: x/y/v must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  auto* st = reinterpret_cast<unsigned long long*>(stats);
  const float keep = (float)(1.0 - 1.0 / (double)tau);          // Python forms 1. - 1. / tau in double; the scalar multiply rounds it to fp32
#define S2F_LEAKY_GO(HV, YBV, DI)                                                                                                   \
  S2F_LAUNCH(true, true, (lif_leaky_fwd_kernel<HV, YBV, DI>), dim3(grid_for(n)), dim3(kBlock), 0, s, x, v_in, y, v_out, mask, st, n, \
             vth, (float)D, tau, keep)
#define S2F_LEAKY_YB(HV, DI) \
  do {                       \
    if (y_bf16)              \
      S2F_LEAKY_GO(HV, true, DI); \
    else                     \
      S2F_LEAKY_GO(HV, false, DI); \
  } while (0)
  if (v_in != nullptr) {
    if (decay_input) S2F_LEAKY_YB(true, true); else S2F_LEAKY_YB(true, false);
  } else {
    if (decay_input) S2F_LEAKY_YB(false, true); else S2F_LEAKY_YB(false, false);
  }
#undef S2F_LEAKY_YB
#undef S2F_LEAKY_GO
  return s2f_check_launch("s2f_lif_leaky_fwd");
}

extern "C" int s2f_lif_leaky_bwd(const float* gy, const float* gv_out, const uint64_t* mask, float* gx, float* gv_in, int64_t n,
                                 float vth, int D, float tau, int decay_input, void* stream) {
  if (n == 0) return S2F_OK;
  S2F_REQUIRE(gy && mask && gx, S2F_EINVAL, "s2f_lif_leaky_bwd: null gy/mask/gx");
  S2F_REQUIRE(n >= 0 && D >= 1 && D <= 255 && tau > 1.0f, S2F_EINVAL, "s2f_lif_leaky_bwd: bad n, D or tau");
  S2F_REQUIRE(s2f_aligned16(gy) && s2f_aligned16(gx) && s2f_aligned16(gv_out) && s2f_aligned16(gv_in), S2F_EALIGN,
              "s2f_lif_leaky_bwd: gy/gx/gv must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const float keep = (float)(1.0 - 1.0 / (double)tau);
#define S2F_LEAKY_BGO(HG, DI)                                                                                                    \
  S2F_LAUNCH(true, true, (lif_leaky_bwd_kernel<HG, DI>), dim3(grid_for(n)), dim3(kBlock), 0, s, gy, gv_out, mask, gx, gv_in, n, vth, \
             (float)D, tau, keep)
  if (gv_out != nullptr) {
    if (decay_input) S2F_LEAKY_BGO(true, true); else S2F_LEAKY_BGO(true, false);
  } else {
    if (decay_input) S2F_LEAKY_BGO(false, true); else S2F_LEAKY_BGO(false, false);
  }
#undef S2F_LEAKY_BGO
  return s2f_check_launch("s2f_lif_leaky_bwd");
}

extern "C" int s2f_sum2_lif_fwd(const float* x, const float* e, const float* pos, void* y_key_out, void* y_value_out,
                                uint64_t* mask_key, uint64_t* mask_value, int64_t TB, int64_t B, int64_t C, int64_t L,
                                float vth, int D, int y_bf16, void* stream) {
  float* y_key = reinterpret_cast<float*>(y_key_out);
  float* y_value = reinterpret_cast<float*>(y_value_out);
  S2F_REQUIRE(x && e && pos && y_key && y_value && mask_key && mask_value, S2F_EINVAL, "s2f_sum2_lif_fwd: null pointer");
  S2F_REQUIRE(TB > 0 && B > 0 && TB % B == 0 && C > 0 && L > 0 && (L & 3) == 0 && D >= 1 && D <= 255, S2F_EINVAL,
              "s2f_sum2_lif_fwd: bad shape (L must be a multiple of 4, TB a multiple of B)");
  S2F_REQUIRE(s2f_aligned16(x) && s2f_aligned16(pos) && s2f_aligned16(y_key) && s2f_aligned16(y_value), S2F_EALIGN,
              "s2f_sum2_lif_fwd: tensors must be 16-byte aligned");
  S2F_REQUIRE(!y_bf16 || s2f_bf16_spikes_exact(D), S2F_EINVAL, "s2f_sum2_lif_fwd: bf16 spikes need D a power of two <= 128 (D=%d)", D);
  const int64_t n = TB * C * L;
  if (y_bf16)
    S2F_LAUNCH(true, true, sum2_lif_fwd_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, e, pos,
               y_key, y_value, mask_key, mask_value, n, (int)C, (int)L, (int)B, vth, (float)D);
  else
    S2F_LAUNCH(true, true, sum2_lif_fwd_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, e, pos,
               y_key, y_value, mask_key, mask_value, n, (int)C, (int)L, (int)B, vth, (float)D);
  return s2f_check_launch("s2f_sum2_lif_fwd");
}

extern "C" int s2f_sum2_lif_bwd(const float* g_key, const float* g_value, const uint64_t* mask_key,
                                const uint64_t* mask_value, float* gx, int64_t n, int D, void* stream) {
  if (n == 0) return S2F_OK;
  S2F_REQUIRE(g_key && g_value && mask_key && mask_value && gx, S2F_EINVAL, "s2f_sum2_lif_bwd: null pointer");
  S2F_REQUIRE(s2f_aligned16(g_key) && s2f_aligned16(g_value) && s2f_aligned16(gx), S2F_EALIGN,
              "s2f_sum2_lif_bwd: tensors must be 16-byte aligned");
  S2F_LAUNCH(true, true, sum2_lif_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, g_key, g_value,
             mask_key, mask_value, gx, (float*)nullptr, n, (float)D, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
  return s2f_check_launch("s2f_sum2_lif_bwd");
}

extern "C" int s2f_sum2_lif_bwd_ports(const float* g_key, const float* g_key2, const float* g_value, const float* g_value2,
                                      const uint64_t* mask_key, const uint64_t* mask_value, const float* skip, float* gx, float* gx_key,
                                      int64_t n, int D, void* stream) {
  if (n == 0) return S2F_OK;
  S2F_REQUIRE((g_key || g_value || g_key2 || g_value2) && mask_key && mask_value && gx, S2F_EINVAL, "s2f_sum2_lif_bwd_ports: null pointer");
  S2F_REQUIRE(s2f_aligned16(g_key) && s2f_aligned16(g_value) && s2f_aligned16(g_key2) && s2f_aligned16(g_value2) && s2f_aligned16(skip) &&
                  s2f_aligned16(gx) && s2f_aligned16(gx_key),
              S2F_EALIGN, "s2f_sum2_lif_bwd_ports: tensors must be 16-byte aligned");
  S2F_LAUNCH(true, true, sum2_lif_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, g_key, g_value,
             mask_key, mask_value, gx, gx_key, n, (float)D, g_key2, g_value2, skip);
  return s2f_check_launch("s2f_sum2_lif_bwd_ports");
}

extern "C" int s2f_sum2_lif_bwd_ex(const float* g_key, const float* g_value, const uint64_t* mask_key,
                                   const uint64_t* mask_value, float* gx, float* gx_key, int64_t n, int D, void* stream) {
  return s2f_sum2_lif_bwd_ports(g_key, nullptr, g_value, nullptr, mask_key, mask_value, nullptr, gx, gx_key, n, D, stream);
}

extern "C" int s2f_scale_affine_fwd(const float* gamma, const float* beta, const float* s, float* w, float* b, int C,
                                    void* stream) {
  S2F_REQUIRE(gamma && beta && s && w && b && C > 0, S2F_EINVAL, "s2f_scale_affine_fwd: null pointer / empty");
  hipLaunchKernelGGL(scale_affine_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, s, w, b,
                     C);
  return s2f_check_launch("s2f_scale_affine_fwd");
}

extern "C" int s2f_scale_affine_bwd(const float* gw, const float* gb, const float* gamma, const float* beta, const float* s,
                                    float* dgamma, float* dbeta, float* ds, int C, void* stream) {
  S2F_REQUIRE(gamma && beta && s && dgamma && dbeta && ds && C > 0, S2F_EINVAL, "s2f_scale_affine_bwd: null pointer / empty");
  hipLaunchKernelGGL(scale_affine_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gw, gb, gamma, beta, s,
                     dgamma, dbeta, ds, C);
  return s2f_check_launch("s2f_scale_affine_bwd");
}

extern "C" int s2f_lif_seq_fwd(const float* x_seq, const float* v0, float* y_seq, float* vT, uint64_t* mask,
                               uint64_t* stats, int T, int64_t n, float vth, int D, void* stream) {
  if (n == 0) return S2F_OK;  // empty tensors are legal (null pointers included)
  S2F_REQUIRE(x_seq && y_seq, S2F_EINVAL, "s2f_lif_seq_fwd: null x/y");
  S2F_REQUIRE(T >= 1 && n >= 0 && D >= 1 && D <= 255, S2F_EINVAL, "s2f_lif_seq_fwd: bad T/n/D");
  S2F_REQUIRE(T == 1 || (n % 4) == 0, S2F_EINVAL, "s2f_lif_seq_fwd: n must be a multiple of 4 when T > 1");
  S2F_REQUIRE(s2f_aligned16(x_seq) && s2f_aligned16(y_seq) && s2f_aligned16(v0) && s2f_aligned16(vT), S2F_EALIGN,
              "s2f_lif_seq_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  auto* st = reinterpret_cast<unsigned long long*>(stats);
  if (v0 != nullptr)
    hipLaunchKernelGGL(lif_seq_fwd_kernel<true>, dim3(grid_for(n)), dim3(kBlock), 0, s, x_seq, v0, y_seq, vT, mask, st,
                       T, n, vth, (float)D);
  else
    hipLaunchKernelGGL(lif_seq_fwd_kernel<false>, dim3(grid_for(n)), dim3(kBlock), 0, s, x_seq, v0, y_seq, vT, mask, st,
                       T, n, vth, (float)D);
  return s2f_check_launch("s2f_lif_seq_fwd");
}

extern "C" int s2f_lif_seq_bwd(const float* gy_seq, const float* gvT, const uint64_t* mask, float* gx_seq, float* gv0,
                               int T, int64_t n, float vth, int D, void* stream) {
  if (n == 0) return S2F_OK;  // empty tensors are legal (null pointers included)
  S2F_REQUIRE(gy_seq && mask && gx_seq, S2F_EINVAL, "s2f_lif_seq_bwd: null gy/mask/gx");
  S2F_REQUIRE(T >= 1 && n >= 0 && D >= 1 && D <= 255, S2F_EINVAL, "s2f_lif_seq_bwd: bad T/n/D");
  S2F_REQUIRE(T == 1 || (n % 4) == 0, S2F_EINVAL, "s2f_lif_seq_bwd: n must be a multiple of 4 when T > 1");
  S2F_REQUIRE(s2f_aligned16(gy_seq) && s2f_aligned16(gx_seq) && s2f_aligned16(gvT) && s2f_aligned16(gv0), S2F_EALIGN,
              "s2f_lif_seq_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(lif_seq_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, gy_seq, gvT, mask,
                     gx_seq, gv0, T, n, vth, (float)D);
  return s2f_check_launch("s2f_lif_seq_bwd");
}
